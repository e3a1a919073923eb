# PMC counters of the geometry-mode kernels (separate passes; no trace domains besides kernel-trace)
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/geopmc; mkdir -p $R/gpurun_out/geopmc
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/geopmc/p$i -o geo -- python $R/tools/geo_app.py 1024 60 > $R/gpurun_out/geopmc/run$i.log 2>&1
done
cd $R/gpurun_out/geopmc
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "geom_los" in k or "quadform_small" in k or "k_accept" in k or "draw_propose" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
rm -rf p*/
