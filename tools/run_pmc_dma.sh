# PMC diagnosis of k_gfstack_dma (bench, 512 chains, ds_read_b64 variant)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
export BEATAMD_GS_DMA=${BEATAMD_GS_DMA:-2}
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --chains 512"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$name -o p -- $CMD > $R/gpurun_out/pmc/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL
run tcc1 TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
run grbm GRBM_GUI_ACTIVE
cd $R/gpurun_out/pmc
python - <<'PY'
import csv, glob, collections
for d in ("sq1","sq2","sq3","tcc1","fetch","grbm"):
    fs = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv; log tail:", open(d+".log").read()[-400:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_gfstack_dma" in r["Kernel_Name"] or "k_gfstack_shared" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d, k, "n=%d mean=%.5g" % (len(v), sum(v)/len(v)))
    for f in glob.glob(d+"/**/*.csv", recursive=True):
        import os; os.remove(f)
PY
