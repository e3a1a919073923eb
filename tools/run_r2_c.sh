# round 2, GPU session C: LDS slots by bank window -- parity, timings, PMC
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_samplers.py -m gpu -q --tb=short -rf > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 900 python tools/exp_variants.py $O/variants.jsonl tools/variants_c.json > $O/variants.log 2>&1
tail -3 $O/variants.log
cd /tmp && export TMPDIR=/tmp
P=$O/pmc; mkdir -p $P
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streaming-leg --no-narrow-leg --no-batch-leg"
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $P/$name -o p -- $CMD > $P/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL
cd $P
python - <<'PY' > summary.txt
import csv, glob, collections, os
for d in ("sq1","sq2","sq3"):
    fs = glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no csv; log tail:", open(d+".log").read()[-400:]); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_gfstack_dma" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d, k, "n=%d mean=%.6g" % (len(v), sum(v)/len(v)))
    for f in glob.glob(d+"/**/*.csv", recursive=True):
        os.remove(f)
PY
cat summary.txt
