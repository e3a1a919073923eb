# rocprofv3 evidence for bench.py (kernel stats + HBM PMC passes); outputs under gpurun_out/prof
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/stats -o bench -- $CMD > $R/gpurun_out/prof/stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/fetch -o bench -- $CMD > $R/gpurun_out/prof/fetch_run.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof/write -o bench -- $CMD > $R/gpurun_out/prof/write_run.log 2>&1
BEATAMD_GF_KERNEL=0 BEATAMD_GF_ORDER=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof/fetch_order0 -o bench -- $CMD > $R/gpurun_out/prof/fetch_order0_run.log 2>&1
cd $R/gpurun_out/prof
find . -type f | head -50
du -sh .
# keep the merged output small: drop the big per-dispatch traces of the library fill, keep gfstack rows
python - <<'PY'
import csv, glob, os
for d in ("fetch", "write", "fetch_order0"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if "k_gfstack" in r.get("Kernel_Name", "")]
        out = d + "_gfstack_counters.csv"
        if keep:
            w = csv.DictWriter(open(out, "w"), fieldnames=list(keep[0].keys()))
            w.writeheader(); w.writerows(keep)
        print(d, len(rows), "rows ->", len(keep), "gfstack rows")
        os.remove(f)
for f in glob.glob("**/*kernel_trace.csv", recursive=True):
    if os.path.getsize(f) > 4e6:
        os.remove(f)
PY
tail -3 $R/gpurun_out/prof/stats_run.log
