# SQ / LDS counters of the stacking kernel (separate --pmc passes, kernel trace only)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-pmc_sq}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-streaming-leg --no-narrow-leg --no-batch-leg $BENCH_ARGS"
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o bench -- $B > $O/p$i.log 2>&1
done
python - <<PY
import glob, sqlite3, json
out = {}
for db in sorted(glob.glob("$O/p*/**/*results.db", recursive=True)):
    d = sqlite3.connect(db)
    try:
        rows = list(d.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                              "where kernel_name like '%k_gfstack%' group by kernel_name, counter_name"))
    except Exception as e:
        print(db, e); continue
    for k, c, v, n in rows:
        out.setdefault(k.replace("void ", "")[:60], {})[c] = v
print(json.dumps(out, indent=1))
json.dump(out, open("$O/sq_counters.json", "w"), indent=1)
PY
find $O -name "*.db" -size +2M -delete
