#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-pmc_cell2}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TD\|TCP\|TCC\|SQ\|GRBM\)_[A-Z0-9_a-z]*" | sort -u > $O/counters.txt
wc -l $O/counters.txt
B="python $R/tools/time_ml.py --reps 3 $TIME_ARGS"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o prof -- $B > $O/p$i.log 2>&1
  grep -i "error\|invalid\|not" $O/p$i.log | head -3
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
json.dump(out, open("$O/counters.json", "w"), indent=1)
PY
find $O -name "*.db" -size +2M -delete
