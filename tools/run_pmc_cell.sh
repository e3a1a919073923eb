#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${RUN_TAG:-pmc_cell3}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/time_ml.py --reps 3 $TIME_ARGS"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o prof -- $B > $O/p$i.log 2>&1
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
