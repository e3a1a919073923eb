# sustained clock of the loader/consumer kernel on the original and on the float-rounded library:
# GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / dispatch duration, per dispatch of the bench's
# main leg (f64 kernel, original values) and of its float-storage leg (float kernel, then the f64
# kernel on the rounded values)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clockpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o bench -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-streaming-leg --no-batch-leg --no-narrow-leg --variant-legs fp32 > $O/run.log 2>&1
python - <<PY
import glob, sqlite3, json
db = sqlite3.connect(glob.glob("$O/p/**/*results.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = list(db.execute("select kernel_name, value, start, end from counters_collection where counter_name='GRBM_GUI_ACTIVE' and kernel_name like '%k_gfstack_ws%' order by start"))
out = []
for k, v, s, e in rows:
    dur = (e - s) * 1e-9
    out.append((k.split('(')[0].replace('void ', '').replace('beatamd::', ''), dur * 1e3, v / 8.0 / dur / 1e9))
# phases in launch order
phase, last = [], None
for k, ms, ghz in out:
    if k != last:
        phase.append([k, []]); last = k
    phase[-1][1].append((ms, ghz))
res = []
for k, l in phase:
    l2 = l[2:] if len(l) > 4 else l
    res.append(dict(kernel=k, dispatches=len(l), avg_ms=sum(x[0] for x in l2) / len(l2), avg_GHz=sum(x[1] for x in l2) / len(l2)))
print(json.dumps(res, indent=1))
json.dump(res, open("$O/clock_summary.json", "w"), indent=1)
PY
find $O -name "*.db" -size +2M -delete
tail -2 $O/run.log | cut -c1-200
