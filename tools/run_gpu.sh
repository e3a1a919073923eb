#!/bin/bash
# ONE entry point for everything that runs on the GPU box (gpurun -- 'bash tools/run_gpu.sh <mode> ...');
# results land under gpurun_out/<tag>, summaries worth keeping are copied to profiles/ by hand.
#
#   round [profile-tag]             full `-m gpu` suite + smoke() + default bench.py (+ `profile <tag>` when given)
#   profile <tag>                   the round's rocprofv3 evidence: kernel stats of the default bench command and, with
#                                   separate FETCH_SIZE / WRITE_SIZE passes, of the nn / multilinear / Toeplitz runs and the
#                                   geometry stage -> gpurun_out/prof_<tag>/out/<tag>_* (tools/summarize_rocpd2.py)
#   profile6 <tag> a|b|c|d          round 6's evidence in four parts (see the mode)
#   stats <tag> <command...>        rocprofv3 --kernel-trace --stats of any command, head of the kernel table
#   pmc <tag> <kernel-like> <command> <counter set> [<counter set> ...]
#                                   one rocprofv3 --pmc pass per counter set (kernel trace only: gpurun refuses --pmc
#                                   together with other trace domains), averages per kernel name matching <kernel-like>
#                                   -> gpurun_out/<tag>/counters.json
#   clock <tag> <kernel-like> <command>   GRBM_GUI_ACTIVE / duration per dispatch: the sustained clock under a kernel
#   ab <tag> <variants.json> <lib.so> [<lib.so> ...]   tools/exp_variants.py under several builds of the library, alternating
#   torchrun1                       the driver's multi-rank launch line with one rank and the RCCL path forced on
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mode=$1; shift
mkdir -p $R/gpurun_out
export TMPDIR=/tmp

summarize_counters() {   # <dir> <kernel-like>
python - "$1" "$2" <<'PY'
import glob, json, sqlite3, sys
O, like = sys.argv[1], sys.argv[2]
out = {}
for db in sorted(glob.glob(O + "/p*/**/*results.db", recursive=True)):
    d = sqlite3.connect(db)
    try:
        rows = list(d.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                              "where kernel_name like ? group by kernel_name, counter_name", ("%" + like + "%",)))
    except Exception as e:
        print(db, e)
        continue
    for k, c, v, n in rows:
        out.setdefault(k.replace("void ", "")[:70], {})[c] = v
print(json.dumps(out, indent=1))
json.dump(out, open(O + "/counters.json", "w"), indent=1)
PY
}

case $mode in
round)
  cd $R
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/pytest_round.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || tail -5 gpurun_out/bench_default.err
  python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value %.0f ms/step %.3f' % (d['value'], d['ms_per_step']), d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])"
  [ -n "$1" ] && bash $0 profile $1 > gpurun_out/profile.log 2>&1 && tail -3 gpurun_out/profile.log
  ;;
profile)
  tag=$1; O=$R/gpurun_out/prof_$tag
  rm -rf $O; mkdir -p $O; cd /tmp
  B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-variant-legs --no-narrow-leg --no-batch-leg"
  prof() { t=$1; shift; mkdir -p $O/$t
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/$t/stats -o bench -- "$@" > $O/$t/stats_run.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$t/fetch -o bench -- "$@" > $O/$t/fetch_run.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$t/write -o bench -- "$@" > $O/$t/write_run.log 2>&1
  }
  prof c512_nn $B
  prof c512_ml $B --interp multilinear --no-streaming-leg
  prof c512_toeplitz $B --covariance toeplitz --no-streaming-leg
  mkdir -p $O/default $O/geometry
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/default/stats -o bench -- python $R/bench.py --no-cpu-baseline > $O/default/stats_run.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/geometry/stats -o bench -- python $R/tools/geo_app.py 1024 200 > $O/geometry/stats_run.log 2>&1
  S="python $R/tools/summarize_rocpd2.py"
  $S $O/c512_nn $O/out ${tag}_bench_c512_nn k_gfstack_ws "k_gfstack<0" k_fast_sweep k_accept > $O/sum_c512_nn.log 2>&1
  $S $O/c512_ml $O/out ${tag}_bench_c512_ml k_gfstack_mlr k_gm_tables > $O/sum_c512_ml.log 2>&1
  $S $O/c512_toeplitz $O/out ${tag}_bench_c512_toeplitz "k_quadform<128>" k_gfstack_ws > $O/sum_c512_toeplitz.log 2>&1
  $S $O/default $O/out ${tag}_bench_default "k_quadform<128>" k_gfstack_mlr "k_gemm_f64<0>" k_gfstack_ws > $O/sum_default.log 2>&1
  $S $O/geometry $O/out ${tag}_geometry_c1024 k_geom_los k_quadform_small k_draw_propose k_accept > $O/sum_geometry.log 2>&1
  grep -h "^{\"metric" $O/*/stats_run.log > $O/out/${tag}_bench_lines_under_profiler.jsonl
  tail -3 $O/default/stats_run.log $O/geometry/stats_run.log | cut -c1-300
  cat $O/sum_*.log | grep -E "kernel\"|avg_us|corrected|write_bytes|no dispatch" | cut -c1-160
  find $O -name "*.db" -size +2M -delete; find $O -name "*.csv" -size +2M -delete
  du -sh $O
  ;;
profile5)
  # round 5: kernel stats + separate FETCH_SIZE / WRITE_SIZE passes of the headline run, the multilinear run and the two
  # runs on the tutorial-grid library (row passes); kernel stats of the default command and of the configs[3] leg
  tag=$1; O=$R/gpurun_out/prof_$tag
  rm -rf $O; mkdir -p $O; cd /tmp
  B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-variant-legs --no-narrow-leg --no-batch-leg"
  G="--samples 512 --ndurations 17 --nstarttimes 41 --duration-min 0 --duration-sampling 0.25 --no-streaming-leg"
  prof() { t=$1; shift; mkdir -p $O/$t
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/$t/stats -o bench -- "$@" > $O/$t/stats_run.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$t/fetch -o bench -- "$@" > $O/$t/fetch_run.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$t/write -o bench -- "$@" > $O/$t/write_run.log 2>&1
  }
  prof c512_nn $B
  prof c512_ml $B --interp multilinear --no-streaming-leg
  prof grid_nn $B $G
  prof grid_ml $B $G --interp multilinear
  prof grid_nn_c2048 $B $G --chains 2048
  prof grid_ml_c2048 $B $G --chains 2048 --interp multilinear
  mkdir -p $O/default $O/config4
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/default/stats -o bench -- python $R/bench.py --no-cpu-baseline > $O/default/stats_run.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/config4/stats -o bench -- python $R/bench.py --no-cpu-baseline --variant-legs config4 --no-streaming-leg --no-batch-leg --no-narrow-leg --steps 10 > $O/config4/stats_run.log 2>&1
  S="python $R/tools/summarize_rocpd2.py"
  $S $O/c512_nn $O/out ${tag}_bench_c512_nn k_gfstack_ws "k_gfstack<0" k_ws_tables k_fast_sweep > $O/sum_c512_nn.log 2>&1
  $S $O/c512_ml $O/out ${tag}_bench_c512_ml k_gfstack_runs k_gm_tables > $O/sum_c512_ml.log 2>&1
  $S $O/grid_nn $O/out ${tag}_grid_c512_nn k_gfstack_ws k_ws_tables > $O/sum_grid_nn.log 2>&1
  $S $O/grid_ml $O/out ${tag}_grid_c512_ml k_gfstack_runs k_gm_tables > $O/sum_grid_ml.log 2>&1
  $S $O/grid_nn_c2048 $O/out ${tag}_grid_c2048_nn k_gfstack_ws > $O/sum_grid_nn2048.log 2>&1
  $S $O/grid_ml_c2048 $O/out ${tag}_grid_c2048_ml k_gfstack_runs > $O/sum_grid_ml2048.log 2>&1
  $S $O/default $O/out ${tag}_bench_default "k_quadform<128>" k_gfstack_runs "k_gemm_f64<0>" k_gfstack_ws > $O/sum_default.log 2>&1
  $S $O/config4 $O/out ${tag}_config4 k_gfstack_runs k_gfstack_ws k_gm_tables k_ws_tables "k_quadform<128>" > $O/sum_config4.log 2>&1
  grep -h "^{\"metric" $O/*/stats_run.log > $O/out/${tag}_bench_lines_under_profiler.jsonl
  tail -2 $O/default/stats_run.log | cut -c1-300
  cat $O/sum_*.log | grep -E "kernel\"|avg_us|corrected|write_bytes|no dispatch" | cut -c1-160
  find $O -name "*.db" -size +2M -delete; find $O -name "*.csv" -size +2M -delete
  du -sh $O
  ;;
profile6)
  # round 6: kernel stats + separate FETCH_SIZE / WRITE_SIZE passes, in PARTS that each fit a gpurun call (the whole set once
  # overran the call's hour and nothing came back): profile6 <tag> a|b|c|d
  #   a  headline run, multilinear run, configs[3] at 120 samples (tools/time_config4.py = the bench leg's problem)
  #   b  the tutorial-grid runs (512 and 2048 chains, nn and multilinear)
  #   c  configs[3] at 4096 samples
  #   d  kernel stats of the default command, then the default bench.py line itself (reads the summaries of a-c from
  #      profiles/: copy them there first)
  # -> gpurun_out/prof_<tag>/out
  tag=$1; part=$2; O=$R/gpurun_out/prof_$tag
  mkdir -p $O/out; cd /tmp
  B="python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-variant-legs --no-narrow-leg --no-batch-leg"
  G="--samples 512 --ndurations 17 --nstarttimes 41 --duration-min 0 --duration-sampling 0.25 --no-streaming-leg"
  prof() { t=$1; shift; rm -rf $O/$t; mkdir -p $O/$t
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/$t/stats -o bench -- "$@" > $O/$t/stats_run.log 2>&1
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/$t/fetch -o bench -- "$@" > $O/$t/fetch_run.log 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/$t/write -o bench -- "$@" > $O/$t/write_run.log 2>&1
  }
  S="python $R/tools/summarize_rocpd2.py"
  c4() { N=$1
    prof c4_${N}_ml python $R/tools/time_config4.py $N multilinear
    $S $O/c4_${N}_ml $O/out ${tag}_config4_N${N}_c512_ml k_gfstack_runs k_gm_tables_w k_gc_order k_geo_stack > $O/sum_c4_${N}_ml.log 2>&1
    prof c4_${N}_nn python $R/tools/time_config4.py $N nearest_neighbor
    $S $O/c4_${N}_nn $O/out ${tag}_config4_N${N}_c512_nn k_gfstack_ws k_ws_tables k_geo_stack > $O/sum_c4_${N}_nn.log 2>&1
  }
  case $part in
  a)
    prof c512_nn $B
    $S $O/c512_nn $O/out ${tag}_bench_c512_nn k_gfstack_ws "k_gfstack<0" k_ws_tables k_fast_sweep > $O/sum_c512_nn.log 2>&1
    prof c512_ml $B --interp multilinear --no-streaming-leg
    $S $O/c512_ml $O/out ${tag}_bench_c512_ml k_gfstack_runs k_gm_tables_w k_gc_order > $O/sum_c512_ml.log 2>&1
    c4 120
    ;;
  b)
    prof grid_nn $B $G
    $S $O/grid_nn $O/out ${tag}_grid_c512_nn k_gfstack_ws k_ws_tables > $O/sum_grid_nn.log 2>&1
    prof grid_ml $B $G --interp multilinear
    $S $O/grid_ml $O/out ${tag}_grid_c512_ml k_gfstack_runs k_gm_tables_w > $O/sum_grid_ml.log 2>&1
    prof grid_nn_c2048 $B $G --chains 2048
    $S $O/grid_nn_c2048 $O/out ${tag}_grid_c2048_nn k_gfstack_ws > $O/sum_grid_nn2048.log 2>&1
    prof grid_ml_c2048 $B $G --chains 2048 --interp multilinear
    $S $O/grid_ml_c2048 $O/out ${tag}_grid_c2048_ml k_gfstack_runs > $O/sum_grid_ml2048.log 2>&1
    ;;
  c)
    c4 4096
    ;;
  d)
    # the line first (the artefact), then the kernel statistics of the command with 10 steps, one timed region and without the sampler / sharded
    # legs (the full command under rocprofv3 did not end within 15 minutes and the profiler then waits for its child beyond
    # any timeout -- this part was lost twice)
    cd $R
    timeout -k 10 1000 python bench.py --full-json $O/out/${tag}_bench_default_full.json > $O/out/${tag}_bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
    tail -c 600 $O/out/${tag}_bench_default.json
    mkdir -p $O/default; cd /tmp
    timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/default/stats -o bench -- python $R/bench.py --no-cpu-baseline --steps 10 --repeats 1 --variant-legs multilinear,toeplitz,default_config,prewhitened,geometry,fp32,config4,realistic_grid --full-json /tmp/bench_full_prof.json > $O/default/stats_run.log 2>&1
    $S $O/default $O/out ${tag}_bench_default "k_quadform<128>" k_gfstack_runs "k_gemm_f64<0>" k_gfstack_ws > $O/sum_default.log 2>&1
    ;;
  esac
  grep -h "^{\"metric" $O/*/stats_run.log > $O/out/${tag}_bench_lines_under_profiler_$part.jsonl 2>/dev/null
  cat $O/sum_*.log 2>/dev/null | grep -E "kernel\"|avg_us\"|corrected|write_bytes|no dispatch" | cut -c1-160
  find $O -name "*.db" -size +2M -delete; find $O -name "*.csv" -size +2M -delete
  du -sh $O
  ;;
stats)
  tag=$1; shift; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -o run -- "$@" > $O/run.log 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -24 "$f" | cut -c1-200
  tail -3 $O/run.log | cut -c1-300
  find $O -name "*trace.csv" -size +3M -delete
  ;;
pmc)
  tag=$1; like=$2; cmd=$3; shift 3; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O; cd /tmp
  i=0
  for set in "$@"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o prof -- $cmd > $O/p$i.log 2>&1
  done
  summarize_counters $O "$like"
  find $O -name "*.db" -size +2M -delete
  ;;
clock)
  tag=$1; like=$2; shift 2; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O; cd /tmp
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o prof -- "$@" > $O/run.log 2>&1
  python - "$O" "$like" <<'PY'
import glob, json, sqlite3, sys
O, like = sys.argv[1], sys.argv[2]
db = sqlite3.connect(glob.glob(O + "/p/**/*results.db", recursive=True)[0])
rows = list(db.execute("select kernel_name, value, start, end from counters_collection where counter_name='GRBM_GUI_ACTIVE' "
                       "and kernel_name like ? order by start", ("%" + like + "%",)))
phase, last = [], None
for k, v, s, e in rows:
    k = k.split('(')[0].replace('void ', '').replace('beatamd::', '')
    dur = (e - s) * 1e-9
    if k != last:
        phase.append([k, []]); last = k
    phase[-1][1].append((dur * 1e3, v / 8.0 / dur / 1e9))      # (the counter is summed over the 8 XCDs)
res = []
for k, l in phase:
    l2 = l[2:] if len(l) > 4 else l
    res.append(dict(kernel=k, dispatches=len(l), avg_ms=sum(x[0] for x in l2) / len(l2), avg_GHz=sum(x[1] for x in l2) / len(l2)))
print(json.dumps(res, indent=1))
json.dump(res, open(O + "/clock_summary.json", "w"), indent=1)
PY
  find $O -name "*.db" -size +2M -delete
  ;;
ab)
  tag=$1; variants=$2; shift 2; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
  for rep in 1 2; do for lib in "$@"; do
    BEATAMD_LIB=$R/beat_amd/$lib timeout 600 python tools/exp_variants.py $O/$lib.$rep.jsonl $variants > $O/$lib.$rep.log 2>&1
  done; done
  ;;
torchrun1)
  cd $R
  BEATAMD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
    --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/torchrun1.json 2> gpurun_out/torchrun1.err
  tail -c 700 gpurun_out/torchrun1.json; tail -3 gpurun_out/torchrun1.err
  ;;
*) sed -n 2,20p $0 ;;
esac
