# round-4 evidence, on the GPU box: gpurun -- 'bash scripts/r4_profiles.sh r04c'.  Outputs under gpurun_out/<tag>/ ; the ones that are
# judged get copied to profiles/ (scripts/r4_collect.sh).  Every rocprofv3 call under `timeout` and with the program itself behind `--`.
tag=${1:-r04c}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. durations of the batched launches of the default form (one launch per stage for the 4 frames of a step)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
python3 scripts/batched_trace.py $out/trace 4 > $out/batched_launches.txt
# 2. HBM-side traffic, FETCH_SIZE and WRITE_SIZE in separate passes, one launch per frame and stage (SOAR_PLAN_BATCHED=0) so that the
#    counters are per frame like the algorithmic bytes
export SOAR_PLAN_BATCHED=0
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/pmc_write.log 2>&1
unset SOAR_PLAN_BATCHED
python3 scripts/make_traffic_json.py $out $out/hbm_traffic.json > $out/hbm_traffic.txt 2>&1
# 3. SQ counters of the two blend kernels (batched launches)
bash scripts/pmc_kernel.sh "render_" > $out/sq_counters.txt 2>&1
# 4. the bench lines: default (with the CPU baselines), the avatar-loss form, the forced RCCL path with two buckets and with one, C5, C2
python3 bench.py --pmc-json $out/hbm_traffic.json > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --loss avatar --no-cpu-baseline > $out/bench_avatar.json 2> $out/bench_avatar.err
python3 bench.py --no-cpu-baseline --no-stage-timers > $out/bench_plain.json 2> $out/bench_plain.err
SOAR_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-stage-timers > $out/bench_forced_dist.json 2> $out/bench_forced_dist.err
SOAR_BENCH_FORCE_DIST=1 SOAR_DP_BUCKETS=1 python3 bench.py --no-cpu-baseline --no-stage-timers > $out/bench_forced_dist_1bucket.json 2> $out/bench_forced_dist_1bucket.err
python3 bench.py --workload C5 --no-cpu-baseline --steps 40 > $out/bench_C5.json 2> $out/bench_C5.err
python3 bench.py --workload C2 --no-cpu-baseline > $out/bench_C2.json 2> $out/bench_C2.err
cat $out/batched_launches.txt
for f in default avatar plain forced_dist forced_dist_1bucket C5 C2; do python3 -c "
import json,sys
d=json.load(open('$out/bench_$f.json')); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'), d.get('ranks'))"; done
# 4b. the avatar-loss form's kernels (durations of the batched launches)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_avatar -o t -- python3 bench.py --loss avatar --steps 20 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace_avatar.log 2>&1
f=$(find $out/trace_avatar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_avatar.csv
# 5. the plugin path
python3 scripts/plugin_time.py 2>&1 | grep -v -E "Warning|amdgpu.ids" > $out/plugin_path.txt
python3 scripts/plugin_host_split.py 2>&1 | grep -v -E "Warning|amdgpu.ids" >> $out/plugin_path.txt
cat $out/plugin_path.txt
