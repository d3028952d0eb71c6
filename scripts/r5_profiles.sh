# round-5 evidence, on the GPU box: gpurun -- 'bash scripts/r5_profiles.sh r05'.  Outputs under gpurun_out/<tag>/ ; the ones that are
# judged get copied to profiles/ (scripts/r5_collect.sh).  Every rocprofv3 call under `timeout` and with the program itself behind `--`.
tag=${1:-r05}
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
# 4. the bench lines: default (with the CPU baselines), the avatar-loss form, C5, C2; the driver's exact command in five fresh processes
python3 bench.py --pmc-json $out/hbm_traffic.json > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --loss avatar --no-cpu-baseline > $out/bench_avatar.json 2> $out/bench_avatar.err
python3 bench.py --workload C5 --no-cpu-baseline --steps 40 > $out/bench_C5.json 2> $out/bench_C5.err
python3 bench.py --workload C2 --no-cpu-baseline > $out/bench_C2.json 2> $out/bench_C2.err
for r in 1 2 3 4 5; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pmc-json $out/hbm_traffic.json 2> /dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('python bench.py --gpus 1 --steps 20 --warmup 5: %.1f frames/s  %.4f ms/step (HIP event pair: %.4f)  host issue %.3f  frac %.4f counter_frac %s' % (d['value'], d['ms_per_step'], d['config']['device_events_ms_per_step'], d['config']['host_issue_ms_per_step'], r['frac'], r.get('counter_frac')))"
done > $out/bench_spread.txt
for r in 1 2 3; do
  python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2> /dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('python bench.py (100 steps): %.1f frames/s  %.4f ms/step (HIP event pair: %.4f)' % (d['value'], d['ms_per_step'], d['config']['device_events_ms_per_step']))"
done >> $out/bench_spread.txt
cat $out/batched_launches.txt $out/bench_spread.txt
for f in default avatar C5 C2; do python3 -c "
import json,sys
d=json.load(open('$out/bench_$f.json')); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('counter_frac'), r.get('traffic'), (r.get('whole_frame') or {}).get('counter_frac'))"; done
# 4b. the avatar-loss form's kernels (durations of the batched launches)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_avatar -o t -- python3 bench.py --loss avatar --steps 20 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace_avatar.log 2>&1
f=$(find $out/trace_avatar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_avatar.csv
# 5. the plugin path: per frame, gt_forward, and the reference's 7-view step (this tree against the round-4 tree when it is there)
python3 scripts/plugin_time.py 2>&1 | grep -v -E "Warning|amdgpu.ids" > $out/plugin_path.txt
python3 scripts/plugin_host_split.py 2>&1 | grep -v -E "Warning|amdgpu.ids" >> $out/plugin_path.txt
if [ -d _r4_tree ]; then bash scripts/r5_refstep_ab.sh 2>&1 | grep -v -E "Warning|amdgpu.ids" > $out/refstep.txt; else SOAR_REFSTEP_FORMS=all python3 scripts/refstep_time.py 2>&1 | grep "reference step" > $out/refstep.txt; fi
cat $out/plugin_path.txt $out/refstep.txt
