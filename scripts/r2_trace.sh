# kernel timeline of the plan-mode bench (default and forced-dist): where are the gaps?
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2c}
mkdir -p $out
cd $GRAFT_REPO_ROOT
python scripts/plan_host_time.py > $out/host_time.log 2>&1; grep -v "^/opt" $out/host_time.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
SOAR_BENCH_FORCE_DIST=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/dist/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/trace_dist.log 2>&1
python3 scripts/trace_step.py $out > $out/step_timeline.txt 2>&1 || true
python3 scripts/trace_step.py $out/dist > $out/step_timeline_dist.txt 2>&1 || true
head -3 $out/step_timeline.txt $out/step_timeline_dist.txt
