# round 2, first GPU pass: the whole -m gpu suite, the default bench, the forced-dist bench, host/GPU balance of the plan
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2b}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
SOAR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-stage-timers > $out/bench_dist.json 2> $out/bench_dist.err; tail -c 400 $out/bench_dist.json
python scripts/plan_host_time.py > $out/host_time.log 2>&1; cat $out/host_time.log | grep graphs
