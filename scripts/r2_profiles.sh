# round-2 evidence: rocprofv3 summaries (plan mode + serialised), HBM counters, SQ counters of the blends, steady-state timeline,
# forced-dist vs plain step time, all-reduce hand-off probe, default bench line.  Outputs under gpurun_out/<tag>/
tag=${1:-r02a}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh $tag > $out/profile_round.log 2>&1
bash scripts/profile_serial.sh ${tag}_serial > $out/profile_serial.log 2>&1
bash scripts/pmc_render.sh > $out/pmc_render.log 2>&1; cp -r gpurun_out/pmc1 gpurun_out/pmc2 $out/ 2>/dev/null
python scripts/plan_phases.py > $out/plan_phases.txt 2>&1
python scripts/allreduce_probe.py > $out/allreduce_probe.txt 2>&1
python bench.py --no-cpu-baseline --no-stage-timers > $out/bench_plain.json 2>/dev/null
SOAR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-stage-timers > $out/bench_forced_dist.json 2>/dev/null
python scripts/plugin_time.py > $out/plugin_time.txt 2>&1
python tests/tools/ref_compare.py > $out/ref_compare.txt 2>&1
tail -2 $out/plan_phases.txt; tail -c 300 $out/bench_default.json
