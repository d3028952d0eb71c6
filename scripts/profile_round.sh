# Run on the GPU box (gpurun -- 'bash scripts/profile_round.sh <tag>'): kernel-trace stats + HBM traffic counters of
# the default bench workload, then the full default bench line (with cpu_baseline).  Outputs under gpurun_out/<tag>/.
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
# HBM traffic of the blend kernels, separate pass with counters only (MI355X_MICROARCH.md, HBM / rocprofv3 section)
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: 3 + 2 of the 4 TCC slots)
# (SOAR_PLAN_BATCHED=0: one launch per frame and stage, so that the counters are per FRAME like the algorithmic bytes they are held
# against; the default at this size launches every stage once for the four frames of a step)
export SOAR_PLAN_BATCHED=0
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-stage-timers > $out/pmc_write.log 2>&1
unset SOAR_PLAN_BATCHED
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
tail -1 $out/bench_default.json
find $out -name "*kernel_stats.csv" | head -3
