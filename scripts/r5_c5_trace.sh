# kernel durations of the C5 line (300k Gaussians, 4K): gpurun -- 'bash scripts/r5_c5_trace.sh'
out=$GRAFT_REPO_ROOT/gpurun_out/r05_c5
mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --workload C5 --steps 20 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_C5.csv
head -30 $out/kernel_stats_C5.csv | cut -c1-150
