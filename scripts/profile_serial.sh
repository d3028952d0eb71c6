# Per-kernel durations without overlap: views serialised on one stream, synchronous mode (what bench.py's per-kernel pass
# measures with HIP events).  gpurun -- 'bash scripts/profile_serial.sh <tag>'
tag=${1:-serial}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SOAR_STREAMS=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-stage-timers --mode sync > $out/trace.log 2>&1
tail -1 $out/trace.log | cut -c1-300
find $out -name "*kernel_stats.csv" | head -2
