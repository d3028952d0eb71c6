out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2ch}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
python3 scripts/trace_chains.py $out/trace
