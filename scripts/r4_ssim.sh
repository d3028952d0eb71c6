# round 4, GPU box: SSIM kernels -- parity tests, then the avatar line + per-kernel durations.  usage: bash scripts/r4_ssim.sh TAG
tag=${1:-ssim}; out=$GRAFT_REPO_ROOT/gpurun_out/r4_$tag; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_plugin_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "ssim or avatar or loss" > $out/tests.txt 2>&1
tail -4 $out/tests.txt
python3 bench.py --loss avatar --steps 100 --warmup 5 --no-cpu-baseline 2> $out/bench.err | tail -1 > $out/bench_avatar_C3.json
python3 - $out/bench_avatar_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("avatar C3: %.1f frames/s  %.3f ms/step" % (d["value"], d["ms_per_step"]))
print({k: round(v, 1) for k, v in d["roofline"]["stage_us_per_step"].items()})
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --loss avatar --steps 20 --warmup 5 --no-cpu-baseline --no-stage-timers > $out/trace.log 2>&1
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_avatar_C3.csv
grep -E "ssim|avatar_pixel|view_finish|occ_backward" $out/kernel_stats_avatar_C3.csv | cut -d, -f1-4 | sed 's/soar::(anonymous namespace):://g' | cut -c1-120
rm -rf $out/trace
