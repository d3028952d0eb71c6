# round 5, GPU box: headline pin test again; frame groups on two streams; backward at three waves per SIMD without spills
out=gpurun_out/r5_exp1; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s > $out/headline.txt 2>&1
tail -8 $out/headline.txt
run() { python "$@" --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %.4f ms/step  %.1f frames/s' % (d['ms_per_step'], d['value']))"; }
for r in 1 2; do
  echo -n "groups=1 "; SOAR_PLAN_GROUPS=1 run bench.py
  echo -n "groups=2 "; SOAR_PLAN_GROUPS=2 run bench.py
  echo -n "groups=4 "; SOAR_PLAN_GROUPS=4 run bench.py
done
bash scripts/ab_variants.sh render_backward bwd_wpe3
