# round 5, GPU box: the step next to a live one-rank RCCL communicator -- plain / two buckets / one bucket / one collective on the step's stream
out=gpurun_out/r5_dist; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run() { python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-stage-timers 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(' %.4f ms/step  %.1f frames/s  %s' % (d['ms_per_step'], d['value'], (d.get('ranks') or {}).get('bucket_wait_us_per_rank')))"; }
for r in 1 2 3; do
  echo -n "plain            "; run
  echo -n "forced, buckets=2"; SOAR_BENCH_FORCE_DIST=1 SOAR_DP_BUCKETS=2 run
  echo -n "forced, buckets=1"; SOAR_BENCH_FORCE_DIST=1 SOAR_DP_BUCKETS=1 run
  echo -n "forced, buckets=0"; SOAR_BENCH_FORCE_DIST=1 SOAR_DP_BUCKETS=0 run
done
