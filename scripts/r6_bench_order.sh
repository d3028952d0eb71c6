# round 6, GPU box: bench.py with the per-kernel pass right behind the timed region (in front of the repeated regions)
out=gpurun_out/r6_bench_order; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stage_us']; print(d['value'], d['ms_per_step'], d['repeats_ms_per_step'], 'frac', r['frac'], 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'sum', round(sum(s.values()),1))"; done
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['stage_us']; print(d['value'], d['ms_per_step'], d['repeats_ms_per_step'], 'frac', r['frac'], 'fwd', s['render_forward'], 'bwd', s['render_backward'], 'sum', round(sum(s.values()),1))"
