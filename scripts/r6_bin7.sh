out=gpurun_out/r6_bin7; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for w in C3 C5 C2; do python bench.py --steps 100 --warmup 5 --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us']; print('$w', d['value'], d['ms_per_step'], {k: s[k] for k in ('tile_lists','tile_ranges','depth_order','block_masks')})"; done | tee $out/bench.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $out/tests.txt
