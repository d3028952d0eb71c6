# round 6, GPU box: the tile-list launch against the number of frames in it (is the launch one round of workgroups or two?)
out=gpurun_out/r6_bin2; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for f in 1 2 3 4 5 6 8; do
  echo -n "frames per step $f: "
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline --frames-per-step $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us']; print('tile_lists %6.1f us  depth_order %5.1f  tile_ranges %5.1f  %.3f ms/step' % (s['tile_lists'], s['depth_order'], s['tile_ranges'], d['ms_per_step']))"
done 2>&1 | tee $out/frames.txt
