# round 6, GPU box: ranks per pass of the blend kernels' grids (development switch SOAR_BLEND_GRID_RANKS), four frames per launch
out=gpurun_out/r6_grid; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for g in 1024 512 640 768 896 1280; do
  echo -n "ranks $g: "
  SOAR_BLEND_GRID_RANKS=$g python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_us']; print('%.3f ms/step  fwd %.1f bwd %.1f' % (d['ms_per_step'], s['render_forward'], s['render_backward']))"
done 2>&1 | tee $out/grid.txt
