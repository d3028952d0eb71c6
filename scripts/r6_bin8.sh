# round 6, GPU box: bin_tiles' grid at C5 (one frame per launch, four frame chains on their own streams; 2040 super-tiles per frame)
out=gpurun_out/r6_bin8; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run() { python "$@" --steps 60 --warmup 5 --no-cpu-baseline --workload C5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f frames/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
{ for r in 1 2; do echo -n "standard (512) "; run bench.py; for n in bin_g256 bin_g1024 bin_g2048; do echo -n "$n  "; run scripts/ab_lib.py soar_amd/_lib/variants/$n.so; done; done; } | tee $out/ab.txt
