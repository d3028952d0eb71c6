# round 6, GPU box: bin_tiles with per-wavefront slices and rings, packed hit test, per-band column extents: parity, then A/B against round 5's body
out=gpurun_out/r6_bin1; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q -k "not c5" 2>&1 | tail -5 | tee $out/tests.txt
bash scripts/ab_variants.sh tile_lists bin_old bin_c512 bin_w8 bin_w8c1k 2>&1 | tee $out/ab.txt
SOAR_BIN_LOG=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep bin_tiles | tee $out/log_new.txt
