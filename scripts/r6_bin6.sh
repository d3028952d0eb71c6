# round 6, GPU box: bin_tiles as a grid of at most SOAR_BIN_GRID workgroups per frame over the list of super-tiles with work (band_place): parity,
# the helpers' path, then the grid size
out=gpurun_out/r6_bin6; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q -k "not c5" 2>&1 | tail -3 | tee $out/tests.txt
SOAR_BIN_SPLIT_AT=32 timeout 1500 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q -k "not c5" 2>&1 | tail -3 | tee $out/tests_split32.txt
bash scripts/ab_variants.sh tile_lists bin_g128 bin_g192 bin_g384 bin_g512 2>&1 | tee $out/ab.txt
SOAR_BIN_LOG=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep bin_tiles | head -3 | tee $out/log.txt
