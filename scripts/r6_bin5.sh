# round 6, GPU box: the helpers' path of bin_tiles on small scenes (threshold lowered by the environment), then the whole rasterizer
# parity file with the threshold at 32 (every band of every scene split)
out=gpurun_out/r6_bin5; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 900 python -m pytest tests/test_rasterizer_gpu.py -m gpu -x -q -k "idle_columns" 2>&1 | tail -5 | tee $out/test.txt
SOAR_BIN_SPLIT_AT=32 timeout 1500 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_build_gpu.py -m gpu -x -q -k "not c5" 2>&1 | tail -3 | tee $out/all_split32.txt
