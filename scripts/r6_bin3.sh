out=gpurun_out/r6_bin3; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
SOAR_BIN_LOG=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep bin_tiles | tee $out/log_new.txt
