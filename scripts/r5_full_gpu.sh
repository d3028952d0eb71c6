out=gpurun_out/r5_full; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -6 $out/tests.txt
grep -n "C5 product\|C5 reference\|headline C3" $out/tests.txt | head
