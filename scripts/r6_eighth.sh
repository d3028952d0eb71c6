# round 6, GPU box, eighth call: LBS matrix blend without the joints no lane of a wavefront follows -- parity and stage times
out=gpurun_out/r6_eighth; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1200 python -m pytest tests/test_lbs_gpu.py tests/test_headline_gpu.py tests/test_training_gpu.py -x -q -m gpu > $out/tests_lbs.txt 2>&1
tail -4 $out/tests_lbs.txt
for st in lbs_warp_forward lbs_warp_backward; do bash scripts/ab_variants.sh $st lbs_dense 2>&1 | tee -a $out/ab_lbs.txt; done
