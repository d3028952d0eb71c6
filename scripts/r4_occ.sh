# round 4, GPU box: the occlusion gradient's tests (fused into the backward blend / separate walk / separate pass)
out=gpurun_out/r4_occ; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_plugin_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "occ or avatar or fused_view or gt_forward or batch_forward" > $out/tests.txt 2>&1
tail -12 $out/tests.txt
