out=gpurun_out/r5_exp4; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_plugin_gpu.py -x -q -m gpu -k "camera_walking or does_not_fit or default_sizes" > $out/tests.txt 2>&1
tail -5 $out/tests.txt
SOAR_REFSTEP_FORMS=all timeout 900 python scripts/refstep_time.py 2>&1 | tail -8
