out=gpurun_out/r5_exp5; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 1500 python -m pytest tests/test_plugin_gpu.py tests/test_integration_stub_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -25 $out/tests.txt
SOAR_REFSTEP_FORMS=all timeout 900 python scripts/refstep_time.py 2>&1 | tail -4
