# round 4, GPU box: the plugin path behind one C call per pose (tests + timing + host split + kernel trace).  Writes gpurun_out/r4_plugin/*
out=gpurun_out/r4_plugin; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python -m pytest tests/test_plugin_gpu.py -x -q -m gpu > $out/tests.txt 2>&1
tail -5 $out/tests.txt
python scripts/plugin_time.py 2>&1 | grep -v amdgpu.ids > $out/plugin_time.txt
cat $out/plugin_time.txt
python scripts/plugin_host_split.py 2>&1 | grep -v amdgpu.ids > $out/host_split.txt
cat $out/host_split.txt
bash scripts/plugin_trace.sh > $out/trace.txt 2>&1
tail -30 $out/trace.txt
