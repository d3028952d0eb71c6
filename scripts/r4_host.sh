export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
python scripts/plan_host_time.py 2>&1 | grep -v amdgpu | head -12
python -m pytest tests/test_plugin_gpu.py tests/test_training_gpu.py -x -q -m gpu -k "step_plan" 2>&1 | tail -3
bash scripts/ab_stages3.sh prev 2>&1 | cut -c1-60
