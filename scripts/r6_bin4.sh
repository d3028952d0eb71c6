# round 6, GPU box: the whole GPU suite on the new tile lists, the log of one launch, bench lines
out=gpurun_out/r6_bin4; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $out/tests.txt
SOAR_BIN_LOG=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep bin_tiles > $out/log.txt; head -3 $out/log.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['repeats_ms_per_step'], d['roofline']['stage_us'])" | tee $out/bench20.txt
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --workload C5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_us'])" | tee $out/benchC5.txt
