# round 6, GPU box: pairs of blocks (-DSOAR_BWD_REGION=2) with float64 accumulation rows: what it costs, and whether the strict C3 bar holds
out=gpurun_out/r6_region2_f64; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run() { python "$@" --steps 100 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%8.1f us  %.3f ms/step' % (d['roofline']['stage_us']['render_backward'], d['ms_per_step']))"; }
{
echo -n "default                       "; run bench.py
echo -n "default, float64 rows         "; SOAR_DETERMINISTIC_BACKWARD=1 run bench.py
echo -n "pairs of blocks               "; run scripts/ab_lib.py soar_amd/_lib/variants/bwd_r2.so
echo -n "pairs of blocks, float64 rows "; SOAR_DETERMINISTIC_BACKWARD=1 run scripts/ab_lib.py soar_amd/_lib/variants/bwd_r2.so
echo -n "default                       "; run bench.py
} 2>&1 | tee $out/timing.txt
for r in 1 2 3 4 5 6 7 8; do
  SOAR_HIP_LIB=$PWD/soar_amd/_lib/variants/bwd_r2.so SOAR_DETERMINISTIC_BACKWARD=1 timeout 600 python -m pytest tests/test_reference_build_gpu.py -m gpu -q -s -k "c3 or C3" 2>&1 | grep -E "passed|failed|dL_drotations" | cut -c1-260
done | tee $out/strict.txt
