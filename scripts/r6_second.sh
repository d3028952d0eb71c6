# round 6, GPU box, second call: full GPU suite; the C5 triangle with L2 in float64; the C5 tests 20 x in fresh processes;
# the strict C3 surfel bar with pairs of blocks 8 x; the driver's bench command
out=gpurun_out/r6_second; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 3000 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1
tail -8 $out/tests.txt
V="div2=soar_amd/_lib/variants/div2.so region2=soar_amd/_lib/variants/region2.so"
timeout 900 python scripts/r6_c5_triangle.py --scene C5 --grads noise --runs 3 --variants $V > $out/tri_C5_noise.txt 2>&1
timeout 900 python scripts/r6_c5_triangle.py --scene C5 --grads loss --runs 3 --variants $V > $out/tri_C5_loss.txt 2>&1
timeout 900 python scripts/r6_c5_triangle.py --scene C3 --grads noise --runs 3 --variants $V > $out/tri_C3_noise.txt 2>&1
timeout 900 python scripts/r6_c5_triangle.py --scene C3 --grads loss --runs 3 --variants $V > $out/tri_C3_loss.txt 2>&1
for i in $(seq 1 20); do
  timeout 600 python -m pytest tests/test_reference_build_gpu.py -q -m gpu -s -k "c5_frame" 2>&1 | grep -E "^C5 |passed|failed|Error|assert" >> $out/c5_20x.txt
done
grep -c "2 passed" $out/c5_20x.txt
for i in $(seq 1 8); do
  SOAR_HIP_LIB=$PWD/soar_amd/_lib/variants/region2.so timeout 600 python -m pytest tests/test_reference_build_gpu.py -q -m gpu -s -k "C3_100k_1080p or c3_frame_with" 2>&1 | grep -E "^C3|passed|failed|dL_drotations" >> $out/region2_c3_8x.txt
done
tail -30 $out/region2_c3_8x.txt | cut -c1-400
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $out/bench.err | tail -1 > $out/bench.json
python - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("driver form: %.1f frames/s  %.4f ms/step repeats %s frac %.4f" % (d["value"], d["ms_per_step"], d["repeats_ms_per_step"], d["roofline"]["frac"]))
print(d.get("reference_same_box"))
print(d["roofline"]["stage_us_per_step"])
PY
