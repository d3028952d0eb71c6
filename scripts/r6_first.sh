# round 6, GPU box, first call: the accuracy triangle (C5 noise, C5 loss, C3 noise, C3 loss) with the default build, float64 rows,
# the IEEE-division build and the pairs-of-blocks build
out=gpurun_out/r6_first; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
V="div2=soar_amd/_lib/variants/div2.so region2=soar_amd/_lib/variants/region2.so"
timeout 1500 python scripts/r6_c5_triangle.py --scene C3 --grads noise --runs 2 --variants $V > $out/tri_C3_noise.txt 2>&1
tail -40 $out/tri_C3_noise.txt
timeout 1500 python scripts/r6_c5_triangle.py --scene C5 --grads noise --runs 3 --variants $V > $out/tri_C5_noise.txt 2>&1
tail -40 $out/tri_C5_noise.txt
timeout 1500 python scripts/r6_c5_triangle.py --scene C3 --grads loss --runs 2 --variants $V > $out/tri_C3_loss.txt 2>&1
tail -40 $out/tri_C3_loss.txt
timeout 1500 python scripts/r6_c5_triangle.py --scene C5 --grads loss --runs 2 --variants $V > $out/tri_C5_loss.txt 2>&1
tail -40 $out/tri_C5_loss.txt
