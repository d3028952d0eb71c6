# round 6, GPU box: the KNN refresh's two launches with fewer, longer wavefronts (runs of queries in a loop)
out=gpurun_out/r6_knn2; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/ab_variants.sh lbs_knn_weights knn_c2 knn_c4 knn_b2 knn_b4 knn_c4b4 2>&1 | tee $out/ab.txt
