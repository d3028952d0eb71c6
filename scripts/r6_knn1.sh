# round 6, GPU box: what bounds the KNN refresh's blend -- the skinning rows' bytes or its instructions? (timing-only variants)
out=gpurun_out/r6_knn1; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/ab_variants.sh lbs_knn_weights knn_norows knn_onerow 2>&1 | tee $out/ab.txt
