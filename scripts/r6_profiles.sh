# round 6, GPU box: the evidence set of the round on the final build (the recipe of round 5, scripts/r5_profiles.sh, under the tag r06),
# then what the driver runs at round end
out=gpurun_out/r6_profiles; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
bash scripts/r5_profiles.sh r06 > $out/profiles.log 2>&1
tail -40 $out/profiles.log
