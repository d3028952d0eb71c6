# round 6, GPU box: the two scenes of the 1500-scene sweep that came out above the bars, replayed alone in fresh processes
out=gpurun_out/r6_fuzz_replay; mkdir -p $out
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for s in 71 526; do for r in 1 2 3 4 5; do
  SOAR_FUZZ_THREADS=8 timeout 600 python tests/tools/fuzz_vs_reference.py 1500 70000 --only=$s 2>&1 | grep -E "^\[$s\]|scenes," | cut -c1-330
done; done | tee $out/replay.txt
