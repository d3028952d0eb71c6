# round 5, GPU box: the reference's 7-view step through the plugin -- the round-4 tree (commit 254e7da: one autograd node per pose, cameras
# by torch ops on the host, stacked outputs copied, one cos_loss call per view) against this tree, interleaved on ONE box
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for r in 1 2 3 4 5; do
  echo -n "round 4 tree: "; (cd _r4_tree && python scripts/refstep_time.py 2>&1 | tail -1 | cut -c1-125)
  echo -n "this tree:    "; python scripts/refstep_time.py 2>&1 | tail -1 | cut -c1-125
done
echo -n "this tree, one node per pose / one forward() per view: "; SOAR_REFSTEP_FORMS=all python scripts/refstep_time.py 2>&1 | tail -2 | cut -c1-160
