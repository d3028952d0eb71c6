# copy what is judged from gpurun_out/<tag>/ (scripts/r5_profiles.sh) into profiles/<tag>_*: bash scripts/r5_collect.sh r05
tag=${1:-r05}
src=gpurun_out/$tag
{ echo "rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timers (C3, default form): the launches with gridDim.y = 4, i.e. every stage of the frame chain once for the four frames of a step (scripts/batched_trace.py; the warm-up steps of bench.py launch per frame and are left out; the KNN refresh, the warps of all frames and Adam are launches without a frame dimension and not in this list)"; cat $src/batched_launches.txt; echo; echo "HBM-side bytes per FRAME of the blend kernels from the same build (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, one launch per frame; ${tag}_hbm_traffic.json):"; python3 -c "
import json
d = json.load(open('$src/hbm_traffic.json'))['kernels']
for k in ('render_forward_kernel', 'render_backward_blocks_kernel', 'tile_order_binned_kernel', 'bin_tiles_kernel', 'geometry_backward_kernel'):
    if k in d: print('  %-32s fetch %7.1f MB  write %7.1f MB' % (k, d[k]['fetch_bytes'] / 1e6, d[k]['WRITE_SIZE_bytes'] / 1e6))
"; } > profiles/${tag}_batched_launches_C3.txt
cp $src/hbm_traffic.json profiles/${tag}_hbm_traffic.json
cp $src/sq_counters.txt profiles/${tag}_sq_counters_blend_kernels.txt
cp $src/bench_default.json profiles/${tag}_bench_default_C3.json
cp $src/bench_avatar.json profiles/${tag}_bench_avatar_C3.json
cp $src/bench_C5.json profiles/${tag}_bench_C5.json
cp $src/bench_C2.json profiles/${tag}_bench_C2.json
[ -f $src/kernel_stats_avatar.csv ] && cp $src/kernel_stats_avatar.csv profiles/${tag}_kernel_stats_avatar_C3.csv
f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${tag}_kernel_stats_C3_plan_mode.csv
{ cat $src/plugin_path.txt; echo; echo "The reference's 7-view training step through the plugin (scripts/refstep_time.py, scripts/r5_refstep_ab.sh): the round-4 tree (commit 254e7da: one autograd node per pose, cameras by torch ops, stacked outputs copied, one cos_loss call per view) against this tree (ONE node / one C call each way for both poses, soar_cameras_from_c2w, stack_views, batched cos_loss), interleaved on one box:"; cat $src/refstep.txt; } > profiles/${tag}_plugin_path.txt
ls profiles | grep $tag
