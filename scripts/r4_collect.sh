# copy what is judged from gpurun_out/<tag>/ (scripts/r4_profiles.sh) into profiles/<tag>_*: bash scripts/r4_collect.sh r04c
tag=${1:-r04c}
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
cp $src/plugin_path.txt profiles/${tag}_plugin_path.txt
[ -f $src/kernel_stats_avatar.csv ] && cp $src/kernel_stats_avatar.csv profiles/${tag}_kernel_stats_avatar_C3.csv
f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${tag}_kernel_stats_C3_plan_mode.csv
python3 - <<PY
import json
p = json.load(open("$src/bench_plain.json")); d = json.load(open("$src/bench_forced_dist.json")); e = json.load(open("$src/bench_forced_dist_1bucket.json"))
open("profiles/${tag}_forced_dist_vs_plain.txt", "w").write(
    "python bench.py --no-cpu-baseline --no-stage-timers (C3, default form, one GPU): %.3f ms per step, %.0f frames/s\\n"
    "SOAR_BENCH_FORCE_DIST=1 (the same with a one-rank RCCL group; the default: two asynchronous all-reduce buckets per step, the optimizer inside the "
    "plan behind them): %.3f ms per step, %.0f frames/s (%+.1f %%); ranks: %s\\n"
    "SOAR_BENCH_FORCE_DIST=1 SOAR_DP_BUCKETS=1 (one collective for the whole buffer): %.3f ms per step, %.0f frames/s (%+.1f %%); ranks: %s\\n"
    % (p["ms_per_step"], p["value"], d["ms_per_step"], d["value"], 100.0 * (d["ms_per_step"] / p["ms_per_step"] - 1.0), json.dumps(d.get("ranks")),
       e["ms_per_step"], e["value"], 100.0 * (e["ms_per_step"] / p["ms_per_step"] - 1.0), json.dumps(e.get("ranks"))))
PY
ls profiles | grep $tag
