# copy what is judged from gpurun_out/<tag>/ (scripts/r3_profiles.sh) into profiles/<tag>_*: bash scripts/r3_collect.sh r03a
tag=${1:-r03a}
src=gpurun_out/$tag
{ echo "rocprofv3 --kernel-trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-stage-timers (C3, default form): the launches with gridDim.y = 4, i.e. every stage of the frame chain once for the four frames of a step (scripts/batched_trace.py; the warm-up steps of bench.py launch per frame and are left out; the KNN refresh, the warps of all frames and Adam are launches without a frame dimension and not in this list)"; cat $src/batched_launches.txt; } > profiles/${tag}_batched_launches_C3.txt
cp $src/hbm_traffic.json profiles/${tag}_hbm_traffic.json
cp $src/sq_counters.txt profiles/${tag}_sq_counters_blend_kernels.txt
cp $src/bench_default.json profiles/${tag}_bench_default_C3.json
cp $src/bench_avatar.json profiles/${tag}_bench_avatar_C3.json
cp $src/bench_C5.json profiles/${tag}_bench_C5.json
cp $src/plugin_path.txt profiles/${tag}_plugin_path.txt
cp $src/gradient_gap.txt profiles/${tag}_gradient_gap_scenes.txt
f=$(find $src/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${tag}_kernel_stats_C3_plan_mode.csv
python3 - <<PY
import json
p = json.load(open("$src/bench_default.json")); d = json.load(open("$src/bench_forced_dist.json"))
open("profiles/${tag}_forced_dist_vs_plain.txt", "w").write(
    "python bench.py (C3, default form, one GPU): %.3f ms per step, %.0f frames/s\\n"
    "SOAR_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-stage-timers (the same with a one-rank RCCL group: two asynchronous "
    "all-reduce buckets per step, the optimizer inside the plan behind them): %.3f ms per step, %.0f frames/s (%+.1f %%)\\n"
    % (p["ms_per_step"], p["value"], d["ms_per_step"], d["value"], 100.0 * (d["ms_per_step"] / p["ms_per_step"] - 1.0)))
PY
if [ -f $src/fuzz.txt ]; then { echo "tests/tools/fuzz_vs_reference.py 1500 70000 on MI355X with the round-3 kernels (entry-lane backward over block masks, fused tile binning): the product against the reference's own kernels (oracle/_ref).  Lines of the run that report a scene:"; grep -E "MISMATCH|fp32 conditioning|scenes," $src/fuzz.txt | cut -c1-900; } > profiles/${tag}_fuzz_vs_reference_kernels.txt; fi
ls profiles | grep $tag
