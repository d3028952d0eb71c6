"""bench.py keeps its output contract: ONE JSON line with the metric fields, the roofline object and the cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["plan", "plan-eager"])
def test_bench_line_schema(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1",
                        "--mode", mode], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "frames/s"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["mode"] == mode
    # ms_per_step is printed to 3 decimals: half a unit of the last one is the tolerance
    assert d["value"] > 0 and abs(d["value"] - 4 * 1000.0 / d["ms_per_step"]) <= d["value"] * (0.00051 / d["ms_per_step"] + 1e-6)
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1
    # the spread of the number and the same-box context are in the line itself
    rep = d["repeats_ms_per_step"]
    assert len(rep) == 4 and all(v > 0 for v in rep)
    assert "reference_same_box" in d
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_rasterizer_fast.so")):
        rb = d["reference_same_box"]
        assert rb["fwd_ms"] > 0 and rb["bwd_ms"] > 0 and rb["product_one_view_fwd_ms"] > 0 and "context only" in rb["build"], rb


def test_bench_avatar_loss_line():
    """`bench.py --loss avatar`: the second line -- the reference's avatar-stage losses in the step plan, Adam also on the occlusion
    values -- keeps the one-line contract; its roofline kernel is one of the stages that are one launch per step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1",
                        "--loss", "avatar", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["config"]["loss"].startswith("avatar") and d["config"]["plan_form"] == "batched, eager" and d["value"] > 0
    assert d["roofline"]["kernel"] not in ("frame_loss", "postops") and d["roofline"]["launches"] == 3


def test_bench_forced_dist_path_runs_rccl():
    """SOAR_BENCH_FORCE_DIST=1: the multi-rank code path (RCCL process group, barriers, bucketed asynchronous all-reduce, the
    same default mode as a real multi-rank job) with a single rank -- the only way to exercise it on a one-GPU box.  stdout
    must still be exactly the one JSON line (RCCL prints banners)."""
    env = dict(os.environ, SOAR_BENCH_FORCE_DIST="1", MASTER_PORT="29537")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["mode"] == "plan-eager" and d["config"]["collectives"].startswith("rccl")
    assert d["value"] > 0
    # what makes a first multi-rank run explain itself: per-rank step time, instances per rank, the stall per gradient bucket
    rk = d["ranks"]
    assert len(rk["per_rank_ms_per_step"]) == 1 and rk["per_rank_ms_per_step"][0] > 0 and rk["imbalance_max_over_mean"] == 1.0
    # (one collective for the whole buffer by default: what measured faster next to a one-rank communicator, round 5)
    assert rk["max_num_rendered_per_rank"][0] > 0 and rk["buckets"] == 1 and len(rk["bucket_wait_us_per_rank"][0]) == 2
    # two buckets -- the positions first, the rest behind the KNN refresh (SOAR_DP_BUCKETS=2) -- and the collective on the step's own
    # stream (SOAR_DP_BUCKETS=0): same line, same shape
    for buckets in ("2", "0"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "4", "--warmup", "1",
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env, SOAR_DP_BUCKETS=buckets))
        assert r.returncode == 0, r.stderr[-2000:]
        d1 = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
        assert d1["ranks"]["buckets"] == int(buckets) and d1["value"] > 0


def test_bench_gpus_flag_must_match_the_job():
    """--gpus 2 on a one-GPU box: no line at all (never n_gpus: 1 for --gpus 2)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_bench_two_gpus_when_there_are_two():
    """On a box with at least two GPUs: `bench.py --gpus 2` starts two ranks over RCCL and prints ONE line with n_gpus == 2 whose
    value is the whole job's (skipped on the one-GPU boxes of this pool: the day a node with more appears, this runs)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "frame-dp2" and d["config"]["collectives"].startswith("rccl")
    assert d["config"]["plan_form"] is not None
    assert abs(d["value"] - 2 * 4 * 1000.0 / d["ms_per_step"]) <= d["value"] * (0.00051 / d["ms_per_step"] + 1e-6)


def test_two_rank_nccl_gradients_equal_one_rank(tmp_path):
    """The real renderer under frame-DP over RCCL: 2 ranks x 2 frames, gradients summed by the bucketed all-reduce == 1 rank x the
    same 4 frames, within 1e-5 (tests/tools/two_rank_check.py; the gloo test of the CPU suite checks the same property on a
    stand-in loss).  Skipped below two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "tests", "tools", "two_rank_check.py"), str(tmp_path)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "two-rank gradients equal the one-rank gradients" in r.stdout


def test_visible_gpu_count_needs_no_hip():
    """bench.visible_gpu_count() (what the rank launcher uses) agrees with torch, from sysfs / the environment alone."""
    import torch
    code = ("import sys; sys.argv=['bench.py']; import bench; n = bench.visible_gpu_count(); import torch; "
            "assert not torch.cuda.is_initialized(); print(n)")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    assert int(r.stdout.strip().splitlines()[-1]) == torch.cuda.device_count()
