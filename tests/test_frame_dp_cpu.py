"""Frame data-parallel path on CPU with the gloo backend, world_size 2: round-robin frame
shards cover a global batch exactly once, the ONE flat gradient all-reduce gives every rank the single-process
gradient, and the densification statistics are summed identically."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from soar_amd import frame_dp


def test_shard_frames_partition():
    for world in (1, 2, 3, 8):
        for step in (0, 1, 7):
            batch = frame_dp.global_batch(step, 4, world, num_frames=400)
            assert len(batch) == 4 * world
            shards = [frame_dp.shard_frames(batch, r, world) for r in range(world)]
            assert all(len(s) == 4 for s in shards)
            assert sorted(f for s in shards for f in s) == sorted(batch)
    # consecutive steps walk the sequence and wrap
    assert frame_dp.global_batch(0, 2, 2, 6) == [0, 1, 2, 3] and frame_dp.global_batch(1, 2, 2, 6) == [4, 5, 0, 1]
    assert frame_dp.shard_frames([], 0, 2) == []


def test_flat_buffer_views_are_the_grads():
    P = 7
    leaves = {n: torch.randn(P, w, requires_grad=True) for n, w in frame_dp.LEAVES[:4]}
    buf = frame_dp.FlatGradBuffer(leaves)
    assert buf.flat.numel() == P * frame_dp.FLOATS_PER_GAUSSIAN
    loss = sum((t * (i + 1)).sum() for i, t in enumerate(leaves.values()))
    loss.backward()
    for i, (n, t) in enumerate(leaves.items()):
        assert t.grad.data_ptr() == buf.views[n].data_ptr() and t.grad.is_contiguous()
        assert torch.equal(buf.views[n], torch.full_like(t, float(i + 1)))
    assert buf.all_reduce() is None                 # no process group: no-op
    buf.zero()
    assert float(buf.flat.abs().sum()) == 0.0
    with pytest.raises(ValueError, match="must be"):
        frame_dp.FlatGradBuffer({"xyz": torch.zeros(P, 4)})


def _frame_loss(leaves, frame):
    """Stand-in for one frame's render + loss: any differentiable function of the shared leaves and the frame id."""
    g = torch.Generator().manual_seed(1000 + frame)
    w = {n: torch.randn(t.shape, generator=g) for n, t in leaves.items()}
    return sum((torch.sin(t * (1 + 0.1 * frame)) * w[n]).sum() for n, t in leaves.items())


def _make_leaves(P):
    g = torch.Generator().manual_seed(0)
    return {n: torch.randn(P, w, generator=g).requires_grad_(True) for n, w in frame_dp.LEAVES[:4]}


def _worker(rank, world, port, P, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        leaves = _make_leaves(P)
        buf = frame_dp.FlatGradBuffer(leaves)
        batch = frame_dp.global_batch(3, 2, world, num_frames=50)
        for f in frame_dp.shard_frames(batch, rank, world):
            _frame_loss(leaves, f).backward()
        buf.n_buckets = 2
        if rank == 0:
            works = buf.all_reduce_buckets()           # two asynchronous buckets: xyz first, then the rest
            assert len(works) == 2
        else:
            works = buf.all_reduce_buckets()
        buf.wait_bucket(0)
        buf.wait_all()
        assert buf.pending == []
        # the same sum as ONE collective (SOAR_DP_BUCKETS=1): a second buffer over copies of the leaves, same frames
        leaves1 = _make_leaves(P)
        buf1 = frame_dp.FlatGradBuffer(leaves1)
        buf1.n_buckets = 1
        for f in frame_dp.shard_frames(batch, rank, world):
            _frame_loss(leaves1, f).backward()
        assert len(buf1.all_reduce_buckets()) == 1
        buf1.wait_bucket(0)
        buf1.wait_all()
        assert torch.equal(buf1.flat, buf.flat)
        # split noise of the densifier: ranks with different RNG states must end up with rank 0's draw
        from soar_amd.densify import SurfelDensifier
        dens = SurfelDensifier.__new__(SurfelDensifier)
        dens.device = torch.device("cpu")
        noise = dens._split_noise(6, torch.Generator().manual_seed(100 + rank))
        acc, den = torch.full((P, 1), float(rank + 1)), torch.full((P, 1), 1.0)
        frame_dp.all_reduce_densification_stats(acc, den)
        acc5, rad = torch.full((5, P), float(rank + 1)), torch.arange(P, dtype=torch.float32) * (1 if rank else -1)
        frame_dp.all_reduce_densifier_stats(acc5, rad)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=buf.flat.numpy(), acc=acc.numpy(), den=den.numpy(),
                 acc5=acc5.numpy(), rad=rad.numpy(), noise=noise.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_all_reduce_equals_single_process(tmp_path):
    P, world = 33, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, P, str(tmp_path)), nprocs=world, join=True)
    # single-process reference: all frames of the same global batch on one set of leaves
    leaves = _make_leaves(P)
    buf = frame_dp.FlatGradBuffer(leaves)
    for f in frame_dp.global_batch(3, 2, world, num_frames=50):
        _frame_loss(leaves, f).backward()
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    np.testing.assert_array_equal(r0["flat"], r1["flat"])                       # every rank holds the same sum
    np.testing.assert_allclose(r0["flat"], buf.flat.numpy(), rtol=1e-5, atol=1e-5)
    assert np.abs(buf.flat.numpy()[P * 13:]).sum() == 0                          # leaves not registered stay zero
    np.testing.assert_array_equal(r0["acc"], np.full((P, 1), 3.0, np.float32))
    np.testing.assert_array_equal(r1["den"], np.full((P, 1), 2.0, np.float32))
    np.testing.assert_array_equal(r0["noise"], r1["noise"])                     # broadcast from rank 0
    np.testing.assert_array_equal(r0["noise"], torch.randn(6, 3, generator=torch.Generator().manual_seed(100)).numpy())
    for r in (r0, r1):                                                          # densifier statistics: sum and max
        np.testing.assert_array_equal(r["acc5"], np.full((5, P), 3.0, np.float32))
        np.testing.assert_array_equal(r["rad"], np.arange(P, dtype=np.float32))


def _worker8(rank, world, port, P, steps, buckets, out_dir):
    """What bench.py --gpus 8 does per rank, on CPU tensors: every step the rank renders ITS frames of the global batch (unequal
    cost: the ranks reach the collectives at different times and in a different order every step), issues the asynchronous bucket
    reductions, waits for the xyz bucket, 'updates' xyz, waits for the rest, updates the rest -- the order FrameStepPlan._run_eager
    keeps -- and sums the densifier's statistics at the end."""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        leaves = _make_leaves(P)
        buf = frame_dp.FlatGradBuffer(leaves)
        buf.n_buckets = buckets
        acc5, rad = torch.zeros(5, P), torch.zeros(P)
        seen = []
        for step in range(steps):
            batch = frame_dp.global_batch(step, 3, world, num_frames=50)            # 24 frames per step over a 50-frame video: wraps
            mine = frame_dp.shard_frames(batch, rank, world)
            seen.append(mine)
            buf.zero()
            for f in mine:
                time.sleep(0.001 * ((f * 7 + rank) % 5))                            # frames of unequal cost
                _frame_loss(leaves, f).backward()
                acc5[:, f % P] += 1.0 + f
                rad[f % P] = max(float(rad[f % P]), float(f))
            buf.all_reduce_buckets()
            buf.wait_bucket(0)
            with torch.no_grad():
                leaves["xyz"] -= 1e-3 * buf.views["xyz"]                            # positions behind the first bucket ...
            buf.wait_all()
            with torch.no_grad():
                for n in ("rot", "scales", "colors"):
                    leaves[n] -= 1e-3 * buf.views[n]                                # ... the rest behind the second
        frame_dp.all_reduce_densifier_stats(acc5, rad)                                # (what SurfelDensifier.sync_stats does before it plans)
        # SurfelDensifier(spatial_order=True) re-orders the model after every densification by a permutation that is a pure function of
        # the replicated positions (densify.spatial_permutation: Morton keys, stable sort): every rank must arrive at the SAME one
        from soar_amd.densify import spatial_permutation
        perm = spatial_permutation(leaves["xyz"] * 0.05)                              # (several points per 2 cm cell: ties, kept in row order)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), acc5=acc5.numpy(), rad=rad.numpy(), seen=np.array(seen), perm=perm.numpy(),
                 **{n: t.detach().numpy() for n, t in leaves.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("buckets", [2, 1])
def test_eight_rank_gloo_training_steps_equal_single_process(tmp_path, buckets):
    """world_size 8 (BASELINE configs C4 / C5: 8 x MI355X frame-DP) on CPU: partition of every global batch, both gradient buckets,
    parameter updates applied behind each bucket, densifier statistics -- so that the first real `bench.py --gpus 8` run cannot
    fail on partition or hand-over logic.  Three steps; every rank must end with the single-process parameters."""
    P, world, steps = 19, 8, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker8, args=(world, port, P, steps, buckets, str(tmp_path)), nprocs=world, join=True)
    leaves = _make_leaves(P)
    buf = frame_dp.FlatGradBuffer(leaves)
    acc5, rad = torch.zeros(5, P), torch.zeros(P)
    for step in range(steps):
        buf.zero()
        for f in frame_dp.global_batch(step, 3, world, num_frames=50):
            _frame_loss(leaves, f).backward()
            acc5[:, f % P] += 1.0 + f
            rad[f % P] = max(float(rad[f % P]), float(f))
        with torch.no_grad():
            for n in leaves:
                leaves[n] -= 1e-3 * buf.views[n]
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for step in range(steps):                                                       # every frame of every batch exactly once
        got = sorted(int(f) for r in ranks for f in r["seen"][step])
        assert got == sorted(frame_dp.global_batch(step, 3, world, num_frames=50))
    for r in ranks:
        for n in leaves:
            np.testing.assert_array_equal(r[n], ranks[0][n])                        # replicas stay bit-identical ...
            np.testing.assert_allclose(r[n], leaves[n].detach().numpy(), rtol=2e-5, atol=2e-6)     # ... and follow the one-process run
    for r in ranks:                                                                 # densifier statistics: sum and max over the ranks
        np.testing.assert_allclose(r["acc5"], acc5.numpy(), rtol=1e-6)
        np.testing.assert_array_equal(r["rad"], rad.numpy())
        # the spatial re-ordering a densifying job applies: one permutation, the same on every rank
        np.testing.assert_array_equal(r["perm"], ranks[0]["perm"])
        assert sorted(r["perm"].tolist()) == list(range(P))


def _bench_line(cmd, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("buckets", ["1", "2"])
def test_bench_dry_run_with_eight_ranks_keeps_the_one_line_contract(buckets):
    """`bench.py --gpus 8 --dry-run`: bench.py's OWN launcher, rendezvous, frame sharding, flat-buffer all-reduce,
    timed region, per-rank / per-bucket aggregation and one-line contract with eight gloo ranks on the CPU and stand-in frames -- the
    functions the real ranks run (bench.timed_region / rank_diagnostics / job_elapsed / launch_ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SOAR_DP_BUCKETS"] = buckets
    d = _bench_line(["bench.py", "--gpus", "8", "--dry-run", "--steps", "3", "--warmup", "1"], env)
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["data"].startswith("dry-run") and d["config"]["parallelism"] == "frame-dp8" and d["config"]["replicas_identical"] is True
    assert abs(d["value"] - 8 * 4 * 1000.0 / d["ms_per_step"]) <= d["value"] * (0.00051 / d["ms_per_step"] + 1e-6)
    rk = d["ranks"]
    assert len(rk["per_rank_ms_per_step"]) == 8 and all(v > 0 for v in rk["per_rank_ms_per_step"]) and rk["imbalance_max_over_mean"] >= 1.0
    assert len(rk["bucket_wait_us_per_rank"]) == 8 and rk["buckets"] == int(buckets)
    assert max(rk["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.001           # the job's step is the slowest rank's (plus the barrier)
    if buckets == "2":
        assert all(w[1] > 0 for w in rk["bucket_wait_us_per_rank"])


def test_bench_dry_run_under_the_drivers_launcher():
    """... and started the way the driver starts N > 1: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _bench_line(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                     "bench.py", "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], env)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "frame-dp2" and len(d["ranks"]["per_rank_ms_per_step"]) == 2
    # --gpus must describe the job: a mismatch prints no line
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, cwd=root,
                       env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_flat_buffer_refuses_stale_views():
    P = 5
    leaves = {n: torch.randn(P, w, requires_grad=True) for n, w in frame_dp.LEAVES[:4]}
    buf = frame_dp.FlatGradBuffer(leaves)
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-3)
    opt.zero_grad()                                    # set_to_none=True: drops the aliases
    with pytest.raises(RuntimeError, match="no longer the view"):
        buf.all_reduce()
    buf.zero()                                         # re-attaches
    buf.all_reduce()
    sum(t.sum() for t in leaves.values()).backward()
    assert float(buf.flat[: P * 13].sum()) == P * 13
    opt.zero_grad(set_to_none=False)                   # keeps the aliases
    buf.all_reduce_buckets()
    buf.invalidate("densified")
    with pytest.raises(RuntimeError, match="stale"):
        buf.all_reduce()
