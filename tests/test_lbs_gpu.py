"""GPU parity tests of the LBS kernels (through the C ABI) against oracle/lbs_oracle.py (torch CPU + autograd)."""
import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo
from soar_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(P=4000, seed=0, V=syn.SMPLX_NUM_VERTS):
    s = syn.make_surfels(P, seed)
    bm = syn.make_body_model(seed, V=V)
    poses = syn.make_pose_sequence(8, seed)
    betas = torch.cat([poses["betas"], poses["expression"][:1]], dim=1)
    A_cano = lo.joint_transforms(betas, torch.zeros(1, 165), bm.v_template[None], bm.shapedirs, bm.J_regressor, bm.parents,
                                 torch.tensor([[0.0, 0.3, 0.0]]))
    A_live = lo.joint_transforms(betas, poses["full_pose"][3:4], bm.v_template[None], bm.shapedirs, bm.J_regressor,
                                 bm.parents, poses["transl"][3:4])
    cano2live = torch.matmul(A_live, torch.linalg.inv(A_cano))[0]
    return s, bm, cano2live


def _rows_equal_or_proved_ties(idx, i_ref, queries, verts):
    """Neighbour index sets are exact except where two candidates tie within fp32 rounding of d2 at the K-th place -- and every row
    that differs is shown to BE such a tie: the distances of the members only one side picked, recomputed in float64, lie within fp32
    rounding of the squared-distance expression (a few ulp of the K-th distance) of each other.  -> mask of the identical rows."""
    same_rows = (np.sort(idx, 1) == np.sort(i_ref, 1)).all(1)
    q64, v64 = queries.astype(np.float64), verts.astype(np.float64)
    for row in np.nonzero(~same_rows)[0]:
        mine, theirs = set(idx[row].tolist()), set(i_ref[row].tolist())
        only_mine, only_theirs = sorted(mine - theirs), sorted(theirs - mine)
        assert len(only_mine) == len(only_theirs) and len(mine) == idx.shape[1], (row, only_mine, only_theirs)
        d2 = lambda ids: ((v64[ids] - q64[row]) ** 2).sum(1)
        kth = max(d2(sorted(theirs)).max(), d2(sorted(mine)).max())
        gap = np.abs(np.sort(d2(only_mine)) - np.sort(d2(only_theirs))).max()
        # fp32 evaluation of |q - v|^2 for coordinates of magnitude c carries ~4 ulp(c^2) of rounding: allow that much, no more
        c2 = max((q64[row] ** 2).sum(), (v64[sorted(mine | theirs)] ** 2).sum(1).max())
        assert gap <= 8 * np.finfo(np.float32).eps * max(kth, c2), (row, gap, kth, only_mine, only_theirs)
    return same_rows


def test_knn_blend_weights_match_oracle():
    from soar_amd import lbs
    s, bm, _ = _setup(P=3000)
    w_ref = lo.query_weights(s.xyz, bm.v_template, bm.lbs_weights)
    d_ref, i_ref = lo.knn_brute(s.xyz, bm.v_template, 30)
    w, idx = lbs.knn_blend_weights(s.xyz.to(DEV), bm.v_template.to(DEV), bm.lbs_weights.to(DEV), return_idx=True)
    idx = idx.cpu().numpy()
    i_ref = i_ref.numpy()
    same_rows = _rows_equal_or_proved_ties(idx, i_ref, s.xyz.numpy(), bm.v_template.numpy())
    assert same_rows.mean() > 0.99            # (ties are rare; nothing depends on this number)
    np.testing.assert_allclose(w.cpu().numpy()[same_rows], w_ref.numpy()[same_rows], rtol=1e-4, atol=1e-6)
    # the rows with a tie blend a vertex at the same distance to the last bits: their weights agree with the oracle's wherever
    # both picked the vertex, and sum to one like every row
    np.testing.assert_allclose(w.sum(1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize("use_T,use_off", [(False, False), (True, False), (True, True)])
def test_warp_forward_backward_match_oracle(use_T, use_off):
    from soar_amd import lbs
    s, bm, cano2live = _setup(P=3000, seed=1)
    w = lo.query_weights(s.xyz, bm.v_template, bm.lbs_weights)
    T = lo.axis_perm_matrix("+z,+x,+y") if use_T else None
    gen = torch.Generator().manual_seed(5)
    off = 0.01 * torch.randn(s.xyz.shape, generator=gen) if use_off else None
    # un-normalised quaternions on purpose: quaternion_to_matrix divides by |q|^2
    rot0 = s.rot * (0.5 + torch.rand(s.rot.shape[0], 1, generator=gen))
    xyz_c = s.xyz.clone().requires_grad_(True)
    rot_c = rot0.clone().requires_grad_(True)
    p_ref, q_ref, mats_ref = lo.warp(xyz_c, rot_c, w, cano2live, off, T)
    gp = torch.randn(p_ref.shape, generator=gen)
    gq = torch.randn(q_ref.shape, generator=gen)
    ((p_ref * gp).sum() + (q_ref * gq).sum()).backward()

    xyz_g = s.xyz.to(DEV).requires_grad_(True)
    rot_g = rot0.to(DEV).requires_grad_(True)
    p, q = lbs.lbs_warp(xyz_g, rot_g, w.to(DEV), cano2live.to(DEV), off.to(DEV) if use_off else None,
                        T.to(DEV) if use_T else None)
    ((p * gp.to(DEV)).sum() + (q * gq.to(DEV)).sum()).backward()
    np.testing.assert_allclose(p.detach().cpu().numpy(), p_ref.detach().numpy(), rtol=1e-5, atol=2e-6)
    # quaternion sign is standardised (non-negative real part) on both sides
    np.testing.assert_allclose(q.detach().cpu().numpy(), q_ref.detach().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(xyz_g.grad.cpu().numpy(), xyz_c.grad.numpy(), rtol=1e-4, atol=1e-5)
    scale = rot_c.grad.abs().max().item()
    assert (rot_g.grad.cpu() - rot_c.grad).abs().max().item() <= 1e-4 * scale
    if not use_T and not use_off:
        mats = lbs.point_transforms(s.xyz.to(DEV), rot0.to(DEV), w.to(DEV), cano2live.to(DEV))
        np.testing.assert_allclose(mats.cpu().numpy(), mats_ref[0].detach().numpy(), rtol=1e-5, atol=2e-6)


def test_warp_ragged_and_empty():
    from soar_amd import lbs
    s, bm, cano2live = _setup(P=300, seed=2, V=512)
    w = lo.query_weights(s.xyz, bm.v_template, bm.lbs_weights)
    for P in (1, 255, 257):
        p, q = lbs.lbs_warp(s.xyz[:P].to(DEV), s.rot[:P].to(DEV), w[:P].to(DEV), cano2live.to(DEV))
        p_ref, q_ref, _ = lo.warp(s.xyz[:P], s.rot[:P], w[:P], cano2live)
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(q.cpu().numpy(), q_ref.numpy(), rtol=1e-4, atol=2e-5)
    p, q = lbs.lbs_warp(s.xyz[:0].to(DEV), s.rot[:0].to(DEV), w[:0].to(DEV), cano2live.to(DEV))
    assert p.shape == (0, 3) and q.shape == (0, 4)


@pytest.mark.parametrize("n,P", [(1, 300), (3, 257), (4, 3000), (6, 1001)])
def test_warp_of_all_frames_in_one_launch_is_the_per_frame_warp(n, P):
    """soar_lbs_warp_forward_batch / soar_lbs_warp_backward_sum (the step plan's forms: every frame of a step in one launch, the
    backward one adding the frames' gradients in frame order and summing two more per-frame blocks on the way) against the
    per-frame entry points, bit for bit."""
    from soar_amd import hip_lib
    L = hip_lib.lib()
    ptr = hip_lib.ptr
    s, bm, cano2live = _setup(P=P, seed=3, V=2048)
    gen = torch.Generator().manual_seed(11)
    w = lo.query_weights(s.xyz, bm.v_template, bm.lbs_weights).to(DEV).contiguous()
    J = w.shape[1]
    mats = torch.stack([cano2live + 0.01 * k * torch.randn(cano2live.shape, generator=gen) for k in range(n)]).to(DEV).contiguous()
    xyz = s.xyz.to(DEV).contiguous()
    rot = (s.rot * (0.5 + torch.rand(P, 1, generator=gen))).to(DEV).contiguous()
    g_p = torch.randn(n, P, 3, generator=gen).to(DEV)
    g_q = torch.randn(n, P, 4, generator=gen).to(DEV)
    extra = [torch.randn(n, P, 3, generator=gen).to(DEV), torch.randn(n, P, 1, generator=gen).to(DEV)]
    stream = torch.cuda.current_stream().cuda_stream

    p_all = torch.full((n, P, 3), float("nan"), device=DEV)
    q_all = torch.full((n, P, 4), float("nan"), device=DEV)
    assert L.soar_lbs_warp_forward_batch(ptr(xyz), ptr(rot), ptr(w), ptr(mats), n, P, J, ptr(p_all), ptr(q_all), stream) == 0
    d_p = torch.full((P, 3), float("nan"), device=DEV)
    d_q = torch.full((P, 4), float("nan"), device=DEV)
    sums = [torch.full((P, 3), float("nan"), device=DEV), torch.full((P, 1), float("nan"), device=DEV)]
    src = torch.tensor([e.data_ptr() for e in extra], dtype=torch.int64)
    dst = torch.tensor([e.data_ptr() for e in sums], dtype=torch.int64)
    width = torch.tensor([3, 1], dtype=torch.int32)
    assert L.soar_lbs_warp_backward_sum(ptr(xyz), ptr(rot), ptr(w), ptr(mats), n, P, J, ptr(g_p), ptr(g_q), ptr(d_p), ptr(d_q),
                                        2, src.data_ptr(), dst.data_ptr(), width.data_ptr(), stream) == 0
    want_dp = torch.zeros(P, 3, device=DEV)
    want_dq = torch.zeros(P, 4, device=DEV)
    for f in range(n):
        p_f = torch.empty(P, 3, device=DEV)
        q_f = torch.empty(P, 4, device=DEV)
        assert L.soar_lbs_warp_forward(ptr(xyz), ptr(rot), ptr(w), ptr(mats[f]), None, None, P, J, ptr(p_f), ptr(q_f), None, stream) == 0
        assert torch.equal(p_all[f], p_f) and torch.equal(q_all[f], q_f), f
        dp_f = torch.empty(P, 3, device=DEV)
        dq_f = torch.empty(P, 4, device=DEV)
        assert L.soar_lbs_warp_backward(ptr(xyz), ptr(rot), ptr(w), ptr(mats[f]), None, P, J, ptr(g_p[f]), ptr(g_q[f]), ptr(dp_f),
                                        ptr(dq_f), stream) == 0
        want_dp = want_dp + dp_f
        want_dq = want_dq + dq_f
    torch.cuda.synchronize()
    assert torch.equal(d_p, want_dp) and torch.equal(d_q, want_dq)
    for e, got in zip(extra, sums):
        want = e[0].clone()
        for f in range(1, n):
            want = want + e[f]
        assert torch.equal(got, want)


def test_dist2_knn3_matches_oracle():
    from soar_amd import lbs
    gen = torch.Generator().manual_seed(3)
    pts = torch.randn(2500, 3, generator=gen)
    pts[10] = pts[11]                                   # duplicate point: distance 0 counts (self excluded by index)
    ref = lo.dist2_knn3(pts.numpy())
    got = lbs.dist2_knn3(pts.to(DEV)).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-9)
    assert got[10] < ref.mean() and lbs.distCUDA2 is lbs.dist2_knn3


def test_full_size_properties_C1():
    """10k canonical Gaussians, SMPL-X-sized vertex set: identity joints leave the cloud unchanged; weights are a
    partition of unity; a rigid global transform moves every point rigidly."""
    from soar_amd import lbs
    s = syn.make_surfels(10000, 4)
    bm = syn.make_body_model(4)
    w = lbs.knn_blend_weights(s.xyz.to(DEV), bm.v_template.to(DEV), bm.lbs_weights.to(DEV))
    assert torch.allclose(w.sum(1), torch.ones(10000, device=DEV), atol=1e-5) and (w >= 0).all()
    eye = torch.eye(4, device=DEV)[None].repeat(55, 1, 1)
    p, q = lbs.lbs_warp(s.xyz.to(DEV), s.rot.to(DEV), w, eye)
    assert torch.allclose(p.cpu(), s.xyz, atol=1e-6)
    assert torch.allclose(lo.quaternion_to_matrix(q.cpu()), lo.quaternion_to_matrix(s.rot), atol=1e-5)
    Rg = lo.batch_rodrigues(torch.tensor([[0.3, -0.2, 0.5]]))[0]
    G = torch.eye(4)
    G[:3, :3] = Rg
    G[:3, 3] = torch.tensor([0.1, 0.2, 0.3])
    p, q = lbs.lbs_warp(s.xyz.to(DEV), s.rot.to(DEV), w, G.to(DEV)[None].repeat(55, 1, 1))
    assert torch.allclose(p.cpu(), s.xyz @ Rg.T + G[:3, 3], atol=2e-6)
    assert torch.allclose(lo.quaternion_to_matrix(q.cpu()), Rg @ lo.quaternion_to_matrix(s.rot), atol=1e-5)


def test_knn_grid_built_once_matches_one_call_form():
    """lbs.KnnGrid (vertex grid built once, queried per step) == knn_blend_weights (grid rebuilt per call), for several
    query sets against the same grid."""
    from soar_amd import lbs
    bm = syn.make_body_model(0)
    grid = lbs.KnnGrid(bm.v_template.to(DEV), bm.lbs_weights.to(DEV))
    for seed, P in ((0, 3000), (5, 777), (9, 64)):
        x = syn.make_surfels(P, seed).xyz.to(DEV)
        w_ref, i_ref = lbs.knn_blend_weights(x, bm.v_template.to(DEV), bm.lbs_weights.to(DEV), return_idx=True)
        w, i = grid.query(x, return_idx=True)
        assert torch.equal(w, w_ref) and torch.equal(i, i_ref)


def test_knn_query_order_reuse_is_exact():
    """KnnGrid.query keeps the cell-grouping order of the queries between calls: with moved points and a stale order the
    weights and neighbour sets are those of a freshly sorted query."""
    from soar_amd import lbs
    bm = syn.make_body_model(0)
    v, w = bm.v_template.to(DEV), bm.lbs_weights.to(DEV)
    grid = lbs.KnnGrid(v, w)
    x0 = syn.make_surfels(3000, 2).xyz.to(DEV)
    grid.query(x0)                                              # sorts and stores the order
    g = torch.Generator().manual_seed(3)
    for step in range(1, 12):                                   # crosses a re-sort (RESORT_EVERY = 8)
        x = x0 + 0.02 * step * torch.randn(x0.shape, generator=g).to(DEV)
        w_got, i_got = grid.query(x, return_idx=True)
        w_ref, i_ref = lbs.knn_blend_weights(x, v, w, return_idx=True)
        # the same exact search with the same tie rule ((distance, grid position) order) whatever the order of the queries: a row
        # that differs has to be proved a tie at the K-th place like the oracle test's, the others carry the same weights
        same = _rows_equal_or_proved_ties(i_got.cpu().numpy(), i_ref.cpu().numpy(), x.cpu().numpy(), v.cpu().numpy())
        torch.testing.assert_close(w_got[torch.from_numpy(same).to(DEV)], w_ref[torch.from_numpy(same).to(DEV)], rtol=1e-6, atol=1e-7)


def test_knn_follower_is_the_full_search_bit_for_bit_under_motion():
    """lbs.KnnFollower (neighbour sets kept between optimizer steps, exactness certificate = displacement below half the gap to
    the 31st neighbour, seeded exact search otherwise) == KnnGrid.query at the same positions, bit for bit, over steps of
    different sizes: small ones (most queries certified), large ones (all searched again) and none at all."""
    from soar_amd import lbs
    bm = syn.make_body_model(0)
    v, w = bm.v_template.to(DEV), bm.lbs_weights.to(DEV)
    grid = lbs.KnnGrid(v, w)
    x = syn.make_surfels(6000, 4).xyz.to(DEV)
    fol = lbs.KnnFollower(grid, x.shape[0])
    g = torch.Generator().manual_seed(11)
    searched_before = 0
    fractions = []
    for step, sigma in enumerate([0.0, 1e-5, 1e-5, 1e-5, 3e-3, 1e-5, 1e-5, 0.0, 5e-2, 1e-4, 1e-4]):
        x = x + sigma * torch.randn(x.shape, generator=g).to(DEV)
        got = fol(x).clone()
        want = grid.query(x)
        assert torch.equal(got, want), (step, sigma, float((got - want).abs().max()))
        n = int(fol.searched.item())
        fractions.append((n - searched_before) / x.shape[0])
        searched_before = n
    # step 0 is the full search; step 1 measures every gap (all searched); then small steps are mostly certified, large ones not
    assert fractions[1] == 1.0 and fractions[2] < 0.3 and fractions[3] < 0.4 and fractions[4] > 0.9 and fractions[7] < 0.01, fractions


@pytest.mark.parametrize("P", [1, 5, 33, 1999, 6007])
def test_knn_follower_ragged_query_counts(P):
    """Query counts that fill neither a pair of the certificate launch, nor the four slots of a blend wavefront, nor a work list's group
    of 32: the follower still equals the full search bit for bit over refreshes of every tier, and leaves its work lists empty (the
    searched counts of later refreshes would be off otherwise)."""
    from soar_amd import lbs
    bm = syn.make_body_model(0)
    v, w = bm.v_template.to(DEV), bm.lbs_weights.to(DEV)
    grid = lbs.KnnGrid(v, w)
    x = syn.make_surfels(max(P, 64), 9).xyz[:P].contiguous().to(DEV)
    fol = lbs.KnnFollower(grid, P)
    g = torch.Generator().manual_seed(17)
    before = 0
    for step, sigma in enumerate([0.0, 1e-5, 2e-5, 4e-3, 1e-5, 0.0, 1e-4]):
        x = x + sigma * torch.randn(x.shape, generator=g).to(DEV)
        got = fol(x).clone()
        assert torch.equal(got, grid.query(x)), (P, step, sigma)
        n = int(fol.searched.item())
        assert 0 <= n - before <= P, (P, step, n, before)
        if step == 1:
            assert n - before == P          # the first refresh after a full search measures every gap
        if step == 5:
            assert n - before <= P // 50    # nothing moved: certified (but for near-ties at the K-th place), and no stale entry left on a list
        before = n


def test_knn_follower_with_ties_and_dense_clusters():
    """Duplicated vertices (exact distance ties at the K-th place, thousands of candidates in one cell): the follower still equals
    the full search bit for bit -- ties are never certified and the seeded search applies the full search's tie rule (grid order)."""
    from soar_amd import lbs
    g = torch.Generator().manual_seed(31)
    base = torch.randn(2500, 3, generator=g) * 0.004
    far = torch.randn(300, 3, generator=g) * torch.tensor([0.3, 0.9, 0.2])
    verts = torch.cat([base, base.clone(), far]).to(DEV)
    w = torch.rand(verts.shape[0], 55, generator=g)
    w = (w / w.sum(1, keepdim=True)).to(DEV)
    grid = lbs.KnnGrid(verts, w)
    x = (base[torch.randint(0, 2500, (2000,), generator=g)] + 0.001 * torch.randn(2000, 3, generator=g)).to(DEV)
    fol = lbs.KnnFollower(grid, x.shape[0])
    for sigma in (0.0, 1e-5, 1e-5, 2e-4, 1e-2):
        x = x + sigma * torch.randn(x.shape, generator=g).to(DEV)
        assert torch.equal(fol(x), grid.query(x)), sigma


def test_knn_follower_survives_queries_that_are_not_numbers():
    """A position that diverged to NaN (or infinity) under the optimizer fails both certificates and goes to the seeded search,
    whose ball never holds a vertex for it: the search loop is bounded, such a query gets NaN weights (what the full search's
    arithmetic gives it), every other query still equals the full search bit for bit, and the query is served again once it is a
    number.  (Run under a watchdog: the failure mode this guards against is a wavefront that spins for ever.)"""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import torch, sys
        sys.path.insert(0, %r)
        from soar_amd import lbs, synthetic as syn
        DEV = "cuda:0"
        bm = syn.make_body_model(0)
        grid = lbs.KnnGrid(bm.v_template.to(DEV), bm.lbs_weights.to(DEV))
        x = syn.make_surfels(3000, 4).xyz.to(DEV)
        fol = lbs.KnnFollower(grid, x.shape[0])
        fol(x); fol(x)
        bad = x.clone()
        bad[17] = float("nan"); bad[1234, 1] = float("nan"); bad[2999] = float("inf")
        got = fol(bad).clone()
        torch.cuda.synchronize()
        want = grid.query(x)
        ok = torch.ones(3000, dtype=torch.bool, device=DEV); ok[[17, 1234, 2999]] = False
        assert torch.equal(got[ok], want[ok])
        assert torch.isnan(got[17]).all() and torch.isnan(got[1234]).all() and not torch.isfinite(got[2999]).all()
        again = fol(x).clone()
        assert torch.equal(again, want)
        print("ok")
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_knn_state_entries_refuse_fewer_vertices_than_the_state_keeps():
    from soar_amd import lbs
    g = torch.Generator().manual_seed(2)
    verts = torch.randn(31, 3, generator=g).to(DEV)
    w = torch.rand(31, 55, generator=g)
    grid = lbs.KnnGrid(verts, (w / w.sum(1, keepdim=True)).to(DEV))
    x = torch.randn(100, 3, generator=g).to(DEV)
    assert torch.isfinite(grid.query(x)).all()                  # K = 30 <= V: the plain search serves it
    fol = lbs.KnnFollower(grid, 100)
    with pytest.raises(RuntimeError, match="keeps 32 vertices"):
        fol(x)


def test_smplx_joint_chain_kernel_matches_reference_lbs_goldens():
    """soar_smplx_joint_mats (Rodrigues + 55-joint chain + transl + right product, all frames in one launch) == the
    reference's lbs() transforms A (tests/golden/smplx_joint_transforms.npz, generated by the reference's own smplx code)
    and the host JointTransformer, which the CPU suite pins on the same goldens."""
    import os
    from soar_amd import smplx_joints as sj
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "smplx_joint_transforms.npz")))
    t = lambda k: torch.from_numpy(g[k])
    jt = sj.JointTransformer(t("v_template"), t("shapedirs"), t("J_regressor"), torch.from_numpy(g["parents"]))
    dev = torch.device("cuda:0")
    A = jt.hip(t("betas").to(dev), t("pose").to(dev), t("transl").to(dev))
    np.testing.assert_allclose(A.cpu().numpy(), g["A_with_transl"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(jt.hip(t("betas").to(dev), t("pose").to(dev)).cpu().numpy(), g["A"], rtol=0, atol=5e-6)
    # shared betas row, right-hand product, a 300-frame batch with large rotations and a zero pose (the 1e-8 quirk)
    gen = torch.Generator().manual_seed(3)
    pose = torch.randn(300, 165, generator=gen) * 1.5
    pose[7] = 0.0
    transl = torch.randn(300, 3, generator=gen)
    betas = t("betas")[:1]
    right = torch.linalg.inv(jt(betas, torch.zeros(1, 165), torch.tensor([[0.0, 0.3, 0.0]])))[0]
    want = torch.matmul(jt(betas, pose, transl), right)
    got = jt.hip(betas.to(dev), pose.to(dev), transl.to(dev), right=right.to(dev)).cpu()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert jt.hip(betas.to(dev), pose[:0].to(dev)).shape == (0, 55, 4, 4)
    with pytest.raises(RuntimeError, match="HIP devices only"):
        jt.hip(betas, pose, transl)
    with pytest.raises(ValueError):
        jt.hip(t("betas")[:3].to(dev), pose.to(dev))


def test_knn_ties_across_candidate_chunks():
    """Exact distance ties at the K-th place in a box with more candidates than one staged chunk (dense, duplicated vertices):
    exactly K neighbours are taken -- the tie counter runs on through the chunks -- their distances are the brute-force ones
    and the blend is the blend of the reported neighbour list."""
    from soar_amd import lbs
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(31)
    base = torch.randn(2500, 3, generator=g) * 0.004                      # thousands of vertices in one grid cell ...
    far = torch.randn(300, 3, generator=g) * torch.tensor([0.3, 0.9, 0.2])
    verts = torch.cat([base, base.clone(), far])                          # ... every one of them twice
    w = torch.rand(verts.shape[0], 55, generator=g)
    w = w / w.sum(1, keepdim=True)
    xyz = base[torch.randint(0, 2500, (3000,), generator=g)] + 0.001 * torch.randn(3000, 3, generator=g)
    out, idx = lbs.KnnGrid(verts.to(dev), w.to(dev)).query(xyz.to(dev), return_idx=True)
    d = torch.cdist(xyz.to(dev).double(), verts.to(dev).double())
    dk = torch.topk(d, 30, dim=1, largest=False).values
    got = torch.gather(d, 1, idx.long())
    assert float((torch.sort(got, 1).values - dk).abs().max()) < 1e-7
    assert all(len(set(r)) == 30 for r in idx[:200].tolist())              # 30 distinct neighbours
    wi = 1.0 / got.clamp(0.0001, 1.0)
    wi = wi / wi.sum(-1, keepdim=True)
    own = (wi[..., None] * w.to(dev).double()[idx.long()]).sum(-2)
    assert float((out.double() - own).abs().max()) < 2e-6
