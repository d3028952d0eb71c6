"""Generates tests/golden/reference_densify.npz by running the REFERENCE's own densification / pruning methods
(TS/geometry/surfel_base.py:850-1136: add_densification_stats, adaptive_prune, adaptive_densify, densify_and_clone,
densify_and_split, densification_postfix, cat_tensors_to_optimizer, prune_points, _prune_optimizer) on a seeded state.

The module itself imports threestudio / pytorch3d / pymeshlab, which are not installed, so the methods are compiled from
the reference's file with `ast` at generation time and bound to a stub object that carries exactly the attributes they read
(only inputs and outputs are stored; nothing of the reference is kept).  The methods hard-code device="cuda": the generator
runs them on the CPU by redirecting that keyword for the duration of the call.  Run here (needs /root/reference):

    python tests/golden/make_densify_golden.py
"""
import ast
import os
import sys

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SOAR_REFERENCE", "/root/reference") + "/soar/threestudio-soar"
METHODS = ("add_densification_stats", "adaptive_prune", "adaptive_densify", "densify_and_clone", "densify_and_split",
           "densification_postfix", "cat_tensors_to_optimizer", "prune_points", "_prune_optimizer", "reset_opacity",
           "replace_tensor_to_optimizer")


def _functions(path, names):
    tree = ast.parse(open(path).read())
    found = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            found.append(node)
    return ast.Module(body=found, type_ignores=[])


class _CpuRedirect:
    """device="cuda" -> cpu for the factory functions the reference methods call; torch.cuda.empty_cache -> no-op."""
    NAMES = ("zeros", "ones", "empty", "full", "tensor")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        for n, fn in self.saved.items():
            setattr(torch, n, (lambda f: lambda *a, **k: f(*a, **{**k, "device": "cpu"} if "device" in k else k))(fn))
        self.saved_empty_cache = torch.cuda.empty_cache
        torch.cuda.empty_cache = lambda: None
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(torch, n, fn)
        torch.cuda.empty_cache = self.saved_empty_cache


def make_stub():
    ns = {"torch": torch, "nn": nn, "np": np}
    exec(compile(_functions(os.path.join(REF, "utils", "general_utils.py"), ("build_rotation", "inverse_sigmoid")), "general_utils", "exec"), ns)
    exec(compile(_functions(os.path.join(REF, "geometry", "surfel_base.py"), METHODS), "surfel_base", "exec"), ns)
    attrs = {m: ns[m] for m in METHODS}
    # the properties and activations the methods read (surfel_base.py:135-143, 442-472)
    attrs.update(get_scaling=property(lambda s: torch.exp(s._scaling)), get_opacity=property(lambda s: torch.sigmoid(s._opacity)),
                 get_xyz=property(lambda s: s._xyz), scaling_inverse_activation=staticmethod(torch.log))
    return type("ReferenceSurfelStub", (), attrs)()


PARAMS = ("xyz", "f_dc", "f_rest", "color", "opacity", "scaling", "rotation")
ATTR = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", color="_colors", opacity="_opacity", scaling="_scaling",
            rotation="_rotation")


def seeded_state(P, seed, surface):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    scaling = torch.log(torch.rand(P, 3, generator=g) * 0.016 + 1e-3)
    scaling[: P // 50] = torch.log(torch.tensor(0.9))                   # oversized -> pruned by scale_max
    scaling[P // 50: P // 25, :2] = torch.log(torch.tensor(3e-6))       # degenerate area -> pruned
    if surface:
        scaling[:, 2] = -1e10
    return dict(xyz=r(P, 3) * 0.5, f_dc=r(P, 1, 3), f_rest=r(P, 3, 3), color=torch.rand(P, 3, generator=g),
                opacity=r(P, 1) * 2.0, scaling=scaling, rotation=r(P, 4))


def run_case(name, P, seed, surface, do_prune, out):
    stub = make_stub()
    st = seeded_state(P, seed, surface)
    for k, v in st.items():
        setattr(stub, ATTR[k], nn.Parameter(v.clone()))
        out[f"{name}_in_{k}"] = v.numpy().copy()
    stub.optimizer = torch.optim.Adam([{"params": [getattr(stub, ATTR[k])], "lr": 1e-3, "name": k} for k in PARAMS], lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(seed + 1)
    # one Adam step so that exp_avg / exp_avg_sq exist and differ per row
    for k in PARAMS:
        getattr(stub, ATTR[k]).grad = torch.randn(getattr(stub, ATTR[k]).shape, generator=g) * 1e-2
    stub.optimizer.step()
    for k in PARAMS:
        out[f"{name}_step_{k}"] = getattr(stub, ATTR[k]).detach().numpy().copy()
        s = stub.optimizer.state[getattr(stub, ATTR[k])]
        out[f"{name}_m_{k}"], out[f"{name}_v_{k}"] = s["exp_avg"].numpy().copy(), s["exp_avg_sq"].numpy().copy()
    stub.percent_dense, stub.config = 0.01, torch.tensor([1.0 if surface else 0.0, 1.0, 1.0, 0.0])
    with _CpuRedirect():
        z = lambda *s: torch.zeros(*s)
        stub.xyz_gradient_accum, stub.scale_gradient_accum, stub.rot_gradient_accum = z(P, 1), z(P, 1), z(P, 1)
        stub.opac_gradient_accum, stub.denom, stub.max_radii2D = z(P, 1), z(P, 1), z(P)
        n_views = 3
        for v in range(n_views):                        # update_states' per-view loop (surfel_base.py:1208-1216)
            radii = torch.randint(0, 40, (P,), generator=g, dtype=torch.int32) * (torch.rand(P, generator=g) > 0.3)
            vis = radii > 0
            view = torch.zeros(P, 3)
            view.grad = torch.randn(P, 3, generator=g) * 4e-4
            stub._scaling.grad = torch.randn(P, 3, generator=g) * 1e-7
            out[f"{name}_view{v}_radii"], out[f"{name}_view{v}_grad2d"] = radii.numpy().copy(), view.grad.numpy().copy()
            out[f"{name}_view{v}_scaling_grad"] = stub._scaling.grad.numpy().copy()
            with torch.no_grad():                       # update_states is @torch.no_grad()
                stub.max_radii2D = torch.max(stub.max_radii2D, radii.float())
                stub.add_densification_stats(view, vis)
        for k in ("xyz_gradient_accum", "scale_gradient_accum", "rot_gradient_accum", "opac_gradient_accum", "denom", "max_radii2D"):
            out[f"{name}_acc_{k}"] = getattr(stub, k).numpy().copy()
        extent, max_grad, min_opac = 1.3, 2e-4, 0.1
        out[f"{name}_scalars"] = np.array([extent, max_grad, min_opac, stub.percent_dense, float(surface), float(do_prune)], np.float32)
        if do_prune:
            with torch.no_grad():
                stub.adaptive_prune(min_opac, extent)
            out[f"{name}_after_prune_xyz"] = stub._xyz.detach().numpy().copy()
        n_before = stub._xyz.shape[0]
        torch.manual_seed(seed + 2)
        captured = {}
        real_normal = torch.normal

        def recording_normal(*a, **k):
            r = real_normal(*a, **k)
            captured["std"], captured["samples"] = k["std"].detach().clone(), r.detach().clone()
            return r
        torch.normal = recording_normal
        try:
            with torch.no_grad():
                stub.adaptive_densify(max_grad, extent)
        finally:
            torch.normal = real_normal
    for k in PARAMS:
        p = getattr(stub, ATTR[k])
        out[f"{name}_out_{k}"] = p.detach().numpy().copy()
        s = stub.optimizer.state[p]
        out[f"{name}_out_m_{k}"], out[f"{name}_out_v_{k}"] = s["exp_avg"].numpy().copy(), s["exp_avg_sq"].numpy().copy()
    for k in ("xyz_gradient_accum", "denom", "max_radii2D"):
        out[f"{name}_out_{k}"] = getattr(stub, k).numpy().copy()
    # reset_opacity (:754-764) on the densified model: new raw opacities, zeroed Adam moments for that tensor
    with torch.no_grad():
        stub.reset_opacity(0.12, 0)
    out[f"{name}_reset_opacity"] = stub._opacity.detach().numpy().copy()
    s = stub.optimizer.state[stub._opacity]
    assert float(s["exp_avg"].abs().sum()) == 0 and float(s["exp_avg_sq"].abs().sum()) == 0
    # the standard normals torch.normal(mean=0, std=stds) consumed: same generator state, same shape -> randn * stds
    std, samples = captured["std"], captured["samples"]
    torch.manual_seed(seed + 2)
    noise = torch.randn(std.shape)
    assert torch.equal(noise * std, samples), "torch.normal(mean=0, std) is not randn * std for this build"
    out[f"{name}_noise"] = noise.numpy().copy()
    n_after = stub._xyz.shape[0]
    print(f"{name}: P {P} -> after prune {n_before} -> final {n_after}")
    out[f"{name}_seed"] = np.array([seed + 2], np.int64)


def main():
    out = {}
    run_case("a", 600, 11, True, True, out)
    run_case("b", 900, 12, False, True, out)
    run_case("c", 500, 13, True, False, out)
    np.savez_compressed(os.path.join(HERE, "reference_densify.npz"), **out)
    print("wrote reference_densify.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
