"""oracle/densify_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

torch-CPU restatement of the reference's densification / pruning state machine (TS/geometry/surfel_base.py:850-1136,
1198-1230) as pure functions over a plain state dict.  PARITY STATUS: pinned -- tests/test_densify_cpu.py checks every
output against tests/golden/reference_densify.npz, which tests/golden/make_densify_golden.py produced by executing the
reference's own methods.  The only liberty: the normal samples of densify_and_split are an explicit `noise` input
(torch.normal(mean=0, std=s) == randn * s for one generator state, asserted by the generator script).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

PARAMS = ("xyz", "f_dc", "f_rest", "color", "opacity", "scaling", "rotation")
ACCUMS = ("xyz_gradient_accum", "scale_gradient_accum", "rot_gradient_accum", "opac_gradient_accum", "denom")


def new_state(params: Dict[str, torch.Tensor], m: Dict[str, torch.Tensor], v: Dict[str, torch.Tensor]) -> dict:
    P = params["xyz"].shape[0]
    st = dict(params={k: params[k].clone() for k in PARAMS}, m={k: m[k].clone() for k in PARAMS},
              v={k: v[k].clone() for k in PARAMS})
    for k in ACCUMS:
        st[k] = torch.zeros(P, 1)
    st["max_radii2D"] = torch.zeros(P)
    return st


def add_densification_stats(st: dict, radii: torch.Tensor, grad2d: torch.Tensor, scaling_grad: torch.Tensor) -> None:
    """update_states' per-view body (:1208-1216) + add_densification_stats (:1102-1128); filter = radii > 0."""
    f = radii > 0
    st["max_radii2D"] = torch.max(st["max_radii2D"], radii.float())
    st["xyz_gradient_accum"][f] += torch.norm(grad2d[f, :2], dim=-1, keepdim=True)
    st["scale_gradient_accum"][f] += scaling_grad[f, :2].sum(1, True)
    st["rot_gradient_accum"][f] += torch.norm(st["params"]["rotation"][f], dim=-1, keepdim=True)     # sic: the rotation itself
    st["opac_gradient_accum"][f] += st["params"]["opacity"][f]                                       # sic: the raw opacity
    st["denom"][f] += 1


def _keep(st: dict, keep: torch.Tensor) -> None:
    """prune_points / _prune_optimizer (:862-906): rows and Adam moments masked, accumulators masked."""
    for grp in ("params", "m", "v"):
        st[grp] = {k: t[keep] for k, t in st[grp].items()}
    for k in ACCUMS + ("max_radii2D",):
        st[k] = st[k][keep]


def _append(st: dict, new: Dict[str, torch.Tensor]) -> None:
    """densification_postfix / cat_tensors_to_optimizer (:908-980): rows appended, zero moments, accumulators reset."""
    for k in PARAMS:
        st["params"][k] = torch.cat([st["params"][k], new[k]], 0)
        st["m"][k] = torch.cat([st["m"][k], torch.zeros_like(new[k])], 0)
        st["v"][k] = torch.cat([st["v"][k], torch.zeros_like(new[k])], 0)
    P = st["params"]["xyz"].shape[0]
    for k in ACCUMS:
        st[k] = torch.zeros(P, 1)
    st["max_radii2D"] = torch.zeros(P)


def adaptive_prune(st: dict, min_opacity: float, extent: float) -> torch.Tensor:
    """:1067-1087.  Returns the prune mask over the incoming rows."""
    scaling = torch.exp(st["params"]["scaling"])
    prune_opac = (torch.sigmoid(st["params"]["opacity"]) < min_opacity).squeeze(-1)
    smin, smax = scaling[:, :2].min(1).values, scaling[:, :2].max(1).values
    prune_scale = (smax > 0.5 * extent) | ((smin * smax) < (1e-8 * extent ** 2))
    prune = prune_opac | (st["denom"] == 0).squeeze(-1) | prune_scale
    _keep(st, ~prune)
    return prune


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """TS/utils/general_utils.py:100-123."""
    q = r / torch.sqrt((r * r).sum(1))[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def adaptive_densify(st: dict, max_grad: float, extent: float, percent_dense: float, surface: bool,
                     noise: Optional[torch.Tensor], N: int = 2) -> dict:
    """:1089-1100 + densify_and_clone (:1032-1065) + densify_and_split (:982-1030).  Returns the masks it used."""
    def ratio(a):
        g = st[a] / st["denom"]
        g[g.isnan()] = 0.0
        return g
    grad_pos, grad_scale, grad_opac = ratio("xyz_gradient_accum"), ratio("scale_gradient_accum"), ratio("opac_gradient_accum")
    pre_mask = (grad_opac <= 2)[:, 0] & (grad_scale <= 1e-7)[:, 0]
    P = st["params"]
    big = torch.exp(P["scaling"]).max(1).values > percent_dense * extent
    clone = (torch.norm(grad_pos, dim=-1) >= max_grad) & ~big & pre_mask
    _append(st, {k: P[k][clone] for k in PARAMS})
    # split: gradients padded with zeros for the clones just appended
    P = st["params"]
    n = P["xyz"].shape[0]
    padded = torch.zeros(n)
    padded[: grad_pos.shape[0]] = grad_pos.squeeze()
    split = (padded >= max_grad) & (torch.exp(P["scaling"]).max(1).values > percent_dense * extent)
    stds = torch.exp(P["scaling"])[split].repeat(N, 1)
    samples = noise[: stds.shape[0]] * stds
    rots = build_rotation(P["rotation"][split]).repeat(N, 1, 1)
    new = {k: P[k][split].repeat(N, *([1] * (P[k].dim() - 1))) for k in PARAMS}
    new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + P["xyz"][split].repeat(N, 1)
    new["scaling"] = torch.log(torch.exp(P["scaling"])[split].repeat(N, 1) / (0.8 * N))
    if surface:
        new["scaling"][:, -1] = -1e10
    _append(st, new)
    keep = torch.cat([~split, torch.ones(N * int(split.sum()), dtype=torch.bool)])
    _keep(st, keep)
    return dict(clone=clone, split=split[: clone.shape[0]])


def reset_opacity(st: dict, ratio: float) -> None:
    """:754-764 + replace_tensor_to_optimizer (:846-860): opacity <- inverse_sigmoid(sigmoid(opacity) * ratio), moments zeroed."""
    x = torch.sigmoid(st["params"]["opacity"]) * ratio
    st["params"]["opacity"] = torch.log(x / (1 - x))
    st["m"]["opacity"] = torch.zeros_like(st["m"]["opacity"])
    st["v"]["opacity"] = torch.zeros_like(st["v"]["opacity"])
