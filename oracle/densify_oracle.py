"""Literal PyTorch restatement of the reference's adaptive density control -- TEST INFRASTRUCTURE (the
checker of gaussian_splatting_amd/densify.py; only tests/ and bench.py's checks may import it).

Restates, on plain tensors held in a dict, what these reference lines do to the six parameter tensors, to
Adam's per-parameter state and to the three accumulators:
    splat_py/trainer.py:68-75      reset_opacity
    splat_py/trainer.py:77-112     add_sh_band
    splat_py/trainer.py:114-121    delete_gaussians          (structs.py:92-100 filter_in_place)
    splat_py/trainer.py:123-161    clone_gaussians           (structs.py:102-114 append)
    splat_py/trainer.py:163-206    split_gaussians
    splat_py/trainer.py:208-295    adaptive_density_control
    splat_py/optimizer_manager.py:44-172   the matching exp_avg / exp_avg_sq surgery
The random numbers of the split (trainer.py:176 torch.rand) are supplied by the caller, so that the
checker and the checked see the same samples.  Pure torch: runs on CPU or GPU tensors.

PINNED (round 6) to the reference's own execution: tests/golden/ref_host_densify_*.npz are states before / after
the reference's SplatTrainer.adaptive_density_control / reset_opacity / add_sh_band run in the authoring
container (tests/golden/make_golden_densify.py imports /root/reference/splat_py/trainer.py with cv2 /
torchmetrics / tyro stubbed at module level); tests/test_densify_golden.py holds this restatement to them bit
for bit -- rows, layout, clones, split samples given the recorded torch.rand stream, every Adam moment.
"""
import math

import torch

PARAMS = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")   # optimizer group order, optimizer_manager.py:15-42


def inverse_sigmoid(x):   # utils.py:11-15
    return math.log(x / (1.0 - x))


def quaternion_to_rotation(q):   # utils.py:40-57
    rot = [
        1 - 2 * q[:, 2] ** 2 - 2 * q[:, 3] ** 2, 2 * q[:, 1] * q[:, 2] - 2 * q[:, 0] * q[:, 3],
        2 * q[:, 3] * q[:, 1] + 2 * q[:, 0] * q[:, 2], 2 * q[:, 1] * q[:, 2] + 2 * q[:, 0] * q[:, 3],
        1 - 2 * q[:, 1] ** 2 - 2 * q[:, 3] ** 2, 2 * q[:, 2] * q[:, 3] - 2 * q[:, 0] * q[:, 1],
        2 * q[:, 3] * q[:, 1] - 2 * q[:, 0] * q[:, 2], 2 * q[:, 2] * q[:, 3] + 2 * q[:, 0] * q[:, 1],
        1 - 2 * q[:, 1] ** 2 - 2 * q[:, 2] ** 2,
    ]
    return torch.stack(rot, dim=1).reshape(-1, 3, 3)


class State:
    """params / exp_avg / exp_avg_sq: dicts name -> tensor (sh may be missing); the three accumulators"""

    def __init__(self, params, exp_avg, exp_avg_sq, uv_grad_accum, xyz_grad_accum, grad_accum_count):
        self.p = {k: v.clone() for k, v in params.items() if v is not None}
        self.m = {k: v.clone() for k, v in exp_avg.items() if v is not None}
        self.v = {k: v.clone() for k, v in exp_avg_sq.items() if v is not None}
        self.uv_grad_accum = uv_grad_accum.clone()
        self.xyz_grad_accum = xyz_grad_accum.clone()
        self.grad_accum_count = grad_accum_count.clone()

    def n(self):
        return self.p["xyz"].shape[0]

    def reset_grad_accum(self):   # trainer.py:50-66
        n, ref = self.n(), self.p["xyz"]
        self.uv_grad_accum = torch.zeros(n, 2, dtype=ref.dtype, device=ref.device)
        self.xyz_grad_accum = torch.zeros(n, 3, dtype=ref.dtype, device=ref.device)
        self.grad_accum_count = torch.zeros(n, dtype=torch.int32, device=ref.device)

    def delete(self, keep):   # trainer.py:114-121, optimizer_manager.py:78-100
        for k in self.p:
            self.p[k] = self.p[k][keep]
        for k in self.m:
            self.m[k] = self.m[k][keep]
            self.v[k] = self.v[k][keep]
        self.uv_grad_accum = self.uv_grad_accum[keep, :]
        self.xyz_grad_accum = self.xyz_grad_accum[keep, :]
        self.grad_accum_count = self.grad_accum_count[keep]

    def append(self, new):   # structs.py:102-114, optimizer_manager.py:102-172 (state of the new rows = 0)
        for k in self.p:
            self.p[k] = torch.cat((self.p[k], new[k]), dim=0)
        for k in self.m:
            self.m[k] = torch.cat((self.m[k], torch.zeros_like(new[k])), dim=0)
            self.v[k] = torch.cat((self.v[k], torch.zeros_like(new[k])), dim=0)


def clone_gaussians(st, clone_mask, xyz_grad_avg):   # trainer.py:123-161
    new = {k: st.p[k][clone_mask].clone() for k in st.p}
    new["xyz"] = new["xyz"] - xyz_grad_avg[clone_mask, :] * 0.01
    st.uv_grad_accum = torch.cat([st.uv_grad_accum, st.uv_grad_accum[clone_mask, :]], dim=0)
    st.xyz_grad_accum = torch.cat([st.xyz_grad_accum, st.xyz_grad_accum[clone_mask, :]], dim=0)
    st.grad_accum_count = torch.cat([st.grad_accum_count, st.grad_accum_count[clone_mask]], dim=0)
    st.append(new)


def split_gaussians(st, split_mask, cfg, rand):   # trainer.py:163-206; rand(n) -> [n, 3] uniform samples
    samples = cfg.num_split_samples
    rep = lambda t: t[split_mask].clone().repeat(samples, *([1] * (t.dim() - 1)))
    new = {k: rep(st.p[k]) for k in st.p}
    random_samples = rand(int(split_mask.sum()) * samples)
    random_samples = random_samples * torch.exp(new["scale"])
    q = new["quaternion"] / torch.norm(new["quaternion"], dim=1, keepdim=True)
    new["quaternion"] = q
    random_samples = torch.bmm(quaternion_to_rotation(q), random_samples.unsqueeze(-1)).squeeze(-1)
    new["xyz"] = new["xyz"] + random_samples
    new["scale"] = torch.log(torch.exp(new["scale"]) / cfg.split_scale_factor)
    st.delete(~split_mask)
    st.append(new)


def adaptive_density_control(st, cfg, it, rand):   # trainer.py:208-295
    """-> dict of what happened (counts, thresholds), for the tests to compare"""
    info = {}
    if not (cfg.use_delete or cfg.use_clone or cfg.use_split):
        return info
    keep = (st.p["opacity"] > inverse_sigmoid(cfg.delete_opacity_threshold)).squeeze(1)
    keep &= ~(st.grad_accum_count == 0)
    keep &= ~(torch.norm(st.uv_grad_accum, dim=1) == 0.0)
    n_delete = int((~keep).sum())
    info["deleted"] = n_delete if cfg.use_delete else 0   # (the reference prints the count either way)
    if n_delete > 0 and cfg.use_delete:
        st.delete(keep)
    if st.n() > cfg.max_gaussians:
        st.reset_grad_accum()
        info["skipped"] = True
        return info
    cnt = st.grad_accum_count.unsqueeze(1).float()
    uv_grad_avg = st.uv_grad_accum / cnt
    xyz_grad_avg = st.xyz_grad_accum / cnt
    uv_norm = torch.norm(uv_grad_avg, dim=1)
    if cfg.use_adaptive_fractional_densification:
        factor = float(cfg.adaptive_control_end - it) / float(cfg.adaptive_control_end - cfg.adaptive_control_start) * 2.0
    else:
        factor = 1.0
    if cfg.use_fractional_densification:
        f = factor if cfg.use_adaptive_fractional_densification else 1.0
        uv_split_val = torch.quantile(uv_norm, 1.0 - (1.0 - cfg.uv_grad_percentile) * f).item()
    else:
        uv_split_val = cfg.uv_grad_threshold
    densify = uv_norm > uv_split_val
    scale_max = st.p["scale"].exp().max(dim=-1).values
    clone_mask = densify & (scale_max <= cfg.clone_scale_threshold)
    info["cloned"] = int(clone_mask.sum()) if cfg.use_clone else 0
    if clone_mask.any() and cfg.use_clone:
        clone_gaussians(st, clone_mask, xyz_grad_avg)
        densify = torch.cat([densify, densify[clone_mask]], dim=0)
        scale_max = torch.cat([scale_max, scale_max[clone_mask]], dim=0)
    split_mask = densify & (scale_max > cfg.clone_scale_threshold)
    scale_split = torch.quantile(scale_max, 1.0 - (1.0 - cfg.scale_norm_percentile) * factor).item()
    split_mask = split_mask | (scale_max > scale_split)
    info["split"] = int(split_mask.sum()) if cfg.use_split else 0
    if split_mask.any() and cfg.use_split:
        split_gaussians(st, split_mask, cfg, rand)
    st.reset_grad_accum()
    return info


def reset_opacity(st, cfg):   # trainer.py:68-75, optimizer_manager.py:44-58
    st.p["opacity"] = torch.ones_like(st.p["opacity"]) * inverse_sigmoid(cfg.reset_opacity_value)
    if "opacity" in st.m:
        st.m["opacity"] = torch.zeros_like(st.m["opacity"])
        st.v["opacity"] = torch.zeros_like(st.v["opacity"])
    st.reset_grad_accum()


def add_sh_band(st, cfg):   # trainer.py:77-112, optimizer_manager.py:60-76
    n, ref = st.n(), st.p["rgb"]
    if cfg.max_sh_band == 0:
        return
    if "sh" not in st.p:
        st.p["sh"] = torch.zeros(n, 3, 3, dtype=ref.dtype, device=ref.device)
        return   # add_param_group: the new group has no optimizer state until its first step
    width = st.p["sh"].shape[2]
    grow = {3: 8, 8: 15}.get(width)
    if grow is None or cfg.max_sh_band <= {3: 1, 8: 2}[width]:
        return
    new_sh = torch.zeros(n, 3, grow, dtype=ref.dtype, device=ref.device)
    new_sh[:, :, :width] = st.p["sh"]
    st.p["sh"] = new_sh
    if "sh" in st.m:   # add_sh_band_to_optimizer: both moments restart at zero
        st.m["sh"] = torch.zeros_like(new_sh)
        st.v["sh"] = torch.zeros_like(new_sh)
