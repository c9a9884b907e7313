"""Adaptive density control on the device (reference: splat_py/trainer.py:50-295 and
splat_py/optimizer_manager.py:44-172).

    ctrl = DensityController(gaussians, optimizer, DensifyConfig())
    ... per iteration:  ctrl.accumulate(uv.grad, culling_mask, camera)            # trainer.py:378-385
    ... every adaptive_control_interval iterations:  ctrl.adaptive_density_control(it)   # :208-295
    ... ctrl.reset_opacity() (:68-75), ctrl.add_sh_band() (:77-112)

The reference performs one density-control step as ~60 PyTorch calls with 10+ host synchronisations
(boolean-mask gathers of six parameter tensors and twelve optimizer-state tensors, per operation, plus
`.item()` / `.cpu().numpy()` prints).  Here the DECISIONS are formed with the reference's own expressions
on the device (same tensors, same torch.quantile -- so the masks are the reference's bit for bit), the
masks become destination indices with prefix sums, and ONE HIP launch (gs_densify_move,
csrc/densify.hip) reads every old row once and writes every new row once for all parameters and their
Adam moments.  One host read per step (the three counts that size the new tensors).

`optimizer` is the torch.optim.Adam (or gaussian_splatting_amd.train_ops.Adam) the reference's
OptimizerManager builds: one param group per tensor in the order xyz, quaternion, scale, opacity, rgb[, sh]
(optimizer_manager.py:15-42).  The controller swaps the parameters inside those groups and re-keys the state
exactly as OptimizerManager does, so `optimizer.step()` keeps working across steps.
"""
import ctypes
import math
from dataclasses import dataclass

import torch

from . import _hip
from .train_ops import accumulate_grad_stats

GROUP_ORDER = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")   # optimizer_manager.py:15-42
_KIND = {"xyz": 1, "scale": 2, "quaternion": 3}


@dataclass
class DensifyConfig:
    """the density-control fields of the reference's SplatConfig with its defaults (config.py:100-157)"""
    reset_opacity_value: float = 0.20
    max_sh_band: int = 3
    use_split: bool = True
    use_clone: bool = True
    use_delete: bool = True
    adaptive_control_start: int = 750
    adaptive_control_end: int = 6500
    adaptive_control_interval: int = 100
    max_gaussians: int = 4250000
    delete_opacity_threshold: float = 0.1
    clone_scale_threshold: float = 0.01
    use_fractional_densification: bool = True
    use_adaptive_fractional_densification: bool = True
    uv_grad_percentile: float = 0.96
    scale_norm_percentile: float = 0.99
    uv_grad_threshold: float = 0.0002
    split_scale_factor: float = 1.6
    num_split_samples: int = 2
    # learning rate of the SH group add_sh_band creates when there is none yet: base_lr * sh_lr_multiplier
    # (config.py:80,93 -> optimizer_manager.py:60-63)
    sh_lr: float = 0.002 * 0.1


def inverse_sigmoid(x):   # utils.py:11-15
    return math.log(x / (1.0 - x))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class DensityController:
    def __init__(self, gaussians, optimizer, config=None):
        self.gaussians = gaussians
        self.optimizer = optimizer
        self.config = config if config is not None else DensifyConfig()
        self.reset_grad_accum()
        self.last_info = {}

    # ---- accumulators (trainer.py:50-66, 378-385) ---------------------------------------------------------
    def reset_grad_accum(self):
        xyz = self.gaussians.xyz
        n, kw = xyz.shape[0], dict(dtype=xyz.dtype, device=xyz.device)
        self.uv_grad_accum = torch.zeros(n, 2, **kw)
        self.xyz_grad_accum = torch.zeros(n, 3, **kw)
        self.grad_accum_count = torch.zeros(n, dtype=torch.int32, device=xyz.device)

    def accumulate(self, uv_grad, culling_mask, camera):
        """the densification statistics of one training iteration (call after backward)"""
        accumulate_grad_stats(uv_grad, culling_mask, self.gaussians.xyz.grad, camera, self.uv_grad_accum,
                              self.xyz_grad_accum, self.grad_accum_count)

    # ---- optimizer plumbing (optimizer_manager.py:44-172) ---------------------------------------------------
    def _names(self):
        return [k for k in GROUP_ORDER if getattr(self.gaussians, k) is not None]

    def _group(self, name):
        return self.optimizer.param_groups[GROUP_ORDER.index(name)]

    def _state(self, name):
        group = self._group(name)
        return self.optimizer.state.get(group["params"][0], None) if group["params"] else None

    def _swap(self, name, new_tensor, exp_avg=None, exp_avg_sq=None, restart=False):
        """puts new_tensor (as a Parameter) into the Gaussians struct and the optimizer group `name`, moving
        the parameter's state entry to the new key with the given moments (None: keep the old ones).
        restart: Adam's step count of the entry goes back to 0 as well -- the reference stores the reset state
        under the integer keys optimizer.state[3] / [5] (optimizer_manager.py:57,76), which orphans it, so its
        optimizer re-initialises the new parameter at the next step(): zero moments AND step 0 (bias
        correction restarts; with the old step count the first updates would be ~3.2x smaller)"""
        group = self._group(name)
        old = group["params"][0]
        state = self.optimizer.state.pop(old, None)
        param = torch.nn.Parameter(new_tensor)
        group["params"] = [param]
        if state:
            if exp_avg is not None:
                state["exp_avg"], state["exp_avg_sq"] = exp_avg, exp_avg_sq
            if restart and "step" in state:
                state["step"] = torch.zeros_like(state["step"]) if torch.is_tensor(state["step"]) else 0
            self.optimizer.state[param] = state
        setattr(self.gaussians, name, param)

    # ---- trainer.py:68-75 ------------------------------------------------------------------------------------
    @torch.no_grad()
    def reset_opacity(self):
        old = self.gaussians.opacity
        st = self._state("opacity")
        zeros = (torch.zeros_like(old), torch.zeros_like(old)) if st else (None, None)
        self._swap("opacity", torch.ones_like(old) * inverse_sigmoid(self.config.reset_opacity_value), *zeros,
                   restart=True)
        self.reset_grad_accum()

    # ---- trainer.py:77-112 -----------------------------------------------------------------------------------
    @torch.no_grad()
    def add_sh_band(self):
        g, cfg = self.gaussians, self.config
        n = g.xyz.shape[0]
        if cfg.max_sh_band == 0:
            return
        if g.sh is None:
            g.sh = torch.nn.Parameter(torch.zeros(n, 3, 3, dtype=g.rgb.dtype, device=g.rgb.device))
            self.optimizer.add_param_group({"params": g.sh, "lr": cfg.sh_lr})
            return
        width = g.sh.shape[2]
        grow = {3: 8, 8: 15}.get(width)
        if grow is None or cfg.max_sh_band <= {3: 1, 8: 2}[width]:
            return
        new_sh = torch.zeros(n, 3, grow, dtype=g.rgb.dtype, device=g.rgb.device)
        new_sh[:, :, :width] = g.sh.detach()
        st = self._state("sh")
        zeros = (torch.zeros_like(new_sh), torch.zeros_like(new_sh)) if st else (None, None)
        self._swap("sh", new_sh, *zeros, restart=True)

    # ---- trainer.py:208-295 ----------------------------------------------------------------------------------
    @torch.no_grad()
    def adaptive_density_control(self, it, rand=None):
        """One density-control step.  rand(n) -> [n, 3] uniform samples for the split (default:
        torch.rand on the device, trainer.py:176).  -> dict with the counts of the step."""
        g, cfg = self.gaussians, self.config
        info = self.last_info = {}
        if not (cfg.use_delete or cfg.use_clone or cfg.use_split):
            return info
        dev = g.xyz.device
        N0 = g.xyz.shape[0]
        i32 = dict(dtype=torch.int32, device=dev)

        # Step 1: the delete decision (trainer.py:213-230), on all N0 rows
        keep = (g.opacity.detach() > inverse_sigmoid(cfg.delete_opacity_threshold)).squeeze(1)
        keep &= ~(self.grad_accum_count == 0)
        keep &= ~(torch.norm(self.uv_grad_accum, dim=1) == 0.0)
        if not cfg.use_delete:
            keep = torch.ones_like(keep)
        # Step 2 works on the survivors: the same expressions on the compacted statistics (the
        # per-row values do not change by compaction; torch.quantile sees exactly the survivors)
        cnt = self.grad_accum_count[keep].unsqueeze(1).float()
        uv_norm = torch.norm(self.uv_grad_accum[keep] / cnt, dim=1)
        scale_max = g.scale.detach()[keep].exp().max(dim=-1).values
        N1 = uv_norm.shape[0]
        if N1 > cfg.max_gaussians:   # trainer.py:232-235: delete only
            clone_s = split_s = torch.zeros(N1, dtype=torch.bool, device=dev)
            split_c = clone_s
            info["skipped"] = True
        else:
            if cfg.use_adaptive_fractional_densification:
                factor = (float(cfg.adaptive_control_end - it)
                          / float(cfg.adaptive_control_end - cfg.adaptive_control_start) * 2.0)
            else:
                factor = 1.0
            if cfg.use_fractional_densification:
                f = factor if cfg.use_adaptive_fractional_densification else 1.0
                uv_split_val = torch.quantile(uv_norm, 1.0 - (1.0 - cfg.uv_grad_percentile) * f) if N1 else 0.0
            else:
                uv_split_val = cfg.uv_grad_threshold
            densify = uv_norm > uv_split_val
            clone_s = densify & (scale_max <= cfg.clone_scale_threshold)
            if not cfg.use_clone:
                clone_s = torch.zeros_like(clone_s)
            # the split decision over [survivors, clones] (trainer.py:268-284): a clone carries its source's
            # densify flag and scale_max, so its decision is its source's, evaluated with the quantile of
            # the concatenated scale_max
            scale_all = torch.cat([scale_max, scale_max[clone_s]], dim=0)
            if scale_all.numel():
                scale_split = torch.quantile(scale_all, 1.0 - (1.0 - cfg.scale_norm_percentile) * factor)
                split_s = (densify & (scale_max > cfg.clone_scale_threshold)) | (scale_max > scale_split)
            else:
                split_s = torch.zeros_like(clone_s)
            if not cfg.use_split:
                split_s = torch.zeros_like(split_s)
            split_c = split_s & clone_s

        # destination indices (survivor space), then scattered back to the N0 source rows
        stay_s = ~split_s
        stay_c = clone_s & ~split_c
        a_self = torch.cumsum(stay_s, 0, dtype=torch.int32) - 1
        n_a_self = int(0)
        counts = torch.stack([stay_s.sum(), stay_c.sum(), split_s.sum(), split_c.sum()]).tolist()   # the host read
        n_a_self, n_a_clone, p_self, p_clone = counts
        a_clone = n_a_self + torch.cumsum(stay_c, 0, dtype=torch.int32) - 1
        r_self = torch.cumsum(split_s, 0, dtype=torch.int32) - 1
        r_clone = p_self + torch.cumsum(split_c, 0, dtype=torch.int32) - 1
        minus = torch.full((), -1, **i32)
        src_rows = torch.nonzero(keep, as_tuple=False).flatten()

        def scatter(values, mask):
            out = torch.full((N0,), -1, **i32)
            out[src_rows] = torch.where(mask, values.to(torch.int32), minus)
            return out

        dst_self, dst_clone = scatter(a_self, stay_s), scatter(a_clone, stay_c)
        spl_self, spl_clone = scatter(r_self, split_s), scatter(r_clone, split_c)
        P = p_self + p_clone
        samples = cfg.num_split_samples
        sample_base = n_a_self + n_a_clone
        n_new = sample_base + P * samples
        info.update(deleted=N0 - N1, cloned=int(n_a_clone + p_clone), split=int(P), n_before=N0, n_after=int(n_new))
        random_samples = (rand(P * samples) if rand is not None else torch.rand(P * samples, 3, device=dev)) \
            if P else torch.zeros(0, 3, device=dev)

        names = self._names()
        src, dst, src_m, dst_m, src_v, dst_v = [], [], [], [], [], []
        for k in names:
            t = getattr(g, k).detach().contiguous()
            st = self._state(k)
            has = bool(st) and "exp_avg" in st
            src.append(t)
            dst.append(torch.empty((n_new,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev))
            src_m.append(st["exp_avg"].contiguous() if has else None)
            src_v.append(st["exp_avg_sq"].contiguous() if has else None)
            dst_m.append(torch.empty_like(dst[-1]) if has else None)
            dst_v.append(torch.empty_like(dst[-1]) if has else None)
        n = len(names)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
        widths = (ctypes.c_int32 * n)(*[int(t[0].numel()) if t.shape[0] else int(math.prod(t.shape[1:])) for t in src])
        kinds = (ctypes.c_int32 * n)(*[_KIND.get(k, 0) for k in names])
        xyz_s, scale_s, quat_s = (src[names.index(k)] for k in ("xyz", "scale", "quaternion"))
        _hip.call("gs_densify_move", n, arr(src), arr(dst), arr(src_m), arr(dst_m), arr(src_v), arr(dst_v), widths, kinds,
                  N0, _p(dst_self), _p(dst_clone), _p(spl_self), _p(spl_clone), _p(xyz_s), _p(scale_s), _p(quat_s),
                  _p(self.xyz_grad_accum), _p(self.grad_accum_count), _p(random_samples.contiguous()), int(P),
                  int(samples), int(sample_base), ctypes.c_float(cfg.split_scale_factor),
                  _hip.current_stream())
        for k, d, m, v in zip(names, dst, dst_m, dst_v):
            self._swap(k, d, m, v)
        self.reset_grad_accum()
        return info
