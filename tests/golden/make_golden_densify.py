"""Generates tests/golden/ref_host_densify_*.npz.  RUNS ONLY IN THE AUTHORING CONTAINER.

Pins adaptive density control, opacity reset and SH-band growth to the REFERENCE'S OWN EXECUTION: it imports
/root/reference/splat_py/trainer.py and optimizer_manager.py and drives their methods on seeded CPU states.
Those lines (trainer.py:50-295, optimizer_manager.py:9-172) are device-generic PyTorch; what keeps trainer.py
from importing here are three module-level imports of packages this image lacks -- `cv2`,
`torchmetrics.image` (trainer.py:1-4) and `tyro` (config.py:2) -- none of which the density-control methods touch.
They are replaced by empty stand-in modules in sys.modules for the duration of this script (the `splat_cuda`
extension the host code imports is the CPU oracle, as in make_golden.py).  The trainer object is made with
`__new__` plus the attributes those methods read; the config is the reference's own `SplatConfig()` (its
defaults, config.py:29-157).  The uniform samples of the split (`torch.rand`, trainer.py:176) are recorded by
wrapping `torch.rand` during the call, so that a checker can be fed the same stream.

Only the .npz files travel (arrays: the state before, the state after, the samples, the counts).

usage: python tests/golden/make_golden_densify.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import gs_oracle as orc  # noqa: E402

# ---- stand-ins for what the image lacks (module-level imports only; nothing below calls into them) --------
mod = types.ModuleType("splat_cuda")
for name in orc.SPLAT_CUDA_API:
    setattr(mod, name, getattr(orc, name))
sys.modules["splat_cuda"] = mod
sys.modules["cv2"] = types.ModuleType("cv2")
tm, tmi = types.ModuleType("torchmetrics"), types.ModuleType("torchmetrics.image")
tmi.StructuralSimilarityIndexMeasure = type("StructuralSimilarityIndexMeasure", (), {})
tm.image = tmi
sys.modules["torchmetrics"], sys.modules["torchmetrics.image"] = tm, tmi
tyro = types.ModuleType("tyro")
tyro.extras = types.SimpleNamespace(subcommand_type_from_defaults=lambda d: d)
sys.modules["tyro"] = tyro
sys.path.insert(0, REF)

from splat_py.config import SplatConfig  # noqa: E402
from splat_py.optimizer_manager import OptimizerManager  # noqa: E402
from splat_py.structs import Gaussians as RefGaussians  # noqa: E402
from splat_py.trainer import SplatTrainer  # noqa: E402

from gaussian_splatting_amd.synthetic import make_scene  # noqa: E402

NAMES = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")   # optimizer_manager.py:15-42 group order


def np_(t):
    return t.detach().cpu().numpy().copy()


def make_trainer(N, deg, seed, steps=3, **flags):
    """a SplatTrainer without its dataset: the attributes adaptive_density_control / reset_opacity /
    add_sh_band read (trainer.py:17-22, 50-66), Adam moments from `steps` real optimizer steps on seeded
    gradients, accumulators as a few views would leave them (some Gaussians unseen, some without gradient)"""
    g, _, _ = make_scene(N, 640, 480, deg, seed=seed)
    P = torch.nn.Parameter
    rg = RefGaussians(P(g.xyz), P(g.rgb), P(g.opacity), P(g.scale), P(g.quaternion),
                      P(g.sh) if g.sh is not None else None)
    cfg = SplatConfig()
    for k, v in flags.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    tr = SplatTrainer.__new__(SplatTrainer)
    tr.gaussians, tr.config = rg, cfg
    tr.optimizer_manager = OptimizerManager(rg, cfg)
    tr.reset_grad_accum()
    gen = torch.Generator().manual_seed(seed + 100)
    step_grads = []
    for _ in range(steps):
        grads = {}
        for k in NAMES:
            p = getattr(rg, k)
            if p is not None:
                p.grad = torch.randn(p.shape, generator=gen) * 1e-3
                grads[k] = p.grad.clone()
        tr.optimizer_manager.optimizer.step()
        step_grads.append(grads)
    tr.grad_accum_count = torch.randint(0, 6, (N,), generator=gen, dtype=torch.int32)
    tr.uv_grad_accum = torch.rand(N, 2, generator=gen) * 1e-3 * tr.grad_accum_count.unsqueeze(1)
    tr.uv_grad_accum[torch.rand(N, generator=gen) < 0.03] = 0.0
    tr.xyz_grad_accum = torch.rand(N, 3, generator=gen) * 1e-2
    return tr


def snapshot(tr, tag, out):
    """parameters, the Adam state FOUND UNDER EACH PARAMETER'S KEY (what optimizer.step() will use), the
    accumulators; has_state = 0 where the reference leaves the new parameter without an entry"""
    g, opt = tr.gaussians, tr.optimizer_manager.optimizer
    names = [k for k in NAMES if getattr(g, k) is not None]
    out[f"{tag}_names"] = np.array(names)
    for i, k in enumerate(names):
        p = getattr(g, k)
        assert opt.param_groups[i]["params"][0] is p, (tag, k)   # the group holds the struct's parameter
        out[f"{tag}_{k}"] = np_(p)
        st = opt.state.get(p, {}) if p in opt.state else {}
        out[f"{tag}_has_state_{k}"] = np.bool_("exp_avg" in st)
        if "exp_avg" in st:
            out[f"{tag}_m_{k}"], out[f"{tag}_v_{k}"] = np_(st["exp_avg"]), np_(st["exp_avg_sq"])
            out[f"{tag}_step_{k}"] = np.float64(float(st["step"]))
        out[f"{tag}_lr_{k}"] = np.float64(opt.param_groups[i]["lr"])
    out[f"{tag}_uv_grad_accum"] = np_(tr.uv_grad_accum)
    out[f"{tag}_xyz_grad_accum"] = np_(tr.xyz_grad_accum)
    out[f"{tag}_grad_accum_count"] = np_(tr.grad_accum_count)


class RandRecorder:
    """records what torch.rand returns inside the reference call (trainer.py:176)"""

    def __enter__(self):
        self.real, self.samples = torch.rand, []

        def rand(*a, **kw):
            r = self.real(*a, **kw)
            self.samples.append(r.clone())
            return r
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.rand = self.real


def densify_case(tag, N, deg, it, seed, **flags):
    tr = make_trainer(N, deg, seed, **flags)
    out = {"iter": np.int64(it)}
    for k, v in flags.items():
        out["flag_" + k] = np.float64(v)
    snapshot(tr, "before", out)
    torch.manual_seed(seed + 7)
    with RandRecorder() as rec:
        tr.adaptive_density_control(it)
    assert len(rec.samples) <= 1
    out["split_rand"] = np_(rec.samples[0]) if rec.samples else np.zeros((0, 3), np.float32)
    snapshot(tr, "after", out)
    np.savez_compressed(os.path.join(HERE, f"ref_host_densify_{tag}.npz"), **out)
    print(tag, "N", N, "->", len(tr.gaussians), "split samples", out["split_rand"].shape[0])


def reset_and_bands_case(tag, N, seed):
    """reset_opacity (trainer.py:68-75) and the three add_sh_band steps (:77-112), each followed by ONE real
    optimizer step on seeded gradients: the reference files the reset moments under an integer key
    (optimizer_manager.py:57,76), so its Adam re-initialises the parameter -- the step pins that restart"""
    tr = make_trainer(N, 0, seed)
    out = {}
    gen = torch.Generator().manual_seed(seed + 200)

    def one_step(stage):
        g = tr.gaussians
        for k in NAMES:
            p = getattr(g, k)
            if p is not None:
                p.grad = torch.randn(p.shape, generator=gen) * 1e-3
                out[f"{stage}_grad_{k}"] = np_(p.grad)
        tr.optimizer_manager.optimizer.step()

    snapshot(tr, "s0", out)
    tr.reset_opacity()
    snapshot(tr, "s1_reset", out)
    one_step("s1")
    snapshot(tr, "s1_stepped", out)
    for j in (2, 3, 4, 5):   # None -> 3 -> 8 -> 15 coefficients, then nothing more
        tr.add_sh_band()
        snapshot(tr, f"s{j}_band", out)
        one_step(f"s{j}")
        snapshot(tr, f"s{j}_stepped", out)
    assert tr.gaussians.sh.shape[2] == 15
    np.savez_compressed(os.path.join(HERE, f"ref_host_densify_{tag}.npz"), **out)
    print(tag, "written; sh", tuple(tr.gaussians.sh.shape))


if __name__ == "__main__":
    densify_case("deg3_it1000", 260, 3, 1000, seed=21)
    densify_case("deg0_it3000_more_clones", 500, 0, 3000, seed=22, clone_scale_threshold=0.12)
    densify_case("deg1_it6400_3samples", 320, 1, 6400, seed=23, num_split_samples=3)
    densify_case("deg2_threshold_mode", 300, 2, 1000, seed=24, use_fractional_densification=False,
                 use_adaptive_fractional_densification=False, uv_grad_threshold=0.0008)
    densify_case("deg0_max_exceeded", 300, 0, 1000, seed=25, max_gaussians=100)
    reset_and_bands_case("reset_and_bands", 48, seed=26)
