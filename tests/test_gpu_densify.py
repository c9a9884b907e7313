"""Adaptive density control on the device (gaussian_splatting_amd.densify, csrc/densify.hip) against the
literal PyTorch restatement of the reference's lines (oracle/densify_oracle.py: trainer.py:68-295,
optimizer_manager.py:44-172).  Decisions, layout, copies, the clones' positions and every optimizer moment:
bit-exact.  The split samples' xyz / scale / quaternion: 1e-6 of max(|value|, 1) (expf / logf / a 3x3 product
whose association torch.bmm does not pin)."""
import pytest
import torch

from gaussian_splatting_amd import fused
from gaussian_splatting_amd.densify import DensifyConfig, DensityController
from gaussian_splatting_amd.synthetic import DEFAULTS, make_grad_image, make_scene
from gaussian_splatting_amd.train_ops import Adam, ssim_l1_loss

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")
LRS = (2e-4, 4e-3, 1e-2, 2e-2, 4e-3, 2e-4)


def oracle():
    from oracle import densify_oracle
    return densify_oracle


def build(N, deg, seed, with_state=True):
    g, cam, T = make_scene(N, 640, 480, deg, seed=seed, device=DEV)
    params = [getattr(g, k) for k in NAMES if getattr(g, k) is not None]
    for p in params:
        p.requires_grad_(True)
    opt = Adam([{"params": p, "lr": lr} for p, lr in zip(params, LRS)])
    gen = torch.Generator(device=DEV).manual_seed(seed + 100)
    if with_state:   # a few optimizer steps on random gradients: non-trivial moments
        for _ in range(3):
            for p in params:
                p.grad = torch.randn(p.shape, generator=gen, device=DEV) * 1e-3
            opt.step()
    ctrl = DensityController(g, opt, DensifyConfig())
    # accumulators as a few hundred training views would leave them: some Gaussians never seen, some
    # seen without gradient
    ctrl.grad_accum_count = torch.randint(0, 6, (N,), generator=gen, device=DEV, dtype=torch.int32)
    ctrl.uv_grad_accum = torch.rand(N, 2, generator=gen, device=DEV) * 1e-3 * ctrl.grad_accum_count.unsqueeze(1)
    ctrl.uv_grad_accum[torch.rand(N, generator=gen, device=DEV) < 0.03] = 0.0
    ctrl.xyz_grad_accum = torch.rand(N, 3, generator=gen, device=DEV) * 1e-2
    return g, opt, ctrl, cam, T


def snapshot(g, opt, ctrl):
    dens = oracle()
    params = {k: getattr(g, k).detach() for k in NAMES if getattr(g, k) is not None}
    m, v = {}, {}
    for k, p in ((k, getattr(g, k)) for k in params):
        st = opt.state.get(p, {})
        if "exp_avg" in st:
            m[k], v[k] = st["exp_avg"], st["exp_avg_sq"]
    return dens.State(params, m, v, ctrl.uv_grad_accum, ctrl.xyz_grad_accum, ctrl.grad_accum_count)


def compare(g, opt, ctrl, st, n_exact_rows):
    """rows [0, n_exact_rows) -- survivors and clones -- bit-exact; the split samples behind them 1e-6"""
    assert g.xyz.shape[0] == st.n()
    for k in st.p:
        got, ref = getattr(g, k).detach(), st.p[k]
        assert got.shape == ref.shape, k
        assert torch.equal(got[:n_exact_rows], ref[:n_exact_rows]), k
        if k in ("xyz", "scale", "quaternion"):
            tail_g, tail_r = got[n_exact_rows:], ref[n_exact_rows:]
            if tail_r.numel():
                err = ((tail_g - tail_r).abs() / tail_r.abs().clamp(min=1.0)).max().item()
                assert err < 1e-6, (k, err)
        else:
            assert torch.equal(got, ref), k
        state = opt.state.get(getattr(g, k), {})
        if k in st.m:
            assert torch.equal(state["exp_avg"], st.m[k]) and torch.equal(state["exp_avg_sq"], st.v[k]), k
        else:
            assert "exp_avg" not in state, k
    assert torch.equal(ctrl.uv_grad_accum, st.uv_grad_accum) and torch.equal(ctrl.grad_accum_count, st.grad_accum_count)
    # the optimizer's groups hold exactly the struct's parameters
    for i, k in enumerate(k for k in NAMES if getattr(g, k) is not None):
        assert opt.param_groups[i]["params"][0] is getattr(g, k)


@pytest.mark.parametrize("N,deg,it,flags", [(20000, 3, 1000, {}), (20000, 0, 3000, {}), (5000, 2, 6400, {}),
                                             (20000, 3, 1000, dict(use_delete=False)),
                                             (20000, 3, 1000, dict(use_clone=False)),
                                             (20000, 3, 1000, dict(use_split=False)),
                                             (20000, 1, 1000, dict(use_fractional_densification=False,
                                                                   uv_grad_threshold=2e-4)),
                                             (20000, 3, 1000, dict(max_gaussians=1000)),
                                             (20000, 3, 1000, dict(num_split_samples=3))])
def test_adaptive_density_control_matches_the_reference_lines(N, deg, it, flags):
    dens = oracle()
    g, opt, ctrl, cam, T = build(N, deg, seed=it)
    for k, v in flags.items():
        setattr(ctrl.config, k, v)
    st = snapshot(g, opt, ctrl)
    pool = torch.rand(4 * N, 3, generator=torch.Generator(device=DEV).manual_seed(5), device=DEV)
    rand = lambda n: pool[:n].clone()
    ref_info = dens.adaptive_density_control(st, ctrl.config, it, rand)
    info = ctrl.adaptive_density_control(it, rand)
    if not ref_info.get("skipped"):
        assert info["deleted"] == ref_info["deleted"] and info["cloned"] == ref_info["cloned"]
        assert info["split"] == ref_info["split"]
    assert info["n_after"] == st.n()
    if not flags and it == 1000:
        assert info["deleted"] > 0 and info["cloned"] > 0 and info["split"] > 0
    n_exact = info["n_after"] - info["split"] * ctrl.config.num_split_samples
    compare(g, opt, ctrl, st, n_exact)


def test_density_control_before_the_first_optimizer_step():
    """no optimizer state yet: parameters move, no moments appear"""
    dens = oracle()
    g, opt, ctrl, cam, T = build(8000, 3, seed=3, with_state=False)
    st = snapshot(g, opt, ctrl)
    pool = torch.rand(32000, 3, device=DEV)
    rand = lambda n: pool[:n].clone()
    dens.adaptive_density_control(st, ctrl.config, 1000, rand)
    info = ctrl.adaptive_density_control(1000, rand)
    compare(g, opt, ctrl, st, info["n_after"] - info["split"] * 2)


def test_reset_opacity_and_sh_band_growth_match_the_reference_lines():
    dens = oracle()
    g, opt, ctrl, cam, T = build(6000, 1, seed=11)
    st = snapshot(g, opt, ctrl)
    dens.reset_opacity(st, ctrl.config)
    ctrl.reset_opacity()
    compare(g, opt, ctrl, st, g.xyz.shape[0])
    # the reference orphans the reset state (optimizer_manager.py:57), i.e. Adam restarts for this parameter:
    # zero moments AND step 0
    assert float(opt.state[g.opacity]["step"]) == 0.0 and float(opt.state[g.xyz]["step"]) > 0.0
    for _ in range(3):   # 3 -> 8 -> 15 coefficients, then nothing more
        dens.add_sh_band(st, ctrl.config)
        ctrl.add_sh_band()
        compare(g, opt, ctrl, st, g.xyz.shape[0])
    assert g.sh.shape[2] == 15 and float(opt.state[g.sh]["step"]) == 0.0
    # from no SH at all: a new parameter group appears
    g, opt, ctrl, cam, T = build(3000, 0, seed=12)
    ctrl.add_sh_band()
    assert g.sh.shape == (3000, 3, 3) and not g.sh.any() and opt.param_groups[5]["params"][0] is g.sh


def test_training_continues_across_density_control():
    """rasterize -> loss -> backward -> Adam -> statistics, with density control, opacity reset and SH growth
    in between: shapes stay consistent and the optimizer keeps stepping"""
    N, W, H = 30000, 320, 240
    g, cam, T = make_scene(N, W, H, 0, seed=2, device=DEV)
    params = [getattr(g, k) for k in NAMES if getattr(g, k) is not None]
    for p in params:
        p.requires_grad_(True)
    opt = Adam([{"params": p, "lr": lr} for p, lr in zip(params, LRS)])
    ctrl = DensityController(g, opt, DensifyConfig(adaptive_control_start=0, adaptive_control_end=100))
    target = torch.rand(H, W, 3, device=DEV)
    bg = torch.zeros(3, device=DEV)
    sizes = []
    for it in range(1, 31):
        opt.zero_grad(set_to_none=True)
        img, culled, uv = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
        uv.retain_grad()
        ssim_l1_loss(img, target, 0.2).backward()
        opt.step()
        ctrl.accumulate(uv.grad, culled, cam)
        if it % 10 == 0:
            info = ctrl.adaptive_density_control(it)
            sizes.append(info["n_after"])
            assert g.xyz.shape[0] == info["n_after"] == ctrl.grad_accum_count.shape[0]
        if it == 15:
            ctrl.reset_opacity()
        if it == 20:
            ctrl.add_sh_band()
            assert g.sh is not None and len(opt.param_groups) == 6
    assert len(set(sizes)) > 1 and all(torch.isfinite(getattr(g, k)).all() for k in NAMES if getattr(g, k) is not None)
