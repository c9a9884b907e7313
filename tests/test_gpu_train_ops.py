"""GPU parity of the training-loop operations behind the rasterizer.  The checker is
the reference's own dependency run here: torch.optim.Adam on the CPU (splat_py/optimizer_manager.py:15-42
builds exactly that optimizer), and the literal PyTorch lines of trainer.py:378-385."""
import pytest
import torch

from gaussian_splatting_amd.synthetic import make_scene
from gaussian_splatting_amd.train_ops import Adam, accumulate_grad_stats

pytestmark = pytest.mark.gpu
DEV = "cuda"
NAMES = ("xyz", "quaternion", "scale", "opacity", "rgb", "sh")
# config.py of the reference: base_lr 0.002 times the per-group multipliers
LRS = dict(xyz=0.002 * 0.1, quaternion=0.002 * 2, scale=0.002 * 5, opacity=0.002 * 10, rgb=0.002 * 2, sh=0.002 * 0.1)


def build(optim_cls, g):
    groups = [{"params": getattr(g, k), "lr": LRS[k]} for k in NAMES[:5]]
    opt = optim_cls(groups)
    opt.add_param_group({"params": g.sh, "lr": LRS["sh"]})   # optimizer_manager.py:37-41
    return opt


@pytest.mark.parametrize("N,deg", [(1000, 3), (4099, 1)])
def test_adam_step_matches_torch_adam(N, deg):
    """10 steps with fresh random gradients: parameters and both moments equal torch.optim.Adam (CPU)
    to a few ulp.  Tolerance: |x - ref| <= 2e-6 * max|ref| per tensor (fp32; ATen's vectorised CPU
    kernels contract lerp into an FMA, this kernel does not)."""
    g_ref, _, _ = make_scene(N, 64, 64, deg, seed=1)
    g_hip, _, _ = make_scene(N, 64, 64, deg, seed=1, device=DEV)
    for k in NAMES:
        getattr(g_ref, k).requires_grad_(True)
        getattr(g_hip, k).requires_grad_(True)
    ref, hip = build(torch.optim.Adam, g_ref), build(Adam, g_hip)
    gen = torch.Generator().manual_seed(5)
    for it in range(10):
        for k in NAMES:
            grad = torch.randn(getattr(g_ref, k).shape, generator=gen) * (10.0 ** (it % 3 - 2))
            if it == 4:
                grad[::3] = 0   # exact zeros: the moments decay, the step shrinks
            getattr(g_ref, k).grad = grad
            getattr(g_hip, k).grad = grad.to(DEV)
        ref.step()
        hip.step()
    for k in NAMES:
        p_ref, p_hip = getattr(g_ref, k), getattr(g_hip, k)
        s_ref, s_hip = ref.state[p_ref], hip.state[p_hip]
        assert float(s_hip["step"]) == float(s_ref["step"]) == 10
        for name, a, b in (("param", p_hip.detach(), p_ref.detach()), ("exp_avg", s_hip["exp_avg"], s_ref["exp_avg"]),
                           ("exp_avg_sq", s_hip["exp_avg_sq"], s_ref["exp_avg_sq"])):
            err = (a.cpu() - b).abs().max() / b.abs().max()
            assert float(err) < 2e-6, (k, name, float(err))


def test_adam_state_layout_survives_the_reference_surgery():
    """OptimizerManager.reset_opacity_exp_avg / delete / add (optimizer_manager.py:44-160) reach into
    optimizer.state[param] and param_groups[i]["params"][0]: same keys and shapes as torch's"""
    g, _, _ = make_scene(500, 64, 64, 0, seed=2, device=DEV)
    for k in NAMES[:5]:
        getattr(g, k).requires_grad_(True)
        getattr(g, k).grad = torch.ones_like(getattr(g, k))
    opt = Adam([{"params": getattr(g, k), "lr": LRS[k]} for k in NAMES[:5]])
    opt.step()
    st = opt.state[opt.param_groups[3]["params"][0]]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"} and st["exp_avg"].shape == g.opacity.shape
    # a tensor torch's kernel would take but this one does not (fp64) goes through torch's own step
    q = torch.zeros(5, dtype=torch.float64, device=DEV, requires_grad=True)
    q.grad = torch.ones_like(q)
    opt2 = Adam([q], lr=0.1)
    opt2.step()
    assert torch.allclose(q.detach(), torch.full_like(q, -0.1))


def test_accumulate_grad_stats_matches_the_trainer_lines():
    N = 5000
    g, cam, _ = make_scene(N, 320, 240, 0, seed=3, device=DEV)
    gen = torch.Generator().manual_seed(6)
    culling_mask = (torch.rand(N, generator=gen) < 0.3).to(DEV)
    V = int((~culling_mask).sum())
    slab = torch.randn(V, 9, generator=gen).to(DEV)
    uv_grad = slab[:, 4:6]                       # the fused path hands out this strided view
    xyz_grad = torch.randn(N, 3, generator=gen).to(DEV)
    uv_acc = torch.rand(N, 2, generator=gen).to(DEV)
    xyz_acc = torch.rand(N, 3, generator=gen).to(DEV)
    count = torch.randint(0, 5, (N,), generator=gen, dtype=torch.int32).to(DEV)
    # trainer.py:378-385, literally
    ref_uv, ref_xyz, ref_count = uv_acc.clone(), xyz_acc.clone(), count.clone()
    ug = uv_grad.detach().clone()
    ug[:, 0] = ug[:, 0] * cam.K[0, 0]
    ug[:, 1] = ug[:, 1] * cam.K[1, 1]
    ref_uv[~culling_mask] += torch.abs(ug)
    ref_xyz += torch.abs(xyz_grad)
    ref_count += (~culling_mask).int()
    before = slab.clone()
    accumulate_grad_stats(uv_grad, culling_mask, xyz_grad, cam, uv_acc, xyz_acc, count)
    assert torch.equal(uv_acc, ref_uv) and torch.equal(xyz_acc, ref_xyz) and torch.equal(count, ref_count)
    assert torch.equal(slab, before)


@pytest.mark.parametrize("H,W,frac", [(40, 50, 0.2), (97, 131, 0.2), (16, 11 + 16, 0.5), (840, 1297, 0.2)])
def test_ssim_l1_loss_matches_the_oracle(H, W, frac):
    """value and gradient against the plain-PyTorch restatement of trainer.py:363-374 +
    torchmetrics 1.2.1 SSIM (oracle/loss_oracle.py, CPU, autograd).  Tolerances (fp32, different
    summation order of the 121-tap windows): loss 1e-5 relative, gradient 1e-4 of its maximum."""
    from gaussian_splatting_amd.train_ops import ssim_l1_loss
    from oracle import loss_oracle
    gen = torch.Generator().manual_seed(H * 1000 + W)
    target = torch.rand(H, W, 3, generator=gen)
    image = (target + 0.15 * torch.randn(H, W, 3, generator=gen)).clamp(0, 1)
    image[: H // 4] = target[: H // 4]          # an exactly matching region: sign(0) = 0, SSIM = 1
    ref_img = image.clone().requires_grad_(True)
    ref_loss, ref_l1, ref_ssim = loss_oracle.ssim_l1_loss(ref_img, target, frac)
    ref_loss.backward()
    img = image.to(DEV).requires_grad_(True)
    loss, terms = ssim_l1_loss(img, target.to(DEV), frac, return_terms=True)
    (2.0 * loss).backward()                      # a non-unit upstream gradient
    terms = terms.cpu()
    ref_loss, ref_l1, ref_ssim = ref_loss.detach(), ref_l1.detach(), ref_ssim.detach()
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
    assert abs(float(terms[1]) - float(ref_l1)) <= 1e-5 * abs(float(ref_l1))
    assert abs(float(terms[2]) - float(ref_ssim)) <= 1e-5
    ref_mse = float(torch.nn.functional.mse_loss(image, target))
    assert abs(float(terms[3]) - ref_mse) <= 1e-5 * ref_mse
    err = (img.grad.cpu() / 2.0 - ref_img.grad).abs().max() / ref_img.grad.abs().max()
    assert float(err) < 1e-4, float(err)


def test_ssim_l1_loss_properties():
    from gaussian_splatting_amd.train_ops import ssim_l1_loss
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(64, 80, 3, generator=gen).to(DEV)
    y = torch.rand(64, 80, 3, generator=gen).to(DEV)
    _, t_same = ssim_l1_loss(x, x.clone(), 0.2, return_terms=True)
    assert float(t_same[0]) == 0.0 and float(t_same[1]) == 0.0 and abs(float(t_same[2]) - 1.0) < 1e-6
    _, t_xy = ssim_l1_loss(x, y, 0.2, return_terms=True)
    _, t_yx = ssim_l1_loss(y, x, 0.2, return_terms=True)
    assert torch.equal(t_xy, t_yx)                                   # symmetric, and bit-reproducible
    _, t_again = ssim_l1_loss(x, y, 0.2, return_terms=True)
    assert torch.equal(t_xy, t_again)
    with pytest.raises(RuntimeError):
        ssim_l1_loss(x[:8], y[:8], 0.2)                              # smaller than the 11x11 window


def test_training_iterations_reduce_the_loss():
    """the pieces of trainer.py:348-385 together: rasterize -> SSIM/L1 loss -> backward -> Adam ->
    densification statistics, on a small scene fitted to a render of its unperturbed self"""
    from gaussian_splatting_amd import fused
    from gaussian_splatting_amd.train_ops import ssim_l1_loss
    N, W, H = 4000, 192, 128
    args = dict(near_thresh=0.3, far_thresh=500.0, cull_mask_padding=100, mh_dist=3.0, use_sh_precompute=True,
                background_rgb=torch.zeros(3, device=DEV))
    g, cam, T = make_scene(N, W, H, 1, seed=11, device=DEV)
    with torch.no_grad():
        target, _, _ = fused.rasterize(g, T, cam, **args)
        gen = torch.Generator().manual_seed(3)
        g.rgb.add_(0.5 * torch.randn(g.rgb.shape, generator=gen).to(DEV))
        g.xyz.add_(0.02 * torch.randn(g.xyz.shape, generator=gen).to(DEV))
    for k in NAMES:
        getattr(g, k).requires_grad_(True)
    opt = build(Adam, g)
    uv_acc, xyz_acc = torch.zeros(N, 2, device=DEV), torch.zeros(N, 3, device=DEV)
    count = torch.zeros(N, dtype=torch.int32, device=DEV)
    losses = []
    for it in range(40):
        opt.zero_grad(set_to_none=True)
        img, culled, uv = fused.rasterize(g, T, cam, **args)
        uv.retain_grad()
        loss = ssim_l1_loss(img, target, 0.2)
        loss.backward()
        opt.step()
        accumulate_grad_stats(uv.grad, culled, g.xyz.grad, cam, uv_acc, xyz_acc, count)
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert int(count.max()) == 40 and float(uv_acc.sum()) > 0 and torch.isfinite(xyz_acc).all()


def test_adam_step_unaligned_views_take_the_scalar_path():
    """parameters that are views at a 4-byte offset (not 16-byte aligned) and of odd length"""
    base = torch.randn(1003, generator=torch.Generator().manual_seed(9))
    grads = torch.randn(1002, generator=torch.Generator().manual_seed(10))
    p_ref = base[1:].clone().requires_grad_(True)
    p_ref.grad = grads.clone()
    ref = torch.optim.Adam([p_ref], lr=1e-2)
    store = base.to(DEV)
    p_hip = store[1:].detach().requires_grad_(True)      # data_ptr is 4 bytes past a 16-byte boundary
    assert p_hip.data_ptr() % 16 != 0
    p_hip.grad = grads.to(DEV)
    hip = Adam([p_hip], lr=1e-2)
    for _ in range(3):
        ref.step()
        hip.step()
    err = (p_hip.detach().cpu() - p_ref.detach()).abs().max() / p_ref.detach().abs().max()
    assert float(err) < 2e-6
    assert float(store[0]) == float(base[0])             # the element before the view is untouched


def test_stream_copy_is_a_copy_and_checks_its_arguments():
    """gs_stream_copy (bench.py's measured-bandwidth denominator): bytes arrive unchanged for a size that is not a
    multiple of the grid, any grid; misaligned pointers / sizes are refused with an error, not copied wrongly"""
    import ctypes

    from gaussian_splatting_amd import _hip
    n = 4 * 1_000_003   # floats: a multiple of 4 (16 bytes), nothing else
    a = torch.randn(n, device=DEV)
    p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    for blocks in (1, 7, 4096):
        b = torch.zeros(n + 8, device=DEV)
        _hip.call("gs_stream_copy", p(b), p(a), ctypes.c_size_t(n * 4), blocks, _hip.current_stream())
        assert torch.equal(b[:n], a) and not b[n:].any()
    b = torch.zeros(n, device=DEV)
    with pytest.raises(RuntimeError):
        _hip.call("gs_stream_copy", p(b, 4), p(a), ctypes.c_size_t(1024), 16, _hip.current_stream())
    with pytest.raises(RuntimeError):
        _hip.call("gs_stream_copy", p(b), p(a), ctypes.c_size_t(1000), 16, _hip.current_stream())
    _hip.call("gs_stream_copy", p(b), p(a), ctypes.c_size_t(0), 16, _hip.current_stream())   # nothing to do: fine
