"""Parity at BASELINE.json's full sizes (config B: 100 k Gaussians @ 1920x1080; config D: 2.86 M @
1297x840) through size-independent properties plus the CPU oracle on a few tile rows of the
full-size frame (the oracle over the whole frame would take too long for the suite)."""
import pytest
import torch

from gaussian_splatting_amd import fused
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

from .helpers import grad_errors, rel_err, report, scaled_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def oracle():
    from oracle import gs_oracle
    gs_oracle.set_modes(0, 0)
    gs_oracle.set_sh_band1_mode(0)
    return gs_oracle


def frame(workload, tile_rows=None, grad_scale=1.0, with_grad=True, seed=0, opacity_shift=0.0):
    N, W, H, deg = WORKLOADS[workload]
    g, cam, T = make_scene(N, W, H, deg, seed=seed, device=DEV)
    if opacity_shift:
        g.opacity.add_(opacity_shift)   # fainter Gaussians: pixels composite deeper into their lists
    if with_grad:
        for k in PARAMS:
            getattr(g, k).requires_grad_(True)
    bg = torch.full((3,), 0.5, device=DEV)
    img, mask, uv, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=tile_rows,
                                         return_aux=True, **DEFAULTS)
    grads = None
    if with_grad:
        for k in ("conic", "opacity", "rgb"):
            aux[k].retain_grad()
        uv.retain_grad()
        img.backward(make_grad_image(W, H, seed=1, device=DEV) * grad_scale)
        grads = {k: getattr(g, k).grad for k in PARAMS}
        grads.update(uv=uv.grad, conic=aux["conic"].grad, opacity_act=aux["opacity"].grad, rgb_render=aux["rgb"].grad)
    return img.detach(), mask, uv.detach(), aux, grads, (W, H)


def check_tile_lists(aux, W, H):
    ranges = aux["tile_ranges"].long()
    sorted_g = aux["sorted_gaussians"].long()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    assert ranges.shape[0] == T + 1 and int(ranges[0]) == 0 and int(ranges[-1]) == sorted_g.numel()
    counts = ranges[1:] - ranges[:-1]
    assert int(counts.min()) >= 0
    tile_of = torch.repeat_interleave(torch.arange(T, device=DEV), counts)
    z = aux["xyz_camera_frame"][:, 2][sorted_g]
    same = tile_of[1:] == tile_of[:-1]
    dz = z[1:] - z[:-1]
    assert bool(((dz >= 0) | ~same).all()), "a tile list is not sorted front to back"
    tie = same & (dz == 0)
    assert bool(((sorted_g[1:] > sorted_g[:-1]) | ~tie).all()), "depth ties must be in Gaussian order"
    assert int(sorted_g.min()) >= 0 and int(sorted_g.max()) < aux["conic"].shape[0]
    return int(counts.max())


def oracle_rows(aux, uv, W, H, rows, grad_image=None):
    """CPU oracle render (and backward) of tile rows [r0, r1) from the GPU's per-splat inputs"""
    orc = oracle()
    c = lambda t: t.detach().cpu().contiguous()
    uvc, conic, opa, rgb = c(uv), c(aux["conic"]), c(aux["opacity"]), c(aux["rgb"])
    ranges, sorted_g = c(aux["tile_ranges"]), c(aux["sorted_gaussians"])
    img = torch.zeros(H, W, 3)
    nsp = torch.zeros(H, W, dtype=torch.int32)
    fw = torch.zeros(H, W)
    bg = torch.full((3,), 0.5)
    orc.render_tiles_cuda(uvc, opa, rgb, conic, torch.zeros(1, 1, 1), ranges, sorted_g, bg, nsp, fw, img,
                          tile_rows=rows)
    out = dict(image=img, nsp=nsp)
    if grad_image is not None:
        V = uvc.shape[0]
        g = [torch.zeros(V, 3), torch.zeros(V, 1), torch.zeros(V, 2), torch.zeros(V, 3)]
        orc.render_tiles_backward_cuda(uvc, opa, rgb, conic, torch.zeros(1, 1, 1), ranges, sorted_g, bg, nsp, fw,
                                       grad_image.cpu().contiguous(), *g, tile_rows=rows)
        out.update(g_rgb=g[0], g_opa=g[1], g_uv=g[2], g_conic=g[3])
        # per element: sum of the magnitudes of its per-pixel terms (scale of its summation noise)
        a = [torch.zeros(V, 3), torch.zeros(V, 1), torch.zeros(V, 2), torch.zeros(V, 3)]
        orc.render_tiles_backward_abs(uvc, opa, rgb, conic, torch.zeros(1, 1, 1), ranges, sorted_g, bg, nsp, fw,
                                      grad_image.cpu().contiguous(), *a, tile_rows=rows)
        out.update(a_rgb=a[0], a_opa=a[1], a_uv=a[2], a_conic=a[3])
    return out


RENDER_GRADS = (("rgb_render", "g_rgb", "a_rgb"), ("opacity_act", "g_opa", "a_opa"), ("uv", "g_uv", "a_uv"),
                ("conic", "g_conic", "a_conic"))


def check_band_backward(tag, grads, ref):
    """render-backward gradients of a band against the oracle: the three element-wise measures are
    reported (tests/helpers.py: grad_errors); asserted: 1e-4 relative with the 1 % floor, 1e-5 of the
    tensor's scale everywhere, and -- floor-free, every element -- 2e-5 of the element's own
    sum of term magnitudes (summation-order noise + a few ulp per term)."""
    for name, key, akey in RENDER_GRADS:
        e = grad_errors(grads[name], ref[key], ref[akey])
        report(tag, tensor=name, **e)
        assert e["scaled"] < 1e-5, (name, e)
        assert e["rel_floor_1e-2"] < 1e-4, (name, e)
        assert e["noise_normalised"] < 2e-5, (name, e)


def test_config_B_full_size_properties():
    img, mask, uv, aux, grads, (W, H) = frame("B")
    max_list = check_tile_lists(aux, W, H)
    assert max_list > 60
    # determinism of the forward
    img2, _, _, _, _, _ = frame("B", with_grad=False)
    assert torch.equal(img, img2)
    # linearity of the backward in grad_image (scaling by 2 is exact in fp32; only the order of the
    # atomic accumulation differs between runs)
    _, _, _, _, grads2, _ = frame("B", grad_scale=2.0)
    for k in PARAMS:
        assert scaled_err(grads2[k], 2 * grads[k]) < 1e-5, k
    # culled Gaussians get exactly zero gradient rows
    assert 0 < int(mask.sum()) < mask.numel()
    for k in PARAMS:
        assert not grads[k][mask].any(), k
    # the oracle on three tile rows of the full-size frame: image bit-exact
    rows = (30, 33)
    ref = oracle_rows(aux, uv, W, H, rows)
    y0, y1 = rows[0] * 16, rows[1] * 16
    assert torch.equal(img[y0:y1].cpu(), ref["image"][y0:y1])


def test_config_B_band_backward_matches_oracle():
    """render backward restricted to tile rows [30, 33) at full size against the oracle on the same
    rows and the same per-splat inputs"""
    rows = (30, 33)
    img, mask, uv, aux, grads, (W, H) = frame("B", tile_rows=rows)
    gi = make_grad_image(W, H, seed=1)
    ref = oracle_rows(aux, uv, W, H, rows, gi)
    assert torch.equal(img.cpu(), ref["image"])
    check_band_backward("config_B_band_backward rows 30-33", grads, ref)


def test_config_C_full_size_properties():
    """1.5 M Gaussians @ 1297x840, SH degree 3 (BASELINE.json configs[2]): tile lists, the image on two
    tile rows bit-exact against the oracle, determinism, dense gradients finite and zero on culled rows"""
    img, mask, uv, aux, grads, (W, H) = frame("C")
    max_list = check_tile_lists(aux, W, H)
    assert max_list > 1000
    rows = (25, 27)
    ref = oracle_rows(aux, uv, W, H, rows)
    y0, y1 = rows[0] * 16, rows[1] * 16
    assert torch.equal(img[y0:y1].cpu(), ref["image"][y0:y1])
    img2, _, _, _, _, _ = frame("C", with_grad=False)
    assert torch.equal(img, img2)
    assert 0 < int(mask.sum()) < mask.numel()
    for k in PARAMS:
        assert torch.isfinite(grads[k]).all(), k
        assert not grads[k][mask].any(), k
        assert grads[k][~mask].any(), k


def test_config_C_band_backward_matches_oracle():
    """workload C, render forward + backward on tile rows [25, 27) against the oracle on the same
    per-splat inputs and tile lists"""
    rows = (25, 27)
    img, mask, uv, aux, grads, (W, H) = frame("C", tile_rows=rows)
    ref = oracle_rows(aux, uv, W, H, rows, make_grad_image(W, H, seed=1))
    assert torch.equal(img.cpu(), ref["image"])
    check_band_backward("config_C_band_backward rows 25-27", grads, ref)


def test_config_D_band_backward_matches_oracle():
    """workload D (2.86 M), tile rows [26, 28): ~2800 splats per tile list, the deepest pixel stops
    inside the first reference chunk (960), so this is the single-chunk backward at full size"""
    rows = (26, 28)
    img, mask, uv, aux, grads, (W, H) = frame("D", tile_rows=rows)
    ref = oracle_rows(aux, uv, W, H, rows, make_grad_image(W, H, seed=1))
    assert torch.equal(img.cpu(), ref["image"])
    y0, y1 = rows[0] * 16, rows[1] * 16
    deepest = int(ref["nsp"][y0:y1].max())
    report("config_D_band_backward rows 26-28", deepest_pixel=deepest)
    check_band_backward("config_D_band_backward rows 26-28", grads, ref)


def test_config_D_faint_band_backward_hits_the_second_reference_chunk():
    """workload D with every opacity logit lowered by 4: pixels composite > 960 splats deep, i.e. past
    the reference's first shared-memory chunk, where render_backward.cu:185 compares a chunk-local
    index with a global count.  Bug-compatible gradients at full size vs the oracle."""
    rows = (26, 27)
    img, mask, uv, aux, grads, (W, H) = frame("D", tile_rows=rows, opacity_shift=-4.0)
    ref = oracle_rows(aux, uv, W, H, rows, make_grad_image(W, H, seed=1))
    y0, y1 = rows[0] * 16, rows[1] * 16
    nsp = ref["nsp"][y0:y1]
    assert int(nsp.max()) > 960 and float((nsp > 960).float().mean()) > 0.05, int(nsp.max())
    report("config_D_faint_band_backward row 26", deepest_pixel=int(nsp.max()),
           pixels_past_first_chunk=float((nsp > 960).float().mean()))
    assert torch.equal(img.cpu(), ref["image"])
    check_band_backward("config_D_faint_band_backward row 26 (Q1 active)", grads, ref)


def test_config_D_full_size_properties():
    """2.86 M Gaussians: ~2800 splats per tile (multi-chunk render, 4096-key LDS sort class)"""
    img, mask, uv, aux, _, (W, H) = frame("D", with_grad=False)
    max_list = check_tile_lists(aux, W, H)
    assert max_list > 2000
    nty = (H + 15) // 16
    # two bands tile the frame bit-exactly
    top, _, _, _, _, _ = frame("D", tile_rows=(0, 20), with_grad=False)
    bot, _, _, _, _, _ = frame("D", tile_rows=(20, nty), with_grad=False)
    assert torch.equal(top + bot, img)
    # oracle on one tile row of the full-size frame
    rows = (26, 27)
    ref = oracle_rows(aux, uv, W, H, rows)
    assert torch.equal(img[rows[0] * 16:rows[1] * 16].cpu(), ref["image"][rows[0] * 16:rows[1] * 16])
    assert torch.isfinite(img).all()


def test_lds_sort_size_classes():
    """tiles of ~600, ~3000 and ~6000 instances exercise the three LDS sort classes"""
    from gaussian_splatting_amd import splat_cuda
    orc = oracle()
    gen = torch.Generator().manual_seed(11)
    chunks = []
    for tile_x, n in ((0, 600), (1, 3000), (2, 6000)):
        uvs = torch.rand(n, 2, generator=gen) * 10 + 3
        uvs[:, 0] += 16 * tile_x
        chunks.append(uvs)
    uv = torch.cat(chunks).contiguous()
    V = uv.shape[0]
    conic = torch.tensor([[0.5, 0.0, 0.5]]).repeat(V, 1).contiguous()
    xyz_c = torch.cat([torch.zeros(V, 2), 1 + 10 * torch.rand(V, 1, generator=gen)], dim=1).contiguous()
    xyz_c[::5, 2] = 3.0   # ties
    ref = orc.get_sorted_gaussian_list(1024, uv, xyz_c, conic, 3, 1, 3.0)
    got = splat_cuda.get_sorted_gaussian_list(1024, uv.to(DEV), xyz_c.to(DEV), conic.to(DEV), 3, 1, 3.0)
    counts = (ref[1][1:] - ref[1][:-1]).tolist()
    assert 500 < counts[0] <= 1024 < counts[1] <= 4096 < counts[2] <= 8192, counts
    assert torch.equal(got[1].cpu(), ref[1]) and torch.equal(got[0].cpu(), ref[0])


def test_prefix_sort_selects_the_nearest_entries():
    """gs_tile_emit_sort(sort_prefix=1024): tiles of 1024 < n <= 8192 entries get exactly the first
    1024 entries of the full order (radix select + sort), shorter and longer tiles the full order"""
    import ctypes
    from gaussian_splatting_amd import _hip
    orc = oracle()
    gen = torch.Generator().manual_seed(12)
    chunks, sizes = [], (700, 1025, 3000, 4097, 8192, 9000, 1, 40, 64, 65, 200, 256, 257, 1024)
    for tile_x, n in enumerate(sizes):
        uvs = torch.rand(n, 2, generator=gen) * 10 + 3
        uvs[:, 0] += 16 * tile_x
        chunks.append(uvs)
    uv = torch.cat(chunks).contiguous()
    V = uv.shape[0]
    conic = torch.tensor([[0.5, 0.0, 0.5]]).repeat(V, 1).contiguous()
    # depths: a wide range in one tile, a narrow one (keys differ in low bits only) in another, many ties
    z = 1 + 10 * torch.rand(V, 1, generator=gen)
    z[700:1725] = 5.0 + 1e-6 * torch.rand(1025, 1, generator=gen)
    z[::7] = 3.0
    xyz_c = torch.cat([torch.zeros(V, 2), z], dim=1).contiguous()
    ntx = len(sizes)
    ref_sorted, ref_ranges = orc.get_sorted_gaussian_list(1024, uv, xyz_c, conic, ntx, 1, 3.0)
    assert (ref_ranges[1:] - ref_ranges[:-1]).tolist() == list(sizes)
    # full mode (the drop-in function) over the same tiles: every size class of the wave-level sorts
    from gaussian_splatting_amd import splat_cuda
    full = splat_cuda.get_sorted_gaussian_list(1024, uv.to(DEV), xyz_c.to(DEV), conic.to(DEV), ntx, 1, 3.0)
    assert torch.equal(full[1].cpu(), ref_ranges) and torch.equal(full[0].cpu(), ref_sorted)

    p = lambda x: ctypes.c_void_p(x.data_ptr())
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    uv_d, xyz_d, conic_d = uv.to(DEV), xyz_c.to(DEV), conic.to(DEV)
    ws = torch.empty(_hip.lib().gs_tile_workspace_ints(ntx), dtype=torch.int32, device=DEV)
    ranges = torch.empty(ntx + 1, dtype=torch.int32, device=DEV)
    _hip.call("gs_tile_count", p(uv_d), p(conic_d), V, None, None, None, ntx, 1, ctypes.c_float(3.0), 0, 1, p(ws),
              p(ranges), None, stream)
    assert torch.equal(ranges.cpu(), ref_ranges)
    S = int(ranges[-1])
    keys = torch.empty(S, dtype=torch.int64, device=DEV)
    got = torch.full((S,), -1, dtype=torch.int32, device=DEV)
    _hip.call("gs_tile_emit_sort", p(uv_d), p(xyz_d), p(conic_d), V, None, None, None, ntx, 1, ctypes.c_float(3.0),
              0, 1, p(ranges), p(ws), p(keys), ctypes.c_int64(S), p(got), _hip.GS_SORT_PREFIX, stream)
    got = got.cpu()
    for t, n in enumerate(sizes):
        s0 = int(ref_ranges[t])
        m = 1024 if 1024 < n <= 8192 else n
        assert torch.equal(got[s0:s0 + m], ref_sorted[s0:s0 + m]), (t, n)
    # repair pass: flag two of the prefix tiles, they come back fully sorted, the others untouched
    flags = torch.tensor([0, 1, 0, 0, 1, 0] + [0] * 8, dtype=torch.int32, device=DEV)
    before = got.clone()
    got_d = got.to(DEV)
    _hip.call("gs_tile_sort_flagged", p(ranges), p(keys), ctypes.c_int64(S), p(flags), ntx, 0, 1, p(got_d), stream)
    got = got_d.cpu()
    for t, n in enumerate(sizes):
        s0 = int(ref_ranges[t])
        if flags[t]:
            assert torch.equal(got[s0:s0 + n], ref_sorted[s0:s0 + n]), (t, n)
        else:
            assert torch.equal(got[s0:s0 + n], before[s0:s0 + n]), (t, n)


def run_both_sort_modes(make, grad_image):
    out = {}
    for mode in (False, True):
        g, cam, T, kw = make()
        for k in PARAMS:
            if getattr(g, k) is not None:
                getattr(g, k).requires_grad_(True)
        fused.SORT_PREFIX = mode
        prev_cut, fused.DEPTH_CUT = fused.DEPTH_CUT, False   # the prefix SORT is what these tests are about
        try:
            img, mask, uv = fused.rasterize(g, T, cam, use_sh_precompute=True,
                                            background_rgb=torch.full((3,), 0.5, device=DEV), **kw)
        finally:
            fused.SORT_PREFIX = True
            fused.DEPTH_CUT = prev_cut
        uv.retain_grad()
        img.backward(grad_image)
        out[mode] = (img.detach(), {k: getattr(g, k).grad for k in PARAMS if getattr(g, k) is not None}, uv.grad)
    return out


def test_prefix_sort_mode_is_exact_on_config_D():
    """dense scene: every tile saturates inside its 1024-entry prefix; image identical, gradients
    equal up to summation order"""
    N, W, H, deg = WORKLOADS["D"]
    fused.last_flags(clear=True)
    out = run_both_sort_modes(lambda: make_scene(N, W, H, deg, seed=0, device=DEV) + (DEFAULTS,),
                              make_grad_image(W, H, seed=1, device=DEV))
    assert fused.last_flags() is not None and int(fused.last_flags().sum()) == 0
    assert torch.equal(out[True][0], out[False][0])
    for k in out[False][1]:
        assert scaled_err(out[True][1][k], out[False][1][k]) < 1e-5, k
    assert scaled_err(out[True][2], out[False][2]) < 1e-5


def test_prefix_sort_mode_repairs_tiles_that_need_more():
    """faint Gaussians: pixels need thousands of splats, so prefix tiles run out and are redone"""
    W = H = 128

    def make():
        g, cam, T = make_scene(120000, W, H, 0, seed=5, device=DEV)
        g.opacity.fill_(-5.0)
        return g, cam, T, DEFAULTS

    fused.last_flags(clear=True)
    out = run_both_sort_modes(make, make_grad_image(W, H, seed=2, device=DEV))
    flags = fused.last_flags()
    assert flags is not None and 0 < int(flags.sum())
    g, cam, T, kw = make()
    _, _, _, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=torch.zeros(3, device=DEV),
                                   return_aux=True, **kw)
    counts = (aux["tile_ranges"][1:] - aux["tile_ranges"][:-1])
    in_prefix_class = (counts > 1024) & (counts <= 8192)
    assert int(in_prefix_class.sum()) > 0 and bool((flags.bool() <= in_prefix_class).all())
    assert torch.equal(out[True][0], out[False][0])
    for k in out[False][1]:
        assert scaled_err(out[True][1][k], out[False][1][k]) < 1e-5, k


def test_config_D_faint_exact_mode_matches_the_oracle_exact_mode():
    """GS_BACKWARD_EXACT at full size where it matters (every pixel past the first reference chunk): against the
    oracle in its exact mode; and it differs from the compat gradients there"""
    from gaussian_splatting_amd import _hip
    rows = (26, 27)
    orc = oracle()
    try:
        _hip.set_backward_mode("exact")
        orc.set_backward_exact(1)
        img, mask, uv, aux, grads, (W, H) = frame("D", tile_rows=rows, opacity_shift=-4.0)
        ref = oracle_rows(aux, uv, W, H, rows, make_grad_image(W, H, seed=1))
    finally:
        _hip.set_backward_mode("compat")
        orc.set_backward_exact(0)
    assert torch.equal(img.cpu(), ref["image"])
    check_band_backward("config_D_faint_band_backward row 26 (exact mode)", grads, ref)
    _, _, _, _, compat, _ = frame("D", tile_rows=rows, opacity_shift=-4.0)
    assert scaled_err(compat["opacity_act"], grads["opacity_act"]) > 1e-3


def test_backward_mode_of_a_frame_is_the_mode_at_its_forward():
    """ABI 5: the render backward takes the gradient mode per call and a fused frame passes the default that was
    in force at its FORWARD -- switching the default while a backward is still to run (the autograd engine
    runs it on its own thread) cannot change that frame's gradients.  D-faint band: the two modes differ."""
    from gaussian_splatting_amd import _hip
    rows = (26, 27)
    N, W, H, deg = WORKLOADS["D"]
    gi = make_grad_image(W, H, seed=1, device=DEV)
    bg = torch.full((3,), 0.5, device=DEV)

    g0, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)   # once: eight scene builds were 20 s of this test
    g0.opacity.add_(-4.0)

    def run(mode_at_forward, mode_at_backward, aux):
        g = type(g0)(g0.xyz, g0.rgb, g0.opacity.detach().clone().requires_grad_(True), g0.scale, g0.quaternion, g0.sh)
        try:
            _hip.set_backward_mode(mode_at_forward)
            out = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows, return_aux=aux,
                                  **DEFAULTS)
            _hip.set_backward_mode(mode_at_backward)
            out[0].backward(gi)
        finally:
            _hip.set_backward_mode("compat")
        return g.opacity.grad

    for aux in (True, False):   # the Python orchestration and the native one (csrc/frame_hip.cpp)
        exact = run("exact", "exact", aux)
        compat = run("compat", "compat", aux)
        assert scaled_err(compat, exact) > 1e-3
        assert scaled_err(run("exact", "compat", aux), exact) < 1e-5
        assert scaled_err(run("compat", "exact", aux), compat) < 1e-5
