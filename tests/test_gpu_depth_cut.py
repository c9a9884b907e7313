"""Depth-bucketed binning ("depth cut", csrc/binning.hip, include/gsplat_hip.h; replaces the emit / sort half of
get_sorted_gaussian_list, tile_culling.cu:124-340, for the fused renderer).

The cut must be invisible in every result: the kept list of a tile is a depth PREFIX of its complete list (the
reference's list), the image and num_splats are those of the complete lists bit for bit -- tiles that run out of
their kept prefix are repaired on the device from the complete list -- and the gradients differ by summation order
only.  Checked against the HIP path without the cut (itself bit-equal to the oracle at these sizes:
tests/test_gpu_fullsize_parity.py, tests/test_gpu_wholeframe_parity.py) and, for the lists, entry by entry."""
import ctypes

import pytest
import torch

from gaussian_splatting_amd import _hip, fused
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

from .helpers import report, scaled_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")
KCUT = 1024


def stage(g, cam, T, depth_cut):
    d = DEFAULTS
    return fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, cam.width, cam.height,
                                    d["near_thresh"], d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None,
                                    0 if not depth_cut else _hip.GS_SORT_PREFIX, depth_cut=depth_cut)


def cut_views(f):
    """b*(t), complete counts, bucket bounds, bucket offsets, control words as tensors (views of the cut workspace)"""
    lib = _hip.lib()
    ws = f.cut.cut_ws
    ptrs = [ctypes.c_void_p() for _ in range(5)]
    _hip.check(lib.gs_cut_debug_views(ctypes.c_void_p(ws.data_ptr()), f.N, f.T, *[ctypes.byref(p) for p in ptrs]))
    off = [(p.value - ws.data_ptr()) // 4 for p in ptrs]
    return (ws[off[0]:off[0] + f.T], ws[off[1]:off[1] + f.T], ws[off[2]:off[2] + 1024], ws[off[3]:off[3] + 1025],
            ws[off[4]:off[4] + 2])


def sortable(z):
    u = z.contiguous().view(torch.int32).long() & 0xffffffff
    return torch.where(u >= 0x80000000, (~u) & 0xffffffff, u | 0x80000000)


@pytest.mark.parametrize("name,N,W,H,deg,seed", [("dense", 400_000, 320, 240, 0, 3), ("D", *WORKLOADS["D"], 0)])
def test_cut_lists_are_depth_prefixes_of_the_complete_lists(name, N, W, H, deg, seed):
    g, cam, T = make_scene(N, W, H, deg, seed=seed, device=DEV)
    full = stage(g, cam, T, False)        # complete lists, sorted in full (bit-equal to the oracle's)
    cut = stage(g, cam, T, True)
    torch.cuda.synchronize()
    bstar, totals, bounds, boff, ctrl = (x.cpu() for x in cut_views(cut))
    nt = cut.T
    fr, fs = full.ranges.cpu().long(), full.sorted_g.cpu()
    cr, cs = cut.ranges.cpu().long(), cut.sorted_g.cpu()
    orng = cut.cut.full_ranges.cpu().long()
    V = full.V
    assert cut.V == V and cut.cut.S_full == full.S == int(fr[-1])
    assert torch.equal(orng, fr)                                   # complete ranges == the reference's ranges
    assert torch.equal(totals.long(), fr[1:] - fr[:-1])
    n_full, n_kept = fr[1:] - fr[:-1], cr[1:] - cr[:-1]
    assert int(n_kept.max()) <= KCUT and bool((n_kept <= n_full).all())
    assert bool((n_kept[n_full <= KCUT] == n_full[n_full <= KCUT]).all())   # short lists are kept whole
    truncated = n_kept < n_full
    assert int(truncated.sum()) > 0.5 * nt, "the scene should be dense enough to cut most tiles"
    # bucket structure: about equal populations, boundaries ascending, every Gaussian in exactly one bucket
    pop = boff[1:] - boff[:-1]
    assert int(boff[0]) == 0 and int(boff[-1]) == V and int(pop.min()) >= 0
    b64 = bounds.long() & 0xffffffff
    assert bool((b64[1:] >= b64[:-1]).all()) and int(b64[-1]) == 0xffffffff
    report(f"depth_cut_lists[{name}]", V=V, S_complete=int(fr[-1]), S_kept=int(cr[-1]), tiles=nt, truncated_tiles=int(truncated.sum()),
           kept_min_of_truncated=int(n_kept[truncated].min()), bucket_population_max_over_mean=float(pop.max()) / (V / 1024.0),
           deepest_wanted_bucket=int(ctrl[0]))
    assert float(pop.max()) < 3.0 * V / 1024.0
    # the kept list is the first n' entries of the complete sorted list, tile by tile (vectorised)
    tile_of = torch.repeat_interleave(torch.arange(nt), n_kept)
    within = torch.arange(int(cr[-1])) - cr[:-1][tile_of]
    assert torch.equal(cs.long(), fs.long()[fr[:-1][tile_of] + within])
    # ... and it ends exactly at a bucket boundary that is as deep as the 1024-entry limit allows: the next complete
    # entry lies in a deeper bucket, and taking that whole bucket would exceed the limit
    zkey = sortable(full.xyz_cam[:V, 2].cpu())
    bucket_of = torch.searchsorted(b64[:1023].contiguous(), zkey, right=False)   # first bound >= key
    tt = torch.nonzero(truncated).flatten()
    last_kept = fs.long()[fr[:-1][tt] + n_kept[tt] - 1]
    first_cut = fs.long()[fr[:-1][tt] + n_kept[tt]]
    has_kept = n_kept[tt] > 0
    assert bool((bucket_of[first_cut] > bstar.long()[tt]).all())
    assert bool((bucket_of[last_kept][has_kept] == bstar.long()[tt][has_kept]).all())
    # (count of the next bucket's entries of the tile, by walking the complete list: sampled tiles)
    for t in tt[:: max(1, len(tt) // 64)].tolist():
        lst = fs.long()[fr[t]:fr[t + 1]]
        nb = bucket_of[lst]
        nxt = int(nb[int(n_kept[t])])
        assert int(n_kept[t]) + int((nb == nxt).sum()) > KCUT, t
    assert int(ctrl[0]) == int(bstar.max())


def run_frame(g, cam, T, gi, depth_cut, native):
    prev = fused.DEPTH_CUT, fused.NATIVE
    fused.DEPTH_CUT, fused.NATIVE = depth_cut, native
    try:
        for k in PARAMS:
            p = getattr(g, k)
            if p is not None:
                p.grad = None
                p.requires_grad_(True)
        bg = torch.full((3,), 0.5, device=DEV)
        img, mask, uv = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
        uv.retain_grad()
        img.backward(gi)
        grads = {k: getattr(g, k).grad.clone() for k in PARAMS if getattr(g, k) is not None}
        return img.detach().clone(), mask.clone(), uv.detach().clone(), uv.grad.clone(), grads
    finally:
        fused.DEPTH_CUT, fused.NATIVE = prev


def same_frame(a, b, tol=1e-5):
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    errs = {"uv.grad": scaled_err(a[3], b[3])}
    for k in b[4]:
        errs[k] = scaled_err(a[4][k], b[4][k])
    assert max(errs.values()) < tol, errs
    return errs


@pytest.mark.parametrize("native", [False, True], ids=["python", "native"])
def test_cut_frame_equals_the_uncut_frame_on_workload_D(native):
    """dense scene: no tile needs more than its kept prefix; image bit-identical, gradients up to summation order"""
    N, W, H, deg = WORKLOADS["D"]
    g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
    gi = make_grad_image(W, H, seed=1, device=DEV)
    ref = run_frame(g, cam, T, gi, False, native)
    # (one cut frame that is not looked at: it brings the shape's capacity hints to THIS scene -- another test of the
    # same process may have left the hints of a sparser view of the same shape, and the miss that follows is the
    # policy working, not what this test is about; seen when test_gpu_wholeframe_parity.py ran first)
    run_frame(g, cam, T, gi, True, native)
    fused.last_flags(clear=True)
    fused.reset_counters()
    got = run_frame(g, cam, T, gi, True, native)
    got2 = run_frame(g, cam, T, gi, True, native)   # (native: the second frame runs on guessed capacities)
    flags = fused.last_flags()
    assert flags is not None and int(flags.sum()) == 0
    errs = same_frame(got, ref)
    same_frame(got2, ref)
    c = fused.counters()
    report(f"depth_cut_frame[D, {'native' if native else 'python'}]", frames=c["frames"],
           depth_cut_frames=c.get("depth_cut_frames", -1), max_grad_scaled_err=max(errs.values()))
    if native:
        assert c["depth_cut_frames"] == 2 and c["capacity_misses"] == 0


@pytest.mark.parametrize("native", [False, True], ids=["python", "native"])
@pytest.mark.parametrize("case", ["faint", "mixed"])
def test_cut_frame_repairs_tiles_that_need_more(case, native):
    """faint Gaussians: pixels composite thousands of splats deep, truncated tiles run out of their kept prefix and
    are redone on the device from their complete lists (overflow buffers); "mixed": only a part of the image is
    faint, so flagged and unflagged tiles coexist in one frame (the backward reads each from its own list)"""
    W, H = 256, 192
    g, cam, T = make_scene(300_000, W, H, 0, seed=5, device=DEV)
    if case == "faint":
        g.opacity.fill_(-5.0)
    else:
        left = g.xyz[:, 0] < 0
        g.opacity[left] = -5.5
    gi = make_grad_image(W, H, seed=2, device=DEV)
    ref = run_frame(g, cam, T, gi, False, native)
    fused.last_flags(clear=True)
    got = run_frame(g, cam, T, gi, True, native)
    flags = fused.last_flags()
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    assert flags is not None and 0 < int(flags.sum())
    if case == "mixed":
        assert int(flags.sum()) < nt
    errs = same_frame(got, ref)
    got2 = run_frame(g, cam, T, gi, True, native)
    same_frame(got2, ref)
    report(f"depth_cut_repair[{case}, {'native' if native else 'python'}]", flagged_tiles=int(flags.sum()), tiles=nt,
           max_grad_scaled_err=max(errs.values()))


def test_auto_policy_takes_the_cut_only_for_long_lists():
    """"auto": the first frame of a shape runs uncut (nothing is known about its lists); dense shapes then switch to
    the cut, sparse ones never do -- and neither do small dense frames (fewer than 1500 tiles with long lists): they
    take the depth-segmented backward, a cut frame takes the unsegmented one, and a shape that switched between the two
    whenever the cut policy changed its mind would change the last bits of its gradients from frame to frame (round-4
    advisor finding; csrc/frame_hip.cpp want_depth_cut)"""
    fused.reset_counters()
    prev = fused.DEPTH_CUT
    fused.DEPTH_CUT = "auto"
    try:
        for N, W, H, expect in ((1_200_000, 1024, 640, 2), (400_000, 320, 240, 0), (60_000, 640, 480, 0)):
            g, cam, T = make_scene(N, W, H, 0, seed=7, device=DEV)
            bg = torch.zeros(3, device=DEV)
            before = fused.counters().get("depth_cut_frames", 0)
            imgs = [fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)[0] for _ in range(3)]
            assert torch.equal(imgs[0], imgs[1]) and torch.equal(imgs[0], imgs[2])
            assert fused.counters().get("depth_cut_frames", 0) - before == expect, (N, W, H)
    finally:
        fused.DEPTH_CUT = prev


def test_cut_frame_with_too_small_capacity_guesses_is_repeated():
    """native orchestration: emit, sort and render (with its repair chain) are enqueued on capacities guessed from
    earlier frames before the host has the frame's counts; guesses that turn out too small (kept entries, or the
    complete count the overflow buffers are sized by) must be detected and the three steps repeated"""
    nat = fused.native()
    if nat is None:
        pytest.skip("native frame module not built")
    W, H = 256, 192
    g, cam, T = make_scene(300_000, W, H, 0, seed=5, device=DEV)
    left = g.xyz[:, 0] < 0
    g.opacity[left] = -5.5            # flagged and unflagged tiles in one frame: the overflow buffers are used
    gi = make_grad_image(W, H, seed=2, device=DEV)
    ref = run_frame(g, cam, T, gi, False, True)
    first = run_frame(g, cam, T, gi, True, True)
    same_frame(first, ref)
    fused.reset_counters()
    for factor in (0.02, 1.0, 0.5, 50.0):
        nat.debug_scale_capacity_hints(factor)
        same_frame(run_frame(g, cam, T, gi, True, True), ref)
    c = fused.counters()
    assert c["depth_cut_frames"] == 4 and c["capacity_misses"] >= 2, c


def test_auto_policy_backs_off_when_most_tiles_need_their_complete_lists():
    """"auto" (native orchestration): a faint scene flags every truncated tile, i.e. the cut emits most lists twice;
    the flagged-tile count of a cut frame reaches the host through a pinned word and switches the cut off for the
    next frames of that shape.  Results are the same either way."""
    nat = fused.native()
    if nat is None:
        pytest.skip("native frame module not built")
    # (1900 tiles: a frame of fewer than 1500 tiles with lists this long keeps the depth segments and is never cut
    # by "auto", csrc/frame_hip.cpp want_depth_cut)
    W, H = 800, 608
    g, cam, T = make_scene(800_000, W, H, 0, seed=11, device=DEV)
    g.opacity.fill_(-5.0)
    gi = make_grad_image(W, H, seed=2, device=DEV)
    ref = run_frame(g, cam, T, gi, False, True)
    fused.reset_counters()
    for frame in range(5):
        same_frame(run_frame(g, cam, T, gi, "auto", True), ref)
        torch.cuda.synchronize()   # (the policy never waits for the word; the test makes its arrival deterministic)
    c = fused.counters()
    # frame 0 of the shape ran uncut as `ref`; frame 1 here takes the cut, is flagged all over, and the rest back off
    assert c["depth_cut_frames"] == 1 and c["depth_cut_backoffs"] == 1, c


def degenerate_scene(case):
    """scenes whose depth distribution defeats the bucket boundaries (sample quantiles): correctness may not depend on
    the boundaries being any good -- only the balance of the workgroups and how many tiles get repaired do"""
    W, H, N = 256, 192, 300_000
    if case == "few":
        W, H, N = 64, 48, 3_000          # most of the 1024 buckets empty, the 256 partition workgroups nearly so
    if case == "long_partition_chunks":
        W, H, N = 160, 128, 3_400_000    # > 12 288 visible Gaussians per partition workgroup: the unstaged scatter
    g, cam, T = make_scene(N, W, H, 0, seed=11, device=DEV)
    z = g.xyz[:, 2]
    if case == "one_depth":
        z.fill_(6.0)                     # one bucket holds everything: no tile can be cut, all are repaired
    elif case == "two_depths":
        z.copy_(torch.where(torch.arange(N, device=DEV) % 2 == 0, 4.0, 9.0))
    elif case == "quantized":
        z.copy_(torch.round(z * 8) / 8)  # ~230 distinct depths: ties inside and across bucket boundaries
    elif case == "sampled_outliers":
        # every Gaussian the depth histogram samples sits at one depth, all the others behind it: the boundaries
        # collapse onto the samples' depth, bucket 0 = the samples, the last bucket = everything else
        stride = _hip.lib().gs_cut_sample_stride(N)
        z.copy_(torch.where(torch.arange(N, device=DEV) % stride == 0, 2.0, 5.0 + 20.0 * torch.rand(N, device=DEV)))
    elif case == "long_runs":
        # 1/16 of the Gaussians (none of them sampled ones) crowd into a depth interval the boundaries do not
        # resolve: a kept list holds unordered runs far longer than the run-aware sort settles in its pass budget
        stride = _hip.lib().gs_cut_sample_stride(N)
        idx = torch.arange(N, device=DEV)
        crowd = (idx % 16 == 1) & (idx % stride != 0)
        z.copy_(torch.where(crowd, 1.6 + 0.01 * torch.rand(N, device=DEV), 8.0 + 20.0 * torch.rand(N, device=DEV)))
    elif case == "none_visible":
        z.fill_(-1.0)
    return g, cam, T, W, H


@pytest.mark.parametrize("native", [False, True], ids=["python", "native"])
@pytest.mark.parametrize("case", ["one_depth", "two_depths", "quantized", "sampled_outliers", "long_runs", "few",
                                  "none_visible", "long_partition_chunks"])
def test_cut_frame_with_degenerate_depth_distributions(case, native):
    g, cam, T, W, H = degenerate_scene(case)
    gi = make_grad_image(W, H, seed=4, device=DEV)
    ref = run_frame(g, cam, T, gi, False, native)
    fused.last_flags(clear=True)
    fused.reset_counters()
    got = run_frame(g, cam, T, gi, True, native)
    flags = fused.last_flags()
    errs = same_frame(got, ref)
    got2 = run_frame(g, cam, T, gi, True, native)   # (native: guessed capacities)
    same_frame(got2, ref)
    c = fused.counters()
    if native and case != "none_visible":
        assert c.get("depth_cut_frames", 0) >= 1, c
    report(f"depth_cut_degenerate[{case}, {'native' if native else 'python'}]",
           flagged_tiles=-1 if flags is None else int(flags.sum()), tiles=((W + 15) // 16) * ((H + 15) // 16),
           max_grad_scaled_err=max(errs.values()))


@pytest.mark.parametrize("case", ["dense_cut", "faint_cut_repair", "mixed_cut_repair", "prefix_no_cut", "band_segments"])
def test_touch_masks_from_the_forward_change_no_result(case):
    """The native frame's backward reads the touch masks its forward left (`_m` render entries, ABI 8) instead of
    rebuilding them.  Same frame with and without the hand-off: image bit-identical and gradients up to summation
    order -- on kept depth prefixes, on tiles repaired from their complete lists (the repair pass rewrites their
    words), on prefix-sorted complete lists deeper than the 1024 entries the buffer covers (the backward builds
    the words beyond itself), and on a band with the depth-segmented backward."""
    nat = fused.native()
    if nat is None:
        pytest.skip("native frame module not built")
    W, H = (320, 240) if case != "band_segments" else (640, 480)
    N = {"dense_cut": 400_000, "faint_cut_repair": 300_000, "mixed_cut_repair": 300_000, "prefix_no_cut": 400_000,
         "band_segments": 600_000}[case]
    g, cam, T = make_scene(N, W, H, 0, seed=9, device=DEV)
    if case == "faint_cut_repair":
        g.opacity.fill_(-5.0)
    if case == "mixed_cut_repair":
        g.opacity[g.xyz[:, 0] < 0] = -5.5
    if case == "prefix_no_cut":
        g.opacity.fill_(-4.0)   # pixels composite deeper than 1024 entries: words beyond the buffer
    gi = make_grad_image(W, H, seed=3, device=DEV)
    cut = case.endswith("cut") or case.endswith("repair")
    rows = (8, 14) if case == "band_segments" else None

    def frame(masks_on):
        nat.set_touch_masks(masks_on)
        prev = fused.DEPTH_CUT, fused.NATIVE
        fused.DEPTH_CUT, fused.NATIVE = cut, True
        try:
            for k in PARAMS:
                p = getattr(g, k)
                if p is not None:
                    p.grad = None
                    p.requires_grad_(True)
            bg = torch.full((3,), 0.5, device=DEV)
            img, mask, uv = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows, **DEFAULTS)
            img.backward(gi)
            return img.detach().clone(), {k: getattr(g, k).grad.clone() for k in PARAMS if getattr(g, k) is not None}
        finally:
            fused.DEPTH_CUT, fused.NATIVE = prev

    try:
        frame(True)   # (capacity hints of the shape)
        img1, g1 = frame(True)
        img0, g0 = frame(False)
    finally:
        nat.set_touch_masks(True)
    assert torch.equal(img1, img0)
    errs = {k: scaled_err(g1[k], g0[k]) for k in g0}
    assert max(errs.values()) < 1e-5, errs
    assert any(bool((v != 0).any()) for v in g1.values())
    report(f"touch_mask_handoff[{case}]", max_grad_scaled_err=max(errs.values()))
