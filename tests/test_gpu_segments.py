"""The depth-segmented render backward (csrc/render.hip "depth segments": one workgroup per (tile, 128-entry
segment of its list), started from the state the forward left at the segment boundaries) against the
unsegmented kernel on the same frame, and against the CPU oracle on full-size bands."""
import pytest
import torch

from gaussian_splatting_amd import _hip, fused
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

from .helpers import scaled_err
from .test_gpu_scale import check_band_backward, frame, oracle, oracle_rows

pytestmark = pytest.mark.gpu
DEV = "cuda"
PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


@pytest.fixture
def segments_forced():
    prev = fused.SEGMENTS
    fused.SEGMENTS = True
    yield
    fused.SEGMENTS = prev


def stages(N, W, H, deg, seed, opacity_shift, sort_prefix):
    g, cam, T = make_scene(N, W, H, deg, seed=seed, device=DEV)
    if opacity_shift:
        g.opacity.add_(opacity_shift)
    d = DEFAULTS
    f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H, d["near_thresh"],
                                 d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None, sort_prefix)
    return f


# (N, W, H, degree, seed, opacity shift, background, tile rows): dense and deep (D, D-faint: up to 8 segments, the
# open-ended last one, Q1 active), unsaturated pixels over a background (sparse / faint: the q factor and the
# background weight of the first contributor), short lists (one segment per tile), a partial tile row and column
CASES = [
    ("D rows 24-30", WORKLOADS["D"] + (0, 0.0, 0.5, (24, 30))),
    ("D-faint rows 26-27", WORKLOADS["D"] + (0, -4.0, 0.5, (26, 27))),
    ("faint 128x128 over a background", (120000, 128, 128, 0, 5, -3.0, 0.5, None)),
    ("sparse 200x120 over a background", (4000, 200, 120, 1, 3, 0.0, 0.25, None)),
    ("B rows 30-34", WORKLOADS["B"] + (0, 0.0, 0.0, (30, 34))),
]


@pytest.mark.parametrize("mode", ["compat", "exact"])
@pytest.mark.parametrize("prefix", [True, False])
@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_segmented_backward_equals_the_unsegmented_backward(name, case, prefix, mode):
    N, W, H, deg, seed, shift, bgv, rows = case
    sort_prefix = _hip.GS_SORT_PREFIX if prefix else 0
    f = stages(N, W, H, deg, seed, shift, sort_prefix)
    V = f.V
    rgb_v = f.rgb_render[:V]
    bg = torch.full((3,), bgv, device=DEV)
    gi = make_grad_image(W, H, seed=seed + 1, device=DEV)
    code = {"compat": _hip.GS_BACKWARD_COMPAT, "exact": _hip.GS_BACKWARD_EXACT}[mode]
    out = {}
    for seg_on in (False, True):
        image, nsp, fw, cost, seg = fused.render_forward(f.packed, rgb_v, f.ranges, f.sorted_g, f.keys, bg, H, W, rows,
                                                         sort_prefix, segments=seg_on)
        assert (seg.numel() > 0) == seg_on
        slab = fused.render_backward(f.packed, rgb_v, f.ranges, f.sorted_g, bg, nsp, fw, gi, H, W, rows, V, None, code, seg)
        out[seg_on] = (image.clone(), nsp.clone(), fw.clone(), slab.clone())
    # the forward's own outputs do not depend on the extra state it leaves
    for a, b in zip(out[False][:3], out[True][:3]):
        assert torch.equal(a, b)
    plain, segmented = out[False][3], out[True][3]
    assert plain.abs().max() > 0
    for j in range(9):
        assert scaled_err(segmented[:, j], plain[:, j]) < 1e-5, (name, j, scaled_err(segmented[:, j], plain[:, j]))
    # rows nobody touches stay exactly zero in both
    assert torch.equal(plain.abs().sum(1) == 0, segmented.abs().sum(1) == 0)


@pytest.mark.parametrize("workload,rows,shift,mode", [("D", (26, 28), 0.0, "compat"), ("C", (25, 27), 0.0, "compat"),
                                                      ("D", (26, 27), -4.0, "compat"), ("D", (26, 27), -4.0, "exact")])
def test_segmented_band_backward_matches_the_oracle(segments_forced, workload, rows, shift, mode):
    """the full-size band checks of tests/test_gpu_scale.py with the segmented backward: image bit-exact,
    gradients against the oracle with the same three criteria"""
    orc = oracle()
    try:
        _hip.set_backward_mode(mode)
        orc.set_backward_exact(1 if mode == "exact" else 0)
        img, mask, uv, aux, grads, (W, H) = frame(workload, tile_rows=rows, opacity_shift=shift)
        ref = oracle_rows(aux, uv, W, H, rows, make_grad_image(W, H, seed=1))
    finally:
        _hip.set_backward_mode("compat")
        orc.set_backward_exact(0)
    assert torch.equal(img.cpu(), ref["image"])
    check_band_backward(f"segmented_band_backward[{workload} rows {rows[0]}-{rows[1]} shift {shift} {mode}]", grads, ref)


@pytest.mark.parametrize("rows", [None, (20, 27)])
def test_native_frame_with_segments(rows):
    """the default path (native orchestration, prefix sort) with and without segments: same image, parameter
    gradients equal up to fp32 rounding; a 1/8 band picks segments by itself ("auto")"""
    N, W, H, deg = WORKLOADS["D"]
    gi = make_grad_image(W, H, seed=1, device=DEV)
    bg = torch.full((3,), 0.5, device=DEV)
    res = {}
    prev = fused.SEGMENTS
    try:
        for setting in (False, True, "auto"):
            fused.SEGMENTS = setting
            g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
            for k in PARAMS:
                getattr(g, k).requires_grad_(True)
            img, mask, uv = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, tile_rows=rows, **DEFAULTS)
            uv.retain_grad()
            img.backward(gi)
            res[setting] = (img.detach(), {k: getattr(g, k).grad for k in PARAMS}, uv.grad)
    finally:
        fused.SEGMENTS = prev
    for setting in (True, "auto"):
        assert torch.equal(res[setting][0], res[False][0])
        for k in PARAMS:
            assert scaled_err(res[setting][1][k], res[False][1][k]) < 1e-5, (setting, k)
        assert scaled_err(res[setting][2], res[False][2]) < 1e-5
