"""Generates the golden fixtures in tests/golden/*.npz.  RUNS ONLY IN THE AUTHORING CONTAINER.

It imports the reference's own Python host code (/root/reference/splat_py: rasterize.py,
cuda_autograd_functions.py, tile_culling.py, depth.py) with the CPU oracle registered as the
`splat_cuda` module it expects, and records inputs + outputs.  The fixtures therefore pin
  (a) the reference's known-answer constants (copied as data from its tests), and
  (b) what the REFERENCE host pipeline returns over the oracle backend, which the host-side
      mirror in gaussian_splatting_amd/splat_py must reproduce exactly on CPU and which the HIP
      path must match on the GPU.
Nothing under /root/reference is read at test time; only the .npz files travel.

usage: python tests/golden/make_golden.py
"""
import ast
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import gs_oracle as orc  # noqa: E402

mod = types.ModuleType("splat_cuda")
for name in orc.SPLAT_CUDA_API:
    setattr(mod, name, getattr(orc, name))
sys.modules["splat_cuda"] = mod
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "test"))

from splat_py.rasterize import rasterize as ref_rasterize  # noqa: E402
from splat_py.depth import render_depth as ref_render_depth  # noqa: E402
from splat_py.structs import Gaussians as RefGaussians, Camera as RefCamera  # noqa: E402
from splat_py.utils import inverse_sigmoid_torch  # noqa: E402
from gaussian_test_data import get_test_data  # noqa: E402

from gaussian_splatting_amd.synthetic import make_scene, make_grad_image  # noqa: E402

orc.set_modes(0, 0)
orc.set_sh_band1_mode(0)


def np_(t):
    return None if t is None else t.detach().cpu().numpy()


def scene_arrays(g, cam, T):
    d = dict(xyz=np_(g.xyz), rgb=np_(g.rgb), opacity=np_(g.opacity), scale=np_(g.scale),
             quaternion=np_(g.quaternion), K=np_(cam.K), camera_T_world=np_(T),
             width=np.int64(cam.width), height=np.int64(cam.height))
    if g.sh is not None:
        d["sh"] = np_(g.sh)
    return d


# ---- (a) known answers held by the reference's tests (data, not code) ------------------------------
def known_answers():
    src = open(os.path.join(REF, "test", "test_tile_culling.py")).read()
    m = re.search(r"expected_sorted_gaussian_idx_by_splat_idx = torch\.tensor\(\s*(\[.*?\])\s*,\s*device", src, re.S)
    tile_list = np.array(ast.literal_eval(m.group(1)), dtype=np.int32)
    assert tile_list.shape[0] == 641
    np.savez_compressed(os.path.join(HERE, "ref_known_answers.npz"),
                        tile_culling_sorted=tile_list,
                        tile_culling_n_ranges=np.int64(1201))
    print("known answers:", tile_list.shape)


# ---- (b) the reference host over the oracle, 6-Gaussian scene (test/gaussian_test_data.py) -------------
def scene6():
    out = {}
    g, cam, T = get_test_data(torch.device("cpu"))
    out.update({"in_" + k: v for k, v in scene_arrays(g, cam, T).items()})
    g.opacity = inverse_sigmoid_torch(g.opacity)
    out["in_opacity_logit"] = np_(g.opacity)
    bg = torch.zeros(3)
    ys = np.arange(0, 480, 7)
    xs = np.arange(0, 640, 7)
    for mode, (sh, pre) in {"nosh": (False, True), "sh_pre": (True, True), "sh_pix": (True, False)}.items():
        g, cam, T = get_test_data(torch.device("cpu"))
        g.opacity = inverse_sigmoid_torch(g.opacity)
        if sh:
            g.sh = torch.ones((6, 3, 15)) * 0.1
        img, mask, uv = ref_rasterize(g, T, cam, 0.3, 100.0, 10, 3.0, pre, bg)
        out[f"{mode}_image_sub"] = np_(img)[np.ix_(ys, xs)]
        out[f"{mode}_image_sum"] = np.float64(np_(img).astype(np.float64).sum())
        out[f"{mode}_px_340_348"] = np_(img[340, 348])
        out[f"{mode}_px_200_348"] = np_(img[200, 348])
        out[f"{mode}_mask"] = np_(mask)
        out[f"{mode}_uv"] = np_(uv)
    g, cam, T = get_test_data(torch.device("cpu"))
    g.opacity = inverse_sigmoid_torch(g.opacity)
    d = ref_render_depth(g, 0.2, T, cam, 0.3, 10, 3.0)
    out["depth_sub"] = np_(d)[np.ix_(ys, xs)]
    out["depth_px"] = np.array([d[340, 348].item(), d[200, 348].item()], dtype=np.float32)
    out["sub_ys"] = ys
    out["sub_xs"] = xs
    np.savez_compressed(os.path.join(HERE, "ref_host_scene6.npz"), **out)
    print("scene6 written")


# ---- (c) the reference host over the oracle, seeded synthetic scenes, forward + backward ----------------
def synth(tag, N, W, H, deg, seed, use_pre, bgval):
    g, cam, T = make_scene(N, W, H, deg, seed=seed)
    params = {}
    for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh"):
        t = getattr(g, k)
        if t is not None:
            params[k] = t.clone().requires_grad_(True)
    rg = RefGaussians(params["xyz"], params["rgb"], params["opacity"], params["scale"], params["quaternion"],
                      params.get("sh"))
    rcam = RefCamera(cam.width, cam.height, cam.K)
    bg = torch.full((3,), bgval)
    img, mask, uv = ref_rasterize(rg, T, rcam, 2.0, 25.0, 5, 3.0, use_pre, bg)
    uv.retain_grad()
    gi = make_grad_image(W, H, seed=seed + 100)
    (img * gi).sum().backward()
    out = {"in_" + k: v for k, v in scene_arrays(g, cam, T).items()}
    out.update(image=np_(img), mask=np_(mask), uv=np_(uv), grad_image=np_(gi), grad_uv=np_(uv.grad),
               near=np.float64(2.0), far=np.float64(25.0), padding=np.int64(5), mh_dist=np.float64(3.0),
               background=np_(bg), use_sh_precompute=np.bool_(use_pre), sh_degree=np.int64(deg))
    for k, p in params.items():
        out["grad_" + k] = np_(p.grad)
    np.savez_compressed(os.path.join(HERE, f"ref_host_synth_{tag}.npz"), **out)
    print("synth", tag, "V =", int((~mask).sum()), "image mean", float(img.detach().mean()))


if __name__ == "__main__":
    known_answers()
    scene6()
    synth("deg0", 300, 96, 80, 0, 11, True, 0.0)
    synth("deg3_pre", 300, 96, 80, 3, 12, True, 0.5)
    synth("deg3_pix", 200, 64, 48, 3, 13, False, 0.5)
