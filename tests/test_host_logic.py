"""Host-side logic that needs neither GPU nor kernels: data contracts, ray helpers, argument
validation of the splat_cuda shim (reference: test/test_structs.py, test/test_utils.py,
src/checks.cuh)."""
import math

import pytest
import torch

from gaussian_splatting_amd import splat_cuda
from gaussian_splatting_amd.splat_py.structs import Camera, Gaussians, Tiles
from gaussian_splatting_amd.splat_py.utils import (compute_rays, compute_rays_in_world_frame,
                                                   transform_points_torch)

from .helpers import scene6


def test_tiles_1080p():
    """test/test_structs.py:10-26"""
    tiles = Tiles(1080, 1920, torch.device("cpu"))
    assert (tiles.image_height_padded, tiles.image_width_padded) == (1088, 1920)
    assert (tiles.y_tiles_count, tiles.x_tiles_count, tiles.tile_count) == (68, 120, 8160)


def test_transform_points():
    """test/test_utils.py:28-46"""
    pts = torch.arange(1.0, 10.0).reshape(-1, 3)
    s = math.sqrt(2) / 2
    transform = torch.eye(4)
    # rotation of q = (0, s, 0, s) (w, x, y, z): 180 deg about (x+z)/sqrt2
    q = torch.tensor([0.0, s, 0.0, s])
    w, x, y, z = q
    R = torch.tensor([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                      [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                      [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])
    transform[:3, :3] = R
    transform[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    out = transform_points_torch(pts, transform)
    expected = torch.tensor([4.0, 0.0, 4.0, 7.0, -3.0, 7.0, 10.0, -6.0, 10.0]).reshape(-1, 3)
    assert out.allclose(expected, atol=1e-6)
    back = transform_points_torch(out, torch.inverse(transform))
    assert back.allclose(pts, atol=1e-5)


def test_rays_known_answers():
    """test/test_utils.py:47-90"""
    _, cam, T, _ = scene6()
    rays = compute_rays(cam).reshape(480, 640, 3)
    exp = {(0, 0): (-0.5403921008110046, -0.4250645041465759, 0.7261518836021423),
           (240, 320): (0.0, 0.0, 1.0),
           (0, 639): (0.5391948819160461, -0.425452321767807, 0.7268144488334656)}
    for (v, u), e in exp.items():
        for k in range(3):
            assert abs(rays[v, u, k].item() - e[k]) < 1e-6
    w = compute_rays_in_world_frame(cam, T)
    assert w.shape == (480, 640, 3)
    exp = {(0, 0): (-0.5390445590019226, -0.6224945187568665, 0.5673900842666626),
           (240, 320): (-0.004399406723678112, -0.2905626893043518, 0.9568459391593933),
           (0, 639): (0.540492832660675, -0.6134769916534424, 0.5757721662521362)}
    for (v, u), e in exp.items():
        for k in range(3):
            assert abs(w[v, u, k].item() - e[k]) < 1e-6


def test_gaussians_filter_append():
    g, _, _, _ = scene6()
    keep = torch.tensor([True, False, True, True, False, True])
    g.filter_in_place(keep)
    assert len(g) == 4 and g.opacity.shape == (4, 1)
    g.append(g.xyz, g.rgb, g.opacity, g.scale, g.quaternion)
    assert len(g) == 8
    with pytest.raises(AssertionError):
        Gaussians(torch.zeros(3, 3), torch.zeros(3, 3), torch.zeros(3), torch.zeros(3, 3), torch.zeros(3, 4))


def test_shim_rejects_cpu_tensors():
    """src/checks.cuh:5: non-device tensors are an error (RuntimeError), never a silent fallback"""
    xyz, K, uv = torch.zeros(4, 3), torch.eye(3), torch.zeros(4, 2)
    with pytest.raises(RuntimeError, match="not a CUDA tensor"):
        splat_cuda.camera_projection_cuda(xyz, K, uv)
    with pytest.raises(RuntimeError, match="not a CUDA tensor"):
        splat_cuda.get_sorted_gaussian_list(1024, uv, xyz, xyz, 4, 4, 3.0)
    with pytest.raises(RuntimeError, match="not a CUDA tensor"):
        splat_cuda.render_depth_cuda(xyz, uv, uv, xyz, uv, uv, 0.2, uv)


def test_get_splats_nan_guard(oracle_backend):
    from gaussian_splatting_amd.splat_py.tile_culling import get_splats
    bad = torch.tensor([[0.0, float("nan"), 1.0]])
    with pytest.raises(FloatingPointError):
        get_splats(torch.zeros(1, 2), Tiles(32, 32, "cpu"), torch.ones(1, 3), bad, 3.0)


def test_product_path_fails_loudly_without_hip_library(monkeypatch):
    """no CPU fallback: a missing libgsplat_hip.so is an error, never a silent oracle/PyTorch path"""
    from gaussian_splatting_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libgsplat_hip.so")
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        _hip.lib()
    with pytest.raises(_hip.HipLibraryError):
        _hip.call("gs_abi_version")


def test_product_package_never_imports_the_oracle():
    """static check: nothing under gaussian_splatting_amd/ refers to oracle/"""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussian_splatting_amd")
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "gs_oracle" not in src, f


def test_train_ops_adam_hands_cpu_tensors_to_torch():
    """gaussian_splatting_amd.train_ops.Adam covers fp32 device tensors with the HIP kernel; anything
    else goes through torch.optim.Adam.step itself (never the oracle)"""
    import torch
    from gaussian_splatting_amd.train_ops import Adam
    a = torch.arange(6, dtype=torch.float32).view(2, 3).requires_grad_(True)
    b = a.detach().clone().requires_grad_(True)
    a.grad = torch.full_like(a, 0.5)
    b.grad = torch.full_like(b, 0.5)
    Adam([a], lr=0.01).step()
    torch.optim.Adam([b], lr=0.01).step()
    assert torch.equal(a, b)


def test_quaternion_to_rotation_and_scale_init():
    """test/test_utils.py:15-25 (rotation matrices are orthonormal) + the two identities, and the
    batched KD-tree initial scale against the reference's per-point formula"""
    import math
    import numpy as np
    import torch
    from scipy.spatial import KDTree
    from gaussian_splatting_amd.splat_py.utils import (compute_initial_scale_from_sparse_points, inverse_sigmoid,
                                                       inverse_sigmoid_torch, quaternion_to_rotation_torch)
    q = torch.tensor([1.0, 0.0, 0.0, 0.0, 0.0, math.sqrt(2) / 2, 0.0, math.sqrt(2) / 2]).reshape(-1, 4)
    R = quaternion_to_rotation_torch(q)
    assert R.shape == (2, 3, 3)
    assert torch.allclose(torch.bmm(R, R.transpose(1, 2)), torch.eye(3).repeat(2, 1, 1), atol=1e-6)
    assert torch.allclose(R[0], torch.eye(3))
    assert torch.allclose(R[1], torch.tensor([[0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 0.0, 0.0]]), atol=1e-6)
    x = np.array([0.0, 0.3, 1.0])
    assert np.allclose(inverse_sigmoid(x), inverse_sigmoid_torch(torch.from_numpy(x)).numpy())
    pts = torch.rand(50, 3, generator=torch.Generator().manual_seed(0))
    got = compute_initial_scale_from_sparse_points(pts, 4, 0.5, 0.1)
    tree = KDTree(pts.numpy())
    for i in (0, 7, 49):
        d, _ = tree.query(pts[i].numpy(), k=4)
        assert np.allclose(got[i].numpy(), np.log(min(np.mean(d), 0.1) * 0.5), atol=1e-6)
    assert got.shape == (50, 3) and got.dtype == torch.float32


def test_fused_slab_detection_and_arena():
    """fused._as_slab: gradients that are the four views of one [V, 9] slab are passed through without a
    copy; anything else (missing outputs, foreign tensors) is packed into a fresh slab.  _Arena: blocks
    are disjoint, 16-byte aligned and sized as asked."""
    import torch
    from gaussian_splatting_amd import fused
    V = 7
    slab = torch.arange(V * 9, dtype=torch.float32).view(V, 9).clone()   # a base tensor, as torch.zeros gives
    views = (slab[:, fused.SLAB_UV], slab[:, fused.SLAB_CONIC], slab[:, fused.SLAB_OPACITY], slab[:, fused.SLAB_RGB])
    assert fused._as_slab(*views, V, slab.device) is slab
    # the prefix view render_backward hands out (slab allocated with max(V, 1) rows)
    big = torch.zeros(V + 3, 9)
    pre = big[:V]
    got = fused._as_slab(pre[:, fused.SLAB_UV], pre[:, fused.SLAB_CONIC], pre[:, fused.SLAB_OPACITY],
                         pre[:, fused.SLAB_RGB], V, big.device)
    assert got is big
    # a consumer replaced one gradient, another output was never used
    g_uv = torch.ones(V, 2)
    packed = fused._as_slab(g_uv, views[1], None, views[3], V, slab.device)
    assert packed is not slab and packed.shape == (V, 9)
    assert torch.equal(packed[:, 4:6], g_uv) and torch.equal(packed[:, 6:9], slab[:, 6:9])
    assert torch.equal(packed[:, 3], torch.zeros(V)) and torch.equal(packed[:, 0:3], slab[:, 0:3])

    ar = fused._Arena(torch.float32, torch.device("cpu"), (3, 10, 0, 5))
    blocks = ar.blocks()
    assert [b.numel() for b in blocks] == [3, 10, 0, 5]
    starts = [b.data_ptr() for b in blocks]
    assert all(p % 16 == 0 for p in starts)
    blocks[0].fill_(1.0); blocks[1].fill_(2.0); blocks[3].fill_(3.0)
    assert float(blocks[0].sum()) == 3 and float(blocks[1].sum()) == 20 and float(blocks[3].sum()) == 15


def test_depth_cut_policy_and_size_helpers(monkeypatch):
    """fused.want_depth_cut ("auto": whole frames in the LDS-histogram regime whose lists averaged >= 1 280 entries per
    tile in an earlier frame of the shape) and the C ABI's size helpers of the depth-bucketed binning (no GPU needed:
    they only compute)."""
    import ctypes
    from gaussian_splatting_amd import _hip, fused
    lib = _hip.lib()
    lib.gs_cut_workspace_ints.restype = ctypes.c_size_t
    # every 44th Gaussian of workload D is sampled (<= 65 536 samples), small scenes sample everything
    assert lib.gs_cut_sample_stride(2_860_000) == 44 and lib.gs_cut_sample_stride(65_536) == 1
    assert lib.gs_cut_sample_stride(65_537) == 2 and lib.gs_cut_sample_stride(1) == 1
    # LDS-histogram regime: at most 16 384 tiles in the row range and more than 128 Gaussians per tile
    ntx, nty = 82, 53
    assert lib.gs_cut_supported(ntx, 0, nty, 2_860_000) == 1
    assert lib.gs_cut_supported(ntx, 0, nty, 128 * ntx * nty) == 0 and lib.gs_cut_supported(ntx, 0, nty, 128 * ntx * nty + 1) == 1
    assert lib.gs_cut_supported(240, 0, 135, 50_000_000) == 0        # 32 400 tiles (4K): not in one range ...
    assert lib.gs_cut_supported(240, 0, 64, 50_000_000) == 1         # ... a band of it is
    assert lib.gs_cut_supported(ntx, 5, 5, 2_860_000) == 0           # empty range
    # workspace: grows with N and T, holds at least the index list, the bucket ids and the 256 x 1024 count matrix
    w = lib.gs_cut_workspace_ints(2_860_000, ntx * nty)
    assert w >= 2_860_000 + 2_860_000 // 2 + 256 * 1024 + 2 * ntx * nty
    assert lib.gs_cut_workspace_ints(2_860_001, ntx * nty) >= w and lib.gs_cut_workspace_ints(2_860_000, ntx * nty + 1) >= w
    assert w < 2 * (2_860_000 + 2_860_000 // 2 + 256 * 1024 + 2 * ntx * nty)

    key = ("shape",)
    T = ntx * nty
    monkeypatch.setattr(fused, "DEPTH_CUT", "auto")
    monkeypatch.setattr(fused, "_mean_list_hint", {})
    assert not fused.want_depth_cut(key, 2_860_000, ntx, 0, nty, True)            # nothing known about the shape yet
    fused._mean_list_hint[key] = fused.DEPTH_CUT_MIN_MEAN_LIST * T - 1
    assert not fused.want_depth_cut(key, 2_860_000, ntx, 0, nty, True)
    fused._mean_list_hint[key] = fused.DEPTH_CUT_MIN_MEAN_LIST * T
    assert fused.want_depth_cut(key, 2_860_000, ntx, 0, nty, True)
    assert not fused.want_depth_cut(key, 2_860_000, ntx, 0, nty, False)           # bands keep the complete lists
    assert not fused.want_depth_cut(key, 100_000, ntx, 0, nty, True)              # outside the regime
    monkeypatch.setattr(fused, "DEPTH_CUT", True)
    assert fused.want_depth_cut(("other",), 2_860_000, ntx, 0, nty, True)         # forced: no history needed ...
    assert not fused.want_depth_cut(("other",), 100_000, ntx, 0, nty, True)       # ... but still only where supported
    monkeypatch.setattr(fused, "DEPTH_CUT", False)
    assert not fused.want_depth_cut(key, 2_860_000, ntx, 0, nty, True)


def test_bench_helpers():
    """bench.py's host-side pieces: the frame's algorithmic bytes are SURVEY.md 8(d)'s formula, camera poses
    are seeded rigid transforms, the median, and the self-spawn command line"""
    import importlib
    import sys
    import types
    bench = importlib.import_module("bench")
    N, V, S, P = 1000, 900, 4000, 50000
    per = bench.algorithmic_bytes(N, V, S, P, 16)
    assert per["frame"] == 248 * N + 384 * V + 144 * S + 40 * P
    per0 = bench.algorithmic_bytes(N, V, S, P, 1)
    assert per0["frame"] == 68 * N + 204 * V + 144 * S + 40 * P
    assert per["gs_render_tiles_backward"] == 76 * S + 20 * P
    poses = bench.camera_poses(5, 7, "cpu", moving=True)
    again = bench.camera_poses(5, 7, "cpu", moving=True)
    assert all(torch.equal(a, b) for a, b in zip(poses, again)) and not torch.equal(poses[0], poses[1])
    for M in poses:
        R = M[:3, :3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-6) and abs(float(torch.det(R)) - 1) < 1e-6
        assert torch.equal(M[3], torch.tensor([0.0, 0.0, 0.0, 1.0]))
    assert all(torch.equal(M, torch.eye(4)) for M in bench.camera_poses(3, 7, "cpu", moving=False))
    assert bench.median([3, 1, 2]) == 2 and bench.median([4, 1, 2, 3]) == 2.5 and bench.median([]) == 0.0
    # --gpus N without a launcher: the command re-executes bench.py under torch.distributed.run with N ranks
    seen = {}
    real = sys.modules.get("subprocess")
    fake = types.SimpleNamespace(call=lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    sys.modules["subprocess"] = fake
    argv = sys.argv
    sys.argv = ["bench.py", "--gpus", "4", "--steps", "3"]
    try:
        assert bench.spawn_ranks(types.SimpleNamespace(gpus=4)) == 0
    finally:
        sys.argv = argv
        sys.modules["subprocess"] = real
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert "127.0.0.1" in cmd and cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


@pytest.mark.parametrize("rnd", ["r02", "r04", "r05"])
def test_committed_traffic_and_valu_json_follow_from_the_committed_pmc_passes(tmp_path, rnd):
    """profiles/<round>/hbm_traffic_D.json and valu_insts_D.json (what bench.py reports as roofline.traffic /
    roofline.valu) are exactly what scripts/make_traffic_json.py derives from the committed rocprofv3 passes
    (r04: the frame binned with the depth cut -- other kernels behind the same entry points)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out, valu = tmp_path / "t.json", tmp_path / "v.json"
    subprocess.run([sys.executable, os.path.join(root, "scripts", "make_traffic_json.py"), f"profiles/{rnd}/pmc_bench_D",
                    str(out), "D", str(valu)], check=True, cwd=root, capture_output=True)
    assert json.load(open(out)) == json.load(open(os.path.join(root, "profiles", rnd, "hbm_traffic_D.json")))
    assert json.load(open(valu)) == json.load(open(os.path.join(root, "profiles", rnd, "valu_insts_D.json")))
    if rnd in ("r04", "r05"):
        assert json.load(open(out))["binning"] == "depth cut"


@pytest.mark.parametrize("name", ["r02/bench_D.json", "r02/bench_B.json", "r02/bench_C.json",
                                  "r02/bench_D_moving_camera.json", "r04/bench_D.json", "r05/bench_D.json"])
def test_committed_bench_lines_keep_the_contract(name):
    """the JSON lines under profiles/ are what `python bench.py` printed: the contract's fields, BASELINE.json's
    metric verbatim, a roofline object whose fraction is achieved / peak and whose achieved rate is the algorithmic
    bytes over the measured launch duration, and (workload D, default flags) the CPU baseline"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", name)))
    name = os.path.basename(name)
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "Mpixels/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["steps"] >= 50 and "workload" in d["config"] and "model" not in d["config"]
    P = d["config"]["P"]
    assert abs(d["value"] - P / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["launch_ms"] * 1e-3) / 1e9) < 1e-2 * r["achieved"]
    assert r["kernel"] in r["entry_ms_per_step"] and r["launch_ms"] <= d["ms_per_step"]
    assert sum(r["entry_ms_per_step"].values()) <= d["ms_per_step"] * 1.02      # the entries fit inside the step
    if name == "bench_D.json":
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mpixels/s" and c["sample"]
        assert d["parity"]["image_max_abs_err"] == 0.0 and d["parity"]["grad_max_rel_err"] < 1e-4
        assert r["traffic"] is not None and r["traffic_source"].startswith("profiles/")
    if name == "bench_D.json" and "hbm_achievable_gbs_guide" in r:   # round 5 on
        assert r["hbm_achievable_gbs_guide"] == 6290.0 and "gs_stream_copy" in r["hbm_copy_kernel"]
        assert abs(r["frame_frac_of_achievable"] - r["frame_frac"] * 8000.0 / 6290.0) < 1e-3
        h = d["parity"]["headline"]
        assert h["grad_max_rel_err_floor_1e-2"] < h["target"] == 1e-4
        assert abs(h["ratio_kernel_to_pure_fp32_reorder_spread"]
                   - h["grad_max_rel_err_floor_1e-6"] / h["fp32_reorder_spread_floor_1e-6"]) < 1e-9
        assert h["ratio_kernel_to_pure_fp32_reorder_spread"] <= 2.0   # the gate tests/test_gpu_wholeframe_parity.py asserts


def test_committed_training_trace_converges():
    """profiles/r05/train_loop_convergence.json is what `bench.py --train-loop 7000` printed: the config-5-sized loop
    fits its training views at every mark; the view-determined problem also rises on the held-out views"""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tl = json.load(open(os.path.join(root, "profiles", "r05", "train_loop_convergence.json")))["train_loop"]
    for run in (tl, tl["well_posed_problem"]):
        c = run["convergence"]
        assert run["iterations"] == 7000 and c["monotone_train_loss"] and c["monotone_train_psnr"]
        assert [q["at"] for q in run["quality_trace"]].count("after opacity reset") == 2
        assert c["train_psnr_db_start_end"][1] > 30.0
    c = tl["well_posed_problem"]["convergence"]
    assert c["monotone_held_out_psnr"] and c["held_out_psnr_db_start_end"][1] - c["held_out_psnr_db_start_end"][0] > 7.0


def test_every_python_file_of_the_repo_compiles(tmp_path):
    """bench.py, the entry module, the package and the scripts byte-compile under this interpreter (a syntax error
    in a file the CPU suite never imports -- a script, a GPU-only branch -- would otherwise wait for the GPU box)"""
    import os
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for top in ("bench.py", "__graft_entry__.py", "gaussian_splatting_amd", "scripts", "tests", "oracle"):
        path = os.path.join(root, top)
        if os.path.isfile(path):
            files.append(path)
        for d, _, names in os.walk(path):
            files += [os.path.join(d, n) for n in names if n.endswith(".py")]
    assert len(files) > 40
    for f in files:
        py_compile.compile(f, doraise=True, cfile=str(tmp_path / "out.pyc"))


def test_background_weight_in_float_equals_the_reference_double_chain():
    """csrc/render.hip background_weight<float>: `1.0 - (alpha * weight + 1.0 - weight)` narrowed to float
    (render_backward.cu:175, the double literals promote everything behind the float product) equals the float
    subtraction `weight - alpha * weight` bit for bit wherever the kernel evaluates it: alpha in [1/255, 0.9999],
    weight = 1 - acc in [1e-4, 1] -- the three double operations are exact there."""
    import numpy as np
    rng = np.random.default_rng(7)

    def mismatches(a, w):
        a, w = a.astype(np.float32), w.astype(np.float32)
        x = (a * w).astype(np.float32)
        ref = (1.0 - ((x.astype(np.float64) + 1.0) - w.astype(np.float64))).astype(np.float32)
        new = (w - x).astype(np.float32)
        return int((ref.view(np.uint32) != new.view(np.uint32)).sum())

    n = 2_000_000
    assert mismatches(rng.uniform(1 / 255, 0.9999, n), rng.uniform(1e-4, 1, n)) == 0
    assert mismatches(np.exp(rng.uniform(np.log(1 / 255), np.log(0.9999), n)), np.exp(rng.uniform(np.log(1e-4), 0, n))) == 0
    # weight as the forward forms it (1 - acc, acc up to 0.9999) and mantissas at the ends of their range
    assert mismatches(rng.uniform(1 / 255, 0.9999, n), 1 - rng.uniform(0, 0.9999, n).astype(np.float32)) == 0
    m = np.concatenate([np.arange(0, 512), np.arange(2 ** 23 - 512, 2 ** 23)]).astype(np.uint32)
    for ea in range(119, 127):
        for ew in range(113, 128):
            A = ((np.uint32(ea) << 23) | m).view(np.float32)
            Wt = ((np.uint32(ew) << 23) | m).view(np.float32)
            AA, WW = [t.ravel() for t in np.meshgrid(A, Wt)]
            ok = (AA >= np.float32(1 / 255)) & (AA <= np.float32(0.9999)) & (WW <= 1) & (WW >= np.float32(1e-4))
            assert mismatches(AA[ok], WW[ok]) == 0, (ea, ew)
