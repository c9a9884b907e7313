"""The C-ABI library loads and exports every symbol include/gsplat_hip.h declares (no compute
calls: runs without a GPU)."""
import ctypes
import os
import re

from gaussian_splatting_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_symbols()
    # one entry point per function of src/bindings.cpp:118-159 (get_sorted_gaussian_list is split
    # in two; the two render entries take the reference's own argument lists, their _packed forms the
    # record of gs_pack_splats)
    for n in ["gs_camera_projection", "gs_camera_projection_backward", "gs_compute_sigma_world",
              "gs_compute_sigma_world_backward", "gs_compute_projection_jacobian",
              "gs_compute_projection_jacobian_backward", "gs_compute_conic", "gs_compute_conic_backward",
              "gs_precompute_rgb_from_sh", "gs_precompute_rgb_from_sh_backward", "gs_tile_count", "gs_tile_workspace_ints",
              "gs_preprocess_forward", "gs_preprocess_backward",
              "gs_tile_emit_sort", "gs_pack_splats", "gs_render_tiles", "gs_render_tiles_backward",
              "gs_render_tiles_packed", "gs_render_tiles_backward_packed",
              "gs_render_depth", "gs_last_error", "gs_abi_version"]:
        assert n in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_hip.LIB_PATH), "build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.gs_abi_version() >= 1


def test_python_loader_lists_the_same_symbols():
    assert sorted(_hip.EXPORTS) == declared_symbols()


def test_native_splat_cuda_module_exports_the_reference_names():
    """csrc/bindings_hip.cpp: the compiled `splat_cuda` module has the 14 functions of
    src/bindings.cpp:118-159 and raises RuntimeError (TORCH_CHECK) for a non-device tensor"""
    import pytest
    import torch

    from gaussian_splatting_amd import splat_cuda, splat_cuda_native
    mod = splat_cuda_native.load()
    names = ["render_tiles_cuda", "render_tiles_backward_cuda", "camera_projection_cuda",
             "camera_projection_backward_cuda", "compute_sigma_world_cuda", "compute_sigma_world_backward_cuda",
             "compute_projection_jacobian_cuda", "compute_projection_jacobian_backward_cuda", "compute_conic_cuda",
             "compute_conic_backward_cuda", "get_sorted_gaussian_list", "precompute_rgb_from_sh_cuda",
             "precompute_rgb_from_sh_backward_cuda", "render_depth_cuda"]
    for n in names:
        assert callable(getattr(mod, n)) and callable(getattr(splat_cuda, n)), n
    with pytest.raises(RuntimeError, match="xyz is not a CUDA tensor"):
        mod.camera_projection_cuda(torch.zeros(3, 3), torch.zeros(3, 3), torch.zeros(3, 2))
    splat_cuda_native.install("splat_cuda_test_alias")
    import sys
    assert sys.modules.pop("splat_cuda_test_alias") is mod


def test_library_shares_the_hip_runtime_torch_loaded():
    """loading the library before `import torch` must not bring a second libamdhip64 into the process
    (the second runtime to initialise finds no device: __graft_entry__.build() followed by smoke())"""
    import subprocess
    import sys
    code = ("from gaussian_splatting_amd import _hip\n_hip.lib()\nimport torch\n"
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "1", out.stdout
