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


def _prototypes():
    """{function name: number of parameters} of include/gsplat_hip.h"""
    src = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(gs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(params.split(","))
    return out


def test_python_call_sites_pass_as_many_arguments_as_the_header_declares():
    """ctypes calls carry no prototype: a call site that drifts from include/gsplat_hip.h would corrupt the
    arguments silently.  Every `_hip.call("gs_...", ...)` / `lib.gs_...(...)` in the package, bench.py and the
    scripts must pass exactly the declared number of arguments."""
    import ast
    protos = _prototypes()
    assert len(protos) == len(declared_symbols()) and protos["gs_abi_version"] == 0 and protos["gs_camera_projection"] == 6
    files = [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    for d in ("gaussian_splatting_amd", "scripts", "tests"):
        files += [os.path.join(ROOT, d, f) for f in sorted(os.listdir(os.path.join(ROOT, d))) if f.endswith(".py")]
    checked, bad = 0, []
    for path in files:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            name, n_args = None, None
            f = node.func
            if isinstance(f, ast.Attribute) and f.attr == "call" and node.args and isinstance(node.args[0], ast.Constant) \
                    and isinstance(node.args[0].value, str) and node.args[0].value.startswith("gs_"):
                name, n_args = node.args[0].value, len(node.args) - 1          # _hip.call("gs_x", a, b, ...)
            elif isinstance(f, ast.Attribute) and f.attr.startswith("gs_") and f.attr in protos:
                name, n_args = f.attr, len(node.args)                            # lib.gs_x(a, b, ...)
            if name is None or any(isinstance(a, ast.Starred) for a in node.args):
                continue
            checked += 1
            if name not in protos or protos[name] != n_args:
                bad.append((os.path.relpath(path, ROOT), node.lineno, name, n_args, protos.get(name)))
    assert checked >= 40, checked
    assert not bad, bad


def test_product_sources_carry_no_experiment_builds():
    """timing-only emulations and A/B switches live in scripts/experiments/*.patch, not in the shipping translation
    units (round-5 review): no GS_EMU_* token and none of the retired A/B macros in csrc/ or include/"""
    banned = re.compile(r"GS_EMU_|GS_CK_NOREC|GS_CK_NOEPI|GS_XCD_SEG|GS_ARITH_FAST|GS_BWD_RCP_REFINE|"
                        r"GS_PRIV_XCD|GS_EMIT_NOSTORE|GS_EMIT_CELL|GS_CUT_EMIT_PREFETCH|GS_CUT_GENERAL_SORT")
    hits = []
    for d in (os.path.join(ROOT, "gaussian_splatting_amd", "csrc"), os.path.join(ROOT, "include")):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
                for i, line in enumerate(open(os.path.join(d, name), errors="replace"), 1):
                    if banned.search(line):
                        hits.append(f"{name}:{i}")
    assert not hits, hits
