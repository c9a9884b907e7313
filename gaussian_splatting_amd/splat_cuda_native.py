"""Loader of the native `splat_cuda` module (csrc/bindings_hip.cpp -> native/splat_cuda.<abi>.so).

The compiled counterpart of gaussian_splatting_amd.splat_cuda (the ctypes shim): same 14 functions,
but a pybind11 module on torch::Tensor exactly like the reference's extension (src/bindings.cpp), so
the per-call cost is the reference's own (one C++ call, no ctypes marshalling).

    from gaussian_splatting_amd import splat_cuda_native
    splat_cuda_native.install()      # sys.modules["splat_cuda"] = the native module
    import splat_py                  # the reference's package now runs on MI355X unchanged

Build: `make -C gaussian_splatting_amd/csrc native` (or __graft_entry__.build()).  No fallback: a
missing module raises.
"""
import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_module = None


def path():
    hits = sorted(glob.glob(os.path.join(_HERE, "native", "splat_cuda*.so")))
    if not hits:
        raise ImportError("gaussian_splatting_amd/native/splat_cuda*.so not found: run "
                          "`make -C gaussian_splatting_amd/csrc native` (there is no fallback)")
    return hits[0]


def load():
    """-> the native module (imports torch first: the extension links against its libraries)"""
    global _module
    if _module is None:
        import torch  # noqa: F401
        spec = importlib.util.spec_from_file_location("splat_cuda", path())
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _module = mod
    return _module


_frame = None


def load_frame():
    """-> the native frame-orchestration module `gsplat_frame` (csrc/frame_hip.cpp)"""
    global _frame
    if _frame is None:
        import torch  # noqa: F401
        hits = sorted(glob.glob(os.path.join(_HERE, "native", "gsplat_frame*.so")))
        if not hits:
            raise ImportError("gaussian_splatting_amd/native/gsplat_frame*.so not found: run "
                              "`make -C gaussian_splatting_amd/csrc native`")
        spec = importlib.util.spec_from_file_location("gsplat_frame", hits[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _frame = mod
    return _frame


def install(name="splat_cuda"):
    sys.modules[name] = load()
    return sys.modules[name]
