"""Selects the module that provides the 14 `splat_cuda` functions (src/bindings.cpp:118-159).

Default and only product backend: gaussian_splatting_amd.splat_cuda (HIP kernels through the
C ABI).  Tests may inject another provider with `use()` -- e.g. the CPU oracle, to exercise the
host-side mirror of the reference's Python code without a GPU.  Nothing in this package ever
selects a CPU provider by itself.
"""
_backend = None


def use(module):
    global _backend
    _backend = module


def get():
    global _backend
    if _backend is None:
        from . import splat_cuda as hip_backend
        _backend = hip_backend
    return _backend
