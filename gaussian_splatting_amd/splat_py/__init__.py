"""Host-side mirror of the reference's `splat_py` package for the rasterization hot path:
same module names, function names, argument order and return values (splat_py/rasterize.py,
cuda_autograd_functions.py, tile_culling.py, depth.py, structs.py, utils.py), so a trainer
written against the reference imports `gaussian_splatting_amd.splat_py` instead."""
