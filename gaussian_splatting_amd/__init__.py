"""gaussian_splatting_amd -- MI355X-native differentiable 3D Gaussian splat rasterizer behind the
API of joeyan/gaussian_splatting (`splat_cuda` functions, six autograd Functions,
`splat_py.rasterize`).  See DESIGN.md and INTEGRATION.md."""
from . import backend  # noqa: F401

__all__ = ["backend"]
