"""How deep does the render go?  Per tile: list length vs the largest num_splats_per_pixel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import _hip, fused, splat_cuda
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene
import ctypes
for wl in sys.argv[1:] or ["D"]:
    N, W, H, deg = WORKLOADS[wl]
    g, cam, T = make_scene(N, W, H, deg, seed=0, device="cuda")
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        img, mask, uv, aux = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, return_aux=True, **DEFAULTS)
        # re-run the render through the drop-in entry to get num_splats_per_pixel
        nsp = torch.zeros(H, W, dtype=torch.int32, device="cuda"); fw = torch.zeros(H, W, device="cuda"); im = torch.zeros(H, W, 3, device="cuda")
        splat_cuda.render_tiles_cuda(uv, aux["opacity"], aux["rgb"], aux["conic"], torch.zeros(1,1,1,device="cuda"), aux["tile_ranges"], aux["sorted_gaussians"], bg, nsp, fw, im)
    ntx, nty = (W + 15)//16, (H + 15)//16
    pad = torch.zeros(nty*16, ntx*16, dtype=torch.int32, device="cuda"); pad[:H, :W] = nsp
    tmax = pad.view(nty, 16, ntx, 16).permute(0, 2, 1, 3).reshape(nty*ntx, 256).max(dim=1).values.float()
    r = aux["tile_ranges"].long(); n = (r[1:] - r[:-1]).float()
    q = torch.tensor([0.5, 0.9, 0.99, 1.0], device="cuda")
    print(wl, "list len mean/max", n.mean().item(), n.max().item(), "| tile-max nsp quantiles", torch.quantile(tmax, q).tolist(),
          "| frac tiles with max nsp <= 1024:", (tmax <= 1024).float().mean().item(), "<=768:", (tmax <= 768).float().mean().item(),
          "<=512:", (tmax <= 512).float().mean().item(), "| mean used/len", (tmax.sum()/n.sum()).item())
