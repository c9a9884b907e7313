import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from gaussian_splatting_amd import fused
from gaussian_splatting_amd.synthetic import DEFAULTS, make_grad_image, make_scene
N, W, H, deg = 12_000_000, 1297, 840, 3
g, cam, T = make_scene(N, W, H, deg, seed=3, device="cuda")
params = [p for p in (g.xyz, g.rgb, g.opacity, g.scale, g.quaternion, g.sh) if p is not None]
gi = make_grad_image(W, H, seed=1, device="cuda"); bg = torch.zeros(3, device="cuda")
outs = []
for aux in (False, True):
    for p in params: p.requires_grad_(True); p.grad = None
    r = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, return_aux=aux, **DEFAULTS)
    img = r[0]
    img.backward(gi); torch.cuda.synchronize()
    outs.append((img.detach().clone(), [p.grad.clone() for p in params]))
print("counters", fused.counters())
print("image equal (prefix vs full sort):", torch.equal(outs[0][0], outs[1][0]), "finite:", bool(torch.isfinite(outs[0][0]).all()))
for a, b, name in zip(outs[0][1], outs[1][1], ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")):
    print(name, "finite", bool(torch.isfinite(a).all()), "scaled diff", float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)))
for _ in range(5):
    for p in params: p.grad = None
    img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS); img.backward(gi)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    for p in params: p.grad = None
    img, _, _ = fused.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS); img.backward(gi)
torch.cuda.synchronize(); print("ms/frame at N = 12 M:", (time.perf_counter() - t) * 100)
