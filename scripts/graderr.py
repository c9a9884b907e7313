import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_parity import cpu_stage_inputs, render_case, oracle
from gaussian_splatting_amd import splat_cuda
from gaussian_splatting_amd.synthetic import make_grad_image
orc = oracle()
for (N, W, H, seed) in [(20000, 640, 472, 21), (9000, 96, 80, 22)]:
    d = cpu_stage_inputs(N, W, H, 0, seed)
    ntx, nty = (W + 15) // 16, (H + 15) // 16
    sg, rg = orc.get_sorted_gaussian_list(1024, d["uv"], d["xyz_c"], d["conic"], ntx, nty, 3.0)
    bg = torch.full((3,), 0.5); gi = make_grad_image(W, H, seed=seed + 1)
    ref = render_case(orc, "cpu", d, d["rgb"], torch.zeros(1,1,1), bg, sg, rg, torch.float32, gi)
    got = render_case(splat_cuda, "cuda", d, d["rgb"], torch.zeros(1,1,1), bg, sg, rg, torch.float32, gi)
    for k in ("g_rgb", "g_opacity", "g_uv", "g_conic"):
        g, r = got[k].double(), ref[k].double()
        mx = r.abs().max()
        line = f"{N} {k:10s} scaled={((g-r).abs().max()/mx).item():.2e}"
        for fl in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2):
            e = ((g-r).abs()/torch.clamp(r.abs(), min=fl*mx)).max().item()
            line += f"  floor{fl:g}={e:.2e}"
        print(line)
