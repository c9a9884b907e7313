"""debug: where does the depth-segmented backward differ from the unsegmented one? (GPU)"""
import json, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussian_splatting_amd import _hip, fused
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_grad_image, make_scene

DEV = "cuda"
N, W, H, deg = WORKLOADS["D"]
rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (26, 28)
prefix = _hip.GS_SORT_PREFIX if (len(sys.argv) > 3 and sys.argv[3] == "prefix") else 0
g, cam, T = make_scene(N, W, H, deg, seed=0, device=DEV)
d = DEFAULTS
f = fused.preprocess_forward(g.xyz, g.quaternion, g.scale, g.opacity, g.rgb, g.sh, T, cam.K, W, H, d["near_thresh"],
                             d["far_thresh"], d["cull_mask_padding"], d["mh_dist"], None, prefix)
V = f.V
rgb_v = f.rgb_render[:V]
bg = torch.full((3,), 0.5, device=DEV)
gi = make_grad_image(W, H, seed=1, device=DEV)
ntx = (W + 15) // 16


def run(seg_on, grad):
    image, nsp, fw, cost, seg = fused.render_forward(f.packed, rgb_v, f.ranges, f.sorted_g, f.keys, bg, H, W, rows, prefix,
                                                     segments=seg_on)
    slab = fused.render_backward(f.packed, rgb_v, f.ranges, f.sorted_g, bg, nsp, fw, grad, H, W, rows, V, None, None, seg)
    return slab.clone(), nsp.clone(), fw.clone(), seg


plain, nsp, fw, _ = run(False, gi)
segd, _, _, seg = run(True, gi)
err = (segd - plain).abs()
scale = plain.abs().max(0).values
print("scaled err per column", (err.max(0).values / scale).tolist())
rel = err / scale
worst = int(rel.max(1).values.argmax())
print("worst gaussian", worst, "plain", plain[worst].tolist(), "seg", segd[worst].tolist())
# tiles that hold it
ranges = f.ranges.cpu()
sorted_g = f.sorted_g.cpu()
tiles = []
for t in range(rows[0] * ntx, rows[1] * ntx):
    s0, s1 = int(ranges[t]), int(ranges[t + 1])
    pos = (sorted_g[s0:s1] == worst).nonzero()
    if pos.numel():
        tiles.append((t, int(pos[0]), s1 - s0))
print("tiles (tile, position in list, list length)", tiles)
P = W * H
T_all = ntx * ((H + 15) // 16)
segc = seg.cpu()
rec = segc[:T_all * 8 * 256 * 4].view(T_all, 8, 256, 4)
kend = segc[T_all * 8 * 256 * 4:T_all * 8 * 256 * 4 + P].view(torch.int32).view(H, W)
oma = segc[T_all * 8 * 256 * 4 + P:T_all * 8 * 256 * 4 + 2 * P].view(H, W)
bgw = segc[T_all * 8 * 256 * 4 + 2 * P:T_all * 8 * 256 * 4 + 3 * P].view(H, W)
for (t, pos, n) in tiles:
    ty, tx = t // ntx, t % ntx
    m = torch.zeros(H, W, 3, device=DEV)
    m[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16] = 1
    a, _, _, _ = run(False, gi * m)
    b, _, _, _ = run(True, gi * m)
    e = ((b[worst] - a[worst]).abs() / scale).max().item()
    print("tile", t, "pos", pos, "len", n, "err of the gaussian from this tile", e)
    if e < 1e-6:
        continue
    # per pixel
    worst_px, worst_e = None, 0
    for py in range(16):
        for px in range(16):
            y, x = ty * 16 + py, tx * 16 + px
            if y >= H or x >= W:
                continue
            m = torch.zeros(H, W, 3, device=DEV)
            m[y, x] = 1
            a, _, _, _ = run(False, gi * m)
            b, _, _, _ = run(True, gi * m)
            e2 = ((b[worst] - a[worst]).abs() / scale).max().item()
            if e2 > worst_e:
                worst_px, worst_e = (y, x), e2
    y, x = worst_px
    print(" worst pixel", worst_px, "err", worst_e, "nsp", int(nsp[y, x]), "kend", int(kend[y, x]), "fw", float(fw[y, x]),
          "oma_last", float(oma[y, x]), "bgw", float(bgw[y, x]))
    # thread index of the pixel inside the tile's workgroup
    py, px = y - ty * 16, x - tx * 16
    wv = (py // 8) * 2 + (px // 8)
    tid = wv * 64 + (py % 8) * 8 + (px % 8)
    print(" records (P, E0, E1, E2) per segment:", rec[t, :, tid].tolist())
    # the pixel's contributors, recomputed on the CPU in double
    s0 = int(ranges[t])
    pk = f.packed[:V].cpu().double()
    out = []
    acc = 0.0
    for k in range(min(n, int(nsp[y, x]) + 3)):
        gidx = int(sorted_g[s0 + k])
        r = pk[gidx]
        du, dv = x - r[0].item(), y - r[1].item()
        mh = (r[6].item() * du * du - 2 * r[5].item() * du * dv + r[4].item() * dv * dv) / r[7].item()
        al = r[3].item() * (2.718281828459045 ** (-0.5 * mh)) if mh > 0 else 0.0
        if al >= 0.00392156862 and du * du + dv * dv <= r[2].item():
            out.append((k, round(al, 6), round(1 - acc, 6)))
            acc += al * (1 - acc)
        if acc > 0.9999:
            break
    print(" contributors (k, alpha, T before):", out[-12:], "count", len(out))
    # per contributor: ratio of the colour gradient (alpha weight Y0 grad_image) segmented / plain, one pixel lit
    m = torch.zeros(H, W, 3, device=DEV)
    m[y, x] = 1
    a, _, _, _ = run(False, gi * m)
    b, _, _, _ = run(True, gi * m)
    a, b = a.cpu().double(), b.cpu().double()
    rows_out = []
    for (k, al, tb) in out:
        gidx = int(sorted_g[s0 + k])
        j = int(a[gidx, :3].abs().argmax())
        rows_out.append((k, round(float(b[gidx, j] / a[gidx, j]) - 1.0, 8), round(float(b[gidx, 3] / a[gidx, 3]) - 1.0, 8)))
    print(" (k, colour ratio - 1, opacity-gradient ratio - 1):", rows_out)
    break
