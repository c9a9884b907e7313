"""The fused band frontend (ABI 8: gs_band_frontend, gs_band_gather_sum, gs_preprocess_backward_gathered) against the
three-call pipeline it replaces in a multi-GPU rank's frame (gs_band_project -> gs_halo_plan_masked ->
gs_preprocess_forward_list; gs_halo_gather_sum -> gs_preprocess_backward): every output bit for bit -- culling mask,
rank, uv, sigmoid(opacity), camera centre, the exchange plan, the send list, the band-compact rows, the gathered rows
and the owned slice's parameter gradients.  The old calls are themselves held to the single-GPU frame and the oracle's
band masks by tests/test_gpu_sharded.py."""
import ctypes

import pytest
import torch

from gaussian_splatting_amd import _hip
from gaussian_splatting_amd.sharded import band_of, owner_blocks
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda"


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def cf(x):
    return ctypes.c_float(float(x))


def ints(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def run_old(g, cam, T, W, H, G, me, rows, oblk):
    N = g.xyz.shape[0]
    d = DEFAULTS
    i32 = dict(dtype=torch.int32, device=DEV)
    f32 = dict(dtype=torch.float32, device=DEV)
    lib = _hip.lib()
    ws = torch.zeros(lib.gs_preprocess_workspace_ints(N), **i32)
    hws = torch.zeros(lib.gs_halo_workspace_ints(N, G), **i32)
    out = dict(center=torch.zeros(4, **f32), count=torch.zeros(1, **i32), culled=torch.zeros(N, dtype=torch.uint8, device=DEV),
               rank=torch.zeros(N, **i32), vis_idx=torch.zeros(N, **i32), mask=torch.zeros(N, **i32),
               uv=torch.zeros(N, 2, **f32), opa=torch.zeros(N, **f32), send=torch.zeros(N, **i32),
               plan=torch.zeros(4 + 2 * G, **i32), uv_l=torch.zeros(N, 2, **f32), xyz_l=torch.zeros(N, 3, **f32),
               conic_l=torch.zeros(N, 3, **f32), packed_l=torch.zeros(N, 12, **f32), ws=ws, hws=hws)
    s = _hip.current_stream()
    n_sh = 1 if g.sh is None else g.sh.shape[2] + 1
    _hip.call("gs_band_project", p(g.xyz), p(g.scale), p(g.opacity), p(T), p(cam.K), N, W, H, cf(d["near_thresh"]),
              cf(d["far_thresh"]), cf(d["cull_mask_padding"]), cf(d["mh_dist"]), ints(rows), G, p(ws), p(out["center"]),
              p(out["count"]), p(out["culled"]), p(out["rank"]), p(out["vis_idx"]), p(out["uv"]), p(out["opa"]),
              p(out["mask"]), p(hws), s)
    _hip.call("gs_halo_plan_masked", p(out["mask"]), N, p(out["count"]), p(ws), ints(oblk), G, me, p(hws), p(out["send"]),
              p(out["plan"]), None, s)
    _hip.call("gs_preprocess_forward_list", p(g.xyz), p(g.quaternion), p(g.scale), p(g.rgb), p(g.sh), n_sh, p(T), p(cam.K),
              p(out["center"]), p(out["send"]), p(out["plan"]), N, p(out["vis_idx"]), p(out["uv"]), p(out["opa"]),
              p(out["uv_l"]), p(out["xyz_l"]), p(out["conic_l"]), p(out["packed_l"]), s)
    return out


def run_new(g, cam, T, W, H, G, me, rows, oblk):
    N = g.xyz.shape[0]
    d = DEFAULTS
    i32 = dict(dtype=torch.int32, device=DEV)
    f32 = dict(dtype=torch.float32, device=DEV)
    lib = _hip.lib()
    ws = torch.zeros(lib.gs_band_frontend_workspace_ints(N, G), **i32)
    out = dict(center=torch.zeros(4, **f32), culled=torch.zeros(N, dtype=torch.uint8, device=DEV), rank=torch.zeros(N, **i32),
               uv=torch.zeros(N, 2, **f32), opa=torch.zeros(N, **f32), send=torch.zeros(N, **i32),
               list_g=torch.zeros(N, **i32), plan=torch.zeros(4 + 2 * G, **i32), uv_l=torch.zeros(N, 2, **f32),
               xyz_l=torch.zeros(N, 3, **f32), conic_l=torch.zeros(N, 3, **f32), packed_l=torch.zeros(N, 12, **f32), ws=ws,
               plan_host=torch.zeros(4 + 2 * G, dtype=torch.int32).pin_memory())
    n_sh = 1 if g.sh is None else g.sh.shape[2] + 1
    _hip.call("gs_band_frontend", p(g.xyz), p(g.quaternion), p(g.scale), p(g.opacity), p(g.rgb), p(g.sh), n_sh, p(T),
              p(cam.K), N, W, H, cf(d["near_thresh"]), cf(d["far_thresh"]), cf(d["cull_mask_padding"]), cf(d["mh_dist"]),
              ints(rows), ints(oblk), G, me, p(ws), p(out["center"]), p(out["culled"]), p(out["rank"]), p(out["uv"]),
              p(out["opa"]), p(out["send"]), p(out["list_g"]), p(out["uv_l"]), p(out["xyz_l"]), p(out["conic_l"]),
              p(out["packed_l"]), p(out["plan"]), p(out["plan_host"]), _hip.current_stream())
    return out


@pytest.mark.parametrize("shape,G,me", [((60_000, 640, 480, 3), 2, 0), ((60_000, 640, 480, 3), 3, 1),
                                         ((60_000, 640, 480, 3), 8, 7), ((40_001, 320, 240, 0), 4, 2),
                                         ((5_000, 320, 240, 1), 8, 0), ((300, 64, 48, 0), 8, 5), ((300, 64, 48, 3), 8, 0),
                                         ((700, 64, 48, 0), 8, 2), ("D", 8, 4), ("D", 2, 1)])
def test_fused_band_frontend_equals_the_three_call_pipeline(shape, G, me):
    N, W, H, deg = WORKLOADS[shape] if isinstance(shape, str) else shape
    g, cam, T = make_scene(N, W, H, deg, seed=3, device=DEV)
    nty = (H + 15) // 16
    rows = [band_of(nty, G, r)[0] for r in range(G)] + [nty]
    oblk = owner_blocks(N, G)
    old = run_old(g, cam, T, W, H, G, me, rows, oblk)
    new = run_new(g, cam, T, W, H, G, me, rows, oblk)
    torch.cuda.synchronize()
    plan = old["plan"].tolist()
    L, V = plan[0], plan[1]
    assert new["plan"].tolist() == plan and new["plan_host"].tolist() == plan
    assert V == int(old["count"]) and 0 <= L <= V   # (64x48: three tile rows, ranks 3.. of 8 own no rows; 300 Gaussians: most ranks own none)
    assert torch.equal(new["center"][:3], old["center"][:3])
    assert torch.equal(new["culled"], old["culled"]) and torch.equal(new["rank"], old["rank"])
    assert torch.equal(new["uv"][:V], old["uv"][:V]) and torch.equal(new["opa"][:V], old["opa"][:V])
    assert torch.equal(new["send"][:L], old["send"][:L])
    assert torch.equal(new["list_g"][:L].long(), old["vis_idx"].long()[old["send"][:L].long()])
    for k in ("uv_l", "xyz_l", "conic_l", "packed_l"):
        assert torch.equal(new[k][:L], old[k][:L]), k
    # the masks the workspace keeps per Gaussian are the old masks by visible index
    nb = (N + 255) // 256
    gm = new["ws"][2 * (G + 1) * (nb + 1):].view(torch.int16)[:N].to(torch.int32) & 0xffff
    vis = (gm >> 15) & 1
    assert torch.equal(vis.bool(), old["culled"] == 0)
    assert torch.equal((gm & 0xff)[vis.bool()], old["mask"][:V] & 0xff)

    # ---- the receive side: random rows from every sender, summed per owned Gaussian -------------------------------
    recv_counts = plan[4 + G:4 + 2 * G]
    v_lo, v_hi = plan[2], plan[3]
    gen = torch.Generator(device=DEV).manual_seed(7)
    recv = torch.randn(max(sum(recv_counts), 1), 9, generator=gen, device=DEV)
    offs = [sum(recv_counts[:s]) for s in range(G)]
    n_own = v_hi - v_lo
    want = torch.full((max(n_own, 1), 9), 7.0, device=DEV)
    got = torch.full((max(n_own, 1), 9), 9.0, device=DEV)
    _hip.call("gs_halo_gather_sum", p(old["mask"]), p(old["hws"]), N, G, me, v_lo, v_hi, p(recv), ints(offs), p(want),
              _hip.current_stream())
    _hip.call("gs_band_gather_sum", p(new["ws"]), N, G, me, ints(oblk), p(new["rank"]), v_lo, p(recv), ints(offs), p(got),
              _hip.current_stream())
    assert torch.equal(got[:n_own], want[:n_own])

    # ---- ... and consumed by the per-Gaussian backward of the owned slice without the [owned, 9] round trip ------------
    i0, i1 = min(N, 256 * oblk[me]), min(N, 256 * oblk[me + 1])
    n = i1 - i0
    n_sh = 1 if g.sh is None else g.sh.shape[2] + 1

    def grads():
        return [torch.full((n, w), 5.0, device=DEV) for w in (3, 4, 3, 1, 3)] + \
               [torch.full((n, 3, n_sh - 1), 5.0, device=DEV) if n_sh > 1 else None]

    a, b = grads(), grads()
    sl = lambda t, w: ctypes.c_void_p(t.data_ptr() + 4 * w * i0)
    if n > 0:
        _hip.call("gs_preprocess_backward", sl(g.xyz, 3), sl(g.quaternion, 4), sl(g.scale, 3), n_sh, p(T), p(cam.K),
                  p(old["center"]), ctypes.c_void_p(old["rank"].data_ptr() + 4 * i0), p(old["opa"]), p(want), v_lo, n,
                  *[p(t) for t in a], _hip.current_stream())
        _hip.call("gs_preprocess_backward_gathered", sl(g.xyz, 3), sl(g.quaternion, 4), sl(g.scale, 3), n_sh, p(T), p(cam.K),
                  p(new["center"]), ctypes.c_void_p(new["rank"].data_ptr() + 4 * i0), p(new["opa"]), p(new["ws"]), N, G, me,
                  ints(oblk), p(recv), ints(offs), n, *[p(t) for t in b], _hip.current_stream())
        torch.cuda.synchronize()
        for x, y in zip(a, b):
            if x is not None:
                assert torch.equal(x, y)
        if sum(recv_counts) > 0:
            assert any(bool((x != 0).any()) for x in b if x is not None)


def test_fused_band_frontend_on_random_shapes():
    """30 seeded random shapes (N 1 .. 30 000 incl. non-multiples of 256, image sizes with partial tiles, SH degree 0-3,
    G 1-8, every rank position incl. ranks without tile rows or without Gaussians): plan, send list and band-compact
    rows of the fused frontend equal the three-call pipeline's"""
    import random
    rnd = random.Random(20260601)
    for trial in range(30):
        N = rnd.choice([1, 37, 255, 256, 257, 1000, 4097, rnd.randrange(2, 30_000)])
        W, H = rnd.randrange(17, 400), rnd.randrange(17, 300)
        deg = rnd.randrange(0, 4)
        G = rnd.randrange(1, 9)
        me = rnd.randrange(0, G)
        g, cam, T = make_scene(N, W, H, deg, seed=100 + trial, device=DEV)
        nty = (H + 15) // 16
        rows = [band_of(nty, G, r)[0] for r in range(G)] + [nty]
        oblk = owner_blocks(N, G)
        old = run_old(g, cam, T, W, H, G, me, rows, oblk)
        new = run_new(g, cam, T, W, H, G, me, rows, oblk)
        torch.cuda.synchronize()
        plan = old["plan"].tolist()
        L, V = plan[0], plan[1]
        tag = (trial, N, W, H, deg, G, me)
        assert new["plan"].tolist() == plan, tag
        assert torch.equal(new["culled"], old["culled"]) and torch.equal(new["rank"], old["rank"]), tag
        assert torch.equal(new["uv"][:V], old["uv"][:V]) and torch.equal(new["opa"][:V], old["opa"][:V]), tag
        assert torch.equal(new["send"][:L], old["send"][:L]), tag
        for k in ("uv_l", "xyz_l", "conic_l", "packed_l"):
            assert torch.equal(new[k][:L], old[k][:L]), (tag, k)
