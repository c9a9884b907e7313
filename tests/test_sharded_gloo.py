"""Multi-process test of the tile-row sharded rasterizer on CPU (gloo, world_size 2 and 3).

The collective logic (band split, image gather, all-reduce of the per-Gaussian render gradients)
is backend-agnostic; here the compute provider is the CPU oracle (test-only injection, the product
path uses the HIP kernels) through the reference-shaped six-node pipeline."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaussian_splatting_amd.sharded import ShardedRasterizer, band_of

PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def test_band_split_covers_all_rows():
    for nty in (1, 5, 53, 68):
        for world in (1, 2, 3, 4, 8):
            bands = [band_of(nty, world, r) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == nty
            for a, b in zip(bands, bands[1:]):
                assert a[1] == b[0]
            sizes = [b[1] - b[0] for b in bands]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(rasterize_fn, deg, requires_grad=True):
    from gaussian_splatting_amd.synthetic import make_grad_image, make_scene
    W, H = 112, 100   # 7 x 7 tiles, partial last row/column
    g, cam, T = make_scene(600, W, H, deg, seed=4)
    for k in PARAMS:
        if getattr(g, k) is not None:
            getattr(g, k).requires_grad_(requires_grad)
    img, mask, uv = rasterize_fn(g, T, cam, 2.0, 25.0, 10, 3.0, True, torch.full((3,), 0.5))
    uv.retain_grad()
    img.backward(make_grad_image(W, H, seed=8))
    grads = {k: getattr(g, k).grad.clone() for k in PARAMS if getattr(g, k) is not None}
    return img.detach(), mask, uv.grad.clone(), grads


def _worker(rank, world, port, deg, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_splatting_amd import backend
        from gaussian_splatting_amd.splat_py.rasterize import rasterize
        from oracle import gs_oracle
        gs_oracle.set_modes(0, 0)
        backend.use(gs_oracle)
        ref_img, ref_mask, ref_guv, ref_grads = _frame(rasterize, deg)
        sharded = ShardedRasterizer(100, world, rank, fused=False)
        img, mask, guv, grads = _frame(sharded.rasterize, deg)
        ok = torch.equal(img, ref_img) and torch.equal(mask, ref_mask)
        err = max(((grads[k] - ref_grads[k]).abs().max() / ref_grads[k].abs().max().clamp(min=1e-30)).item()
                  for k in grads)
        err = max(err, ((guv - ref_guv).abs().max() / ref_guv.abs().max()).item())
        # every rank must end with the same gradients (replicated parameters stay in sync)
        flat = torch.cat([grads[k].reshape(-1) for k in sorted(grads)])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same = all(torch.equal(gathered[0], x) for x in gathered)
        out.put((rank, bool(ok), float(err), bool(same), sharded.tile_rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,deg", [(2, 0), (2, 3), (3, 0)])
def test_sharded_matches_single_process(world, deg):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, deg, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = sorted(r[4] for r in results)
    assert rows[0][0] == 0 and rows[-1][1] == 7
    for rank, ok, err, same, _ in results:
        assert ok, f"rank {rank}: sharded image differs from the single-process image"
        assert err < 1e-6, f"rank {rank}: gradient error {err}"
        assert same, f"rank {rank}: gradients differ between ranks"
