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

from gaussian_splatting_amd.sharded import ShardedRasterizer, balanced_bounds, band_of

PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def test_band_split_covers_all_rows():
    for nty in (1, 5, 53, 68):
        for world in (1, 2, 3, 4, 8):
            bands = [band_of(nty, world, r) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == nty
            for a, b in zip(bands, bands[1:]):
                assert a[1] == b[0]
            sizes = [b[1] - b[0] for b in bands]
            full = -(-nty // world)   # every band is full-sized until the rows run out
            assert all(x == min(full, max(0, nty - r * full)) for r, x in enumerate(sizes))


def test_balanced_bounds_minimise_the_largest_band():
    import itertools
    import random
    rnd = random.Random(3)
    for _ in range(200):
        R, G = rnd.randint(1, 9), rnd.randint(1, 4)
        cost = [rnd.choice([0, 1, 2, 5, 40]) for _ in range(R)]
        b = balanced_bounds(cost, G)
        assert len(b) == G + 1 and b[0] == 0 and b[-1] == R and all(x <= y for x, y in zip(b, b[1:]))
        got = max(sum(cost[b[r]:b[r + 1]]) for r in range(G))
        best = min(max(sum(cost[c[r]:c[r + 1]]) for r in range(G))
                   for cuts in itertools.combinations_with_replacement(range(R + 1), G - 1)
                   for c in [(0,) + cuts + (R,)])
        assert got == best, (cost, G, b)
    # a uniform image splits like the equal policy's largest band
    assert max(y - x for x, y in zip(balanced_bounds([7] * 53, 8), balanced_bounds([7] * 53, 8)[1:])) == 7


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


def _cost_worker(rank, world, port, out):
    """replicated gradients over cost-balanced bands (row costs given), plus the loss-contract check"""
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
        ref_img, ref_mask, ref_guv, ref_grads = _frame(rasterize, 0)
        sharded = ShardedRasterizer(100, world, rank, fused=False, band_policy="cost", check_grad_image=True)
        sharded.set_row_costs([50, 1, 1, 1, 1, 1, 30])   # 7 tile rows: a heavy first row
        bounds = list(sharded.bounds)
        img, mask, guv, grads = _frame(sharded.rasterize, 0)
        ok = torch.equal(img, ref_img) and torch.equal(mask, ref_mask)
        err = max(((grads[k] - ref_grads[k]).abs().max() / ref_grads[k].abs().max().clamp(min=1e-30)).item()
                  for k in grads)
        # a per-rank loss violates the contract: the check must raise on every rank
        raised = False
        try:
            from gaussian_splatting_amd.synthetic import make_grad_image, make_scene
            g, cam, T = make_scene(600, 112, 100, 0, seed=4)
            g.xyz.requires_grad_(True)
            img2, _, _ = sharded.rasterize(g, T, cam, 2.0, 25.0, 10, 3.0, True, torch.full((3,), 0.5))
            img2.backward(make_grad_image(112, 100, seed=8) * (rank + 1))
        except RuntimeError as e:
            raised = "grad_image differs" in str(e)
        out.put((rank, bool(ok), float(err), bounds, raised))
    finally:
        dist.destroy_process_group()


def test_cost_balanced_bands_match_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cost_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err, bounds, raised in results:
        assert bounds == [0, 1, 7], bounds
        assert ok, f"rank {rank}: image over cost-balanced bands differs from the single-process image"
        assert err < 1e-6, f"rank {rank}: gradient error {err}"
        assert raised, f"rank {rank}: a per-rank loss was not detected"


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


# ---- grad_mode "owner": sparse exchange of the partial render gradients --------------------------------
def test_owner_ranges_partition_the_gaussians():
    from gaussian_splatting_amd.sharded import OWNER_BLOCK, owner_blocks, owner_range
    for N in (1, 255, 256, 257, 600, 100000, 2860000):
        for world in (1, 2, 3, 8):
            ranges = [owner_range(N, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == N
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            assert all(lo % OWNER_BLOCK == 0 for lo, _ in ranges)
            assert len(owner_blocks(N, world)) == world + 1


def test_halo_plan_reference_is_a_consistent_exchange():
    """all ranks' plans (pure bookkeeping, no process group): what s sends to r is what r expects
    from s, and pack -> route -> unpack equals the dense sum restricted to the owned range"""
    from gaussian_splatting_amd.sharded import HaloPlan
    gen = torch.Generator().manual_seed(3)
    G, V = 4, 5000
    mask = torch.randint(0, 2 ** G, (V,), generator=gen, dtype=torch.int32)
    mask[::17] = 0
    v_bounds = [0, 1300, 1300, 4100, V]   # an empty owner range included
    plans = [HaloPlan.reference(mask, v_bounds, G, r) for r in range(G)]
    for s in range(G):
        for r in range(G):
            assert plans[s].send_splits[r] == plans[r].recv_splits[s]
    # slab of rank s: non-zero only where bit s is set (a band's partial sums)
    slabs = [torch.randn(V, 9, generator=gen) * ((mask >> s) & 1).unsqueeze(1) for s in range(G)]
    dense = sum(slabs)
    sends = [plans[s].pack(slabs[s]) for s in range(G)]
    for r in range(G):
        chunks = []
        for s in range(G):
            off = sum(plans[s].send_splits[:r])
            chunks.append(sends[s][off:off + plans[s].send_splits[r]])
        owned = plans[r].unpack(torch.cat(chunks))
        assert owned.shape[0] == v_bounds[r + 1] - v_bounds[r]
        assert torch.allclose(owned, dense[v_bounds[r]:v_bounds[r + 1]], atol=1e-6)


def _owner_worker(rank, world, port, deg, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_splatting_amd import backend
        from gaussian_splatting_amd.sharded import owned_slice, owner_range
        from gaussian_splatting_amd.splat_py.rasterize import rasterize
        from gaussian_splatting_amd.synthetic import make_grad_image, make_scene
        from oracle import gs_oracle
        gs_oracle.set_modes(0, 0)
        backend.use(gs_oracle)
        ref_img, ref_mask, _, ref_grads = _frame(rasterize, deg)
        W, H = 112, 100
        g, cam, T = make_scene(600, W, H, deg, seed=4)
        owned = owned_slice(g, world, rank)
        sharded = ShardedRasterizer(H, world, rank, fused=False, grad_mode="owner")
        img, mask, uv = sharded.rasterize(g, T, cam, 2.0, 25.0, 10, 3.0, True, torch.full((3,), 0.5), owned=owned)
        img.backward(make_grad_image(W, H, seed=8))
        i0, i1 = owner_range(600, world, rank)
        ok = torch.equal(img.detach(), ref_img) and torch.equal(mask, ref_mask)
        err = 0.0
        for k in PARAMS:
            if getattr(owned, k) is None:
                continue
            got, ref = getattr(owned, k).grad, ref_grads[k][i0:i1]
            assert got.shape == ref.shape
            err = max(err, ((got - ref).abs().max() / ref_grads[k].abs().max().clamp(min=1e-30)).item())
        plan = sharded.last_plan
        sparse = plan is not None and sum(plan.send_splits) < int(mask.numel() - mask.sum())
        out.put((rank, bool(ok), float(err), bool(sparse), (i0, i1)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,deg", [(2, 0), (3, 3)])
def test_owner_mode_matches_single_process(world, deg):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, deg, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    spans = sorted(r[4] for r in results)
    assert spans[0][0] == 0 and spans[-1][1] == 600
    for rank, ok, err, sparse, _ in results:
        assert ok, f"rank {rank}: sharded image differs from the single-process image"
        assert err < 1e-6, f"rank {rank}: owned-slice gradient error {err}"
        assert sparse, f"rank {rank}: the exchange was not sparse"
