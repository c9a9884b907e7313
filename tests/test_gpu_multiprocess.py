"""The sharded frame with MORE THAN ONE RANK on a one-GPU box: `world` processes share the device and exchange
through gloo (c10d's CUDA-tensor collectives of that backend), so the real all-gather of the band images and the
real all_to_all of the partial gradients run -- through the native C++ orchestration (the same c10d calls the RCCL
job makes) and through the Python one -- and every rank checks its frame against the single-GPU frame it computes
itself.  RCCL itself refuses two ranks on one device; its world-size-1 run is tests/test_gpu_fused.py's."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
PARAMS = ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, native, deg, out, band_policy="equal"):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_splatting_amd import fused, sharded
        from gaussian_splatting_amd.synthetic import DEFAULTS, make_grad_image, make_scene
        W, H = 400, 300   # 25 x 19 tiles: bands of unequal height at world 2 and 3, partial last row
        g, cam, T = make_scene(30000, W, H, deg, seed=5, device="cuda")
        gi = make_grad_image(W, H, seed=6, device="cuda")
        bg = torch.full((3,), 0.25, device="cuda")
        # single-GPU frame on this very process
        ref_g = type(g)(*[None if getattr(g, k) is None else getattr(g, k).detach().clone().requires_grad_(True)
                          for k in ("xyz", "rgb", "opacity", "scale", "quaternion", "sh")])
        ref_img, ref_mask, ref_uv = fused.rasterize(ref_g, T, cam, use_sh_precompute=True, background_rgb=bg, **DEFAULTS)
        ref_img.backward(gi)
        rast = sharded.ShardedRasterizer(H, grad_mode=mode, native=native, band_policy=band_policy)
        assert rast.world_size == world and rast.rank == rank
        if mode == "owner":
            holder = sharded.owned_slice(g, world, rank)
            i0, i1 = rast.owned_range(g.xyz.shape[0])
        else:
            for k in PARAMS:
                if getattr(g, k) is not None:
                    getattr(g, k).requires_grad_(True)
            holder, (i0, i1) = g, (0, g.xyz.shape[0])
        worst = 0.0
        same = True
        for frame in range(3):   # later frames take the speculative-capacity path (and, cost policy, re-cut bands)
            for k in PARAMS:
                if getattr(holder, k) is not None:
                    getattr(holder, k).grad = None
            image, mask, uv = rast.rasterize(g, T, cam, use_sh_precompute=True, background_rgb=bg,
                                             owned=holder if mode == "owner" else None, **DEFAULTS)
            image.backward(gi)
            same = same and bool(torch.equal(image.detach(), ref_img.detach())) and bool(torch.equal(mask, ref_mask))
            for k in PARAMS:
                if getattr(holder, k) is None:
                    continue
                ref = getattr(ref_g, k).grad
                err = (getattr(holder, k).grad.double() - ref[i0:i1].double()).abs().max() / ref.abs().max().double()
                worst = max(worst, float(err))
        plan = rast.last_plan
        native_used = type(plan).__name__ == "SimpleNamespace"   # the native branch records its plan as a namespace
        out.put((rank, same, worst, None if plan is None else (sum(plan.send_splits), sum(plan.recv_splits), native_used),
                 tuple(rast.tile_rows)))
    except Exception as e:   # noqa: BLE001 -- reported to the parent, which fails the test
        import traceback
        out.put((rank, False, float("inf"), "".join(traceback.format_exception(type(e), e, e.__traceback__))[-2000:], None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,native,policy", [
    (2, "owner", True, "equal"), (3, "owner", True, "equal"), (8, "owner", True, "equal"),
    (2, "owner", False, "equal"), (2, "replicated", None, "equal"),
    # cost-balanced bands: the per-row costs ride on the image gather and the next frame's bands follow them
    (3, "owner", False, "cost"), (3, "owner", True, "cost"), (2, "owner", True, "cost")])
def test_sharded_frame_on_several_ranks_of_one_gpu(world, mode, native, policy):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, native, 3, out, policy)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in range(world):
            results.append(out.get(timeout=240))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    results.sort()
    for rank, same, worst, info, rows in results:
        assert same, f"rank {rank}: image or mask differs from the single-GPU frame ({info})"
        assert worst < 2e-5, f"rank {rank}: gradient error {worst} ({info})"
    if mode == "owner":
        # what the ranks send is what the ranks receive
        assert sum(r[3][0] for r in results) == sum(r[3][1] for r in results) > 0
        assert all(r[3][2] == bool(native) for r in results), "the orchestration asked for is not the one that ran"
    # the ranks' bands tile the rows in rank order
    rows = [r[4] for r in results]
    assert rows[0][0] == 0 and rows[-1][1] == 19 and all(a[1] == b[0] for a, b in zip(rows, rows[1:])), rows


@pytest.mark.parametrize("inject", [None, "owner"])
def test_bench_line_with_two_ranks_on_one_gpu(inject):
    """bench.py's N > 1 code (the command the driver launches for the scaling record) on a one-GPU box: two ranks
    under torch.distributed.run, gloo instead of RCCL.  The times mean nothing; the line must be complete, carry
    the in-run check of the sharded frame against the single-GPU frame, and be the last line on stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GS_BENCH_BACKEND="gloo")
    if inject:   # the native orchestration "fails" on every rank: the same frame with the Python orchestration is timed
        env["GS_BENCH_INJECT_FAILURE"] = inject
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "2", "--workload", "B", "--spinup-steps", "2"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 2 and line["scaling"] == "strong"
    assert line["metric"] and line["value"] > 0 and line["config"]["parallelism"].startswith("tile-rows x2")
    m = line["multi_gpu"]
    assert m["world_size_seen_by_rccl"] == 2 and m["backend"] == "gloo"
    if inject:
        assert m["headline_grad_mode"] == "owner_python" and "injected failure" in m["modes"]["owner_error"]
        assert m["sharded_check"]["mode"] == "owner_python"
    else:
        assert m["headline_grad_mode"] == "owner", m["modes"]   # the native orchestration ran: no fallback was needed
    assert m["sharded_check"]["image_equals_single_gpu_image_on_every_rank"] is True
    assert m["sharded_check"]["grad_max_err_over_tensor_scale"] < 2e-5
    assert len(m["ranks"]) == 2 and set(m["modes"]) >= {"owner_python" if inject else "owner", "replicated"}
    assert "rank 0/2" in r.stderr and "rank 1/2" in r.stderr
