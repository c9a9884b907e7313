"""Training-loop operations right behind the rasterizer, over the C ABI.

    Adam                      drop-in for the torch.optim.Adam the reference builds in
                              splat_py/optimizer_manager.py:15-42 and steps in trainer.py:376: same
                              constructor, param_groups and state layout ("step", "exp_avg",
                              "exp_avg_sq"), so OptimizerManager's parameter surgery works on it
                              unchanged; step() is ONE HIP launch over all parameter tensors
    accumulate_grad_stats     trainer.py:378-385 (densification statistics) in one launch, without
                              the boolean-mask index_put and its host sync

Anything the HIP kernel does not cover (amsgrad, weight decay, maximize, capturable, sparse or
non-fp32 / non-device tensors) is handed to torch.optim.Adam.step itself.
"""
import ctypes

import torch

from . import _hip

_MAX_TENSORS = 8   # per launch (csrc/train_ops.hip)


class Adam(torch.optim.Adam):
    def _hip_ok(self, group):
        if group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"] or group.get("capturable", False):
            return False
        if group.get("differentiable", False):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse
                    and p.grad.dtype == torch.float32):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._hip_ok(g) for g in self.param_groups):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:   # torch/optim/adam.py _init_group
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                work.append((p, grad, state["exp_avg"], state["exp_avg_sq"], float(group["lr"]),
                             int(state["step"]), beta1, beta2, group["eps"]))
        stream = _hip.current_stream()
        i = 0
        while i < len(work):
            # one launch per run of tensors that share (beta1, beta2, eps): all of them, normally
            j = i
            while j < len(work) and j - i < _MAX_TENSORS and work[j][6:] == work[i][6:]:
                j += 1
            chunk = work[i:j]
            n = len(chunk)
            ptrs = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in chunk])
            _hip.call("gs_adam_step", n, ptrs(0), ptrs(1), ptrs(2), ptrs(3),
                      (ctypes.c_int64 * n)(*[t[0].numel() for t in chunk]),
                      (ctypes.c_double * n)(*[t[4] for t in chunk]),
                      (ctypes.c_int64 * n)(*[t[5] for t in chunk]),
                      ctypes.c_double(chunk[0][6]), ctypes.c_double(chunk[0][7]), ctypes.c_double(chunk[0][8]),
                      stream)
            i = j
        # the kernel wrote through raw pointers: bump the version counters like an in-place torch op would,
        # so that autograd's saved-tensor check still catches a backward over stale parameters
        for p, _, m, v, *_ in work:
            torch.autograd.graph.increment_version(p)
            torch.autograd.graph.increment_version(m)
            torch.autograd.graph.increment_version(v)
        return loss


def accumulate_grad_stats(uv_grad, culling_mask, xyz_grad, camera, uv_grad_accum, xyz_grad_accum, grad_accum_count):
    """trainer.py:378-385:

        uv_grad[:, 0] *= K[0, 0]; uv_grad[:, 1] *= K[1, 1]
        uv_grad_accum[~culling_mask] += |uv_grad|; xyz_grad_accum += |xyz.grad|
        grad_accum_count += (~culling_mask).int()

    uv_grad: [V, 2] (any row stride, e.g. the view of the fused path's slab); it is NOT modified.
    The focal lengths are read from camera.K on the device (no host sync)."""
    N = culling_mask.shape[0]
    keep = ~culling_mask
    rank = torch.where(keep, torch.cumsum(keep, 0, dtype=torch.int32) - 1,
                       torch.full((), -1, dtype=torch.int32, device=keep.device)).to(torch.int32)
    if uv_grad.stride(1) != 1:
        uv_grad = uv_grad.contiguous()
    K = camera.K if (camera.K.dtype == torch.float32 and camera.K.is_contiguous()) else \
        camera.K.to(torch.float32).contiguous()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _hip.call("gs_accumulate_grad_stats", p(uv_grad), int(uv_grad.stride(0)) if uv_grad.shape[0] else 2, p(rank),
              p(xyz_grad), p(K), N, p(uv_grad_accum), p(xyz_grad_accum),
              p(grad_accum_count), _hip.current_stream())


class _SsimL1Loss(torch.autograd.Function):
    """value and d loss / d image from one launch; backward scales the stored gradient"""

    @staticmethod
    def forward(ctx, image, target, ssim_frac):
        H, W = image.shape[0], image.shape[1]
        dev = image.device
        ws = torch.empty(_hip.lib().gs_ssim_l1_workspace_bytes(H, W) // 8, dtype=torch.float64, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        need_grad = image.requires_grad
        grad = torch.empty_like(image) if need_grad else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _hip.call("gs_ssim_l1_loss", p(image), p(target), H, W, ctypes.c_float(float(ssim_frac)), p(ws), p(out),
                  p(grad), _hip.current_stream())
        ctx.grad = grad
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_loss, _unused):
        if ctx.grad is None or g_loss is None:
            return None, None, None
        return ctx.grad * g_loss, None, None


def ssim_l1_loss(image, target, ssim_frac=0.2, return_terms=False):
    """The reference's training loss (trainer.py:363-374):

        (1 - ssim_frac) * l1_loss(image, target) + ssim_frac * (1 - SSIM(image, target))

    image, target: [H, W, 3] fp32 device tensors (the rasterizer's layout; the reference permutes
    them to NCHW for torchmetrics).  -> loss (0-d, differentiable w.r.t. image); with
    return_terms=True also the detached (loss, l1, ssim, mse) tensor (psnr = -10 log10(mse),
    trainer.py:366-367)."""
    if not (image.is_cuda and image.dtype == torch.float32 and image.dim() == 3 and image.shape[2] == 3):
        raise RuntimeError("ssim_l1_loss takes [H, W, 3] float32 device tensors")
    if target.shape != image.shape or target.dtype != image.dtype or target.device != image.device:
        raise RuntimeError("ssim_l1_loss: target must match image")
    loss, terms = _SsimL1Loss.apply(image.contiguous(), target.contiguous(), ssim_frac)
    return (loss, terms) if return_terms else loss
