"""CPU restatement of the reference's training loss (TEST INFRASTRUCTURE ONLY: used by tests/ as the
checker of gs_ssim_l1_loss, never by the product path).

Reference call site: splat_py/trainer.py:363-374

    l1_loss = torch.nn.functional.l1_loss(image, gt_image)
    ssim_loss = 1.0 - self.ssim(image[None].permute(0, 3, 1, 2), gt_image[None].permute(0, 3, 1, 2))
    loss = (1.0 - ssim_frac) * l1_loss + ssim_frac * ssim_loss

with self.ssim = torchmetrics.image.StructuralSimilarityIndexMeasure(data_range=1.0) (trainer.py:24),
torchmetrics pinned at 1.2.1 (requirements.txt:13).  torchmetrics is NOT importable in this image, so
the algorithm of torchmetrics/functional/image/ssim.py (_ssim_update, 1.2.1) is restated from its
published source; PARITY UNPINNED for this function: no golden vector of the reference exists for it.
Restated behaviour (defaults gaussian_kernel=True, sigma=1.5, kernel_size=11, k1=0.01, k2=0.03):
  * 1-D Gaussian g[i] = exp(-((i - 5) / 1.5)^2 / 2) normalised to sum 1; 2-D kernel = g^T g per channel
  * inputs reflect-padded by 5, the five maps (x, y, x*x, y*y, x*y) convolved ("valid") with it
  * SSIM map = ((2 mx my + c1)(2 sxy + c2)) / ((mx^2 + my^2 + c1)(sxx + syy + c2)), c1 = 0.01^2, c2 = 0.03^2
  * the map is cropped by 5 pixels on every side AGAIN (so only windows that lie entirely inside
    the image count and the padding never reaches the result), then averaged over all elements
"""
import torch
import torch.nn.functional as F

KERNEL_SIZE, SIGMA, K1, K2 = 11, 1.5, 0.01, 0.03


def gaussian_kernel_1d(dtype=torch.float32):
    dist = torch.arange(start=(1 - KERNEL_SIZE) / 2, end=(1 + KERNEL_SIZE) / 2, step=1, dtype=dtype)
    gauss = torch.exp(-torch.pow(dist / SIGMA, 2) / 2)
    return (gauss / gauss.sum()).unsqueeze(0)   # [1, 11]


def ssim(image, target, data_range=1.0):
    """image, target: [H, W, C] -> scalar (mean SSIM), differentiable"""
    x = image.permute(2, 0, 1).unsqueeze(0)
    y = target.permute(2, 0, 1).unsqueeze(0)
    c = x.shape[1]
    c1, c2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    g = gaussian_kernel_1d(x.dtype)
    kernel = torch.matmul(g.t(), g).expand(c, 1, KERNEL_SIZE, KERNEL_SIZE)
    pad = (KERNEL_SIZE - 1) // 2
    xp = F.pad(x, (pad, pad, pad, pad), mode="reflect")
    yp = F.pad(y, (pad, pad, pad, pad), mode="reflect")
    maps = torch.cat((xp, yp, xp * xp, yp * yp, xp * yp))
    out = F.conv2d(maps, kernel, groups=c).split(1)
    mu_x_sq, mu_y_sq, mu_xy = out[0].pow(2), out[1].pow(2), out[0] * out[1]
    s_xx, s_yy, s_xy = out[2] - mu_x_sq, out[3] - mu_y_sq, out[4] - mu_xy
    upper = 2 * s_xy + c2
    lower = s_xx + s_yy + c2
    full = ((2 * mu_xy + c1) * upper) / ((mu_x_sq + mu_y_sq + c1) * lower)
    return full[..., pad:-pad, pad:-pad].reshape(1, -1).mean(-1)[0]


def ssim_l1_loss(image, target, ssim_frac):
    """trainer.py:363-374 -> (loss, l1, ssim)"""
    l1 = F.l1_loss(image, target)
    s = ssim(image, target)
    return (1.0 - ssim_frac) * l1 + ssim_frac * (1.0 - s), l1, s
