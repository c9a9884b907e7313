"""The loss oracle (oracle/loss_oracle.py) is a restatement of torchmetrics 1.2.1 SSIM, which is not
importable here (PARITY UNPINNED, see its header).  These tests hold it to an independent numpy
evaluation of the published formula and to the properties the formula implies."""
import numpy as np
import torch

from oracle import loss_oracle


def numpy_ssim(x, y):
    g = np.exp(-((np.arange(11) - 5) / 1.5) ** 2 / 2)
    g /= g.sum()
    w = np.outer(g, g)
    H, W, C = x.shape
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    vals = []
    for c in range(C):
        for i in range(5, H - 5):
            for j in range(5, W - 5):
                a, b = x[i - 5:i + 6, j - 5:j + 6, c], y[i - 5:i + 6, j - 5:j + 6, c]
                mx, my = (w * a).sum(), (w * b).sum()
                sxx, syy, sxy = (w * a * a).sum() - mx * mx, (w * b * b).sum() - my * my, (w * a * b).sum() - mx * my
                vals.append(((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2)))
    return float(np.mean(vals))


def test_ssim_restatement_equals_the_formula_on_interior_windows():
    rng = np.random.default_rng(0)
    x, y = rng.random((20, 23, 3)), rng.random((20, 23, 3))
    got = loss_oracle.ssim(torch.from_numpy(x), torch.from_numpy(y))
    assert abs(float(got) - numpy_ssim(x, y)) < 1e-12


def test_loss_terms():
    torch.manual_seed(0)
    x, y = torch.rand(30, 31, 3), torch.rand(30, 31, 3)
    loss, l1, s = loss_oracle.ssim_l1_loss(x, y, 0.2)
    assert abs(float(loss) - (0.8 * float((x - y).abs().mean()) + 0.2 * (1 - float(s)))) < 1e-7
    assert abs(float(loss_oracle.ssim(x, x)) - 1.0) < 1e-6
    assert abs(float(loss_oracle.ssim(x, y)) - float(loss_oracle.ssim(y, x))) < 1e-7
