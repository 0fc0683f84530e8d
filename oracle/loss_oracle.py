"""CPU oracle of the L1 + SSIM training loss -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu legs).

Restates utils/loss_utils.py:22-25 (l1_loss), :32-41 (gaussian / create_window), :47-81 (ssim / _ssim) of the
reference and their combination at train.py:144-151, in float64 torch on the CPU with autograd for the gradient.
The window taps are the reference's float32 values (float32 exponentials normalised by their float32 sum, 2-D window =
float32 outer product) promoted to float64, so the only difference to the reference is accumulation precision.
Pinned by tests/golden/loss_l1_ssim.npz (outputs and autograd gradients of the imported reference functions).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def window_1d(window_size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return g / g.sum()                                                         # loss_utils.py:32-34


def window_2d(channel, window_size=11):
    w1 = window_1d(window_size).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)                       # :38-39
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def l1_ssim(image, gt, lambda_dssim, dtype=torch.float64):
    """image, gt: numpy [C,H,W].  Returns dict(loss, l1_errors, ssim_errors, ssim_map, grad) as numpy (grad = dloss/dimage)."""
    x = torch.tensor(np.asarray(image), dtype=dtype).requires_grad_(True)
    y = torch.tensor(np.asarray(gt), dtype=dtype)
    Cn = x.shape[0]
    w = window_2d(Cn).to(dtype)
    pad = 11 // 2
    conv = lambda a: F.conv2d(a.unsqueeze(0), w, padding=pad, groups=Cn)[0]   # :61-62 (3-D input = unbatched)
    mu1, mu2 = conv(x), conv(y)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(x * x) - mu1_sq                                          # :68-70
    sigma2_sq = conv(y * y) - mu2_sq
    sigma12 = conv(x * y) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))   # :75
    ll1 = torch.abs(x - y).mean()                                             # :22-25
    loss = (1.0 - lambda_dssim) * ll1 + lambda_dssim * (1.0 - ssim_map.mean())   # train.py:145
    loss.backward()
    return dict(loss=loss.item(), l1_errors=torch.abs(x - y).mean(0).detach().numpy(), ssim_errors=ssim_map.mean(0).detach().numpy(),
                ssim_map=ssim_map.detach().numpy(), grad=x.grad.numpy())
