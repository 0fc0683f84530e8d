"""Fused L1 + SSIM training loss: one HIP forward + one HIP backward through the C ABI of
include/ex4d_loss.h, instead of the reference's l1_loss + ssim and their autograd graph
(utils/loss_utils.py:22-25, :47-81 as combined by train.py:144-151 of the reference).

    loss, l1_errors, ssim_errors = l1_ssim_loss(image, gt_image, lambda_dssim)
    loss, l1_errors, ssim_errors, hook_tensor = l1_ssim_loss(image, gt_image, lambda_dssim, acc=acc)

`loss` is differentiable w.r.t. `image`; the two [H,W] error maps are the per-pixel channel means the reference hands
to its densification statistics (train.py:149-150) and carry no gradient.  No CPU fallback.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _C

EXPORTS = ("ex4d_loss_last_error", "ex4d_l1_ssim_scratch_floats", "ex4d_l1_ssim_forward", "ex4d_l1_ssim_backward")
WINDOW_SIZE = 11


def gaussian_window(window_size=WINDOW_SIZE, sigma=1.5):
    """The 1-D taps of loss_utils.py:32-34: float32 tensor of Python-double exponentials, divided by its float32 sum."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)], dtype=torch.float32)
    return (g / g.sum()).numpy()


_WINDOW = gaussian_window()


def _lib():
    lib = _C.load()
    if not getattr(lib, "_loss_ready", False):
        lib.ex4d_loss_last_error.restype = C.c_char_p
        lib.ex4d_l1_ssim_scratch_floats.restype = C.c_size_t
        lib.ex4d_l1_ssim_scratch_floats.argtypes = [C.c_int32, C.c_int32]
        lib.ex4d_l1_ssim_forward.restype = C.c_int
        lib.ex4d_l1_ssim_forward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 2 + [C.c_float] + [C.c_void_p] * 7
        lib.ex4d_l1_ssim_backward.restype = C.c_int
        lib.ex4d_l1_ssim_backward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 2 + [C.c_float] + [C.c_void_p] * 5
        lib._loss_ready = True
    return lib


class _L1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim, errors):
        lib = _lib()
        if not image.is_cuda:
            raise RuntimeError(f"image is on {image.device}: the fused L1+SSIM loss only runs on a ROCm GPU (no CPU fallback)")
        if image.dim() != 3 or image.shape != gt.shape or image.dtype != torch.float32 or gt.dtype != torch.float32 or gt.device != image.device:
            raise RuntimeError("image and gt_image must be float32 [C,H,W] tensors of the same shape on the same ROCm device")
        image, gt = image.contiguous(), gt.contiguous()
        Cn, H, W = image.shape
        f32 = dict(dtype=torch.float32, device=image.device)
        loss = torch.empty(1, **f32)
        l1e, sse = errors if errors is not None else (torch.empty(H, W, **f32), torch.empty(H, W, **f32))
        dmaps = torch.empty(3, Cn, H, W, **f32)
        scratch = torch.empty(lib.ex4d_l1_ssim_scratch_floats(H, W), **f32)
        with torch.cuda.device(image.device):
            rc = lib.ex4d_l1_ssim_forward(Cn, H, W, image.data_ptr(), gt.data_ptr(), float(lambda_dssim), _WINDOW.ctypes.data,
                                          loss.data_ptr(), l1e.data_ptr(), sse.data_ptr(), dmaps.data_ptr(), scratch.data_ptr(),
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(lib.ex4d_loss_last_error().decode())
        ctx.lam = float(lambda_dssim)
        ctx.save_for_backward(image, gt, dmaps)
        ctx.mark_non_differentiable(l1e, sse)
        return loss.reshape(()), l1e, sse

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        lib = _lib()
        image, gt, dmaps = ctx.saved_tensors
        Cn, H, W = image.shape
        g = g_loss.reshape(1).to(torch.float32).contiguous()
        grad = torch.empty_like(image)
        with torch.cuda.device(image.device):
            rc = lib.ex4d_l1_ssim_backward(Cn, H, W, image.data_ptr(), gt.data_ptr(), ctx.lam, _WINDOW.ctypes.data, dmaps.data_ptr(),
                                           g.data_ptr(), grad.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc:
            raise RuntimeError(lib.ex4d_loss_last_error().decode())
        return grad, None, None, None


def l1_ssim_loss(image, gt_image, lambda_dssim=0.2, acc=None):
    """(loss, l1_errors[H,W], ssim_errors[H,W]) of train.py:144-151 for a [C,H,W] render and its ground truth.
    With `acc` (the rasterizer's [1,H,W] accumulation output) a fourth value is returned: the [3,H,W] tensor
    stack([acc[0], l1_errors, ssim_errors]) the reference installs as the gradient of the flow image (train.py:151-152);
    the two error maps are then written straight into it (they are views of it)."""
    if acc is None:
        return _L1SSIM.apply(image, gt_image, lambda_dssim, None)
    H, W = image.shape[-2:]
    hook = torch.empty(3, H, W, dtype=torch.float32, device=image.device)
    hook[0].copy_(acc.detach()[0])
    loss, l1e, sse = _L1SSIM.apply(image, gt_image, lambda_dssim, (hook[1], hook[2]))
    return loss, l1e, sse, hook


def l1_ssim_loss_unfused(image, gt_image, lambda_dssim=0.2):
    """The same quantities as the composition of torch ops the reference executes (five depthwise conv2d per ssim call, two
    ssim calls, ~25 element-wise kernels and their autograd graph).  Kept ONLY so bench.py can time "before" on the same
    GPU; it is not a fallback of l1_ssim_loss."""
    import torch.nn.functional as F
    Cn = image.shape[0]
    w1 = torch.from_numpy(_WINDOW).to(image.device).unsqueeze(1)
    window = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0).expand(Cn, 1, WINDOW_SIZE, WINDOW_SIZE).contiguous()

    def ssim_map(a, b):
        mu1 = F.conv2d(a, window, padding=WINDOW_SIZE // 2, groups=Cn); mu2 = F.conv2d(b, window, padding=WINDOW_SIZE // 2, groups=Cn)
        mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
        s1 = F.conv2d(a * a, window, padding=WINDOW_SIZE // 2, groups=Cn) - mu1_sq
        s2 = F.conv2d(b * b, window, padding=WINDOW_SIZE // 2, groups=Cn) - mu2_sq
        s12 = F.conv2d(a * b, window, padding=WINDOW_SIZE // 2, groups=Cn) - mu1_mu2
        C1, C2 = 0.01 ** 2, 0.03 ** 2
        return ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    loss = (1.0 - lambda_dssim) * torch.abs(image - gt_image).mean() + lambda_dssim * (1.0 - ssim_map(image, gt_image).mean())
    l1e = (image - gt_image).abs().mean(dim=0)
    sse = ssim_map(image, gt_image).mean(dim=0)
    return loss, l1e.detach(), sse.detach()


# ---- the reference's metric functions by name (utils/loss_utils.py:22-25, :47-52; utils/image_utils.py:17-19), so that
# `from utils.loss_utils import l1_loss, ssim` / `from utils.image_utils import psnr` can point here -------------------------
def l1_loss(network_output, gt, mask=None):
    if mask is not None:
        return torch.abs((network_output - gt)[mask]).mean()
    return torch.abs(network_output - gt).mean()


def ssim(img1, img2, window_size=WINDOW_SIZE, size_average=True, reduce=True):
    """SSIM with the reference's signature, on [C,H,W] or [1,C,H,W] tensors, through the fused HIP op (differentiable w.r.t.
    img1).  reduce=True -> scalar mean; reduce=False -> the [C,H,W] map (one fused call per channel, not differentiable)."""
    if window_size != WINDOW_SIZE or not size_average:
        raise NotImplementedError("only the configuration the reference uses: window_size=11, size_average=True")
    a = img1[0] if img1.dim() == 4 else img1
    b = img2[0] if img2.dim() == 4 else img2
    if img1.dim() == 4 and img1.shape[0] != 1:
        raise NotImplementedError("batched SSIM is not used by the reference (render.py:77 passes one image)")
    if reduce:
        loss, _, _ = l1_ssim_loss(a, b.detach(), 1.0)           # lambda = 1: loss = 1 - mean(ssim_map)
        return 1.0 - loss
    with torch.no_grad():
        maps = [l1_ssim_loss(a[c:c + 1].contiguous(), b[c:c + 1].contiguous(), 1.0)[2] for c in range(a.shape[0])]
        out = torch.stack(maps)
    return out.unsqueeze(0) if img1.dim() == 4 else out


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
