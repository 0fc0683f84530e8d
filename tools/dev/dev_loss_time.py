"""Dev script (not a test): times the reference-style torch L1+SSIM loss against the fused HIP op at N3V resolution."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from math import exp
from ex4dgs_amd.loss import l1_ssim_loss
def gaussian(ws, sigma):
    g = torch.Tensor([exp(-(x - ws // 2) ** 2 / float(2 * sigma ** 2)) for x in range(ws)]); return g / g.sum()
def create_window(ws, ch):
    w1 = gaussian(ws, 1.5).unsqueeze(1); w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(ch, 1, ws, ws).contiguous()
def ssim(img1, img2, reduce=True):
    ch = img1.size(-3); window = create_window(11, ch).to(img1.device)
    mu1 = F.conv2d(img1, window, padding=5, groups=ch); mu2 = F.conv2d(img2, window, padding=5, groups=ch)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=5, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=5, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=5, groups=ch) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if reduce else m
H, W = 1014, 1352
img = torch.rand(3, H, W, device="cuda", requires_grad=True); gt = torch.rand(3, H, W, device="cuda")
def step_torch():
    img.grad = None
    Ll1 = (img - gt).abs().mean()
    loss = 0.8 * Ll1 + 0.2 * (1.0 - ssim(img, gt))
    l1e = (img - gt).abs().mean(dim=0); se = ssim(img, gt, reduce=False).mean(dim=0)
    loss.backward()
    return loss, l1e, se
def step_fused():
    img.grad = None
    loss, l1e, se = l1_ssim_loss(img, gt, 0.2)
    loss.backward()
    return loss, l1e, se
for name, fn in (("torch", step_torch), ("fused HIP", step_fused)):
    for _ in range(3): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = fn()
    torch.cuda.synchronize(); print(f"{name} L1+SSIM loss fwd+bwd (+ error maps): {1e3 * (time.perf_counter() - t0) / 20:.3f} ms  loss={r[0].item():.7f}")
    g = img.grad.clone()
    if name == "torch": g0, r0 = g, r
print("max |grad diff|", (g - g0).abs().max().item(), "max |grad|", g0.abs().max().item(), "ssim map diff", (r[2] - r0[2]).abs().max().item())
