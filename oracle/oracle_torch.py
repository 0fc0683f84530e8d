"""Pure-PyTorch CPU rasterizer (TEST INFRASTRUCTURE ONLY; also the "non-optimised pure-PyTorch CPU
baseline" BASELINE.json asks to be timed on the host cores).

Independent, vectorised formulation (pixels x Gaussians dense tensors) of the same algorithm the C
oracle restates line by line: preprocess CR/forward.cu:165-269, binning CR/rasterizer_impl.cu:72-140,
compositing CR/forward.cu:274-462 (CR/ = submodules/diff_gaussian_rasterization_df/cuda_rasterizer/).
It is differentiable, so torch autograd pins the TRUE-gradient subset of the reference's analytical
backward: colour path only, alpha-clamp passed straight through (CR/backward.cu:588),
covariance->mean path cut (CR/backward.cu:414 overwrites it), gradients w.r.t. w = opacity*coef.
Only usable for small scenes: memory is O(H*W*P).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def eval_sh_rgb(deg, sh, dirs):
    """sh [P,16,3], dirs [P,3] normalised -> rgb before the +0.5 / clamp (CR/forward.cu:30-62)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
             + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
             + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
             + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
             + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return r


def rasterize(means3D, dir3D, opacities, shs, scales, rotations, *, bg, viewmatrix, projmatrix, campos,
              image_height, image_width, tanfovx, tanfovy, kernel_size, sh_degree, subpixel_offset=None,
              scale_modifier=1.0, min_depth=0.2, max_depth=100.0, colors_precomp=None, cut_cov_mean_path=True, return_dense=False):
    """Returns dict(color[3,H,W], radii[P], depth[1,H,W], flow[3,H,W], acc[1,H,W], idx[1,H,W], n_contrib, final_T,
    num_rendered, plus differentiable intermediates means2D_pix / conic / w / rgb with retained grads)."""
    H, W = int(image_height), int(image_width)
    P = means3D.shape[0]
    f32 = torch.float32
    V, PM = viewmatrix.to(f32), projmatrix.to(f32)
    ones = torch.ones(P, 1, dtype=f32)
    hom = torch.cat([means3D, ones], 1) @ PM            # row-vector convention: mem[c*4+r] = M[r][c]
    p_w = 1.0 / (hom[:, 3] + 0.0000001)
    p_proj = hom[:, :3] * p_w[:, None]
    p_view = (torch.cat([means3D, ones], 1) @ V)[:, :3]
    vis = ~((p_view[:, 2] <= min_depth) | (p_view[:, 2] > max_depth) | (p_proj[:, 0].double().abs() > 1.3) | (p_proj[:, 1].double().abs() > 1.3))

    # cov3D = R S^2 R^T with the raw quaternion (CR/forward.cu:128-162)
    q = rotations
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).view(P, 3, 3)
    S = scale_modifier * scales
    Mm = Rm * S[:, None, :]
    Sigma = Mm @ Mm.transpose(1, 2)

    # cov2D (CR/forward.cu:74-124)
    mean_for_cov = means3D.detach() if cut_cov_mean_path else means3D
    t = (torch.cat([mean_for_cov, ones], 1) @ V)[:, :3]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    tx = torch.clamp(t[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(t[:, 1] / tz, -limy, limy) * tz
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], -1).view(P, 2, 3)
    W3 = V[:3, :3].t()                                  # math view rotation
    Tm = J @ W3
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a0, b, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]
    det0 = torch.clamp_min((a0 * c0 - b * b).double(), 1e-6).float()
    det1 = torch.clamp_min(((a0 + kernel_size) * (c0 + kernel_size) - b * b).double(), 1e-6).float()
    coef = torch.sqrt(det0.double() / (det1.double() + 1e-6) + 1e-6).float()
    coef = torch.where((det0.double() <= 1e-6) | (det1.double() <= 1e-6), torch.zeros_like(coef), coef)
    a, c = a0 + kernel_size, c0 + kernel_size
    det = a * c - b * b
    vis = vis & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    lam2 = mid - torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam, lam2))).detach()
    pix = torch.stack([((p_proj[:, 0].double() + 1.0) * W - 1.0) * 0.5, ((p_proj[:, 1].double() + 1.0) * H - 1.0) * 0.5], -1).float()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ri = radius.to(torch.int32).float()
    pd = pix.detach()
    rmin_x = ((pd[:, 0] - ri) / 16).trunc().clamp(0, gx).int()
    rmin_y = ((pd[:, 1] - ri) / 16).trunc().clamp(0, gy).int()
    rmax_x = ((pd[:, 0] + ri + 16 - 1) / 16).trunc().clamp(0, gx).int()
    rmax_y = ((pd[:, 1] + ri + 16 - 1) / 16).trunc().clamp(0, gy).int()
    tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
    vis = vis & (tiles > 0)
    radii = torch.where(vis, radius.to(torch.int32), torch.zeros(P, dtype=torch.int32))

    if colors_precomp is None:
        d = means3D - campos[None]
        dirs = d / d.norm(dim=1, keepdim=True)
        raw = eval_sh_rgb(sh_degree, shs, dirs) + 0.5
        rgb = torch.clamp_min(raw, 0.0)
    else:
        rgb = colors_precomp
    # the reference computes, then drops, every gradient through coef (CR/backward.cu:201-218)
    w = opacities.reshape(P) * coef.detach()
    for tns in (pix, conic, w, rgb):
        if tns.requires_grad:
            tns.retain_grad()

    # binning: stable sort by depth bits, membership by tile rect (equivalent to the (tile|depth) key sort)
    depth = p_view[:, 2]
    order = torch.sort(torch.where(vis, depth.detach(), torch.full_like(depth, float("inf"))), stable=True)[1]
    nv = int(vis.sum())
    order = order[:nv]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    sub = torch.zeros(H, W, 2) if subpixel_offset is None else subpixel_offset.to(f32)
    pfx = (xs.float() + sub[..., 0]).reshape(-1, 1)
    pfy = (ys.float() + sub[..., 1]).reshape(-1, 1)
    tix, tiy = (xs // 16).reshape(-1, 1), (ys // 16).reshape(-1, 1)
    o = order
    member = (tix >= rmin_x[o][None]) & (tix < rmax_x[o][None]) & (tiy >= rmin_y[o][None]) & (tiy < rmax_y[o][None])  # [HW, nv]

    dx = pix[o, 0][None] - pfx
    dy = pix[o, 1][None] - pfy
    cn = conic[o]
    power = -0.5 * (cn[:, 0][None] * dx * dx + cn[:, 2][None] * dy * dy) - cn[:, 1][None] * dx * dy
    G = torch.exp(power)
    raw_alpha = w[o][None] * G
    alpha = raw_alpha + (torch.clamp(raw_alpha, max=0.99) - raw_alpha).detach()      # straight-through clamp
    contrib = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones(H * W, 1), T_incl[:, :-1]], 1)
    stop = contrib & (T_incl.detach() < 0.0001)
    alive = (torch.cumsum(stop.int(), 1) == 0)
    use = contrib & alive
    wgt = torch.where(use, a_eff * T_excl, torch.zeros_like(alpha))
    color = wgt @ rgb[o]
    acc = wgt.sum(1)
    Dsum = wgt @ depth[o]
    Fsum = wgt @ dir3D[o]
    # final T = product over used contributors only
    T_final = torch.prod(torch.where(use, one_m, torch.ones_like(one_m)), 1)
    color = color + T_final[:, None] * bg[None]
    nz = acc != 0
    accs = torch.where(nz, acc, torch.ones_like(acc))
    depth_out = torch.where(nz, Dsum / accs, torch.full_like(acc, max_depth))
    flow_out = torch.where(nz[:, None], Fsum / accs[:, None], torch.zeros_like(Fsum))
    # dominant index: first arg-max of wgt (strict >, first wins); -1 where nothing contributed
    mx, am = wgt.detach().max(1)
    idx = torch.where(mx > 0, o[am].int(), torch.full_like(am, -1).int())
    # n_contrib: position (1-based) in the tile's list of the last used contributor
    pos_in_list = torch.cumsum(member.int(), 1)
    n_contrib = torch.where(use, pos_in_list, torch.zeros_like(pos_in_list)).max(1)[0]
    dense = None
    if return_dense:
        # the pixels x (visible Gaussians in depth order) tensors, for closed-form checks (tests: quirk terms of the backward)
        dense = dict(order=o, use=use, alpha=a_eff.detach(), G=G.detach(), dx=dx.detach(), dy=dy.detach(), T_before=T_excl.detach(),
                     conic=cn.detach(), w=w[o].detach(), rgb=rgb[o].detach(), depth=depth[o].detach())
    return dict(dense=dense, color=color.t().reshape(3, H, W), radii=radii, depth=depth_out.reshape(1, H, W),
                flow=flow_out.t().reshape(3, H, W), acc=acc.reshape(1, H, W), idx=idx.reshape(1, H, W),
                n_contrib=n_contrib.reshape(H, W), final_T=T_final.reshape(H, W), num_rendered=int(tiles[vis].sum()),
                means2D_pix=pix, conic=conic, w=w, rgb=rgb, tiles_touched=torch.where(vis, tiles, torch.zeros_like(tiles)),
                depths=depth, vis=vis)
