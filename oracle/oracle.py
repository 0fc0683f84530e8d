"""ctypes/numpy front-end of the CPU oracle (oracle/ex4d_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (ex4dgs_amd/).

Mirrors the staging of CudaRasterizer::Rasterizer::forward / ::backward
(submodules/diff_gaussian_rasterization_df/cuda_rasterizer/rasterizer_impl.cu:204-363, :367-486) and
the tensor marshalling of RasterizeGaussiansCUDA / RasterizeGaussiansBackwardCUDA
(submodules/diff_gaussian_rasterization_df/rasterize_points.cu:35-133, :135-234), exposing every
intermediate the reference keeps in its opaque geom/binning/img byte buffers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile libex4d_oracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "libex4d_oracle.so")
    src = os.path.join(_HERE, "ex4d_oracle.c")
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libex4d_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ex4d_oracle_preprocess.restype = C.c_int64
        _LIB.ex4d_oracle_getHigherMsb.restype = C.c_uint32
    return _LIB


def _np(x, dtype):
    """torch tensor / array-like -> contiguous numpy array of dtype (None stays None, empty -> None)."""
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(x), dtype=dtype)
    if a.size == 0:
        return None
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def get_higher_msb(n):
    return int(lib().ex4d_oracle_getHigherMsb(C.c_uint32(n)))


def mark_visible(means3D, viewmatrix, projmatrix, min_depth, max_depth):
    m = _np(means3D, np.float32)
    P = 0 if m is None else m.shape[0]
    present = np.zeros(P, np.uint8)
    if P:
        lib().ex4d_oracle_mark_visible(C.c_int(P), _p(m), _p(_np(viewmatrix, np.float32)), _p(_np(projmatrix, np.float32)),
                                       C.c_float(min_depth), C.c_float(max_depth), _p(present))
    return present.astype(bool)


def forward(means3D, dir3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, bg, viewmatrix, projmatrix, campos, image_height, image_width,
            tanfovx, tanfovy, kernel_size, subpixel_offset=None, scale_modifier=1.0, sh_degree=0,
            prefiltered=False, min_depth=0.2, max_depth=100.0, want_fragile=True, frag_eps=1e-4):
    """Returns a dict with the six public outputs (color, radii, depth, flow, acc, idx), num_rendered,
    and every internal array (geometry / binning / image state)."""
    L = lib()
    H, W = int(image_height), int(image_width)
    means3D = _np(means3D, np.float32)
    P = 0 if means3D is None else means3D.shape[0]
    shs = _np(shs, np.float32)
    colors_precomp = _np(colors_precomp, np.float32)
    scales = _np(scales, np.float32)
    rotations = _np(rotations, np.float32)
    cov3D_precomp = _np(cov3D_precomp, np.float32)
    opacities = _np(opacities, np.float32)
    dir3D = _np(dir3D, np.float32)
    bg = _np(bg, np.float32)
    vm = _np(viewmatrix, np.float32)
    pm = _np(projmatrix, np.float32)
    cp = _np(campos, np.float32)
    if subpixel_offset is None:
        subpixel_offset = np.zeros((H, W, 2), np.float32)
    sub = _np(subpixel_offset, np.float32)
    M = 0 if shs is None else shs.shape[1]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy

    out = dict(P=P, W=W, H=H, M=M, D=int(sh_degree), grid=(gx, gy))
    out["color"] = np.zeros((3, H, W), np.float32)
    out["radii"] = np.zeros(P, np.int32)
    out["depth"] = np.zeros((1, H, W), np.float32)
    out["acc"] = np.zeros((1, H, W), np.float32)
    out["flow"] = np.zeros((3, H, W), np.float32)
    out["idx"] = np.full((1, H, W), -1, np.int32)
    out["num_rendered"] = 0
    if P == 0:
        return out

    g = dict(
        means2D=np.zeros((P, 2), np.float32), depths=np.zeros(P, np.float32), cov3D=np.zeros((P, 6), np.float32),
        rgb=np.zeros((P, 3), np.float32), conic_opacity=np.zeros((P, 4), np.float32),
        tiles_touched=np.zeros(P, np.uint32), clamped=np.zeros((P, 3), np.uint8), point_offsets=np.zeros(P, np.uint32))
    R = L.ex4d_oracle_preprocess(
        C.c_int(P), C.c_int(int(sh_degree)), C.c_int(M),
        _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations), _p(opacities), _p(shs),
        _p(cov3D_precomp), _p(colors_precomp), _p(vm), _p(pm), _p(cp),
        C.c_int(W), C.c_int(H), C.c_float(tanfovx), C.c_float(tanfovy), C.c_float(kernel_size),
        C.c_float(min_depth), C.c_float(max_depth), C.c_int(int(bool(prefiltered))),
        _p(out["radii"]), _p(g["means2D"]), _p(g["depths"]), _p(g["cov3D"]), _p(g["rgb"]), _p(g["conic_opacity"]),
        _p(g["tiles_touched"]), _p(g["clamped"]), _p(g["point_offsets"]))
    if R < 0:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    R = int(R)
    out.update(g)
    out["num_rendered"] = R

    b = dict(keys_unsorted=np.zeros(R, np.uint64), values_unsorted=np.zeros(R, np.uint32),
             keys_sorted=np.zeros(R, np.uint64), point_list=np.zeros(R, np.uint32), ranges=np.zeros((T, 2), np.uint32))
    L.ex4d_oracle_binning(C.c_int(P), C.c_int(W), C.c_int(H), C.c_int64(R),
                          _p(out["radii"]), _p(g["means2D"]), _p(g["depths"]), _p(g["point_offsets"]),
                          _p(b["keys_unsorted"]), _p(b["values_unsorted"]), _p(b["keys_sorted"]), _p(b["point_list"]),
                          _p(b["ranges"]))
    out.update(b)

    features = colors_precomp if colors_precomp is not None else g["rgb"]
    if dir3D is None:
        dir3D = np.zeros((P, 3), np.float32)
    out["final_T"] = np.zeros((H, W), np.float32)
    out["n_contrib"] = np.zeros((H, W), np.uint32)
    out["fragile"] = np.ones((H, W), np.float32) if want_fragile else None
    out["idx_margin"] = np.ones((H, W), np.float32) if want_fragile else None
    # summed blending weight of the decisions within frag_eps of their threshold: bounds what a flipped decision moves (per pixel)
    out["flip_w"] = np.zeros((H, W), np.float32) if want_fragile else None
    L.ex4d_oracle_render_fwd(
        C.c_int(W), C.c_int(H), _p(b["ranges"]), _p(b["point_list"]), _p(sub), _p(g["means2D"]), _p(features),
        _p(g["conic_opacity"]), _p(g["depths"]), _p(dir3D), _p(bg), C.c_float(min_depth), C.c_float(max_depth),
        _p(out["final_T"]), _p(out["n_contrib"]), _p(out["color"]), _p(out["depth"]), _p(out["acc"]), _p(out["flow"]),
        _p(out["idx"]), _p(out["fragile"]), _p(out["idx_margin"]), C.c_float(frag_eps), _p(out["flip_w"]))
    out["_inputs"] = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations,
                          cov3D_precomp=cov3D_precomp, bg=bg, vm=vm, pm=pm, cp=cp, sub=sub, tanfovx=float(tanfovx),
                          tanfovy=float(tanfovy), kernel_size=float(kernel_size), scale_modifier=float(scale_modifier),
                          min_depth=float(min_depth), max_depth=float(max_depth))
    return out


def backward(fwd, grad_color, grad_depth, grad_flow, grad_acc, want_sums=True, pixel_order=None, stage=True, state_delta=None):
    """fwd = dict returned by forward().  Returns the nine tensors of
    RasterizeGaussiansBackwardCUDA (rasterize_points.cu:233) plus internals.
    pixel_order: optional permutation of the H*W pixel ids -- the order in which the pixels' terms reach the float32 accumulators
    (the reference's atomicAdd order is arbitrary; see backward_noise).  stage=False stops after the compositing backward.
    state_delta: optional (|delta out_depth|, |delta out_acc|, |delta final_T|) per pixel, [H,W] each -- how far ANOTHER forward
    evaluation's per-pixel state lies from fwd's; res['state13'] then holds the first-order bound on what that difference moves in
    each of the 13 accumulators (the conditioning of the backward with respect to the forward state, e.g. dL_ddepth / acc)."""
    L = lib()
    P, W, H, M, D = fwd["P"], fwd["W"], fwd["H"], fwd["M"], fwd["D"]
    res = dict(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
        dL_ddir=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 4), np.float32))
    if P == 0:
        return res
    i = fwd["_inputs"]
    gc = _np(grad_color, np.float32)
    gd = _np(grad_depth, np.float32)
    gf = _np(grad_flow, np.float32)
    ga = _np(grad_acc, np.float32)
    colors = i["colors_precomp"] if i["colors_precomp"] is not None else fwd["rgb"]
    res["sum13"] = np.zeros((P, 13), np.float64) if want_sums else None
    res["abs13"] = np.zeros((P, 13), np.float64) if want_sums else None
    # like abs13, with the parts of dL_dalpha taken before they cancel (ex4d_oracle.c: cmag13); the other seven accumulators = abs13
    res["cmag13"] = np.zeros((P, 13), np.float64) if want_sums else None
    dstate = None
    res["state13"] = None
    if state_delta is not None:
        dstate = np.ascontiguousarray(np.stack([np.abs(_np(x, np.float32)).reshape(H, W) for x in state_delta]), dtype=np.float32)
        res["state13"] = np.zeros((P, 13), np.float64)
    L.ex4d_oracle_render_bwd(
        C.c_int(W), C.c_int(H), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(i["sub"]), _p(i["bg"]),
        _p(fwd["means2D"]), _p(fwd["conic_opacity"]), _p(colors), _p(fwd["depths"]),
        _p(fwd["depth"]), _p(fwd["acc"]), C.c_float(i["min_depth"]), C.c_float(i["max_depth"]),
        _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(gc), _p(gd), _p(gf), _p(ga),
        _p(res["dL_dmeans2D"]), _p(res["dL_dconic"]), _p(res["dL_ddir"]), _p(res["dL_dopacity"]), _p(res["dL_dcolors"]),
        _p(res["sum13"]), _p(res["abs13"]), _p(None if pixel_order is None else np.ascontiguousarray(pixel_order, dtype=np.uint32)),
        _p(dstate), _p(res["state13"]), _p(res["cmag13"]))
    if want_sums:
        res["cmag13"] = np.maximum(res["cmag13"], res["abs13"])
    if stage:
        preprocess_backward(fwd, res)
    return res


def acc13_of(res):
    """The 13 accumulated quantities per Gaussian of a backward() result as one [P,13] float32 array (index convention of
    ex4d_oracle_render_bwd: 0..2 dL_dmean2D, 3..5 dL_dconic.(x,y,w), 6 dL_dopacity, 7..9 dL_dcolor, 10..12 dL_ddir)."""
    c = res["dL_dconic"]
    return np.concatenate([res["dL_dmeans2D"], c[:, 0:2], c[:, 3:4], res["dL_dopacity"].reshape(-1, 1), res["dL_dcolors"], res["dL_ddir"]],
                          axis=1).astype(np.float32)


def backward_noise(fwd, grad_color, grad_depth, grad_flow, grad_acc, orders=8, seed=0, threads=None):
    """The reference's own float32 noise floor: its compositing backward adds every pixel's terms with float atomicAdd
    (CR/backward.cu:613-679), so the summation order -- and with it the float32 result -- changes from run to run.  Replays the
    oracle's backward with the pixel loop in `orders` random orders (float32 accumulation, like the reference) and returns the
    list of [P,13] float32 accumulator arrays, one per order.  The spread of these around the exact (double) sums is what two
    runs of the reference itself differ by; the parity tolerances are stated in multiples of it (tests/helpers.py)."""
    from concurrent.futures import ThreadPoolExecutor
    HW = fwd["H"] * fwd["W"]
    rng = np.random.default_rng(seed)
    perms = [rng.permutation(HW).astype(np.uint32) for _ in range(orders)]

    def run(perm):
        r = backward(fwd, grad_color, grad_depth, grad_flow, grad_acc, want_sums=False, pixel_order=perm, stage=False)
        return acc13_of(r)
    if threads is None:
        threads = min(orders, os.cpu_count() or 1)
    if threads <= 1:
        return [run(p) for p in perms]
    with ThreadPoolExecutor(threads) as ex:           # ctypes releases the GIL inside the C call
        return list(ex.map(run, perms))


def preprocess_backward(fwd, res):
    """Second half of Rasterizer::backward (rasterizer_impl.cu:457-485) on res['dL_dmeans2D'/'dL_dconic'/'dL_dcolors'];
    fills dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations in place."""
    L = lib()
    i = fwd["_inputs"]
    P, W, H, M, D = fwd["P"], fwd["W"], fwd["H"], fwd["M"], fwd["D"]
    for k in ("dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        res[k][...] = 0
    cov3D_ptr = i["cov3D_precomp"] if i["cov3D_precomp"] is not None else fwd["cov3D"]
    L.ex4d_oracle_preprocess_bwd(
        C.c_int(P), C.c_int(D), C.c_int(M), _p(i["means3D"]), _p(fwd["radii"]), _p(i["shs"]), _p(fwd["clamped"]),
        _p(i["scales"]), _p(i["rotations"]), C.c_float(i["scale_modifier"]), _p(cov3D_ptr), _p(i["vm"]), _p(i["pm"]),
        C.c_int(W), C.c_int(H), C.c_float(i["tanfovx"]), C.c_float(i["tanfovy"]), C.c_float(i["kernel_size"]), _p(i["cp"]),
        _p(res["dL_dmeans2D"]), _p(res["dL_dconic"]), _p(res["dL_dmeans3D"]), _p(res["dL_dcolors"]), _p(res["dL_dcov3D"]),
        _p(res["dL_dsh"]), _p(res["dL_dscales"]), _p(res["dL_drotations"]))
    return res
