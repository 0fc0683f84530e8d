"""GPU tests of round 5's binning: the MSD-first depth sort (one global partition on the top key bits, every bucket finished in LDS,
the packed rects carried along) against the 3-pass LSD sort it replaces and against a stable host sort of the depth bits."""
import numpy as np
import pytest
import torch

from tests import helpers as h
from tests.test_gpu_round4 import _raw_forward

pytestmark = pytest.mark.gpu


def _frame(ins, st, **options):
    """One forward with the given library options; returns what the binning decides (and the image)."""
    from ex4dgs_amd import _C
    saved = {k: _C.get_option(k) for k in options}
    try:
        for k, v in options.items():
            _C.set_option(k, v)
        s, f = _raw_forward(ins, st)
        torch.cuda.synchronize()
    finally:
        for k, v in saved.items():
            _C.set_option(k, v)
    R, color, radii, geom, binning, img = f[0], f[1], f[2], f[3], f[4], f[5]
    P = ins["means3D"].shape[0]
    H, W = st["image_height"], st["image_width"]
    g = _C.geom_views(geom, P)
    return dict(R=R, color=color.clone(), radii=radii.clone(), depth_order=g["depth_order"].clone(), depths=g["depths"].clone(),
                point_list=_C.binning_views(binning, R, W, H)["point_list"][:R].clone(), ranges=_C.img_views(img, W, H)["ranges"].clone(),
                depth=f[6].clone(), acc=f[7].clone(), idx=f[9].clone())


def _host_depth_order(fr):
    """Stable ascending sort of the visible Gaussians' depth bit patterns (CR/rasterizer_impl.cu:106: the low key word), ties in
    ascending id, invisible Gaussians behind them in id order."""
    vis = fr["radii"] > 0
    bits = fr["depths"].contiguous().view(torch.int32).to(torch.int64)
    key = torch.where(vis, bits, torch.full_like(bits, 1 << 40))
    return torch.sort(key, stable=True).indices.to(torch.int32)


def _same(a, b, what):
    for k in ("R",):
        assert a[k] == b[k], (what, k)
    for k in ("depth_order", "point_list", "ranges", "radii", "color", "depth", "acc", "idx"):
        assert torch.equal(a[k], b[k]), (what, k)


SCENES = [("cfg2", 20000, 0), ("cfg3", 12000, 137), ("cfg5", 6000, 0), ("cfg1", None, 0), ("cfg2", 1, 0), ("cfg2", 2049, 0)]


@pytest.mark.parametrize("cfg,P,t", SCENES)
def test_msd_depth_sort_equals_the_lsd_sort_and_a_host_sort(hip_lib, cfg, P, t):
    """depth_order / point_list / ranges / the image: bit-equal between the MSD-first depth sort (what the default, auto, runs on such scenes), the same with every bucket
    forced through global memory (local capacity 1: the path of oversize buckets), the variant that keeps the tile scan as a kernel of
    its own (depth_sort_msd = 1), and the 3-pass LSD sort; depth_order equal to a stable host sort of the depth bits."""
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    ins = {k: v.cuda() for k, v in ins.items()}
    lsd = _frame(ins, st, depth_sort_msd=0)
    msd = _frame(ins, st, depth_sort_msd=2)
    mem = _frame(ins, st, depth_sort_msd=2, depth_sort_local_cap=1, depth_sort_local_threads=512)
    mid = _frame(ins, st, depth_sort_msd=2, depth_sort_local_cap=7, depth_sort_local_threads=256)
    big = _frame(ins, st, depth_sort_msd=2, depth_sort_local_threads=512)
    unfused = _frame(ins, st, depth_sort_msd=1, depth_sort_local_cap=5)
    assert torch.equal(lsd["depth_order"], _host_depth_order(lsd)), "LSD depth order differs from the host sort"
    _same(msd, lsd, "msd vs lsd")
    _same(mem, lsd, "msd through memory vs lsd")
    _same(mid, lsd, "msd mixed vs lsd")
    _same(big, lsd, "msd with 512-thread bucket workgroups and the histogram kernel vs lsd")
    _same(unfused, lsd, "msd with the scan kernel vs lsd")


def _squeezed(P, z_lo, z_hi, ties=0, seed=5):
    """cfg2 with every depth squeezed into [z_lo, z_hi]: few MSD buckets, many Gaussians each; `ties` of them at exactly one depth."""
    ins, st = h.scene_inputs("cfg2", P=P)
    g = torch.Generator().manual_seed(seed)
    m = ins["means3D"]
    z = z_lo + (z_hi - z_lo) * torch.rand(P, generator=g)
    scale = (z / m[:, 2]).unsqueeze(1)
    ins["means3D"] = (m * scale).contiguous()            # same image position, new depth (identity view: x/z, y/z unchanged)
    ins["means3D"][:, 2] = z
    ins["scales"] = (ins["scales"] * scale).contiguous()
    if ties:
        idx = torch.randperm(P, generator=g)[:ties]
        ins["means3D"][idx, 2] = 0.5 * (z_lo + z_hi)
    return {k: v.cuda() for k, v in ins.items()}, st


@pytest.mark.parametrize("P,z_lo,z_hi,ties", [(30000, 10.0, 10.02, 0), (30000, 6.0, 6.4, 9000), (12000, 5.0, 5.0, 0), (40000, 4.5, 80.0, 20000)])
def test_msd_depth_sort_oversize_buckets_and_ties(hip_lib, P, z_lo, z_hi, ties):
    """Depths squeezed into a sliver of [min_depth, max_depth] (buckets of > 8192 Gaussians: sorted through global memory), thousands of
    exact depth ties (order falls back to ascending id: CR/rasterizer_impl.cu:321-326 is a stable sort), all depths equal."""
    ins, st = _squeezed(P, z_lo, z_hi, ties)
    lsd = _frame(ins, st, depth_sort_msd=0)
    msd = _frame(ins, st, depth_sort_msd=2)
    assert torch.equal(lsd["depth_order"], _host_depth_order(lsd))
    _same(msd, lsd, "msd vs lsd")
    _same(_frame(ins, st, depth_sort_msd=1, depth_sort_local_threads=512), lsd, "msd with the scan kernel vs lsd")
    vis = int((lsd["radii"] > 0).sum())
    assert vis > P // 2


def test_depth_sort_auto_mode_leaves_the_msd_sort_after_an_oversize_bucket(hip_lib):
    """depth_sort_msd = 3 (the default): frames take the MSD sort until its bucket kernel reports a bucket beyond the LDS capacity (a pinned
    host word); the next forward sees the report and orders the following 64 frames with the LSD sort, the hold doubling with every
    further report.  The frames themselves are bit-equal whichever sort ran."""
    from ex4dgs_amd import _C
    assert _C.get_option("depth_sort_msd") == 3, "auto is the library default"
    spread, st = h.scene_inputs("cfg2", P=20000)
    spread = {k: v.cuda() for k, v in spread.items()}
    # 14000 Gaussians at one depth: one bucket of > 8192 visible ones (round 6: up to 1.3 M Gaussians the bucket workgroups have 512 threads)
    wall, st_w = _squeezed(30000, 6.0, 6.4, 14000)
    ref_spread = _frame(spread, st, depth_sort_msd=0)
    ref_wall = _frame(wall, st_w, depth_sort_msd=0)
    try:
        _C.set_option("depth_sort_msd", 3)                    # (setting the option resets the hold and the counters)
        for _ in range(3):
            _same(_frame(spread, st), ref_spread, "auto on a spread scene")
        assert _C.get_option("depth_sort_trips") == 0 and _C.get_option("depth_sort_hold") == 0
        _same(_frame(wall, st_w), ref_wall, "auto, first wall frame (MSD, the oversize bucket through memory)")
        assert _C.get_option("depth_sort_trips") == 0        # the host has not looked yet
        _same(_frame(wall, st_w), ref_wall, "auto, second wall frame (LSD)")
        assert _C.get_option("depth_sort_trips") == 1 and _C.get_option("depth_sort_hold") == 63
        for _ in range(63):
            _frame(wall, st_w)
        assert _C.get_option("depth_sort_hold") == 0 and _C.get_option("depth_sort_trips") == 1
        _same(_frame(wall, st_w), ref_wall, "auto, probe frame after the hold (MSD again)")
        _same(_frame(spread, st), ref_spread, "auto, the frame that sees the second report")
        assert _C.get_option("depth_sort_trips") == 2 and _C.get_option("depth_sort_hold") == 127
        _C.set_option("depth_sort_msd", 3)
        assert _C.get_option("depth_sort_hold") == 0 and _C.get_option("depth_sort_trips") == 0
        # every depth equal: nothing to sort inside the one bucket, but one workgroup scans all of it -- reported like an oversize sort
        flat, st_f = _squeezed(12000, 5.0, 5.0, 0)
        ref_flat = _frame(flat, st_f, depth_sort_msd=0)
        _C.set_option("depth_sort_msd", 3)
        _same(_frame(flat, st_f), ref_flat, "auto, all depths equal (MSD)")
        _same(_frame(flat, st_f), ref_flat, "auto, all depths equal (LSD)")
        assert _C.get_option("depth_sort_trips") == 1 and _C.get_option("depth_sort_hold") == 63
    finally:
        _C.set_option("depth_sort_msd", 3)


def test_msd_depth_sort_full_size_1M(hip_lib):
    """BASELINE config 3 at its full 1.0 M Gaussians: MSD == LSD == host sort."""
    ins, st = h.scene_inputs("cfg3", t=137)
    ins = {k: v.cuda() for k, v in ins.items()}
    lsd = _frame(ins, st, depth_sort_msd=0)
    msd = _frame(ins, st, depth_sort_msd=2)
    assert torch.equal(lsd["depth_order"], _host_depth_order(lsd))
    _same(msd, lsd, "msd vs lsd at 1.0 M")
    _same(_frame(ins, st, depth_sort_msd=1, depth_sort_local_threads=512), lsd, "msd with the scan kernel vs lsd at 1.0 M")


@pytest.mark.parametrize("cfg,P,t,prepare", [("cfg2", 20000, 0, False), ("cfg3", 12001, 137, True), ("cfg5", 6000, 0, True), ("cfg2", 63, 0, True)])
def test_specialised_preprocess_kernel_is_bit_identical_to_the_generic_one(hip_lib, cfg, P, t, prepare):
    """preprocess_fwd_kernel<FAST> (one [P,16,3] SH tensor at degree 3, scale + rotation: constants at compile time, plain SH loads for
    full waves) against the generic kernel: every word of the geometry buffer the rest of the frame reads, and the frame itself."""
    from ex4dgs_amd import _C
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    ins = {k: v.cuda() for k, v in ins.items()}
    out = []
    for fast in (0, 1):
        _C.set_option("preprocess_fast_path", fast)
        try:
            s, f = _raw_forward(ins, st, prepare_backward=prepare)
            torch.cuda.synchronize()
        finally:
            _C.set_option("preprocess_fast_path", 1)
        n = ins["means3D"].shape[0]
        g = _C.geom_views(f[3], n)
        vis = f[2] > 0
        out.append(dict(R=f[0], color=f[1].clone(), radii=f[2].clone(), records=g["records"][vis].clone(), clamped=g["clamped"][vis].clone(),
                        rects=g["rects"].clone(), order=g["depth_order"].clone(), depth=f[6].clone(), acc=f[7].clone(), flow=f[8].clone(), idx=f[9].clone(),
                        geom=f[3].clone()))
    a, b = out
    assert a["R"] == b["R"] and int((a["radii"] > 0).sum()) > 0
    for k in ("color", "radii", "records", "clamped", "rects", "order", "depth", "acc", "flow", "idx"):
        assert torch.equal(a[k].view(torch.uint8) if a[k].dtype == torch.float32 else a[k], b[k].view(torch.uint8) if b[k].dtype == torch.float32 else b[k]), k
    if prepare:
        # the SH direction sums the forward leaves for the backward (36 B per visible Gaussian): the last array of the geometry buffer
        import ctypes
        n = ins["means3D"].shape[0]
        lay = _C.GeomLayout(); _C.load().ex4d_geom_layout(n, ctypes.byref(lay))
        off = lay.total - ((36 * n + 255) // 256) * 256
        vis = a["radii"] > 0
        da, db = (x["geom"][off: off + 36 * n].view(torch.int32).view(n, 9)[vis] for x in (a, b))
        assert torch.equal(da, db), "SH direction sums differ"
