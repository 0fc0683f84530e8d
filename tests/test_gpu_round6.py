"""GPU tests of round 6: the tile sort at row-segment granularity (ex4d_rowsort.hip: point_list and the tile ranges straight from the
rects in depth order) against the duplication + pair sort of rounds 2-5 it replaces, behind both depth sorts."""
import pytest
import torch

from tests import helpers as h
from tests.test_gpu_round5 import _frame, _same, _squeezed

pytestmark = pytest.mark.gpu

SCENES = [("cfg2", 20000, 0), ("cfg3", 12000, 137), ("cfg5", 6000, 0), ("cfg1", None, 0), ("cfg2", 1, 0), ("cfg2", 2049, 0), ("cfg5", 40000, 0)]


@pytest.mark.parametrize("cfg,P,t", SCENES)
def test_row_segment_tile_sort_equals_the_pair_sort(hip_lib, cfg, P, t):
    """point_list / ranges / the image: bit-equal between the row-segment sort (default) and duplication + MSD pair sort, behind the MSD
    depth sort and behind the LSD depth sort (whose scan kernel hands the rects over in the 8-byte form)."""
    from ex4dgs_amd import _C
    assert _C.get_option("tile_sort_rows") == 1, "the row-segment sort is the library default"
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    ins = {k: v.cuda() for k, v in ins.items()}
    ref = _frame(ins, st, tile_sort_rows=0, depth_sort_msd=0)
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=2), ref, "rows behind the MSD depth sort")
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=0), ref, "rows behind the LSD depth sort")
    _same(_frame(ins, st, tile_sort_rows=0, depth_sort_msd=2), ref, "pairs behind the MSD depth sort")


def test_row_segment_tile_sort_giant_gaussians(hip_lib):
    """Rects of hundreds of tiles (scales x 30): blocks whose segments / instances exceed the LDS stage take the unstaged write path."""
    ins, st = h.scene_inputs("cfg2", P=6000)
    ins["scales"] = ins["scales"] * 30.0
    ins = {k: v.cuda() for k, v in ins.items()}
    ref = _frame(ins, st, tile_sort_rows=0, depth_sort_msd=0)
    assert ref["R"] > 200 * 6000 * 0.5, "the scene is meant to have rects of hundreds of tiles"
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=2), ref, "rows, giant rects")
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=0), ref, "rows, giant rects, LSD depth sort")


def test_row_segment_tile_sort_depth_ties(hip_lib):
    ins, st = _squeezed(30000, 6.0, 6.4, 9000)
    ref = _frame(ins, st, tile_sort_rows=0, depth_sort_msd=0)
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=2), ref, "rows, depth ties")


def test_clustered_scene_cfg3c_against_the_oracle(hip_lib):
    """Config 3c: 200 anisotropic clusters + two planes, scale log-std 1.2 -- tile lists ten times the mean, rects of
    hundreds of tiles (one covering the whole image), popular Gaussians.  Forward integers bit-exact, pixels and gradients at the bars."""
    from tests.test_gpu_parity import _fwd_bwd
    _fwd_bwd("cfg3c", P=20000, t=137)


def test_clustered_scene_tile_sorts_agree(hip_lib):
    ins, st = h.scene_inputs("cfg3c", P=60000, t=0)
    ins = {k: v.cuda() for k, v in ins.items()}
    ref = _frame(ins, st, tile_sort_rows=0, depth_sort_msd=0)
    _same(_frame(ins, st, tile_sort_rows=1, depth_sort_msd=2), ref, "rows, clustered scene")


def test_msd_depth_sort_above_two_million_gaussians(hip_lib):
    """beyond 2 M Gaussians the MSD depth sort's partition runs with 16 items per thread (72 KB of LDS), its histogram kernel
    likewise, and the bucket kernel with 512 threads at real bucket sizes -- MSD = LSD = a stable host sort there too; the row-segment
    tile sort behind both (forward only: the image and the lists)."""
    from tests.test_gpu_round5 import _host_depth_order
    ins, st = h.scene_inputs("cfg2", P=2_100_000)
    ins = {k: v.cuda() for k, v in ins.items()}
    lsd = _frame(ins, st, depth_sort_msd=0)
    msd = _frame(ins, st, depth_sort_msd=2)
    assert torch.equal(lsd["depth_order"], _host_depth_order(lsd)), "LSD depth order differs from the host sort"
    _same(msd, lsd, "msd vs lsd at 2.1 M")
    _same(_frame(ins, st, depth_sort_msd=2, tile_sort_rows=0), lsd, "msd + pair sort vs lsd at 2.1 M")


def test_ranking_by_lds_atomics_is_probed_and_equals_ballot_ranking(hip_lib):
    """The scatter kernels rank by the return value of an LDS atomic where the device hands the lanes of one ds_add_rtn their pre-op values
    in lane order (probed by the library on its first forward: option "rank_lds_atomics" = -1, the default) -- on gfx950 it does; the
    ballot ranking (option 0) gives the same lists bit for bit."""
    from ex4dgs_amd import _C
    assert _C.get_option("rank_lds_atomics") == -1, "probing is the library default"
    ins, st = h.scene_inputs("cfg3", P=30000, t=137)
    ins = {k: v.cuda() for k, v in ins.items()}
    auto = _frame(ins, st)
    assert _C.get_option("rank_lds_atomics_in_use") == 1, "the probe found the LDS atomics of this device out of lane order"
    try:
        ballots = _frame(ins, st, rank_lds_atomics=0)
        assert _C.get_option("rank_lds_atomics_in_use") == 0
        for mode in (0, 2):
            _same(_frame(ins, st, rank_lds_atomics=1, depth_sort_msd=mode, tile_sort_rows=0), ballots, f"forced LDS ranking, pair sort, depth sort {mode}")
    finally:
        _C.set_option("rank_lds_atomics", -1)
    _same(auto, ballots, "probed LDS ranking vs ballot ranking")


def test_read_back_on_a_side_stream_equals_the_in_stream_copy(hip_lib):
    """The synchronous forward's one read-back (instance / segment counts, frame flags) runs on a stream of its own behind an event
    (option "readback_side_stream" = 1, the default): same frame as with the copy on the caller's stream, also from a non-default
    stream and over frames of different sizes back to back (the pinned buffer and the side stream are reused)."""
    from ex4dgs_amd import _C
    assert _C.get_option("readback_side_stream") == 1, "the side-stream read-back is the library default"
    for cfg, P in (("cfg2", 20000), ("cfg3", 12000), ("cfg2", 1), ("cfg5", 6000)):
        ins, st = h.scene_inputs(cfg, P=P, t=0)
        ins = {k: v.cuda() for k, v in ins.items()}
        ref = _frame(ins, st, readback_side_stream=0)
        _same(_frame(ins, st, readback_side_stream=1), ref, f"side-stream read-back, {cfg} {P}")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            other = _frame(ins, st, readback_side_stream=1)
        torch.cuda.current_stream().wait_stream(s)
        _same(other, ref, f"side-stream read-back from a non-default stream, {cfg} {P}")


@pytest.mark.parametrize("cfg,P,t", SCENES + [("cfg3c", 30000, 0)])
def test_msd_depth_sort_digit_width_9_and_10_bits_agree(hip_lib, cfg, P, t):
    """The MSD depth sort cuts its top digit 9 bits wide up to 1.3 M Gaussians (511 visible buckets, 512-thread bucket workgroups) and
    10 bits beyond (option "depth_sort_msd_bits" = 0: by count; 9 / 10: forced): same order, same lists as the LSD sort -- also behind
    the pair sort with the fused tile scan (bucket bases over 512 or 1024 buckets) and with 256-thread bucket workgroups under the 9-bit
    digit (twice the Gaussians per bucket: more of them take the through-memory path)."""
    from ex4dgs_amd import _C
    assert _C.get_option("depth_sort_msd_bits") == 0
    ins, st = h.scene_inputs(cfg, P=P, t=t)
    ins = {k: v.cuda() for k, v in ins.items()}
    ref = _frame(ins, st, depth_sort_msd=0)
    for bits in (9, 10):
        _same(_frame(ins, st, depth_sort_msd=2, depth_sort_msd_bits=bits), ref, f"{bits}-bit digit, row-segment sort")
        _same(_frame(ins, st, depth_sort_msd=2, depth_sort_msd_bits=bits, tile_sort_rows=0), ref, f"{bits}-bit digit, pair sort with the fused tile scan")
        _same(_frame(ins, st, depth_sort_msd=2, depth_sort_msd_bits=bits, depth_sort_local_threads=256, depth_sort_local_cap=64), ref, f"{bits}-bit digit, small buckets in LDS only")


def test_msd_depth_sort_9_bit_digit_at_the_bench_size(hip_lib):
    """1.0 M Gaussians (BASELINE config 3, what bench.py times): the 9-bit digit is what the count selects, no bucket leaves the LDS path
    (the auto mode's watch word stays clear), the order equals the LSD sort's and a stable host sort."""
    from ex4dgs_amd import _C
    from tests.test_gpu_round5 import _host_depth_order
    ins, st = h.scene_inputs("cfg3", t=137)
    ins = {k: v.cuda() for k, v in ins.items()}
    lsd = _frame(ins, st, depth_sort_msd=0)
    _C.set_option("depth_sort_msd", 3)          # (resets the auto mode's counters)
    auto = _frame(ins, st)
    assert _C.get_option("depth_sort_trips") == 0 and _C.get_option("depth_sort_hold") == 0, "a bucket of the 9-bit digit exceeded the LDS capacity at 1.0 M"
    _frame(ins, st)
    assert _C.get_option("depth_sort_trips") == 0
    assert torch.equal(lsd["depth_order"], _host_depth_order(lsd))
    _same(auto, lsd, "auto (9-bit MSD) vs LSD at 1.0 M")
    _same(_frame(ins, st, depth_sort_msd=2, depth_sort_msd_bits=10), lsd, "10-bit MSD vs LSD at 1.0 M")
