#!/usr/bin/env python
"""Per-kernel HBM traffic and roofline fractions from separate rocprofv3 --pmc passes (tools/prof_round.sh):

    pmc_traffic.py FETCH.txt WRITE.txt SQ_valu.txt kernel_stats.txt out.json [iter_FETCH iter_WRITE iter_stats knn_FETCH knn_WRITE knn_stats]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is in KB.  achieved = bytes / average launch duration of the
kernel-trace pass; frac = achieved / 8 TB/s.  valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs * duration * 2.4 GHz)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hashes():
    """sha256 (first 16 hex digits) of every kernel source: bench.py only quotes these counters while the sources are the ones
    they were collected from (a committed PMC file silently went stale the moment a kernel changed)."""
    d = os.path.join(ROOT, "ex4dgs_amd", "csrc")
    return {f: hashlib.sha256(open(os.path.join(d, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))}


def parse_table(path):
    lines = open(path).read().splitlines()
    hdr = lines[0].split()
    cols = hdr[2:]
    d = {}
    for line in lines[1:]:
        m = re.match(r"^(.*?)\s+(\d+)((?:\s+[0-9.e+-]+)+)\s*$", line)
        if m:
            vals = [float(x) for x in m.group(3).split()]
            d[m.group(1).strip()] = dict(zip(cols, vals))
    return d


def parse_stats(path, col=2):
    """{kernel name: column `col` of the stats table} (2 = avg_us, 0 = calls); the name column ends where the header's `calls` begins."""
    lines = open(path).read().splitlines()
    width = lines[1].index(" calls") - 1 if len(lines) > 1 and " calls" in lines[1] else 70
    d = {}
    for line in lines[2:]:
        name, rest = line[:width].strip(), line[width:].split()
        if len(rest) >= 3:
            d[name] = float(rest[col])
    return d


ALIAS = [("rows_seg_hist_kernel", "row_partition_histogram"), ("rows_scan_kernel", "row_partition_row_scan"), ("rows_seg_scatter_kernel", "row_partition_scatter"),
         ("ts_histogram_kernel", "tile_sort_histogram_pass_b"),
         ("composite_bwd_scan_kernel", "composite_bwd"), ("composite_bwd_kernel", "composite_bwd_per_pixel"), ("composite_fwd_kernel", "composite_fwd"),
         ("preprocess_geom_kernel", "preprocess_fwd"), ("preprocess_color_kernel", "preprocess_color"), ("preprocess_fwd_kernel", "preprocess_fwd"),
         ("preprocess_bwd_kernel", "preprocess_bwd"), ("duplicate_kernel", "duplicate"),
         ("depth_local_sort_kernel", "depth_sort_bucket_pass"), ("dls_histogram_kernel", "depth_sort_msd_histogram"), ("dls_range_kernel", "depth_sort_key_range"),
         ("rs_scatter_kernel<16", "tile_sort_scatter_pass"), ("rs_scatter_kernel<8", "depth_sort_scatter_pass"),
         ("rs_histogram_kernel<16", "tile_sort_histogram_pass"), ("rs_histogram_kernel<8", "depth_sort_histogram_pass"),
         ("rs_scan_rows_kernel", "radix_row_scan"), ("tile_ranges_kernel", "tile_ranges"), ("scan_tiles_local_kernel", "scan_tiles"),
         ("radam_kernel", "radam"), ("l1_ssim_fwd", "l1_ssim_forward"), ("l1_ssim_bwd", "l1_ssim_backward"),
         ("l1_ssim_finish", "l1_ssim_finish"), ("attributes_fwd", "attributes_forward"), ("attributes_bwd", "attributes_backward"),
         ("features_kernel", "attributes_sh_gather"), ("knn3_kernel", "knn3_search"), ("morton_kernel", "knn_morton"),
         ("gather_kernel", "knn_gather"), ("boxes_kernel", "knn_boxes")]


def alias(name):
    for pat, a in ALIAS:
        if pat in name:
            return a
    return None


def collect(fetch, write, stats, sq=None):
    f, w, st = parse_table(fetch), parse_table(write), parse_stats(stats)
    s = parse_table(sq) if sq else {}
    out = {}
    for name in sorted(set(f) | set(w)):
        a = alias(name)
        if a is None or a in out:
            continue
        fk, wk = f.get(name, {}).get("FETCH_SIZE", 0.0), w.get(name, {}).get("WRITE_SIZE", 0.0)
        us = st.get(name)          # (full names on both sides: exact match or nothing)
        row = {"kernel": name, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
        if us:
            row["avg_us"] = us
            row["hbm_GBps"] = round(row["hbm_bytes_per_launch"] / us / 1e3, 1)
            row["frac_of_8TBps"] = round(row["hbm_GBps"] / 8000.0, 3)
        if name in s and us:
            row["SQ_INSTS_VALU"] = s[name].get("SQ_INSTS_VALU")
            row["valu_busy_frac"] = round(4 * s[name].get("SQ_ACTIVE_INST_VALU", 0.0) / (1024 * us * 1e-6 * 2.4e9), 3)
        out[a] = row
    return out


RASTER_SOURCES = ("ex4d_preprocess.hip", "ex4d_binning.hip", "ex4d_rowsort.hip", "ex4d_composite.hip", "ex4d_api.hip", "ex4d_internal.h")


def frame_totals(fetch, write, stats, sq):
    """Everything ONE rasterizer frame (forward + backward) launches in the bench command, summed from the counter passes: every
    kernel (and runtime fill / copy) whose launch count is a multiple of the frame count, launches per frame x per-launch
    counters.  This is the frame's REAL HBM traffic and instruction count -- the figures a roofline of the frame is made of
    (the section-8(d) 'algorithmic bytes' price the REFERENCE's algorithm, 6-pass 64-bit sort included)."""
    f, w, st, s = parse_table(fetch), parse_table(write), parse_stats(stats), parse_table(sq)
    calls = {}
    for line in open(fetch).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)((?:\s+[0-9.e+-]+)+)\s*$", line)
        if m:
            calls[m.group(1).strip()] = int(m.group(2))
    n_fwd = next((n for k, n in calls.items() if "composite_fwd_kernel" in k), None)
    if not n_fwd:
        return None
    rows, tot_b, tot_us, tot_valu, launches = [], 0.0, 0.0, 0.0, 0
    for name, n in calls.items():
        ours = "anonymous namespace" in name or "__amd_rocclr_fillBuffer" in name or "__amd_rocclr_copyBuffer" in name
        per = int(round(n / n_fwd))
        if not ours or per < 1 or alias(name) in ("radam", "l1_ssim_forward", "l1_ssim_backward", "l1_ssim_finish", "attributes_forward", "attributes_backward"):
            continue
        b = (2 * f.get(name, {}).get("FETCH_SIZE", 0.0) + w.get(name, {}).get("WRITE_SIZE", 0.0)) * 1024
        us = st.get(name, 0.0)
        valu = s.get(name, {}).get("SQ_INSTS_VALU", 0.0)
        rows.append({"kernel": name, "launches_per_frame": per, "hbm_bytes_per_launch": int(b), "avg_us": us, "SQ_INSTS_VALU": valu})
        tot_b += per * b; tot_us += per * us; tot_valu += per * valu; launches += per
    # the same frame from the kernel-trace pass: every launch of the stats table, per forward call -- the two have to agree, or a row went
    # missing above (one instantiation of the scatter template had fallen out of the frame's rows)
    st_calls = parse_stats(stats, col=0)
    n_fwd_st = next((n for k, n in st_calls.items() if "composite_fwd_kernel" in k), None)
    stats_launches = sum(int(round(n / n_fwd_st)) for k, n in st_calls.items()
                         if ("anonymous namespace" in k or "__amd_rocclr_fillBuffer" in k or "__amd_rocclr_copyBuffer" in k) and int(round(n / n_fwd_st)) >= 1
                         and alias(k) not in ("radam", "l1_ssim_forward", "l1_ssim_backward", "l1_ssim_finish", "attributes_forward", "attributes_backward")) if n_fwd_st else None
    assert stats_launches is None or stats_launches == launches, f"frame rows: {launches} launches per frame from the counter pass, {stats_launches} from the kernel trace"
    missing = [r["kernel"] for r in rows if not r["avg_us"]]
    assert not missing, f"kernels of the counter pass without a duration in the kernel trace: {missing}"
    return {"launches_per_frame": launches, "hbm_bytes_per_frame": int(tot_b), "kernel_us_per_frame": round(tot_us, 2),
            "SQ_INSTS_VALU_per_frame": tot_valu, "frac_of_8TBps_at_kernel_sum": round(tot_b / (tot_us * 1e-6) / 8e12, 3) if tot_us else None,
            "rows": rows}


def main():
    a = sys.argv[1:]
    kernels = collect(a[0], a[1], a[3], a[2])
    frame = frame_totals(a[0], a[1], a[3], a[2])
    if len(a) >= 8:
        kernels.update({k: v for k, v in collect(a[5], a[6], a[7]).items() if k not in kernels})
    if len(a) >= 11:
        kernels.update({k: v for k, v in collect(a[8], a[9], a[10]).items() if k not in kernels})
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes, tools/prof_round.sh): bench.py cfg3 1.0M Gaussians; "
                         "tools/dev/dev_iter_profile.py (fused training iteration at 1.0M); tools/dev/dev_knn_time.py",
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
               "source_sha16": kernel_source_hashes(),
               "frame": frame,
               "kernels": kernels}, open(a[4], "w"), indent=1)
    if frame:
        print(f"frame: {frame['launches_per_frame']} launches, {frame['hbm_bytes_per_frame'] / 1e9:.3f} GB HBM traffic, {frame['kernel_us_per_frame']:.1f} us of kernels "
              f"-> {frame['frac_of_8TBps_at_kernel_sum']} of 8 TB/s; {frame['SQ_INSTS_VALU_per_frame']:.4g} VALU wave-instructions")
    for k, v in kernels.items():
        print(f"{k:28s} {v.get('avg_us', 0):9.2f} us  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {v.get('hbm_GBps', 0):8.1f} GB/s  frac {v.get('frac_of_8TBps', 0):.3f}  valu_busy {v.get('valu_busy_frac', '')}")


if __name__ == "__main__":
    main()
