#!/usr/bin/env python
"""profiles/<tag>_pmc_FETCH_SIZE.txt + <tag>_pmc_WRITE_SIZE.txt (tools/pmc.sh, separate passes) -> profiles/r01_pmc_traffic.json.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 B
(MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is in KB."""
import json, re, sys
fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]


def parse(path):
    d = {}
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([0-9.e+]+)\s*$", line)
        if m:
            name = m.group(1)
            key = re.search(r"(\w+_kernel|\w+)(<[^>]*>)?\(", name)
            d[(key.group(1) + (key.group(2) or "")) if key else name] = float(m.group(3))
    return d


f, w = parse(fetch), parse(write)
alias = {"composite_bwd_kernel<4>": "composite_bwd", "composite_fwd_kernel<4>": "composite_fwd", "preprocess_fwd_kernel": "preprocess_fwd",
         "preprocess_bwd_kernel": "preprocess_bwd", "duplicate_kernel": "duplicate", "rs_scatter_kernel<16>": "tile_sort_scatter_pass",
         "rs_scatter_kernel<4>": "depth_sort_scatter_pass", "rs_histogram_kernel<16>": "tile_sort_histogram_pass",
         "tile_ranges_kernel": "tile_ranges", "scan_tiles_local_kernel": "scan_tiles"}
kernels = {}
for k in sorted(set(f) | set(w)):
    if k in alias:
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        kernels[alias[k]] = {"FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): {fetch}, {write}; bench.py cfg3 1.0M Gaussians",
           "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE tallies 128-B requests at 64 B)",
           "kernels": kernels}, open(out, "w"), indent=1)
print(json.dumps(kernels, indent=1))
