#!/usr/bin/env python
"""Turn a rocprofv3 rocpd SQLite database (--kernel-trace [--stats]) into a per-kernel text summary
(the `--stats` table: calls, total, average, min, max, %, plus VGPR/SGPR/LDS per kernel)."""
import sqlite3
import sys

NAME_W = 160      # full template argument lists: `rs_scatter_kernel<8, 1024, 3, 10>` and `<16, 256, 2, 7>` are different kernels


def main(db_path, out=sys.stdout):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
        "max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {db_path}", file=out)
    print(f"{'kernel':{NAME_W}s} {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>9s} {'wg':>4s}", file=out)
    for n, c, s, a, mn, mx, vg, ag, sg, lds, gx, wx in rows:
        print(f"{n[:NAME_W]:{NAME_W}s} {c:6d} {s / 1e3:11.1f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / total:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:6d} {gx or 0:9d} {wx or 0:4d}", file=out)


if __name__ == "__main__":
    main(sys.argv[1], open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
