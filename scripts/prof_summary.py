#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) into a text table:
   per kernel family (total / calls / avg) and per (kernel, grid, LDS) shape.  Usage: prof_summary.py run.db > out.txt"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("at::native") or "at::native" in n[:40]:
        m = re.search(r"(distribution_\w+|vectorized_elementwise_kernel|indexSelect\w+|CatArray\w+|\w+_kernel)", n)
        return "torch:" + (m.group(1) if m else n[:40])
    return re.sub(r"\(.*$", "", n)


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, duration from kernels").fetchall()
    tot = sum(r[6] for r in rows)
    fam, shp = {}, {}
    for name, gx, wx, lds, vg, sg, d in rows:
        k = short(name)
        f = fam.setdefault(k, [0, 0])
        f[0] += 1
        f[1] += d
        s = shp.setdefault((k, gx // max(wx, 1), wx, lds, vg, sg), [0, 0])
        s[0] += 1
        s[1] += d
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}\n# total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches\n")
    print("## per kernel\n%-58s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-58s %8d %12.3f %10.2f %6.2f%%" % (k[:58], c, d / 1e6, d / c / 1e3, 100 * d / tot))
    print("\n## per (kernel, workgroups, wg size, LDS bytes, vgpr, sgpr)  [top 60]\n%-50s %7s %5s %7s %5s %5s %7s %10s %10s" %
          ("kernel", "WGs", "wg", "lds", "vgpr", "sgpr", "calls", "avg_us", "total_ms"))
    for (k, g, wx, lds, vg, sg), (c, d) in sorted(shp.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%-50s %7d %5d %7d %5d %5d %7d %10.2f %10.3f" % (k[:50], g, wx, lds, vg, sg, c, d / c / 1e3, d / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
