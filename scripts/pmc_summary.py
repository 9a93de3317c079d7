"""Summarise rocprofv3 --pmc passes (one counter per pass, CSV output) into per-kernel-family HBM traffic.

usage: python scripts/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <steps> [out.json]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE
tallies 128-byte read requests at 64 B, so wide coalesced streaming reads are under-counted by exactly 2x -> doubled
here; WRITE_SIZE is uncalibrated on gfx950 and is reported raw (flagged).  Infinity-cache hits are counted, so this is
L2<->fabric traffic, an upper bound on HBM traffic."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"\s*\[clone.*$", "", name)
    if "conv1d_mfma" in name or "conv1d_strip" in name:      # one family in bench.py's roofline: every dense conv launch
        return "conv1d_mfma"
    m = re.match(r"(?:void\s+)?(?:\(anonymous namespace\)::)?([A-Za-z_0-9:]+)", name)
    return (m.group(1) if m else name).split("::")[-1].replace("_kernel", "")


def load(d, counter):
    per = defaultdict(lambda: [0.0, set()])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                per[k][0] += float(row["Counter_Value"])
                per[k][1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return {k: (v[0], len(v[1])) for k, v in per.items()}, files


def main():
    fdir, wdir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else None
    fetch, ff = load(fdir, "FETCH_SIZE")
    write, wf = load(wdir, "WRITE_SIZE")
    fams = defaultdict(lambda: dict(fetch_kib=0.0, write_kib=0.0, launches=0))
    for k, (v, n) in fetch.items():
        fams[family(k)]["fetch_kib"] += v
        fams[family(k)]["launches"] += n
    for k, (v, n) in write.items():
        fams[family(k)]["write_kib"] += v
    rows = []
    for f, d in sorted(fams.items(), key=lambda kv: -(kv[1]["fetch_kib"] + kv[1]["write_kib"])):
        rd = d["fetch_kib"] * 1024 * 2          # gfx950: x2
        wr = d["write_kib"] * 1024              # uncalibrated
        n = max(d["launches"], 1)
        rows.append(dict(family=f, launches=d["launches"], read_bytes_per_launch=rd / n, write_bytes_per_launch=wr / n,
                         hbm_bytes_per_launch=(rd + wr) / n, hbm_bytes_per_step=(rd + wr) / steps))
    print(f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {steps} steps; read = FETCH_SIZE KiB x1024 x2 "
          "(gfx950 128-B requests tallied at 64 B), write = WRITE_SIZE KiB x1024 (uncalibrated)")
    print(f"# files: {len(ff)} fetch, {len(wf)} write")
    print(f"{'family':28s} {'launches':>9s} {'read MB/launch':>15s} {'write MB/launch':>16s} {'MB/step':>10s}")
    for r in rows:
        print(f"{r['family']:28s} {r['launches']:9d} {r['read_bytes_per_launch'] / 1e6:15.3f} "
              f"{r['write_bytes_per_launch'] / 1e6:16.3f} {r['hbm_bytes_per_step'] / 1e6:10.2f}")
    if out:
        dom = next((r for r in rows if r["family"] == "conv1d_mfma"), None)
        import importlib.util
        spec = importlib.util.spec_from_file_location("svc_build", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                   "so-vits-svc_amd", "csrc", "build.py"))
        bld = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bld)
        json.dump(dict(hbm_bytes_per_launch=dom["hbm_bytes_per_launch"] if dom else None,
                       hbm_bytes_per_step=dom["hbm_bytes_per_step"] if dom else None, steps=steps,
                       csrc_sha=bld.source_hash(), families=rows,
                       note="read = FETCH_SIZE*1024*2 (gfx950 correction), write = WRITE_SIZE*1024 (uncalibrated); "
                            "separate --pmc passes"), open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
