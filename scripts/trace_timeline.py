"""One hipGraph replay of the clip as the GPU ran it: from a `rocprofv3 --kernel-trace --output-format csv` of `bench.py --mode infer`
(graph replay, the three MRF chains on their streams), take the LAST complete replay (the kernels between two long idle gaps) and print
its wall span, the union of the kernel intervals (time with at least one kernel running), the idle remainder, the sum of kernel
durations (> span where launches overlap), and the same per section — front (everything before the generator's first MRF conv),
then each MRF stage (split at the ConvTranspose launches).  usage: trace_timeline.py <dir with *kernel_trace.csv>"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# replays are separated by host-side gaps; take the last run of kernels with no gap > 200 us whose length looks like a clip
runs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(x[1] for x in cur[-8:]) > 200_000:
        runs.append(cur)
        cur = []
    cur.append(b)
runs.append(cur)
runs = [r for r in runs if len(r) > 100]
run = runs[-2] if len(runs) > 1 else runs[-1]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def report(name, ks):
    if not ks:
        return
    span = max(k[1] for k in ks) - min(k[0] for k in ks)
    u = union([(k[0], k[1]) for k in ks])
    ssum = sum(k[1] - k[0] for k in ks)
    print(f"{name:34s} launches {len(ks):4d}  span {span / 1e3:8.1f} us  busy(union) {u / 1e3:8.1f}  idle {(span - u) / 1e3:7.1f}  "
          f"sum of durations {ssum / 1e3:8.1f}  overlap factor {ssum / max(u, 1):.2f}")


print(f"# {len(runs)} replays found; analysing one with {len(run)} kernels")
report("whole replay", run)
is_up = lambda n: "convt" in n.lower() or ("conv1d_mfma_direct_kernel<2, 0, 2>" in n) or ("conv1d_mfma_kernel" in n and False)
# sections: split at the transposed-conv (ups) launches, recognised by the phases-as-rows instantiations' names if present, else by
# the first strip / tiled launch
names = [k[2] for k in run]
first_gen = next((i for i, n in enumerate(names) if "strip" in n or "conv1d_mfma_kernel" in n or "conv1d_h_kernel" in n or "respair" in n
                  or "conv1d_hl_kernel" in n), len(run))
report("front (encoder, flow, source ...)", run[:first_gen])
report("generator (from first MRF conv)", run[first_gen:])
# gaps > 2 us inside the replay
ivs = sorted((k[0], k[1]) for k in run)
gaps, ce = [], ivs[0][1]
for s, e in ivs[1:]:
    if s - ce > 2000:
        gaps.append((s - ce) / 1e3)
    ce = max(ce, e)
print(f"idle gaps > 2 us: {len(gaps)}, total {sum(gaps):.1f} us, largest {max(gaps) if gaps else 0:.1f} us")
