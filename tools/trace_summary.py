#!/usr/bin/env python3
"""Per-kernel durations of a `rocprofv3 --kernel-trace` run, split by launch shape.

bench.py's default line runs several workloads through the same kernels (the 1M-peer graph, the 64K-peer graph,
4096 small sets), so rocprofv3's own per-kernel average mixes them; this groups launches by shape and, inside one
shape, by duration cluster.

  python tools/trace_summary.py gpurun_out/<dir> [substring of the kernel names, default k_]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_"
groups = defaultdict(list)
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if want not in name:
                continue
            key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]))
            groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-60s %8s %7s %8s %7s %10s %10s %10s" % ("kernel", "blocks", "threads", "LDS", "calls", "avg_us", "min_us", "max_us"))
for (name, blocks, threads, lds), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    # launches of different workloads with the same shape (the static LDS size and the grid do not tell a 1M-peer
    # graph from a 64K-peer one): cut the sorted durations wherever one is more than 1.5 times the one before
    d.sort()
    clusters, cur = [], [d[0]]
    for v in d[1:]:
        if v > 1.5 * cur[-1]:
            clusters.append(cur)
            cur = []
        cur.append(v)
    clusters.append(cur)
    for c in reversed(clusters):
        print("%-60s %8d %7d %8d %7d %10.2f %10.2f %10.2f" % (name[:60], blocks, threads, lds, len(c), sum(c) / len(c) / 1e3, min(c) / 1e3, max(c) / 1e3))
