#!/usr/bin/env python3
"""Host-only count for a row-sharded graph: how many distinct sources the rows of one share cite.

A share of an N-way run gathers only the entries of the trust vector that its own rows cite.  This prints,
for share 0 of N, the number of distinct sources per producing share, as a fraction of all peers, and how
many 128-byte lines of the vector those sources occupy in the global position order against a compact
order that keeps only the cited ones.  No GPU needed.

  python tools/share_sources.py --workload c4 --ranks 8
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4", choices=["c3", "c4"])
    ap.add_argument("--ranks", type=int, nargs="+", default=[8])
    args = ap.parse_args()

    import numpy as np
    import protocol_amd as pa
    from protocol_amd.graph import Plan
    import bench

    kind, n, target_edges, _desc = bench.WORKLOADS[args.workload]
    t0 = time.time()
    frm, to, val = pa.synth_graph(kind, n, target_edges, seed=42, value_max=255)
    print("generated %d attestations in %.1f s" % (len(frm), time.time() - t0), file=sys.stderr, flush=True)
    for world in args.ranks:
        t0 = time.time()
        plan = Plan(n, frm, to, value_u32=val, n_seg=world, seg=0)
        t_plan = time.time() - t0
        stride = int(plan.info.rows_cap + plan.info.nd_cap)
        pos = plan.pos_of_slot.astype(np.int64)
        valid = pos != 0xFFFFFFFF
        seg_of = np.where(valid, pos // stride, -1)
        # surviving edges of share 0: destination in segment 0, attester a member with opinions (self-edges dropped)
        keep = (seg_of[to] == 0) & valid[frm] & (frm != to)
        src = np.unique(frm[keep])
        src_pos = np.sort(pos[src])
        out = {"ranks": world, "peers": n, "plan_s": round(t_plan, 1), "edges_of_share": int(keep.sum()),
               "distinct_sources": int(len(src)), "fraction_of_peers": round(len(src) / n, 4),
               "per_producer": [int(((src_pos // stride) == p).sum()) for p in range(world)],
               "f64_lines_global_order": int(len(np.unique(src_pos // 16))), "f64_lines_compact": int((len(src) + 15) // 16),
               "exact_lines_global_order": int(len(np.unique(src_pos // 4))), "exact_lines_compact": int((len(src) + 3) // 4)}
        print(json.dumps(out), flush=True)
        del plan


if __name__ == "__main__":
    main()
