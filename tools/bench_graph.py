#!/usr/bin/env python3
"""Times the step kernels on one synthetic graph of any shape on one GPU.

  python tools/bench_graph.py --kind 2 --peers 1048576 --edges 16777216      # uniform out-degree
  python tools/bench_graph.py --kind 3 --peers 2097152 --edges 33554432      # power law

Prints one JSON line: microseconds per iteration launch of the f64 and of the exact variant (HIP events around the
launches of a 20-iteration converge, best of three), and how the plan's sweeps are laid out.
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
    ap.add_argument("--kind", type=int, default=3, choices=[2, 3])
    ap.add_argument("--peers", type=int, default=1 << 20)
    ap.add_argument("--edges", type=int, default=1 << 24)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--launches", type=int, default=20)
    args = ap.parse_args()

    import torch
    import protocol_amd as pa
    from protocol_amd.graph import Graph

    frm, to, val = pa.synth_graph(args.kind, args.peers, args.edges, seed=args.seed, value_max=255)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream(torch.device("cuda", 0)).cuda_stream
    t0 = time.time()
    g = Graph(args.peers, frm, to, value_u32=val, variants=3, device=0)
    out = {"kind": args.kind, "peers": args.peers, "edges": int(g.info.nnz), "build_s": round(time.time() - t0, 2)}

    def timed(run):
        g.event_times()                                  # empty the event ring
        for _ in range(5):
            run()                                        # one pair of events around the launches of a converge
        torch.cuda.synchronize()
        ms, pairs = g.event_times()
        return ms * 1e3 / (pairs * args.launches)

    g.run_f64(args.launches, 1000.0, 0.0, st)
    g.run_fr(args.launches, 1000, st)
    torch.cuda.synchronize()
    out["f64_us"] = round(timed(lambda: g.run_f64(args.launches, 1000.0, 0.0, st, record=True)), 2)
    out["fr_us"] = round(timed(lambda: g.run_fr(args.launches, 1000, st, record=True)), 2)
    out["exact_kernel"] = g.exact_kernel_name()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
