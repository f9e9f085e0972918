#!/usr/bin/env python3
"""Times the step kernels of ONE rank's share of a row-sharded graph on one GPU (no exchange).

The vector a share reads spans all peers whatever the number of ranks, so a share's launch does not
shrink in proportion to its rows; this prints what a rank of an N-way run computes per iteration.

  python tools/bench_share.py --workload c4 --ranks 1 2 4 8
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
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--chunks", type=int, default=1, help="pieces of a chunked exchange: an iteration = that many launches")
    ap.add_argument("--full-vector", action="store_true", help="every share's whole segment in the vectors (all-gather exchange) instead of the cited entries")
    args = ap.parse_args()

    import torch
    import protocol_amd as pa
    from protocol_amd.graph import Graph
    import bench

    kind, n, target_edges, _desc = bench.WORKLOADS[args.workload]
    frm, to, val = pa.synth_graph(kind, n, target_edges, seed=42, value_max=255)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    for world in args.ranks:
        t0 = time.time()
        g = Graph(n, frm, to, value_u32=val, n_seg=world, seg=0, variants=3, device=0, n_chunks=args.chunks, full_vector=args.full_vector)
        t_build = time.time() - t0
        info = g.info
        vec_len = int(info.vec_len)
        ex = g.exchange()
        out = {"ranks": world, "rows": int(info.n_valid) // world, "edges": int(info.nnz_local), "build_s": round(t_build, 2),
               "chunks": int(info.n_chunks), "gather_vector": bool(ex.gather), "vector_entries": vec_len, "all_peers_entries": int(ex.full_len),
               "entries_received_per_iteration": sum(int(ex.recv_count[c]) for c in range(world)),
               "entries_sent_per_iteration": int(ex.send_total) if ex.gather else (world - 1) * int(ex.own_len)}
        for kind_, shape, dt in (("f64", (vec_len,), torch.float64), ("fr", (vec_len, 4), torch.int64)):
            x = torch.zeros(shape, dtype=dt, device=dev)
            y = torch.zeros(shape, dtype=dt, device=dev)
            init, step = (g.init_f64, g.step_f64) if kind_ == "f64" else (g.init_fr, g.step_fr)
            init(x.data_ptr(), 1000, st)
            # the other shares' entries: any non-zero values do, copies of this share's
            stride = int(info.rows_cap + info.nd_cap)
            for off in range(stride, vec_len, stride):
                ln = min(stride, vec_len - off)
                x[off:off + ln].copy_(x[:ln])
            y.copy_(x)
            bufs = (x, y)

            def launch(it):
                a, b = bufs[it & 1], bufs[(it + 1) & 1]
                if kind_ == "f64":
                    step(a.data_ptr(), b.data_ptr(), it, 0.0, None, st, stats=False)
                else:
                    step(a.data_ptr(), b.data_ptr(), it, st, stats=False)      # as every step of a converge but the last
            for it in range(3):
                launch(it)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(args.launches):
                launch(it)
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / args.launches
            out[kind_ + "_us"] = round(us, 1)
            out[kind_ + "_edges_per_s"] = float("%.3g" % (int(info.nnz_local) / (us * 1e-6)))
            del x, y
        import resource
        out["host_peak_rss_gib"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 1)
        print(json.dumps(out), flush=True)
        g.close()
        del g


if __name__ == "__main__":
    main()
