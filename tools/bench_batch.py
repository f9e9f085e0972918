#!/usr/bin/env python3
"""Times the batched small-set kernel: 4096 independent 256-slot sets (240-256 members each, about
5 % of the rows all-zero), 20 iterations, Fr-exact.  Prints one JSON line.  Timing only: the checked run with
the CPU figure beside it is `bench.py --workload c5 --c5-sets S --c5-slots N`.
Edge-iterations count the dense edges the reference would iterate over: n_valid*(n_valid-1) per set."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_batch(n_sets, n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    values = rng.integers(1, 256, (n_sets, n, n), dtype=np.uint32)
    member = np.ones((n_sets, n), dtype=np.uint8)
    n_members = rng.integers(n - n // 16, n + 1, n_sets)
    for s in range(n_sets):
        member[s, n_members[s]:] = 0
    values[rng.random((n_sets, n)) < 0.05] = 0
    return values, member


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=4096)
    ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import protocol_amd as pa
    torch.cuda.set_device(0)
    values, member = make_batch(args.sets, args.slots, 42 + 5)
    t0 = time.time()
    b = pa.Batch(values, member)
    t_create = time.time() - t0
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        b.run_fr(20, 1000, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.run_fr(20, 1000, st)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    out, status = b.result_fr(st)
    print(json.dumps({"workload": "c5: %d sets x %d slots, 20 iterations, Fr-exact" % (args.sets, args.slots),
                      "ms_per_batch": el * 1e3, "sets_per_s": args.sets / el,
                      "edge_iterations_per_s": b.edges * 20 / el, "all_status_ok": bool((status == 0).all()),
                      "create_s": t_create}))


if __name__ == "__main__":
    main()
