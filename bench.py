#!/usr/bin/env python3
"""Benchmark of the EigenTrust convergence hot path on MI355X.

A "step" is one whole converge on a device-resident graph: initial vector, NUM_ITERATIONS = 20
iteration launches (dynamic_sets/native.rs:319-329), result back in slot order.  Inputs are
synthetic (the reference never runs above a dozen peers) and already resident in HBM when the timed
region starts.

  N = 1   : 1 048 576 peers, 16 777 216 attestations, power-law out-degree (the configuration the
            metric is quoted on); f64 and Fr-exact variants are both timed, `value` is the f64 one.
            The same line carries the other single-GPU configurations under "configs": the 64K-peer
            graph, the batch of 4096 small sets and the latency of one converge of the reference's
            own small sets beside the dense CPU restatement.
  N > 1   : 16 777 216 peers, 268 435 456 attestations row-sharded over the N GPUs, one all-gather
            of the trust vector per iteration (torch.distributed, backend nccl = RCCL).  Started
            without a launcher, the script starts its own N ranks (before it touches the GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_ITER = 20
INITIAL = 1000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

WORKLOADS = {
    # name: (kind, peers, attestations, description)
    "c2": (2, 1 << 16, 1 << 20, "64K peers / 1M attestations, uniform out-degree 16"),
    "c3": (3, 1 << 20, 1 << 24, "1M peers / 16M attestations, power-law out-degree, 1% dangling"),
    "c4": (3, 1 << 24, 1 << 28, "16M peers / 256M attestations, power-law out-degree, 1% dangling"),
}
C5_SETS, C5_SLOTS = 4096, 256   # batched small sets (ZK witness generation), split over the GPUs without communication


def algorithmic_bytes(variant, nnz, rows):
    """Bytes one iteration launch must move when every array is touched once.
    f64: CSR of C^T with pre-normalised weights, nnz*(8+4) + rows*(4+8+8) -- the per-edge figure the
    metric's roofline is quoted on; the kernel's own edge records are smaller (5 B: slot + source
    offset, opinion byte).
    Fr: the kernel iterates the factored form (raw opinion + source per edge, s*inv(sum) per row),
    whose compulsory traffic is nnz*(4+4) + rows*(4+32*3); the weighted CSR form would be
    nnz*36 + rows*68 and is reported beside it."""
    if variant == "f64":
        return nnz * 12 + rows * 20
    return nnz * 8 + rows * 100


def own_bytes(variant, nnz, rows, value_bytes=1):
    """Compulsory bytes of the form the kernels iterate (every array touched once, each gathered entry counted once):
    per edge a 4-byte record (accumulator slot + source offset; one base position per 64 edges) and the opinion,
    per row the gathered entry, its old entry, its factor, its new entry and a group entry per 16 rows."""
    entry = 8 if variant == "f64" else 32
    return nnz * (4 + value_bytes) + nnz // 16 + rows * 4 * entry + rows // 4


def kernel_sources_sha():
    """Identifies the kernels a counter record was taken on: hash of the step kernels' sources and of the layout
    they run on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "protocol_amd", "csrc")
    for name in ("engine.hip", "fr_dev.hpp", "layout.hpp", "builder.cpp"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def batch_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "protocol_amd", "csrc")
    for name in ("batch.hip", "fr_dev.hpp"):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def batch_counter_record(kernel, sets, slots, ms_per_step):
    """Vector-ALU counters of the small-set kernel from its PMC passes (tools/c5_counters_record.py): valid for the kernel
    and batch shape it names, built from the sources it was taken on, while the batch time is within 10 %."""
    path = os.path.join(ROOT, "profiles", "c5_counters.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    if rec.get("kernel") != kernel or rec.get("batch_sha") != batch_sources_sha() or (rec.get("sets"), rec.get("slots")) != (sets, slots):
        return None
    if abs(rec.get("ms_per_step", 0.0) - ms_per_step) > 0.10 * ms_per_step:
        return None
    return rec


def counter_record(wl, kernel, launch_us):
    """Fabric traffic per launch from the PMC passes of this same workload (tools/pmc_passes.sh +
    tools/traffic_record.py; rocprofv3 cannot be nested inside the bench).  The record counts only for the
    kernel it names, built from the sources it was taken on, and only while the launch time measured now is
    within 10 % of the one measured then; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)
    k = rec.get("kernels", {}).get(kernel)
    if not k or rec.get("workload") != wl or rec.get("csrc_sha") != kernel_sources_sha():
        return None
    if abs(k.get("launch_us", 0.0) - launch_us) > 0.10 * launch_us:
        return None
    return k


def spawn_ranks(n):
    """--gpus N without a launcher: start N ranks with torch.distributed.run as a child process.  The parent
    has not touched the GPU, and it waits for the child instead of replacing itself."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def bench_small_sets(args, world, rank, device, red, steps, warmup, with_cpu):
    """4096 independent 256-slot sets (240-256 members each, ~5 % all-zero rows), 20 iterations, exact:
    every rank converges its own 4096 / world sets, one workgroup per set, no communication.
    Edge-iterations count the dense edges the reference iterates over: n_valid * (n_valid - 1) per set."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import protocol_amd as pa

    sets, slots = args.c5_sets, args.c5_slots      # defaults: the 4096 x 256 configuration
    lo, hi = sets * rank // world, sets * (rank + 1) // world
    # every rank draws only its own sets, each from a seed of its own
    values = np.empty((hi - lo, slots, slots), dtype=np.uint32)
    member = np.ones((hi - lo, slots), dtype=np.uint8)
    for s in range(lo, hi):
        rng = np.random.default_rng([args.seed + 5, s])
        v = rng.integers(1, 256, (slots, slots), dtype=np.uint32)
        v[rng.random(slots) < 0.05] = 0
        values[s - lo] = v
        member[s - lo, int(rng.integers(slots - slots // 16, slots + 1)):] = 0
    t0 = time.time()
    batch = pa.Batch(values, member)
    t_create = time.time() - t0
    st = torch.cuda.current_stream(device).cuda_stream

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(warmup):
        batch.run_fr(NUM_ITER, INITIAL, st)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run_fr(NUM_ITER, INITIAL, st)
    barrier()
    elapsed = time.perf_counter() - t0
    edges = float(batch.edges)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([edges], dtype=torch.float64, device=red)
        dist.all_reduce(e)
        elapsed, edges = float(t[0]), float(e[0])
    out, status = batch.result_fr(st)
    ok = bool((status == 0).all())
    cpu = None
    if rank == 0 and with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        sample = [0, (hi - lo) // 2, hi - lo - 1]
        cpu_edges = 0
        for s in sample:
            rows = [[int(v) for v in values[s, i]] for i in range(slots)]
            t1 = time.perf_counter()
            rc, exp, _ = c_oracle.dense_converge_fr(slots, NUM_ITER, INITIAL, member[s], member[s], rows)
            secs = time.perf_counter() - t1
            ok = ok and rc == 0 and pa.fr_ints_from_bytes(out[s]) == exp
            nv = int(member[s].sum())
            cpu_edges += nv * (nv - 1) * NUM_ITER / secs
        cpu = {"value": cpu_edges / len(sample), "unit": "edge-iterations/s", "cores": 1, "kind": "port",
               "sample": "3 of the %d sets, dense loop nest of the reference restated in C, one thread" % sets}
    ops = sets * NUM_ITER * steps * 2.0 * 32 * slots * slots / elapsed / 1e12
    kernel = "k_batch_fr_mfma_packed" if slots <= 32 else ("k_batch_fr_mfma" if slots <= 256 else "k_batch_fr_mfma_stream")
    # What bounds the kernel is the per-element field arithmetic on the vector ALUs: the roofline is the issue rate of
    # vector instructions, 1024 SIMDs x one wave-instruction per 4 cycles at 2.4 GHz (an MFMA holds a SIMD's issue for 8).
    # The instruction counts come from the PMC passes of this same batch shape; the launch time is measured here.
    rec = batch_counter_record(kernel, sets, slots, elapsed / steps * 1e3) if world == 1 else None
    valu_peak = 1024 * 2.4e9 / 4 / 1e9                       # G wave-instructions / s
    if rec:
        mfma = rec.get("SQ_INSTS_MFMA", 0.0)
        slots_used = (rec["SQ_INSTS_VALU"] - mfma) + 2.0 * mfma           # in 4-cycle issue slots
        valu_rate = slots_used / (elapsed / steps) / 1e9
        roof = {"bound": "valu", "achieved": valu_rate, "peak": valu_peak, "unit": "G wave-instructions/s (4-cycle issue slots)",
                "frac": valu_rate / valu_peak, "traffic": None, "kernel": kernel,
                "valu_instructions_per_launch": rec["SQ_INSTS_VALU"], "mfma_instructions_per_launch": mfma,
                # the same against the cycles the counters saw (the clock under this load is below 2.4 GHz)
                "frac_of_measured_cycles": slots_used * 4.0 / (1024 * rec["GRBM_GUI_ACTIVE"] / 8.0) if rec.get("GRBM_GUI_ACTIVE") else None,
                "mfma_i8_TOPs": ops, "mfma_frac_of_i8_peak": ops / 5000.0,
                "note": "bound by 32-bit vector-ALU field arithmetic; the byte-limb product runs on the matrix cores beside it"}
    else:
        roof = {"bound": "valu", "achieved": None, "peak": valu_peak, "unit": "G wave-instructions/s (4-cycle issue slots)", "frac": None,
                "traffic": None, "kernel": kernel, "mfma_i8_TOPs": ops, "mfma_frac_of_i8_peak": ops / 5000.0,
                "note": "no counter record for this kernel, batch shape and build (profiles/c5_counters.json)"}
    return {
        "metric": "edge-iterations/sec", "value": edges * NUM_ITER * steps / elapsed, "unit": "edge-iterations/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "fr254", "data": "synthetic",
        "config": {"workload": "c5: %d independent sets x %d slots, %d iterations, exact; sets split over the GPUs" % (sets, slots, NUM_ITER),
                   "sets": sets, "slots": slots, "iterations": NUM_ITER, "parallelism": "sets split x%d" % world,
                   "seed": args.seed},
        "sets_per_s": sets * steps / elapsed,
        "roofline": roof,
        "cpu_baseline": cpu,
        "checks": {"all_status_ok": ok, "sampled_sets_bit_exact": ok},
        "setup_s": {"create": t_create},
    }


def bench_set_latency(args):
    """The reference's own configuration: ONE converge of one small set through the set API
    (EigenTrustSet::converge, native.rs:286-337), wall time per call including every host step, for the
    production size (4 slots, circuits/mod.rs:39), the reference's test sizes (12 slots / 10 iterations,
    native.rs:410-413) and a 256-slot set; beside it the dense loop nest restated in C on one thread."""
    import numpy as np
    import protocol_amd as pa
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    rows = []
    for n, iters in ((4, 20), (12, 10), (256, 20)):
        rng = np.random.default_rng(args.seed + n)
        vals = rng.integers(1, 256, (n, n), dtype=np.uint32)
        s = pa.EigenTrustSet(42, n, iters, INITIAL)
        for i in range(n):
            s.add_member(1000 + i)
        about = np.arange(n, dtype=np.uint32)
        for i in range(n):
            s.update_op_u32(i, about, vals[i])
        first = time.perf_counter()
        got = s.converge()
        first = time.perf_counter() - first
        reps = 200 if n <= 12 else 50
        t0 = time.perf_counter()
        for _ in range(reps):
            s.converge()
        gpu = (time.perf_counter() - t0) / reps
        ones = np.ones(n, dtype=np.uint8)
        rc, exp, secs = c_oracle.dense_converge_fr(n, iters, INITIAL, ones, ones, [[int(v) for v in r] for r in vals])
        rows.append({"slots": n, "iterations": iters, "gpu_us_per_converge": gpu * 1e6, "gpu_first_call_us": first * 1e6,
                     "cpu_port_us_per_converge": secs * 1e6, "bit_exact": bool(rc == 0 and got == exp)})
    return {"workload": "c1: one EigenTrustSet::converge through the set API, wall time per call", "sets": rows,
            "cpu_baseline": {"kind": "port", "cores": 1, "sample": "dense loop nest of native.rs:286-337 restated in C, one call"}}


def bench_graph(args, wl, world, rank, device, red, steps, warmup, with_cpu, value_max=255, alpha=0.0, variant=None):
    """value_max: opinions are drawn uniformly from 1..value_max (255: the reference's u8 attestation value,
    eigentrust/src/attestation.rs:304; 2^32 - 1: the stress set for opinions that do not fit a byte).
    alpha: damping of the f64 variant with the initial scores as pre-trust (0: the reference's update)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import protocol_amd as pa
    from protocol_amd.sharded import ShardedGraph

    kind, n, target_edges, desc = WORKLOADS[wl]
    t0 = time.time()
    variant = variant or args.variant
    if world == 1:
        frm, to, val = pa.synth_graph(kind, n, target_edges, seed=args.seed, value_max=value_max)
    else:
        # every rank ingests the attestations of its own range of attesters only; the ranks agree on the layout and
        # send each edge to the rank that owns its destination row (no rank holds the whole list)
        frm, to, val = pa.synth_graph_range(kind, n, target_edges, n * rank // world, n * (rank + 1) // world,
                                            seed=args.seed, value_max=value_max)
    t_gen = time.time() - t0
    variants = {"both": 3, "fr": 1, "f64": 2}[variant]
    t0 = time.time()
    # the graph is converged over and over: worth the planner's extra time (build time is reported, not part of the metric)
    sg = ShardedGraph(n, frm, to, value_u32=val, device=device, variants=variants, n_chunks=args.chunks,
                      local_attestations=world > 1, **({"reused": True} if world == 1 else {}))
    t_build = time.time() - t0
    info = sg.graph.info
    nnz = int(info.nnz)
    if world > 1:
        del frm, to, val  # the host copy is not needed after upload

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # what a plain device-to-device copy reaches on this card (read + write bytes), beside the 8 TB/s peak
    copy_gbs = None
    if world == 1 and wl == "c3":
        src = torch.empty(1 << 27, dtype=torch.float64, device=device).fill_(1.0)
        dst = torch.empty_like(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dst.copy_(src)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize(device)
        copy_gbs = 5 * 2 * src.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst

    fr_kernel = sg.graph.exact_kernel_name()
    results = {}
    value_bytes = 1 if value_max <= 255 else 4
    for variant in (["f64", "fr"] if variant == "both" else [variant]):
        if variant == "f64":
            def run(rec):
                return sg.run_f64(NUM_ITER, float(INITIAL), alpha, record=rec)
        else:
            def run(rec):
                return sg.run_fr(NUM_ITER, INITIAL, record=rec)
        for _ in range(warmup):
            run(False)
        barrier()
        sg.graph.event_times()  # empty the event ring
        t0 = time.perf_counter()
        for _ in range(steps):
            run(True)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ev_ms, ev_pairs = sg.graph.event_times()
        launches = ev_pairs * NUM_ITER if world == 1 else ev_pairs / max(1, sg.n_chunks)
        # world == 1 brackets the 20 back-to-back launches of a step with one pair of events;
        # the sharded loop brackets every launch (an all-gather sits between launches; with a chunked exchange an
        # iteration is n_chunks launches, counted as one)
        launch_s = ev_ms * 1e-3 / max(1, launches)
        rows_local = int(info.n_valid) // world
        alg = algorithmic_bytes(variant, int(info.nnz_local), rows_local)
        achieved = alg / launch_s / 1e9
        kernel = ("k_step_f64_blocked<u8>" if value_bytes == 1 else "k_step_f64_blocked<u32>") if variant == "f64" else fr_kernel
        rec = counter_record(wl, kernel, launch_s * 1e6) if world == 1 and value_max == 255 and alpha == 0.0 else None
        own = own_bytes(variant, int(info.nnz_local), rows_local, value_bytes)
        sent, received = sg.exchange_bytes(variant)
        per_iteration_us = elapsed / steps / NUM_ITER * 1e6
        results[variant] = {
            "value": nnz * NUM_ITER * steps / elapsed,
            "ms_per_step": elapsed / steps * 1e3,
            # N > 1: what an iteration spends outside this rank's launches (the exchange and the waits for the slowest
            # rank), from the wall time of a step and the event time of rank 0's launches
            "iteration_us": {"total": per_iteration_us, "launches": launch_s * 1e6,
                             "exchange_and_waits": max(0.0, per_iteration_us - launch_s * 1e6)} if world > 1 else None,
            # N > 1: what one rank sends and receives between two iterations (a gather vector holds only the entries the
            # rank's rows cite; the all-gather of whole segments would receive the last figure)
            "exchange": {"gather_vector": bool(sg.gather), "bytes_sent_per_rank": sent, "bytes_received_per_rank": received,
                         "vector_entries": sg.vec_len, "all_peers_entries": sg.full_len,
                         "all_gather_bytes_received_per_rank": sg.seg_stride * (8 if variant == "f64" else 32) * (world - 1)} if world > 1 else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": rec["traffic_bytes"] if rec else None,
                         # the same launch priced two more ways: by the bytes the counters saw cross the fabric, and by the
                         # compulsory bytes of the form the kernel iterates (smaller records than the CSR figure above)
                         "frac_traffic": rec["traffic_bytes"] / launch_s / 1e9 / HBM_PEAK_GBS if rec else None,
                         "own_bytes_per_launch": own,
                         "frac_own_bytes": own / launch_s / 1e9 / HBM_PEAK_GBS,
                         "kernel": kernel,
                         "launch_us": launch_s * 1e6, "algorithmic_bytes_per_launch": alg,
                         "device_copy_GBs": copy_gbs},
        }
        if variant == "fr":
            r = results[variant]["roofline"]
            # share of the vector-ALU issue slots the exact kernel used in the counter passes: wave-instructions x 4
            # cycles (a SIMD issues one full-rate wave64 instruction per 4 cycles; the 64-bit multiply-adds take
            # longer, so this is a lower bound) over 1024 SIMDs x the cycles of the launch
            r["valu_issue_frac"] = (rec["SQ_INSTS_VALU"] * 4.0 / (1024 * rec["GRBM_GUI_ACTIVE"] / 8.0)
                                    if rec and "SQ_INSTS_VALU" in rec and "GRBM_GUI_ACTIVE" in rec else None)
            # the same with the measured issue cost of the kernel's instruction mix: 5.7 cycles for a 64-bit multiply-add
            # (tools/microbench/mad_vs_fma64.hip), 4 for the others; the multiply-adds are counted from the kernel's code
            # (8 per edge in the sweep, 145 per row)
            if r["valu_issue_frac"] is not None:
                mads = (8.0 * int(info.nnz_local) + 145.0 * rows_local) / 64.0
                r["valu_issue_frac_weighted"] = ((rec["SQ_INSTS_VALU"] - mads) * 4.0 + mads * 5.7) / (1024 * rec["GRBM_GUI_ACTIVE"] / 8.0)
            r["form"] = "factored (raw opinions, pre-scaled vector)"
            r["weighted_csr_bytes_per_launch"] = int(info.nnz_local) * 36 + rows_local * 68

    # parity of the timed configuration: conservation of the score sum (native.rs:331-334)
    checks = {}
    st = torch.cuda.current_stream(device).cuda_stream
    if "fr" in results:
        s, changed = sg.graph.stats_fr(NUM_ITER - 1, st)
        total = sg._sum_mod_r(s)
        checks["fr_sum_conserved"] = bool(total == INITIAL * int(info.n_valid) % pa.R_MODULUS)
    if "f64" in results:
        res, s64 = sg.graph.stats_f64(NUM_ITER - 1, st)
        if world > 1:
            t = torch.tensor([res, s64], dtype=torch.float64, device=red)
            dist.all_reduce(t)
            res, s64 = float(t[0]), float(t[1])
        checks["f64_sum_rel_err"] = abs(s64 - INITIAL * float(info.n_valid)) / (INITIAL * float(info.n_valid))
        checks["f64_residual_l1"] = res

    cpu = None
    if rank == 0 and world == 1 and with_cpu:
        # the CPU checker (a port: the reference is Rust and cannot be built in this image),
        # timed on this box's host cores over the same graph and the same 20 iterations
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        threads = c_oracle.max_threads()
        member = np.ones(n, dtype=np.uint8)
        cpu = {"unit": "edge-iterations/s", "cores": threads, "kind": "port",
               "sample": "%s, %d iterations, OpenMP over rows, iteration loop only; *_1_thread: same graph on one "
                         "thread (f64 %d iterations, exact variant 4 iterations)" % (wl, NUM_ITER, NUM_ITER)}
        if "f64" in results:
            rc, _o, _l, _r, cnnz, secs = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=NUM_ITER,
                                                                       initial=float(INITIAL), n_threads=threads)
            assert rc == 0 and cnnz == nnz
            cpu["value"] = cnnz * NUM_ITER / secs
        if "fr" in results:
            rc, _o, cnnz, secs = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=NUM_ITER,
                                                             initial=INITIAL, n_threads=threads)
            assert rc == 0 and cnnz == nnz
            cpu["fr_value"] = cnnz * NUM_ITER / secs
            if "value" not in cpu:
                cpu["value"] = cpu["fr_value"]
        # one thread beside it (the reference's converge is single-threaded); the exact variant on 4 iterations
        if "f64" in results:
            rc, _o, _l, _r, cnnz, secs = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=NUM_ITER,
                                                                       initial=float(INITIAL), n_threads=1)
            cpu["value_1_thread"] = cnnz * NUM_ITER / secs
        if "fr" in results and wl != "c4":
            rc, _o, cnnz, secs = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=4,
                                                             initial=INITIAL, n_threads=1)
            cpu["fr_value_1_thread"] = cnnz * 4 / secs

    head = results.get("f64") or results["fr"]
    line = {
        "metric": "edge-iterations/sec", "value": head["value"], "unit": "edge-iterations/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak" if world == 1 else "strong", "vs_baseline": None,
        "dtype": "f64" if "f64" in results else "fr254", "data": "synthetic",
        "config": {"workload": "%s: %s; %d iterations per step; opinions 1..%d kept raw in the edge records%s" % (
                       wl, desc, NUM_ITER, value_max, "; damping alpha = %g, pre-trust = the initial scores" % alpha if alpha else ""),
                   "peers": n, "edges": nnz, "iterations": NUM_ITER, "parallelism": "rows sharded x%d" % world,
                   "exchange_pieces": sg.n_chunks, "seed": args.seed},
        "roofline": head["roofline"],
        "cpu_baseline": cpu,
        "checks": checks,
        "setup_s": {"generate": t_gen, "build_and_upload": t_build},
    }
    if world > 1:
        line["iteration_us"] = head["iteration_us"]
        line["exchange"] = head["exchange"]
    if "fr" in results and "f64" in results:
        line["fr_exact"] = dict(results["fr"], dtype="fr254 (8x32-bit limbs)", unit="edge-iterations/s")
    sg.graph.close()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--variant", default="both", choices=["both", "f64", "fr"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="N = 1, workload auto: only the 1M-peer graph, not the other configurations")
    ap.add_argument("--chunks", type=int, default=1, help="N > 1: exchange a segment in this many pieces beside the iteration's launches (1 = one all-gather per iteration)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--value-max", type=int, default=255, help="graph workloads: opinions are drawn from 1..value_max (255: the reference's u8; 4294967295: the 32-bit stress set)")
    ap.add_argument("--alpha", type=float, default=0.0, help="graph workloads: damping of the f64 variant, pre-trust = the initial scores (0: the reference's update)")
    ap.add_argument("--c5-sets", type=int, default=C5_SETS, help="small-set workload: number of sets")
    ap.add_argument("--c5-slots", type=int, default=C5_SLOTS, help="small-set workload: slots per set (<= 1024)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))       # nothing above touched the GPU
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): ET_BENCH_BACKEND=gloo stages the exchange through host memory and lets several
    # ranks share a card (RCCL refuses two ranks of one communicator on one GPU); ET_BENCH_DEVICE pins every rank to
    # one card.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL.
    backend = os.environ.get("ET_BENCH_BACKEND", "nccl")
    if "ET_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["ET_BENCH_DEVICE"])
    elif backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())   # counting devices does not initialise the GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    # small reductions travel on the device with nccl, on the host in a gloo rehearsal
    red = device if backend == "nccl" else torch.device("cpu")
    with_cpu = not args.no_cpu_baseline

    wl = args.workload
    if wl == "auto":
        wl = "c3" if world == 1 else "c4"
    if wl == "c1":
        line = {"metric": "us per converge", "unit": "us", "higher_is_better": False, "n_gpus": 1, "data": "synthetic",
                "dtype": "fr254", "config": bench_set_latency(args)}
    elif wl == "c5":
        line = bench_small_sets(args, world, rank, device, red, args.steps, args.warmup, with_cpu)
    else:
        line = bench_graph(args, wl, world, rank, device, red, args.steps, args.warmup, with_cpu and args.value_max == 255,
                           value_max=args.value_max, alpha=args.alpha)
        if args.workload == "auto" and world == 1 and not args.no_extras:
            # the other single-GPU configurations, briefly, inside the same line
            saved = args.variant
            c2 = bench_graph(args, "c2", 1, 0, device, red, max(3, args.steps // 2), 2, with_cpu)
            args.variant = saved
            c5 = bench_small_sets(args, 1, 0, device, red, max(3, args.steps // 2), 2, with_cpu)
            few = max(3, args.steps // 2)
            # opinions that do not fit a byte (the reference's opinion is any field element, native.rs:128), the damped f64
            # update, and the 16M-peer graph on ONE GPU (the N = 1 point of the multi-GPU curve; no CPU leg: it takes minutes)
            u32 = bench_graph(args, "c3", 1, 0, device, red, few, 2, False, value_max=0xFFFFFFFF)
            damped = bench_graph(args, "c3", 1, 0, device, red, few, 2, False, alpha=0.15, variant="f64")
            big = bench_graph(args, "c4", 1, 0, device, red, 2, 1, False)
            keep = ("value", "unit", "ms_per_step", "roofline", "fr_exact", "cpu_baseline", "checks", "sets_per_s", "dtype", "setup_s")
            pick = lambda r: dict({k: r[k] for k in keep if k in r}, workload=r["config"]["workload"])
            line["configs"] = {
                "c2": pick(c2), "c5": pick(c5), "c1": bench_set_latency(args),
                "c3_u32": pick(u32), "c3_alpha015": pick(damped), "c4_one_gpu": pick(big),
            }
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
