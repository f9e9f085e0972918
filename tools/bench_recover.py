#!/usr/bin/env python3
"""Times batched attester recovery (Poseidon hash + secp256k1 key recovery + Keccak address per signed
attestation): the GPU kernel (one lane per record, host<->device copies included) against the host
threads of the same library.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=262144)
    ap.add_argument("--distinct", type=int, default=2048)
    ap.add_argument("--host-sample", type=int, default=4096)
    args = ap.parse_args()
    import protocol_amd as pa
    from protocol_amd import _ffi
    rng = random.Random(1)
    n_order = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    base = [pa.sign_attestation(rng.randrange(1, n_order), bytes(rng.randrange(256) for _ in range(20)), rng.randrange(256),
                                rng.randrange(1, n_order)) for _ in range(args.distinct)]
    arr = (_ffi.EtAttestation * args.records)()
    cbase = [b._to_c() for b in base]
    for i in range(args.records):
        arr[i] = cbase[i % args.distinct]
    who = (C.c_uint8 * (20 * args.records))()
    lib = pa.lib()
    _ffi.check(lib.et_attestations_recover_gpu(arr, min(args.records, 1024), who, None, -1))   # warm-up: module load
    t0 = time.perf_counter()
    _ffi.check(lib.et_attestations_recover_gpu(arr, args.records, who, None, -1))
    t_gpu = time.perf_counter() - t0
    hs = min(args.host_sample, args.records)
    who_h = (C.c_uint8 * (20 * hs))()
    t0 = time.perf_counter()
    _ffi.check(lib.et_attestations_recover(arr, hs, who_h, None))
    t_host = time.perf_counter() - t0
    same = bytes(who[:20 * hs]) == bytes(who_h)
    print(json.dumps({"records": args.records, "gpu_s": t_gpu, "gpu_records_per_s": args.records / t_gpu,
                      "host_sample": hs, "host_threads": min(16, os.cpu_count() or 1), "host_records_per_s": hs / t_host,
                      "identical_on_sample": same}))


if __name__ == "__main__":
    main()
