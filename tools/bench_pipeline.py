#!/usr/bin/env python3
"""Times calculate_scores for large signed attestation lists (et_scores_from_attestations): attester recovery on
the GPU, participant set and slot lookups on the host threads, graph build, converge.  Signs --records records
with the library's own signer (slow, Python loop), replicates them --rep times and reports both sizes.

  PYTHONPATH=. python tools/bench_pipeline.py --records 300000 --rep 8
"""
import argparse
import random, time, ctypes as C
import numpy as np
import protocol_amd as pa
from protocol_amd import _ffi
from protocol_amd.scores import _c_attestations

ap = argparse.ArgumentParser()
ap.add_argument('--records', type=int, default=300000)
ap.add_argument('--keys', type=int, default=20000)
ap.add_argument('--rep', type=int, default=8)
args = ap.parse_args()
rng = random.Random(7)
N_KEYS, N_ATT = args.keys, args.records
t0 = time.time()
secrets = [rng.randrange(1, 2**200) for _ in range(N_KEYS)]
addrs = [pa.public_key_of_secret(s)[1] for s in secrets]
atts = []
for k in range(N_ATT):
    i = rng.randrange(N_KEYS)
    j = rng.randrange(N_KEYS)
    atts.append(pa.sign_attestation(secrets[i], addrs[j], rng.randrange(1, 256), rng.randrange(1, 2**250)))
print("generated", N_ATT, "in %.1f s" % (time.time() - t0), flush=True)
arr = _c_attestations(atts)
# replicate to a bigger list (same signatures, still valid)
REP = args.rep
big = (_ffi.EtAttestation * (N_ATT * REP))()
C.memmove(big, arr, C.sizeof(arr))
for r in range(1, REP):
    C.memmove(C.byref(big, r * C.sizeof(arr)), arr, C.sizeof(arr))
for a, name in ((arr, "%d records" % N_ATT), (big, "%d records" % (N_ATT * REP))):
    for _ in range(2):
        t0 = time.perf_counter()
        who = pa.recover_attesters(a, gpu=True)
        t1 = time.perf_counter()
        out = pa.scores_from_attestations(a)
        t2 = time.perf_counter()
    print(name, "recover %.3f s (incl. python list)  scores_from_attestations %.3f s  participants %d" % (t1 - t0, t2 - t1, len(out[0])), flush=True)
