"""MI355X-native EigenTrust convergence engine (HIP kernels behind a C ABI).

`EigenTrustSet` mirrors the reference's set API; `Graph` exposes the device-resident engine.
Everything computes on the GPU through protocol_amd/lib/libeigentrust_hip.so; there is no CPU path.
"""
from ._ffi import EigenTrustError, LIB_PATH, lib  # noqa: F401
from .eigentrust_set import (EigenTrustSet, INITIAL_SCORE, MIN_PEER_COUNT, NUM_ITERATIONS,  # noqa: F401
                             NUM_NEIGHBOURS)
from .graph import Batch, Graph, Plan, R_MODULUS, fr_bytes_from_ints, fr_ints_from_bytes, synth_graph  # noqa: F401
from .scores import (ETPublicInputs, Score, SignedAttestationRaw, Threshold, big_to_fe_rat, calculate_scores,  # noqa: F401
                     pack_score, read_attestations_csv, scalar_from_address, write_attestations_csv, write_scores_csv)
