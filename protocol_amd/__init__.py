"""MI355X-native EigenTrust convergence engine (HIP kernels behind a C ABI).

`EigenTrustSet` mirrors the reference's set API; `Graph` exposes the device-resident engine.
Everything computes on the GPU through protocol_amd/lib/libeigentrust_hip.so; there is no CPU path.
"""
from ._ffi import EigenTrustError, LIB_PATH, lib  # noqa: F401
from .eigentrust_set import (EigenTrustSet, INITIAL_SCORE, MIN_PEER_COUNT, NUM_ITERATIONS,  # noqa: F401
                             NUM_NEIGHBOURS)
from .graph import (Batch, DistributedPlan, Graph, MultiGpuGraph, Plan, R_MODULUS, fr_bytes_from_ints,  # noqa: F401
                    fr_ints_from_bytes, synth_graph, synth_graph_range)
from .scores import (ETPublicInputs, Score, SignedAttestationRaw, Threshold, attestation_hash, big_to_fe_rat,  # noqa: F401
                     calculate_scores, et_circuit_setup, keccak256, opinion_validate, pack_score, poseidon5_permute,
                     poseidon5_sponge, public_key_of_secret, read_attestations_csv, recover_attesters,
                     scalar_from_address, scores_from_attestations, sign_attestation, write_attestations_csv, write_scores_csv)
