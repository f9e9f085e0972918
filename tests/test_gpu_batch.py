"""Batched small sets (one workgroup per set) against the dense CPU restatement, bit for bit."""
import numpy as np
import pytest

import c_oracle
import eigentrust_ref as ref
import protocol_amd as pa

pytestmark = pytest.mark.gpu


def dense_expect(values, member, iters, initial):
    n = values.shape[0]
    rows = [[int(v) for v in values[i]] for i in range(n)]
    return c_oracle.dense_converge_fr(n, iters, initial, member, member, rows)


@pytest.mark.parametrize("n,n_sets,vmax", [(4, 9, 255), (12, 5, 1000), (64, 6, 255), (256, 8, 255), (256, 3, 2**32 - 1),
                                          (300, 2, 255), (700, 1, 70000),
                                          # byte opinions: every tile count of the matrix-core kernel with resident operands
                                          # (n <= 256), then the one that streams them (n <= 1024)
                                          (33, 4, 3), (100, 3, 255), (130, 2, 200), (192, 2, 255), (200, 2, 255),
                                          # up to 32 slots: several sets share a wavefront (last group partly filled)
                                          (4, 37, 255), (12, 11, 255), (32, 5, 255), (7, 20, 255), (2, 70, 255),
                                          (400, 2, 255), (1000, 1, 255)])
def test_batch_matches_dense_restatement(n, n_sets, vmax):
    _check_batch(n, n_sets, vmax)


@pytest.mark.parametrize("n,n_sets,vmax", [(33, 4, 3), (130, 2, 200), (256, 2, 255), (400, 2, 255), (1000, 1, 255)])
def test_batch_vector_kernel_on_small_byte_sets(n, n_sets, vmax, monkeypatch):
    # the byte-valued vector kernel (uneven column split, matrix in LDS / left in global memory, >256 sources per
    # part), which byte sets reach only with the matrix cores switched off
    monkeypatch.setenv("ET_BATCH_NO_MFMA", "1")
    _check_batch(n, n_sets, vmax)


@pytest.mark.parametrize("n,iters", [(256, 20), (200, 20), (64, 1), (64, 0), (5, 3), (512, 20), (1024, 3), (700, 0)])
def test_batch_extreme_bytes(n, iters):
    # the largest column sums the byte-limb product can see (every opinion 255), and a 0/255 mix
    rng = np.random.default_rng(n + iters)
    values = np.full((2, n, n), 255, dtype=np.uint32)
    values[1] = rng.integers(0, 2, (n, n), dtype=np.uint32) * 255
    out, status = pa.Batch(values).converge_fr(iters, 1000)
    for s in range(2):
        rc, exp, _ = dense_expect(values[s], np.ones(n, dtype=np.uint8), iters, 1000)
        assert rc == 0 and status[s] == 0 and pa.fr_ints_from_bytes(out[s]) == exp, (n, s)


def _check_batch(n, n_sets, vmax):
    rng = np.random.default_rng(n * 7 + n_sets)
    values = rng.integers(0, vmax + 1, (n_sets, n, n), dtype=np.uint64).astype(np.uint32)
    member = (rng.random((n_sets, n)) > 0.06).astype(np.uint8)
    member[:, :2] = 1
    values[rng.random((n_sets, n)) < 0.05] = 0          # all-zero rows -> dangling peers
    values[0, 1, :] = 0
    if n_sets > 2:
        member[2, :] = 0
        member[2, 0] = 1                                 # a single peer: insufficient
    b = pa.Batch(values, member)
    out, status = b.converge_fr(20, 1000)
    for s in range(n_sets):
        rc, exp, _ = dense_expect(values[s], member[s], 20, 1000)
        if rc == -1:
            assert status[s] == pa._ffi.ET_E_INSUFFICIENT_PEERS
            continue
        assert rc == 0 and status[s] == 0
        assert pa.fr_ints_from_bytes(out[s]) == exp, (n, s)


def test_batch_reproduces_reference_test_matrices(scenarios):
    for name in ("test_scaling_1", "closed_graph_5", "threshold_4"):
        sc = next(s for s in scenarios if s["name"] == name)
        n = sc["num_neighbours"]
        rows = [[int(v, 16) for v in st[2]] for st in sc["steps"] if st[0] == "op"]
        values = np.array([rows], dtype=np.uint32)
        out, status = pa.Batch(values).converge_fr(sc["num_iterations"], sc["initial_score"])
        assert status[0] == 0
        assert [hex(v) for v in pa.fr_ints_from_bytes(out[0])] == sc["checkpoints"][0]["scores_fr"]


def test_batch_huge_initial_score():
    rng = np.random.default_rng(1)
    values = rng.integers(0, 256, (2, 8, 8), dtype=np.uint32)
    init = 2**127 + 12345
    out, status = pa.Batch(values).converge_fr(20, init)
    for s in range(2):
        rc, exp, _ = dense_expect(values[s], np.ones(8, dtype=np.uint8), 20, init)
        assert rc == 0 and status[s] == 0 and pa.fr_ints_from_bytes(out[s]) == exp
