"""The entry point that allocates no device memory (et_batch_converge_fr_noalloc) against the entry point it is a
variant of (et_batch_converge_fr): same scores and statuses bit for bit, on calls in a row that reuse the one
resident scratch array, each smaller than the one before."""
import ctypes as C

import numpy as np
import pytest

import protocol_amd as pa

pytestmark = pytest.mark.gpu


def _case(seed, n_sets, n, vmax):
    rng = np.random.default_rng(seed)
    values = rng.integers(0, vmax + 1, (n_sets, n, n), dtype=np.uint64).astype(np.uint32)
    member = (rng.random((n_sets, n)) > 0.06).astype(np.uint8)
    member[:, :2] = 1
    values[rng.random((n_sets, n)) < 0.05] = 0          # peers without opinions
    if n_sets > 2:
        member[2, :] = 0
        member[2, 0] = 1                                 # a single peer: ET_E_INSUFFICIENT_PEERS for that set
    return values, member


def _call(fn, values, member, iters=20, initial=1000):
    n_sets, n = values.shape[0], values.shape[1]
    out = np.full((n_sets, n, 32), 0xEE, dtype=np.uint8)
    status = np.full(n_sets, 77, dtype=np.int32)
    rc = fn(n_sets, n, member.ctypes.data_as(C.c_void_p), values.ctypes.data_as(C.c_void_p), iters,
            initial & (2**64 - 1), initial >> 64, out.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p))
    return rc, out, status


def test_no_alloc_entry_point_matches_the_allocating_one():
    lib = pa.lib()
    # a large call first, then smaller ones into the same array (which keeps the earlier calls' bytes): 256-slot
    # sets on the matrix cores, 12-slot sets packed five to a wavefront, opinions above a byte on the vector ALUs
    for seed, n_sets, n, vmax in ((1, 6, 256, 255), (2, 3, 12, 255), (3, 3, 100, 70000), (4, 1, 4, 255)):
        values, member = _case(seed, n_sets, n, vmax)
        rc0, out0, st0 = _call(lib.et_batch_converge_fr, values, member)
        rc1, out1, st1 = _call(lib.et_batch_converge_fr_noalloc, values, member)
        assert rc0 == 0 and rc1 == 0, (n_sets, n)
        assert np.array_equal(st0, st1), (n_sets, n)
        assert np.array_equal(out0, out1), (n_sets, n)
        assert (st0 == 0).sum() >= 1


def test_no_alloc_entry_point_refuses_what_does_not_fit():
    values, member = _case(5, 80, 256, 255)               # 80 x 64 KiB of opinion bytes > the 4 MiB array
    rc, _out, _st = _call(pa.lib().et_batch_converge_fr_noalloc, values, member)
    assert rc == pa._ffi.ET_E_BUFFER_TOO_SMALL
    assert b"scratch" in pa.lib().et_last_error_message()
    # and the array still serves the next call
    values, member = _case(6, 2, 64, 255)
    rc0, out0, st0 = _call(pa.lib().et_batch_converge_fr, values, member)
    rc1, out1, st1 = _call(pa.lib().et_batch_converge_fr_noalloc, values, member)
    assert rc0 == 0 and rc1 == 0 and np.array_equal(out0, out1) and np.array_equal(st0, st1)
