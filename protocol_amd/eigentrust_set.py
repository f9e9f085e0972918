"""Host-side mirror of the reference's `EigenTrustSet`
(eigentrust-zk/src/circuits/dynamic_sets/native.rs:109-392): same method names, same argument
meaning, same failure conditions (the reference panics; here the C ABI's error code is raised as
EigenTrustError).  All state and all computation live behind the C ABI (et_set_*); this class is
the binding a caller in Python uses, nothing more.
"""
import ctypes as C
from fractions import Fraction

import numpy as np

from . import _ffi
from ._ffi import check, lib

# production constants, circuits/mod.rs:39-45
NUM_NEIGHBOURS = 4
NUM_ITERATIONS = 20
INITIAL_SCORE = 1000
MIN_PEER_COUNT = 2


def _fr(v):
    return _ffi.EtFr.from_buffer_copy(int(v).to_bytes(32, "little"))


class EigenTrustSet:
    """EigenTrustSet::new(domain), with the const generics as constructor arguments."""

    def __init__(self, domain=0, num_neighbours=NUM_NEIGHBOURS, num_iterations=NUM_ITERATIONS,
                 initial_score=INITIAL_SCORE):
        self.num_neighbours = num_neighbours
        self.num_iterations = num_iterations
        self.initial_score = initial_score
        h = C.c_void_p()
        d = _fr(domain)
        check(lib().et_set_new(num_neighbours, num_iterations, initial_score & (2**64 - 1),
                               initial_score >> 64, C.byref(d), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().et_set_free(self._h)
                self._h = None
        except Exception:
            pass

    def add_member(self, addr):
        a = _fr(addr)
        check(lib().et_set_add_member(self._h, C.byref(a)))

    def remove_member(self, addr):
        a = _fr(addr)
        check(lib().et_set_remove_member(self._h, C.byref(a)))

    def update_op(self, from_addr, scores):
        """`scores`: what Opinion::validate (opinion/native.rs:63-109) returns for this sender --
        one field element per slot, zero where the attestation is absent or invalid."""
        if len(scores) != self.num_neighbours:
            raise ValueError("an opinion has one score per slot")
        a = _fr(from_addr)
        buf = (C.c_uint8 * (32 * len(scores)))()
        for i, v in enumerate(scores):
            buf[32 * i:32 * i + 32] = int(v).to_bytes(32, "little")
        check(lib().et_set_update_op(self._h, C.byref(a), buf))

    def update_op_u32(self, from_slot, about_slots, values):
        about = np.ascontiguousarray(about_slots, dtype=np.uint32)
        vals = np.ascontiguousarray(values, dtype=np.uint32)
        check(lib().et_set_update_op_u32(self._h, from_slot, len(about), about.ctypes.data_as(C.c_void_p),
                                         vals.ctypes.data_as(C.c_void_p)))

    def members(self):
        n = self.num_neighbours
        addrs = (C.c_uint8 * (32 * n))()
        scores = (C.c_uint8 * (32 * n))()
        check(lib().et_set_members(self._h, addrs, scores))
        ab, sb = bytes(addrs), bytes(scores)
        return ([int.from_bytes(ab[32 * i:32 * i + 32], "little") for i in range(n)],
                [int.from_bytes(sb[32 * i:32 * i + 32], "little") for i in range(n)])

    def converge(self):
        """Field-exact scores for every slot (python ints < r), computed on the GPU."""
        n = self.num_neighbours
        out = (C.c_uint8 * (32 * n))()
        check(lib().et_set_converge(self._h, out))
        b = bytes(out)
        return [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(n)]

    def converge_f64(self, alpha=0.0, pretrust=None):
        n = self.num_neighbours
        out = np.zeros(n, dtype=np.float64)
        res = C.c_double(0)
        p = None
        if pretrust is not None:
            pretrust = np.ascontiguousarray(pretrust, dtype=np.float64)
            p = pretrust.ctypes.data_as(C.c_void_p)
        check(lib().et_set_converge_f64(self._h, alpha, p, out.ctypes.data_as(C.c_void_p), C.byref(res)))
        return out, res.value

    def converge_rational(self):
        """Exact fractions (fractions.Fraction) for every slot."""
        n = self.num_neighbours
        need = C.c_size_t(0)
        rc = lib().et_set_converge_rational(self._h, None, None, 0, C.byref(need))
        if rc not in (_ffi.ET_OK, _ffi.ET_E_BUFFER_TOO_SMALL):
            check(rc)
        stride = max(1, need.value)
        num = (C.c_uint8 * (stride * n))()
        den = (C.c_uint8 * (stride * n))()
        check(lib().et_set_converge_rational(self._h, num, den, stride, C.byref(need)))
        nb, db = bytes(num), bytes(den)
        return [Fraction(int.from_bytes(nb[i * stride:(i + 1) * stride], "little"),
                         int.from_bytes(db[i * stride:(i + 1) * stride], "little")) for i in range(n)]
