"""ctypes binding of oracle/libet_oracle.so (the C restatement) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libet_oracle.so")
_lib = None
_max_threads = 1

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)


def build(force=False):
    src = os.path.join(_HERE, "et_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libet_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_dense_converge_fr.restype = C.c_int
        _lib.oracle_sparse_converge_fr.restype = C.c_int
        _lib.oracle_sparse_converge_f64.restype = C.c_int
        global _max_threads
        _max_threads = int(_lib.oracle_max_threads())     # before any call narrows the OpenMP team (a call with n_threads sets it)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def fr_binop(name, a, b):
    """a, b python ints -> python int, through the C field arithmetic."""
    out = (C.c_uint8 * 32)()
    fa = (C.c_uint8 * 32).from_buffer_copy(int(a).to_bytes(32, "little"))
    fb = (C.c_uint8 * 32).from_buffer_copy(int(b).to_bytes(32, "little"))
    rc = getattr(lib(), "oracle_fr_" + name)(fa, fb, out)
    assert rc == 0
    return int.from_bytes(bytes(out), "little")


def fr_inv(a):
    out = (C.c_uint8 * 32)()
    fa = (C.c_uint8 * 32).from_buffer_copy(int(a).to_bytes(32, "little"))
    assert lib().oracle_fr_inv(fa, out) == 0
    return int.from_bytes(bytes(out), "little")


def ints_to_fr_bytes(vals):
    buf = bytearray(32 * len(vals))
    for i, v in enumerate(vals):
        buf[32 * i:32 * i + 32] = int(v).to_bytes(32, "little")
    return np.frombuffer(bytes(buf), dtype=np.uint8).copy()


def fr_bytes_to_ints(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint8).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def dense_converge_fr(n, num_iter, initial, member, has_row, ops_rows):
    """ops_rows: n x n python ints.  Returns (rc, scores as ints, loop seconds)."""
    member = np.ascontiguousarray(member, dtype=np.uint8)
    has_row = np.ascontiguousarray(has_row, dtype=np.uint8)
    flat = [v for row in ops_rows for v in row]
    ops = ints_to_fr_bytes(flat)
    out = np.zeros(32 * n, dtype=np.uint8)
    secs = C.c_double(0)
    rc = lib().oracle_dense_converge_fr(C.c_uint32(n), C.c_uint32(num_iter),
                                        C.c_uint64(initial & (2**64 - 1)), C.c_uint64(initial >> 64),
                                        _p(member, u8p), _p(has_row, u8p), _p(ops, u8p),
                                        _p(out, u8p), C.byref(secs))
    return rc, fr_bytes_to_ints(out), secs.value


def sparse_converge_fr(n, member, frm, to, value_u32=None, value_fr=None, num_iter=20,
                       initial=1000, n_threads=1):
    """Returns (rc, out bytes [n,32] uint8, nnz, loop seconds)."""
    member = np.ascontiguousarray(member, dtype=np.uint8)
    frm = np.ascontiguousarray(frm, dtype=np.uint32)
    to = np.ascontiguousarray(to, dtype=np.uint32)
    if value_u32 is not None:
        value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
    if value_fr is not None:
        value_fr = np.ascontiguousarray(value_fr, dtype=np.uint8)
    out = np.zeros((n, 32), dtype=np.uint8)
    nnz = C.c_uint64(0)
    secs = C.c_double(0)
    rc = lib().oracle_sparse_converge_fr(C.c_uint32(n), _p(member, u8p), C.c_uint64(len(frm)),
                                         _p(frm, u32p), _p(to, u32p), _p(value_u32, u32p),
                                         _p(value_fr, u8p), C.c_uint32(num_iter),
                                         C.c_uint64(initial & (2**64 - 1)), C.c_uint64(initial >> 64),
                                         C.c_int(n_threads), _p(out, u8p), C.byref(nnz), C.byref(secs))
    return rc, out, nnz.value, secs.value


def sparse_converge_f64(n, member, frm, to, value_u32=None, value_fr=None, num_iter=20,
                        initial=1000.0, alpha=0.0, pretrust=None, n_threads=1, want_ld=False):
    """Returns (rc, out f64[n], out long double[n] or None, residual_l1, nnz, loop seconds)."""
    member = np.ascontiguousarray(member, dtype=np.uint8)
    frm = np.ascontiguousarray(frm, dtype=np.uint32)
    to = np.ascontiguousarray(to, dtype=np.uint32)
    if value_u32 is not None:
        value_u32 = np.ascontiguousarray(value_u32, dtype=np.uint32)
    if value_fr is not None:
        value_fr = np.ascontiguousarray(value_fr, dtype=np.uint8)
    if pretrust is not None:
        pretrust = np.ascontiguousarray(pretrust, dtype=np.float64)
    out = np.zeros(n, dtype=np.float64)
    out_ld = np.zeros(n, dtype=np.longdouble) if want_ld else None
    res = C.c_double(0)
    nnz = C.c_uint64(0)
    secs = C.c_double(0)
    rc = lib().oracle_sparse_converge_f64(C.c_uint32(n), _p(member, u8p), C.c_uint64(len(frm)),
                                          _p(frm, u32p), _p(to, u32p), _p(value_u32, u32p),
                                          _p(value_fr, u8p), C.c_uint32(num_iter), C.c_double(initial),
                                          C.c_double(alpha),
                                          _p(pretrust, C.POINTER(C.c_double)), C.c_int(n_threads),
                                          _p(out, C.POINTER(C.c_double)),
                                          out_ld.ctypes.data_as(C.c_void_p) if want_ld else None,
                                          C.byref(res), C.byref(nnz), C.byref(secs))
    return rc, out, out_ld, res.value, nnz.value, secs.value


def max_threads():
    """Threads the checker may use on this host: what OpenMP offered when the library was loaded (a converge called with
    n_threads = 1 narrows the team for later calls, so asking again would answer 1)."""
    lib()
    return _max_threads
