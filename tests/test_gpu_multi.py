"""The row-sharded driver behind the C ABI (et_sharded_*): one host thread, several shares.
On the one-GPU box every share sits on device 0 and the exchange runs as device copies ordered by events; the
RCCL path is exercised with the one rank a single GPU allows.  Results must equal the CPU checker's whatever the
number of shares: bit for bit in the field, <= 1e-9 relative L1 in f64."""
import numpy as np
import pytest

import c_oracle
import protocol_amd as pa
from protocol_amd import _ffi

pytestmark = pytest.mark.gpu


def _graph(n, seed):
    frm, to, val = pa.synth_graph(3, n, 16 * n, seed=seed, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    member[::97] = 0
    return frm, to, val, member


def _rel_l1(a, b):
    a = np.asarray(a, dtype=np.longdouble)
    b = np.asarray(b, dtype=np.longdouble)
    return float(np.abs(a - b).sum() / np.abs(b).sum())


@pytest.mark.parametrize("world,n", [(1, 30000), (2, 50000), (3, 50000), (8, 60000), (2, 300000), (4, 3000)])
def test_shares_on_one_device_match_oracle(world, n):
    frm, to, val, member = _graph(n, 5 + world)
    iters = 20
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, eres, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                             n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0] * world)
    assert mg.exchange == _ffi.ET_EXCHANGE_COPY
    assert sum(int(mg.info(r).nnz_local) for r in range(world)) == nnz
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)
    o64, res = mg.converge_f64(iters, 1000.0)
    assert _rel_l1(o64, eld) <= 1e-9
    assert abs(res - eres) <= 1e-6 * max(1.0, abs(eres))
    # a second converge on the same object, and one of zero iterations (native.rs:317: the initial scores)
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)
    assert pa.fr_ints_from_bytes(mg.converge_fr(0, 1000)) == [1000 if m else 0 for m in member]


def test_damping_and_pretrust_over_shares():
    n = 20000
    frm, to, val, member = _graph(n, 77)
    rng = np.random.default_rng(3)
    p = rng.random(n) * 2000.0 * member
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0, 0, 0], variants=_ffi.ET_VARIANT_F64)
    for pre in (None, p):
        rc, _e, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0, alpha=0.15,
                                                               pretrust=pre if pre is not None else 1000.0 * member, want_ld=True)
        assert rc == 0
        o64, _ = mg.converge_f64(20, 1000.0, alpha=0.15, pretrust=pre)
        assert _rel_l1(o64, eld) <= 1e-9
    with pytest.raises(pa.EigenTrustError) as ei:
        mg.converge_fr(20, 1000)                        # built without the exact variant
    assert ei.value.code == _ffi.ET_E_UNSUPPORTED


def test_rccl_exchange_with_the_one_rank_a_single_gpu_allows():
    n = 40000
    frm, to, val, member = _graph(n, 9)
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=5, initial=1000)
    assert rc == 0
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0], exchange=_ffi.ET_EXCHANGE_RCCL, variants=1)
    assert mg.exchange == _ffi.ET_EXCHANGE_RCCL
    assert np.array_equal(mg.converge_fr(5, 1000), exp)
    # two ranks of one communicator cannot share a device
    with pytest.raises(pa.EigenTrustError) as ei:
        pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0, 0], exchange=_ffi.ET_EXCHANGE_RCCL)
    assert ei.value.code == _ffi.ET_E_RCCL
    with pytest.raises(pa.EigenTrustError) as ei:
        pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0, 99])
    assert ei.value.code == _ffi.ET_E_OUT_OF_RANGE
