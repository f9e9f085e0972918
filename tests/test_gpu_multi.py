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


def test_shares_large_enough_to_order_the_cited_entries_by_workgroup():
    """From a thousand rows per workgroup upwards a share asks for the other shares' entries grouped by the workgroup that
    cites them, and the f64 sweep's workgroups follow the exact sweep's: the same scores."""
    n = 1 << 20
    frm, to, val = pa.synth_graph(3, n, 12 * n, seed=29, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    member[::1013] = 0
    iters = 6
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                           n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    for world in (2, 3):
        mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0] * world)
        info = mg.info(0)
        assert int(info.gather) == 1 and int(info.exact_sweep_blocks) % int(info.f64_sweep_blocks) == 0
        assert int(info.rows_cap) // int(info.exact_sweep_blocks) >= 1024
        assert sum(int(mg.info(r).nnz_local) for r in range(world)) == nnz
        assert np.array_equal(mg.converge_fr(iters, 1000), exp)
        o64, _ = mg.converge_f64(iters, 1000.0)
        assert _rel_l1(o64, eld) <= 1e-9
        mg.close()


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


@pytest.mark.parametrize("world,n,chunks", [(2, 60000, 2), (4, 100000, 4), (8, 200000, 3), (3, 70000, 8)])
def test_chunked_exchange_matches_oracle(world, n, chunks):
    """Segments exchanged in pieces on a second stream per share while the iteration's launches run, launch q waiting
    for piece q only: the same scores as one exchange per iteration, whatever the number of pieces."""
    frm, to, val, member = _graph(n, 21 + world)
    iters = 12
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, eres, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0,
                                                             n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0] * world, n_chunks=chunks)
    info = mg.info(0)
    assert int(info.n_chunks) == chunks and int(info.exact_sweep_blocks) != 0
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)
    o64, res = mg.converge_f64(iters, 1000.0)
    assert _rel_l1(o64, eld) <= 1e-9
    assert abs(res - eres) <= 1e-6 * max(1.0, abs(eres))
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)            # again on the same object
    assert np.array_equal(mg.converge_fr(1, 1000),
                          c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=1, initial=1000)[1])
    assert pa.fr_ints_from_bytes(mg.converge_fr(0, 1000)) == [1000 if m else 0 for m in member]


def test_chunked_rccl_exchange_with_one_rank():
    """The chunked exchange over RCCL -- an all-gather per piece into the staging buffer, the strided copy into place,
    the launches waiting for their pieces -- with the one rank a single GPU allows."""
    n = 70000
    frm, to, val, member = _graph(n, 31)
    iters = 6
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, _e, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0, want_ld=True)
    assert rc == 0
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0], exchange=_ffi.ET_EXCHANGE_RCCL, n_chunks=3)
    assert mg.exchange == _ffi.ET_EXCHANGE_RCCL and int(mg.info(0).n_chunks) == 3
    assert np.array_equal(mg.converge_fr(iters, 1000), exp)
    o64, _ = mg.converge_f64(iters, 1000.0)
    assert _rel_l1(o64, eld) <= 1e-9


def test_chunks_fall_back_where_the_sweeps_do_not_apply():
    # fewer than 16K rows per share: the exact variant runs row per lane there and the exchange stays whole
    n = 20000
    frm, to, val, member = _graph(n, 3)
    mg = pa.MultiGpuGraph(n, frm, to, value_u32=val, member=member, devices=[0, 0], n_chunks=4)
    assert int(mg.info(0).n_chunks) == 1
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=5, initial=1000)
    assert rc == 0 and np.array_equal(mg.converge_fr(5, 1000), exp)


def test_one_graph_stepped_in_phases():
    """et_graph_step_* on a graph built for a chunked exchange runs an iteration as its launches in a row; stepping
    the phases one by one gives the same vector."""
    import torch
    n = 70000
    frm, to, val, member = _graph(n, 8)
    iters = 4
    rc, exp, _nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    rc, _e, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0, want_ld=True)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member, n_chunks=3)
    assert int(g.info.n_chunks) == 3
    assert np.array_equal(g.converge_fr(iters, 1000), exp)
    o64, _ = g.converge_f64(iters, 1000.0)
    assert _rel_l1(o64, eld) <= 1e-9
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    vec_len = int(g.info.vec_len)
    x = torch.full((vec_len, 4), -1, dtype=torch.int64, device=dev)
    y = torch.full((vec_len, 4), -1, dtype=torch.int64, device=dev)
    g.init_fr(x.data_ptr(), 1000, st)
    for it in range(iters):
        for q in range(3):
            g.step_fr(x.data_ptr(), y.data_ptr(), it, st, final=it + 1 == iters, phase=q)
        x, y = y, x
    out = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    g.unpermute_fr(x.data_ptr(), out.data_ptr(), st)
    torch.cuda.synchronize(dev)
    assert np.array_equal(out.cpu().numpy().view(np.uint8).reshape(n, 32), exp)
    for bad in (3, 0xFFFFFFFF):                          # a phase beyond the launches of an iteration is refused, never "all"
        with pytest.raises(pa.EigenTrustError) as ei:
            g.step_fr(x.data_ptr(), y.data_ptr(), 0, st, phase=bad)
        assert ei.value.code == _ffi.ET_E_OUT_OF_RANGE
