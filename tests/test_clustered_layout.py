"""Positions clustered by consumer (cluster_cold_sources in csrc/builder.cpp): the plan of a graph large enough for it
must still be a permutation of the peers, keep the workgroups' edge counts level, and pull fewer 128-byte lines of the
trust vector per launch than the plain degree order.  CPU only: et_plan_build needs no GPU."""
import numpy as np
import pytest

import protocol_amd as pa

N, M = 1 << 19, 1 << 23          # 2048 rows per workgroup of a 256-workgroup sweep


@pytest.fixture(scope="module")
def graph():
    return pa.synth_graph(3, N, M, seed=11, value_max=255)


def _plan(graph, monkeypatch, affinity):
    frm, to, val = graph
    monkeypatch.setenv("ET_AFFINITY", "1" if affinity else "0")
    return pa.Plan(N, frm, to, value_u32=val)


def test_clustered_positions_are_a_permutation_with_level_workgroups_and_fewer_line_fills(graph, monkeypatch):
    plain = _plan(graph, monkeypatch, False)
    clustered = _plan(graph, monkeypatch, True)
    for variant in ("f64", "fr"):
        a, b = plain.sweep[variant], clustered.sweep[variant]
        assert a["blocks"] == b["blocks"] == 256
        assert a["clustered"] == 0 and b["clustered"] == 1
        assert a["edges"] == b["edges"] == int(plain.info.nnz)
        # a launch ends with its slowest workgroup: the deal must stay level
        assert b["edges_most"] <= 1.01 * b["edges"] / 256
        # fewer lines, for the busiest workgroup as for all of them
        assert b["line_fills"] < 0.92 * a["line_fills"], (variant, a, b)
        assert b["line_fills_most"] < 0.94 * a["line_fills_most"], (variant, a, b)
    # the same peers hold the same set of positions; only who sits where changes
    pa_, pb = plain.pos_of_slot, clustered.pos_of_slot
    assert np.array_equal(np.sort(pa_), np.sort(pb))
    members = pb != 0xFFFFFFFF
    assert len(np.unique(pb[members])) == int(members.sum()) == int(plain.info.n_valid)
    assert not np.array_equal(pa_, pb)
    # factors travel with the peers: the factor at a peer's position is the same number in both plans
    assert np.array_equal(plain.scale_f64[pa_[members]], clustered.scale_f64[pb[members]])


def test_clustering_is_asked_for_by_the_caller_who_reuses_the_graph(graph, monkeypatch):
    """Without a word from the caller a single share keeps the degree order (the plan is a third faster); ET_BUILD_REUSED
    asks for the clustered positions; the environment overrides both."""
    frm, to, val = graph
    monkeypatch.delenv("ET_AFFINITY", raising=False)
    assert pa.Plan(N, frm, to, value_u32=val).sweep["f64"]["clustered"] == 0
    assert pa.Plan(N, frm, to, value_u32=val, reused=True).sweep["f64"]["clustered"] == 1
    monkeypatch.setenv("ET_AFFINITY", "0")
    assert pa.Plan(N, frm, to, value_u32=val, reused=True).sweep["f64"]["clustered"] == 0


def test_small_graphs_and_several_shares_keep_the_degree_order(monkeypatch):
    frm, to, val = pa.synth_graph(3, 1 << 15, 1 << 19, seed=3, value_max=255)
    monkeypatch.setenv("ET_AFFINITY", "1")
    small = pa.Plan(1 << 15, frm, to, value_u32=val)
    assert small.sweep["f64"]["clustered"] == 0 and small.sweep["fr"]["clustered"] == 0
    frm, to, val = pa.synth_graph(3, 1 << 19, 1 << 22, seed=3, value_max=255)
    share = pa.Plan(1 << 19, frm, to, value_u32=val, n_seg=2, seg=1)
    assert share.sweep["f64"]["clustered"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(1 << 19, 1 << 23), (5 << 18, 3 << 22)])
def test_clustered_and_plain_positions_give_the_same_scores_on_the_gpu(monkeypatch, n, m):
    """Positions are internal: with the peers clustered by consumer or in plain degree order the exact variant returns
    the same bits (and the CPU checker's), the f64 variant the same scores to 1e-9 relative L1.  Empty slots, peers
    without opinions and repeated attestation pairs included.  The second size runs the exact sweep on twice the f64
    sweep's workgroups: every cluster of the planner is shared by two of them."""
    import c_oracle

    frm, to, val = pa.synth_graph(3, n, m, seed=5, value_max=255)
    rng = np.random.default_rng(5)
    member = (rng.random(n) > 0.02).astype(np.uint8)
    dup = rng.integers(0, len(frm), size=50000)          # later attestations for the same pair replace earlier ones
    frm = np.concatenate([frm, frm[dup]])
    to = np.concatenate([to, to[dup]])
    val = np.concatenate([val, rng.integers(0, 256, size=len(dup)).astype(np.uint32)])
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=20, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    outs = {}
    for affinity in ("0", "1"):
        monkeypatch.setenv("ET_AFFINITY", affinity)
        g = pa.Graph(n, frm, to, value_u32=val, member=member)
        assert g.info.nnz == nnz
        outs[affinity] = (g.converge_fr(20, 1000), g.converge_f64(20, 1000.0)[0])
        g.close()
    assert np.array_equal(outs["0"][0], exp)
    assert np.array_equal(outs["1"][0], exp)
    for affinity in ("0", "1"):
        a = np.asarray(outs[affinity][1], dtype=np.longdouble)
        assert float(np.abs(a - eld).sum() / np.abs(eld).sum()) <= 1e-9


def test_the_clustered_plan_does_not_depend_on_the_number_of_host_threads(graph, monkeypatch):
    """Votes are counted on all host threads, placement is sequential: one thread or many give the same positions."""
    monkeypatch.setenv("ET_BUILD_THREADS", "1")
    one = _plan(graph, monkeypatch, True)
    monkeypatch.setenv("ET_BUILD_THREADS", "7")
    many = _plan(graph, monkeypatch, True)
    assert np.array_equal(one.pos_of_slot, many.pos_of_slot)
    assert one.sweep == many.sweep


def test_two_shares_ask_for_the_cited_entries_grouped_by_workgroup():
    """Gather vectors of shares laid out from edge records (et_dplan_*): each share asks for the other share's entries in the
    order its workgroups want them.  Against the shares et_plan_build lays out from the whole list, which ask in ascending
    order: the same entries, the same geometry, level workgroups, fewer lines per launch -- and what one share says it needs
    is what the other is told to pack."""
    n, m, world = 1 << 20, 1 << 24, 2
    frm, to, val = pa.synth_graph(3, n, m, seed=5, value_max=255)
    plans = [pa.DistributedPlan(n, frm[frm % world == r], to[frm % world == r], val[frm % world == r]) for r in range(world)]
    stats = [p.local_stats() for p in plans]
    for r, p in enumerate(plans):
        p.set_global_stats(sum(s[0] for s in stats), sum(s[1] for s in stats), sum(s[2] for s in stats), sum(s[3] for s in stats),
                           max(s[4] for s in stats), world, r)
    for r, p in enumerate(plans):
        for q in plans:
            p.add_incoming(*q.outgoing(r))
    needs = {(r, q): p.needs(q).copy() for r, p in enumerate(plans) for q in range(world) if q != r}
    for (r, q), rows in needs.items():
        plans[q].add_wanted(r, rows)
    for r, p in enumerate(plans):
        got = p.finish_host()
        ref = pa.Plan(n, frm, to, value_u32=val, n_seg=world, seg=r)
        q = 1 - r
        assert got.exchange.gather == ref.exchange.gather == 1
        assert list(got.exchange.recv_off) == list(ref.exchange.recv_off) and list(got.exchange.recv_count) == list(ref.exchange.recv_count)
        assert list(got.exchange.send_count) == list(ref.exchange.send_count)
        assert np.array_equal(got.need[q], needs[(r, q)]) and np.array_equal(got.send[q], needs[(q, r)])
        # the same entries in another order
        assert not np.array_equal(got.need[q], ref.need[q]) and np.array_equal(np.sort(got.need[q]), ref.need[q])
        for variant in ("f64", "fr"):
            a, b = ref.sweep[variant], got.sweep[variant]
            assert a["blocks"] == b["blocks"] and a["edges"] == b["edges"]
            assert b["clustered"] == 1 and a["clustered"] == 0
            assert b["edges_most"] <= 1.01 * b["edges"] / b["blocks"]
            assert b["line_fills"] < 0.9 * a["line_fills"], (variant, a, b)


def test_sweep_stats_refuses_bad_arguments():
    import ctypes as C
    from protocol_amd import _ffi
    from protocol_amd.graph import lib, _ptr

    frm, to, val = pa.synth_graph(2, 1 << 10, 1 << 13, seed=1, value_max=255)
    opts = _ffi.EtBuildOpts(1, 0, 0, 0, 0, -1, 1)
    h = C.c_void_p()
    assert lib().et_plan_build(1 << 10, None, len(frm), _ptr(frm), _ptr(to), _ptr(val), None, C.byref(opts), C.byref(h)) == 0
    try:
        ss = _ffi.EtSweepStats()
        assert lib().et_plan_sweep_stats(h, 0, C.byref(ss)) == _ffi.ET_E_INVALID_ARG
        assert lib().et_plan_sweep_stats(h, 3, C.byref(ss)) == _ffi.ET_E_INVALID_ARG
        assert lib().et_plan_sweep_stats(None, _ffi.ET_VARIANT_F64, C.byref(ss)) == _ffi.ET_E_INVALID_ARG
        assert lib().et_plan_sweep_stats(h, _ffi.ET_VARIANT_F64, None) == _ffi.ET_E_INVALID_ARG
        # a graph this small runs the f64 sweep but not the exact one (row per lane below 16K rows)
        assert lib().et_plan_sweep_stats(h, _ffi.ET_VARIANT_F64, C.byref(ss)) == 0 and ss.blocks > 0 and ss.edges > 0
        assert lib().et_plan_sweep_stats(h, _ffi.ET_VARIANT_FR, C.byref(ss)) == 0 and ss.blocks == 0
    finally:
        lib().et_plan_free(h)
