"""GPU parity: the HIP path, called through the C ABI, against the CPU checker and the goldens.

Bit-exact for the field (Fr) variant; <= 1e-9 relative L1 for the f64 variant (the tolerance the
path's specification states, checked against exact rationals where they exist and against the
long-double restatement elsewhere).
"""
import hashlib
import random
from fractions import Fraction

import numpy as np
import pytest

import c_oracle
import eigentrust_ref as ref
import protocol_amd as pa

pytestmark = pytest.mark.gpu
R = ref.R_MODULUS
F64_TOL = 1e-9


def rel_l1(a, b):
    a = np.asarray(a, dtype=np.longdouble)
    b = np.asarray(b, dtype=np.longdouble)
    return float(np.abs(a - b).sum() / np.abs(b).sum())


def test_device_field_multiply_matches_integers():
    rng = random.Random(7)
    a = [rng.randrange(R) for _ in range(4096)] + [0, 1, R - 1, R - 1, 2**253, 5]
    b = [rng.randrange(R) for _ in range(4096)] + [R - 1, R - 1, R - 1, 1, 2**253, 0]
    out = np.zeros((len(a), 32), dtype=np.uint8)
    ab, bb = pa.fr_bytes_from_ints(a), pa.fr_bytes_from_ints(b)   # keep the buffers alive across the call
    pa._ffi.check(pa.lib().et_dev_fr_mul(len(a), ab.ctypes.data, bb.ctypes.data, out.ctypes.data))
    assert pa.fr_ints_from_bytes(out) == [x * y % R for x, y in zip(a, b)]


def replay(sc):
    s = pa.EigenTrustSet(sc["domain"], sc["num_neighbours"], sc["num_iterations"], sc["initial_score"])
    k = 0
    for st in sc["steps"]:
        if st[0] == "add":
            s.add_member(int(st[1], 16))
        elif st[0] == "remove":
            s.remove_member(int(st[1], 16))
        elif st[0] == "op":
            s.update_op(int(st[1], 16), [int(v, 16) for v in st[2]])
        else:
            yield s, sc["checkpoints"][k]
            k += 1


def test_set_api_converge_matches_golden_scenarios(scenarios):
    for sc in scenarios:
        for s, exp in replay(sc):
            got = s.converge()
            assert [hex(v) for v in got] == exp["scores_fr"], sc["name"]


def test_set_api_converge_f64_within_tolerance_of_rationals(scenarios):
    for sc in scenarios:
        if sc["name"] in ("full_field_values", "row_sum_multiple_of_r"):
            continue  # 254-bit opinions: the f64 variant is checked on the u8/u32 ranges it is meant for
        for s, exp in replay(sc):
            got, _res = s.converge_f64()
            q = [Fraction(int(a.split("/")[0], 16), int(a.split("/")[1], 16)) for a in exp["scores_rational"]]
            num = sum(abs(Fraction(float(g)) - x) for g, x in zip(got, q))
            den = sum(q)
            assert num / den <= F64_TOL, (sc["name"], float(num / den))


def test_sparse_goldens_bit_exact(sparse_goldens):
    for gd in sparse_goldens:
        e = np.array(gd["edges"], dtype=np.uint64)
        g = pa.Graph(gd["n"], e[:, 0], e[:, 1], value_u32=e[:, 2], member=np.array(gd["member"], dtype=np.uint8))
        out = g.converge_fr(gd["num_iterations"], gd["initial_score"])
        assert hashlib.sha256(out.tobytes()).hexdigest() == gd["sha256"], gd["n"]
        got = pa.fr_ints_from_bytes(out)
        for i, v in gd["samples"].items():
            assert hex(got[int(i)]) == v


def random_graph(rng, n, avg_deg, p_empty, p_dangling, vmax, hub=False):
    member = (rng.random(n) >= p_empty).astype(np.uint8)
    if member.sum() < 2:
        member[:2] = 1
    deg = rng.poisson(avg_deg, n)
    deg[rng.random(n) < p_dangling] = 0
    frm = np.repeat(np.arange(n, dtype=np.uint32), deg)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    if hub:  # a few very popular peers: rows longer than one wavefront pass
        sel = rng.random(len(frm)) < 0.3
        to[sel] = rng.integers(0, 3, sel.sum(), dtype=np.uint32)
    val = rng.integers(0, vmax + 1, len(frm), dtype=np.uint32)
    return member, frm, to, val


@pytest.mark.parametrize("n,avg,pe,pd,vmax,hub", [
    (2, 1, 0.0, 0.0, 255, False), (3, 2, 0.0, 0.5, 255, False), (64, 5, 0.1, 0.1, 255, False),
    (65, 3, 0.0, 0.0, 1, False), (1000, 8, 0.05, 0.02, 255, True), (5000, 20, 0.01, 0.3, 2**32 - 1, True),
    (20000, 16, 0.0, 0.01, 255, True), (777, 0, 0.2, 0.0, 255, False)])
def test_random_graphs_fr_bit_exact_and_f64_close(n, avg, pe, pd, vmax, hub):
    rng = np.random.default_rng(n * 31 + avg)
    member, frm, to, val = random_graph(rng, n, avg, pe, pd, vmax, hub)
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=20, initial=1000)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert g.info.nnz == nnz
    out = g.converge_fr(20, 1000)
    assert np.array_equal(out, exp)
    rc, e64, eld, eres, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0, want_ld=True)
    assert rc == 0
    o64, res = g.converge_f64(20, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL
    assert abs(res - eres) <= 1e-6 * max(1.0, abs(eres))


@pytest.mark.parametrize("n,edges,force", [(400000, 3000, False), (20000, 300000, True)])
def test_f64_byte_opinions_without_packed_records(n, edges, force, monkeypatch):
    # the 5-byte edge records hold a 16-bit source offset per 64-edge chunk: a few edges spread over very many
    # peers do not fit and take the 7-byte records, as does any graph with the measurement knob set
    if force:
        monkeypatch.setenv("ET_F64_UNPACKED", "1")
    rng = np.random.default_rng(n + edges)
    frm = rng.integers(0, n, edges, dtype=np.uint32)
    to = rng.integers(0, n, edges, dtype=np.uint32)
    val = rng.integers(1, 256, edges, dtype=np.uint32)
    member = np.ones(n, dtype=np.uint8)
    rc, e64, eld, eres, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0, want_ld=True)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member, variants=2)
    o64, res = g.converge_f64(20, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL


def test_field_valued_opinions_bit_exact():
    rng = random.Random(99)
    n = 300
    frm, to, vals = [], [], []
    for i in range(n):
        for t in rng.sample(range(n), 7):
            frm.append(i); to.append(t); vals.append(rng.randrange(R))
    member = np.ones(n, dtype=np.uint8)
    member[[5, 17]] = 0
    vb = pa.fr_bytes_from_ints(vals)
    rc, exp, _, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_fr=vb, num_iter=20, initial=1000)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_fr=vb, member=member)
    assert np.array_equal(g.converge_fr(20, 1000), exp)


def test_alpha_and_pretrust_f64():
    rng = np.random.default_rng(5)
    n = 3000
    member, frm, to, val = random_graph(rng, n, 10, 0.02, 0.05, 255)
    p = rng.random(n) * 2000.0 * member
    rc, e64, eld, _, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0,
                                                         alpha=0.15, pretrust=p, want_ld=True)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    o64, _ = g.converge_f64(20, 1000.0, alpha=0.15, pretrust=p)
    assert rel_l1(o64, eld) <= F64_TOL


def test_insufficient_peers_is_an_error():
    with pytest.raises(pa.EigenTrustError) as ei:
        pa.Graph(4, [0], [1], value_u32=[5], member=np.array([1, 0, 0, 0], dtype=np.uint8))
    assert ei.value.code == pa._ffi.ET_E_INSUFFICIENT_PEERS
    s = pa.EigenTrustSet(42, 12, 10, 1000)
    s.add_member(7)
    with pytest.raises(pa.EigenTrustError) as ei:
        s.converge()   # native.rs:484-510 test_one_member_converge
    assert ei.value.code == pa._ffi.ET_E_INSUFFICIENT_PEERS


@pytest.mark.parametrize("iters", [0, 1, 2, 7])
def test_iteration_counts_including_zero(iters):
    rng = np.random.default_rng(40 + iters)
    n = 900
    member, frm, to, val = random_graph(rng, n, 6, 0.1, 0.1, 255)
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    rc, exp, _, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000)
    assert rc == 0
    assert np.array_equal(g.converge_fr(iters, 1000), exp)
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=1000.0, want_ld=True)
    o64, _ = g.converge_f64(iters, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL
    if iters == 0:   # native.rs:317: the initial scores, zero for empty slots
        assert pa.fr_ints_from_bytes(g.converge_fr(0, 1000)) == [1000 if m else 0 for m in member]
    # the same graph object serves converge after converge (native.rs:873-878 calls converge repeatedly)
    assert np.array_equal(g.converge_fr(iters, 1000), exp)


@pytest.mark.parametrize("fused", [True, False])
def test_converges_in_a_row_with_other_initial_scores_and_iteration_counts(fused, monkeypatch):
    # A whole converge keeps its first input vector between calls and writes slot order from its last launch
    # (ET_NO_FUSED_RUN=1: init and un-permutation as launches of their own).  Peers without opinions keep a plain score
    # inside the graph object that the previous converge left behind, and empty slots must read zero every time.
    if not fused:
        monkeypatch.setenv("ET_NO_FUSED_RUN", "1")
    rng = np.random.default_rng(77)
    n = 6000
    member, frm, to, val = random_graph(rng, n, 9, 0.07, 0.2, 255, hub=True)
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    for iters, initial in ((20, 1000), (20, 1000), (3, 777), (1, 777), (20, 1000), (0, 5), (2, 5)):
        rc, exp, _, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=iters, initial=initial)
        assert rc == 0
        got = g.converge_fr(iters, initial)
        assert np.array_equal(got, exp), (iters, initial)
        assert all(v == 0 for v, m in zip(pa.fr_ints_from_bytes(got), member) if not m)
        rc, _e, eld, eres, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=iters, initial=float(initial), want_ld=True)
        assert rc == 0
        o64, res = g.converge_f64(iters, float(initial))
        assert rel_l1(o64, eld) <= F64_TOL, (iters, initial)
        assert np.all(o64[member == 0] == 0.0)
        if iters:
            assert abs(res - eres) <= 1e-6 * max(1.0, abs(eres))
    # damping with the default pre-trust (the initial scores) goes through the same launches
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0, alpha=0.15,
                                                         pretrust=1000.0 * member, want_ld=True)
    assert rc == 0
    o64, _ = g.converge_f64(20, 1000.0, alpha=0.15)
    assert rel_l1(o64, eld) <= F64_TOL


def test_steps_without_statistics_write_the_same_scores():
    # ET_STEP_NO_STATS: a step that forms neither residual nor score sum nor the count of changed rows (what every step of a
    # converge but the last asks for) writes the same vector as one that does
    import torch
    rng = np.random.default_rng(123)
    n = 9000
    member, frm, to, val = random_graph(rng, n, 12, 0.03, 0.15, 255, hub=True)
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    vec_len = int(g.info.vec_len)
    for kind in ("fr", "f64"):
        outs = []
        for stats in (True, False):
            mk = (lambda: torch.zeros((vec_len, 4), dtype=torch.int64, device=dev)) if kind == "fr" else (lambda: torch.zeros(vec_len, dtype=torch.float64, device=dev))
            x, y = mk(), mk()
            (g.init_fr if kind == "fr" else g.init_f64)(x.data_ptr(), 1000, st)
            for it in range(3):
                if kind == "fr":
                    g.step_fr(x.data_ptr(), y.data_ptr(), it, st, final=it == 2, stats=stats)
                else:
                    g.step_f64(x.data_ptr(), y.data_ptr(), it, 0.0, None, st, final=it == 2, stats=stats)
                x, y = y, x
            torch.cuda.synchronize(dev)
            outs.append(x.cpu().numpy().copy())
            if kind == "fr":
                total, changed = g.stats_fr(1, st)
                assert not stats or (total == 1000 * int(member.sum()) % R and changed > 0)     # without: unspecified
            else:
                res, total = g.stats_f64(1, st)
                assert not stats or (res > 0.0 and abs(total - 1000.0 * member.sum()) < 1e-6 * total)
        if kind == "fr":
            assert np.array_equal(outs[0], outs[1])
        else:
            assert np.allclose(outs[0], outs[1], rtol=1e-12, atol=0.0)


def test_no_attestations_at_all():
    # every member is dangling: each gives 1 to every other member (native.rs:263-278), scores stay put
    n = 70
    member = np.ones(n, dtype=np.uint8)
    member[[3, 69]] = 0
    g = pa.Graph(n, [], [], value_u32=[], member=member)
    assert pa.fr_ints_from_bytes(g.converge_fr(20, 1000)) == [1000 if m else 0 for m in member]
    o64, res = g.converge_f64(20, 1000.0)
    assert np.allclose(o64, 1000.0 * member, rtol=1e-12) and res < 1e-6


def test_one_row_receives_every_attestation():
    # a single destination row far longer than any workgroup's share of an even deal
    n = 60000
    frm = np.arange(1, n, dtype=np.uint32)
    to = np.zeros(n - 1, dtype=np.uint32)
    val = (np.arange(n - 1, dtype=np.uint32) % 255) + 1
    member = np.ones(n, dtype=np.uint8)
    rc, exp, _, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=5, initial=1000)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert np.array_equal(g.converge_fr(5, 1000), exp)
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=5, initial=1000.0, want_ld=True)
    o64, _ = g.converge_f64(5, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL


@pytest.mark.parametrize("row_per_lane", [False, True])
def test_both_exact_kernels_above_half_a_million_rows(row_per_lane, monkeypatch):
    """One share of 512K to 2M rows of byte opinions runs the exact variant as a blocked sweep; the row-per-lane
    kernel stays selectable there.  Both against the C restatement on a graph with empty slots, peers without
    opinions and three rows far longer than a workgroup's even share."""
    if row_per_lane:
        monkeypatch.setenv("ET_FR_ROW_PER_LANE", "1")
    rng = np.random.default_rng(77)
    n = 600000
    member, frm, to, val = random_graph(rng, n, 6, 0.03, 0.05, 255, hub=True)
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=6, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member, variants=1)
    assert g.info.nnz == nnz
    assert (int(g.info.exact_sweep_blocks) == 0) == row_per_lane
    assert np.array_equal(g.converge_fr(6, 1000), exp)
    assert np.array_equal(g.converge_fr(1, 1000),
                          c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=1, initial=1000,
                                                      n_threads=c_oracle.max_threads())[1])


def test_exact_sweep_with_32_bit_opinions(monkeypatch):
    """Opinions above a byte (the reference's opinion is any field element, native.rs:128): the exact variant runs the blocked
    sweep with rows of nine limbs and sixteen bits, the f64 sweep 8-byte packed records.  Hubs rated by tens of thousands
    with the largest opinion stress the carries of a row's sum."""
    rng = np.random.default_rng(78)
    n = 600000
    member, frm, to, val = random_graph(rng, n, 3, 0.0, 0.02, 2**32 - 1, False)
    sel = rng.random(len(frm)) < 0.1                       # three hubs of about 60 000 in-edges each
    to[sel] = rng.integers(0, 3, sel.sum(), dtype=np.uint32)
    big = np.flatnonzero(to < 3)
    val[big[: len(big) // 2]] = 0xFFFFFFFF
    assert 30000 < np.bincount(to[frm != to], minlength=n).max() < 2**17
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=5, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    rc, _e, eld, _r, _n, _s = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=5, initial=1000.0,
                                                           n_threads=c_oracle.max_threads(), want_ld=True)
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert int(g.info.exact_sweep_blocks) != 0 and int(g.info.nnz) == nnz
    assert np.array_equal(g.converge_fr(5, 1000), exp)
    assert np.array_equal(g.converge_fr(1, 1000),
                          c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=1, initial=1000,
                                                      n_threads=c_oracle.max_threads())[1])
    o64, _ = g.converge_f64(5, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL
    g.close()
    # unpacked records (source, slot and opinion apart) run the same arithmetic
    monkeypatch.setenv("ET_FR_UNPACKED", "1")
    monkeypatch.setenv("ET_F64_UNPACKED", "1")
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    assert int(g.info.exact_sweep_blocks) != 0
    assert np.array_equal(g.converge_fr(5, 1000), exp)
    assert rel_l1(g.converge_f64(5, 1000.0)[0], eld) <= F64_TOL
    g.close()
    # the row-per-lane kernel on the same graph (what 32-bit opinions ran before the sweep took them)
    monkeypatch.setenv("ET_FR_SWEEP_BYTES_ONLY", "1")
    g = pa.Graph(n, frm, to, value_u32=val, member=member, variants=1)
    assert int(g.info.exact_sweep_blocks) == 0
    assert np.array_equal(g.converge_fr(5, 1000), exp)
    g.close()
    monkeypatch.delenv("ET_FR_SWEEP_BYTES_ONLY")
    monkeypatch.delenv("ET_FR_UNPACKED")
    monkeypatch.delenv("ET_F64_UNPACKED")
    # a row with 2^17 in-edges or more does not fit the sweep's sixteen spare bits: such a graph keeps the row-per-lane kernel
    member, frm, to, val = random_graph(rng, n, 3, 0.0, 0.02, 2**32 - 1, True)     # three hubs of about 180 000 in-edges
    assert np.bincount(to[frm != to], minlength=n).max() >= 2**17
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=3, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0
    g = pa.Graph(n, frm, to, value_u32=val, member=member, variants=1)
    assert int(g.info.exact_sweep_blocks) == 0
    assert np.array_equal(g.converge_fr(3, 1000), exp)


def test_full_size_benchmark_graph_parity():
    """The graph the metric is quoted on (1M peers / 16M attestations, 20 iterations), whole result:
    exact variant bit for bit against the C restatement, f64 against its long-double restatement."""
    n = 1 << 20
    frm, to, val = pa.synth_graph(3, n, 1 << 24, seed=42, value_max=255)
    member = np.ones(n, dtype=np.uint8)
    g = pa.Graph(n, frm, to, value_u32=val, member=member)
    rc, exp, nnz, _ = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, num_iter=20, initial=1000,
                                                  n_threads=c_oracle.max_threads())
    assert rc == 0 and nnz == g.info.nnz == 1 << 24
    out = g.converge_fr(20, 1000)
    assert hashlib.sha256(out.tobytes()).hexdigest() == hashlib.sha256(exp.tobytes()).hexdigest()
    rc, _e, eld, _r, _, _ = c_oracle.sparse_converge_f64(n, member, frm, to, value_u32=val, num_iter=20, initial=1000.0,
                                                         n_threads=c_oracle.max_threads(), want_ld=True)
    o64, _ = g.converge_f64(20, 1000.0)
    assert rel_l1(o64, eld) <= F64_TOL


def test_full_size_properties_linearity_and_relabelling():
    """Size-independent properties on the 1M-peer graph, exact variant: scores are linear in the initial score
    (native.rs:317-329 has no constant term), and renaming the peers renames the scores."""
    n = 1 << 20
    frm, to, val = pa.synth_graph(3, n, 1 << 24, seed=42, value_max=255)
    g = pa.Graph(n, frm, to, value_u32=val, variants=1)
    a = pa.fr_ints_from_bytes(g.converge_fr(20, 1000)[:4096])
    b = pa.fr_ints_from_bytes(g.converge_fr(20, 3000)[:4096])
    assert [3 * x % R for x in a] == b
    one = g.converge_fr(20, 1000)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n).astype(np.uint32)          # peer i becomes peer perm[i]
    shuffle = rng.permutation(len(frm))                  # and the attestations arrive in another order
    g2 = pa.Graph(n, perm[frm][shuffle], perm[to][shuffle], value_u32=val[shuffle], variants=1)
    two = g2.converge_fr(20, 1000)
    assert np.array_equal(two[perm], one)
