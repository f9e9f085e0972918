"""The CPU checker against everything the reference itself pins (no GPU needed)."""
import hashlib
import random
from fractions import Fraction

import numpy as np
import pytest

import c_oracle
import eigentrust_ref as ref

R = ref.R_MODULUS


def replay(sc, cls=ref.EigenTrustSet):
    s = cls(sc["num_neighbours"], sc["num_iterations"], sc["initial_score"], sc["domain"])
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


def test_cli_asset_literal_scores():
    # eigentrust-cli/assets/attestations.csv:2 -> scores.csv:2-3 (both peers 0x3e8, 1000/1)
    a = 0x70997970c51812dc3a010c7d01b50e0d17dc79c8
    b = 0xf39fd6e51aad88f6f4ce6ab8827279cfffb92266
    s = ref.EigenTrustSet(4, 20, 1000, 0)
    s.add_member(a)
    s.add_member(b)
    s.update_op(b, [5, 0, 0, 0])
    assert s.converge() == [0x3e8, 0x3e8, 0, 0]
    assert s.converge_rational()[:2] == [Fraction(1000, 1), Fraction(1000, 1)]


def test_python_oracle_reproduces_golden_and_invariants(scenarios):
    for sc in scenarios:
        for s, exp in replay(sc):
            f = s.converge()  # asserts conservation (native.rs:331-334) inside
            q = s.converge_rational()
            assert [hex(v) for v in f] == exp["scores_fr"], sc["name"]
            assert ["%x/%x" % (x.numerator, x.denominator) for x in q] == exp["scores_rational"]
            for fi, qi in zip(f, q):  # threshold/native.rs:75-81
                assert ref.rational_to_fr(qi) == fi


def test_circuit_statement_agrees_with_native_statement(scenarios):
    """The reference checks its native converge against the circuit with MockProver
    (dynamic_sets/mod.rs:744-868).  The circuit's constraints restated (circuit_converge) must give the native
    scores wherever every slot is occupied; with an empty slot the two statements differ by design (the circuit
    seeds and redistributes empty slots too) and the circuit's fixed total must still hold."""
    full = 0
    for sc in scenarios:
        for s, exp in replay(sc):
            addrs = [a for a, _ in s.set]
            ops = [list(s.ops.get(a, [0] * s.n)) if a != 0 else [0] * s.n for a in addrs]
            c = ref.circuit_converge(addrs, ops, s.num_iterations, s.initial_score)   # asserts the fixed total inside
            if all(a != 0 for a in addrs):
                assert [hex(v) for v in c] == exp["scores_fr"], sc["name"]
                full += 1
    assert full >= 3        # the reference's 10x10, 5x5 and 4x4 test matrices fill every slot
    rng = random.Random(4)
    for n in (2, 4, 7):     # random full sets, incl. all-zero rows and opinions about oneself
        s = ref.EigenTrustSet(n, 12, 1000, 42)
        addrs = [rng.randrange(1, R) for _ in range(n)]
        for a in addrs:
            s.add_member(a)
        ops = [[rng.choice([0, 0, rng.randrange(256), rng.randrange(R)]) for _ in range(n)] for _ in range(n)]
        ops[1] = [0] * n
        for a, row in zip(addrs, ops):
            s.update_op(a, row)
        assert ref.circuit_converge(addrs, ops, 12, 1000) == s.converge()


def test_known_peer_score_of_scaling_matrix(scenarios):
    sc = next(s for s in scenarios if s["name"] == "test_scaling_1")
    assert sc["checkpoints"][0]["scores_fr"][0] == "0x17942a6a1e6d480a143b3a2bc6cc20e49c4af5440c6d2da3db85036ae13cb096"


def test_threshold_known_answers(threshold_kats):
    # literal inputs and expected booleans of threshold/native.rs:135-226
    for k in threshold_kats:
        got = ref.check_threshold(int(k["score_fr"], 16), Fraction(k["num"], k["den"]), k["threshold"],
                                  k["num_limbs"], k["power_of_ten"], 4, 1000)
        assert got == k["expect"]


def test_reference_panics_are_reproduced():
    s = ref.EigenTrustSet(12, 10, 1000, 42)
    s.add_member(7)
    with pytest.raises(AssertionError):
        s.add_member(7)            # native.rs:455-482
    with pytest.raises(ref.InsufficientPeers):
        s.converge()               # native.rs:484-510
    t = ref.EigenTrustSet(2, 10, 1000, 42)
    t.add_member(1)
    t.add_member(2)
    with pytest.raises(AssertionError):
        t.add_member(3)            # native.rs:181 unwrap on a full set
    with pytest.raises(AssertionError):
        t.remove_member(9)         # native.rs:192


def test_c_field_arithmetic_matches_integers():
    rng = random.Random(11)
    for _ in range(300):
        a, b = rng.randrange(R), rng.randrange(R)
        assert c_oracle.fr_binop("mul", a, b) == a * b % R
        assert c_oracle.fr_binop("add", a, b) == (a + b) % R
        assert c_oracle.fr_binop("sub", a, b) == (a - b) % R
    for a in (1, 2, R - 1, 5, rng.randrange(R)):
        assert c_oracle.fr_inv(a) * a % R == 1
    assert c_oracle.fr_inv(0) == 0  # invert().unwrap_or(ZERO), native.rs:308


def test_c_dense_restatement_matches_golden(scenarios):
    for sc in scenarios:
        n = sc["num_neighbours"]
        for s, exp in replay(sc):
            member = [1 if a != 0 else 0 for a, _ in s.set]
            has = [1 if (a != 0 and a in s.ops) else 0 for a, _ in s.set]
            ops = [s.ops.get(a, [0] * n) if a != 0 else [0] * n for a, _ in s.set]
            rc, out, _ = c_oracle.dense_converge_fr(n, sc["num_iterations"], sc["initial_score"], member, has, ops)
            assert rc == 0
            assert [hex(v) for v in out] == exp["scores_fr"], sc["name"]


def test_c_sparse_restatement_matches_golden_graphs(sparse_goldens):
    for gd in sparse_goldens:
        e = np.array(gd["edges"], dtype=np.uint64)
        rc, out, _nnz, _ = c_oracle.sparse_converge_fr(gd["n"], np.array(gd["member"], dtype=np.uint8), e[:, 0], e[:, 1],
                                                       value_u32=e[:, 2], num_iter=gd["num_iterations"],
                                                       initial=gd["initial_score"])
        assert rc == 0
        assert hashlib.sha256(out.tobytes()).hexdigest() == gd["sha256"]


def test_c_sparse_threads_do_not_change_results():
    rng = np.random.default_rng(4)
    n = 5000
    frm = np.repeat(np.arange(n, dtype=np.uint32), 9)
    to = rng.integers(0, n, len(frm), dtype=np.uint32)
    val = rng.integers(0, 256, len(frm), dtype=np.uint32)
    member = np.ones(n, dtype=np.uint8)
    a = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, n_threads=1)
    b = c_oracle.sparse_converge_fr(n, member, frm, to, value_u32=val, n_threads=4)
    assert a[0] == 0 and b[0] == 0 and np.array_equal(a[1], b[1])


def test_c_f64_restatement_close_to_rationals(scenarios):
    sc = next(s for s in scenarios if s["name"] == "test_scaling_1")
    for s, exp in replay(sc):
        n = sc["num_neighbours"]
        frm, to, val = [], [], []
        for i, (a, _) in enumerate(s.set):
            for j, v in enumerate(s.ops.get(a, [])):
                if v:
                    frm.append(i); to.append(j); val.append(v)
        rc, o64, old, _res, _nnz, _ = c_oracle.sparse_converge_f64(n, np.ones(n, dtype=np.uint8), frm, to, value_u32=val,
                                                                  num_iter=sc["num_iterations"], initial=1000.0, want_ld=True)
        assert rc == 0
        q = [Fraction(int(x.split("/")[0], 16), int(x.split("/")[1], 16)) for x in exp["scores_rational"]]
        err = sum(abs(Fraction(float(g)) - x) for g, x in zip(o64, q)) / sum(q)
        assert err <= 1e-12
