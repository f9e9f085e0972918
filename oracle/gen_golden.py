#!/usr/bin/env python3
"""Generates tests/golden/*.json from the Python restatement -- TEST INFRASTRUCTURE ONLY.

The reference is Rust and cannot run in this image, so the expected outputs come from
oracle/eigentrust_ref.py.  Every vector is re-checked here, before it is written, against the
things the reference itself pins:
  * the conservation assert native.rs:331-334 (inside EigenTrustSet.converge),
  * the identity score_fr == num * den^-1 (mod r) of threshold/native.rs:75-81,
  * for the CLI asset, the literal scores in eigentrust-cli/assets/scores.csv:2-3.
Input matrices are the literals of the reference's own tests (cited per scenario).

Run:  python oracle/gen_golden.py        (rewrites tests/golden/)
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import eigentrust_ref as ref  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
R = ref.R_MODULUS


def _ser(st):
    if st[0] == "op":
        return ["op", hex(st[1]), [hex(v) for v in st[2]]]
    if st[0] in ("add", "remove"):
        return [st[0], hex(st[1])]
    return [st[0]]


def run_scenario(name, cite, n, num_iter, initial, steps, domain=42):
    """steps: list of ("add", addr) | ("remove", addr) | ("op", addr, row) | ("converge",).
    Every ("converge",) records the Fr scores, the rational scores and the slot layout."""
    s = ref.EigenTrustSet(n, num_iter, initial, domain)
    checkpoints = []
    for st in steps:
        if st[0] == "add":
            s.add_member(st[1])
        elif st[0] == "remove":
            s.remove_member(st[1])
        elif st[0] == "op":
            s.update_op(st[1], st[2])
        elif st[0] == "converge":
            f = s.converge()
            q = s.converge_rational()
            for fi, qi in zip(f, q):
                assert ref.rational_to_fr(qi) == fi, name  # threshold/native.rs:75-81
            checkpoints.append({
                "slots": [hex(a) for a, _ in s.set],
                "scores_fr": [hex(v) for v in f],
                "scores_rational": ["%x/%x" % (x.numerator, x.denominator) for x in q],  # hex num/den
            })
    return {
        "name": name, "reference": cite, "num_neighbours": n, "num_iterations": num_iter,
        "initial_score": initial, "domain": domain,
        "steps": [_ser(st) for st in steps],
        "checkpoints": checkpoints,
    }


def dense_scenario(name, cite, ops, num_iter, initial=1000, n=None):
    n = n or len(ops)
    addrs = [0x1000 + i for i in range(len(ops))]
    steps = [("add", a) for a in addrs]
    for a, row in zip(addrs, ops):
        steps.append(("op", a, list(row) + [0] * (n - len(row))))
    steps.append(("converge",))
    return run_scenario(name, cite, n, num_iter, initial, steps)


def main():
    os.makedirs(OUT, exist_ok=True)
    scen = []

    # eigentrust-cli/assets/attestations.csv:2 -> scores.csv:2-3; slot order is ascending
    # address (eigentrust/src/lib.rs:345-361), attester 0xf39f... (key 0 of eth.rs:153-160)
    a = 0x70997970c51812dc3a010c7d01b50e0d17dc79c8
    b = 0xf39fd6e51aad88f6f4ce6ab8827279cfffb92266
    cli = run_scenario("cli_asset", "eigentrust-cli/assets/attestations.csv:2, scores.csv:2-3",
                       4, 20, 1000,
                       [("add", a), ("add", b), ("op", b, [5, 0, 0, 0]), ("converge",)], domain=0)
    assert cli["checkpoints"][0]["scores_fr"] == ["0x3e8", "0x3e8", "0x0", "0x0"]
    assert cli["checkpoints"][0]["scores_rational"][:2] == ["3e8/1", "3e8/1"]  # 1000/1
    scen.append(cli)

    # native.rs:1101-1117 (test_scaling_1: N=10, 30 iterations)
    ops10 = [[0, 15, 154, 165, 0, 123, 56, 222, 253, 12], [210, 0, 10, 210, 20, 10, 20, 30, 440, 50],
             [40, 10, 0, 20, 30, 410, 20, 445, 23, 2], [10, 18, 20, 0, 310, 134, 45, 12, 439, 12],
             [30, 130, 44, 210, 0, 12, 445, 62, 12, 55], [0, 15, 154, 165, 123, 0, 56, 222, 253, 12],
             [210, 20, 10, 210, 20, 10, 0, 30, 440, 50], [40, 10, 445, 20, 30, 410, 20, 0, 23, 2],
             [10, 18, 20, 439, 310, 134, 45, 12, 0, 12], [30, 130, 44, 210, 55, 12, 445, 62, 12, 0]]
    scen.append(dense_scenario("test_scaling_1", "dynamic_sets/native.rs:1095-1128", ops10, 30))
    # dynamic_sets/mod.rs:747-753 (closed graph, N=5, 20 iterations)
    ops5 = [[0, 200, 300, 500, 0], [100, 0, 100, 100, 700], [400, 100, 0, 200, 300],
            [100, 100, 700, 0, 100], [300, 100, 400, 200, 0]]
    scen.append(dense_scenario("closed_graph_5", "dynamic_sets/mod.rs:744-868", ops5, 20))
    # threshold/mod.rs:861-866 (N=4, 20 iterations)
    ops4 = [[0, 200, 300, 500], [100, 0, 600, 300], [400, 100, 0, 500], [100, 200, 700, 0]]
    scen.append(dense_scenario("threshold_4", "threshold/mod.rs:850-889", ops4, 20))

    # unit-test scenarios of native.rs:512-945 (N=12, 10 iterations, INITIAL 1000, DOMAIN 42)
    N, IT = 12, 10
    a1, a2, a3 = 0xa1, 0xa2, 0xa3

    def row(**kw):
        r = [0] * N
        for k, v in kw.items():
            r[int(k[1:])] = v
        return r
    scen.append(run_scenario("two_members_without_opinions", "native.rs:512-541", N, IT, 1000,
                             [("add", a1), ("add", a2), ("converge",)]))
    scen.append(run_scenario("two_members_one_opinion", "native.rs:543-586", N, IT, 1000,
                             [("add", a1), ("add", a2), ("op", a1, row(i1=1000)), ("converge",)]))
    scen.append(run_scenario("two_members_with_opinions", "native.rs:588-645", N, IT, 1000,
                             [("add", a1), ("add", a2), ("op", a1, row(i1=1000)),
                              ("op", a2, row(i0=1000)), ("converge",)]))
    three = [("add", a1), ("add", a2), ("add", a3), ("op", a1, row(i1=300, i2=700)),
             ("op", a2, row(i0=600, i2=400)), ("op", a3, row(i0=600, i1=400))]
    scen.append(run_scenario("three_members_with_opinions", "native.rs:647-726", N, IT, 1000,
                             three + [("converge",)]))
    scen.append(run_scenario("three_members_two_opinions", "native.rs:728-792", N, IT, 1000,
                             three[:5] + [("converge",)]))
    scen.append(run_scenario("three_members_quit_one", "native.rs:794-879", N, IT, 1000,
                             three + [("converge",), ("remove", a2), ("converge",)]))
    scen.append(run_scenario("filter_peers_ops", "native.rs:947-1038", N, IT, 1000,
                             [("add", a1), ("add", a2), ("add", a3), ("op", a1, row(i0=10, i1=10)),
                              ("op", a2, row(i2=30)), ("op", a3, row(i0=10)), ("converge",)]))

    # edge cases the reference's rules imply
    scen.append(run_scenario("rejoin_into_freed_slot", "native.rs:180-181,189-198", 4, 20, 1000,
                             [("add", 1), ("add", 2), ("add", 3), ("op", 1, [0, 7, 9, 0]),
                              ("op", 2, [3, 0, 1, 0]), ("op", 3, [1, 1, 0, 0]), ("converge",),
                              ("remove", 2), ("converge",), ("add", 9), ("converge",),
                              ("op", 9, [5, 0, 5, 0]), ("converge",)]))
    scen.append(run_scenario("all_zero_opinions", "native.rs:263-278", 4, 20, 1000,
                             [("add", 1), ("add", 2), ("add", 3), ("add", 4)] +
                             [("op", k, [0, 0, 0, 0]) for k in (1, 2, 3, 4)] + [("converge",)]))
    scen.append(run_scenario("self_only_opinion", "native.rs:254-259", 4, 20, 1000,
                             [("add", 1), ("add", 2), ("add", 3), ("op", 1, [99, 0, 0, 0]),
                              ("op", 2, [1, 0, 2, 0]), ("converge",)]))
    # a row whose entries sum to a multiple of r counts as zero (native.rs:263-264)
    scen.append(run_scenario("row_sum_multiple_of_r", "native.rs:263-264", 4, 20, 1000,
                             [("add", 1), ("add", 2), ("add", 3), ("op", 1, [0, R - 5, 5, 0]),
                              ("op", 2, [4, 0, 6, 0]), ("op", 3, [1, 2, 0, 0]), ("converge",)]))
    rng = random.Random(20240205)
    big = [[0 if i == j else rng.randrange(R) for j in range(6)] for i in range(6)]
    scen.append(dense_scenario("full_field_values", "native.rs:305-329", big, 20))
    u32 = [[0 if i == j else rng.randrange(2 ** 32) for j in range(8)] for i in range(8)]
    scen.append(dense_scenario("u32_values", "native.rs:305-329", u32, 20))
    u8 = [[rng.randrange(256) for j in range(4)] for i in range(4)]
    scen.append(dense_scenario("u8_values_4_peers", "threshold/native.rs:309-345", u8, 20))
    scen.append(dense_scenario("huge_initial_score", "native.rs:184", ops4, 20, initial=2 ** 127 + 12345))

    with open(os.path.join(OUT, "scenarios.json"), "w") as f:
        json.dump(scen, f, indent=1)

    # threshold known answers, threshold/native.rs:135-226 (literal inputs and expected booleans)
    kats = []
    for num, den, th, limbs, p10, expect in [(345111, 1000, 346, 2, 3, False), (345111, 1000, 344, 2, 3, True),
                                            (347123456789123, 1984263563965, 346, 5, 3, True)]:
        from fractions import Fraction
        score = num % R * ref.fr_inv(den) % R
        got = ref.check_threshold(score, Fraction(num, den), th, limbs, p10, 4, 1000)
        assert got == expect
        kats.append({"num": num, "den": den, "threshold": th, "num_limbs": limbs, "power_of_ten": p10,
                     "score_fr": hex(score), "expect": expect})
    with open(os.path.join(OUT, "threshold_kats.json"), "w") as f:
        json.dump({"reference": "threshold/native.rs:135-226", "kats": kats}, f, indent=1)

    # seeded random sparse graphs: edge lists + sha256 of the n*32 score bytes (and a few samples)
    sparse = []
    for n, deg, n_empty, n_dangling, vmax, seed in [(64, 6, 5, 4, 255, 1), (257, 9, 0, 11, 255, 2),
                                                    (1000, 12, 37, 50, 2 ** 32 - 1, 3)]:
        rg = random.Random(seed)
        member = [1] * n
        for i in rg.sample(range(n), n_empty):
            member[i] = 0
        dang = set(rg.sample(range(n), n_dangling))
        edges = []
        for i in range(n):
            if i in dang:
                continue
            for t in rg.sample(range(n), min(n, rg.randrange(1, 2 * deg))):
                edges.append((i, t, rg.randrange(1, vmax + 1)))   # may include self / empty targets
        # duplicates: re-attest a few pairs with a new value (last write wins, lib.rs:415)
        for _ in range(n // 8):
            f_, t_, _v = edges[rg.randrange(len(edges))]
            edges.append((f_, t_, rg.randrange(1, vmax + 1)))
        exp = sparse_reference(n, member, edges, 20, 1000)
        blob = b"".join(ref.fr_to_le_bytes(v) for v in exp)
        sparse.append({"n": n, "num_iterations": 20, "initial_score": 1000, "member": member,
                       "edges": edges, "sha256": hashlib.sha256(blob).hexdigest(),
                       "samples": {str(i): hex(exp[i]) for i in range(0, n, max(1, n // 16))}})
    with open(os.path.join(OUT, "sparse_graphs.json"), "w") as f:
        json.dump({"reference": "native.rs:234-337 evaluated through the dense definition", "graphs": sparse}, f)
    print("wrote", len(scen), "scenarios,", len(kats), "threshold KATs,", len(sparse), "sparse graphs")


def sparse_reference(n, member, edges, num_iter, initial):
    """Dense-definition evaluation (EigenTrustSet) of an attestation list."""
    s = ref.EigenTrustSet(n, num_iter, initial, 42)
    for i in range(n):
        s.add_member(i + 1)                # address i+1 sits in slot i
    rows = {}
    for f_, t_, v in edges:
        rows.setdefault(f_, [0] * n)[t_] = v
    for f_, r_ in rows.items():
        if member[f_]:
            s.update_op(f_ + 1, r_)
    for i in range(n):
        if not member[i]:
            s.remove_member(i + 1)
    return s.converge()


if __name__ == "__main__":
    main()
