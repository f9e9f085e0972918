"""calculate_scores end to end (eigentrust/src/lib.rs:201-236 over :339-440): attestation records
in, score records out, converge on the GPU.  The reference's committed CLI assets are the literal
golden: attestations.csv -> scores.csv byte for byte (the attester of the single record is Anvil
account 0, eigentrust/src/eth.rs:153-160; signature recovery is outside this library)."""
import os
import random
from fractions import Fraction

import pytest

import eigentrust_ref as ref
import protocol_amd as pa
from protocol_amd import _ffi
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ASSETS = os.path.join(GOLDEN, "cli_assets")
ANVIL_0 = "0xf39fd6e51aad88f6f4ce6ab8827279cfffb92266"


def test_cli_assets_scores_csv_byte_for_byte():
    atts = pa.read_attestations_csv(open(os.path.join(ASSETS, "attestations.csv")).read())
    scores = pa.calculate_scores(atts, [ANVIL_0])
    assert pa.write_scores_csv(scores) == open(os.path.join(ASSETS, "scores.csv")).read()
    for s in scores:   # the threshold check the reference runs on each score (lib.rs:480-505)
        fr = int.from_bytes(s.score_fr, "big")
        assert pa.Threshold(fr, s.rational, 999).check_threshold() is True
        assert pa.Threshold(fr, s.rational, 1001).check_threshold() is False


def att(about, value, domain=bytes(20)):
    return pa.SignedAttestationRaw(about, domain, value, bytes(32), bytes(32), bytes(32), 0)


def expected_scores(atts, attesters, nn, iters, init):
    """The reference's set building order restated with the Python checker."""
    parts = sorted({a.about for a in atts} | set(attesters))
    s = ref.EigenTrustSet(nn, iters, init, 0)
    for p in parts:
        s.add_member(int.from_bytes(p, "big"))
    matrix = {p: [0] * nn for p in parts}
    for a, who in zip(atts, attesters):
        matrix[who][parts.index(a.about)] = a.value
    for p in parts:
        if p in set(attesters):
            s.update_op(int.from_bytes(p, "big"), matrix[p])
    return parts, s.converge(), s.converge_rational()


@pytest.mark.parametrize("nn,iters,seed", [(4, 20, 1), (4, 3, 2), (8, 4, 3), (16, 2, 4), (6, 20, 5)])
def test_random_attestation_lists_match_the_restatement(nn, iters, seed):
    rng = random.Random(seed)
    peers = [bytes(rng.randrange(256) for _ in range(20)) for _ in range(nn)]
    atts, who = [], []
    for _ in range(rng.randrange(nn, 3 * nn)):
        f, t = rng.sample(peers[: max(2, nn - 1)], 2)
        atts.append(att(t, rng.randrange(0, 256)))
        who.append(f)
    atts.append(att(atts[0].about, 77))     # a later record of the same pair replaces the earlier one
    who.append(who[0])
    parts, want_fr, want_rat = expected_scores(atts, who, nn, iters, 1000)
    fits = all(max(q.numerator, q.denominator).bit_length() <= 256 for q in want_rat[: len(parts)])
    if not fits:                             # lib.rs:222-229 panics in the reference
        with pytest.raises(pa.EigenTrustError) as e:
            pa.calculate_scores(atts, who, num_neighbours=nn, num_iterations=iters)
        assert e.value.code == _ffi.ET_E_RATIONAL_TOO_LARGE
        return
    got = pa.calculate_scores(atts, who, num_neighbours=nn, num_iterations=iters)
    assert [g.address for g in got] == parts
    for g, fr, q in zip(got, want_fr, want_rat):
        assert int.from_bytes(g.score_fr, "big") == fr
        assert g.rational == q
        assert int.from_bytes(g.score_hex, "big") == q.numerator // q.denominator


def test_calculate_scores_errors():
    a, b, c, d, e = (bytes([i]) * 20 for i in range(1, 6))
    with pytest.raises(pa.EigenTrustError) as err:           # lib.rs:366-369
        pa.calculate_scores([att(b, 1), att(c, 1), att(d, 1), att(e, 1)], [a, a, a, a], num_neighbours=4)
    assert err.value.code == _ffi.ET_E_SET_FULL
    with pytest.raises(pa.EigenTrustError) as err:           # lib.rs:371-374
        pa.calculate_scores([att(a, 1)], [a])
    assert err.value.code == _ffi.ET_E_INSUFFICIENT_PEERS
    with pytest.raises(pa.EigenTrustError) as err:           # opinion/native.rs:76
        pa.calculate_scores([att(b, 1, domain=bytes([9]) * 20)], [a])
    assert err.value.code == _ffi.ET_E_BAD_DOMAIN


# ---- signed path: recovery and verification included (lib.rs:339-466) -------------------------------
import crypto_ref as cref  # noqa: E402


def signed(secret, about, value, rng, domain=bytes(20)):
    h = cref.attestation_hash(about, domain, value, bytes(32))
    r, s, odd = cref.sign(secret, h, rng.randrange(1, cref.SECP_N))
    return pa.SignedAttestationRaw(about, domain, value, bytes(32), r.to_bytes(32, "little"), s.to_bytes(32, "little"), odd)


def test_cli_assets_from_signatures_alone():
    atts = pa.read_attestations_csv(open(os.path.join(ASSETS, "attestations.csv")).read())
    scores, pub = pa.et_circuit_setup(atts)
    assert pa.write_scores_csv(scores) == open(os.path.join(ASSETS, "scores.csv")).read()
    a = atts[0]
    assert pub.participants == [int.from_bytes(a.about, "big"), int(ANVIL_0, 16), 0, 0]
    assert pub.scores == [1000, 1000, 0, 0] and pub.domain == 0
    h = cref.attestation_hash(a.about, a.domain, a.value, a.message)
    assert pub.opinion_hash == cref.sponge([cref.sponge([h, 0, 0, 0])])     # opinion/native.rs:103-105, lib.rs:455-457
    assert pa.calculate_scores(atts) == scores


@pytest.mark.parametrize("nn,iters,seed", [(4, 20, 11), (4, 4, 12), (8, 3, 13)])
def test_signed_attestation_lists_match_the_restatement(nn, iters, seed):
    rng = random.Random(seed)
    secrets = [rng.randrange(1, cref.SECP_N) for _ in range(nn - 1)]
    addr = [cref.address_of(cref.public_key(s)) for s in secrets]
    atts, who = [], []
    for _ in range(2 * nn):
        i, j = rng.sample(range(len(secrets)), 2)
        atts.append(signed(secrets[i], addr[j], rng.randrange(1, 256), rng))
        who.append(addr[i])
    parts, want_fr, want_rat = expected_scores(atts, who, nn, iters, 1000)
    fits = all(max(q.numerator, q.denominator).bit_length() <= 256 for q in want_rat[: len(parts)])
    if not fits:
        with pytest.raises(pa.EigenTrustError) as e:
            pa.calculate_scores(atts, num_neighbours=nn, num_iterations=iters)
        assert e.value.code == _ffi.ET_E_RATIONAL_TOO_LARGE
        return
    assert pa.recover_attesters(atts) == who
    got, pub = pa.et_circuit_setup(atts, num_neighbours=nn, num_iterations=iters)
    assert [g.address for g in got] == parts
    for g, fr, q in zip(got, want_fr, want_rat):
        assert int.from_bytes(g.score_fr, "big") == fr and g.rational == q
    assert pub.scores == want_fr and pub.participants == [int.from_bytes(p, "big") for p in parts] + [0] * (nn - len(parts))
    assert pa.calculate_scores(atts, who, num_neighbours=nn, num_iterations=iters) == got


def test_scores_from_attestations_any_size():
    """The large-set path: attesters recovered on the GPU, graph engine for the iteration."""
    atts = pa.read_attestations_csv(open(os.path.join(ASSETS, "attestations.csv")).read())
    addrs, scores = pa.scores_from_attestations(atts)
    assert [a.hex() for a in addrs] == ["70997970c51812dc3a010c7d01b50e0d17dc79c8", ANVIL_0[2:]] and scores == [1000, 1000]

    rng = random.Random(77)
    n_peers = 180
    secrets = [rng.randrange(1, cref.SECP_N) for _ in range(n_peers)]
    keys = [pa.public_key_of_secret(s)[1] for s in secrets]
    silent = [bytes(rng.randrange(256) for _ in range(20)) for _ in range(15)]      # rated, never attesting: dangling
    atts, who = [], []
    for _ in range(2500):
        i = rng.randrange(n_peers)
        about = rng.choice(keys + silent)
        atts.append(pa.sign_attestation(secrets[i], about, rng.randrange(0, 256), rng.randrange(1, cref.SECP_N)))
        who.append(keys[i])
    allp = sorted({a.about for a in atts} | set(who))
    s = ref.EigenTrustSet(len(allp), 20, 1000, 0)
    for p in allp:
        s.add_member(int.from_bytes(p, "big"))
    matrix = {p: [0] * len(allp) for p in set(who)}
    pos = {p: i for i, p in enumerate(allp)}
    for a, w in zip(atts, who):
        matrix[w][pos[a.about]] = a.value
    for p in allp:
        if p in matrix:
            s.update_op(int.from_bytes(p, "big"), matrix[p])
    want = s.converge()
    addrs, scores, d64 = pa.scores_from_attestations(atts, f64=True)
    assert addrs == allp and scores == want
    # the f64 variant tracks the same scores (loosely: the field scores are not comparable digit by digit)
    assert abs(float(d64.sum()) - 1000.0 * len(allp)) <= 1e-6 * 1000.0 * len(allp)
    with pytest.raises(pa.EigenTrustError) as e:
        pa.scores_from_attestations(atts, domain=bytes([1]) * 20)
    assert e.value.code == _ffi.ET_E_BAD_DOMAIN


def test_cli_local_scores_regenerates_the_reference_asset(tmp_path):
    import shutil
    from protocol_amd import cli
    shutil.copy(os.path.join(ASSETS, "attestations.csv"), tmp_path / "attestations.csv")
    assert cli.main(["local-scores", "--assets", str(tmp_path)]) == 0
    assert (tmp_path / "scores.csv").read_text() == open(os.path.join(ASSETS, "scores.csv")).read()
    assert cli.main(["threshold", "--assets", str(tmp_path), "--peer", ANVIL_0, "--threshold", "999"]) == 0
    assert cli.main(["threshold", "--assets", str(tmp_path), "--peer", ANVIL_0, "--threshold", "1001"]) == 1
    assert cli.main(["local-scores", "--assets", str(tmp_path), "--large"]) == 0
    assert (tmp_path / "scores.csv").read_text().splitlines()[1].endswith("03e8")
