"""Host-side neighbours of the converge path, through the C ABI, without a GPU: threshold check,
decimal decomposition, score packaging, the client's CSV formats, public-input layout.
Checked against the reference's literal answers (threshold/native.rs:135-226), its committed CLI
assets (eigentrust-cli/assets/attestations.csv, scores.csv, copied under tests/golden/cli_assets as
data fixtures) and the Python restatement in oracle/."""
import os
import random
from fractions import Fraction

import pytest

import eigentrust_ref as ref
import protocol_amd as pa
from protocol_amd import _ffi
from conftest import GOLDEN

R = pa.R_MODULUS
ASSETS = os.path.join(GOLDEN, "cli_assets")


def rat(text):
    n, d = text.split("/")
    return Fraction(int(n, 16), int(d, 16))


def test_threshold_known_answers(threshold_kats):
    for k in threshold_kats:
        t = pa.Threshold(int(k["score_fr"], 16), Fraction(k["num"], k["den"]), k["threshold"], k["num_limbs"],
                         k["power_of_ten"], 4, 1000)
        assert t.check_threshold() == k["expect"]


def test_threshold_matches_restatement_on_golden_scores(scenarios):
    checked = 0
    for sc in scenarios:
        cp = sc["checkpoints"][-1]
        nn, init = sc["num_neighbours"], sc["initial_score"]
        for fr_hex, rt in zip(cp["scores_fr"], cp["scores_rational"]):
            ratio, fr = rat(rt), int(fr_hex, 16)
            big = max(ratio.numerator, ratio.denominator)
            digits = len(str(big)) if big.bit_length() < 4000 else 10**6   # int -> str is capped at 4300 digits
            for thr in (0, 1, init // 2, init, nn * init - 1):
                t = pa.Threshold(fr, ratio, thr, 2, 72, nn, init)
                if digits > 144:   # utils.rs:340 underflows in the reference
                    with pytest.raises(pa.EigenTrustError) as e:
                        t.check_threshold()
                    assert e.value.code == _ffi.ET_E_RATIO_DIGITS
                    continue
                try:
                    want = ref.check_threshold(fr, ratio, thr, 2, 72, nn, init)
                except AssertionError:
                    with pytest.raises(pa.EigenTrustError):
                        t.check_threshold()
                    continue
                assert t.check_threshold() == want
                checked += 1
    assert checked > 50


def test_threshold_error_codes():
    ok = (1000, Fraction(1000, 1), 500)
    assert pa.Threshold(*ok).check_threshold() is True
    with pytest.raises(pa.EigenTrustError) as e:          # threshold/native.rs:81
        pa.Threshold(999, Fraction(1000, 1), 500).check_threshold()
    assert e.value.code == _ffi.ET_E_RATIO_MISMATCH
    with pytest.raises(pa.EigenTrustError) as e:          # :61-63 threshold >= NUM_NEIGHBOURS * INITIAL_SCORE
        pa.Threshold(1000, Fraction(1000, 1), 4000).check_threshold()
    assert e.value.code == _ffi.ET_E_THRESHOLD_RANGE
    with pytest.raises(pa.EigenTrustError) as e:          # :34-37 limb bound
        pa.Threshold(1000, Fraction(1000, 1), 5, num_limbs=1, power_of_ten=75).check_threshold()
    assert e.value.code == _ffi.ET_E_THRESHOLD_RANGE
    with pytest.raises(pa.EigenTrustError) as e:          # utils.rs:340
        pa.Threshold(1000, Fraction(10**7, 1), 5, num_limbs=2, power_of_ten=3).check_threshold()
    assert e.value.code == _ffi.ET_E_RATIO_DIGITS


def test_decimal_decomposition_matches_restatement():
    rng = random.Random(5)
    for _ in range(200):
        limbs, p10 = rng.choice([(2, 72), (2, 3), (5, 20), (1, 40)])
        digits = rng.randrange(1, limbs * p10 + 1)
        a, b = rng.randrange(1, 10**digits), rng.randrange(1, 10**digits)
        ratio = Fraction(a, b)
        got_n, got_d = pa.big_to_fe_rat(ratio, limbs, p10)
        scale = 10 ** (limbs * p10 - len(str(max(ratio.numerator, ratio.denominator))))
        want_n = ref._decompose_big_decimal(ratio.numerator * scale, limbs, p10)
        want_d = ref._decompose_big_decimal(ratio.denominator * scale, limbs, p10)
        assert got_n == want_n and got_d == want_d


def test_scalar_from_address():
    a = "0x70997970c51812dc3a010c7d01b50e0d17dc79c8"
    assert pa.scalar_from_address(a) == int(a, 16)      # eth.rs:77-95: reversed bytes, little-endian scalar
    assert pa.scalar_from_address(bytes(20)) == 0


def test_attestations_csv_round_trip_of_the_reference_asset():
    text = open(os.path.join(ASSETS, "attestations.csv")).read()
    atts = pa.read_attestations_csv(text)
    assert len(atts) == 1
    a = atts[0]
    assert a.about.hex() == "70997970c51812dc3a010c7d01b50e0d17dc79c8" and a.value == 5 and a.rec_id == 1
    assert a.domain == bytes(20) and a.message == bytes(32)
    assert a.sig_r.hex() == "1cc60bf64279111e233842283422f63cc34235adb4622a905aa325b6c8c44c4f"
    assert pa.write_attestations_csv(atts) == text
    for bad in ("nonsense\n", text.replace(",5,", ",256,"), text.replace("0x7099", "0x70zz"), text.replace("c8,", "c8ff,")):
        with pytest.raises(pa.EigenTrustError) as e:    # storage.rs:286-331
            pa.read_attestations_csv(bad)
        assert e.value.code == _ffi.ET_E_PARSE


def test_scores_csv_of_the_reference_asset_from_packed_scores():
    text = open(os.path.join(ASSETS, "scores.csv")).read()
    scores = [pa.pack_score(addr, 1000, Fraction(1000, 1)) for addr in
              ("0x70997970c51812dc3a010c7d01b50e0d17dc79c8", "0xf39fd6e51aad88f6f4ce6ab8827279cfffb92266")]
    assert pa.write_scores_csv(scores) == text
    assert scores[0].score_fr == (1000).to_bytes(32, "big") and scores[0].rational == 1000


def test_score_packaging():
    rng = random.Random(9)
    for _ in range(100):
        num, den = rng.randrange(1, 2**rng.randrange(1, 257)), rng.randrange(1, 2**rng.randrange(1, 257))
        ratio = Fraction(num, den)
        fr = ref.rational_to_fr(ratio)
        s = pa.pack_score(bytes(range(20)), fr, ratio)
        assert s.score_fr == fr.to_bytes(32, "big")                       # lib.rs:215-216
        assert int.from_bytes(s.score_rat[0], "big") == ratio.numerator   # :218-226
        assert int.from_bytes(s.score_rat[1], "big") == ratio.denominator
        assert int.from_bytes(s.score_hex, "big") == ratio.numerator // ratio.denominator   # :220
    with pytest.raises(pa.EigenTrustError) as e:     # lib.rs:222-229 panics: 33-byte numerator
        pa.pack_score(bytes(20), 5, Fraction(2**256 + 1, 2))
    assert e.value.code == _ffi.ET_E_RATIONAL_TOO_LARGE


def test_public_inputs_layout():
    rng = random.Random(3)
    n = 4
    pi = pa.ETPublicInputs([rng.randrange(R) for _ in range(n)], [rng.randrange(R) for _ in range(n)], rng.randrange(R), rng.randrange(R))
    raw = pi.to_bytes()
    assert len(raw) == (2 * n + 2) * 32
    assert raw == b"".join(v.to_bytes(32, "little") for v in pi.to_vec())          # circuit.rs:114-123
    back = pa.ETPublicInputs.from_bytes(raw, n)
    assert back.to_vec() == pi.to_vec()
    with pytest.raises(pa.EigenTrustError) as e:                                     # circuit.rs:128-132
        pa.ETPublicInputs.from_bytes(raw[:-1], n)
    assert e.value.code == _ffi.ET_E_PARSE
    with pytest.raises(pa.EigenTrustError) as e:
        pa.ETPublicInputs.from_bytes(raw[:-32] + b"\xff" * 32, n)
    assert e.value.code == _ffi.ET_E_NONCANONICAL_FR
