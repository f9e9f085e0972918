"""Opinion-validation crypto through the C ABI, without a GPU: Poseidon 5x5 and its sponge, Keccak-256,
attestation hash, secp256k1 recovery, Opinion::validate.  Checked against the reference's own known
answers (poseidon/native/mod.rs:122-146; the CLI asset whose single record must recover to the
attester scores.csv lists) and the Python restatement oracle/crypto_ref.py."""
import os
import random

import pytest

import crypto_ref as cref
import protocol_amd as pa
from protocol_amd import _ffi
from conftest import GOLDEN

R = pa.R_MODULUS
ASSETS = os.path.join(GOLDEN, "cli_assets")
ANVIL = [(0xac0974bec39a17e36ba4a6b4d238ff944bacb478cbed5efcae784d7bf4f2ff80, "f39fd6e51aad88f6f4ce6ab8827279cfffb92266"),
         (0x59c6995e998f97a5a0044966f0945389dc9e86dae88c7a8412f4603b6b78690d, "70997970c51812dc3a010c7d01b50e0d17dc79c8")]


def test_poseidon_known_answer_of_the_reference():
    out = pa.poseidon5_permute([0, 1, 2, 3, 4])   # poseidon/native/mod.rs:122-146
    assert out == [0x299c867db6c1fdd79dcefa40e4510b9837e60ebb1ce0663dbaa525df65250465,
                   0x1148aaef609aa338b27dafd89bb98862d8bb2b429aceac47d86206154ffe053d,
                   0x24febb87fed7462e23f6665ff9a0111f4044c38ee1672c1ac6b0637d34f24907,
                   0x0eb08f6d809668a981c186beaf6110060707059576406b248e5d9cf6e78b3d3e,
                   0x07748bc6877c9b82c8b98666ee9d0626ec7f5be4205f79ee8528ef1c4a376fc7]
    assert cref.poseidon_permute([0, 1, 2, 3, 4]) == out


def test_poseidon_and_sponge_match_restatement():
    rng = random.Random(1)
    for _ in range(20):
        st = [rng.randrange(R) for _ in range(5)]
        assert pa.poseidon5_permute(st) == cref.poseidon_permute(st)
    for n in (0, 1, 4, 5, 6, 10, 13):
        v = [rng.randrange(R) for _ in range(n)]
        assert pa.poseidon5_sponge(v) == cref.sponge(v)


def test_keccak256():
    assert pa.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert pa.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    rng = random.Random(2)
    for n in (1, 55, 135, 136, 137, 272, 1000):
        data = bytes(rng.randrange(256) for _ in range(n))
        assert pa.keccak256(data) == cref.keccak256(data)


def test_public_keys_and_addresses():
    for secret, addr in ANVIL + [(1, "7e5f4552091a69125d5dfcb7b8c2659029395bdf")]:
        key, got = pa.public_key_of_secret(secret)
        assert got.hex() == addr
        x, y = cref.public_key(secret)
        assert key == x.to_bytes(32, "little") + y.to_bytes(32, "little")


def test_cli_asset_record_recovers_to_its_attester():
    atts = pa.read_attestations_csv(open(os.path.join(ASSETS, "attestations.csv")).read())
    assert [a.hex() for a in pa.recover_attesters(atts)] == [ANVIL[0][1]]
    a = atts[0]
    assert pa.attestation_hash(a) == cref.attestation_hash(a.about, a.domain, a.value, a.message)


def signed(secret, about, value, rng, domain=bytes(20), message=bytes(32)):
    h = cref.attestation_hash(about, domain, value, message)
    r, s, odd = cref.sign(secret, h, rng.randrange(1, cref.SECP_N))
    return pa.SignedAttestationRaw(about, domain, value, message, r.to_bytes(32, "little"), s.to_bytes(32, "little"), odd)


def test_recovery_of_fresh_signatures():
    rng = random.Random(3)
    atts, want = [], []
    for _ in range(12):
        secret = rng.randrange(1, cref.SECP_N)
        about = bytes(rng.randrange(256) for _ in range(20))
        msg = bytes(rng.randrange(256) for _ in range(32)) if rng.random() < 0.5 else bytes(32)
        atts.append(signed(secret, about, rng.randrange(256), rng, bytes([7]) * 20, msg))
        want.append(cref.address_of(cref.public_key(secret)))
    assert pa.recover_attesters(atts) == want
    bad = atts[0]
    bad.sig_r = (cref.SECP_N + 5).to_bytes(32, "little")           # ecdsa/native.rs:202 Fq::from_bytes(..).unwrap()
    with pytest.raises(pa.EigenTrustError) as e:
        pa.recover_attesters([bad])
    assert e.value.code == _ffi.ET_E_BAD_SIGNATURE
    bad.sig_r = (5).to_bytes(32, "little")                         # x = 5 is not on the curve (:316 unwrap)
    with pytest.raises(pa.EigenTrustError) as e:
        pa.recover_attesters([bad])
    assert e.value.code == _ffi.ET_E_BAD_SIGNATURE


def test_opinion_validate_matches_restatement():
    rng = random.Random(4)
    nn = 5
    secrets = [rng.randrange(1, cref.SECP_N) for _ in range(3)]
    addrs = [cref.address_of(cref.public_key(s)) for s in secrets] + [bytes(rng.randrange(256) for _ in range(20)), bytes(20)]
    domain = bytes([3]) * 20
    me = secrets[0]
    row, entries = [], []
    for i, about in enumerate(addrs):
        if i == 0 or i == 2:
            row.append(None)                                       # filled with empty_with_about (dynamic_sets/native.rs:209-219)
            entries.append(None)
            continue
        a = signed(me if i != 3 else secrets[1], about, 10 + i, rng, domain)   # entry 3 is signed by someone else: invalid
        row.append(a)
        entries.append((a.value, a.message, int.from_bytes(a.sig_r, "little"), int.from_bytes(a.sig_s, "little")))
    key, _ = pa.public_key_of_secret(me)
    scores, op_hash = pa.opinion_validate(key, row, addrs, domain)
    want_scores, want_hash = cref.opinion_validate(cref.public_key(me), entries, addrs, domain)
    assert scores == want_scores == [0, 11, 0, 0, 0]               # slot 4 is empty: a valid signature still scores 0 (:89-90)
    assert op_hash == want_hash
    # the reference's asserts
    with pytest.raises(pa.EigenTrustError) as e:                   # opinion/native.rs:67
        pa.opinion_validate(pa.public_key_of_secret(12345)[0], row, addrs, domain)
    assert e.value.code == _ffi.ET_E_NOT_MEMBER
    with pytest.raises(pa.EigenTrustError) as e:                   # :76
        pa.opinion_validate(key, row, addrs, bytes(20))
    assert e.value.code == _ffi.ET_E_BAD_DOMAIN
    swapped = list(row)
    swapped[1], swapped[3] = row[3], row[1]
    with pytest.raises(pa.EigenTrustError) as e:                   # :75
        pa.opinion_validate(key, swapped, addrs, domain)
    assert e.value.code == _ffi.ET_E_BAD_ABOUT


def test_device_lane_code_compiled_for_the_host_matches_host_recovery():
    """The GPU kernel's lane function (ecdsa.hip) built for the CPU: same source, same answers."""
    import ctypes as C
    rng = random.Random(21)
    recs = [pa.sign_attestation(rng.randrange(1, cref.SECP_N), bytes(rng.randrange(256) for _ in range(20)), rng.randrange(256),
                                rng.randrange(1, cref.SECP_N), bytes([rng.randrange(256)]) * 20,
                                bytes([0xff] * 32) if k % 3 == 0 else bytes(32)) for k in range(24)]
    recs += pa.read_attestations_csv(open(os.path.join(ASSETS, "attestations.csv")).read())
    want_addr, want_keys = pa.recover_attesters(recs, with_keys=True)
    for rec, addr, key in zip(recs, want_addr, want_keys):
        c = rec._to_c()
        a, k = (C.c_uint8 * 20)(), (C.c_uint8 * 64)()
        _ffi.check(pa.lib().et_attestation_recover_lane_host(C.byref(c), a, k))
        assert bytes(a) == addr and bytes(k) == key
