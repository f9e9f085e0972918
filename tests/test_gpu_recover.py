"""Batched attester recovery on the GPU (one lane per record) against the host implementation, the
Python restatement and the reference's CLI asset."""
import os
import random

import pytest

import crypto_ref as cref
import protocol_amd as pa
from protocol_amd import _ffi
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def make_records(count, seed):
    rng = random.Random(seed)
    recs, secrets = [], []
    for _ in range(count):
        secret = rng.randrange(1, cref.SECP_N)
        about = bytes(rng.randrange(256) for _ in range(20))
        domain = bytes(rng.randrange(256) for _ in range(20)) if rng.random() < 0.5 else bytes(20)
        # messages above the scalar field's modulus exercise the reduction of attestation.rs:107-113
        message = bytes([0xff] * 8 + [rng.randrange(256) for _ in range(24)]) if rng.random() < 0.3 else bytes(32)
        recs.append(pa.sign_attestation(secret, about, rng.randrange(256), rng.randrange(1, cref.SECP_N), domain, message))
        secrets.append(secret)
    return recs, secrets


def test_cli_asset_record_on_the_gpu():
    atts = pa.read_attestations_csv(open(os.path.join(GOLDEN, "cli_assets", "attestations.csv")).read())
    assert [a.hex() for a in pa.recover_attesters(atts, gpu=True)] == ["f39fd6e51aad88f6f4ce6ab8827279cfffb92266"]


def test_gpu_recovery_equals_host_recovery():
    recs, secrets = make_records(3000, 7)
    host_addr, host_keys = pa.recover_attesters(recs, with_keys=True)
    gpu_addr, gpu_keys = pa.recover_attesters(recs, with_keys=True, gpu=True)
    assert gpu_addr == host_addr and gpu_keys == host_keys
    for k in (0, 1, 2999):   # and both equal the restatement
        assert gpu_addr[k] == cref.address_of(cref.public_key(secrets[k]))


def test_gpu_recovery_rejects_what_the_reference_panics_on():
    recs, _ = make_records(70, 9)
    recs[33].sig_r = (cref.SECP_N + 1).to_bytes(32, "little")     # ecdsa/native.rs:202 unwrap
    with pytest.raises(pa.EigenTrustError) as e:
        pa.recover_attesters(recs, gpu=True)
    assert e.value.code == _ffi.ET_E_BAD_SIGNATURE and "attestation 33" in str(e.value)
    recs[33].sig_r = (5).to_bytes(32, "little")                   # :316 unwrap: 5^3 + 7 is not a square mod p
    with pytest.raises(pa.EigenTrustError) as e:
        pa.recover_attesters(recs, gpu=True)
    assert e.value.code == _ffi.ET_E_BAD_SIGNATURE and "attestation 33" in str(e.value)
    assert pa.recover_attesters([], gpu=True) == []
