"""Host-side neighbours of the converge path, with the reference's names: the threshold check
(eigentrust-zk/src/circuits/threshold/native.rs:33-96), score records and the client's CSV files
(eigentrust/src/storage.rs:184-307, eigentrust/src/lib.rs:201-236), the ET public-input layout
(eigentrust/src/circuit.rs:84-151).  Thin wrappers over the C ABI; nothing is computed in Python.
"""
import ctypes as C
from dataclasses import dataclass
from fractions import Fraction

import numpy as np

from . import _ffi
from ._ffi import check, lib

NUM_DECIMAL_LIMBS = 2   # circuits/mod.rs:53
POWER_OF_TEN = 72       # circuits/mod.rs:55


def _le(v):
    n = max(1, (int(v).bit_length() + 7) // 8)
    return (C.c_uint8 * n).from_buffer_copy(int(v).to_bytes(n, "little")), n


def _fr(v):
    return _ffi.EtFr.from_buffer_copy(int(v).to_bytes(32, "little"))


def _addr(a):
    if isinstance(a, str):
        a = bytes.fromhex(a[2:] if a.startswith("0x") else a)
    elif isinstance(a, int):
        a = a.to_bytes(20, "big")
    if len(a) != 20:
        raise ValueError("an address has 20 bytes")
    return (C.c_uint8 * 20).from_buffer_copy(bytes(a))


def big_to_fe_rat(ratio, num_limbs=NUM_DECIMAL_LIMBS, power_of_ten=POWER_OF_TEN):
    """utils.rs:332-354: decimal limbs (least significant first) of the scaled numerator and denominator."""
    num, nn = _le(ratio.numerator)
    den, dn = _le(ratio.denominator)
    out_n = (_ffi.EtFr * num_limbs)()
    out_d = (_ffi.EtFr * num_limbs)()
    check(lib().et_ratio_decompose(num, nn, den, dn, num_limbs, power_of_ten, out_n, out_d))
    conv = lambda arr: [int.from_bytes(bytes(e.b), "little") for e in arr]
    return conv(out_n), conv(out_d)


class Threshold:
    """Threshold::new(score, ratio, threshold).check_threshold() (threshold/native.rs:33-96)."""

    def __init__(self, score, ratio, threshold, num_limbs=NUM_DECIMAL_LIMBS, power_of_ten=POWER_OF_TEN,
                 num_neighbours=4, initial_score=1000):
        self.score, self.ratio, self.threshold = int(score), Fraction(ratio), int(threshold)
        self.num_limbs, self.power_of_ten = num_limbs, power_of_ten
        self.num_neighbours, self.initial_score = num_neighbours, initial_score

    def check_threshold(self):
        num, nn = _le(self.ratio.numerator)
        den, dn = _le(self.ratio.denominator)
        passed = C.c_int(0)
        s, t = _fr(self.score), _fr(self.threshold)
        check(lib().et_threshold_check(C.byref(s), num, nn, den, dn, C.byref(t), self.num_limbs, self.power_of_ten,
                                       self.num_neighbours, self.initial_score & (2**64 - 1), self.initial_score >> 64,
                                       C.byref(passed)))
        return bool(passed.value)


def scalar_from_address(address):
    """eth.rs:77-95."""
    out = _ffi.EtFr()
    a = _addr(address)
    check(lib().et_scalar_from_address(a, C.byref(out)))
    return int.from_bytes(bytes(out.b), "little")


@dataclass
class Score:
    """eigentrust/src/lib.rs Score: big-endian byte strings as the reference stores them."""
    address: bytes
    score_fr: bytes
    score_rat: tuple
    score_hex: bytes

    @property
    def rational(self):
        return Fraction(int.from_bytes(self.score_rat[0], "big"), int.from_bytes(self.score_rat[1], "big"))

    @classmethod
    def _from_c(cls, c):
        return cls(bytes(c.address), bytes(c.score_fr), (bytes(c.numerator), bytes(c.denominator)), bytes(c.score))

    def _to_c(self):
        c = _ffi.EtScore()
        for name, val in (("address", self.address), ("score_fr", self.score_fr), ("numerator", self.score_rat[0]),
                          ("denominator", self.score_rat[1]), ("score", self.score_hex)):
            C.memmove(getattr(c, name), val, len(val))
        return c


def pack_score(address, score_fr, ratio):
    """lib.rs:207-233 for one participant; raises (ET_E_RATIONAL_TOO_LARGE) where the reference panics."""
    num, nn = _le(ratio.numerator)
    den, dn = _le(ratio.denominator)
    out = _ffi.EtScore()
    s = _fr(score_fr)
    check(lib().et_score_pack(_addr(address), C.byref(s), num, nn, den, dn, C.byref(out)))
    return Score._from_c(out)


@dataclass
class SignedAttestationRaw:
    """attestation.rs:283-330 flattened: the record of attestations.csv."""
    about: bytes
    domain: bytes
    value: int
    message: bytes
    sig_r: bytes
    sig_s: bytes
    rec_id: int

    def _to_c(self):
        c = _ffi.EtAttestation()
        for name in ("about", "domain", "message", "sig_r", "sig_s"):
            val = getattr(self, name)
            C.memmove(getattr(c, name), val, len(val))
        c.value, c.rec_id = self.value, self.rec_id
        return c


def _c_attestations(atts):
    arr = (_ffi.EtAttestation * max(1, len(atts)))()
    for i, a in enumerate(atts):
        arr[i] = a._to_c()
    return arr


def read_attestations_csv(text):
    """storage.rs:247-307 (CSVFileStorage::<AttestationRecord>::load + TryFrom)."""
    raw = text.encode() if isinstance(text, str) else bytes(text)
    count = C.c_size_t(0)
    check(lib().et_attestations_csv_parse(raw, len(raw), None, 0, C.byref(count)))
    arr = (_ffi.EtAttestation * max(1, count.value))()
    check(lib().et_attestations_csv_parse(raw, len(raw), arr, count.value, C.byref(count)))
    return [SignedAttestationRaw(bytes(a.about), bytes(a.domain), int(a.value), bytes(a.message), bytes(a.sig_r),
                                 bytes(a.sig_s), int(a.rec_id)) for a in arr[:count.value]]


def _format(fn, arr, n):
    needed = C.c_size_t(0)
    rc = fn(arr, n, None, 0, C.byref(needed))
    if rc not in (_ffi.ET_OK, _ffi.ET_E_BUFFER_TOO_SMALL):
        check(rc)
    buf = C.create_string_buffer(needed.value)
    check(fn(arr, n, buf, needed.value, C.byref(needed)))
    return buf.value.decode()


def write_attestations_csv(atts):
    return _format(lib().et_attestations_csv_format, _c_attestations(atts), len(atts))


def write_scores_csv(scores):
    """storage.rs:184-217 (ScoreRecord::from_score + CSV save)."""
    arr = (_ffi.EtScore * max(1, len(scores)))()
    for i, s in enumerate(scores):
        arr[i] = s._to_c()
    return _format(lib().et_scores_csv_format, arr, len(scores))


def poseidon5_permute(state):
    """poseidon/native/mod.rs:34-97 on five field elements."""
    inp = (_ffi.EtFr * 5)(*[_fr(v) for v in state])
    out = (_ffi.EtFr * 5)()
    check(lib().et_poseidon5_permute(inp, out))
    return [int.from_bytes(bytes(e.b), "little") for e in out]


def poseidon5_sponge(values):
    """poseidon/native/sponge.rs:46-68."""
    n = len(values)
    inp = (_ffi.EtFr * max(1, n))(*[_fr(v) for v in values])
    out = _ffi.EtFr()
    check(lib().et_poseidon5_sponge(inp, n, C.byref(out)))
    return int.from_bytes(bytes(out.b), "little")


def keccak256(data):
    out = (C.c_uint8 * 32)()
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\0")
    check(lib().et_keccak256(buf, len(data), out))
    return bytes(out)


def attestation_hash(att):
    out = _ffi.EtFr()
    c = att._to_c()
    check(lib().et_attestation_hash(C.byref(c), C.byref(out)))
    return int.from_bytes(bytes(out.b), "little")


def recover_attesters(attestations, with_keys=False, gpu=False):
    """SignedAttestationEth::recover_public_key + address_from_ecdsa_key for every record, on host
    threads or (`gpu=True`) one GPU lane per record."""
    n = len(attestations)
    arr = attestations if isinstance(attestations, C.Array) else _c_attestations(attestations)
    who = (C.c_uint8 * max(20, 20 * n))()
    keys = (C.c_uint8 * max(64, 64 * n))()
    if gpu:
        check(lib().et_attestations_recover_gpu(arr, n, who, keys, -1))
    else:
        check(lib().et_attestations_recover(arr, n, who, keys))
    addrs = [bytes(who[20 * k:20 * k + 20]) for k in range(n)]
    if not with_keys:
        return addrs
    return addrs, [bytes(keys[64 * k:64 * k + 64]) for k in range(n)]


def public_key_of_secret(secret):
    """(64-byte key: x, y little-endian; 20-byte address) of a secp256k1 secret."""
    sk = (C.c_uint8 * 32).from_buffer_copy(int(secret).to_bytes(32, "little"))
    pk, ad = (C.c_uint8 * 64)(), (C.c_uint8 * 20)()
    check(lib().et_secp256k1_public_key(sk, pk, ad))
    return bytes(pk), bytes(ad)


def sign_attestation(secret, about, value, nonce, domain=bytes(20), message=bytes(32)):
    """A signed record as the reference's client produces it (attestation.rs:117-140 over
    ecdsa/native.rs:264-296), with the caller's nonce."""
    att = SignedAttestationRaw(bytes(_addr(about)), bytes(_addr(domain)), int(value), bytes(message), bytes(32), bytes(32), 0)
    h = _fr(attestation_hash(att))
    sk = (C.c_uint8 * 32).from_buffer_copy(int(secret).to_bytes(32, "little"))
    k = (C.c_uint8 * 32).from_buffer_copy(int(nonce).to_bytes(32, "little"))
    r, s, rec = (C.c_uint8 * 32)(), (C.c_uint8 * 32)(), C.c_uint8(0)
    check(lib().et_secp256k1_sign(sk, C.byref(h), k, r, s, C.byref(rec)))
    att.sig_r, att.sig_s, att.rec_id = bytes(r), bytes(s), int(rec.value)
    return att


def opinion_validate(public_key, attestations, set_addresses, domain=bytes(20)):
    """Opinion::validate (opinion/native.rs:63-109): `attestations[i]` is a SignedAttestationRaw or
    None (filled with empty_with_about as update_op does).  Returns (scores, op_hash)."""
    nn = len(set_addresses)
    arr = (_ffi.EtAttestation * nn)()
    present = (C.c_uint8 * nn)()
    for i, a in enumerate(attestations):
        if a is not None:
            arr[i] = a._to_c()
            present[i] = 1
    addrs = (C.c_uint8 * (20 * nn))()
    for i, a in enumerate(set_addresses):
        C.memmove(C.byref(addrs, 20 * i), bytes(_addr(a)), 20)
    pk = (C.c_uint8 * 64).from_buffer_copy(bytes(public_key))
    scores = (_ffi.EtFr * nn)()
    op_hash = _ffi.EtFr()
    check(lib().et_opinion_validate(pk, arr, present, addrs, nn, _addr(domain), scores, C.byref(op_hash)))
    conv = lambda e: int.from_bytes(bytes(e.b), "little")
    return [conv(e) for e in scores], conv(op_hash)


def et_circuit_setup(attestations, domain=bytes(20), num_neighbours=4, num_iterations=20, initial_score=1000):
    """Client::et_circuit_setup without the circuit (lib.rs:339-466): signatures are recovered and
    verified, opinions validated, converge runs on the GPU.  Returns (scores, ETPublicInputs)."""
    n = len(attestations)
    arr = _c_attestations(attestations)
    count = C.c_size_t(0)
    dom = _addr(domain)
    args = (arr, n, dom, num_neighbours, num_iterations, initial_score & (2**64 - 1), initial_score >> 64)
    check(lib().et_et_setup(*args, None, 0, C.byref(count), None, 0))
    out = (_ffi.EtScore * max(1, count.value))()
    pub = (C.c_uint8 * ((2 * num_neighbours + 2) * 32))()
    check(lib().et_et_setup(*args, out, count.value, C.byref(count), pub, len(pub)))
    return [Score._from_c(c) for c in out[:count.value]], ETPublicInputs.from_bytes(bytes(pub), num_neighbours)


def calculate_scores(attestations, attesters=None, domain=bytes(20), num_neighbours=4, num_iterations=20, initial_score=1000):
    """Client::calculate_scores (lib.rs:201-236).  With `attesters=None` the attester of every record
    is recovered from its signature and every opinion is verified, as in the reference; a list of
    addresses (`attesters[k]` for attestation k) skips recovery and verification.
    Runs converge on the GPU and converge_rational on the host."""
    if attesters is None:
        return et_circuit_setup(attestations, domain, num_neighbours, num_iterations, initial_score)[0]
    n = len(attestations)
    arr = _c_attestations(attestations)
    who = (C.c_uint8 * max(20, 20 * n))()
    for k, a in enumerate(attesters):
        C.memmove(C.byref(who, 20 * k), bytes(_addr(a)), 20)
    count = C.c_size_t(0)
    dom = _addr(domain)
    args = (arr, who, n, dom, num_neighbours, num_iterations, initial_score & (2**64 - 1), initial_score >> 64)
    check(lib().et_local_scores(*args, None, 0, C.byref(count)))
    out = (_ffi.EtScore * max(1, count.value))()
    check(lib().et_local_scores(*args, out, count.value, C.byref(count)))
    return [Score._from_c(c) for c in out[:count.value]]


def scores_from_attestations(attestations, domain=bytes(20), num_iterations=20, initial_score=1000, f64=False):
    """calculate_scores for attestation lists of any size, all on the GPU except the address bookkeeping:
    returns (addresses ascending, field scores as integers[, f64 scores])."""
    n = len(attestations)
    arr = attestations if isinstance(attestations, C.Array) else _c_attestations(attestations)
    count = C.c_size_t(0)
    dom = _addr(domain)
    head = (arr, n, dom, num_iterations, initial_score & (2**64 - 1), initial_score >> 64)
    # one call (a size query would recover every signature a second time): at most 2 participants per record
    cap = max(2, 2 * n)
    addrs = (C.c_uint8 * (20 * cap))()
    fr = np.zeros((cap, 32), dtype=np.uint8)
    d64 = np.zeros(cap, dtype=np.float64) if f64 else None
    check(lib().et_scores_from_attestations(*head, addrs, fr.ctypes.data, d64.ctypes.data if f64 else None, cap, C.byref(count)))
    m = count.value
    fr = fr[:m]
    d64 = d64[:m] if f64 else None
    out_addr = [bytes(addrs[20 * i:20 * i + 20]) for i in range(m)]
    ints = [int.from_bytes(fr[i].tobytes(), "little") for i in range(m)]
    return (out_addr, ints, d64) if f64 else (out_addr, ints)


class ETPublicInputs:
    """circuit.rs:84-151."""

    def __init__(self, participants, scores, domain, opinion_hash):
        self.participants, self.scores = [int(p) for p in participants], [int(s) for s in scores]
        self.domain, self.opinion_hash = int(domain), int(opinion_hash)

    def to_vec(self):
        return self.participants + self.scores + [self.domain, self.opinion_hash]

    def to_bytes(self):
        n = len(self.participants)
        pa = (_ffi.EtFr * max(1, n))(*[_fr(p) for p in self.participants])
        sc = (_ffi.EtFr * max(1, n))(*[_fr(s) for s in self.scores])
        d, h = _fr(self.domain), _fr(self.opinion_hash)
        out = (C.c_uint8 * ((2 * n + 2) * 32))()
        check(lib().et_public_inputs_pack(pa, sc, n, C.byref(d), C.byref(h), out, len(out)))
        return bytes(out)

    @classmethod
    def from_bytes(cls, data, participants):
        n = participants
        pa, sc = (_ffi.EtFr * max(1, n))(), (_ffi.EtFr * max(1, n))()
        d, h = _ffi.EtFr(), _ffi.EtFr()
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) or b"\0")
        check(lib().et_public_inputs_unpack(buf, len(data), n, pa, sc, C.byref(d), C.byref(h)))
        conv = lambda e: int.from_bytes(bytes(e.b), "little")
        return cls([conv(e) for e in pa[:n]], [conv(e) for e in sc[:n]], conv(d), conv(h))
