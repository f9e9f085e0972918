"""CPU restatement of the opinion-validation crypto, TEST INFRASTRUCTURE ONLY (Python integers).

Follows the reference:
  * Poseidon 5x5 permutation and sponge   eigentrust-zk/src/poseidon/native/mod.rs:34-97, native/sponge.rs:46-68
  * its parameters: the published Grain-LFSR generation for (prime field, n=254, t=5, R_F=8, R_P=60);
    the first constants, the MDS matrix and the known answer of poseidon/native/mod.rs:122-146 pin it
  * attestation scalars and hash          eigentrust/src/attestation.rs:81-115, dynamic_sets/native.rs:96-104
  * ECDSA sign / recover / verify         eigentrust-zk/src/ecdsa/native.rs:264-331, :382-395
  * Ethereum address of a key             ecdsa/native.rs:90-111, eigentrust/src/eth.rs:70-74
  * Opinion::validate                     circuits/opinion/native.rs:63-109
Pinned by: the Poseidon known answer above; Keccak-256 of the empty string and the address of secret
key 1 (public knowledge); the reference's CLI asset (attestations.csv:2 must recover to Anvil account 0,
the attester whose scores scores.csv lists).
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # params/rns/bn256.rs:6
SECP_P = 2**256 - 2**32 - 977
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
T, RF, RP = 5, 8, 60


def _params():
    def bits(v, w):
        return [(v >> (w - 1 - i)) & 1 for i in range(w)]
    st = bits(1, 2) + bits(0, 4) + bits(254, 12) + bits(T, 12) + bits(RF, 10) + bits(RP, 10) + [1] * 30

    def upd():
        nb = st[62] ^ st[51] ^ st[38] ^ st[23] ^ st[13] ^ st[0]
        st.pop(0)
        st.append(nb)
        return nb
    for _ in range(160):
        upd()

    def bit():
        b = upd()
        while b == 0:
            upd()
            b = upd()
        return upd()

    def rnd():
        v = 0
        for _ in range(254):
            v = (v << 1) | bit()
        return v
    rc = []
    while len(rc) < (RF + RP) * T:
        v = rnd()
        if v < R:
            rc.append(v)
    while True:
        xy = [rnd() % R for _ in range(2 * T)]
        if len(set(xy)) == 2 * T and all((xy[i] + xy[T + j]) % R for i in range(T) for j in range(T)):
            break
    mds = [[pow(xy[i] + xy[T + j], R - 2, R) for j in range(T)] for i in range(T)]
    return rc, mds


_RC, _MDS = _params()


def poseidon_permute(state):
    st = [s % R for s in state]
    k = 0
    for rnd in range(RF + RP):
        st = [(s + _RC[k + i]) % R for i, s in enumerate(st)]
        k += T
        full = rnd < RF // 2 or rnd >= RF // 2 + RP
        for i in range(T if full else 1):
            st[i] = pow(st[i], 5, R)
        st = [sum(_MDS[i][j] * st[j] for j in range(T)) % R for i in range(T)]
    return st


def sponge(inputs):
    inputs = list(inputs) or [0]
    state = [0] * T
    for off in range(0, len(inputs), T):
        chunk = inputs[off:off + T] + [0] * (T - len(inputs[off:off + T]))
        state = poseidon_permute([(c + s) % R for c, s in zip(chunk, state)])
    return state[0]


def keccak256(data):
    rc = [0x0000000000000001, 0x0000000000008082, 0x800000000000808a, 0x8000000080008000, 0x000000000000808b,
          0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008a, 0x0000000000000088,
          0x0000000080008009, 0x000000008000000a, 0x000000008000808b, 0x800000000000008b, 0x8000000000008089,
          0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800a, 0x800000008000000a,
          0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    m64 = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & m64 if n else v
    a = [[0] * 5 for _ in range(5)]
    data = bytearray(data)
    rate = 136
    data.append(0x01)
    while len(data) % rate:
        data.append(0)
    data[-1] |= 0x80
    for off in range(0, len(data), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(data[off + 8 * i:off + 8 * i + 8], "little")
        for rnd in range(24):
            c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
            b = [[0] * 5 for _ in range(5)]
            x, y = 1, 0
            b[0][0] = a[0][0]
            rot = 0
            for t in range(24):
                rot = (rot + t + 1) % 64
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], rot)
                x, y = y, (2 * x + 3 * y) % 5
            a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & m64 & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
            a[0][0] ^= rc[rnd]
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


# ---- secp256k1, affine, None = infinity ----
def _add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % SECP_P == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], SECP_P - 2, SECP_P) % SECP_P
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], SECP_P - 2, SECP_P) % SECP_P
    x = (lam * lam - p[0] - q[0]) % SECP_P
    return x, (lam * (p[0] - x) - p[1]) % SECP_P


def _mul(p, k):
    r = None
    while k:
        if k & 1:
            r = _add(r, p)
        p = _add(p, p)
        k >>= 1
    return r


def public_key(secret):
    return _mul(G, secret % SECP_N)


def address_of(pub):
    return keccak256(pub[0].to_bytes(32, "big") + pub[1].to_bytes(32, "big"))[12:]


def sign(secret, msg, k):
    """ecdsa/native.rs:264-296 with the caller's nonce k: (r, s, y_is_odd)."""
    pt = _mul(G, k)
    r = pt[0] % SECP_N
    s = pow(k, SECP_N - 2, SECP_N) * (msg + r * secret) % SECP_N
    return r, s, pt[1] & 1


def recover(r, s, y_odd, msg):
    """ecdsa/native.rs:298-331."""
    y = pow(r**3 + 7, (SECP_P + 1) // 4, SECP_P)
    assert y * y % SECP_P == (r**3 + 7) % SECP_P
    if (y & 1) != (1 if y_odd else 0):
        y = SECP_P - y
    ri = pow(r, SECP_N - 2, SECP_N)
    pk = _add(_mul(G, (-ri * msg) % SECP_N), _mul((r, y), ri * s % SECP_N))
    assert verify(r, s, msg, pk)
    return pk


def verify(r, s, msg, pk):
    """ecdsa/native.rs:382-395."""
    if not (0 < r < SECP_N and 0 < s < SECP_N) or pk is None:
        return False
    si = pow(s, SECP_N - 2, SECP_N)
    pt = _add(_mul(G, msg * si % SECP_N), _mul(pk, r * si % SECP_N))
    return pt is not None and pt[0] % SECP_N == r


# ---- attestations ----
def attestation_scalars(about, domain, value, message):
    """attestation.rs:81-115: 20-byte big-endian addresses and the 32-byte message as scalars."""
    return (int.from_bytes(about, "big"), int.from_bytes(domain, "big"), value, int.from_bytes(message, "big") % R)


def attestation_hash(about, domain, value, message):
    a, d, v, m = attestation_scalars(about, domain, value, message)
    return poseidon_permute([a, d, v, m, 0])[0]


def opinion_validate(pub, entries, set_addresses, domain):
    """opinion/native.rs:63-109.  entries[i] = (value, message, r, s) or None (empty_with_about);
    set_addresses[i] = 20 bytes (zero = empty slot).  Returns (scores, op_hash)."""
    assert address_of(pub) in set_addresses
    scores, hashes = [], []
    for about, e in zip(set_addresses, entries):
        value, message, r, s = e if e is not None else (0, bytes(32), 1, 1)
        h = attestation_hash(about, domain, value, message)
        ok = verify(r, s, h, pub) and about != bytes(20)
        scores.append(value if ok else 0)
        hashes.append(h if ok else 0)
    return scores, sponge(hashes)
