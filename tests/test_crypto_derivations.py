"""The algebra behind two shortcuts of the recovery kernel (protocol_amd/csrc/ecdsa.hip), checked with Python integers:
the sparse form of Poseidon's partial rounds, and the endomorphism constants that split a secp256k1 scalar."""
import random

import crypto_ref as c

R, T, RF, RP = c.R, c.T, c.RF, c.RP


def _matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(len(B))) % R for j in range(len(B[0]))] for i in range(len(A))]


def _matvec(A, v):
    return [sum(A[i][j] * v[j] for j in range(len(v))) % R for i in range(len(A))]


def _inverse(A):
    n = len(A)
    a = [row[:] + [int(i == j) for j in range(n)] for i, row in enumerate(A)]
    for col in range(n):
        p = next(r for r in range(col, n) if a[r][col] % R)
        a[col], a[p] = a[p], a[col]
        iv = pow(a[col][col], R - 2, R)
        a[col] = [x * iv % R for x in a[col]]
        for r in range(n):
            if r != col and a[r][col]:
                f = a[r][col]
                a[r] = [(x - f * y) % R for x, y in zip(a[r], a[col])]
    return [row[n:] for row in a]


def test_sparse_partial_rounds_equal_the_plain_permutation():
    """M_k = S_k * P_k with S_k sparse (first row, first column, identity) and P_k = diag(1, lower block): P_k commutes
    with the partial S-box and joins the matrix of the round before; constants are carried through the sparse
    matrices.  The kernel's host code derives the same matrices from the generated parameters."""
    M = c._MDS
    RC = [c._RC[k * T:(k + 1) * T] for k in range(RF + RP)]
    H = RF // 2
    cur, sparse, P = M, [None] * RP, [None] * RP
    for k in range(RP - 1, -1, -1):
        low = [row[1:] for row in cur[1:]]
        low_inv = _inverse(low)
        u = [sum(cur[0][1 + i] * low_inv[i][j] for i in range(T - 1)) % R for j in range(T - 1)]
        sparse[k] = ([cur[0][0]] + u, [cur[i][0] for i in range(1, T)])
        Pk = [[int(i == j) for j in range(T)] for i in range(T)]
        for i in range(1, T):
            for j in range(1, T):
                Pk[i][j] = low[i - 1][j - 1]
        P[k] = Pk
        cur = _matmul(Pk, M)
    pre = cur

    def sp(k, z):
        row0, col = sparse[k]
        return [sum(row0[j] * z[j] for j in range(T)) % R] + [(col[i - 1] * z[0] + z[i]) % R for i in range(1, T)]

    first = _matvec(P[0], RC[H])
    const0, carry = [0] * RP, [0] * T
    for k in range(1, RP):
        eff = [(a + b) % R for a, b in zip(_matvec(P[k], RC[H + k]), carry)]
        const0[k] = eff[0]
        carry = sp(k, [0] + eff[1:])

    def permute(state):
        st = [s % R for s in state]
        for rnd in range(H):
            st = [pow((s + RC[rnd][i]) % R, 5, R) for i, s in enumerate(st)]
            st = _matvec(pre if rnd == H - 1 else M, st)
        st = [(s + first[i]) % R for i, s in enumerate(st)]
        for k in range(RP):
            if k:
                st[0] = (st[0] + const0[k]) % R
            st[0] = pow(st[0], 5, R)
            st = sp(k, st)
        st = [(s + carry[i]) % R for i, s in enumerate(st)]
        for rnd in range(H + RP, RF + RP):
            st = [pow((s + RC[rnd][i]) % R, 5, R) for i, s in enumerate(st)]
            st = _matvec(M, st)
        return st

    rng = random.Random(1)
    for _ in range(4):
        x = [rng.randrange(R) for _ in range(T)]
        assert permute(x) == c.poseidon_permute(x)


def test_endomorphism_constants_split_scalars():
    """The constants written into the kernel's table: beta^3 = 1 mod p, lambda^3 = 1 mod n,
    lambda * G = (beta * Gx, Gy); (a1, b1), (a2, b2) lie in the lattice x + y * lambda = 0 mod n; the rounded
    multipliers split every scalar into two parts of at most 128 bits."""
    p, n, G = c.SECP_P, c.SECP_N, c.G
    beta = 0x7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee
    lam = 0x5363ad4cc05c30e0a5261c028812645a122e22ea20816678df02967c1b23bd72
    a1, nb1 = 0x3086d221a7d46bcde86c90e49284eb15, 0xe4437ed6010e88286f547fa90abfe4c3
    a2, b2 = 0x114ca50f7a8e2f3f657c1108d9d44cfd8, 0x3086d221a7d46bcde86c90e49284eb15
    g1 = 0x3086d221a7d46bcde86c90e49284eb153daa8a1471e8ca7fe893209a45dbb031
    g2 = 0xe4437ed6010e88286f547fa90abfe4c4221208ac9df506c61571b4ae8ac47f71
    assert pow(beta, 3, p) == 1 and beta != 1 and pow(lam, 3, n) == 1 and lam != 1
    assert c._mul(G, lam) == (beta * G[0] % p, G[1])
    assert (a1 - nb1 * lam) % n == 0 and (a2 + b2 * lam) % n == 0
    assert g1 == (2 * (b2 << 384) + n) // (2 * n) and g2 == (2 * (nb1 << 384) + n) // (2 * n)
    rng = random.Random(5)
    for _ in range(2000):
        k = rng.randrange(n)
        c1 = (k * g1 + (1 << 383)) >> 384
        c2 = (k * g2 + (1 << 383)) >> 384
        k1 = k - c1 * a1 - c2 * a2
        k2 = c1 * nb1 - c2 * b2
        assert (k1 + k2 * lam - k) % n == 0
        assert abs(k1).bit_length() <= 128 and abs(k2).bit_length() <= 128
