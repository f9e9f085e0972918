"""CPU restatement of the reference's native EigenTrust set -- TEST INFRASTRUCTURE ONLY.

This module is the checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product package never does.

It restates, with Python integers (mod r) and fractions.Fraction, exactly what
/root/reference/eigentrust-zk/src/circuits/dynamic_sets/native.rs does:

  EigenTrustSet.new              native.rs:165-172
  EigenTrustSet.add_member       native.rs:175-186
  EigenTrustSet.remove_member    native.rs:189-198
  EigenTrustSet.update_op        native.rs:201-231  (takes the row that Opinion::validate,
                                 opinion/native.rs:63-109, would have produced: one field
                                 element per slot, already zero where the attestation is invalid)
  EigenTrustSet.filter_peers_ops native.rs:234-283
  EigenTrustSet.converge         native.rs:286-337
  EigenTrustSet.converge_rational native.rs:340-392
  circuit_converge               dynamic_sets/mod.rs:469-689, the halo2 circuit's statement of the
                                 same filter / normalise / iterate, as a second, independent check

Field arithmetic: the reference computes in halo2curves 0.3.2 `bn256::Fr`
(Cargo.lock:1768-1770), which is absent from /root/reference.  Its results are observable
only as canonical integers modulo r (utils.rs:327-329), so `int % R_MODULUS` is the same
function.  r is quoted at eigentrust-zk/src/params/rns/bn256.rs:6.

Pinned against: the CLI asset pair eigentrust-cli/assets/attestations.csv:2 ->
scores.csv:2-3, the conservation assert native.rs:331-334, the Fr == num/den invariant of
threshold/native.rs:75-81 and its literal known answers :135-226 (tests/test_oracle.py).
"""
from fractions import Fraction

# BN254 scalar field modulus (eigentrust-zk/src/params/rns/bn256.rs:6)
R_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def fr(x):
    return x % R_MODULUS


def fr_inv(x):
    """Field inverse; the reference's `invert().unwrap_or(ZERO)` maps 0 -> 0 (native.rs:308)."""
    x %= R_MODULUS
    if x == 0:
        return 0
    return pow(x, R_MODULUS - 2, R_MODULUS)


def fr_to_le_bytes(x):
    """Canonical 32-byte little-endian repr == Fr::to_repr() (utils.rs:327-329)."""
    return int(x % R_MODULUS).to_bytes(32, "little")


def fr_from_le_bytes(b):
    v = int.from_bytes(bytes(b), "little")
    if v >= R_MODULUS:
        raise ValueError("non-canonical field element")
    return v


class InsufficientPeers(AssertionError):
    pass


class EigenTrustSet:
    """Runtime-sized restatement of the const-generic reference struct (native.rs:109-131)."""

    def __init__(self, num_neighbours, num_iterations, initial_score, domain=0):
        self.n = int(num_neighbours)
        self.num_iterations = int(num_iterations)
        self.initial_score = int(initial_score)
        self.domain = fr(domain)
        # native.rs:167  set = [(0, 0); N]
        self.set = [(0, 0) for _ in range(self.n)]
        # native.rs:168  ops: HashMap<addr, Vec<N>>
        self.ops = {}

    # native.rs:175-186
    def add_member(self, addr):
        addr = fr(addr)
        assert all(a != addr for a, _ in self.set), "member already in set"  # :176-178
        idx = next((i for i, (a, _) in enumerate(self.set) if a == 0), None)  # :180
        assert idx is not None, "set is full"  # :181 unwrap
        self.set[idx] = (addr, fr(self.initial_score))  # :184-185

    # native.rs:189-198
    def remove_member(self, addr):
        addr = fr(addr)
        idx = next((i for i, (a, _) in enumerate(self.set) if a == addr), None)
        assert idx is not None, "not a member"  # :192
        self.set[idx] = (0, 0)  # :195
        self.ops.pop(addr, None)  # :197

    # native.rs:201-231 with Opinion::validate's output given directly
    def update_op(self, from_addr, scores):
        from_addr = fr(from_addr)
        assert any(a == from_addr for a, _ in self.set), "sender not in set"  # opinion/native.rs:66-67
        assert len(scores) == self.n
        # opinion/native.rs:92-98: a score about an empty slot is zeroed at validation time
        row = [0 if self.set[j][0] == 0 else fr(v) for j, v in enumerate(scores)]
        self.ops[from_addr] = row  # :228

    # native.rs:234-283
    def filter_peers_ops(self):
        n = self.n
        filtered = {}
        for i in range(n):
            addr_i = self.set[i][0]
            if addr_i == 0:
                continue
            ops_i = list(self.ops.get(addr_i, [0] * n))  # :244-245
            for j in range(n):
                addr_j = self.set[j][0]
                if addr_j == 0 or addr_j == addr_i:  # :254-259
                    ops_i[j] = 0
            s = 0
            for v in ops_i:
                s = (s + v) % R_MODULUS  # :263 (sum in the field)
            if s == 0:  # :264
                for j in range(n):
                    addr_j = self.set[j][0]
                    if addr_j != addr_i and addr_j != 0:  # :271-276
                        ops_i[j] = 1
            filtered[addr_i] = ops_i
        return filtered

    def _filtered_rows(self):
        n = self.n
        filtered = self.filter_peers_ops()
        rows = []
        for i in range(n):
            addr = self.set[i][0]
            rows.append([0] * n if addr == 0 else list(filtered[addr]))  # :294-302 / :344-353
        return rows

    # native.rs:286-337
    def converge(self):
        n = self.n
        valid = sum(1 for a, _ in self.set if a != 0)
        if valid < 2:  # :288-289
            raise InsufficientPeers("Insufficient peers for calculation!")
        ops = self._filtered_rows()
        ops_norm = [[0] * n for _ in range(n)]
        for i in range(n):  # :305-314
            inv = fr_inv(sum(ops[i]) % R_MODULUS)
            for j in range(n):
                ops_norm[i][j] = ops[i][j] * inv % R_MODULUS
        s = [sc for _, sc in self.set]  # :317
        for _ in range(self.num_iterations):  # :319-329
            new_s = [0] * n
            for i in range(n):
                acc = 0
                for j in range(n):
                    acc = (ops_norm[j][i] * s[j] + acc) % R_MODULUS
                new_s[i] = acc
            s = new_s
        sum_initial = sum(sc for _, sc in self.set) % R_MODULUS  # :332
        sum_final = sum(s) % R_MODULUS  # :333
        assert sum_initial == sum_final, "reputation leak"  # :334
        return s

    # native.rs:340-392
    def converge_rational(self):
        n = self.n
        ops = self._filtered_rows()  # field elements as integers (fe_to_big, :350-351)
        s = [Fraction(self.initial_score) for _ in range(n)]  # :362-364 (ALL slots)
        ops_norm = [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):  # :367-377
            tot = sum(ops[i])
            if tot == 0:
                tot = 1
            for j in range(n):
                ops_norm[i][j] = Fraction(ops[i][j], tot)
        for _ in range(self.num_iterations):  # :380-390
            new_s = [Fraction(0)] * n
            for i in range(n):
                acc = Fraction(0)
                for j in range(n):
                    acc = ops_norm[j][i] * s[j] + acc
                new_s[i] = acc
            s = new_s
        return s


def circuit_converge(set_addrs, ops, num_iterations, initial_score):
    """The reference's SECOND statement of the same algorithm: what the halo2 circuit constrains
    (eigentrust-zk/src/circuits/dynamic_sets/mod.rs:469-689), restated over plain field integers.
    `set_addrs`: N slot addresses (0 = empty); `ops`: N x N opinion values, row i = slot i's opinions
    (an empty slot's row is whatever the prover supplied; the circuit filters it like any other).

    It differs from native.rs only where a slot is empty: the circuit filters and redistributes the
    row of EVERY slot (mod.rs:469-600 loops over all i, so an empty slot's row sums to zero and
    becomes ones over the occupied slots), seeds every slot with INITIAL_SCORE (mod.rs:642) and
    fixes the total at INITIAL_SCORE * NUM_NEIGHBOURS (mod.rs:326-329, :674-689).  With every slot
    occupied the two statements must agree on every score."""
    n = len(set_addrs)
    rows = []
    for i in range(n):
        addr_i = set_addrs[i]
        ops_i = []
        for j in range(n):
            addr_j = set_addrs[j]
            # mod.rs:479-512: nullified when set_addr_j == 0 or set_addr_j == addr_i
            ops_i.append(0 if (addr_j == 0 or addr_j == addr_i) else fr(ops[i][j]))
        total = 0
        for v in ops_i:  # mod.rs:515-523
            total = (total + v) % R_MODULUS
        is_sum_zero = total == 0  # mod.rs:525-530
        for j in range(n):  # mod.rs:532-592
            addr_j = set_addrs[j]
            if addr_j != addr_i and addr_j != 0 and is_sum_zero:
                ops_i[j] = 1
        rows.append(ops_i)
    norm = []
    for i in range(n):  # mod.rs:602-638
        total = 0
        for v in rows[i]:
            total = (total + v) % R_MODULUS
        inv = fr_inv(total)
        norm.append([(v * inv) % R_MODULUS for v in rows[i]])
    s = [fr(initial_score)] * n  # mod.rs:642
    for _ in range(num_iterations):  # mod.rs:643-657
        new_s = [0] * n
        for i in range(n):
            for j in range(n):
                new_s[i] = (norm[j][i] * s[j] + new_s[i]) % R_MODULUS
        s = new_s
    # mod.rs:674-689: the sum of the scores is constrained to INITIAL_SCORE * NUM_NEIGHBOURS
    assert sum(s) % R_MODULUS == fr(initial_score * n), "total reputation changed"
    return s


def rational_to_fr(q):
    """num * den^-1 mod r: the identity threshold/native.rs:75-81 asserts between the two paths."""
    return q.numerator % R_MODULUS * fr_inv(q.denominator) % R_MODULUS


# ---- Threshold::check_threshold (threshold/native.rs:33-96), used to pin Fr == rational ----

def _decompose_big_decimal(x, num_limbs, power_of_ten):
    base = 10 ** power_of_ten
    limbs = []
    for _ in range(num_limbs):
        limbs.append(x % base)
        x //= base
    return limbs


def _compose_big_decimal_f(limbs, power_of_ten):
    base = pow(10, power_of_ten, R_MODULUS)
    val = 0
    for limb in reversed(limbs):
        val = (val * base + limb) % R_MODULUS
    return val


def check_threshold(score_fr, ratio, threshold, num_limbs, power_of_ten, num_neighbours,
                    initial_score):
    """threshold/native.rs:33-96.  Raises AssertionError where the reference would panic."""
    max_score = num_neighbours * initial_score
    max_limb = 10 ** power_of_ten - 1
    assert max_score * max_limb < R_MODULUS - 1  # :34-37
    num, den = ratio.numerator, ratio.denominator
    max_len = num_limbs * power_of_ten
    dig_len = len(str(max(num, den)))
    diff = max_len - dig_len
    assert diff >= 0  # usize underflow panics in the reference (:44)
    scale = 10 ** diff
    num_dec = _decompose_big_decimal(num * scale, num_limbs, power_of_ten)
    den_dec = _decompose_big_decimal(den * scale, num_limbs, power_of_ten)
    assert threshold < max_score  # :61-63
    composed_num = _compose_big_decimal_f(num_dec, power_of_ten)
    composed_den = _compose_big_decimal_f(den_dec, power_of_ten)
    assert composed_den != 0
    res = composed_num * fr_inv(composed_den) % R_MODULUS
    assert res == score_fr % R_MODULUS  # :81
    last_num, last_den = num_dec[-1], den_dec[-1]
    assert last_den != 0  # :90
    comp = last_den * threshold % R_MODULUS
    return last_num >= comp  # :95
