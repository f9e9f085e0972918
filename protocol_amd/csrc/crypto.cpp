// Opinion validation crypto that feeds update_op in the reference, host-side:
//   - Poseidon 5x5 over bn256::Fr, x^5 S-box, 8 full + 60 partial rounds
//     (eigentrust-zk/src/poseidon/native/mod.rs:34-97) and its sponge (native/sponge.rs:46-68);
//     round constants and MDS matrix are the published Grain-LFSR parameters for
//     (field = prime, n = 254, t = 5, R_F = 8, R_P = 60), generated here at first use -- the
//     reference stores the same numbers as a table (params/hasher/poseidon_bn254_5x5.rs);
//   - attestation hash (circuits/dynamic_sets/native.rs:96-104, eigentrust/src/attestation.rs:81-115);
//   - secp256k1 ECDSA public-key recovery and verification (ecdsa/native.rs:298-331, :382-395),
//     Keccak-256 and the Ethereum address of a key (ecdsa/native.rs:90-111, eigentrust/src/eth.rs:70-74);
//   - Opinion::validate's scores and hashes (circuits/opinion/native.rs:63-109).
#include <array>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "engine.hpp"
#include "fr_host.hpp"
#include "mont256.hpp"
#include "crypto.hpp"

using namespace et;

namespace {


const Mont& FP() { static const Mont f(SECP_P); return f; }
const Mont& FN() { static const Mont f(SECP_N); return f; }

// Jacobian point on y^2 = x^3 + 7, coordinates in Montgomery form; z == 0 is the point at infinity
struct Pt { U256h x, y, z; };
Pt pt_inf() { return Pt{{{0, 0, 0, 0}}, FP().one, {{0, 0, 0, 0}}}; }
Pt pt_double(const Pt& p) {
    const Mont& F = FP();
    if (p.z.is_zero() || p.y.is_zero()) return pt_inf();
    U256h a = F.mul(p.x, p.x), b = F.mul(p.y, p.y), c = F.mul(b, b);
    U256h t = F.add(p.x, b);
    U256h d = F.sub(F.sub(F.mul(t, t), a), c);
    d = F.add(d, d);
    U256h e = F.add(F.add(a, a), a), f = F.mul(e, e);
    Pt r;
    r.x = F.sub(f, F.add(d, d));
    U256h c8 = F.add(c, c); c8 = F.add(c8, c8); c8 = F.add(c8, c8);
    r.y = F.sub(F.mul(e, F.sub(d, r.x)), c8);
    U256h yz = F.mul(p.y, p.z);
    r.z = F.add(yz, yz);
    return r;
}
Pt pt_add(const Pt& p, const Pt& q) {
    const Mont& F = FP();
    if (p.z.is_zero()) return q;
    if (q.z.is_zero()) return p;
    U256h z1z1 = F.mul(p.z, p.z), z2z2 = F.mul(q.z, q.z);
    U256h u1 = F.mul(p.x, z2z2), u2 = F.mul(q.x, z1z1);
    U256h s1 = F.mul(F.mul(p.y, q.z), z2z2), s2 = F.mul(F.mul(q.y, p.z), z1z1);
    if (u1 == u2) return s1 == s2 ? pt_double(p) : pt_inf();
    U256h h = F.sub(u2, u1), r = F.sub(s2, s1);
    U256h hh = F.mul(h, h), hhh = F.mul(hh, h), v = F.mul(u1, hh);
    Pt o;
    o.x = F.sub(F.sub(F.mul(r, r), hhh), F.add(v, v));
    o.y = F.sub(F.mul(r, F.sub(v, o.x)), F.mul(s1, hhh));
    o.z = F.mul(F.mul(p.z, q.z), h);
    return o;
}
Pt pt_mul(const Pt& p, const U256h& k) {   // k: plain integer
    Pt r = pt_inf();
    for (int i = 255; i >= 0; --i) {
        r = pt_double(r);
        if ((k.l[i / 64] >> (i % 64)) & 1) r = pt_add(r, p);
    }
    return r;
}
bool pt_affine(const Pt& p, U256h& x, U256h& y) {   // plain integers out
    const Mont& F = FP();
    if (p.z.is_zero()) return false;
    U256h zi = F.inv(p.z), zi2 = F.mul(zi, zi);
    x = F.from(F.mul(p.x, zi2));
    y = F.from(F.mul(p.y, F.mul(zi2, zi)));
    return true;
}
Pt generator() { return Pt{FP().to(SECP_GX), FP().to(SECP_GY), FP().one}; }

// ------------------------------------------------------------------------------------------------
// Keccak-256 (original Keccak padding 0x01, as Ethereum uses)
// ------------------------------------------------------------------------------------------------
void keccak_f(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(s[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}
void keccak256(const uint8_t* in, size_t len, uint8_t out[32]) {
    uint64_t s[25];
    std::memset(s, 0, sizeof s);
    const size_t rate = 136;
    uint8_t block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; std::memcpy(&w, in + 8 * i, 8); s[i] ^= w; }
        keccak_f(s);
        in += rate;
        len -= rate;
    }
    std::memset(block, 0, rate);
    std::memcpy(block, in, len);
    block[len] ^= 0x01;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; std::memcpy(&w, block + 8 * i, 8); s[i] ^= w; }
    keccak_f(s);
    std::memcpy(out, s, 32);
}

// ------------------------------------------------------------------------------------------------
// Poseidon 5x5
// ------------------------------------------------------------------------------------------------
struct PoseidonParams {
    static const int T = 5, RF = 8, RP = 60;
    std::vector<Fr> rc;   // (RF + RP) * T
    Fr mds[T][T];
    PoseidonParams() {
        // Grain LFSR in self-shrinking mode, as published with Poseidon: 80-bit state seeded with
        // field type (2 bits) | S-box (4) | field size (12) | t (12) | R_F (10) | R_P (10) | thirty ones
        std::vector<uint8_t> st;
        auto push = [&](uint32_t v, int w) { for (int i = w - 1; i >= 0; --i) st.push_back((v >> i) & 1); };
        push(1, 2); push(0, 4); push(254, 12); push(T, 12); push(RF, 10); push(RP, 10);
        for (int i = 0; i < 30; ++i) st.push_back(1);
        size_t head = 0;   // st is used as a ring of 80 bits starting at `head`
        auto at = [&](size_t i) -> uint8_t { return st[(head + i) % 80]; };
        auto update = [&]() -> uint8_t {
            uint8_t nb = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
            st[head] = nb;
            head = (head + 1) % 80;
            return nb;
        };
        for (int i = 0; i < 160; ++i) update();
        auto next_bit = [&]() -> uint8_t {
            uint8_t b = update();
            while (b == 0) { update(); b = update(); }
            return update();
        };
        auto next_254 = [&](uint64_t l[4]) {   // 254 bits, most significant first
            l[0] = l[1] = l[2] = l[3] = 0;
            for (int i = 253; i >= 0; --i) l[i / 64] |= (uint64_t)next_bit() << (i % 64);
        };
        rc.resize((size_t)(RF + RP) * T);
        for (auto& c : rc) {
            FrRaw raw;
            do next_254(raw.l); while (!fr_raw_is_canonical(raw.l));   // rejection sampling
            c = Fr::from_raw(raw);
        }
        // Cauchy matrix 1 / (x_i + y_j) from 2t further elements (reduced, not rejected)
        Fr xy[2 * T];
        for (;;) {
            bool distinct = true;
            for (int i = 0; i < 2 * T; ++i) {
                uint64_t l[4];
                next_254(l);
                FrRaw raw;
                std::memcpy(raw.l, l, 32);
                if (!fr_raw_is_canonical(raw.l)) {   // value < 2^254 < 2r: one subtraction reduces it
                    uint64_t br = 0;
                    for (int k = 0; k < 4; ++k) { u128 t = (u128)raw.l[k] - FR_MOD[k] - br; raw.l[k] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
                }
                xy[i] = Fr::from_raw(raw);
                for (int k = 0; k < i; ++k) distinct = distinct && xy[k] != xy[i];
            }
            bool ok = distinct;
            for (int i = 0; ok && i < T; ++i)
                for (int j = 0; ok && j < T; ++j) ok = !(xy[i] + xy[T + j]).is_zero();
            if (ok) break;
        }
        for (int i = 0; i < T; ++i)
            for (int j = 0; j < T; ++j) mds[i][j] = (xy[i] + xy[T + j]).invert_or_zero();
    }
};
const PoseidonParams& poseidon_params() { static const PoseidonParams p; return p; }

void poseidon_permute(Fr st[5]) {   // poseidon/native/mod.rs:34-97
    const PoseidonParams& P = poseidon_params();
    const int T = PoseidonParams::T;
    size_t k = 0;
    auto round = [&](bool full) {
        for (int i = 0; i < T; ++i) st[i] = st[i] + P.rc[k + i];
        k += T;
        for (int i = 0; i < (full ? T : 1); ++i) { Fr x2 = st[i] * st[i]; st[i] = x2 * x2 * st[i]; }
        Fr o[5];
        for (int i = 0; i < T; ++i) {
            Fr acc = Fr::zero();
            for (int j = 0; j < T; ++j) acc = acc + P.mds[i][j] * st[j];
            o[i] = acc;
        }
        for (int i = 0; i < T; ++i) st[i] = o[i];
    };
    for (int r = 0; r < PoseidonParams::RF / 2; ++r) round(true);
    for (int r = 0; r < PoseidonParams::RP; ++r) round(false);
    for (int r = 0; r < PoseidonParams::RF / 2; ++r) round(true);
}
Fr sponge_hash(const std::vector<Fr>& in) {   // poseidon/native/sponge.rs:46-68
    std::vector<Fr> inputs = in;
    if (inputs.empty()) inputs.push_back(Fr::zero());
    Fr state[5];
    for (size_t off = 0; off < inputs.size(); off += 5) {
        Fr x[5];
        for (size_t i = 0; i < 5; ++i) x[i] = (off + i < inputs.size() ? inputs[off + i] : Fr::zero()) + state[i];
        poseidon_permute(x);
        for (int i = 0; i < 5; ++i) state[i] = x[i];
    }
    return state[0];
}

// a 32-byte big-endian integer reduced mod r (Scalar::from_uniform_bytes on its reversed bytes, attestation.rs:107-113)
Fr fr_from_be32_reduced(const uint8_t be[32]) {
    FrRaw raw;
    for (int i = 0; i < 4; ++i) {
        uint64_t w = 0;
        for (int b = 0; b < 8; ++b) w = (w << 8) | be[(3 - i) * 8 + b];
        raw.l[i] = w;
    }
    // value < 2^256 < 6r: subtract r while >= r
    while (!fr_raw_is_canonical(raw.l)) {
        uint64_t br = 0;
        for (int k = 0; k < 4; ++k) { u128 t = (u128)raw.l[k] - FR_MOD[k] - br; raw.l[k] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1; }
    }
    return Fr::from_raw(raw);
}
Fr fr_from_address(const uint8_t a[20]) {   // eth.rs:77-95
    FrRaw raw{{0, 0, 0, 0}};
    for (int i = 0; i < 20; ++i) raw.l[i / 8] |= (uint64_t)a[19 - i] << (8 * (i % 8));
    return Fr::from_raw(raw);
}
Fr hash_fields(const Fr& about, const Fr& domain, const Fr& value, const Fr& message) {   // dynamic_sets/native.rs:96-104
    Fr st[5] = {about, domain, value, message, Fr::zero()};
    poseidon_permute(st);
    return st[0];
}
Fr attestation_hash(const et_attestation& a) {
    return hash_fields(fr_from_address(a.about), fr_from_address(a.domain), Fr::from_u64(a.value), fr_from_be32_reduced(a.message));
}
U256h u256_of_fr(const Fr& f) {
    FrRaw r = f.to_raw();
    U256h u;
    std::memcpy(u.l, r.l, 32);
    return u;
}

// ECDSA verify (ecdsa/native.rs:382-395): x(u1*G + u2*Q) mod n == r with u1 = m/s, u2 = r/s
bool ecdsa_verify(const U256h& r, const U256h& s, const U256h& msg, const Pt& q) {
    const Mont& N = FN();
    if (r.is_zero() || s.is_zero() || cmp256(r, SECP_N) >= 0 || cmp256(s, SECP_N) >= 0) return false;
    U256h si = N.inv(N.to(s));
    U256h m = msg;
    if (cmp256(m, SECP_N) >= 0) { U256h t; sub256(t, m, SECP_N); m = t; }
    U256h u1 = N.from(N.mul(N.to(m), si)), u2 = N.from(N.mul(N.to(r), si));
    Pt R = pt_add(pt_mul(generator(), u1), pt_mul(q, u2));
    U256h x, y;
    if (!pt_affine(R, x, y)) return false;
    if (cmp256(x, SECP_N) >= 0) { U256h t; sub256(t, x, SECP_N); x = t; }
    return x == r;
}
// ECDSA public-key recovery (ecdsa/native.rs:298-331).  sig_r / sig_s are little-endian, rec_id != 0
// means R has an odd y (ecdsa/native.rs:196-208, :33-57).
int ecdsa_recover(const uint8_t sig_r[32], const uint8_t sig_s[32], uint8_t rec_id, const U256h& msg, Pt& q_out, U256h& qx, U256h& qy) {
    const Mont& F = FP();
    const Mont& N = FN();
    U256h r, s;
    std::memcpy(r.l, sig_r, 32);
    std::memcpy(s.l, sig_s, 32);
    if (cmp256(r, SECP_N) >= 0 || cmp256(s, SECP_N) >= 0 || r.is_zero() || s.is_zero()) { set_error("signature scalar out of range"); return ET_E_BAD_SIGNATURE; }
    // R = (r, y) with y^2 = r^3 + 7 and the requested parity
    U256h xm = F.to(r);
    U256h rhs = F.add(F.mul(F.mul(xm, xm), xm), F.to(U256h{{7, 0, 0, 0}}));
    U256h e;   // (p + 1) / 4
    {
        U256h p1, one{{1, 0, 0, 0}};
        add256(p1, SECP_P, one);   // p + 1 = 2^256 - 2^32 - 976 still fits 256 bits
        for (int i = 0; i < 4; ++i) e.l[i] = (p1.l[i] >> 2) | ((i < 3 ? p1.l[i + 1] : 0) << 62);
    }
    U256h ym = F.pow(rhs, e);
    if (!(F.mul(ym, ym) == rhs)) { set_error("signature r is not the x coordinate of a curve point"); return ET_E_BAD_SIGNATURE; }
    if ((F.from(ym).l[0] & 1) != (rec_id != 0 ? 1u : 0u)) ym = F.neg(ym);
    Pt R{xm, ym, F.one};
    U256h ri = N.inv(N.to(r));
    U256h m = msg;
    if (cmp256(m, SECP_N) >= 0) { U256h t; sub256(t, m, SECP_N); m = t; }
    U256h u1 = N.from(N.neg(N.mul(ri, N.to(m)))), u2 = N.from(N.mul(ri, N.to(s)));
    Pt q = pt_add(pt_mul(generator(), u1), pt_mul(R, u2));
    if (!pt_affine(q, qx, qy)) { set_error("recovered key is the point at infinity"); return ET_E_BAD_SIGNATURE; }
    q_out = Pt{F.to(qx), F.to(qy), F.one};
    if (!ecdsa_verify(r, s, msg, q_out)) { set_error("recovered key does not verify the signature"); return ET_E_BAD_SIGNATURE; }   // :322-327
    return ET_OK;
}
void address_of_key(const U256h& qx, const U256h& qy, uint8_t out[20]) {   // ecdsa/native.rs:90-111, eth.rs:70-74
    uint8_t buf[64], h[32];
    for (int i = 0; i < 32; ++i) { buf[i] = (uint8_t)(qx.l[(31 - i) / 8] >> (8 * ((31 - i) % 8))); buf[32 + i] = (uint8_t)(qy.l[(31 - i) / 8] >> (8 * ((31 - i) % 8))); }
    keccak256(buf, 64, h);
    std::memcpy(out, h + 12, 20);
}

struct Recovered { Pt key; uint8_t address[20]; U256h x, y; Fr hash; };
int recover_one(const et_attestation& a, Recovered& out) {
    out.hash = attestation_hash(a);
    int rc = ecdsa_recover(a.sig_r, a.sig_s, a.rec_id, u256_of_fr(out.hash), out.key, out.x, out.y);
    if (rc) return rc;
    address_of_key(out.x, out.y, out.address);
    return ET_OK;
}

template <typename F>
void parallel_for(size_t n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = std::max(1u, std::min(16u, hw ? hw : 1u));
    if (n < 4) nt = 1;
    std::atomic<size_t> next(0);
    auto work = [&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= n) break; f(i); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}

}  // namespace

namespace et {
void poseidon5_parameters(std::vector<FrRaw>& round_constants, std::vector<FrRaw>& mds) {
    const PoseidonParams& P = poseidon_params();
    round_constants.clear();
    for (const Fr& c : P.rc) round_constants.push_back(c.to_raw());
    mds.clear();
    for (int i = 0; i < PoseidonParams::T; ++i)
        for (int j = 0; j < PoseidonParams::T; ++j) mds.push_back(P.mds[i][j].to_raw());
}
}  // namespace et

extern "C" {

int et_poseidon5_permute(const et_fr* in, et_fr* out) {
    if (!in || !out) return ET_E_INVALID_ARG;
    Fr st[5];
    for (int i = 0; i < 5; ++i) {
        FrRaw r;
        std::memcpy(r.l, in[i].b, 32);
        if (!fr_raw_is_canonical(r.l)) return ET_E_NONCANONICAL_FR;
        st[i] = Fr::from_raw(r);
    }
    poseidon_permute(st);
    for (int i = 0; i < 5; ++i) { FrRaw r = st[i].to_raw(); std::memcpy(out[i].b, r.l, 32); }
    return ET_OK;
}

int et_poseidon5_sponge(const et_fr* in, size_t n, et_fr* out) {
    if ((!in && n) || !out) return ET_E_INVALID_ARG;
    std::vector<Fr> v(n);
    for (size_t i = 0; i < n; ++i) {
        FrRaw r;
        std::memcpy(r.l, in[i].b, 32);
        if (!fr_raw_is_canonical(r.l)) return ET_E_NONCANONICAL_FR;
        v[i] = Fr::from_raw(r);
    }
    FrRaw r = sponge_hash(v).to_raw();
    std::memcpy(out->b, r.l, 32);
    return ET_OK;
}

int et_keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    if ((!data && len) || !out) return ET_E_INVALID_ARG;
    keccak256(data ? data : (const uint8_t*)"", len, out);
    return ET_OK;
}

int et_attestation_hash(const et_attestation* att, et_fr* out) {
    if (!att || !out) return ET_E_INVALID_ARG;
    FrRaw r = attestation_hash(*att).to_raw();
    std::memcpy(out->b, r.l, 32);
    return ET_OK;
}

int et_attestations_recover(const et_attestation* atts, size_t n, uint8_t* attesters, uint8_t* public_keys) {
    if ((!atts && n) || (!attesters && n)) return ET_E_INVALID_ARG;
    std::vector<int> rcs(n, ET_OK);
    std::vector<std::string> errs(n);
    parallel_for(n, [&](size_t k) {
        Recovered rec;
        rcs[k] = recover_one(atts[k], rec);
        if (rcs[k]) { errs[k] = last_error(); return; }
        std::memcpy(attesters + 20 * k, rec.address, 20);
        if (public_keys) { std::memcpy(public_keys + 64 * k, rec.x.l, 32); std::memcpy(public_keys + 64 * k + 32, rec.y.l, 32); }
    });
    for (size_t k = 0; k < n; ++k)
        if (rcs[k]) { set_error("attestation " + std::to_string(k) + ": " + errs[k]); return rcs[k]; }
    return ET_OK;
}

int et_secp256k1_public_key(const uint8_t secret_le[32], uint8_t public_key[64], uint8_t address[20]) {
    if (!secret_le || !public_key) return ET_E_INVALID_ARG;
    U256h k, x, y;
    std::memcpy(k.l, secret_le, 32);
    if (k.is_zero() || cmp256(k, SECP_N) >= 0) return ET_E_INVALID_ARG;
    if (!pt_affine(pt_mul(generator(), k), x, y)) return ET_E_INVALID_ARG;
    std::memcpy(public_key, x.l, 32);
    std::memcpy(public_key + 32, y.l, 32);
    if (address) address_of_key(x, y, address);
    return ET_OK;
}

int et_secp256k1_sign(const uint8_t secret_le[32], const et_fr* msg_hash, const uint8_t nonce_le[32], uint8_t sig_r[32], uint8_t sig_s[32],
                      uint8_t* rec_id) {
    if (!secret_le || !msg_hash || !nonce_le || !sig_r || !sig_s || !rec_id) return ET_E_INVALID_ARG;
    const Mont& N = FN();
    U256h d, k, m, x, y;
    std::memcpy(d.l, secret_le, 32);
    std::memcpy(k.l, nonce_le, 32);
    std::memcpy(m.l, msg_hash->b, 32);
    if (d.is_zero() || k.is_zero() || cmp256(d, SECP_N) >= 0 || cmp256(k, SECP_N) >= 0) return ET_E_INVALID_ARG;
    if (!pt_affine(pt_mul(generator(), k), x, y)) return ET_E_INVALID_ARG;   // ecdsa/native.rs:270-276
    U256h r = x;
    if (cmp256(r, SECP_N) >= 0) { U256h t; sub256(t, r, SECP_N); r = t; }
    if (cmp256(m, SECP_N) >= 0) { U256h t; sub256(t, m, SECP_N); m = t; }
    // s = k^-1 * (m + r * d)  (:282-284)
    U256h s = N.from(N.mul(N.inv(N.to(k)), N.add(N.to(m), N.mul(N.to(r), N.to(d)))));
    if (r.is_zero() || s.is_zero()) return ET_E_INVALID_ARG;
    std::memcpy(sig_r, r.l, 32);
    std::memcpy(sig_s, s.l, 32);
    *rec_id = (uint8_t)(y.l[0] & 1);                                          // :288-289
    return ET_OK;
}

int et_opinion_validate(const uint8_t public_key[64], const et_attestation* atts, const uint8_t* present, const uint8_t* set_addresses,
                        uint32_t num_neighbours, const uint8_t domain[20], et_fr* scores, et_fr* op_hash) {
    if (!public_key || !atts || !set_addresses || !domain || !scores || !op_hash || num_neighbours == 0) return ET_E_INVALID_ARG;
    const Mont& F = FP();
    U256h qx, qy;
    std::memcpy(qx.l, public_key, 32);
    std::memcpy(qy.l, public_key + 32, 32);
    const bool default_pk = qx.is_zero() && qy.is_zero();           // opinion/native.rs:69
    uint8_t from[20];
    address_of_key(qx, qy, from);
    bool member = false;
    for (uint32_t i = 0; i < num_neighbours; ++i) member = member || !std::memcmp(set_addresses + 20 * i, from, 20);
    if (!member) { set_error("attester is not in the set"); return ET_E_NOT_MEMBER; }   // :66-67
    const Pt q{F.to(qx), F.to(qy), F.one};
    const Fr dom = fr_from_address(domain);
    std::vector<Fr> hashes(num_neighbours);
    std::vector<int> rcs(num_neighbours, ET_OK);
    parallel_for(num_neighbours, [&](size_t i) {
        const uint8_t* about = set_addresses + 20 * i;
        const bool have = !present || present[i];
        Fr h;
        U256h r{{1, 0, 0, 0}}, s{{1, 0, 0, 0}};                     // empty_with_about: r = s = 1 (dynamic_sets/native.rs:55-60)
        Fr value = Fr::zero();
        if (have) {
            const et_attestation& a = atts[i];
            if (std::memcmp(a.about, about, 20) != 0) { rcs[i] = ET_E_BAD_ABOUT; return; }          // :75
            if (std::memcmp(a.domain, domain, 20) != 0) { rcs[i] = ET_E_BAD_DOMAIN; return; }       // :76
            h = attestation_hash(a);
            std::memcpy(r.l, a.sig_r, 32);
            std::memcpy(s.l, a.sig_s, 32);
            value = Fr::from_u64(a.value);
        } else {
            h = hash_fields(fr_from_address(about), dom, Fr::zero(), Fr::zero());
        }
        bool zero_addr = true;
        for (int b = 0; b < 20; ++b) zero_addr = zero_addr && about[b] == 0;
        const bool valid = !default_pk && !zero_addr && ecdsa_verify(r, s, u256_of_fr(h), q);   // :84-92
        FrRaw sv = (valid ? value : Fr::zero()).to_raw();
        std::memcpy(scores[i].b, sv.l, 32);
        hashes[i] = valid ? h : Fr::zero();
    });
    for (uint32_t i = 0; i < num_neighbours; ++i)
        if (rcs[i]) { set_error("attestation " + std::to_string(i) + " does not match the set or the domain"); return rcs[i]; }
    FrRaw oh = sponge_hash(hashes).to_raw();                        // :103-105
    std::memcpy(op_hash->b, oh.l, 32);
    return ET_OK;
}

int et_et_setup(const et_attestation* atts, size_t n_att, const uint8_t domain[20], uint32_t num_neighbours, uint32_t num_iterations,
                uint64_t initial_lo, uint64_t initial_hi, et_score* out, size_t cap, size_t* n_participants, uint8_t* public_inputs,
                size_t public_inputs_cap) {
    if ((!atts && n_att) || !domain || !n_participants || num_neighbours == 0) return ET_E_INVALID_ARG;
    typedef std::array<uint8_t, 20> Addr;
    // recover every attester (lib.rs:351-358)
    std::vector<uint8_t> who(20 * std::max<size_t>(1, n_att)), keys(64 * std::max<size_t>(1, n_att));
    int rc = et_attestations_recover(atts, n_att, who.data(), keys.data());
    if (rc) return rc;
    std::set<Addr> parts;
    std::map<Addr, std::array<uint8_t, 64>> key_of;
    for (size_t k = 0; k < n_att; ++k) {
        Addr about, origin;
        std::memcpy(about.data(), atts[k].about, 20);
        std::memcpy(origin.data(), who.data() + 20 * k, 20);
        parts.insert(about);
        parts.insert(origin);
        std::array<uint8_t, 64> key;
        std::memcpy(key.data(), keys.data() + 64 * k, 64);
        key_of[origin] = key;
    }
    std::vector<Addr> address_set(parts.begin(), parts.end());
    *n_participants = address_set.size();
    if (address_set.size() > num_neighbours) { set_error("Number of participants exceeds maximum number of neighbours"); return ET_E_SET_FULL; }
    if (address_set.size() < 2) { set_error("Number of participants is less than the minimum number of neighbours"); return ET_E_INSUFFICIENT_PEERS; }
    if (!out) return ET_OK;
    if (cap < address_set.size()) return ET_E_BUFFER_TOO_SMALL;
    if (public_inputs && public_inputs_cap < ((size_t)2 * num_neighbours + 2) * 32) return ET_E_BUFFER_TOO_SMALL;
    std::map<Addr, uint32_t> index;
    for (uint32_t i = 0; i < address_set.size(); ++i) index[address_set[i]] = i;
    // set addresses, zero padded (lib.rs:377-386)
    std::vector<uint8_t> set_addr((size_t)20 * num_neighbours, 0);
    for (size_t i = 0; i < address_set.size(); ++i) std::memcpy(set_addr.data() + 20 * i, address_set[i].data(), 20);
    // attestation matrix (lib.rs:400-419)
    std::vector<std::vector<et_attestation>> matrix(address_set.size(), std::vector<et_attestation>(num_neighbours));
    std::vector<std::vector<uint8_t>> present(address_set.size(), std::vector<uint8_t>(num_neighbours, 0));
    for (size_t k = 0; k < n_att; ++k) {
        Addr about, origin;
        std::memcpy(about.data(), atts[k].about, 20);
        std::memcpy(origin.data(), who.data() + 20 * k, 20);
        matrix[index[origin]][index[about]] = atts[k];
        present[index[origin]][index[about]] = 1;
    }
    et_fr dom_fr;
    et_scalar_from_address(domain, &dom_fr);
    et_set* set = nullptr;
    if ((rc = et_set_new(num_neighbours, num_iterations, initial_lo, initial_hi, &dom_fr, &set))) return rc;
    struct Guard { et_set* s; ~Guard() { et_set_free(s); } } guard{set};
    std::vector<et_fr> scalars(num_neighbours);
    std::memset(scalars.data(), 0, scalars.size() * sizeof(et_fr));
    for (size_t i = 0; i < address_set.size(); ++i) {
        et_scalar_from_address(address_set[i].data(), &scalars[i]);
        if ((rc = et_set_add_member(set, &scalars[i]))) return rc;
    }
    // opinions of the participants that attested, in set order (lib.rs:428-434)
    std::vector<Fr> op_hashes;
    for (uint32_t i = 0; i < address_set.size(); ++i) {
        auto it = key_of.find(address_set[i]);
        if (it == key_of.end()) continue;
        std::vector<et_fr> scores(num_neighbours);
        et_fr oh;
        if ((rc = et_opinion_validate(it->second.data(), matrix[i].data(), present[i].data(), set_addr.data(), num_neighbours, domain,
                                      scores.data(), &oh))) return rc;
        if ((rc = et_set_update_op(set, &scalars[i], scores.data()))) return rc;
        FrRaw r;
        std::memcpy(r.l, oh.b, 32);
        op_hashes.push_back(Fr::from_raw(r));
    }
    std::vector<et_fr> scores(num_neighbours);
    if ((rc = et_set_converge(set, scores.data()))) return rc;            // lib.rs:438 (GPU)
    size_t stride = 64, needed = 0;
    std::vector<uint8_t> num, den;
    for (;;) {                                                            // lib.rs:437
        num.assign(stride * num_neighbours, 0);
        den.assign(stride * num_neighbours, 0);
        rc = et_set_converge_rational(set, num.data(), den.data(), stride, &needed);
        if (rc != ET_E_BUFFER_TOO_SMALL) break;
        stride = needed;
    }
    if (rc) return rc;
    if (public_inputs) {                                                  // lib.rs:455-462
        FrRaw oh = sponge_hash(op_hashes).to_raw();
        et_fr ohf;
        std::memcpy(ohf.b, oh.l, 32);
        if ((rc = et_public_inputs_pack(scalars.data(), scores.data(), num_neighbours, &dom_fr, &ohf, public_inputs, public_inputs_cap))) return rc;
    }
    for (size_t i = 0; i < address_set.size(); ++i)                       // lib.rs:207-233
        if ((rc = et_score_pack(address_set[i].data(), &scores[i], num.data() + i * stride, stride, den.data() + i * stride, stride, out + i))) return rc;
    return ET_OK;
}

}  // extern "C"
