// What sits on either side of the converge path in the reference's client crate, host-side:
//   - attestations.csv reader / scores.csv writer          (eigentrust/src/storage.rs:184-307)
//   - participant set building and score packaging          (eigentrust/src/lib.rs:201-236, :339-440)
//   - the threshold check on a rational score               (eigentrust-zk/src/circuits/threshold/native.rs:33-96,
//                                                            utils.rs:332-354, params/rns/mod.rs:202-241)
//   - the ET public-input byte layout                       (eigentrust/src/circuit.rs:84-151)
// The scores themselves come from the GPU through the set API (et_set_converge); only the exact
// fractions (et_set_converge_rational) and the bookkeeping here run on the host.
#include <algorithm>
#include <atomic>
#include <array>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/eigentrust_hip.h"
#include "bigrat.hpp"
#include "engine.hpp"
#include "fr_host.hpp"

using namespace et;

namespace {

BigUint big_from_le(const uint8_t* p, size_t len) {
    BigUint r;
    r.d.assign((len + 3) / 4, 0);
    for (size_t i = 0; i < len; ++i) r.d[i / 4] |= (uint32_t)p[i] << (8 * (i % 4));
    r.trim();
    return r;
}
BigUint big_pow10(uint32_t k) {
    BigUint r(1), ten9(1000000000ULL);
    while (k >= 9) { r = BigUint::mul(r, ten9); k -= 9; }
    uint64_t t = 1;
    while (k--) t *= 10;
    return BigUint::mul(r, BigUint(t));
}
std::string big_to_decimal(BigUint v) {
    if (v.is_zero()) return "0";
    std::vector<uint32_t> chunks;  // base 10^9, least significant first
    const BigUint ten9(1000000000ULL);
    while (!v.is_zero()) {
        BigUint q, r;
        BigUint::divmod(v, ten9, q, r);
        chunks.push_back(r.is_zero() ? 0u : r.d[0]);
        v = q;
    }
    char buf[16];
    std::string s;
    snprintf(buf, sizeof buf, "%u", chunks.back());
    s += buf;
    for (size_t i = chunks.size() - 1; i-- > 0;) {
        snprintf(buf, sizeof buf, "%09u", chunks[i]);
        s += buf;
    }
    return s;
}
BigUint modulus_big() { return BigUint::from_le_u64(FR_MOD, 4); }
// big_to_fe (utils.rs:320-324): e mod r as a field element
Fr big_to_fr(const BigUint& e) {
    BigUint q, rem;
    BigUint::divmod(e, modulus_big(), q, rem);
    FrRaw raw{{0, 0, 0, 0}};
    for (size_t i = 0; i < rem.d.size(); ++i) raw.l[i / 2] |= (uint64_t)rem.d[i] << (32 * (i % 2));
    return Fr::from_raw(raw);
}
BigUint fr_to_big(const Fr& f) {
    FrRaw r = f.to_raw();
    return BigUint::from_le_u64(r.l, 4);
}
bool read_fr(const et_fr* in, Fr& out) {
    FrRaw r;
    std::memcpy(r.l, in->b, 32);
    if (!fr_raw_is_canonical(r.l)) return false;
    out = Fr::from_raw(r);
    return true;
}
void write_fr(const Fr& f, et_fr* out) {
    FrRaw r = f.to_raw();
    std::memcpy(out->b, r.l, 32);
}

// decompose_big_decimal (params/rns/mod.rs:202-214): num_limbs limbs of power_of_ten decimal digits,
// least significant first; anything above the last limb is dropped, as there
std::vector<BigUint> decompose_decimal(BigUint e, uint32_t num_limbs, uint32_t power_of_ten) {
    const BigUint scale = big_pow10(power_of_ten);
    std::vector<BigUint> limbs(num_limbs);
    for (uint32_t i = 0; i < num_limbs; ++i) {
        BigUint q, rem;
        BigUint::divmod(e, scale, q, rem);
        limbs[i] = rem;
        e = q;
    }
    return limbs;
}
// big_to_fe_rat (utils.rs:332-354): scale numerator and denominator by the same power of ten so
// that the larger one has exactly num_limbs * power_of_ten digits, then decompose both
int ratio_decompose(const BigUint& num, const BigUint& den, uint32_t num_limbs, uint32_t power_of_ten,
                    std::vector<BigUint>& nl, std::vector<BigUint>& dl) {
    const size_t max_len = (size_t)num_limbs * power_of_ten;
    const BigUint& bigger = BigUint::cmp(num, den) >= 0 ? num : den;
    const size_t dig_len = big_to_decimal(bigger).size();
    if (dig_len > max_len) {   // `max_len - dig_len` underflows and panics in the reference (utils.rs:340)
        set_error("ratio has " + std::to_string(dig_len) + " decimal digits, the limbs hold " + std::to_string(max_len));
        return ET_E_RATIO_DIGITS;
    }
    const BigUint scale = big_pow10((uint32_t)(max_len - dig_len));
    nl = decompose_decimal(BigUint::mul(num, scale), num_limbs, power_of_ten);
    dl = decompose_decimal(BigUint::mul(den, scale), num_limbs, power_of_ten);
    return ET_OK;
}
// compose_big_decimal_f (params/rns/mod.rs:232-243)
Fr compose_decimal_fr(const std::vector<BigUint>& limbs, uint32_t power_of_ten) {
    const Fr scale = big_to_fr(big_pow10(power_of_ten));
    Fr val = big_to_fr(limbs.back());
    for (size_t i = limbs.size() - 1; i-- > 0;) val = val * scale + big_to_fr(limbs[i]);
    return val;
}

int hex_nibble(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}
// "0x" + exactly 2*len hex digits (storage.rs:310-331 accept only full-width values)
bool parse_hex_fixed(const std::string& s, uint8_t* out, size_t len) {
    size_t off = (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) ? 2 : 0;
    if (s.size() - off != 2 * len) return false;
    for (size_t i = 0; i < len; ++i) {
        int h = hex_nibble(s[off + 2 * i]), l = hex_nibble(s[off + 2 * i + 1]);
        if (h < 0 || l < 0) return false;
        out[i] = (uint8_t)(h << 4 | l);
    }
    return true;
}
void append_hex(std::string& s, const uint8_t* p, size_t len) {
    static const char* digits = "0123456789abcdef";
    s += "0x";
    for (size_t i = 0; i < len; ++i) { s += digits[p[i] >> 4]; s += digits[p[i] & 15]; }
}
bool parse_u8(const std::string& s, uint8_t& out) {
    if (s.empty() || s.size() > 3) return false;
    unsigned v = 0;
    for (char c : s) { if (c < '0' || c > '9') return false; v = v * 10 + (unsigned)(c - '0'); }
    if (v > 255) return false;
    out = (uint8_t)v;
    return true;
}
int copy_out(const std::string& s, char* buf, size_t cap, size_t* needed) {
    if (needed) *needed = s.size() + 1;
    if (!buf || cap < s.size() + 1) return ET_E_BUFFER_TOO_SMALL;
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return ET_OK;
}
std::string big_endian_decimal(const uint8_t* be, size_t len) {
    std::vector<uint8_t> le(be, be + len);
    std::reverse(le.begin(), le.end());
    return big_to_decimal(big_from_le(le.data(), len));
}
// a big-endian magnitude right-aligned in 32 bytes (lib.rs:218-230)
bool put_be32(const BigUint& v, uint8_t out[32]) {
    if (v.byte_len() > 32) return false;
    uint8_t le[32];
    v.to_le_bytes(le, 32);
    for (int i = 0; i < 32; ++i) out[i] = le[31 - i];
    return true;
}

}  // namespace

extern "C" {

int et_ratio_decompose(const uint8_t* num_le, size_t num_len, const uint8_t* den_le, size_t den_len, uint32_t num_limbs,
                       uint32_t power_of_ten, et_fr* num_limbs_out, et_fr* den_limbs_out) {
    if (!num_le || !den_le || !num_limbs_out || !den_limbs_out || num_limbs == 0 || power_of_ten == 0) return ET_E_INVALID_ARG;
    std::vector<BigUint> nl, dl;
    int rc = ratio_decompose(big_from_le(num_le, num_len), big_from_le(den_le, den_len), num_limbs, power_of_ten, nl, dl);
    if (rc) return rc;
    for (uint32_t i = 0; i < num_limbs; ++i) { write_fr(big_to_fr(nl[i]), num_limbs_out + i); write_fr(big_to_fr(dl[i]), den_limbs_out + i); }
    return ET_OK;
}

int et_threshold_check(const et_fr* score_fr, const uint8_t* num_le, size_t num_len, const uint8_t* den_le, size_t den_len,
                       const et_fr* threshold, uint32_t num_limbs, uint32_t power_of_ten, uint32_t num_neighbours,
                       uint64_t initial_lo, uint64_t initial_hi, int* passed) {
    if (!score_fr || !num_le || !den_le || !threshold || !passed || num_limbs == 0 || power_of_ten == 0) return ET_E_INVALID_ARG;
    Fr score, thr;
    if (!read_fr(score_fr, score) || !read_fr(threshold, thr)) return ET_E_NONCANONICAL_FR;
    // Threshold::new (threshold/native.rs:33-56)
    const uint64_t init_l[2] = {initial_lo, initial_hi};
    const BigUint max_score = BigUint::mul(BigUint(num_neighbours), BigUint::from_le_u64(init_l, 2));
    const BigUint limb_bound = big_pow10(power_of_ten);
    const BigUint max_f = BigUint::sub(modulus_big(), BigUint(1));
    if (BigUint::cmp(BigUint::mul(max_score, BigUint::sub(limb_bound, BigUint(1))), max_f) >= 0) {
        set_error("num_neighbours * initial_score * (10^power_of_ten - 1) does not fit the field");   // :34-37
        return ET_E_THRESHOLD_RANGE;
    }
    std::vector<BigUint> nl, dl;
    int rc = ratio_decompose(big_from_le(num_le, num_len), big_from_le(den_le, den_len), num_limbs, power_of_ten, nl, dl);
    if (rc) return rc;
    // check_threshold (:60-96); the limb range asserts (:66-73) hold by construction
    if (BigUint::cmp(fr_to_big(thr), max_score) >= 0) { set_error("threshold is not below the maximum score"); return ET_E_THRESHOLD_RANGE; }
    const Fr cn = compose_decimal_fr(nl, power_of_ten), cd = compose_decimal_fr(dl, power_of_ten);
    if (cd.is_zero()) { set_error("composed denominator is zero in the field"); return ET_E_ZERO_LIMB; }   // :79 unwrap
    if (cn * cd.invert_or_zero() != score) { set_error("numerator / denominator does not equal the field score"); return ET_E_RATIO_MISMATCH; }  // :81
    if (dl.back().is_zero()) { set_error("top decimal limb of the denominator is zero"); return ET_E_ZERO_LIMB; }  // :90
    const BigUint comp = fr_to_big(big_to_fr(dl.back()) * thr);   // :92-93, in the field as there
    *passed = BigUint::cmp(nl.back(), comp) >= 0 ? 1 : 0;         // :95
    return ET_OK;
}

int et_scalar_from_address(const uint8_t address[20], et_fr* out) {
    if (!address || !out) return ET_E_INVALID_ARG;
    // eth.rs:77-95: the address bytes reversed are the low 20 bytes of the little-endian scalar
    std::memset(out->b, 0, 32);
    for (int i = 0; i < 20; ++i) out->b[i] = address[19 - i];
    return ET_OK;
}

int et_score_pack(const uint8_t address[20], const et_fr* score_fr, const uint8_t* num_le, size_t num_len,
                  const uint8_t* den_le, size_t den_len, et_score* out) {
    if (!address || !score_fr || !num_le || !den_le || !out) return ET_E_INVALID_ARG;
    std::memcpy(out->address, address, 20);
    for (int i = 0; i < 32; ++i) out->score_fr[i] = score_fr->b[31 - i];   // lib.rs:215-216
    const BigUint num = big_from_le(num_le, num_len), den = big_from_le(den_le, den_len);
    if (den.is_zero()) return ET_E_INVALID_ARG;
    BigUint q, rem;
    BigUint::divmod(num, den, q, rem);                                       // to_integer(), lib.rs:220
    // the reference indexes `32 - len` and panics when a magnitude needs more than 32 bytes (lib.rs:222-229)
    if (!put_be32(num, out->numerator) || !put_be32(den, out->denominator) || !put_be32(q, out->score)) {
        set_error("numerator or denominator of the rational score exceeds 32 bytes");
        return ET_E_RATIONAL_TOO_LARGE;
    }
    return ET_OK;
}

int et_attestations_csv_parse(const char* text, size_t len, et_attestation* out, size_t cap, size_t* count) {
    if (!text || !count) return ET_E_INVALID_ARG;
    std::vector<et_attestation> recs;
    size_t pos = 0, line_no = 0;
    while (pos < len) {
        size_t eol = pos;
        while (eol < len && text[eol] != '\n') ++eol;
        std::string line(text + pos, eol - pos);
        pos = eol + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        ++line_no;
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t a = 0;
        for (;;) {
            size_t c = line.find(',', a);
            f.push_back(line.substr(a, c == std::string::npos ? std::string::npos : c - a));
            if (c == std::string::npos) break;
            a = c + 1;
        }
        if (line_no == 1) {   // header (storage.rs:247-262 field names)
            static const char* names[7] = {"about", "domain", "value", "message", "sig_r", "sig_s", "rec_id"};
            bool ok = f.size() == 7;
            for (size_t i = 0; ok && i < 7; ++i) ok = f[i] == names[i];
            if (!ok) { set_error("attestations.csv: unexpected header"); return ET_E_PARSE; }
            continue;
        }
        et_attestation r;
        if (f.size() != 7 || !parse_hex_fixed(f[0], r.about, 20) || !parse_hex_fixed(f[1], r.domain, 20) || !parse_u8(f[2], r.value) ||
            !parse_hex_fixed(f[3], r.message, 32) || !parse_hex_fixed(f[4], r.sig_r, 32) || !parse_hex_fixed(f[5], r.sig_s, 32) ||
            !parse_u8(f[6], r.rec_id)) {
            set_error("attestations.csv: cannot parse line " + std::to_string(line_no));   // storage.rs:286-307
            return ET_E_PARSE;
        }
        recs.push_back(r);
    }
    *count = recs.size();
    if (!out) return ET_OK;
    if (cap < recs.size()) return ET_E_BUFFER_TOO_SMALL;
    if (!recs.empty()) std::memcpy(out, recs.data(), recs.size() * sizeof(et_attestation));
    return ET_OK;
}

int et_attestations_csv_format(const et_attestation* recs, size_t n, char* buf, size_t cap, size_t* needed) {
    if (!recs && n) return ET_E_INVALID_ARG;
    std::string s = "about,domain,value,message,sig_r,sig_s,rec_id\n";
    for (size_t i = 0; i < n; ++i) {   // storage.rs:264-281
        const et_attestation& r = recs[i];
        append_hex(s, r.about, 20); s += ',';
        append_hex(s, r.domain, 20); s += ',';
        s += std::to_string((unsigned)r.value); s += ',';
        append_hex(s, r.message, 32); s += ',';
        append_hex(s, r.sig_r, 32); s += ',';
        append_hex(s, r.sig_s, 32); s += ',';
        s += std::to_string((unsigned)r.rec_id); s += '\n';
    }
    return copy_out(s, buf, cap, needed);
}

int et_scores_csv_format(const et_score* scores, size_t n, char* buf, size_t cap, size_t* needed) {
    if (!scores && n) return ET_E_INVALID_ARG;
    std::string s = "peer_address,score_fr,numerator,denominator,score\n";
    for (size_t i = 0; i < n; ++i) {   // storage.rs:207-217
        const et_score& r = scores[i];
        append_hex(s, r.address, 20); s += ',';
        append_hex(s, r.score_fr, 32); s += ',';
        s += big_endian_decimal(r.numerator, 32); s += ',';
        s += big_endian_decimal(r.denominator, 32); s += ',';
        s += big_endian_decimal(r.score, 32); s += '\n';
    }
    return copy_out(s, buf, cap, needed);
}

int et_public_inputs_pack(const et_fr* participants, const et_fr* scores, uint32_t n, const et_fr* domain,
                          const et_fr* opinions_hash, uint8_t* out, size_t cap) {
    if (!participants || !scores || !domain || !opinions_hash || !out) return ET_E_INVALID_ARG;
    const size_t need = ((size_t)2 * n + 2) * 32;   // circuit.rs:114-123
    if (cap < need) return ET_E_BUFFER_TOO_SMALL;
    std::memcpy(out, participants, (size_t)n * 32);
    std::memcpy(out + (size_t)n * 32, scores, (size_t)n * 32);
    std::memcpy(out + (size_t)2 * n * 32, domain->b, 32);
    std::memcpy(out + ((size_t)2 * n + 1) * 32, opinions_hash->b, 32);
    return ET_OK;
}

int et_public_inputs_unpack(const uint8_t* bytes, size_t len, uint32_t n, et_fr* participants, et_fr* scores, et_fr* domain,
                            et_fr* opinions_hash) {
    if (!bytes || !participants || !scores || !domain || !opinions_hash) return ET_E_INVALID_ARG;
    if (len != ((size_t)2 * n + 2) * 32) { set_error("Invalid bytes length."); return ET_E_PARSE; }   // circuit.rs:128-132
    for (size_t i = 0; i < (size_t)2 * n + 2; ++i) {
        uint64_t l[4];
        std::memcpy(l, bytes + i * 32, 32);
        if (!fr_raw_is_canonical(l)) return ET_E_NONCANONICAL_FR;
    }
    std::memcpy(participants, bytes, (size_t)n * 32);
    std::memcpy(scores, bytes + (size_t)n * 32, (size_t)n * 32);
    std::memcpy(domain->b, bytes + (size_t)2 * n * 32, 32);
    std::memcpy(opinions_hash->b, bytes + ((size_t)2 * n + 1) * 32, 32);
    return ET_OK;
}

int et_local_scores(const et_attestation* atts, const uint8_t* attesters, size_t n_att, const uint8_t domain[20],
                    uint32_t num_neighbours, uint32_t num_iterations, uint64_t initial_lo, uint64_t initial_hi, et_score* out,
                    size_t cap, size_t* n_participants) {
    if ((!atts || !attesters) && n_att) return ET_E_INVALID_ARG;
    if (!domain || !n_participants || num_neighbours == 0) return ET_E_INVALID_ARG;
    typedef std::array<uint8_t, 20> Addr;
    // participants in ascending address order: the BTreeSet of lib.rs:344-361
    std::set<Addr> parts;
    std::set<Addr> has_key;
    for (size_t k = 0; k < n_att; ++k) {
        Addr about, origin;
        std::memcpy(about.data(), atts[k].about, 20);
        std::memcpy(origin.data(), attesters + 20 * k, 20);
        parts.insert(about);
        parts.insert(origin);
        has_key.insert(origin);
        if (std::memcmp(atts[k].domain, domain, 20) != 0) {   // opinion/native.rs:76
            set_error("attestation " + std::to_string(k) + " belongs to another domain");
            return ET_E_BAD_DOMAIN;
        }
    }
    std::vector<Addr> address_set(parts.begin(), parts.end());
    *n_participants = address_set.size();
    if (address_set.size() > num_neighbours) { set_error("Number of participants exceeds maximum number of neighbours"); return ET_E_SET_FULL; }   // lib.rs:366-369
    if (address_set.size() < 2) { set_error("Number of participants is less than the minimum number of neighbours"); return ET_E_INSUFFICIENT_PEERS; }  // :371-374
    if (!out) return ET_OK;
    if (cap < address_set.size()) return ET_E_BUFFER_TOO_SMALL;
    std::map<Addr, uint32_t> index;
    for (uint32_t i = 0; i < address_set.size(); ++i) index[address_set[i]] = i;

    et_fr dom_fr;
    et_scalar_from_address(domain, &dom_fr);
    et_set* set = nullptr;
    int rc = et_set_new(num_neighbours, num_iterations, initial_lo, initial_hi, &dom_fr, &set);
    if (rc) return rc;
    struct Guard { et_set* s; ~Guard() { et_set_free(s); } } guard{set};
    std::vector<et_fr> scalars(address_set.size());
    for (size_t i = 0; i < address_set.size(); ++i) {                         // lib.rs:377-386, :423-425
        et_scalar_from_address(address_set[i].data(), &scalars[i]);
        if ((rc = et_set_add_member(set, &scalars[i]))) return rc;
    }
    // attestation matrix: a later attestation of the same (origin, about) pair replaces an earlier one (lib.rs:404-419)
    std::vector<std::vector<uint32_t>> matrix(address_set.size(), std::vector<uint32_t>(num_neighbours, 0));
    for (size_t k = 0; k < n_att; ++k) {
        Addr about, origin;
        std::memcpy(about.data(), atts[k].about, 20);
        std::memcpy(origin.data(), attesters + 20 * k, 20);
        matrix[index[origin]][index[about]] = atts[k].value;
    }
    for (uint32_t i = 0; i < address_set.size(); ++i) {                       // lib.rs:428-434: only peers that attested
        if (!has_key.count(address_set[i])) continue;
        std::vector<uint32_t> about_slot, values;
        for (uint32_t j = 0; j < address_set.size(); ++j)
            if (matrix[i][j]) { about_slot.push_back(j); values.push_back(matrix[i][j]); }
        // slot i holds address_set[i]: add_member takes the first empty slot (native.rs:180-185)
        if ((rc = et_set_update_op_u32(set, i, (uint32_t)values.size(), about_slot.data(), values.data()))) return rc;
    }
    std::vector<et_fr> scores(num_neighbours);
    if ((rc = et_set_converge(set, scores.data()))) return rc;                // lib.rs:438 (GPU)
    size_t stride = 64, needed = 0;
    std::vector<uint8_t> num, den;
    for (;;) {                                                                // lib.rs:437
        num.assign(stride * num_neighbours, 0);
        den.assign(stride * num_neighbours, 0);
        rc = et_set_converge_rational(set, num.data(), den.data(), stride, &needed);
        if (rc != ET_E_BUFFER_TOO_SMALL) break;
        stride = needed;
    }
    if (rc) return rc;
    for (size_t i = 0; i < address_set.size(); ++i)                           // lib.rs:207-233
        if ((rc = et_score_pack(address_set[i].data(), &scores[i], num.data() + i * stride, stride, den.data() + i * stride, stride, out + i))) return rc;
    return ET_OK;
}

int et_scores_from_attestations(const et_attestation* atts, size_t n_att, const uint8_t domain[20], uint32_t num_iterations,
                                uint64_t initial_lo, uint64_t initial_hi, uint8_t* addresses, et_fr* scores_fr, double* scores_f64,
                                size_t cap, size_t* n_participants) {
    if ((!atts && n_att) || !domain || !n_participants) return ET_E_INVALID_ARG;
    typedef std::array<uint8_t, 20> Addr;
    for (size_t k = 0; k < n_att; ++k)
        if (std::memcmp(atts[k].domain, domain, 20) != 0) {   // opinion/native.rs:76
            set_error("attestation " + std::to_string(k) + " belongs to another domain");
            return ET_E_BAD_DOMAIN;
        }
    // attesters, one GPU lane per record (lib.rs:351-358)
    std::vector<uint8_t> who(20 * std::max<size_t>(1, n_att));
    int rc = et_attestations_recover_gpu(atts, n_att, who.data(), nullptr, -1);
    if (rc) return rc;
    // participants in ascending address order (the BTreeSet of lib.rs:344-361).  Addresses are hashes, so their
    // first byte splits them into 256 even buckets whose concatenation is the sorted order: scatter, then sort and
    // deduplicate the buckets on the host threads.
    const unsigned T = host_threads();
    const size_t n_all = 2 * n_att;
    auto addr_of = [&](size_t i) -> const uint8_t* { return (i & 1) ? who.data() + 20 * (i >> 1) : atts[i >> 1].about; };
    std::vector<size_t> cnt((size_t)T * 256, 0);
    run_threads(T, [&](unsigned t) {
        size_t* c = cnt.data() + (size_t)t * 256;
        for (size_t i = n_all * t / T; i < n_all * (t + 1) / T; ++i) ++c[addr_of(i)[0]];
    });
    std::vector<size_t> bucket_begin(257, 0);
    {
        size_t run = 0;
        for (unsigned b = 0; b < 256; ++b) {
            bucket_begin[b] = run;
            for (unsigned t = 0; t < T; ++t) { const size_t c = cnt[(size_t)t * 256 + b]; cnt[(size_t)t * 256 + b] = run; run += c; }
        }
        bucket_begin[256] = run;
    }
    std::vector<Addr> all(n_all);
    run_threads(T, [&](unsigned t) {
        size_t* c = cnt.data() + (size_t)t * 256;
        for (size_t i = n_all * t / T; i < n_all * (t + 1) / T; ++i) {
            const uint8_t* a = addr_of(i);
            std::memcpy(all[c[a[0]]++].data(), a, 20);
        }
    });
    std::vector<size_t> uniq(257, 0);
    {
        std::atomic<unsigned> next(0);
        run_threads(T, [&](unsigned) {
            for (;;) {
                const unsigned b = next.fetch_add(1);
                if (b >= 256) break;
                auto first = all.begin() + bucket_begin[b], last = all.begin() + bucket_begin[b + 1];
                std::sort(first, last);
                uniq[b + 1] = (size_t)(std::unique(first, last) - first);
            }
        });
    }
    for (unsigned b = 0; b < 256; ++b) uniq[b + 1] += uniq[b];    // uniq[b] = first participant of bucket b
    std::vector<Addr> parts(uniq[256]);
    run_threads(T, [&](unsigned t) {
        for (unsigned b = 256 * t / T; b < 256 * (t + 1) / T; ++b)
            std::copy(all.begin() + bucket_begin[b], all.begin() + bucket_begin[b] + (uniq[b + 1] - uniq[b]), parts.begin() + uniq[b]);
    });
    { std::vector<Addr>().swap(all); }
    *n_participants = parts.size();
    if (parts.size() < 2) { set_error("Number of participants is less than the minimum number of neighbours"); return ET_E_INSUFFICIENT_PEERS; }
    if (parts.size() > 0xFFFFFFF0u) return ET_E_UNSUPPORTED;
    if (!addresses && !scores_fr && !scores_f64) return ET_OK;
    if (cap < parts.size()) return ET_E_BUFFER_TOO_SMALL;
    auto index_of = [&](const uint8_t* a) -> uint32_t {
        Addr key;
        std::memcpy(key.data(), a, 20);
        return (uint32_t)(std::lower_bound(parts.begin() + uniq[a[0]], parts.begin() + uniq[a[0] + 1], key) - parts.begin());
    };
    // one edge per record, in record order: a later record of the same (attester, about) pair replaces an
    // earlier one (lib.rs:404-419); self-attestations and zero values drop out in the filter (native.rs:248-260)
    std::vector<uint32_t> from(n_att), to(n_att), val(n_att);
    run_threads(T, [&](unsigned t) {
        for (size_t k = n_att * t / T; k < n_att * (t + 1) / T; ++k) {
            from[k] = index_of(who.data() + 20 * k);
            to[k] = index_of(atts[k].about);
            val[k] = atts[k].value;
        }
    });
    et_build_opts opts;
    std::memset(&opts, 0, sizeof opts);
    opts.n_seg = 1;
    opts.variants = (scores_fr ? ET_VARIANT_FR : 0) | (scores_f64 ? ET_VARIANT_F64 : 0);
    opts.device = -1;
    if (!opts.variants) opts.variants = ET_VARIANT_FR;
    et_graph* g = nullptr;
    if ((rc = et_graph_build((uint32_t)parts.size(), nullptr, n_att, from.data(), to.data(), val.data(), nullptr, &opts, &g))) return rc;
    struct Guard { et_graph* g; ~Guard() { et_graph_free(g); } } guard{g};
    if (scores_fr && (rc = et_graph_converge_fr(g, num_iterations, initial_lo, initial_hi, scores_fr))) return rc;
    if (scores_f64) {
        const double init = (double)initial_lo + 18446744073709551616.0 * (double)initial_hi;
        if ((rc = et_graph_converge_f64(g, num_iterations, init, 0.0, nullptr, scores_f64, nullptr))) return rc;
    }
    if (addresses) for (size_t i = 0; i < parts.size(); ++i) std::memcpy(addresses + 20 * i, parts[i].data(), 20);
    return ET_OK;
}

}  // extern "C"
