// Minimal arbitrary-precision unsigned integers and non-negative fractions for converge_rational
// (dynamic_sets/native.rs:340-392, which uses num-bigint / num-rational).  Every quantity on that
// path is non-negative, so no sign handling is needed.  Fractions are kept reduced, as
// BigRational::new does; only the reduced final values are observable.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace et {

class BigUint {
  public:
    std::vector<uint32_t> d;  // little-endian, no trailing zero limbs; empty = 0

    BigUint() {}
    explicit BigUint(uint64_t v) {
        while (v) { d.push_back((uint32_t)v); v >>= 32; }
    }
    static BigUint from_le_u64(const uint64_t* l, int count) {
        BigUint r;
        for (int i = 0; i < count; ++i) { r.d.push_back((uint32_t)l[i]); r.d.push_back((uint32_t)(l[i] >> 32)); }
        r.trim();
        return r;
    }
    bool is_zero() const { return d.empty(); }
    bool is_one() const { return d.size() == 1 && d[0] == 1; }
    void trim() { while (!d.empty() && d.back() == 0) d.pop_back(); }
    size_t byte_len() const {
        if (d.empty()) return 0;
        uint32_t top = d.back();
        size_t b = 0;
        while (top) { ++b; top >>= 8; }
        return (d.size() - 1) * 4 + b;
    }
    void to_le_bytes(uint8_t* out, size_t len) const {
        for (size_t i = 0; i < len; ++i) {
            size_t limb = i / 4;
            out[i] = limb < d.size() ? (uint8_t)(d[limb] >> (8 * (i % 4))) : 0;
        }
    }
    static int cmp(const BigUint& a, const BigUint& b) {
        if (a.d.size() != b.d.size()) return a.d.size() < b.d.size() ? -1 : 1;
        for (size_t i = a.d.size(); i-- > 0;)
            if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
        return 0;
    }
    static BigUint add(const BigUint& a, const BigUint& b) {
        BigUint r;
        size_t n = std::max(a.d.size(), b.d.size());
        r.d.resize(n + 1);
        uint64_t c = 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t t = c + (i < a.d.size() ? a.d[i] : 0) + (i < b.d.size() ? b.d[i] : 0);
            r.d[i] = (uint32_t)t;
            c = t >> 32;
        }
        r.d[n] = (uint32_t)c;
        r.trim();
        return r;
    }
    static BigUint sub(const BigUint& a, const BigUint& b) {  // a >= b
        BigUint r;
        r.d.resize(a.d.size());
        int64_t br = 0;
        for (size_t i = 0; i < a.d.size(); ++i) {
            int64_t t = (int64_t)a.d[i] - (i < b.d.size() ? b.d[i] : 0) - br;
            br = t < 0;
            r.d[i] = (uint32_t)(t + (br ? (1LL << 32) : 0));
        }
        r.trim();
        return r;
    }
    static BigUint mul(const BigUint& a, const BigUint& b) {
        BigUint r;
        if (a.is_zero() || b.is_zero()) return r;
        r.d.assign(a.d.size() + b.d.size(), 0);
        for (size_t i = 0; i < a.d.size(); ++i) {
            uint64_t c = 0;
            for (size_t j = 0; j < b.d.size(); ++j) {
                uint64_t t = (uint64_t)a.d[i] * b.d[j] + r.d[i + j] + c;
                r.d[i + j] = (uint32_t)t;
                c = t >> 32;
            }
            r.d[i + b.d.size()] = (uint32_t)c;
        }
        r.trim();
        return r;
    }
    // Knuth algorithm D.  q = a / b, rem = a % b, b != 0.
    static void divmod(const BigUint& a, const BigUint& b, BigUint& q, BigUint& rem) {
        if (cmp(a, b) < 0) { q = BigUint(); rem = a; return; }
        if (b.d.size() == 1) {
            q.d.assign(a.d.size(), 0);
            uint64_t r = 0;
            for (size_t i = a.d.size(); i-- > 0;) {
                uint64_t cur = (r << 32) | a.d[i];
                q.d[i] = (uint32_t)(cur / b.d[0]);
                r = cur % b.d[0];
            }
            q.trim();
            rem = BigUint(r);
            return;
        }
        int s = __builtin_clz(b.d.back());
        std::vector<uint32_t> v(b.d.size()), u(a.d.size() + 1);
        for (size_t i = b.d.size(); i-- > 0;) v[i] = (b.d[i] << s) | (s && i ? b.d[i - 1] >> (32 - s) : 0);
        u[a.d.size()] = s ? a.d.back() >> (32 - s) : 0;
        for (size_t i = a.d.size(); i-- > 0;) u[i] = (a.d[i] << s) | (s && i ? a.d[i - 1] >> (32 - s) : 0);
        size_t n = v.size(), m = a.d.size() - n;
        q.d.assign(m + 1, 0);
        for (size_t j = m + 1; j-- > 0;) {
            uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
            uint64_t qhat = num / v[n - 1], rhat = num % v[n - 1];
            while (qhat >= (1ULL << 32) || qhat * v[n - 2] > ((rhat << 32) | u[j + n - 2])) {
                --qhat;
                rhat += v[n - 1];
                if (rhat >= (1ULL << 32)) break;
            }
            int64_t borrow = 0;
            uint64_t carry = 0;
            for (size_t i = 0; i < n; ++i) {
                uint64_t p = qhat * v[i] + carry;
                carry = p >> 32;
                int64_t t = (int64_t)u[i + j] - (int64_t)(uint32_t)p - borrow;
                borrow = t < 0;
                u[i + j] = (uint32_t)t;
            }
            int64_t t = (int64_t)u[j + n] - (int64_t)carry - borrow;
            borrow = t < 0;
            u[j + n] = (uint32_t)t;
            if (borrow) {
                --qhat;
                uint64_t c = 0;
                for (size_t i = 0; i < n; ++i) {
                    uint64_t x = (uint64_t)u[i + j] + v[i] + c;
                    u[i + j] = (uint32_t)x;
                    c = x >> 32;
                }
                u[j + n] += (uint32_t)c;
            }
            q.d[j] = (uint32_t)qhat;
        }
        q.trim();
        rem.d.assign(n, 0);
        for (size_t i = 0; i < n; ++i) rem.d[i] = (u[i] >> s) | (s ? (uint32_t)((uint64_t)u[i + 1] << (32 - s)) : 0);
        rem.trim();
    }
    static BigUint gcd(BigUint a, BigUint b) {
        while (!b.is_zero()) {
            BigUint q, r;
            divmod(a, b, q, r);
            a = std::move(b);
            b = std::move(r);
        }
        return a;
    }
};

struct BigFrac {  // num / den, reduced, den >= 1
    BigUint num, den;
    BigFrac() : den(1) {}
    BigFrac(const BigUint& n, const BigUint& d) : num(n), den(d) { reduce(); }
    void reduce() {
        if (num.is_zero()) { den = BigUint(1); return; }
        BigUint g = BigUint::gcd(num, den);
        if (!g.is_one()) {
            BigUint q, r;
            BigUint::divmod(num, g, q, r); num = q;
            BigUint::divmod(den, g, q, r); den = q;
        }
    }
    static BigFrac add(const BigFrac& a, const BigFrac& b) {
        if (a.num.is_zero()) return b;
        if (b.num.is_zero()) return a;
        if (BigUint::cmp(a.den, b.den) == 0) return BigFrac(BigUint::add(a.num, b.num), a.den);
        return BigFrac(BigUint::add(BigUint::mul(a.num, b.den), BigUint::mul(b.num, a.den)), BigUint::mul(a.den, b.den));
    }
    static BigFrac mul(const BigFrac& a, const BigFrac& b) {
        return BigFrac(BigUint::mul(a.num, b.num), BigUint::mul(a.den, b.den));
    }
};

}  // namespace et
