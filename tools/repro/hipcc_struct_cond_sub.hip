// Re-check of a device-compiler pitfall met while writing the signature-recovery kernel (ecdsa.hip).
//
// A modular reduction written on the whole 8-limb struct,
//        if (geq(a, m)) a = sub(a, m);
// gave wrong results from hipcc at -O2 / -O3 in some inlining contexts (a sum of products came out off by the
// modulus' low limb in limb 0 only; -O1 and the host compilation of the same source were right).  The library
// therefore computes the difference and its borrow once and selects limb by limb (f_cond_sub in ecdsa.hip,
// u256_sub_p_if_geq in fr_dev.hpp).  This program runs both forms on the same inputs: a five-product
// accumulation mod a run-time 256-bit modulus, reduced (a) with the whole-struct form and (b) with the
// limb-select form, each compared with a host computation in unsigned __int128.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/repro hipcc_struct_cond_sub.hip && /tmp/repro
// prints per form "ok" or the number of wrong lanes.  After a toolchain change: if form (a) is "ok" at every
// optimisation level the workaround is no longer needed (the guard test stays: tests/test_gpu_recover.py).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct F256 { uint32_t v[8]; };
struct Mod { uint32_t m[8]; uint32_t n0; };

#define DEV __host__ __device__ __forceinline__

DEV bool geq(const F256& a, const uint32_t* m) {
    uint64_t br = 0;
    for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)a.v[i] - m[i] - br; br = (d >> 32) & 1; }
    return br == 0;
}
DEV F256 sub_m(const F256& a, const uint32_t* m) {
    F256 d; uint64_t br = 0;
    for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a.v[i] - m[i] - br; d.v[i] = (uint32_t)t; br = (t >> 32) & 1; }
    return d;
}
DEV F256 cond_sub_select(const F256& a, const uint32_t* m, bool force) {
    F256 d; uint64_t br = 0;
    for (int i = 0; i < 8; ++i) { uint64_t t = (uint64_t)a.v[i] - m[i] - br; d.v[i] = (uint32_t)t; br = (t >> 32) & 1; }
    const bool take = force || br == 0;
    F256 r;
    for (int i = 0; i < 8; ++i) r.v[i] = take ? d.v[i] : a.v[i];
    return r;
}
// sum of five 8x8-limb products (17 limbs), eight Montgomery rounds, then ONE conditional subtraction
template <bool WHOLE_STRUCT>
DEV F256 dot5(const F256* a, const F256* b, const Mod& M) {
    uint32_t t[18];
    for (int i = 0; i < 18; ++i) t[i] = 0;
    for (int k = 0; k < 5; ++k) {
        for (int i = 0; i < 8; ++i) {
            uint64_t c = 0;
            for (int j = 0; j < 8; ++j) {
                uint64_t x = (uint64_t)a[k].v[j] * b[k].v[i] + t[i + j] + c;
                t[i + j] = (uint32_t)x; c = x >> 32;
            }
            for (int j = i + 8; j < 18 && c; ++j) { uint64_t x = (uint64_t)t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
        }
    }
    for (int i = 0; i < 8; ++i) {
        const uint32_t q = t[i] * M.n0;
        uint64_t c = 0;
        for (int j = 0; j < 8; ++j) { uint64_t x = (uint64_t)q * M.m[j] + t[i + j] + c; t[i + j] = (uint32_t)x; c = x >> 32; }
        for (int j = i + 8; j < 18 && c; ++j) { uint64_t x = (uint64_t)t[j] + c; t[j] = (uint32_t)x; c = x >> 32; }
    }
    // value = t[8..16] < 6m: bring below m with up to five subtractions
    F256 r;
    for (int i = 0; i < 8; ++i) r.v[i] = t[8 + i];
    uint32_t top = t[16];
    for (int s = 0; s < 6; ++s) {
        if (WHOLE_STRUCT) {
            if (top || geq(r, M.m)) {                 // the form that went wrong
                uint64_t br = 0;
                for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)r.v[i] - M.m[i] - br; br = (d >> 32) & 1; }
                r = sub_m(r, M.m);
                top -= (uint32_t)br;
            }
        } else {
            uint64_t br = 0;
            for (int i = 0; i < 8; ++i) { uint64_t d = (uint64_t)r.v[i] - M.m[i] - br; br = (d >> 32) & 1; }
            const bool take = top != 0 || br == 0;
            r = cond_sub_select(r, M.m, top != 0);
            if (take) top -= (uint32_t)br;
        }
    }
    return r;
}
template <bool WHOLE_STRUCT>
__global__ void k(const F256* a, const F256* b, F256* out, Mod M, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dot5<WHOLE_STRUCT>(a + 5 * (size_t)i, b + 5 * (size_t)i, M);
}

// host check: (sum a_k b_k) * 2^-256 mod m with 64-bit limbs
typedef unsigned __int128 u128;
static void host_ref(const F256* a, const F256* b, const Mod& M, F256& out) {
    // plain big integers through the same routine compiled for the host (the host build was always right) ...
    out = dot5<false>(a, b, M);
    // ... and cross-checked against the whole-struct form on the host
    F256 o2 = dot5<true>(a, b, M);
    for (int i = 0; i < 8; ++i) if (o2.v[i] != out.v[i]) { printf("host forms disagree\n"); exit(2); }
}

int main() {
    const int n = 1 << 16;
    // secp256k1 base field: the modulus the kernel met the problem with
    Mod M = {{0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, 0};
    uint32_t inv = 1;
    for (int i = 0; i < 5; ++i) inv *= 2 - M.m[0] * inv;
    M.n0 = ~inv + 1;
    std::mt19937 rng(5);
    std::vector<F256> a(5 * (size_t)n), b(5 * (size_t)n), want(n), got(n);
    for (auto* v : {&a, &b})
        for (auto& x : *v) {
            for (int i = 0; i < 8; ++i) x.v[i] = rng();
            if (rng() % 4 == 0) for (int i = 1; i < 8; ++i) x.v[i] = 0xFFFFFFFFu;   // values next to the modulus
            x.v[7] &= 0xFFFFFFFEu;                                                  // below m
        }
    for (int i = 0; i < n; ++i) host_ref(&a[5 * (size_t)i], &b[5 * (size_t)i], M, want[i]);
    F256 *da, *db, *dout;
    if (hipMalloc(&da, a.size() * 32) || hipMalloc(&db, b.size() * 32) || hipMalloc(&dout, (size_t)n * 32)) { printf("no device memory\n"); return 1; }
    hipMemcpy(da, a.data(), a.size() * 32, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 32, hipMemcpyHostToDevice);
    int bad_total = 0;
    for (int form = 0; form < 2; ++form) {
        if (form == 0) hipLaunchKernelGGL(k<true>, dim3(n / 256), dim3(256), 0, 0, da, db, dout, M, n);
        else hipLaunchKernelGGL(k<false>, dim3(n / 256), dim3(256), 0, 0, da, db, dout, M, n);
        if (hipMemcpy(got.data(), dout, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
        int bad = 0;
        for (int i = 0; i < n; ++i)
            for (int l = 0; l < 8; ++l) if (got[i].v[l] != want[i].v[l]) { ++bad; break; }
        printf("%s: %s", form == 0 ? "(a) whole-struct conditional subtraction" : "(b) limb-select conditional subtraction", bad ? "" : "ok\n");
        if (bad) printf("%d of %d lanes WRONG\n", bad, n);
        if (form == 1) bad_total = bad;
    }
    return bad_total ? 1 : 0;   // only the form the library uses decides the exit status
}
