// Do the cache-policy bits of a global load (sc0, sc1, nt) change what a gather costs?  A miss in the vector L1 fetches a
// whole 128-byte line for 8 or 32 useful bytes; this measures whether any policy makes the L1 ask for less.
// Per block a sorted list of element indices (the blocked sweep's order) over a table of 1M elements.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gcp gather_cache_policy.hip && /tmp/gcp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static const int BLOCKS = 256, THREADS = 1024, PER_BLOCK = 65536, U = 4;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define KERNEL32(NAME, BITS)                                                                                         \
    __global__ __launch_bounds__(1024) void NAME(const char* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* out) { \
        const uint32_t* my = idx + (size_t)blockIdx.x * PER_BLOCK;                                                   \
        uint32_t acc = 0;                                                                                            \
        for (int e = threadIdx.x; e < PER_BLOCK; e += U * 1024) {                                                    \
            uint32_t c[U];                                                                                           \
            u32x4 lo[U], hi[U];                                                                                      \
            _Pragma("unroll") for (int u = 0; u < U; ++u) c[u] = my[e + u * 1024];                                   \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
                const char* p = tab + (size_t)c[u] * 32;                                                             \
                asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(lo[u]) : "v"(p) : "memory");            \
                asm volatile("global_load_dwordx4 %0, %1, off offset:16 " BITS : "=v"(hi[u]) : "v"(p) : "memory");  \
            }                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
                asm volatile("" : "+v"(lo[u]), "+v"(hi[u]));                                                         \
                acc ^= lo[u].x ^ lo[u].w ^ hi[u].y ^ hi[u].z;                                                        \
            }                                                                                                        \
        }                                                                                                            \
        if (acc == 0x12345678u) out[0] = acc;                                                                        \
    }
#define KERNEL8(NAME, BITS)                                                                                          \
    __global__ __launch_bounds__(1024) void NAME(const char* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* out) { \
        const uint32_t* my = idx + (size_t)blockIdx.x * PER_BLOCK;                                                   \
        uint32_t acc = 0;                                                                                            \
        for (int e = threadIdx.x; e < PER_BLOCK; e += U * 1024) {                                                    \
            uint32_t c[U];                                                                                           \
            u32x2 v[U];                                                                                              \
            _Pragma("unroll") for (int u = 0; u < U; ++u) c[u] = my[e + u * 1024];                                   \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
                const char* p = tab + (size_t)c[u] * 8;                                                              \
                asm volatile("global_load_dwordx2 %0, %1, off " BITS : "=v"(v[u]) : "v"(p) : "memory");             \
            }                                                                                                        \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
                asm volatile("" : "+v"(v[u]));                                                                       \
                acc ^= v[u].x ^ v[u].y;                                                                              \
            }                                                                                                        \
        }                                                                                                            \
        if (acc == 0x12345678u) out[0] = acc;                                                                        \
    }

KERNEL32(k32_plain, "")
KERNEL32(k32_nt, "nt")
KERNEL32(k32_sc0, "sc0")
KERNEL32(k32_sc1, "sc1")
KERNEL32(k32_sc0_sc1, "sc0 sc1")
KERNEL32(k32_sc0_nt, "sc0 nt")
KERNEL32(k32_sc1_nt, "sc1 nt")
KERNEL32(k32_all, "sc0 sc1 nt")
KERNEL8(k8_plain, "")
KERNEL8(k8_nt, "nt")
KERNEL8(k8_sc0, "sc0")
KERNEL8(k8_sc1, "sc1")
KERNEL8(k8_sc0_sc1, "sc0 sc1")
KERNEL8(k8_sc0_nt, "sc0 nt")
KERNEL8(k8_sc1_nt, "sc1 nt")
KERNEL8(k8_all, "sc0 sc1 nt")

typedef void (*Kern)(const char*, const uint32_t*, uint32_t*);

static float time_it(Kern k, const char* tab, const uint32_t* idx, uint32_t* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(BLOCKS), dim3(THREADS), 0, 0, tab, idx, out);
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(BLOCKS), dim3(THREADS), 0, 0, tab, idx, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 20 * 1000.f;
}

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;       // run one variant only (for a counter pass)
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    const size_t entries = size_t(1) << 20;
    std::mt19937_64 rng(7);
    std::vector<uint32_t> idx((size_t)BLOCKS * PER_BLOCK);
    for (int b = 0; b < BLOCKS; ++b) {
        uint32_t* p = idx.data() + (size_t)b * PER_BLOCK;
        for (int i = 0; i < PER_BLOCK; ++i) p[i] = (uint32_t)(rng() % entries);
        std::sort(p, p + PER_BLOCK);
    }
    uint32_t* d_idx; char* d_tab;
    CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_tab, entries * 32)); CK(hipMemset(d_tab, 1, entries * 32));
    struct V { const char* name; Kern k32, k8; } vs[] = {
        {"plain", k32_plain, k8_plain}, {"nt", k32_nt, k8_nt}, {"sc0", k32_sc0, k8_sc0}, {"sc1", k32_sc1, k8_sc1},
        {"sc0 sc1", k32_sc0_sc1, k8_sc0_sc1}, {"sc0 nt", k32_sc0_nt, k8_sc0_nt}, {"sc1 nt", k32_sc1_nt, k8_sc1_nt},
        {"sc0 sc1 nt", k32_all, k8_all}};
    int i = 0;
    for (const V& v : vs) {
        if (only >= 0 && only != i++) continue;
        float t32 = time_it(v.k32, d_tab, d_idx, d_out);
        float t8 = time_it(v.k8, d_tab, d_idx, d_out);
        printf("%-12s : 32-byte gathers %6.1f us | 8-byte gathers %6.1f us   (16.8M gathers, 1M-element table)\n", v.name, t32, t8);
    }
    return 0;
}
