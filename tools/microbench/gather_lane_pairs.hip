// 32-byte gathers: one lane per element with two 16-byte loads, against two adjacent lanes per element with one
// 16-byte load each.  Per block a list of element indices (random, or sorted ascending: the blocked sweep's order).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/glp gather_lane_pairs.hip && /tmp/glp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static const int BLOCKS = 256, THREADS = 1024, PER_BLOCK = 65536;

template <int U>
__global__ __launch_bounds__(1024) void k_one_lane(const uint4* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* out) {
    const uint32_t* my = idx + (size_t)blockIdx.x * PER_BLOCK;
    uint32_t acc = 0;
    for (int e = threadIdx.x; e < PER_BLOCK; e += U * 1024) {
        uint32_t c[U];
        uint4 lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = my[e + u * 1024];
#pragma unroll
        for (int u = 0; u < U; ++u) { lo[u] = tab[2 * (size_t)c[u]]; hi[u] = tab[2 * (size_t)c[u] + 1]; }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= lo[u].x ^ lo[u].w ^ hi[u].y ^ hi[u].z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(1024) void k_lane_pair(const uint4* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* out) {
    const uint32_t* my = idx + (size_t)blockIdx.x * PER_BLOCK;
    const int half = threadIdx.x & 1;
    uint32_t acc = 0;
    for (int e = threadIdx.x >> 1; e < PER_BLOCK; e += U * 512) {
        uint32_t c[U];
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = my[e + u * 512];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = tab[2 * (size_t)c[u] + half];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// 8-byte gathers for comparison (the f64 sweep's access)
template <int U>
__global__ __launch_bounds__(1024) void k_f64(const double* __restrict__ tab, const uint32_t* __restrict__ idx, uint32_t* out) {
    const uint32_t* my = idx + (size_t)blockIdx.x * PER_BLOCK;
    double acc = 0;
    for (int e = threadIdx.x; e < PER_BLOCK; e += U * 1024) {
        uint32_t c[U];
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = my[e + u * 1024];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = tab[c[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 1.2345) out[0] = 1;
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 20 * 1000.f;
}

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    for (size_t entries : {size_t(256), size_t(8192), size_t(131072), size_t(1) << 20}) {
        for (int sorted = 0; sorted < 2; ++sorted) {
            std::mt19937_64 rng(7);
            std::vector<uint32_t> idx((size_t)BLOCKS * PER_BLOCK);
            for (int b = 0; b < BLOCKS; ++b) {
                uint32_t* p = idx.data() + (size_t)b * PER_BLOCK;
                for (int i = 0; i < PER_BLOCK; ++i) p[i] = (uint32_t)(rng() % entries);
                if (sorted) std::sort(p, p + PER_BLOCK);
            }
            uint32_t* d_idx; uint4* d_tab;
            CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
            CK(hipMalloc(&d_tab, entries * 32)); CK(hipMemset(d_tab, 1, entries * 32));
            float a1 = time_it([&] { hipLaunchKernelGGL(k_one_lane<1>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float b1 = time_it([&] { hipLaunchKernelGGL(k_lane_pair<1>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float b2 = time_it([&] { hipLaunchKernelGGL(k_lane_pair<2>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float f1 = time_it([&] { hipLaunchKernelGGL(k_f64<1>, dim3(BLOCKS), dim3(THREADS), 0, 0, (const double*)d_tab, d_idx, d_out); });
            float f2 = time_it([&] { hipLaunchKernelGGL(k_f64<2>, dim3(BLOCKS), dim3(THREADS), 0, 0, (const double*)d_tab, d_idx, d_out); });
            printf("entries %8zu %s : one lane U=1 %6.1f | lane pair U=1 %6.1f U=2 %6.1f | f64 U=1 %6.1f U=2 %6.1f us\n", entries,
                   sorted ? "sorted" : "random", a1, b1, b2, f1, f2);
            float a2 = time_it([&] { hipLaunchKernelGGL(k_one_lane<2>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float a4 = time_it([&] { hipLaunchKernelGGL(k_one_lane<4>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float b4 = time_it([&] { hipLaunchKernelGGL(k_lane_pair<4>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float b8 = time_it([&] { hipLaunchKernelGGL(k_lane_pair<8>, dim3(BLOCKS), dim3(THREADS), 0, 0, d_tab, d_idx, d_out); });
            float f4 = time_it([&] { hipLaunchKernelGGL(k_f64<4>, dim3(BLOCKS), dim3(THREADS), 0, 0, (const double*)d_tab, d_idx, d_out); });
            printf("entries %8zu %s : one lane U=2 %6.1f U=4 %6.1f | lane pair U=4 %6.1f U=8 %6.1f | f64 U=4 %6.1f us\n", entries,
                   sorted ? "sorted" : "random", a2, a4, b4, b8, f4);
            CK(hipFree(d_idx)); CK(hipFree(d_tab));
        }
    }
    return 0;
}
