// Issue rate of the 32x32+64-bit integer multiply-add (v_mad_u64_u32) against the double-precision fused multiply-add
// (v_fma_f64) and add: what a field multiplication built from either would pay per instruction.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mvf mad_vs_fma64.hip && /tmp/mvf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static const int ITERS = 4096, CHAINS = 8;

__global__ __launch_bounds__(1024) void k_mad(unsigned long long* out, uint32_t m) {
    unsigned long long a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = (unsigned long long)(uint32_t)a[c] * m + a[c];
    unsigned long long s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= a[c];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(1024) void k_fma(double* out, double m) {
    double a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], m, 1.0);
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    if (s == 0.1234567) out[0] = s;
}
__global__ __launch_bounds__(1024) void k_add(double* out, double m) {
    double a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = a[c] + m;
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c];
    if (s == 0.1234567) out[0] = s;
}
__global__ __launch_bounds__(1024) void k_mul32(uint32_t* out, uint32_t m) {
    uint32_t a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = a[c] * m + 1u;
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= a[c];
    if (s == 0x1234567) out[0] = s;
}
__global__ __launch_bounds__(1024) void k_add32(uint32_t* out, uint32_t m) {
    uint32_t a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = threadIdx.x + c;
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) a[c] = (a[c] + m) ^ (a[c] >> 3);
    uint32_t s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= a[c];
    if (s == 0x1234567) out[0] = s;
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5 * 1000.f;
}

int main() {
    void* d; CK(hipMalloc(&d, 64));
    const double per_wave_instr = (double)ITERS * CHAINS;     // instructions of the kind per wavefront
    // 256 workgroups x 16 wavefronts: 4 wavefronts per SIMD, so a SIMD issues 4 * per_wave_instr instructions
    auto report = [&](const char* name, float us, int instr_per_step) {
        printf("%-22s %8.1f us  -> %5.2f cycles per wavefront instruction at 2.4 GHz\n", name, us,
               us * 1e-6 * 2.4e9 / (4 * per_wave_instr * instr_per_step));
    };
    report("v_mad_u64_u32", time_it([&] { hipLaunchKernelGGL(k_mad, dim3(256), dim3(1024), 0, 0, (unsigned long long*)d, 12345u); }), 1);
    report("v_fma_f64", time_it([&] { hipLaunchKernelGGL(k_fma, dim3(256), dim3(1024), 0, 0, (double*)d, 1.0000001); }), 1);
    report("v_add_f64", time_it([&] { hipLaunchKernelGGL(k_add, dim3(256), dim3(1024), 0, 0, (double*)d, 1.0000001); }), 1);
    report("v_mul_lo_u32 + add", time_it([&] { hipLaunchKernelGGL(k_mul32, dim3(256), dim3(1024), 0, 0, (uint32_t*)d, 12345u); }), 1);
    report("add + shift + xor u32", time_it([&] { hipLaunchKernelGGL(k_add32, dim3(256), dim3(1024), 0, 0, (uint32_t*)d, 12345u); }), 3);
    return 0;
}
