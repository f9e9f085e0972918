// What does it cost to keep a whole converge inside ONE launch?  An iteration of the step kernels ends with every
// workgroup's rows written and begins with every workgroup reading all of them, so a converge in one launch needs a
// barrier across the grid per iteration.  This prices that barrier against the gap between two launches of the same
// grid on one stream.
//
//   launches : K back-to-back launches of a kernel in which every workgroup writes its slice of a vector and reads a
//              few entries other workgroups wrote in the previous launch;
//   barrier  : ONE launch doing the same K rounds, the workgroups meeting at a barrier between rounds: one thread per
//              workgroup publishes its stores (agent-scope release), increments a counter in device memory and polls it
//              (agent-scope acquire), the others wait at the workgroup's own barrier.
// The grid is one workgroup of 1024 threads per compute unit, as the sweeps launch (all of them resident at once, which
// a grid barrier needs); the poll is bounded, so a grid that cannot be resident ends with an error instead of hanging.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/gb tools/microbench/grid_barrier_vs_launch.hip && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double round_of_work(const double* __restrict__ in, double* __restrict__ out, uint32_t n, uint32_t round) {
    // every thread: one entry of this workgroup's slice, from four entries scattered over the other workgroups' slices
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) s += in[(i * 2654435761u + k * 40503u + round) % n];
    out[i] = s * 0.25 + 1.0;
    return s;
}

__global__ __launch_bounds__(1024) void k_round(const double* in, double* out, uint32_t n, uint32_t round) {
    round_of_work(in, out, n, round);
}

__global__ __launch_bounds__(1024) void k_all_rounds(double* a, double* b, uint32_t n, uint32_t rounds, unsigned* counter, unsigned* failed) {
    for (uint32_t r = 0; r < rounds; ++r) {
        round_of_work(r & 1 ? b : a, r & 1 ? a : b, n, r);
        // barrier across the grid: stores of this round visible to every workgroup before the next round's loads
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (r + 1) * gridDim.x;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 4000000u) { *failed = 1u; break; }      // never spin for ever: report and go on
                __builtin_amdgcn_s_sleep(2);
            }
            __threadfence();
        }
        __syncthreads();
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount, threads = 1024, n = blocks * threads, rounds = 20, reps = 50;
    double *a = nullptr, *b = nullptr;
    unsigned *counter = nullptr, *failed = nullptr;
    CHECK(hipMalloc(&a, n * sizeof(double)));
    CHECK(hipMalloc(&b, n * sizeof(double)));
    CHECK(hipMalloc(&counter, sizeof(unsigned)));
    CHECK(hipMalloc(&failed, sizeof(unsigned)));
    CHECK(hipMemset(a, 0, n * sizeof(double)));
    CHECK(hipMemset(b, 0, n * sizeof(double)));
    CHECK(hipMemset(failed, 0, sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    // K launches per converge
    for (uint32_t rep = 0; rep < reps + 5; ++rep) {
        if (rep == 5) CHECK(hipEventRecord(e0, 0));
        for (uint32_t r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_round, dim3(blocks), dim3(threads), 0, 0, r & 1 ? b : a, r & 1 ? a : b, n, r);
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double launches_us = ms * 1e3 / (reps * rounds);
    std::vector<double> ref(n);
    CHECK(hipMemcpy(ref.data(), rounds & 1 ? b : a, n * sizeof(double), hipMemcpyDeviceToHost));
    // one launch per converge, a grid barrier per round
    CHECK(hipMemset(a, 0, n * sizeof(double)));
    CHECK(hipMemset(b, 0, n * sizeof(double)));
    for (uint32_t rep = 0; rep < reps + 5; ++rep) {
        if (rep == 5) CHECK(hipEventRecord(e0, 0));
        CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), 0));
        hipLaunchKernelGGL(k_all_rounds, dim3(blocks), dim3(threads), 0, 0, a, b, n, rounds, counter, failed);
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double barrier_us = ms * 1e3 / (reps * rounds);
    unsigned f = 0;
    CHECK(hipMemcpy(&f, failed, sizeof(unsigned), hipMemcpyDeviceToHost));
    // the two forms compute the same vector when the barrier makes every round's stores visible (every converge restarts
    // from what the one before left, in both forms, so only the last converge's input differs: compare one converge each)
    CHECK(hipMemset(a, 0, n * sizeof(double)));
    CHECK(hipMemset(b, 0, n * sizeof(double)));
    for (uint32_t r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_round, dim3(blocks), dim3(threads), 0, 0, r & 1 ? b : a, r & 1 ? a : b, n, r);
    CHECK(hipMemcpy(ref.data(), rounds & 1 ? b : a, n * sizeof(double), hipMemcpyDeviceToHost));
    CHECK(hipMemset(a, 0, n * sizeof(double)));
    CHECK(hipMemset(b, 0, n * sizeof(double)));
    CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned), 0));
    hipLaunchKernelGGL(k_all_rounds, dim3(blocks), dim3(threads), 0, 0, a, b, n, rounds, counter, failed);
    std::vector<double> got(n);
    CHECK(hipMemcpy(got.data(), rounds & 1 ? b : a, n * sizeof(double), hipMemcpyDeviceToHost));
    uint32_t differ = 0;
    for (uint32_t i = 0; i < n; ++i) differ += got[i] != ref[i];
    printf("{\"workgroups\": %u, \"rounds\": %u, \"us_per_round_as_launches\": %.2f, \"us_per_round_with_grid_barrier\": %.2f, "
           "\"barrier_gave_up\": %u, \"entries_that_differ\": %u}\n", blocks, rounds, launches_us, barrier_us, f, differ);
    return 0;
}
