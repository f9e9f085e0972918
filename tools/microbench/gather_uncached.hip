// Microbenchmark 5: random gathers from a table in cached (hipMalloc) versus uncached
// (hipDeviceMallocUncached) device memory: does an L2-bypassing table cut the 128-byte line fetch per miss?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int U, int ESZ>   // ESZ doubles per entry (1 = 8 B, 4 = 32 B)
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ col, const double* __restrict__ x, double* out, size_t n) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x, stride=(size_t)gridDim.x*256;
  double acc=0;
  for (size_t i=tid; i+ (U-1)*stride < n; i+=U*stride) {
    uint32_t c[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
    if (ESZ==1) {
      double xv[U];
#pragma unroll
      for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
      for(int u=0;u<U;++u) acc+=xv[u];
    } else {
      double4 xv[U];
#pragma unroll
      for(int u=0;u<U;++u) xv[u]=reinterpret_cast<const double4*>(x)[c[u]];
#pragma unroll
      for(int u=0;u<U;++u) acc+=xv[u].x+xv[u].y+xv[u].z+xv[u].w;
    }
  }
  if (acc==12345.678) out[tid]=acc;
}
template<typename F> float timeit(F f, int reps=10){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<2;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  size_t n = 16777216;
  std::vector<uint32_t> h(n);
  uint32_t* col; double* out; double* xc; double* xu;
  size_t T = 1u<<20;            // entries
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&out,(size_t)4096*256*8));
  CK(hipMalloc(&xc,T*32));
  hipError_t e = hipExtMallocWithFlags((void**)&xu, T*32, hipDeviceMallocUncached);
  if (e != hipSuccess) { printf("uncached alloc failed: %s\n", hipGetErrorString(e)); return 0; }
  CK(hipMemset(xc,0,T*32)); CK(hipMemset(xu,0,T*32));
  std::mt19937_64 rng(1);
  for (auto& v: h) v = rng()%T;
  CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice));
  float a = timeit([&]{ hipLaunchKernelGGL((k_gather<8,1>),dim3(2048),dim3(256),0,0,col,xc,out,n); });
  float b = timeit([&]{ hipLaunchKernelGGL((k_gather<8,1>),dim3(2048),dim3(256),0,0,col,xu,out,n); });
  float c = timeit([&]{ hipLaunchKernelGGL((k_gather<4,4>),dim3(2048),dim3(256),0,0,col,xc,out,n); });
  float d = timeit([&]{ hipLaunchKernelGGL((k_gather<4,4>),dim3(2048),dim3(256),0,0,col,xu,out,n); });
  printf("uniform random, 1M entries: 8B entries (8MB): cached %.1f us, uncached %.1f us | 32B entries (32MB): cached %.1f us, uncached %.1f us\n", a,b,c,d);
  return 0;
}
