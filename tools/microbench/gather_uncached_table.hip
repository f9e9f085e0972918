// Microbenchmark: random 32-byte gathers from a 32 MB table that lives in ordinary device memory, in memory
// allocated uncached (hipDeviceMallocUncached) or fine-grained: launch time, and -- under
// `rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum` --
// the size of the fabric requests each allocation kind produces.  The kernels are named by mode.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k_gather32(const uint32_t* __restrict__ col, const uint4* __restrict__ x, uint4* out, size_t n) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x, stride=(size_t)gridDim.x*256;
  uint4 acc = make_uint4(0,0,0,0);
  constexpr int U = 4;
  for (size_t i=tid; i+(U-1)*stride < n; i+=U*stride) {
    uint32_t c[U]; uint4 lo[U], hi[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) { lo[u]=x[2*(size_t)c[u]]; hi[u]=x[2*(size_t)c[u]+1]; }
#pragma unroll
    for(int u=0;u<U;++u) { acc.x^=lo[u].x^hi[u].x; acc.y+=lo[u].y+hi[u].w; }
  }
  if (acc.x==0x12345678u && acc.y==77u) out[tid]=acc;
}
template<typename F> float timeit(F f, int reps=10){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<2;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  const size_t n = 16777216, T = 1u<<20;
  std::vector<uint32_t> h(n);
  std::mt19937_64 rng(1);
  for (auto& v: h) v = rng()%T;
  uint32_t* col; uint4* out; uint4 *x0=nullptr,*x1=nullptr,*x2=nullptr;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&out,(size_t)2048*256*16));
  CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice));
  CK(hipMalloc(&x0,T*32));
  hipError_t e1 = hipExtMallocWithFlags((void**)&x1, T*32, hipDeviceMallocUncached);
  hipError_t e2 = hipExtMallocWithFlags((void**)&x2, T*32, hipDeviceMallocFinegrained);
  printf("uncached alloc: %s, fine-grained alloc: %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  CK(hipMemset(x0,1,T*32));
  float a = timeit([&]{ hipLaunchKernelGGL((k_gather32<0>),dim3(2048),dim3(256),0,0,col,x0,out,n); });
  printf("ordinary memory   : %.1f us\n", a);
  if (e1==hipSuccess) { CK(hipMemset(x1,1,T*32)); float b = timeit([&]{ hipLaunchKernelGGL((k_gather32<1>),dim3(2048),dim3(256),0,0,col,x1,out,n); }); printf("uncached memory   : %.1f us\n", b); }
  if (e2==hipSuccess) { CK(hipMemset(x2,1,T*32)); float c = timeit([&]{ hipLaunchKernelGGL((k_gather32<2>),dim3(2048),dim3(256),0,0,col,x2,out,n); }); printf("fine-grained memory: %.1f us\n", c); }
  return 0;
}
