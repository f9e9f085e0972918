// Microbenchmark 2: cost of random 8-byte gathers by table size and cache policy (plain / nt / sc1 / sc0 sc1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int POLICY> __device__ __forceinline__ void gload(double& v, const double* p) {
  if (POLICY==0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p));
  if (POLICY==1) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p));
  if (POLICY==2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p));
  if (POLICY==3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p));
  if (POLICY==4) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p));
}
template<int U, int POLICY>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ col, const double* __restrict__ x, double* out, size_t n) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x, stride=(size_t)gridDim.x*256;
  double acc=0;
  for (size_t i=tid; i+ (U-1)*stride < n; i+=U*stride) {
    uint32_t c[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) gload<POLICY>(xv[u], x + c[u]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for(int u=0;u<U;++u) acc+= xv[u];
  }
  if (acc==12345.678) out[tid]=acc;
}
template<typename F> float timeit(F f, int reps=20){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<3;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  size_t n = 16777216;
  std::vector<uint32_t> h(n);
  uint32_t* col; double* x; double* out;
  size_t TMAX = 1u<<24;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&x,TMAX*8)); CK(hipMalloc(&out,(size_t)4096*1024*8));
  CK(hipMemset(x,0,TMAX*8));
  std::mt19937_64 rng(1);
  for (size_t T : {size_t(1)<<10, size_t(1)<<11, size_t(1)<<12, size_t(1)<<14, size_t(1)<<17, size_t(1)<<19, size_t(1)<<20, size_t(1)<<22}) {
    for (auto& v: h) v = rng()%T;
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice));
    float a = timeit([&]{ hipLaunchKernelGGL((k_gather<8,0>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float b = timeit([&]{ hipLaunchKernelGGL((k_gather<8,1>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float c = timeit([&]{ hipLaunchKernelGGL((k_gather<8,2>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float d = timeit([&]{ hipLaunchKernelGGL((k_gather<8,3>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float e = timeit([&]{ hipLaunchKernelGGL((k_gather<8,4>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float f = timeit([&]{ hipLaunchKernelGGL((k_gather<16,0>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    float g = timeit([&]{ hipLaunchKernelGGL((k_gather<16,2>),dim3(2048),dim3(256),0,0,col,x,out,n); });
    printf("unif T=%8zu (%7.3f MB): plain %.1f | nt %.1f | sc1 %.1f | sc0sc1 %.1f | sc0 %.1f | plain U16 %.1f | sc1 U16 %.1f us\n", T, T*8/1048576.0, a,b,c,d,e,f,g);
  }
  return 0;
}
