// Microbenchmark 3: does a synchronised ascending sweep (every lane walks its own sorted index list)
// keep an 8 MB / 32 MB gather table L2-resident?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int U, int ESZ, bool NT>
__global__ __launch_bounds__(256) void k_sweep(const uint32_t* __restrict__ col, const double* __restrict__ w, const double* __restrict__ x, double* out, int L, size_t nt) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x;
  double acc=0;
  for (int k=0;k+U<=L;k+=U) {
    uint32_t c[U]; double ww[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]= NT ? __builtin_nontemporal_load(col+(size_t)(k+u)*nt+tid) : col[(size_t)(k+u)*nt+tid];
#pragma unroll
    for(int u=0;u<U;++u) ww[u]= NT ? __builtin_nontemporal_load(w+(size_t)(k+u)*nt+tid) : w[(size_t)(k+u)*nt+tid];
#pragma unroll
    for(int u=0;u<U;++u) { xv[u]=x[(size_t)c[u]*ESZ]; if (ESZ==4) xv[u]+=x[(size_t)c[u]*ESZ+2]; }
#pragma unroll
    for(int u=0;u<U;++u) acc+= ww[u]*xv[u];
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
  size_t T = 1u<<20;
  uint32_t* col; double* w; double* x; double* out;
  size_t n = 16777216;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n*8)); CK(hipMalloc(&x,T*8*4)); CK(hipMalloc(&out,n*8/16));
  CK(hipMemset(w,0,n*8)); CK(hipMemset(x,0,T*8*4));
  std::mt19937_64 rng(1);
  for (int blocks : {1024, 2048, 4096}) {
    size_t nt = (size_t)blocks*256; int L = (int)(n/nt);
    std::vector<uint32_t> h(n);
    for (int sorted=0; sorted<2; ++sorted) {
      std::vector<uint32_t> tmp(L);
      for (size_t t=0;t<nt;++t) {
        for (int k=0;k<L;++k) tmp[k]=rng()%T;
        if (sorted) std::sort(tmp.begin(),tmp.end());
        for (int k=0;k<L;++k) h[(size_t)k*nt+t]=tmp[k];
      }
      CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice));
      float a = timeit([&]{ hipLaunchKernelGGL((k_sweep<4,1,false>),dim3(blocks),dim3(256),0,0,col,w,x,out,L,nt); });
      float b = timeit([&]{ hipLaunchKernelGGL((k_sweep<4,1,true>),dim3(blocks),dim3(256),0,0,col,w,x,out,L,nt); });
      float c = timeit([&]{ hipLaunchKernelGGL((k_sweep<4,4,false>),dim3(blocks),dim3(256),0,0,col,w,x,out,L,nt); });
      float d = timeit([&]{ hipLaunchKernelGGL((k_sweep<4,4,true>),dim3(blocks),dim3(256),0,0,col,w,x,out,L,nt); });
      printf("blocks %d L %d %s: 8B entries (8MB table) plain %.1f us, nt-stream %.1f | 32B entries (32MB table) plain %.1f, nt-stream %.1f\n", blocks, L, sorted?"sorted-sweep":"random      ", a,b,c,d);
    }
  }
  return 0;
}
