// Microbenchmark: the exact variant as a blocked sweep WITH its per-row field arithmetic, at several workgroup
// shapes: does the arithmetic of one workgroup hide behind the sweep of another when two or more share a CU?
// Per edge: 7 B record, one 32-byte gather, eight 64-bit LDS atomic adds; per row: fold the eight column sums,
// one 2^288 Montgomery reduction, one field multiply (the epilogue of k_step_fr_raw).  Uniform random sources.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include "../../protocol_amd/csrc/fr_dev.hpp"
using namespace et;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int THREADS, bool EPILOGUE>
__global__ __launch_bounds__(THREADS) void k_blocked(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                     const uint4* __restrict__ x, uint4* __restrict__ y, const uint4* __restrict__ scale,
                                                     uint32_t edges_per_block, uint32_t rows_per_block) {
  extern __shared__ unsigned long long acc[];   // [8][rows_per_block]
  for (uint32_t i=threadIdx.x;i<8*rows_per_block;i+=THREADS) acc[i]=0;
  __syncthreads();
  constexpr int U = 4;
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  const uint32_t nb = edges_per_block/(U*THREADS);
  uint32_t e=threadIdx.x;
  for (uint32_t k=0;k<nb;++k, e+=U*THREADS) {
    uint32_t c[U]; uint32_t ww[U]; uint32_t r[U]; uint4 lo[U], hi[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*THREADS]; ww[u]=w[base+e+u*THREADS]; r[u]=lrow[base+e+u*THREADS]; }
#pragma unroll
    for(int u=0;u<U;++u) { lo[u]=x[2*(size_t)c[u]]; hi[u]=x[2*(size_t)c[u]+1]; }
#pragma unroll
    for(int u=0;u<U;++u) {
      unsigned long long* a = acc + r[u];
      atomicAdd(a + 0*rows_per_block, (unsigned long long)lo[u].x*ww[u]);
      atomicAdd(a + 1*rows_per_block, (unsigned long long)lo[u].y*ww[u]);
      atomicAdd(a + 2*rows_per_block, (unsigned long long)lo[u].z*ww[u]);
      atomicAdd(a + 3*rows_per_block, (unsigned long long)lo[u].w*ww[u]);
      atomicAdd(a + 4*rows_per_block, (unsigned long long)hi[u].x*ww[u]);
      atomicAdd(a + 5*rows_per_block, (unsigned long long)hi[u].y*ww[u]);
      atomicAdd(a + 6*rows_per_block, (unsigned long long)hi[u].z*ww[u]);
      atomicAdd(a + 7*rows_per_block, (unsigned long long)hi[u].w*ww[u]);
    }
    if (k & 1) __syncthreads();
  }
  __syncthreads();
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=THREADS) {
    const size_t row = (size_t)blockIdx.x*rows_per_block+i;
    if (EPILOGUE) {
      Acc full = acc_zero();
      uint64_t c = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { c += acc[i + q*rows_per_block] & 0xFFFFFFFFull; full.v[q] = (uint32_t)c; c = (c >> 32) + (acc[i + q*rows_per_block] >> 32); }
      full.v[8] = (uint32_t)c; full.v[9] = (uint32_t)(c >> 32);
      U256 v = acc_redc288(full);
      U256 out = fr_mul_redc288(v, u256_load(scale + 2*row));
      u256_store(y + 2*row, out);
    } else {
      unsigned long long s0=acc[i]^acc[i+rows_per_block]^acc[i+2*rows_per_block]^acc[i+3*rows_per_block];
      y[2*row]=make_uint4((uint32_t)s0,(uint32_t)(s0>>32),1,2); y[2*row+1]=make_uint4(1,2,3,4);
    }
  }
}
template<typename F> float timeit(F f, int reps=10){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<3;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
template<int THREADS> void run(int nblk, const uint32_t* col, const uint8_t* w, const uint16_t* lrow, const uint4* x, uint4* y, const uint4* scale, size_t n, uint32_t T) {
  const uint32_t epb=n/nblk, rpb=T/nblk;
  const size_t lds=(size_t)rpb*64;
  CK(hipFuncSetAttribute((const void*)k_blocked<THREADS,true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
  CK(hipFuncSetAttribute((const void*)k_blocked<THREADS,false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
  float a = timeit([&]{ hipLaunchKernelGGL((k_blocked<THREADS,false>),dim3(nblk),dim3(THREADS),lds,0,col,w,lrow,x,y,scale,epb,rpb); });
  float b = timeit([&]{ hipLaunchKernelGGL((k_blocked<THREADS,true>),dim3(nblk),dim3(THREADS),lds,0,col,w,lrow,x,y,scale,epb,rpb); });
  printf("%4d workgroups x %4d threads, %4u rows each (LDS %3zu KB): sweep only %.1f us, with the row arithmetic %.1f us\n", nblk, THREADS, rpb, lds/1024, a, b);
}
int main(){
  const size_t n=16777216; const uint32_t T=1u<<20;
  uint32_t* col; uint8_t* w; uint16_t* lrow; uint4 *x, *y, *scale;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*32)); CK(hipMalloc(&y,(size_t)T*32)); CK(hipMalloc(&scale,(size_t)T*32));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,1,(size_t)T*32)); CK(hipMemset(scale,3,(size_t)T*32));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  for (int nblk : {512, 1024, 2048}) {
    uint32_t epb=n/nblk, rpb=T/nblk;
    for (int b=0;b<nblk;++b) { uint32_t* p=h.data()+(size_t)b*epb; for (uint32_t e=0;e<epb;++e) p[e]=rng()%T; std::sort(p,p+epb); }
    for (auto& v: hr) v=rng()%rpb;
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
    if (nblk == 512) run<1024>(nblk,col,w,lrow,x,y,scale,n,T);
    if (nblk == 1024) { run<1024>(nblk,col,w,lrow,x,y,scale,n,T); run<512>(nblk,col,w,lrow,x,y,scale,n,T); }
    if (nblk == 2048) { run<512>(nblk,col,w,lrow,x,y,scale,n,T); run<256>(nblk,col,w,lrow,x,y,scale,n,T); }
  }
  return 0;
}
