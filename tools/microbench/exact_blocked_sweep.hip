// Microbenchmark 8: would a blocked sweep pay for the exact (field) variant?  The sweep alone: 122 us for
// two rounds of 256 workgroups (row-per-lane kernel: 196 us in total), but the per-row field arithmetic
// (~40 us) cannot overlap it -- one workgroup fills a CU -- and the engine version came out at 203 us.
// Per edge: 7 B record, one 32-byte gather,
// eight 64-bit LDS atomic adds (32-bit limb x 8-bit opinion into per-row column sums).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int U, int SYNC_EVERY>
__global__ __launch_bounds__(1024) void k_blocked(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                  const uint4* __restrict__ x, uint4* __restrict__ y, uint32_t edges_per_block, uint32_t rows_per_block) {
  extern __shared__ unsigned long long acc[];   // [8][rows_per_block]
  for (uint32_t i=threadIdx.x;i<8*rows_per_block;i+=1024) acc[i]=0;
  __syncthreads();
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  const uint32_t nb = edges_per_block/(U*1024);
  uint32_t e=threadIdx.x;
  for (uint32_t k=0;k<nb;++k, e+=U*1024) {
    uint32_t c[U]; uint32_t ww[U]; uint32_t r[U]; uint4 lo[U], hi[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
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
    if (SYNC_EVERY && (k % SYNC_EVERY)==SYNC_EVERY-1) __syncthreads();
  }
  __syncthreads();
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) {
    unsigned long long s0=acc[i]^acc[i+rows_per_block]^acc[i+2*rows_per_block]^acc[i+3*rows_per_block];
    unsigned long long s1=acc[i+4*rows_per_block]^acc[i+5*rows_per_block]^acc[i+6*rows_per_block]^acc[i+7*rows_per_block];
    y[2*((size_t)blockIdx.x*rows_per_block+i)]=make_uint4((uint32_t)s0,(uint32_t)(s0>>32),(uint32_t)s1,(uint32_t)(s1>>32));
    y[2*((size_t)blockIdx.x*rows_per_block+i)+1]=make_uint4(1,2,3,4);
  }
}
template<typename F> float timeit(F f, int reps=10){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<3;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  const size_t n=16777216; const uint32_t T=1u<<20;
  uint32_t* col; uint8_t* w; uint16_t* lrow; uint4* x; uint4* y;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*32)); CK(hipMalloc(&y,(size_t)T*32));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,1,(size_t)T*32));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  for (int nblk : {512, 1024}) {
    uint32_t epb=n/nblk, rpb=T/nblk;
    for (int b=0;b<nblk;++b) { uint32_t* p=h.data()+(size_t)b*epb; for (uint32_t e=0;e<epb;++e) p[e]=rng()%T; std::sort(p,p+epb); }
    for (auto& v: hr) v=rng()%rpb;
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
    const size_t lds=(size_t)rpb*64;
    CK(hipFuncSetAttribute((const void*)k_blocked<4,0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
    CK(hipFuncSetAttribute((const void*)k_blocked<4,1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
    CK(hipFuncSetAttribute((const void*)k_blocked<4,2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
    CK(hipFuncSetAttribute((const void*)k_blocked<2,2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
#define RUN(U,S) printf("blocks %d (lds %zu KB) U=%d sync_every=%d : %.1f us\n",nblk,lds/1024,U,S,timeit([&]{ hipLaunchKernelGGL((k_blocked<U,S>),dim3(nblk),dim3(1024),lds,0,col,w,lrow,x,y,epb,rpb); }));
    RUN(4,0) RUN(4,1) RUN(4,2) RUN(2,2)
  }
  return 0;
}
