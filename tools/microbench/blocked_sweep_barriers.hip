// Microbenchmark 7: blocked sweep with the wavefronts of a workgroup held together by barriers
// (every batch, every second batch), with and without requesting the next batch before the barrier,
// for 1, 2 and 4 workgroups per CU.  256 workgroups: 61.0 us free-running, 56.4 us with a barrier
// every second batch; more workgroups per CU are slower (72 / 86 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int U, int SYNC_EVERY, bool PIPE>
__global__ __launch_bounds__(1024) void k_blocked(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                  const double* __restrict__ x, double* __restrict__ y, uint32_t edges_per_block, uint32_t rows_per_block) {
  extern __shared__ double acc[];
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) acc[i]=0.0;
  __syncthreads();
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  if (!PIPE) {
    int k=0;
    for (uint32_t e=threadIdx.x; e + (U-1)*1024 < edges_per_block; e+=U*1024, ++k) {
      uint32_t c[U]; uint8_t ww[U]; uint16_t r[U]; double xv[U];
#pragma unroll
      for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
      for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
      for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
      if (SYNC_EVERY && (k % SYNC_EVERY)==SYNC_EVERY-1) __syncthreads();
    }
  } else {
    // stream records of batch k+1 are requested before the barrier that ends batch k
    uint32_t c[U]; uint8_t ww[U]; uint16_t r[U];
    uint32_t e=threadIdx.x;
    const uint32_t nb = edges_per_block/(U*1024);
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
    for (uint32_t k=0;k<nb;++k) {
      double xv[U];
#pragma unroll
      for(int u=0;u<U;++u) xv[u]=x[c[u]];
      uint32_t c2[U]; uint8_t w2[U]; uint16_t r2[U];
      const uint32_t e2 = e + U*1024;
      if (k+1<nb) {
#pragma unroll
        for(int u=0;u<U;++u) { c2[u]=col[base+e2+u*1024]; w2[u]=w[base+e2+u*1024]; r2[u]=lrow[base+e2+u*1024]; }
      }
#pragma unroll
      for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
      if (SYNC_EVERY && (k % SYNC_EVERY)==SYNC_EVERY-1) __syncthreads();
      if (k+1<nb) {
#pragma unroll
        for(int u=0;u<U;++u) { c[u]=c2[u]; ww[u]=w2[u]; r[u]=r2[u]; }
      }
      e=e2;
    }
  }
  __syncthreads();
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) y[(size_t)blockIdx.x*rows_per_block+i]=acc[i];
}
template<typename F> float timeit(F f, int reps=20){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<3;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  const size_t n=16777216; const uint32_t T=1u<<20;
  uint32_t* col; uint8_t* w; uint16_t* lrow; double* x; double* y;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*8)); CK(hipMalloc(&y,(size_t)T*8));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,0,(size_t)T*8));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  for (int nblk : {256, 512, 1024}) {
  uint32_t epb=n/nblk, rpb=T/nblk;
  for (int b=0;b<nblk;++b) { uint32_t* p=h.data()+(size_t)b*epb; for (uint32_t e=0;e<epb;++e) p[e]=rng()%T; std::sort(p,p+epb); }
  for (auto& v: hr) v=rng()%rpb;
  CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
#define RUN(U,S,P) printf("blocks %d U=%d sync_every=%d pipe=%d : %.1f us\n",nblk,U,S,P,timeit([&]{ hipLaunchKernelGGL((k_blocked<U,S,P>),dim3(nblk),dim3(1024),rpb*8,0,col,w,lrow,x,y,epb,rpb); }));
  RUN(4,0,false) RUN(4,1,false) RUN(4,2,false) RUN(2,1,false) RUN(2,2,false) RUN(8,1,false) RUN(4,1,true)
  }
  return 0;
}
