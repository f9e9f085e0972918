// Microbenchmark 9: blocked sweep where 45 % of the edges cite 12288 hot sources -- does serving those from an
// LDS copy pay?  No: 48.9 us from memory, 49.6 us with the copy (the hot gathers already hit L1; the time goes
// into the cold gathers' 128-byte L1 fills).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
template<int U, bool TILE>
__global__ __launch_bounds__(1024) void k_blocked(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                  const double* __restrict__ x, double* __restrict__ y, uint32_t edges_per_block, uint32_t rows_per_block,
                                                  const uint32_t* __restrict__ hot_end, uint32_t H) {
  extern __shared__ double lds[];
  double* acc = lds;
  double* tile = lds + rows_per_block;
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) acc[i]=0.0;
  if (TILE) for (uint32_t i=threadIdx.x;i<H;i+=1024) tile[i]=x[i];
  __syncthreads();
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  const uint32_t nb = edges_per_block/(U*1024);
  const uint32_t hb = TILE ? hot_end[blockIdx.x]/(U*1024) : 0;   // whole batches of hot edges
  uint32_t e=threadIdx.x;
  uint32_t k=0;
  for (;k<hb;++k, e+=U*1024) {
    uint32_t c[U]; uint8_t ww[U]; uint16_t r[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
    for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*tile[c[u]]);
  }
  __syncthreads();
  for (;k<nb;++k, e+=U*1024) {
    uint32_t c[U]; uint8_t ww[U]; uint16_t r[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
    for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
    for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
    if (k&1) __syncthreads();
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
  const size_t n=16777216; const uint32_t T=1u<<20; const int nblk=256; const uint32_t H=12288;
  uint32_t* col; uint8_t* w; uint16_t* lrow; double* x; double* y; uint32_t* hot_end;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*8)); CK(hipMalloc(&y,(size_t)T*8)); CK(hipMalloc(&hot_end,nblk*4));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,0,(size_t)T*8));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n), he(nblk); std::vector<uint16_t> hr(n);
  uint32_t epb=n/nblk, rpb=T/nblk;
  for (double hot_frac : {0.0, 0.45}) {
    for (int b=0;b<nblk;++b) { uint32_t* p=h.data()+(size_t)b*epb;
      for (uint32_t e=0;e<epb;++e) p[e]= (rng()%1000 < hot_frac*1000) ? rng()%H : H + rng()%(T-H);
      std::sort(p,p+epb); he[b]=(uint32_t)(std::lower_bound(p,p+epb,H)-p); }
    for (auto& v: hr) v=rng()%rpb;
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice)); CK(hipMemcpy(hot_end,he.data(),nblk*4,hipMemcpyHostToDevice));
    const size_t lds=(size_t)(rpb+H)*8;
    CK(hipFuncSetAttribute((const void*)k_blocked<4,true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
    CK(hipFuncSetAttribute((const void*)k_blocked<4,false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150*1024));
    float a=timeit([&]{ hipLaunchKernelGGL((k_blocked<4,false>),dim3(nblk),dim3(1024),lds,0,col,w,lrow,x,y,epb,rpb,hot_end,H); });
    float b=timeit([&]{ hipLaunchKernelGGL((k_blocked<4,true>),dim3(nblk),dim3(1024),lds,0,col,w,lrow,x,y,epb,rpb,hot_end,H); });
    printf("hot fraction %.2f: all gathers from memory %.1f us, hot ones from an LDS copy %.1f us\n", hot_frac, a, b);
  }
  return 0;
}
