// Microbenchmark 14: what the f64 blocked sweep would gain from denser sources.  256 workgroups of 65536 edges
// each; workgroup b takes its sources from window b % S of S equal windows of the 8 MB vector (S = 1: the whole
// vector, as the engine's sweep does) and keeps S times as many row accumulators in LDS (a row block S times as
// tall, its edge list cut into S source windows, one workgroup per window).  The partial sums of the S
// workgroups of a row block are written out and added by a second kernel, timed with it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
__global__ __launch_bounds__(1024) void k_sweep(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                const double* __restrict__ x, double* __restrict__ part, uint32_t edges_per_block, uint32_t rows_per_block) {
  extern __shared__ double acc[];
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) acc[i]=0.0;
  __syncthreads();
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  int k=0;
  for (uint32_t e=threadIdx.x; e + 3*1024 < edges_per_block; e+=4*1024, ++k) {
    uint32_t c[4]; uint8_t ww[4]; uint16_t r[4]; double xv[4];
#pragma unroll
    for(int u=0;u<4;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
    for(int u=0;u<4;++u) xv[u]=x[c[u]];
#pragma unroll
    for(int u=0;u<4;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
    if (k & 1) __syncthreads();
  }
  __syncthreads();
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) part[(size_t)blockIdx.x*rows_per_block+i]=acc[i];
}
// row block g = workgroups g*S .. g*S+S-1; y[row] = sum of their partials times a per-row factor
__global__ __launch_bounds__(256) void k_join(const double* __restrict__ part, const double* __restrict__ scale, double* __restrict__ y,
                                              uint32_t rows_per_block, uint32_t S, uint32_t rows) {
  const uint32_t i = blockIdx.x*256 + threadIdx.x;
  if (i >= rows) return;
  const uint32_t g = i / rows_per_block, r = i % rows_per_block;
  double s = 0.0;
  for (uint32_t q=0;q<S;++q) s += part[((size_t)g*S+q)*rows_per_block + r];
  y[i] = s * scale[i];
}
template<typename F> float timeit(F f, int reps=20){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for(int i=0;i<3;++i) f();
  CK(hipEventRecord(a)); for(int i=0;i<reps;++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps*1000.f;
}
int main(){
  const size_t n=16777216; const uint32_t T=1u<<20; const int nblk=256;
  uint32_t* col; uint8_t* w; uint16_t* lrow; double* x; double* part; double* y; double* scale;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*8)); CK(hipMalloc(&y,(size_t)T*8));
  CK(hipMalloc(&scale,(size_t)T*8)); CK(hipMalloc(&part,(size_t)T*8*8));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,0,(size_t)T*8)); CK(hipMemset(scale,0,(size_t)T*8));
  CK(hipFuncSetAttribute((const void*)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024-64));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  const uint32_t epb=n/nblk;
  for (int skew=0; skew<2; ++skew)
  for (uint32_t S : {1u, 2u, 4u}) {
    const uint32_t rpb = T/nblk*S;      // 4096, 8192, 16384 accumulators: 32, 64, 128 KB of LDS
    for (int b=0;b<nblk;++b) {
      uint32_t* p=h.data()+(size_t)b*epb;
      if (!skew) { const uint32_t win=T/S, lo=(b%S)*win; for (uint32_t e=0;e<epb;++e) p[e]=lo+rng()%win; }
      else {
        // skewed: 40 % of a row block's edges on the first 16384 sources, the rest uniform; the block's sorted list
        // of S*epb edges cut into S equal pieces, this workgroup takes piece b % S
        std::vector<uint32_t> all((size_t)S*epb);
        std::mt19937_64 r2(1000 + b/S);
        for (auto& v: all) v = (r2()%100 < 40) ? r2()%16384 : r2()%T;
        std::sort(all.begin(), all.end());
        std::copy(all.begin()+(size_t)(b%S)*epb, all.begin()+(size_t)(b%S+1)*epb, p);
      }
      std::sort(p,p+epb);
    }
    for (auto& v: hr) v=rng()%rpb;
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
    float t1 = timeit([&]{ hipLaunchKernelGGL(k_sweep,dim3(nblk),dim3(1024),rpb*8,0,col,w,lrow,x,part,epb,rpb); });
    float t2 = timeit([&]{ hipLaunchKernelGGL(k_sweep,dim3(nblk),dim3(1024),rpb*8,0,col,w,lrow,x,part,epb,rpb);
                           hipLaunchKernelGGL(k_join,dim3(T/256),dim3(256),0,0,part,scale,y,rpb,S,T); });
    printf("%s sources, %u windows, %u accumulators per workgroup: sweep %.1f us, sweep + join %.1f us\n", skew?"skewed":"uniform", S, rpb, t1, t2);
  }
  return 0;
}
