// Microbenchmark 5: microbenchmark 4 with the width of the per-edge value as a parameter (f64 weight,
// u32 or u8 raw opinion to be multiplied with a pre-scaled vector).
// One workgroup per CU owns a block of rows with f64 accumulators in LDS and walks
// its edges in ascending source order: coalesced 14-byte edge records, one 8-byte gather per edge,
// one LDS atomic add per edge.  All workgroups sweep the source range together, so the part of the
// gathered vector in use at any time is small (L2-resident) although the vector (8 MB) is not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int U, typename WT>
__global__ __launch_bounds__(1024) void k_blocked(const uint32_t* __restrict__ col, const WT* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                  const double* __restrict__ x, double* __restrict__ y, uint32_t edges_per_block, uint32_t rows_per_block) {
  extern __shared__ double acc[];
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) acc[i]=0.0;
  __syncthreads();
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  for (uint32_t e=threadIdx.x; e + (U-1)*1024 < edges_per_block; e+=U*1024) {
    uint32_t c[U]; WT ww[U]; uint16_t r[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
    for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
    for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
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
  uint32_t* col; double* w; uint16_t* lrow; double* x; double* y;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n*8)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*8)); CK(hipMalloc(&y,(size_t)T*8));
  CK(hipMemset(w,0,n*8)); CK(hipMemset(x,0,(size_t)T*8));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  for (int nblk : {256}) {
    uint32_t epb=n/nblk, rpb=T/nblk;
    for (int mode=0;mode<3;++mode) {  // 0 uniform sorted, 1 skewed sorted, 2 skewed unsorted
      std::uniform_real_distribution<double> U01(0,1);
      for (int b=0;b<nblk;++b) {
        uint32_t* p=h.data()+(size_t)b*epb;
        for (uint32_t e=0;e<epb;++e) { if(mode==0) p[e]=rng()%T; else { double u=U01(rng); uint32_t v=(uint32_t)(T*u*u*u); p[e]= v>=T?T-1:v; } }
        if (mode<2) std::sort(p,p+epb);
      }
      for (auto& v: hr) v=rng()%rpb;
      CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
      float a=timeit([&]{ hipLaunchKernelGGL((k_blocked<4,double>),dim3(nblk),dim3(1024),rpb*8,0,col,w,lrow,x,y,epb,rpb); });
      float b4=timeit([&]{ hipLaunchKernelGGL((k_blocked<4,uint32_t>),dim3(nblk),dim3(1024),rpb*8,0,col,(const uint32_t*)w,lrow,x,y,epb,rpb); });
      float c=timeit([&]{ hipLaunchKernelGGL((k_blocked<4,uint8_t>),dim3(nblk),dim3(1024),rpb*8,0,col,(const uint8_t*)w,lrow,x,y,epb,rpb); });
      float c8=timeit([&]{ hipLaunchKernelGGL((k_blocked<8,uint8_t>),dim3(nblk),dim3(1024),rpb*8,0,col,(const uint8_t*)w,lrow,x,y,epb,rpb); });
      printf("blocks %4d rows/block %5u %s: f64 %.1f us, u32 %.1f us, u8 %.1f us (U8: %.1f)\n", nblk, rpb, mode==0?"uniform sorted ":mode==1?"skewed sorted  ":"skewed unsorted", a, b4, c, c8);
    }
  }
  return 0;
}
