// Microbenchmark: what do coalesced streams and random 8-byte gathers cost on this GPU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template<int U>
__global__ __launch_bounds__(256) void k_stream(const uint32_t* __restrict__ col, const double* __restrict__ w, double* out, size_t n) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x, stride=(size_t)gridDim.x*256;
  double acc=0;
  for (size_t i=tid; i+ (U-1)*stride < n; i+=U*stride) {
    uint32_t c[U]; double ww[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) ww[u]=w[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) acc+=ww[u]*(double)c[u];
  }
  if (acc==12345.678) out[tid]=acc;
}
template<int U, bool WITHW>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ col, const double* __restrict__ w, const double* __restrict__ x, double* out, size_t n) {
  size_t tid = (size_t)blockIdx.x*256+threadIdx.x, stride=(size_t)gridDim.x*256;
  double acc=0;
  for (size_t i=tid; i+ (U-1)*stride < n; i+=U*stride) {
    uint32_t c[U]; double ww[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
    if (WITHW) {
#pragma unroll
    for(int u=0;u<U;++u) ww[u]=w[i+u*stride];
    }
#pragma unroll
    for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
    for(int u=0;u<U;++u) acc+= (WITHW? ww[u]:1.0)*xv[u];
  }
  if (acc==12345.678) out[tid]=acc;
}
// gather from an LDS tile of H doubles when idx < H else from global
template<int U, int H>
__global__ __launch_bounds__(1024) void k_gather_lds(const uint32_t* __restrict__ col, const double* __restrict__ w, const double* __restrict__ x, double* out, size_t n) {
  extern __shared__ double tile[];
  for (int i=threadIdx.x;i<H;i+=1024) tile[i]=x[i];
  __syncthreads();
  size_t tid = (size_t)blockIdx.x*1024+threadIdx.x, stride=(size_t)gridDim.x*1024;
  double acc=0;
  for (size_t i=tid; i+ (U-1)*stride < n; i+=U*stride) {
    uint32_t c[U]; double ww[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) c[u]=col[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) ww[u]=w[i+u*stride];
#pragma unroll
    for(int u=0;u<U;++u) xv[u]= c[u] < (uint32_t)H ? tile[c[u]] : x[c[u]];
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
int main(int argc,char**argv){
  size_t n = 16777216; 
  std::vector<uint32_t> h(n);
  uint32_t* col; double* w; double* x; double* out;
  size_t TMAX = 1u<<24; // 16M doubles = 128MB
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n*8)); CK(hipMalloc(&x,TMAX*8)); CK(hipMalloc(&out,(size_t)4096*1024*8));
  CK(hipMemset(w,0,n*8)); CK(hipMemset(x,0,TMAX*8));
  std::mt19937_64 rng(1);
  {
    float us = timeit([&]{ hipLaunchKernelGGL((k_stream<4>),dim3(2048),dim3(256),0,0,col,w,out,n); });
    printf("stream col+w U4 2048blk: %.1f us  %.2f TB/s\n",us, n*12.0/us*1e-6);
    us = timeit([&]{ hipLaunchKernelGGL((k_stream<8>),dim3(2048),dim3(256),0,0,col,w,out,n); });
    printf("stream col+w U8 2048blk: %.1f us  %.2f TB/s\n",us, n*12.0/us*1e-6);
    us = timeit([&]{ hipLaunchKernelGGL((k_stream<8>),dim3(4096),dim3(256),0,0,col,w,out,n); });
    printf("stream col+w U8 4096blk: %.1f us  %.2f TB/s\n",us, n*12.0/us*1e-6);
  }
  for (int mode=0; mode<2; ++mode) {   // 0 uniform, 1 skewed like the benchmark graph
   for (size_t T : {size_t(1)<<14, size_t(1)<<17, size_t(1)<<19, size_t(1)<<20, size_t(1)<<22, size_t(1)<<24}) {
    if (mode==0) { for (auto& v: h) v = rng()%T; }
    else { // power-law-ish: idx = T * u^3
      std::uniform_real_distribution<double> U01(0,1);
      for (auto& v: h) { double u=U01(rng); v=(uint32_t)(T*u*u*u); if(v>=T) v=T-1; }
    }
    CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice));
    float a = timeit([&]{ hipLaunchKernelGGL((k_gather<4,false>),dim3(2048),dim3(256),0,0,col,w,x,out,n); });
    float b = timeit([&]{ hipLaunchKernelGGL((k_gather<4,true>),dim3(2048),dim3(256),0,0,col,w,x,out,n); });
    float c = timeit([&]{ hipLaunchKernelGGL((k_gather<8,true>),dim3(2048),dim3(256),0,0,col,w,x,out,n); });
    float d = timeit([&]{ hipLaunchKernelGGL((k_gather<8,true>),dim3(4096),dim3(256),0,0,col,w,x,out,n); });
    float e = timeit([&]{ hipLaunchKernelGGL((k_gather_lds<4,16384>),dim3(256),dim3(1024),16384*8,0,col,w,x,out,n); });
    float f = timeit([&]{ hipLaunchKernelGGL((k_gather_lds<8,16384>),dim3(256),dim3(1024),16384*8,0,col,w,x,out,n); });
    printf("%s T=%8zu (%6.1f MB): gather-only U4 %.1f us | +w U4 %.1f | +w U8 %.1f | +w U8 4096blk %.1f | lds16K U4 %.1f | lds16K U8 %.1f us\n", mode?"skew":"unif", T, T*8/1048576.0, a,b,c,d,e,f);
   }
  }
  return 0;
}
