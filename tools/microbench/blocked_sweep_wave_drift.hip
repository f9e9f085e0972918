// Microbenchmark 6: timestamps of the blocked sweep per workgroup and XCD.  The oldest wavefront of a
// workgroup finishes its sixteen batches at ~32 us while the workgroup ends at ~55 us: the wavefronts
// of one workgroup drift apart far more than the workgroups of an XCD do (~2.5 us).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
constexpr int NT = 18;  // timestamps per block: start, after each of 16 batches, end
__global__ __launch_bounds__(1024) void k_blocked(const uint32_t* __restrict__ col, const uint8_t* __restrict__ w, const uint16_t* __restrict__ lrow,
                                                  const double* __restrict__ x, double* __restrict__ y, uint32_t edges_per_block, uint32_t rows_per_block,
                                                  unsigned long long* ts, uint32_t* xcc) {
  extern __shared__ double acc[];
  constexpr int U = 4;
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) acc[i]=0.0;
  __syncthreads();
  if (threadIdx.x==0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x]=id; ts[blockIdx.x*NT]=wall_clock64(); }
  const size_t base=(size_t)blockIdx.x*edges_per_block;
  int k=1;
  for (uint32_t e=threadIdx.x; e + (U-1)*1024 < edges_per_block; e+=U*1024, ++k) {
    uint32_t c[U]; uint8_t ww[U]; uint16_t r[U]; double xv[U];
#pragma unroll
    for(int u=0;u<U;++u) { c[u]=col[base+e+u*1024]; ww[u]=w[base+e+u*1024]; r[u]=lrow[base+e+u*1024]; }
#pragma unroll
    for(int u=0;u<U;++u) xv[u]=x[c[u]];
#pragma unroll
    for(int u=0;u<U;++u) atomicAdd(&acc[r[u]], (double)ww[u]*xv[u]);
    if (threadIdx.x==0 && k<NT-1) ts[blockIdx.x*NT+k]=wall_clock64();
  }
  __syncthreads();
  for (uint32_t i=threadIdx.x;i<rows_per_block;i+=1024) y[(size_t)blockIdx.x*rows_per_block+i]=acc[i];
  if (threadIdx.x==0) ts[blockIdx.x*NT+NT-1]=wall_clock64();
}
int main(){
  const size_t n=16777216; const uint32_t T=1u<<20; const int nblk=256;
  uint32_t* col; uint8_t* w; uint16_t* lrow; double* x; double* y; unsigned long long* ts; uint32_t* xcc;
  CK(hipMalloc(&col,n*4)); CK(hipMalloc(&w,n)); CK(hipMalloc(&lrow,n*2)); CK(hipMalloc(&x,(size_t)T*8)); CK(hipMalloc(&y,(size_t)T*8));
  CK(hipMalloc(&ts,nblk*NT*8)); CK(hipMalloc(&xcc,nblk*4));
  CK(hipMemset(w,1,n)); CK(hipMemset(x,0,(size_t)T*8));
  std::mt19937_64 rng(1);
  std::vector<uint32_t> h(n); std::vector<uint16_t> hr(n);
  uint32_t epb=n/nblk, rpb=T/nblk;
  for (int b=0;b<nblk;++b) { uint32_t* p=h.data()+(size_t)b*epb; for (uint32_t e=0;e<epb;++e) p[e]=rng()%T; std::sort(p,p+epb); }
  for (auto& v: hr) v=rng()%rpb;
  CK(hipMemcpy(col,h.data(),n*4,hipMemcpyHostToDevice)); CK(hipMemcpy(lrow,hr.data(),n*2,hipMemcpyHostToDevice));
  for (int rep=0;rep<5;++rep) hipLaunchKernelGGL(k_blocked,dim3(nblk),dim3(1024),rpb*8,0,col,w,lrow,x,y,epb,rpb,ts,xcc);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> t(nblk*NT); std::vector<uint32_t> xc(nblk);
  CK(hipMemcpy(t.data(),ts,nblk*NT*8,hipMemcpyDeviceToHost)); CK(hipMemcpy(xc.data(),xcc,nblk*4,hipMemcpyDeviceToHost));
  unsigned long long t0=~0ull; for (int b=0;b<nblk;++b) t0=std::min(t0,t[b*NT]);
  // wall_clock64 ticks at 100 MHz: 10 ns per tick
  printf("per XCD: blocks, start spread, time at batch 4/8/12/16 (min..max over the XCD's blocks), end (min..max), all in us\n");
  for (int xid=0;xid<8;++xid) {
    int cnt=0; double smin=1e9,smax=0,emin=1e9,emax=0; double qmin[4]={1e9,1e9,1e9,1e9},qmax[4]={0,0,0,0};
    for (int b=0;b<nblk;++b) if ((xc[b]&0xF)==(uint32_t)xid) { ++cnt;
      double s=(t[b*NT]-t0)*0.01, e=(t[b*NT+NT-1]-t0)*0.01; smin=std::min(smin,s); smax=std::max(smax,s); emin=std::min(emin,e); emax=std::max(emax,e);
      for (int q=0;q<4;++q) { double v=(t[b*NT+4*(q+1)]-t0)*0.01; qmin[q]=std::min(qmin[q],v); qmax[q]=std::max(qmax[q],v); } }
    printf("xcd %d: %3d blocks start %.1f..%.1f | b4 %.1f..%.1f b8 %.1f..%.1f b12 %.1f..%.1f b16 %.1f..%.1f | end %.1f..%.1f\n",xid,cnt,smin,smax,qmin[0],qmax[0],qmin[1],qmax[1],qmin[2],qmax[2],qmin[3],qmax[3],emin,emax);
  }
  printf("xcc raw ids of blocks 0..15:"); for (int b=0;b<16;++b) printf(" %u",xc[b]); printf("\n");
  return 0;
}
