// Development microbenchmark: what streaming pattern reaches the best HBM rate on this MI355X?
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/dev/membench scripts/dev/membench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// grid-stride over 16 B vectors, UNR independent loads per thread per step, stride = grid*block between them
template<int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_gs(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n){
  size_t stride=(size_t)gridDim.x*blockDim.x;
  size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;
  for(; i+ (UNR-1)*stride < n; i+=UNR*stride){
    u32x4 v[UNR];
#pragma unroll
    for(int j=0;j<UNR;j++) v[j]= NT? __builtin_nontemporal_load(&s[i+j*stride]) : s[i+j*stride];
#pragma unroll
    for(int j=0;j<UNR;j++){ if(NT) __builtin_nontemporal_store(v[j],&d[i+j*stride]); else d[i+j*stride]=v[j]; }
  }
  for(; i<n; i+=stride) d[i]=s[i];
}
// unit-based: block handles units of UNR*4KiB contiguous (like the ipx lut kernel), units grid-strided
template<int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_unit(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n){
  size_t nunits = n/(256*UNR);
  for(size_t u=blockIdx.x; u<nunits; u+=gridDim.x){
    size_t base=u*256*UNR+threadIdx.x;
    u32x4 v[UNR];
#pragma unroll
    for(int j=0;j<UNR;j++) v[j]= NT? __builtin_nontemporal_load(&s[base+j*256]) : s[base+j*256];
#pragma unroll
    for(int j=0;j<UNR;j++){ if(NT) __builtin_nontemporal_store(v[j],&d[base+j*256]); else d[base+j*256]=v[j]; }
  }
}
// one unit per block, huge grid
template<int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n){
  size_t base=(size_t)blockIdx.x*256*UNR+threadIdx.x;
  u32x4 v[UNR];
#pragma unroll
  for(int j=0;j<UNR;j++) if(base+j*256<n) v[j]= NT? __builtin_nontemporal_load(&s[base+j*256]) : s[base+j*256];
#pragma unroll
  for(int j=0;j<UNR;j++) if(base+j*256<n){ if(NT) __builtin_nontemporal_store(v[j],&d[base+j*256]); else d[base+j*256]=v[j]; }
}
template<int UNR>
__global__ __launch_bounds__(256) void read_only(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n){
  size_t stride=(size_t)gridDim.x*blockDim.x; size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; u32x4 acc={0,0,0,0};
  for(; i+(UNR-1)*stride<n; i+=UNR*stride){
#pragma unroll
    for(int j=0;j<UNR;j++){ u32x4 v=s[i+j*stride]; acc^=v; }
  }
  if(acc[0]==0x12345678 && acc[1]==1) d[0]=acc;
}
template<int UNR>
__global__ __launch_bounds__(256) void write_only(u32x4* __restrict__ d, size_t n){
  size_t stride=(size_t)gridDim.x*blockDim.x; size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x; u32x4 v={1,2,3,(uint32_t)i};
  for(; i<n; i+=stride) d[i]=v;
}
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t sizes[]={133ull<<20, 400ull<<20, 1024ull<<20};
  for(size_t bytes: sizes){
    u32x4 *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
    size_t n=bytes/16;
    auto run=[&](const char* name, auto launch, double traffic){
      for(int i=0;i<10;i++) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0); for(int i=0;i<20;i++) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms,e0,e1); double sec=ms/20*1e-3;
      printf("%5zu MiB %-28s %8.1f us %8.1f GB/s\n", bytes>>20, name, sec*1e6, traffic/sec/1e9); fflush(stdout);
    };
    for(int g: {1024,2048,4096,8192,16384}){
      char nm[64];
      snprintf(nm,64,"gs4 grid=%d",g);  run(nm,[&]{ hipLaunchKernelGGL((copy_gs<4,false>),dim3(g),dim3(256),0,0,s,d,n); },2.0*bytes);
      snprintf(nm,64,"gs8 grid=%d",g);  run(nm,[&]{ hipLaunchKernelGGL((copy_gs<8,false>),dim3(g),dim3(256),0,0,s,d,n); },2.0*bytes);
      snprintf(nm,64,"gs4nt grid=%d",g);  run(nm,[&]{ hipLaunchKernelGGL((copy_gs<4,true>),dim3(g),dim3(256),0,0,s,d,n); },2.0*bytes);
      snprintf(nm,64,"unit4 grid=%d",g);  run(nm,[&]{ hipLaunchKernelGGL((copy_unit<4,false>),dim3(g),dim3(256),0,0,s,d,n); },2.0*bytes);
      snprintf(nm,64,"unit4nt grid=%d",g);  run(nm,[&]{ hipLaunchKernelGGL((copy_unit<4,true>),dim3(g),dim3(256),0,0,s,d,n); },2.0*bytes);
    }
    run("flat1",[&]{ hipLaunchKernelGGL((copy_flat<1,false>),dim3((unsigned)((n+255)/256)),dim3(256),0,0,s,d,n); },2.0*bytes);
    run("flat2",[&]{ hipLaunchKernelGGL((copy_flat<2,false>),dim3((unsigned)((n+511)/512)),dim3(256),0,0,s,d,n); },2.0*bytes);
    run("flat4",[&]{ hipLaunchKernelGGL((copy_flat<4,false>),dim3((unsigned)((n+1023)/1024)),dim3(256),0,0,s,d,n); },2.0*bytes);
    run("flat4nt",[&]{ hipLaunchKernelGGL((copy_flat<4,true>),dim3((unsigned)((n+1023)/1024)),dim3(256),0,0,s,d,n); },2.0*bytes);
    run("flat8",[&]{ hipLaunchKernelGGL((copy_flat<8,false>),dim3((unsigned)((n+2047)/2048)),dim3(256),0,0,s,d,n); },2.0*bytes);
    run("read_only gs8 4096",[&]{ hipLaunchKernelGGL((read_only<8>),dim3(4096),dim3(256),0,0,s,d,n); },1.0*bytes);
    run("read_only gs8 16384",[&]{ hipLaunchKernelGGL((read_only<8>),dim3(16384),dim3(256),0,0,s,d,n); },1.0*bytes);
    run("write_only 4096",[&]{ hipLaunchKernelGGL((write_only<1>),dim3(4096),dim3(256),0,0,d,n); },1.0*bytes);
    run("hipMemcpyDtoD",[&]{ hipMemcpyAsync(d,s,bytes,hipMemcpyDeviceToDevice,0); },2.0*bytes);
    CK(hipFree(s)); CK(hipFree(d));
  }
  return 0;
}
