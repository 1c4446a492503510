// Dev microbenchmark 3: does the 48-byte lane stride of the BGR->gray loads cost bandwidth?  3 B in : 1 B out streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
// block = 256 threads, unit = U * 4096 px.  MODE 0: lane loads its own 48 contiguous bytes (3 loads, 48-B lane stride).
// MODE 1: each load instruction covers 1 KiB contiguous per wave (lane stride 16 B), 3 instructions 1 KiB apart.
template<int MODE,int U,bool NT>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, size_t npx){
  size_t px0=(size_t)blockIdx.x*4096*U;
  const int wave=threadIdx.x>>6, lane=threadIdx.x&63;
  u32x4 v[U][3];
#pragma unroll
  for(int j=0;j<U;j++){
    size_t p=px0+(size_t)j*4096+wave*1024;           // wave's 1024 px = 3 KiB
    const uint8_t* b=s+p*3;
#pragma unroll
    for(int t=0;t<3;t++){
      const uint8_t* a = MODE==0 ? b+lane*48+16*t : b+t*1024+lane*16;
      v[j][t]= NT? __builtin_nontemporal_load((const u32x4*)a) : *(const u32x4*)a;
    }
  }
#pragma unroll
  for(int j=0;j<U;j++){
    size_t p=px0+(size_t)j*4096+wave*1024+lane*16;
    u32x4 o=v[j][0]^v[j][1]^v[j][2];
    if(NT) __builtin_nontemporal_store(o,(u32x4*)(d+p)); else *(u32x4*)(d+p)=o;
  }
}
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t npx=(size_t)16*2160*3840;
  uint8_t *s,*d; CK(hipMalloc(&s,npx*3)); CK(hipMalloc(&d,npx)); CK(hipMemset(s,1,npx*3)); CK(hipMemset(d,0,npx));
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<10;i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for(int i=0;i<20;i++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); double sec=ms/20*1e-3;
    printf("%-40s %8.1f us %8.1f GB/s\n", name, sec*1e6, 4.0*npx/sec/1e9); fflush(stdout);
  };
  for(int i=0;i<100;i++) hipLaunchKernelGGL((k<0,2,false>),dim3(npx/8192),dim3(256),0,0,s,d,npx);
#define R(M,U,NT) run("mode=" #M " U=" #U " nt=" #NT,[&]{ hipLaunchKernelGGL((k<M,U,NT>),dim3(npx/(4096*U)),dim3(256),0,0,s,d,npx); });
  R(0,1,false) R(1,1,false) R(0,2,false) R(1,2,false) R(0,4,false) R(1,4,false)
  R(0,1,true) R(1,1,true) R(0,2,true) R(1,2,true) R(0,4,true) R(1,4,true)
  return 0;
}
