// Dev microbenchmark: does CHAINING vertically adjacent band tiles in one block (the second tile loads only its R new rows into
// the LDS ring, the 2 HR rows it shares with the first are already there) stream faster than one tile per block?  Copy-like
// kernels (out row = xor of its 2 HR + 1 source rows from LDS), rows of L bytes split into tiles of TW bytes, LDS-DMA staging as
// in the library's band kernels.  NC = tiles per block (1 = the library's structure).
// hipcc --offload-arch=gfx950 -O3 -o bandchain bandchain.hip ; ./bandchain [L] [H]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);}}while(0)
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks){
  const unsigned q = nblocks >> 3, r = nblocks & 7u, x = bid & 7u;
  const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
__global__ __launch_bounds__(256) void flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16){
  size_t k0 = (size_t)blockIdx.x*1024 + threadIdx.x; u32x4 v[4];
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) v[j]=__builtin_nontemporal_load(s+k); }
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) __builtin_nontemporal_store(v[j],d+k); }
}
// block = 256 lanes = one 4 KiB-wide tile column; ring of (R + 2 HR) LDS rows of 4096 bytes
template<int R, int HR>
__global__ __launch_bounds__(256) void chain(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int ntx, int nchains, int NC, unsigned nblocks){
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int RING = R + 2*HR;
  const unsigned blk = xcd_remap(blockIdx.x, nblocks);
  const int tx = blk % ntx, ch = blk / ntx;
  const int x0 = tx*4096, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (x0 + 16*(int)threadIdx.x + 16 > L) { /* lanes beyond the row keep taking part in the barriers */ }
  const bool act = x0 + 16*(int)threadIdx.x + 16 <= L;
  const int ybase = ch*NC*R;
  auto stage=[&](int first_row, int count, int slot0){          // image rows first_row .. +count-1 -> ring slots (slot0 + i) % RING
    for(int i=wave;i<count;i+=4){
      int y = first_row+i; y = y<0?0:(y>H-1?H-1:y);
      const uint8_t* rp = s + (size_t)y*L + x0;
      uint8_t* lp = lds + ((slot0+i)%RING)*4096;
      for(int c0=0;c0<256;c0+=64) if(x0+16*(c0+lane)+16<=L)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rp+16*(c0+lane)),(void __attribute__((address_space(3)))*)(lp+16*c0),16,0,0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  stage(ybase-HR, RING, 0);
  for(int t=0;t<NC;++t){
    const int y0 = ybase + t*R;
    if(y0 >= H) break;
    const int s0 = (t*R) % RING;                                  // ring slot of image row y0 - HR
    if(act){
      for(int r=0;r<R && y0+r<H;++r){
        u32x4 acc = *(const u32x4*)(lds + ((s0+r)%RING)*4096 + 16*threadIdx.x);
#pragma unroll
        for(int q=1;q<=2*HR;q++) acc ^= *(const u32x4*)(lds + ((s0+r+q)%RING)*4096 + 16*threadIdx.x);
        __builtin_nontemporal_store(acc,(u32x4*)(d + (size_t)(y0+r)*L + x0 + 16*threadIdx.x));
      }
    }
    if(t+1<NC && y0+R<H){
      __syncthreads();                                            // every lane is done with the R oldest rows
      stage(y0+R+HR, R, (s0+RING)%RING);                          // the next tile's R new rows replace them
    }
  }
}
int main(int argc,char**argv){
  int L = argc>1?atoi(argv[1]):11520, H = argc>2?atoi(argv[2]):2160*24;
  size_t bytes=(size_t)L*H;
  uint8_t *s,*d,*o1,*o2; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMalloc(&o1,bytes)); CK(hipMalloc(&o2,bytes));
  { std::vector<uint32_t> rnd(bytes/4); uint32_t x=12345; for(auto&v:rnd){ x=x*1664525u+1013904223u; v=x; } CK(hipMemcpy(s,rnd.data(),bytes,hipMemcpyHostToDevice)); }
  CK(hipMemset(o1,2,bytes)); CK(hipMemset(o2,0,bytes));
  size_t n16=bytes/16;
  auto other=[&](){ hipLaunchKernelGGL(flat,dim3((n16+1023)/1024),dim3(256),0,0,(const u32x4*)o1,(u32x4*)o2,n16); };
  auto timeit=[&](const char*name, auto fn){
    std::vector<float> ts; hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i=0;i<12;i++){ other(); CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); other(); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(i>=2) ts.push_back(ms); }
    std::sort(ts.begin(),ts.end()); float ms=ts[ts.size()/2];
    printf("%-40s %8.1f us  %6.2f TB/s\n",name,ms*1e3,2.0*bytes/ms/1e9); fflush(stdout);
  };
  printf("L=%d H=%d (%.0f MB each way)\n",L,H,bytes/1e6);
  for(int i=0;i<30;i++) other();
  timeit("flat", [&](){ hipLaunchKernelGGL(flat,dim3((n16+1023)/1024),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,n16); });
  const int ntx=(L+4095)/4096;
  CK(hipFuncSetAttribute((const void*)chain<8,2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
  for(int NC: {1,2,3,4,8}){
    int nchains=(H+8*NC-1)/(8*NC); unsigned nb=(unsigned)ntx*nchains; char nm[64]; snprintf(nm,64,"chain R=8 halo=2 tiles/block=%d",NC);
    timeit(nm,[&](){ hipLaunchKernelGGL((chain<8,2>),dim3(nb),dim3(256),12*4096,0,s,d,L,H,ntx,nchains,NC,nb); });
  }
  // tile heights at one tile per block (LDS: (R + 4) x 4 KiB per block)
#define ONE(RR) { CK(hipFuncSetAttribute((const void*)chain<RR,2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024)); \
    int nchains=(H+RR-1)/RR; unsigned nb=(unsigned)ntx*nchains; char nm[64]; snprintf(nm,64,"one tile per block, R=%d halo=2",RR); \
    timeit(nm,[&](){ hipLaunchKernelGGL((chain<RR,2>),dim3(nb),dim3(256),(RR+4)*4096,0,s,d,L,H,ntx,nchains,1,nb); }); }
  ONE(4) ONE(6) ONE(8) ONE(12) ONE(16) ONE(24)
  // checksum of the last variant against the first
  std::vector<uint8_t> a(1<<20), b(1<<20);
  { int nchains=(H+7)/8; unsigned nb=(unsigned)ntx*nchains; hipLaunchKernelGGL((chain<8,2>),dim3(nb),dim3(256),12*4096,0,s,d,L,H,ntx,nchains,1,nb); CK(hipDeviceSynchronize()); CK(hipMemcpy(a.data(),d+bytes/2,a.size(),hipMemcpyDeviceToHost)); }
  { int nchains=(H+31)/32; unsigned nb=(unsigned)ntx*nchains; hipLaunchKernelGGL((chain<8,2>),dim3(nb),dim3(256),12*4096,0,s,d,L,H,ntx,nchains,4,nb); CK(hipDeviceSynchronize()); CK(hipMemcpy(b.data(),d+bytes/2,b.size(),hipMemcpyDeviceToHost)); }
  printf("chained == unchained: %s\n", a==b?"yes":"NO");
  return 0;
}
