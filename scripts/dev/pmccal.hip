// Dev calibration of the TCC read counters (FETCH_SIZE, TCC_EA0_RDREQ*) on KNOWN byte counts, for the load paths the library uses:
//   flat     : 16-byte nt loads, block-contiguous (the table pass / BGR->gray path)
//   dma R HR : band tiles staged by global_load_lds_dwordx4 exactly as sep_pow2_band_kernel / conv3_band_kernel do (a block of 256
//              threads takes R output rows x 4 KiB of a row plus HR halo rows above and below, 1 KiB per wave instruction), then
//              stores its R rows.  HR = 0: every input byte is requested exactly once; HR = 2: (R + 4) / R times at the instruction
//              level (what reaches HBM depends on whether vertically adjacent tiles meet in an L2).
// Each kernel runs 3 times on a 597 MB buffer (11520 x 51840, the bench's 24 x 4K BGR batch) with a 1.2 GB flush between launches.
// hipcc --offload-arch=gfx950 -O3 -o scripts/dev/_build/pmccal scripts/dev/pmccal.hip ; run under rocprofv3 --pmc ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);}}while(0)

__global__ __launch_bounds__(256) void cal_flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16){
  size_t k0 = (size_t)blockIdx.x*1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) v[j]=__builtin_nontemporal_load(s+k); }
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) __builtin_nontemporal_store(v[j],d+k); }
}
__global__ __launch_bounds__(256) void cal_flush(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16){
  size_t k = (size_t)blockIdx.x*256 + threadIdx.x;
  if(k<n16) d[k] = s[k];
}

// tile = R rows x TWB bytes (TWB = 3840: three tiles per 11520-byte row); staged rows = R + 2 HR
template<int R, int HR>
__global__ __launch_bounds__(256) void cal_dma(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int ntx){
  constexpr int TWB = 3840, C16 = TWB/16, ROWS = R + 2*HR;
  __shared__ __attribute__((aligned(16))) uint8_t lds[ROWS*TWB];
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx, x0 = tx*TWB, y0 = ty*R;
  const int wave = threadIdx.x>>6, lane = threadIdx.x&63;
  for(int r = wave; r < ROWS; r += 4){
    int y = y0 - HR + r; y = y<0?0:(y>H-1?H-1:y);
    const uint8_t* rp = s + (size_t)y*L + x0;
    for(int c0=0;c0<C16;c0+=64){
      if(c0+lane<C16)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rp + 16*(c0+lane)),
                                         (void __attribute__((address_space(3)))*)(lds + r*TWB + 16*c0), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for(int k=threadIdx.x;k<R*C16;k+=256){
    int row=k/C16, col=k-row*C16;
    if(y0+row<H){
      u32x4 acc = *(const u32x4*)(lds + (row+HR)*TWB + 16*col);
      if(HR){ acc ^= *(const u32x4*)(lds + (row)*TWB + 16*col); acc ^= *(const u32x4*)(lds + (row+2*HR)*TWB + 16*col); }
      __builtin_nontemporal_store(acc,(u32x4*)(d + (size_t)(y0+row)*L + x0 + 16*col));
    }
  }
}

int main(){
  const int L=11520, H=2160*24; const size_t bytes=(size_t)L*H, fb = 1200u<<20;
  uint8_t *s,*d,*f0,*f1; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMalloc(&f0,fb)); CK(hipMalloc(&f1,fb));
  CK(hipMemset(s,7,bytes)); CK(hipMemset(f0,1,fb));
  auto flush=[&](){ hipLaunchKernelGGL(cal_flush,dim3((fb/16+255)/256),dim3(256),0,0,(const u32x4*)f0,(u32x4*)f1,fb/16); };
  const int ntx=L/3840;
  for(int rep=0;rep<3;rep++){
    flush(); hipLaunchKernelGGL(cal_flat,dim3((bytes/16+1023)/1024),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,bytes/16);
    flush(); hipLaunchKernelGGL((cal_dma<8,0>),dim3(ntx*((H+7)/8)),dim3(256),0,0,s,d,L,H,ntx);
    flush(); hipLaunchKernelGGL((cal_dma<8,2>),dim3(ntx*((H+7)/8)),dim3(256),0,0,s,d,L,H,ntx);
  }
  CK(hipDeviceSynchronize());
  printf("input bytes per launch: %zu (%.1f MB); cal_dma<8,2> requests %.1f MB at the instruction level\n", bytes, bytes/1e6, bytes*12.0/8/1e6);
  return 0;
}
