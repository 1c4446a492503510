// Dev microbenchmark: 2-D "strip walker" copy patterns vs flat copy on MI355X (what does the memory system want?)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// wave owns a strip of P KiB (P pieces of 64 lanes x 16 B) and walks R rows, D rows of loads in flight.
// items ordered strip-major within a row segment; 4 waves per block.
template<int P, int D, bool NTS>
__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R, int nstrips, int nsegs){
  unsigned item = blockIdx.x*4 + (threadIdx.x>>6);
  if(item >= (unsigned)(nstrips*nsegs)) return;
  int strip = item % nstrips, seg = item / nstrips;
  int y0 = seg*R, y1 = min(y0+R, H);
  int b0 = strip*P*1024 + (threadIdx.x&63)*16;
  u32x4 buf[D][P];
  auto ld=[&](int slot,int y){
#pragma unroll
    for(int p=0;p<P;p++){ int b=b0+p*1024; if(b+16<=L) buf[slot][p]=*(const u32x4*)(s+(size_t)y*L+b); }
  };
#pragma unroll
  for(int i=0;i<D;i++) ld(i, min(y0+i,y1-1));
  for(int y=y0;y<y1;y+=D){
#pragma unroll
    for(int i=0;i<D;i++){
      if(y+i<y1){
#pragma unroll
        for(int p=0;p<P;p++){ int b=b0+p*1024; if(b+16<=L){ u32x4 v=buf[i][p]; if(NTS) __builtin_nontemporal_store(v,(u32x4*)(d+(size_t)(y+i)*L+b)); else *(u32x4*)(d+(size_t)(y+i)*L+b)=v; } }
      }
      ld(i, min(y+D+i, y1-1));
    }
  }
}
template<int UNR, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n){
  size_t base=(size_t)blockIdx.x*256*UNR+threadIdx.x;
  u32x4 v[UNR];
#pragma unroll
  for(int j=0;j<UNR;j++) if(base+j*256<n) v[j]= NT? __builtin_nontemporal_load(&s[base+j*256]) : s[base+j*256];
#pragma unroll
  for(int j=0;j<UNR;j++) if(base+j*256<n){ if(NT) __builtin_nontemporal_store(v[j],&d[base+j*256]); else d[base+j*256]=v[j]; }
}
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L=11520, H=2160*16; size_t bytes=(size_t)L*H;
  uint8_t *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<10;i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for(int i=0;i<20;i++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); double sec=ms/20*1e-3;
    printf("%-40s %8.1f us %8.1f GB/s\n", name, sec*1e6, 2.0*bytes/sec/1e9); fflush(stdout);
  };
  for(int k=0;k<30;k++) hipLaunchKernelGGL((copy_flat<4,true>),dim3((unsigned)((bytes/16+1023)/1024)),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,bytes/16);
  run("flat4nt",[&]{ hipLaunchKernelGGL((copy_flat<4,true>),dim3((unsigned)((bytes/16+1023)/1024)),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,bytes/16); });
  run("flat4",[&]{ hipLaunchKernelGGL((copy_flat<4,false>),dim3((unsigned)((bytes/16+1023)/1024)),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,bytes/16); });
  char nm[96];
#define WALK(P,D,NTS) for(int R: {8,16,24,48,96}){ int nstrips=(L+P*1024-1)/(P*1024); int nsegs=(H+R-1)/R; unsigned items=nstrips*nsegs; \
    snprintf(nm,96,"walk P=%d D=%d nts=%d R=%d (%u waves)",P,D,NTS,R,items); \
    run(nm,[&]{ hipLaunchKernelGGL((walk<P,D,NTS>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs); }); }
  WALK(1,4,true) WALK(1,8,true) WALK(1,4,false) WALK(2,4,true) WALK(4,2,true) WALK(4,4,true)
  return 0;
}
