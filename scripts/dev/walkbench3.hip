// Dev microbenchmark 3: flat copy vs strip-walker copy when NOTHING fits the Infinity Cache (2 x 796 MB),
// i.e. the regime of BASELINE configs[2] on one GPU.  hipcc --offload-arch=gfx950 -O3 -o walkbench3 walkbench3.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template<bool NTL>
__global__ __launch_bounds__(256) void flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16, int per){
  size_t base = ((size_t)blockIdx.x*256 + threadIdx.x);       // each wave: `per` consecutive 1 KiB pieces
  size_t wave = base>>6, lane = base&63;
  for(int i=0;i<per;i++){ size_t k=(wave*per+i)*64+lane; if(k<n16){ u32x4 v= NTL? __builtin_nontemporal_load(s+k) : s[k]; __builtin_nontemporal_store(v,d+k);} }
}
template<int D, bool NTL>
__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R,
                                            int nstrips, int nsegs, int HALO){
  unsigned item = blockIdx.x*4 + (threadIdx.x>>6);
  if(item >= (unsigned)(nstrips*nsegs)) return;
  int strip = item % nstrips, seg = item / nstrips;
  int lane = threadIdx.x&63;
  int y0 = seg*R, y1 = min(y0+R, H);
  int b0 = strip*1024 + lane*16;
  if(b0+16 > L) return;
  const int hr = HALO/2;
  auto rowp=[&](int y){ y = y<0?0:(y>H-1?H-1:y); return s+(size_t)y*L+b0; };
  u32x4 buf[D];
  int t0 = y0-hr, t1 = y1+hr;
#pragma unroll
  for(int i=0;i<D;i++) buf[i]= NTL? __builtin_nontemporal_load((const u32x4*)rowp(min(t0+i,t1-1))) : *(const u32x4*)rowp(min(t0+i,t1-1));
  u32x4 acc = {0,0,0,0};
  for(int t=t0;t<t1;t+=D){
#pragma unroll
    for(int i=0;i<D;i++){
      if(t+i<t1){
        u32x4 v=buf[i]; acc ^= v;
        int y=t+i-hr;
        if(y>=y0 && y<y1) __builtin_nontemporal_store(HALO? (v^acc) : v,(u32x4*)(d+(size_t)y*L+b0));
      }
      buf[i]= NTL? __builtin_nontemporal_load((const u32x4*)rowp(min(t+D+i,t1-1))) : *(const u32x4*)rowp(min(t+D+i,t1-1));
    }
  }
}
template<int D, int WPB>
__global__ __launch_bounds__(64*WPB) void walk_lock(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R,
                                                 int nstrips, int nsegs){
  const int bps = (nstrips + WPB - 1) / WPB;                  // blocks per row segment
  int sub = blockIdx.x % bps, seg = blockIdx.x / bps;
  int strip = sub*WPB + (threadIdx.x>>6);
  int lane = threadIdx.x&63;
  int y0 = seg*R, y1 = min(y0+R, H);
  int b0 = strip*1024 + lane*16;
  const bool live = strip < nstrips && b0+16 <= L;            // dead lanes still take part in the barriers
  if(!live) b0 = 0;
  auto rowp=[&](int y){ y = y<0?0:(y>H-1?H-1:y); return s+(size_t)y*L+b0; };
  u32x4 buf[D];
#pragma unroll
  for(int i=0;i<D;i++) buf[i]= *(const u32x4*)rowp(min(y0+i,y1-1));
  for(int t=y0;t<y1;t+=D){
    __builtin_amdgcn_s_barrier();                             // the block's waves read their rows together
#pragma unroll
    for(int i=0;i<D;i++){
      u32x4 v=buf[i];
      if(live && t+i<y1) __builtin_nontemporal_store(v,(u32x4*)(d+(size_t)(t+i)*L+b0));
      buf[i]= *(const u32x4*)rowp(min(t+D+i,y1-1));
    }
  }
}
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L=11520, H=2160*32; size_t bytes=(size_t)L*H;
  uint8_t *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<5;i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for(int i=0;i<10;i++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); double sec=ms/10*1e-3;
    printf("%-60s %8.1f us %8.1f GB/s\n", name, sec*1e6, 2.0*bytes/sec/1e9); fflush(stdout);
  };
  char nm[128];
  size_t n16=bytes/16;
  for(int per: {1,4,16,64}){
    unsigned blocks=(unsigned)((n16/64+per-1)/per/4+1);
    snprintf(nm,128,"flat copy nt, %d KiB per wave",per); run(nm,[&]{ hipLaunchKernelGGL((flat<true>),dim3(blocks),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,n16,per); });
    snprintf(nm,128,"flat copy (default loads), %d KiB per wave",per); run(nm,[&]{ hipLaunchKernelGGL((flat<false>),dim3(blocks),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,n16,per); });
  }
  for(int halo: {0,4}) for(int R: {12,24,48,96}){
    int nstrips=(L+1023)/1024; int nsegs=(H+R-1)/R; unsigned items=nstrips*nsegs;
    snprintf(nm,128,"walker D=2 halo=%d R=%d (%u waves)",halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<2,false>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,halo); });
    snprintf(nm,128,"walker D=4 halo=%d R=%d (%u waves)",halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<4,false>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,halo); });
    snprintf(nm,128,"walker D=4 nt loads halo=%d R=%d",halo,R);
    run(nm,[&]{ hipLaunchKernelGGL((walk<4,true>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,halo); });
  }
  for(int R: {12,24,48}){
    int nstrips=(L+1023)/1024; int nsegs=(H+R-1)/R;
    { constexpr int W=4; int bps=(nstrips+W-1)/W; snprintf(nm,128,"lockstep walker D=4, 4 waves/block, R=%d",R);
      run(nm,[&]{ hipLaunchKernelGGL((walk_lock<4,W>),dim3(bps*nsegs),dim3(64*W),0,0,s,d,L,H,R,nstrips,nsegs); }); }
    { constexpr int W=12; int bps=(nstrips+W-1)/W; snprintf(nm,128,"lockstep walker D=4, 12 waves/block (whole rows), R=%d",R);
      run(nm,[&]{ hipLaunchKernelGGL((walk_lock<4,W>),dim3(bps*nsegs),dim3(64*W),0,0,s,d,L,H,R,nstrips,nsegs); }); }
    { constexpr int W=12; int bps=(nstrips+W-1)/W; snprintf(nm,128,"lockstep walker D=2, 12 waves/block (whole rows), R=%d",R);
      run(nm,[&]{ hipLaunchKernelGGL((walk_lock<2,W>),dim3(bps*nsegs),dim3(64*W),0,0,s,d,L,H,R,nstrips,nsegs); }); }
  }
  return 0;
}
