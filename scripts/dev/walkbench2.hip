// Dev microbenchmark 2: what do strip alignment and the vertical halo re-read cost the strip walker? (copy only)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

// wave = LANES lanes x 16 B starting at byte OFF + strip*LANES*16; walks R output rows, reading HALO extra rows
// (HALO/2 above, HALO/2 below, clamped) like a (HALO+1)-tap vertical filter would; D rows of loads in flight.
template<int D, bool NTL>
__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R,
                                            int nstrips, int nsegs, int OFF, int LANES, int HALO){
  unsigned item = blockIdx.x*4 + (threadIdx.x>>6);
  if(item >= (unsigned)(nstrips*nsegs)) return;
  int strip = item % nstrips, seg = item / nstrips;
  int lane = threadIdx.x&63;
  if(lane >= LANES) return;
  int y0 = seg*R, y1 = min(y0+R, H);
  int b0 = OFF + strip*LANES*16 + lane*16;
  if(b0+16 > L) return;
  const int hr = HALO/2;
  auto rowp=[&](int y){ y = y<0?0:(y>H-1?H-1:y); return s+(size_t)y*L+b0; };
  u32x4 buf[D];
  // source rows y0-hr .. y1-1+hr; output row y gets source row y+hr xor-folded with the row hr*2 earlier (keeps loads alive)
  int t0 = y0-hr, t1 = y1+hr;      // [t0,t1)
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
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L=11520, H=2160*16; size_t bytes=(size_t)L*H;
  uint8_t *s,*d; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes));
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<10;i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for(int i=0;i<20;i++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); double sec=ms/20*1e-3;
    printf("%-52s %8.1f us %8.1f GB/s(alg)\n", name, sec*1e6, 2.0*bytes/sec/1e9); fflush(stdout);
  };
  for(int k=0;k<200;k++) hipLaunchKernelGGL((walk<4,false>),dim3(4320),dim3(256),0,0,s,d,L,H,24,12,1440,0,64,0);
  char nm[128];
  struct Cfg{int off,lanes,halo;};
  Cfg cfgs[]={{0,64,0},{0,64,4}};
  for(auto c: cfgs) for(int R: {8,16,24,32,48}){
    int nstrips=(L-c.off+c.lanes*16-1)/(c.lanes*16); int nsegs=(H+R-1)/R; unsigned items=nstrips*nsegs;
    snprintf(nm,128,"walk D=4 off=%d lanes=%d halo=%d R=%d (%u waves)",c.off,c.lanes,c.halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<4,false>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,c.off,c.lanes,c.halo); });
  }
  for(auto c: {Cfg{16,60,4},Cfg{0,48,4}}) for(int R: {16,24,32}){
    int nstrips=(L-c.off+c.lanes*16-1)/(c.lanes*16); int nsegs=(H+R-1)/R; unsigned items=nstrips*nsegs;
    snprintf(nm,128,"walk D=6 off=%d lanes=%d halo=%d R=%d (%u waves)",c.off,c.lanes,c.halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<6,false>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,c.off,c.lanes,c.halo); });
  }
  for(auto c: {Cfg{0,64,0},Cfg{0,64,4},Cfg{0,64,2}}) for(int R: {8,16,24,32,48}){
    int nstrips=(L-c.off+c.lanes*16-1)/(c.lanes*16); int nsegs=(H+R-1)/R; unsigned items=nstrips*nsegs;
    snprintf(nm,128,"walk NTL D=4 off=%d lanes=%d halo=%d R=%d (%u waves)",c.off,c.lanes,c.halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<4,true>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,c.off,c.lanes,c.halo); });
    snprintf(nm,128,"walk NTL D=6 off=%d lanes=%d halo=%d R=%d (%u waves)",c.off,c.lanes,c.halo,R,items);
    run(nm,[&]{ hipLaunchKernelGGL((walk<6,true>),dim3((items+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,c.off,c.lanes,c.halo); });
  }
  return 0;
}
