// Dev microbenchmark: does a BAND-structured copy (block loads a contiguous band of rows flat, all loads up front, into
// LDS; then writes its output rows) stream like a flat copy, where the strip walker (waves walking down rows) does not?
// Every timed launch sits between two events and between launches of another kernel (no overlap at the edges).
// hipcc --offload-arch=gfx950 -O3 -o bandbench bandbench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <dlfcn.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);}}while(0)

template<bool NTL, bool NTS>
__global__ __launch_bounds__(256) void flat2(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16){
  size_t k0 = (size_t)blockIdx.x*1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) v[j]= NTL? __builtin_nontemporal_load(s+k) : s[k]; }
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16){ if(NTS) __builtin_nontemporal_store(v[j],d+k); else d[k]=v[j]; } }
}
__global__ __launch_bounds__(256) void flat(const u32x4* __restrict__ s, u32x4* __restrict__ d, size_t n16){
  size_t k0 = (size_t)blockIdx.x*1024 + threadIdx.x;
  u32x4 v[4];
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) v[j]=__builtin_nontemporal_load(s+k); }
#pragma unroll
  for(int j=0;j<4;j++){ size_t k=k0+256*j; if(k<n16) __builtin_nontemporal_store(v[j],d+k); }
}

// walker copy: wave = 1 KiB strip x R rows, D rows ahead, HR halo rows each side re-read
template<int D>
__global__ __launch_bounds__(256) void walk(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R, int nstrips, int nsegs, int HR){
  unsigned item = blockIdx.x*4 + (threadIdx.x>>6);
  if(item >= (unsigned)(nstrips*nsegs)) return;
  int strip = item % nstrips, seg = item / nstrips, lane = threadIdx.x&63;
  int y0 = seg*R, y1 = min(y0+R, H), b0 = strip*1024 + lane*16;
  if(b0+16 > L) return;
  auto rowp=[&](int y){ y = y<0?0:(y>H-1?H-1:y); return s+(size_t)y*L+b0; };
  u32x4 buf[D]; int t0=y0-HR, t1=y1+HR;
#pragma unroll
  for(int i=0;i<D;i++) buf[i]= *(const u32x4*)rowp(min(t0+i,t1-1));
  u32x4 acc={0,0,0,0};
  for(int t=t0;t<t1;t+=D){
#pragma unroll
    for(int i=0;i<D;i++){
      if(t+i<t1){ u32x4 v=buf[i]; acc^=v; int y=t+i-HR; if(y>=y0&&y<y1) __builtin_nontemporal_store(v^acc,(u32x4*)(d+(size_t)y*L+b0)); }
      buf[i]= *(const u32x4*)rowp(min(t+D+i,t1-1));
    }
  }
}

// walker with knobs: XR = xcd remap of the block id, HL = also load 4 halo bytes left and right per row, VW = dummy VALU work per row
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks){
  const unsigned q = nblocks >> 3, r = nblocks & 7u, x = bid & 7u;
  const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
template<int D, bool XR, bool HL, int VW>
__global__ __launch_bounds__(256) void walk2(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R, int nstrips, int nsegs, int HR){
  extern __shared__ uint8_t dummy_lds[];
  unsigned blk = XR ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  unsigned item = blk*4 + (threadIdx.x>>6);
  if(item >= (unsigned)(nstrips*nsegs)) return;
  int strip = item % nstrips, seg = item / nstrips, lane = threadIdx.x&63;
  int y0 = seg*R, y1 = min(y0+R, H), b0 = strip*1024 + lane*16;
  if(b0+16 > L) return;
  auto rowp=[&](int y){ y = y<0?0:(y>H-1?H-1:y); return s+(size_t)y*L+b0; };
  u32x4 buf[D]; uint32_t hl[D], hr[D]; int t0=y0-HR, t1=y1+HR;
  const int lo = b0>=4? -4:0, hi = b0+20<=L? 16:12;
#pragma unroll
  for(int i=0;i<D;i++){ const uint8_t* p=rowp(min(t0+i,t1-1)); buf[i]= *(const u32x4*)p; if(HL){ hl[i]=*(const uint32_t*)(p+lo); hr[i]=*(const uint32_t*)(p+hi);} }
  u32x4 acc={0,0,0,0};
  for(int t=t0;t<t1;t+=D){
#pragma unroll
    for(int i=0;i<D;i++){
      if(t+i<t1){ u32x4 v=buf[i]; if(HL){ v[0]^=hl[i]; v[3]^=hr[i]; }
        if(VW){
#pragma unroll
          for(int q=0;q<VW;q++){ v[q&3] = v[q&3]*2654435761u + v[(q+1)&3]; }
        }
        acc^=v; int y=t+i-HR; if(y>=y0&&y<y1) __builtin_nontemporal_store(v^acc,(u32x4*)(d+(size_t)y*L+b0)); }
      const uint8_t* p=rowp(min(t+D+i,t1-1)); buf[i]= *(const u32x4*)p; if(HL){ hl[i]=*(const uint32_t*)(p+lo); hr[i]=*(const uint32_t*)(p+hi);}
    }
  }
}

// band copy: block = T threads, tile = R output rows x TW bytes at column x0 (TW = L: whole rows, contiguous band);
// loads (R + 2 HR) rows x TW bytes up front (NL 16-byte loads per lane) -> LDS; barrier; out = xor of 2 HR + 1 rows.
template<int T, int NL>
__global__ __launch_bounds__(T) void band(const uint8_t* __restrict__ s, uint8_t* __restrict__ d, int L, int H, int R, int TW, int ntx, int HR){
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
  const int x0 = tx*TW, y0 = ty*R;
  const int tw = min(TW, L-x0), c16 = tw/16;          // chunks per tile row
  const int rows_in = R + 2*HR, nch = rows_in*c16;
  u32x4 v[NL];
#pragma unroll
  for(int j=0;j<NL;j++){
    int k = threadIdx.x + T*j; if(k>=nch) k = nch-1;
    int row = k / c16, col = k - row*c16;
    int y = y0 - HR + row; y = y<0?0:(y>H-1?H-1:y);
    v[j] = *(const u32x4*)(s + (size_t)y*L + x0 + 16*col);
  }
#pragma unroll
  for(int j=0;j<NL;j++){ int k = threadIdx.x + T*j; if(k<nch) *(u32x4*)(lds + 16*(size_t)k) = v[j]; }
  __syncthreads();
  const int nout = R*c16;
  for(int k=threadIdx.x;k<nout;k+=T){
    int row = k / c16, col = k - row*c16;
    if(y0+row>=H) break;
    u32x4 acc = *(const u32x4*)(lds + 16*(size_t)((row)*c16+col));
    for(int q=1;q<=2*HR;q++) acc ^= *(const u32x4*)(lds + 16*(size_t)((row+q)*c16+col));
    __builtin_nontemporal_store(acc,(u32x4*)(d + (size_t)(y0+row)*L + x0 + 16*col));
  }
}

int main(int argc,char**argv){
  int L = argc>1?atoi(argv[1]):3840, H = argc>2?atoi(argv[2]):2160*24;
  size_t bytes=(size_t)L*H;
  const size_t ob = bytes * (argc>3?atoi(argv[3]):4);   // the kernel between timed launches moves ob bytes each way: flushes the Infinity Cache
  uint8_t *s,*d,*o1,*o2; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&d,bytes)); CK(hipMalloc(&o1,ob)); CK(hipMalloc(&o2,ob));
  CK(hipMemset(s,1,bytes)); CK(hipMemset(d,0,bytes)); CK(hipMemset(o1,2,ob)); CK(hipMemset(o2,0,ob));
  { std::vector<uint32_t> rnd(bytes/4); uint32_t x=12345; for(auto&v:rnd){ x=x*1664525u+1013904223u; v=x; } CK(hipMemcpy(s,rnd.data(),bytes,hipMemcpyHostToDevice)); }
  size_t n16=bytes/16, on16=ob/16;
  const int omode = getenv("OTHER_MODE") ? atoi(getenv("OTHER_MODE")) : 0;   // neighbour kernel: bit 0 = default-policy loads, bit 1 = default-policy stores
  auto other=[&](){ dim3 g((on16+1023)/1024), b(256);
    if(omode==0) hipLaunchKernelGGL((flat2<true,true>),g,b,0,0,(const u32x4*)o1,(u32x4*)o2,on16);
    else if(omode==1) hipLaunchKernelGGL((flat2<false,true>),g,b,0,0,(const u32x4*)o1,(u32x4*)o2,on16);
    else if(omode==2) hipLaunchKernelGGL((flat2<true,false>),g,b,0,0,(const u32x4*)o1,(u32x4*)o2,on16);
    else hipLaunchKernelGGL((flat2<false,false>),g,b,0,0,(const u32x4*)o1,(u32x4*)o2,on16); };
  auto timeit=[&](const char*name, auto fn){
    std::vector<float> ts; hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int i=0;i<12;i++){ other(); CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); other(); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(i>=2) ts.push_back(ms); }
    std::sort(ts.begin(),ts.end()); float ms=ts[ts.size()/2];
    printf("%-34s %8.1f us  %6.2f TB/s\n",name,ms*1e3,2.0*bytes/ms/1e9); fflush(stdout);
  };
  printf("L=%d H=%d (%.0f MB each way)\n",L,H,bytes/1e6);
  for(int i=0;i<30;i++) other();
  CK(hipDeviceSynchronize());
  if(const char* wu=getenv("WARM_SECONDS")){   // sustained load first: does the chip hold these rates once it is warm?
    double secs=atof(wu); int ns=(L+1023)/1024, nsg=(H+15)/16; unsigned gr=(ns*nsg+3)/4;
    hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float tot=0;
    while(tot<secs*1e3){ CK(hipEventRecord(a)); for(int i=0;i<20;i++){ hipLaunchKernelGGL((walk2<4,true,true,160>),dim3(gr),dim3(256),0,0,s,d,L,H,16,ns,nsg,1); other(); } CK(hipEventRecord(b)); CK(hipDeviceSynchronize()); float ms; CK(hipEventElapsedTime(&ms,a,b)); tot+=ms; }
    printf("warmed for %.1f s\n",tot/1e3);
  }
  timeit("flat", [&](){ hipLaunchKernelGGL(flat,dim3((n16+1023)/1024),dim3(256),0,0,(const u32x4*)s,(u32x4*)d,n16); });
  if(const char* lp=getenv("IPX_LIB")){        // the library's own 3x3 kernels in THIS harness (gray: n = H / 2160 images)
    void* h=dlopen(lp,RTLD_NOW); if(!h){ printf("dlopen: %s\n",dlerror()); return 1; }
    typedef int (*conv_t)(void*,const uint8_t*,uint8_t*,int,int,int,int,int64_t,int64_t,int64_t,int64_t,const uint8_t*,const uint8_t*);
    conv_t box=(conv_t)dlsym(h,"ipx_box3_u8"), sob=(conv_t)dlsym(h,"ipx_sobel_u8");
    int c = L==3840?1:3, n=H/2160, w=L/c;
    timeit("libipx box3", [&](){ if(box(nullptr,s,d,n,2160,w,c,L,(int64_t)L*2160,L,(int64_t)L*2160,nullptr,nullptr)) printf("box3 failed\n"); });
    timeit("libipx sobel", [&](){ if(sob(nullptr,s,d,n,2160,w,c,L,(int64_t)L*2160,L,(int64_t)L*2160,nullptr,nullptr)) printf("sobel failed\n"); });
    timeit("libipx box3 as ONE tall image", [&](){ if(box(nullptr,s,d,1,H,w,c,L,(int64_t)L*H,L,(int64_t)L*H,nullptr,nullptr)) printf("box3 failed\n"); });
  }
  int nstrips=(L+1023)/1024;
  for(int R: {16,24}) for(int HR: {1,2}){
    int nsegs=(H+R-1)/R; char nm[64]; snprintf(nm,64,"walk D=4 R=%d halo=%d",R,HR);
    timeit(nm,[&](){ hipLaunchKernelGGL(walk<4>,dim3((nstrips*nsegs+3)/4),dim3(256),0,0,s,d,L,H,R,nstrips,nsegs,HR); });
  }
  {
    int R=16, HR=1, nsegs=(H+R-1)/R; unsigned grid=(nstrips*nsegs+3)/4;
#define W2(NAME, XR, HL, VW, LDS) timeit(NAME,[&](){ hipLaunchKernelGGL((walk2<4,XR,HL,VW>),dim3(grid),dim3(256),LDS,0,s,d,L,H,R,nstrips,nsegs,HR); })
    W2("walk2 base", false,false,0,0);
    W2("walk2 xcdremap", true,false,0,0);
    W2("walk2 halo-loads", false,true,0,0);
    W2("walk2 xcd+halo", true,true,0,0);
    W2("walk2 valu64", false,false,64,0);
    W2("walk2 valu160", false,false,160,0);
    W2("walk2 valu320", false,false,320,0);
    W2("walk2 occ5 (lds 32K)", false,false,0,32*1024);
    W2("walk2 occ4 (lds 40K)", false,false,0,40*1024);
    W2("walk2 occ3 (lds 53K)", false,false,0,53*1024);
    W2("walk2 occ5+halo+valu160+xcd", true,true,160,32*1024);
    W2("walk2 occ5+halo+valu160", false,true,160,32*1024);
  }
  auto run_band=[&](auto kern,int T,int NL,int R,int TW,int HR){
    int ntx=(L+TW-1)/TW, nty=(H+R-1)/R; size_t lds=(size_t)(R+2*HR)*TW;
    if((size_t)T*NL*16 < lds){ printf("band T=%d NL=%d R=%d TW=%d: tile does not fit the loads\n",T,NL,R,TW); return; }
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    char nm[80]; snprintf(nm,80,"band T=%d R=%d TW=%d halo=%d lds=%zuK",T,R,TW,HR,lds/1024);
    timeit(nm,[&](){ hipLaunchKernelGGL(kern,dim3(ntx*nty),dim3(T),lds,0,s,d,L,H,R,TW,ntx,HR); });
  };
  if(L==3840){
    run_band(band<256,10>,256,10,8,3840,1);  run_band(band<256,8>,256,8,6,3840,1);  run_band(band<256,6>,256,6,4,3840,1);
    run_band(band<512,8>,512,8,14,3840,1);   run_band(band<512,5>,512,5,8,3840,1);  run_band(band<1024,4>,1024,4,14,3840,1);
    run_band(band<256,12>,256,12,8,3840,2);  run_band(band<512,6>,512,6,8,3840,2);  run_band(band<256,6>,256,6,4,1920,1);
    run_band(band<256,10>,256,10,8,1024,1);
  } else {
    run_band(band<1024,6>,1024,6,4,11520,2); run_band(band<1024,5>,1024,5,4,11520,1); run_band(band<512,9>,512,9,2,11520,2);
    run_band(band<256,12>,256,12,8,3840,2);  run_band(band<512,6>,512,6,8,3840,2);   run_band(band<256,10>,256,10,8,3840,1);
    run_band(band<512,8>,512,8,12,3840,2);   run_band(band<256,10>,256,10,12,2048,2); run_band(band<256,12>,256,12,8,3840,2);
    run_band(band<512,12>,512,12,12,5760,2); run_band(band<256,9>,256,9,8,2880,2);
  }
  return 0;
}
