// Dev microbenchmark 4: how fast can 2-D source tiles with halos be streamed in (loads only)?  8K BGR x 2 geometry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
// block (256 thr) per tile step: thread (lrow = tid>>4, lch = tid&15) loads chunk lch (< nch) of rows lrow+16i (< rows)
// of the tile at (tx*128 + xoff, ty*64 - 15).  MODE 0: tiles linear + stride G; MODE 1: 12x8 patches per XCD.
template<int NLD>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ s, uint32_t* __restrict__ out, int L, int H, int tiles_x, int tiles_y, int nimg,
                                         int rows, int nch, int xoff, int mode){
  const int tid=threadIdx.x, lrow=tid>>4, lch=min(tid&15,nch-1);
  u32x4 acc={0,0,0,0};
  unsigned ntiles=tiles_x*tiles_y*nimg;
  auto body=[&](int tx,int ty,int img){
    int xs=tx*128+xoff, ys=ty*64-15;
    xs=max(0,min(xs,L-16*nch)); ys=max(0,min(ys,H-rows));
    const uint8_t* base=s+(size_t)img*L*H+(size_t)ys*L+xs+16*lch;
#pragma unroll
    for(int i=0;i<NLD;i++){ int r=min(lrow+16*i,rows-1); acc^=*(const u32x4*)(base+(size_t)r*L); }
  };
  if(mode==0){
    for(unsigned t=blockIdx.x;t<ntiles;t+=gridDim.x) body(t%tiles_x,(t/tiles_x)%tiles_y,t/(tiles_x*tiles_y));
  } else {
    int q=blockIdx.x>>3, qx=q%12, qy=q/12; int px_n=(tiles_x+11)/12, py_n=(tiles_y+7)/8; unsigned np=px_n*py_n*nimg;
    for(unsigned P=blockIdx.x&7;P<np;P+=8){ int tx=(P%px_n)*12+qx, ty=((P/px_n)%py_n)*8+qy; if(tx<tiles_x&&ty<tiles_y) body(tx,ty,P/(px_n*py_n)); }
  }
  out[blockIdx.x*256+tid]=acc[0]^acc[1]^acc[2]^acc[3];
}
int main(){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int L=7680*3, H=4320, N=2; size_t bytes=(size_t)L*H*N;
  uint8_t *s; uint32_t* o; CK(hipMalloc(&s,bytes)); CK(hipMalloc(&o,4<<20)); CK(hipMemset(s,1,bytes));
  const int tiles_x=L/128, tiles_y=(H+63)/64;
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<5;i++) launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for(int i=0;i<10;i++) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms,e0,e1); double sec=ms/10*1e-3;
    printf("%-56s %8.1f us  (%.0f GB/s of image bytes)\n", name, sec*1e6, bytes/sec/1e9); fflush(stdout);
  };
  char nm[128];
  struct Cfg{int rows,nch,xoff,mode,grid;};
  Cfg cfgs[]={{94,14,-48,0,768},{94,14,-48,1,768},{94,16,-64,1,768},{94,16,0,1,768},{64,8,0,1,768},{64,8,0,0,768},{94,8,0,1,768},{64,14,-48,1,768},
              {94,14,-48,1,1536},{94,14,-48,1,2304},{94,16,-64,1,2304},{64,8,0,1,2304}};
  for(auto c: cfgs){
    snprintf(nm,128,"rows=%d nch=%d xoff=%d mode=%d grid=%d",c.rows,c.nch,c.xoff,c.mode,c.grid);
    run(nm,[&]{ hipLaunchKernelGGL((k<6>),dim3(c.grid),dim3(256),0,0,s,o,L,H,tiles_x,tiles_y,N,c.rows,c.nch,c.xoff,c.mode); });
  }
  return 0;
}
