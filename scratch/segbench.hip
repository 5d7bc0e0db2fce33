// scratch: bandwidth of "64 rows x W columns" tile streaming (x-direction kernels' fetch pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
template<int W>  // W doubles per row segment per tile; lane->(row,colpair)
__global__ void __launch_bounds__(64) k_tiles(const double* __restrict__ a, double* __restrict__ b, int nxp, int ntile){
  const int lane=threadIdx.x; const long slab=(long)blockIdx.x*64*nxp;
  constexpr int LPR=W/2, RPF=64/LPR, TLD=W/2;
  const int cp=(lane%LPR)*2;
  for(int t=0;t<ntile;t++){
    double2 v[TLD];
    #pragma unroll
    for(int i=0;i<TLD;i++){ int row=lane/LPR+RPF*i; v[i]=*(const double2*)(a+slab+(long)row*nxp+t*W+cp); }
    #pragma unroll
    for(int i=0;i<TLD;i++){ int row=lane/LPR+RPF*i; v[i].x+=1.0; *(double2*)(b+slab+(long)row*nxp+t*W+cp)=v[i]; }
  }
}
int main(){
  const int nx=512,ny=512,nz=512; size_t n=(size_t)nx*ny*nz; double *a,*b; hipMalloc(&a,n*8); hipMalloc(&b,n*8); hipMemset(a,0,n*8);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run=[&](const char* nm, auto f){ for(int i=0;i<2;i++) f(); hipEventRecord(e0); for(int i=0;i<10;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=10; printf("%-28s %7.3f ms %8.1f GB/s\n",nm,ms,2.0*n*8/ms*1e-6); };
  int nw=ny*nz/64;
  run("tile W=16 (128B seg)",[&]{ hipLaunchKernelGGL(k_tiles<16>,dim3(nw),dim3(64),0,0,a,b,nx,nx/16); });
  run("tile W=32 (256B seg)",[&]{ hipLaunchKernelGGL(k_tiles<32>,dim3(nw),dim3(64),0,0,a,b,nx,nx/32); });
  run("tile W=64 (512B seg)",[&]{ hipLaunchKernelGGL(k_tiles<64>,dim3(nw),dim3(64),0,0,a,b,nx,nx/64); });
  run("tile W=128 (1KB seg)",[&]{ hipLaunchKernelGGL(k_tiles<128>,dim3(nw),dim3(64),0,0,a,b,nx,nx/128); });
  return 0; }
