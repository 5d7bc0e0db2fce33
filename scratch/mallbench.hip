// scratch: does the Infinity Cache absorb write->read reuse of a small scratch buffer?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_w(double2* __restrict__ b, size_t n2, double v){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n2;i+=st){ double2 t; t.x=v; t.y=v+1; b[i]=t; } }
__global__ void k_r(const double2* __restrict__ a, double* __restrict__ out, size_t n2){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; double s=0; for(;i<n2;i+=st){ double2 t=a[i]; s+=t.x+t.y; } if(s==123.456) out[0]=s; }
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x; for(;i<n2;i+=st) b[i]=a[i]; }
int main(){
  double *big,*sm,*out; size_t NB=(size_t)1<<27; hipMalloc(&big,NB*8*2); hipMalloc(&sm,(size_t)1<<28); hipMalloc(&out,8);
  hipMemset(big,0,NB*16);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(size_t mb : {16,32,64,128,192,256,512,1024}){
    size_t n2=mb*1024*1024/16; if(n2*16 > ((size_t)1<<28) && mb>256) { /* use big */ }
    double* buf = (mb<=256)? sm : big;
    // pattern: write buf, then read buf, repeated; in between stream 0 extra bytes
    for(int it=0;it<2;it++){ hipLaunchKernelGGL(k_w,dim3(2048),dim3(256),0,0,(double2*)buf,n2,1.0); hipLaunchKernelGGL(k_r,dim3(2048),dim3(256),0,0,(const double2*)buf,out,n2);}    
    hipEventRecord(e0); const int R=20; for(int it=0;it<R;it++){ hipLaunchKernelGGL(k_w,dim3(2048),dim3(256),0,0,(double2*)buf,n2,(double)it); hipLaunchKernelGGL(k_r,dim3(2048),dim3(256),0,0,(const double2*)buf,out,n2);} hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=R;
    printf("write+read %5zu MB buffer: %7.3f ms per pair -> %8.1f GB/s (w+r bytes)\n", mb, ms, 2.0*mb*1.048576/ms);
  }
  // write small scratch while streaming a big copy in between (pollution): w(sm 64MB), copy 256MB, r(sm)
  for(size_t mb : {32,64,96}){
    size_t n2=mb*1024*1024/16; size_t c2=(size_t)128*1024*1024/16;
    hipEventRecord(e0); const int R=20; for(int it=0;it<R;it++){ hipLaunchKernelGGL(k_w,dim3(2048),dim3(256),0,0,(double2*)sm,n2,(double)it); hipLaunchKernelGGL(k_copy,dim3(2048),dim3(256),0,0,(const double2*)big,(double2*)(big+NB),c2/2); hipLaunchKernelGGL(k_r,dim3(2048),dim3(256),0,0,(const double2*)sm,out,n2);} hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); ms/=R;
    printf("w(%zu MB) + copy(64MB->64MB) + r: %7.3f ms ; if all HBM at 5TB/s: %7.3f ms\n", mb, ms, (2.0*mb+128.0)*1.048576/5000.0);
  }
  return 0; }
