// Scratch microbenchmark: streaming bandwidth of the access patterns the
// Cartesian-layout derivative kernels would use (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// plain grid-stride copy, double2 per lane
__global__ void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2){
  size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t st = (size_t)gridDim.x*blockDim.x;
  for(; i<n2; i+=st) b[i]=a[i];
}
// walk: each thread owns one pencil (lane stride 1), walks n rows with stride `rs` elements.
// pencils indexed p = blockIdx.x*blockDim.x+threadIdx.x -> (lane offset = (p % w) + (p / w) * ps)
// chunks: blockIdx.y selects chunk of rows
__global__ void k_walk(const double* __restrict__ a, double* __restrict__ b,
                       int w, size_t ps, size_t rs, int rows_per_chunk, int unroll_dummy){
  size_t p = blockIdx.x*(size_t)blockDim.x + threadIdx.x;
  size_t base = (p % w) + (p / w) * ps + (size_t)blockIdx.y*rows_per_chunk*rs;
  double carry = 0.0;
  #pragma unroll 8
  for(int j=0;j<rows_per_chunk;j++){
    double v = a[base + j*rs];
    carry = 0.3*carry + v;
    b[base + j*rs] = carry;
  }
}
// x-pencil: a wave handles one contiguous pencil of n doubles, lane holds n/64 consecutive
template<int Q>
__global__ void k_xwave(const double* __restrict__ a, double* __restrict__ b, size_t npencils, int n){
  size_t wave = (blockIdx.x*(size_t)blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  size_t nw = ((size_t)gridDim.x*blockDim.x)>>6;
  for(size_t p=wave; p<npencils; p+=nw){
    const double* src = a + p*(size_t)n + lane*Q;
    double v[Q];
    #pragma unroll
    for(int q=0;q<Q;q+=2){ double2 t = *(const double2*)(src+q); v[q]=t.x; v[q+1]=t.y; }
    double c=0;
    #pragma unroll
    for(int q=0;q<Q;q++){ c = 0.3*c + v[q]; v[q]=c; }
    double up = __shfl_up(c,1); if(lane==0) up=0;
    #pragma unroll
    for(int q=0;q<Q;q++) v[q]+=0.01*up;
    double* dst = b + p*(size_t)n + lane*Q;
    #pragma unroll
    for(int q=0;q<Q;q+=2){ double2 t; t.x=v[q]; t.y=v[q+1]; *(double2*)(dst+q)=t; }
  }
}
int main(){
  const int nx=512, ny=512, nz=512;
  size_t n=(size_t)nx*ny*nz;
  double *a,*b; CK(hipMalloc(&a,n*8)); CK(hipMalloc(&b,n*8));
  CK(hipMemset(a,0,n*8)); CK(hipMemset(b,0,n*8));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit=[&](const char* name, auto f){
    for(int i=0;i<3;i++) f();
    hipEventRecord(e0); const int R=10; for(int i=0;i<R;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1); ms/=R;
    printf("%-40s %8.3f ms  %8.1f GB/s (r+w)\n", name, ms, 2.0*n*8/ms*1e-6);
  };
  timeit("copy double2 grid=2048x256", [&]{ hipLaunchKernelGGL(k_copy,dim3(2048),dim3(256),0,0,(const double2*)a,(double2*)b,n/2); });
  timeit("copy double2 grid=8192x256", [&]{ hipLaunchKernelGGL(k_copy,dim3(8192),dim3(256),0,0,(const double2*)a,(double2*)b,n/2); });
  timeit("hipMemcpyDtoD", [&]{ hipMemcpyAsync(b,a,n*8,hipMemcpyDeviceToDevice,0); });
  // y-walk: pencil (x,z): w=nx, ps=nx*ny (next z), rs=nx ; total pencils nx*nz
  for(int chunks : {1,2,4,8,16}){
    for(int bs : {64,256}){
      char nm[128]; snprintf(nm,128,"y-walk chunks=%d bs=%d",chunks,bs);
      timeit(nm,[&]{ hipLaunchKernelGGL(k_walk,dim3((size_t)nx*nz/bs,chunks),dim3(bs),0,0,a,b,nx,(size_t)nx*ny,(size_t)nx,ny/chunks,0); });
    }
  }
  // z-walk: pencil (x,y): contiguous index p, rs=nx*ny
  for(int chunks : {1,2,4,8,16}){
    for(int bs : {64,256}){
      char nm[128]; snprintf(nm,128,"z-walk chunks=%d bs=%d",chunks,bs);
      timeit(nm,[&]{ hipLaunchKernelGGL(k_walk,dim3((size_t)nx*ny/bs,chunks),dim3(bs),0,0,a,b,nx*ny,(size_t)0,(size_t)nx*ny,nz/chunks,0); });
    }
  }
  // x wave-per-pencil
  for(int g : {1024,2048,4096,8192,16384}){
    char nm[128]; snprintf(nm,128,"x-wave Q=8 grid=%d bs=256",g);
    timeit(nm,[&]{ hipLaunchKernelGGL(k_xwave<8>,dim3(g),dim3(256),0,0,a,b,(size_t)ny*nz,nx); });
  }
  return 0;
}
