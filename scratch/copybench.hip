// Scratch microbenchmark (not part of the product): what does a plain streaming copy reach on this pool's MI355X?
// The guide (MI355X_MICROARCH.md, "HBM") states 6.29 TB/s for a float4 copy; scratch/membench.hip measured 4.8-5.7.
// Variants: bytes per lane and instruction, loads in flight per lane (unroll), grid shape (grid-stride vs one chunk
// per workgroup), nontemporal loads / stores, read-only and write-only streams, 2R:1W and 3R:1W mixes, buffer sizes.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/copybench scratch/copybench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);}}while(0)

typedef float v4 __attribute__((ext_vector_type(4)));

template<int NT> __device__ __forceinline__ v4 ld(const v4* p){
  if (NT) return __builtin_nontemporal_load(p); else return *p;
}
template<int NT> __device__ __forceinline__ void st(v4* p, v4 v){
  if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// grid-stride copy, U independent 16-byte loads in flight per lane
template<int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_copy_gs(const v4* __restrict__ a, v4* __restrict__ b, size_t n){
  size_t st_ = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * st_ < n; i += U * st_) {
    v4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld<NTL>(a + i + u * st_);
#pragma unroll
    for (int u = 0; u < U; u++) st<NTS>(b + i + u * st_, v[u]);
  }
  for (; i < n; i += st_) st<NTS>(b + i, ld<NTL>(a + i));
}

// one contiguous chunk per workgroup: workgroup w copies [w*chunk, (w+1)*chunk); U loads in flight
template<int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_copy_chunk(const v4* __restrict__ a, v4* __restrict__ b, size_t chunk){
  const v4* s = a + blockIdx.x * chunk;
  v4* d = b + blockIdx.x * chunk;
  for (size_t i = threadIdx.x; i < chunk; i += 256 * U) {
    v4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld<NTL>(s + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) st<NTS>(d + i + u * 256, v[u]);
  }
}

template<int U, int NTL>
__global__ void __launch_bounds__(256) k_read(const v4* __restrict__ a, float* __restrict__ out, size_t n){
  size_t st_ = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  v4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * st_ < n; i += U * st_) {
    v4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ld<NTL>(a + i + u * st_);
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}

template<int U, int NTS>
__global__ void __launch_bounds__(256) k_write(v4* __restrict__ b, size_t n){
  size_t st_ = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  v4 v = {1, 2, 3, 4};
  for (; i + (U - 1) * st_ < n; i += U * st_) {
#pragma unroll
    for (int u = 0; u < U; u++) st<NTS>(b + i + u * st_, v);
  }
}

// R inputs summed into one output (the mix of the derivative kernels: 2R:1W, 3R:1W, 4R:1W)
template<int R, int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_mix(const v4* __restrict__ a, v4* __restrict__ b, size_t n, size_t fld){
  size_t st_ = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * st_ < n; i += U * st_) {
    v4 v[U][R];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < R; r++) v[u][r] = ld<NTL>(a + r * fld + i + u * st_);
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4 s = v[u][0];
#pragma unroll
      for (int r = 1; r < R; r++) s += v[u][r];
      st<NTS>(b + i + u * st_, s);
    }
  }
}

// the same mix with one contiguous chunk per workgroup (the shape that reaches 6.2 TB/s as a copy)
template<int R, int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_mix_chunk(const v4* __restrict__ a, v4* __restrict__ b, size_t chunk, size_t fld){
  const v4* s = a + blockIdx.x * chunk;
  v4* d = b + blockIdx.x * chunk;
  for (size_t i = threadIdx.x; i < chunk; i += 256 * U) {
    v4 v[U][R];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < R; r++) v[u][r] = ld<NTL>(s + r * fld + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4 t = v[u][0];
#pragma unroll
      for (int r = 1; r < R; r++) t += v[u][r];
      st<NTS>(d + i + u * 256, t);
    }
  }
}
// 2R:2W (k_xscan_tds_lin's first stage: u, rhs in; u_new, du out) and 3R:2W
template<int R, int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_mix2w_chunk(const v4* __restrict__ a, v4* __restrict__ b, size_t chunk, size_t fld){
  const v4* s = a + blockIdx.x * chunk;
  v4* d = b + blockIdx.x * chunk;
  for (size_t i = threadIdx.x; i < chunk; i += 256 * U) {
    v4 v[U][R];
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int r = 0; r < R; r++) v[u][r] = ld<NTL>(s + r * fld + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) {
      v4 t = v[u][0];
#pragma unroll
      for (int r = 1; r < R; r++) t += v[u][r];
      st<NTS>(d + i + u * 256, t);
      st<NTS>(d + fld + i + u * 256, t * 2.0f);
    }
  }
}

// in-place update b += a (the accumulating kernels' rhs read-modify-write): 2R:1W with the write onto a read line
template<int U>
__global__ void __launch_bounds__(256) k_rmw(const v4* __restrict__ a, v4* __restrict__ b, size_t n){
  size_t st_ = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * st_ < n; i += U * st_) {
    v4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; u++) { v[u] = a[i + u * st_]; w[u] = b[i + u * st_]; }
#pragma unroll
    for (int u = 0; u < U; u++) b[i + u * st_] = v[u] + w[u];
  }
}

int main(int argc, char** argv){
  size_t bytes = (argc > 1 ? atof(argv[1]) : 1.0) * (1ull << 30);   // per buffer
  size_t n = bytes / 16;
  v4 *a, *b; float* out;
  CK(hipMalloc(&a, 4 * bytes)); CK(hipMalloc(&b, 2 * bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, 4 * bytes)); CK(hipMemset(b, 0, 2 * bytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int dev; hipGetDevice(&dev); hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
  printf("# %s, %d CUs, buffers of %.2f GiB\n", pr.name, pr.multiProcessorCount, bytes / double(1ull << 30));
  auto timeit = [&](const char* name, double moved, auto f){
    for (int i = 0; i < 2; i++) f();
    std::vector<float> t;
    for (int r = 0; r < 7; r++) {
      hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-58s best %7.3f ms median %7.3f ms  %7.1f GB/s (median)  %7.1f (best)\n", name, t[0], t[3], moved / t[3] * 1e-6, moved / t[0] * 1e-6);
    fflush(stdout);
  };
  char nm[160];
  timeit("hipMemcpyAsync DtoD", 2.0 * bytes, [&]{ CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
#define GS(U, NTL, NTS, G) snprintf(nm, 160, "copy grid-stride 16B U=%d ntl=%d nts=%d grid=%d", U, NTL, NTS, G); \
  timeit(nm, 2.0 * bytes, [&]{ hipLaunchKernelGGL((k_copy_gs<U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n); });
  for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64, 256 * 256}) {
    GS(1, 0, 0, g) GS(2, 0, 0, g) GS(4, 0, 0, g) GS(8, 0, 0, g)
  }
  for (int g : {256 * 8, 256 * 32}) {
    GS(4, 1, 0, g) GS(4, 0, 1, g) GS(4, 1, 1, g) GS(8, 1, 1, g)
  }
#define CH(U, NTL, NTS, G) snprintf(nm, 160, "copy chunk-per-workgroup 16B U=%d ntl=%d nts=%d grid=%d", U, NTL, NTS, G); \
  timeit(nm, 2.0 * bytes, [&]{ hipLaunchKernelGGL((k_copy_chunk<U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n / (G)); });
  for (int g : {256 * 8, 256 * 32, 256 * 128}) { CH(4, 0, 0, g) CH(8, 0, 0, g) CH(4, 1, 1, g) }
#define RD(U, NTL, G) snprintf(nm, 160, "read-only 16B U=%d ntl=%d grid=%d", U, NTL, G); \
  timeit(nm, 1.0 * bytes, [&]{ hipLaunchKernelGGL((k_read<U, NTL>), dim3(G), dim3(256), 0, 0, a, out, n); });
  for (int g : {256 * 8, 256 * 32}) { RD(4, 0, g) RD(8, 0, g) RD(8, 1, g) }
#define WR(U, NTS, G) snprintf(nm, 160, "write-only 16B U=%d nts=%d grid=%d", U, NTS, G); \
  timeit(nm, 1.0 * bytes, [&]{ hipLaunchKernelGGL((k_write<U, NTS>), dim3(G), dim3(256), 0, 0, b, n); });
  for (int g : {256 * 8, 256 * 32}) { WR(4, 0, g) WR(8, 0, g) WR(8, 1, g) }
#define MX(R, U, NTL, NTS, G) snprintf(nm, 160, "mix %dR:1W 16B U=%d ntl=%d nts=%d grid=%d", R, U, NTL, NTS, G); \
  timeit(nm, (R + 1.0) * bytes, [&]{ hipLaunchKernelGGL((k_mix<R, U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n, n); });
  for (int g : {256 * 8, 256 * 32}) { MX(2, 4, 0, 0, g) MX(3, 2, 0, 0, g) MX(3, 4, 0, 0, g) MX(4, 2, 0, 0, g) MX(3, 2, 1, 1, g) }
#define RM(U, G) snprintf(nm, 160, "in-place b += a (2R:1W, rmw) 16B U=%d grid=%d", U, G); \
  timeit(nm, 3.0 * bytes, [&]{ hipLaunchKernelGGL((k_rmw<U>), dim3(G), dim3(256), 0, 0, a, b, n); });
  for (int g : {256 * 8, 256 * 32}) { RM(2, g) RM(4, g) }
#define MC(R, U, NTL, NTS, G) snprintf(nm, 160, "mix %dR:1W chunk-per-workgroup U=%d ntl=%d nts=%d grid=%d", R, U, NTL, NTS, G); \
  timeit(nm, (R + 1.0) * bytes, [&]{ hipLaunchKernelGGL((k_mix_chunk<R, U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n / (G), n); });
#define M2(R, U, NTL, NTS, G) snprintf(nm, 160, "mix %dR:2W chunk-per-workgroup U=%d ntl=%d nts=%d grid=%d", R, U, NTL, NTS, G); \
  timeit(nm, (R + 2.0) * bytes, [&]{ hipLaunchKernelGGL((k_mix2w_chunk<R, U, NTL, NTS>), dim3(G), dim3(256), 0, 0, a, b, n / (G), n); });
  for (int g : {256 * 32, 256 * 128}) {
    MC(2, 2, 0, 0, g) MC(2, 2, 1, 1, g) MC(3, 2, 0, 0, g) MC(3, 2, 1, 1, g) MC(3, 1, 1, 1, g) MC(4, 1, 1, 1, g)
    M2(2, 2, 0, 0, g) M2(2, 2, 1, 1, g) M2(3, 2, 1, 1, g) M2(3, 1, 0, 0, g)
  }
  return 0;
}
