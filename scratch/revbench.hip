// scratch: 1 GiB written front-to-back, then read front-to-back vs back-to-front: does the last-written
// quarter come out of the 256 MB Infinity Cache?
#include <hip/hip_runtime.h>
#include <cstdio>
// chunked sweeps: block b handles chunk order[b]; chunks of 1 MiB, processed in launch order (grid = nchunks)
__global__ void k_w(double2* __restrict__ b, size_t per, int nch, int rev, double v){
  int c = rev ? nch - 1 - blockIdx.x : blockIdx.x; double2* p = b + (size_t)c * per;
  for (size_t i = threadIdx.x; i < per; i += blockDim.x) { double2 t; t.x = v; t.y = v + 1; p[i] = t; } }
__global__ void k_r(const double2* __restrict__ a, double* __restrict__ out, size_t per, int nch, int rev){
  int c = rev ? nch - 1 - blockIdx.x : blockIdx.x; const double2* p = a + (size_t)c * per; double s = 0;
  for (size_t i = threadIdx.x; i < per; i += blockDim.x) { double2 t = p[i]; s += t.x + t.y; }
  if (s == 123.456) out[0] = s; }
int main(){
  const size_t bytes = (size_t)1 << 30, chunk = (size_t)1 << 18;  // 256 KiB chunks -> 4096 blocks
  const int nch = (int)(bytes / chunk); const size_t per = chunk / 16;
  double *buf, *out; hipMalloc(&buf, bytes); hipMalloc(&out, 8); hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rev = 0; rev < 2; rev++) for (int rep = 0; rep < 2; rep++) {
    float tw = 0, tr = 0; const int R = 10;
    for (int it = 0; it < R; it++) {
      float ms;
      hipEventRecord(e0); hipLaunchKernelGGL(k_w, dim3(nch), dim3(256), 0, 0, (double2*)buf, per, nch, 0, (double)it); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tw += ms;
      hipEventRecord(e0); hipLaunchKernelGGL(k_r, dim3(nch), dim3(256), 0, 0, (const double2*)buf, out, per, nch, rev); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tr += ms;
    }
    printf("write fwd %.3f ms (%.0f GB/s); read %s %.3f ms (%.0f GB/s)\n", tw / R, 1073.7 / (tw / R), rev ? "REVERSED" : "forward ", tr / R, 1073.7 / (tr / R));
  }
  // read -> read reuse: read fwd then read rev
  for (int rev = 0; rev < 2; rev++) {
    float tr = 0; const int R = 10;
    for (int it = 0; it < R; it++) {
      float ms;
      hipLaunchKernelGGL(k_r, dim3(nch), dim3(256), 0, 0, (const double2*)buf, out, per, nch, 0);
      hipEventRecord(e0); hipLaunchKernelGGL(k_r, dim3(nch), dim3(256), 0, 0, (const double2*)buf, out, per, nch, rev); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); tr += ms;
    }
    printf("after a forward read: read %s %.3f ms (%.0f GB/s)\n", rev ? "REVERSED" : "forward ", tr / R, 1073.7 / (tr / R));
  }
  return 0; }
