// LDS read throughput per instruction kind: 16 waves per CU, every lane reads lane-contiguous doubles
//   hipcc -O2 --offload-arch=gfx950 scratch/ldsbench.hip -o scratch/ldsbench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(1024) k(double *out, int iters)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.5;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned addr = lane * (KIND == 2 ? 16 : 8);
    // round 4: what does a table read cost when most lanes want the SAME value (periodic operators on a uniform grid:
    // lanes 8..55 of every entry are bitwise equal)?  KIND 3: all lanes one address (broadcast); 4: the compressed
    // layout (lanes 0..7 | one address for 8..55 | 56..63: 17 doubles per entry); 5 / 6: ds_read_b64 with only the 16
    // edge lanes (0..7, 56..63) / only lanes 0..15 active
    if (KIND == 3) addr = 0;
    if (KIND == 4) addr = (lane < 8 ? lane : (lane > 55 ? lane - 47 : 8)) * 8;
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        if (KIND == 5 || KIND == 6) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
            if (KIND == 5 ? (lane < 8 || lane > 55) : lane < 16) {
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                             "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr) : "memory");
            }
            acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        } else if (KIND == 0 || KIND == 3 || KIND == 4) {  // 8 x ds_read_b64, entries 512 B apart
            double a0, a1, a2, a3, a4, a5, a6, a7;
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                         "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr) : "memory");
            acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        } else if (KIND == 1) {  // 4 x ds_read2st64_b64 (the same 8 entries)
            double2 b0, b1, b2, b3;
            asm volatile("ds_read2st64_b64 %0, %4 offset1:1\n ds_read2st64_b64 %1, %4 offset0:2 offset1:3\n"
                         "ds_read2st64_b64 %2, %4 offset0:4 offset1:5\n ds_read2st64_b64 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr) : "memory");
            acc += b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y;
        } else {  // 4 x ds_read_b128, pairs of entries adjacent per lane: [pair][lane][2]
            double2 b0, b1, b2, b3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr) : "memory");
            acc += b0.x + b0.y + b1.x + b1.y + b2.x + b2.y + b3.x + b3.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND> void run(const char *name, double *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 131072, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 131072, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: 16 waves x iters x 8 entries of 512 B
    const double bytes = 16.0 * iters * 8 * 512, clk = ms * 1e-3 * 2.4e9;
    printf("%-22s %7.3f ms  %.1f B/clk/CU (at 2.4 GHz)  %.2f clk per entry-read per wave\n", name, ms, bytes / clk, clk / (16.0 * iters * 8));
}
int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8);
    hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    run<0>("ds_read_b64", out); run<1>("ds_read2st64_b64", out); run<2>("ds_read_b128", out);
    run<0>("ds_read_b64", out); run<1>("ds_read2st64_b64", out); run<2>("ds_read_b128", out);
    hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int r = 0; r < 2; r++) {
        run<3>("b64, all lanes one address", out); run<4>("b64, compressed layout", out);
        run<5>("b64, 16 edge lanes active", out); run<6>("b64, lanes 0..15 active", out);
    }
    return 0;
}
