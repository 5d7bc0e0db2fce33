// z stage of the slab 010 solver (csrc/sfft010.hip): W[nz][cols] complex, transform along nz for every column.
// rocFFT's strided 1-D plan on W itself against 32 x 32 LDS-tiled transposes around a contiguous plan.
//   hipcc -O2 --offload-arch=gfx950 scratch/zfft_bench.hip -lhipfft -o scratch/zfft_bench && scratch/zfft_bench
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <vector>
#include <functional>

__global__ void __launch_bounds__(256) k_transpose(double2 *__restrict__ dst, const double2 *__restrict__ src, int nA, int nB)
{
    __shared__ double2 tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int a0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    for (int r = 0; r < 4; r++) {
        const int bb = b0 + ty + 8 * r, aa = a0 + tx;
        if (aa < nA && bb < nB) tile[ty + 8 * r][tx] = src[(long)bb * nA + aa];
    }
    __syncthreads();
    for (int r = 0; r < 4; r++) {
        const int aa = a0 + ty + 8 * r, bb = b0 + tx;
        if (aa < nA && bb < nB) dst[(long)aa * nB + bb] = tile[tx][ty + 8 * r];
    }
}

static float run(hipStream_t st, int reps, const std::function<void()> &f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipEventRecord(a, st);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main()
{
    const int cases[3][2] = {{512, 256 * 513}, {4096, 256 * 65}, {1024, 256 * 257}};
    for (auto &c : cases) {
        const int nz = c[0], cols = c[1];
        const size_t n = (size_t)nz * cols;
        double2 *w, *t;
        hipMalloc(&w, n * sizeof(double2)); hipMalloc(&t, n * sizeof(double2));
        hipMemset(w, 0, n * sizeof(double2));
        hipfftHandle ps, pc;
        int len[1] = {nz}, emb[1] = {nz};
        size_t ws;
        hipfftCreate(&ps); hipfftCreate(&pc);
        if (hipfftMakePlanMany(ps, 1, len, emb, cols, 1, emb, cols, 1, HIPFFT_Z2Z, cols, &ws) != HIPFFT_SUCCESS) { printf("strided plan failed\n"); return 1; }
        if (hipfftMakePlanMany(pc, 1, len, emb, 1, nz, emb, 1, nz, HIPFFT_Z2Z, cols, &ws) != HIPFFT_SUCCESS) { printf("contiguous plan failed\n"); return 1; }
        const float s = run(0, 5, [&] { hipfftExecZ2Z(ps, (hipfftDoubleComplex *)w, (hipfftDoubleComplex *)w, HIPFFT_FORWARD); });
        const float tr = run(0, 5, [&] { hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (nz + 31) / 32), dim3(256), 0, 0, t, w, cols, nz); });
        const float cc = run(0, 5, [&] { hipfftExecZ2Z(pc, (hipfftDoubleComplex *)t, (hipfftDoubleComplex *)t, HIPFFT_FORWARD); });
        printf("nz %5d cols %7d (%.2f GB): strided plan %.3f ms | transpose %.3f + contiguous plan %.3f (+ transpose back) = %.3f ms per direction\n",
               nz, cols, n * 16.0 / 1e9, s, tr, cc, 2 * tr + cc);
        hipfftDestroy(ps); hipfftDestroy(pc);
        hipFree(w); hipFree(t);
    }
    return 0;
}
