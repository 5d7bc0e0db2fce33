// DPP lane-movement semantics on gfx950, checked against the intended definitions used by xscan.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int RM>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(-1, v, CTRL, RM, 0xf, false); }
__global__ void k(int *o)
{
    const int l = threadIdx.x, v = l;
    o[0 * 64 + l] = dppi<0x111, 0xf>(v);  // row_shr:1   expect l-1 within row else -1
    o[1 * 64 + l] = dppi<0x101, 0xf>(v);  // row_shl:1   expect l+1 within row else -1
    o[2 * 64 + l] = dppi<0x138, 0xf>(v);  // wave_shr:1  expect l-1, lane 0: -1
    o[3 * 64 + l] = dppi<0x130, 0xf>(v);  // wave_shl:1  expect l+1, lane 63: -1
    o[4 * 64 + l] = dppi<0x13C, 0xf>(v);  // wave_ror:1  expect (l+63)%64
    o[5 * 64 + l] = dppi<0x134, 0xf>(v);  // wave_rol:1  expect (l+1)%64
    o[6 * 64 + l] = dppi<0x142, 0xA>(v);  // row_bcast:15 rows 1,3: expect 15 / 47, rows 0,2: -1
    o[7 * 64 + l] = dppi<0x143, 0xC>(v);  // row_bcast:31 rows 2,3: expect 31, rows 0,1: -1
    o[8 * 64 + l] = dppi<0x118, 0xf>(v);  // row_shr:8
    o[9 * 64 + l] = dppi<0x108, 0xf>(v);  // row_shl:8
}
int main()
{
    int *d, h[640];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int r = l >> 4, p = l & 15;
        const int e[10] = {p >= 1 ? l - 1 : -1, p <= 14 ? l + 1 : -1, l >= 1 ? l - 1 : -1, l <= 62 ? l + 1 : -1,
                           (l + 63) % 64, (l + 1) % 64, (r == 1 || r == 3) ? 16 * r - 1 : -1, r >= 2 ? 31 : -1,
                           p >= 8 ? l - 8 : -1, p <= 7 ? l + 8 : -1};
        for (int t = 0; t < 10; t++)
            if (h[t * 64 + l] != e[t]) { if (bad < 20) printf("test %d lane %d: got %d expect %d\n", t, l, h[t * 64 + l], e[t]); bad++; }
    }
    printf("dpptest: %d mismatches\n", bad);
    return bad != 0;
}
