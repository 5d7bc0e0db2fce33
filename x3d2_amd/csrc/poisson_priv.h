// the spectral Poisson solver's state, shared by csrc/poisson.hip (000 / 010 / 100 / 110 solves) and csrc/zfirst.hip
// (the z-first form of the 000 solve at 512^3)
#pragma once
#include <hipfft/hipfft.h>

#include "common.h"

struct x3d_poisson {
    x3d_backend *b;
    int nx, ny, nz;       // cell dims
    int nxm, nxs;         // nxm = nx/2+1 modes per row; nxs = the row PITCH of every spectral-side array: nxm rounded
                          // up to 8 complex numbers (128 B), so that the 128 / 256-byte row segments of the strided
                          // y / z passes are line-aligned (dense rows of 257 made every segment straddle a third
                          // 128-byte line: 1.44-1.6 x the compulsory fetch, round-1 PMC).  Pad columns hold zeros
                          // (waves: ones) and are carried through every kernel; host arrays stay dense.
    hipfftHandle plan_fw, plan_bw;
    real2_t *c;           // spectral workspace [nz][ny][nxs]
    real_t *waves;        // [nz][ny][nxs]
    real_t *rwT;          // [ny][nxs][nz]: -1 / waves (0 where waves < 1e-16), for the fused z pass of fft512.hip
    real_t *ab;           // ax bx ay by az bz
    void *work;
    size_t work_size;
    // stretched y (010): factored pentadiagonal operators, [5][nz][n][nxs] each
    int stretched, sym;   // sym: odd/even rows decoupled (centred, top-bottom); else one full system
    real_t *lu[2];        // sym: odd, even; else lu[0] only
    real_t *luz[2];       // the same operators for the z-first form of the solve (zfirst.hip): [5][nz/2+1][n][nx + 16] --
                          // ALL x modes (mirrored above nx / 2 like the z modes of lu), the z modes 0 .. nz / 2
    size_t c_elems;       // complex numbers allocated behind c (room for either layout)
    // ny = nz = 512: rocFFT does only the contiguous x pass, the strided y / z passes are ours (fft512.hip)
    int fast512;
    int r2c512;  // own single-kernel r2c x pass (fft512.hip) instead of rocFFT's two kernels
    hipfftHandle plan_x_fw, plan_x_bw;
    // ny = 256 (010, the channel case): x and z through 1-D rocFFT plans, y LAST inside the fused y pass (y010.hip)
    int y010;             // 0: not tried yet, 1: plans made, -1: not available for these sizes / switched off
    hipfftHandle plan_x010_fw, plan_x010_bw;  // 2-D over (z, x), batched over the y rows
    real_t *rwZ;          // z-first solve: [nz/2+1][nx][ny] reciprocal wave numbers (built on first use, zfirst.hip)
    // PROXY of a z-first solve that lives outside this object (round 6: the Fortran shim's y-slab solve, csrc/sfftz.hip +
    // fortran/m_hip_backend.f90): only b and the fields below are valid.  The three hooks and x3d_poisson_solve_000 then
    // mean: z transform of the field onto ext_c ; ext_middle(ext_user) = everything between the two z transforms (x, the
    // all-to-alls, y + division, back) ; inverse z transform -- and the deferred layer's z-first rewrite (the z transforms
    // on the tiles of the neighbouring z operator pairs) applies to the recorded hooks as on one rank
    int (*ext_middle)(void *user);
    void *ext_user;
    real2_t *ext_c;       // C[257][ext_ny][ext_px]
    int ext_ny;
    long ext_px;
};
