#!/usr/bin/env python3
"""z stage of the slab Poisson solver for pz = 1, 2, 4, 8 ranks, timed in ONE process (the stage is local: transposes
+ rocFFT + spectral kernel, or the single fused kernel k_fft512_peers): python scratch/zstage_bench.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from x3d2_amd import Mesh, _lib
from x3d2_amd.backend import HipBackend
VP = ctypes.c_void_p
mesh = Mesh((512, 512, 512), (1, 1, 1), (6.283185307179586,) * 3, ("periodic",) * 2, ("periodic",) * 2, ("periodic",) * 2)
b = HipBackend(mesh)
lib = b.lib
for fused in (1, 0):
    os.environ["X3D_NO_SLAB_FUSED_Z"] = "0" if fused else "1"
    for pz in (1, 2, 4, 8):
        h = VP()
        _lib.check(lib.x3d_sfft_create_parts(b.h, ctypes.byref(h), _lib.ints(512, 512, 512 * pz), pz, 0, 4))
        sz = (ctypes.c_long * 4)()
        _lib.check(lib.x3d_sfft_sizes(h, sz))
        chunk, zl, ys, nxs = [int(v) for v in sz]
        nz = 512 * pz
        w = np.ones((ys, nxs, nz)); one = lambda n: np.ones(n)
        keep = [np.ascontiguousarray(a) for a in (w, one(512), one(512), one(512), one(512), one(nz), one(nz))]
        _lib.check(lib.x3d_sfft_set_waves(h, *[a.ctypes.data_as(_lib.c_double_p) for a in keep]))
        rb = torch.randn(2 * pz * chunk, dtype=torch.float64, device=b.device)
        def stage():
            for m in range(4):
                _lib.check(lib.x3d_sfft_fft_z_part(h, rb.data_ptr(), 0, m))
                _lib.check(lib.x3d_sfft_postprocess_000_part(h, rb.data_ptr(), m))
                _lib.check(lib.x3d_sfft_fft_z_part(h, rb.data_ptr(), 1, m))
        for _ in range(2):
            stage()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            stage()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        gb = 2.0 * pz * chunk * 8 / 1e9
        print("fused %d  pz %d: z stage %.3f ms  (%.2f GB array; %.2f TB/s on read + write once)" % (fused, pz, ms, gb, 2 * gb / ms))
        lib.x3d_sfft_destroy(h)
