"""ctypes binding of libx3d2_hip.so (the C ABI declared in include/x3d2_hip.h).

The HIP library is the product: there is no CPU fallback.  Importing this
module never needs a GPU, but every compute entry point does."""
import ctypes
import os
import subprocess

from .common import X3dError

HERE = os.path.dirname(os.path.abspath(__file__))
# X3D_SINGLE_PREC=1 (read once, at import): the FP32 flavour of the library (libx3d2_hip_sp.so = the same sources compiled
# with -DX3D_SINGLE_PREC; the reference's -DSINGLE_PREC, src/common.f90:6-12) -- fields, tables, scalars and transforms
# on 4-byte reals.  The host side keeps computing coefficients and wave numbers in float64 and converts at this boundary.
SINGLE = os.environ.get("X3D_SINGLE_PREC") == "1"
LIB_PATH = os.path.join(HERE, "libx3d2_hip_sp.so" if SINGLE else "libx3d2_hip.so")
CSRC = os.path.join(HERE, "csrc")

REAL = ctypes.c_float if SINGLE else ctypes.c_double  # x3d_real of include/x3d2_hip.h
NP_REAL = "float32" if SINGLE else "float64"
c_double_p = ctypes.POINTER(REAL)  # (name kept from the FP64-only days: pointer to the library's real kind)
c_int_p = ctypes.POINTER(ctypes.c_int)
VP = ctypes.c_void_p
I, D, SZT = ctypes.c_int, REAL, ctypes.c_size_t


def torch_real():
    import torch
    return torch.float32 if SINGLE else torch.float64

# name -> (restype, argtypes); kept in step with include/x3d2_hip.h
# (tests/test_abi.py parses the header and compares)
PROTOTYPES = {
    "x3d_last_error": (ctypes.c_char_p, []),
    "x3d_abi_version": (I, []),
    "x3d_real_bytes": (I, []),
    "x3d_backend_create": (I, [ctypes.POINTER(VP), c_int_p, I, VP]),
    "x3d_backend_destroy": (I, [VP]),
    "x3d_backend_create_like": (I, [ctypes.POINTER(VP), VP, c_int_p]),
    "x3d_lazy_enable": (I, [VP, I]),
    "x3d_lazy_set_dist_transeq": (I, [VP, ctypes.c_uint, VP, VP]),
    "x3d_lazy_set_dist_tds": (I, [VP, ctypes.c_uint, VP, VP]),  # (fn: a C function pointer; the Fortran shim's use)
    "x3d_lazy_flush": (I, [VP]),
    "x3d_lazy_sync": (I, [VP]),
    "x3d_lazy_register_block": (I, [VP, VP]),
    "x3d_lazy_unregister_block": (I, [VP, VP]),
    "x3d_tds_pair_zfirst_ok": (I, [VP, VP, VP, c_int_p]),
    "x3d_block_discard": (I, [VP, VP]),
    "x3d_lazy_stats": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_backend_set_stream": (I, [VP, VP]),
    "x3d_backend_set_comm_reserve": (I, [VP, I]),
    "x3d_backend_set_ring": (I, [VP, I, I]),
    "x3d_block_elems": (SZT, [VP]),
    "x3d_padded_dims": (I, [VP, c_int_p]),
    "x3d_device_sync": (I, [VP]),
    "x3d_block_alloc": (I, [VP, ctypes.POINTER(VP)]),
    "x3d_block_free": (I, [VP, VP]),
    "x3d_transpose_xy": (I, [VP, VP, VP, VP, I, I, I]),
    "x3d_transpose_xyz_zxy": (I, [VP, VP, VP, VP, I, I, I]),
    "x3d_transpose_zxy_xyz": (I, [VP, VP, VP, VP, I, I, I]),
    "x3d_poisson_enforce_periodicity_z": (I, [VP, VP, VP]),
    "x3d_poisson_undo_periodicity_z": (I, [VP, VP, VP]),
    "x3d_poisson_postprocess_011": (I, [VP]),
    "x3d_device_alloc": (I, [VP, ctypes.POINTER(VP), ctypes.c_long]),
    "x3d_device_free": (I, [VP, VP]),
    "x3d_copy_to_host": (I, [VP, VP, VP, ctypes.c_long]),
    "x3d_copy_to_device": (I, [VP, VP, VP, ctypes.c_long]),
    "x3d_device_count": (I, [c_int_p]),
    "x3d_ipc_export": (I, [VP, VP, VP]),
    "x3d_ipc_open": (I, [VP, VP, ctypes.POINTER(VP)]),
    "x3d_ipc_close": (I, [VP, VP]),
    "x3d_copy_device": (I, [VP, VP, VP, ctypes.c_long]),
    "x3d_block_fill": (I, [VP, VP, D]),
    "x3d_tdsops_create": (I, [VP, ctypes.POINTER(VP), I, I, I, I] + [c_double_p] * 10),
    "x3d_tdsops_destroy": (I, [VP]),
    "x3d_tdsops_set_penta": (I, [VP, D, D, D] + [c_double_p] * 6 + [I]),
    "x3d_tds_penta_solve": (I, [VP, VP, VP, VP, I, VP, VP]),
    "x3d_tds_solve": (I, [VP, VP, VP, VP, I]),
    "x3d_tds_solve_acc": (I, [VP, VP, VP, VP, I, I, D]),
    "x3d_transeq_acc": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, I]),
    "x3d_npencils": (I, [VP, I]),
    "x3d_pack_halos": (I, [VP, VP, VP, VP, I, I]),
    "x3d_tds_dist_fwd": (I, [VP, VP, VP, VP, VP, VP, VP, VP, I]),
    "x3d_tds_solve_pair": (I, [VP, I, I, VP, VP, VP, VP, VP, VP]),
    "x3d_tds_solve_pair_yperm": (I, [VP, I, VP, VP, VP, VP, VP, VP, I, c_int_p]),
    "x3d_tdsops_halo_rows": (I, [VP, c_int_p]),
    "x3d_tdsops_dims": (I, [VP, c_int_p]),
    "x3d_halo_row_size": (ctypes.c_long, [VP, I]),
    "x3d_pack_halos_multi": (I, [VP, VP, ctypes.POINTER(VP), I, I, I]),
    "x3d_transeq_tile": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, I, VP, VP, I, I, c_int_p]),
    "x3d_transeq_halo_fix": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP]),
    "x3d_tds_pair_tile": (I, [VP, I, I, VP, VP, VP, VP, VP, VP, VP, VP, I, I, c_int_p]),
    "x3d_tds_pair_halo_fix": (I, [VP, I, I, VP, VP, VP, VP, VP]),
    "x3d_tds_pair_tile_yperm": (I, [VP, I, VP, VP, VP, VP, VP, VP, VP, VP, I, c_int_p]),
    "x3d_tds_pair_halo_fix_yperm": (I, [VP, I, VP, VP, VP, VP, VP, I]),
    "x3d_tds_solve_lincomb": (I, [VP, I, VP, VP, VP, VP, I, c_double_p, ctypes.POINTER(VP)]),
    "x3d_tds_solve_lincomb_wall": (I, [VP, I, VP, VP, VP, VP, I, c_double_p, ctypes.POINTER(VP), VP]),
    "x3d_tds_solve_mean": (I, [VP, VP, VP, VP, I, c_int_p, D, D, ctypes.POINTER(VP)]),
    "x3d_tds_solve_lincomb_wall_mean": (I, [VP, I, VP, VP, VP, VP, I, c_double_p, ctypes.POINTER(VP), VP, c_int_p, D, D,
                                            ctypes.POINTER(VP)]),
    "x3d_tds_dist_bwd": (I, [VP, VP, VP, VP, VP, VP, I]),
    "x3d_tds_dist_bwd_acc": (I, [VP, VP, VP, VP, VP, VP, I, I, D]),
    "x3d_transeq": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP]),
    "x3d_compute_vorticity": (I, [VP, VP, ctypes.POINTER(VP)]),
    "x3d_compute_qcriterion": (I, [VP, VP, ctypes.POINTER(VP)]),
    "x3d_transeq_species": (I, [VP, I, VP, VP, VP, D, VP, VP, VP, I]),
    "x3d_transeq_dist_fwd": (I, [VP, I, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "x3d_transeq_dist_bwd": (I, [VP, I, VP, VP, VP, VP, VP, D, VP, VP, VP]),
    "x3d_transeq_dist_bwd_acc": (I, [VP, I, VP, VP, VP, VP, VP, D, VP, VP, VP, I]),
    "x3d_reorder": (I, [VP, VP, VP, I]),
    "x3d_sum_intox": (I, [VP, VP, VP, I]),
    "x3d_veccopy": (I, [VP, VP, VP]),
    "x3d_vecadd": (I, [VP, D, VP, D, VP]),
    "x3d_vecmult": (I, [VP, VP, VP]),
    "x3d_field_scale": (I, [VP, VP, D]),
    "x3d_field_shift": (I, [VP, VP, D]),
    "x3d_backend_counter": (ctypes.c_long, [VP, I]),
    "x3d_lincomb": (I, [VP, VP, VP, I, c_double_p, ctypes.POINTER(VP)]),
    "x3d_transeq_x_rot": (I, [VP, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, D, VP, c_int_p]),
    "x3d_transeq_x_update": (I, [VP, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, VP, VP, VP, VP, VP, D, c_int_p]),
    "x3d_transeq_x_update_rot": (I, [VP, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, VP, VP, VP, VP, VP, D, D, VP, c_int_p]),
    "x3d_transeq_defer": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, c_int_p]),
    "x3d_pending_flush": (I, [VP, I, VP, VP]),
    "x3d_transeq_stage_ok": (I, [VP, I, VP, VP, VP, VP]),
    "x3d_transeq_lincomb": (I, [VP, I, I, VP, VP, D, VP, VP, VP, VP, VP, VP, I, c_double_p, ctypes.POINTER(VP), I, I]),
    "x3d_transeq_lincomb3": (I, [VP, I, VP, VP, VP, VP, VP, VP, D, VP, VP, VP, VP, ctypes.POINTER(VP), ctypes.POINTER(VP), c_int_p,
                                 c_double_p, ctypes.POINTER(VP), c_int_p, c_int_p, c_int_p]),
    "x3d_lincomb_pending": (I, [VP, I, VP, VP, I, c_double_p, ctypes.POINTER(VP), I, VP, I]),
    "x3d_scalar_product": (I, [VP, VP, VP, c_int_p, c_double_p]),
    "x3d_field_max_sum": (I, [VP, VP, c_int_p, c_double_p, c_double_p]),
    "x3d_field_volume_integral": (I, [VP, VP, c_int_p, c_double_p]),
    "x3d_field_shift_to_mean": (I, [VP, VP, c_int_p, D, D]),
    "x3d_field_mean_shift": (I, [VP, VP, c_int_p, D, D, ctypes.POINTER(VP)]),
    "x3d_field_shift_by": (I, [VP, VP, VP]),
    "x3d_wall_noise": (I, [VP, VP, c_int_p, D, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "x3d_slice_max_sum": (I, [VP, VP, c_int_p, I, I, c_double_p, c_double_p]),
    "x3d_field_set_face": (I, [VP, VP, c_int_p, D, D, I]),
    "x3d_field_set_face_from_field": (I, [VP, VP, VP, c_int_p, D, I, D]),
    "x3d_set_field_data": (I, [VP, VP, c_double_p, c_int_p]),
    "x3d_get_field_data": (I, [VP, c_double_p, VP, c_int_p]),
    "x3d_set_field_data_pitched": (I, [VP, VP, c_double_p, I, I, c_int_p]),
    "x3d_get_field_data_pitched": (I, [VP, c_double_p, VP, I, I, c_int_p]),
    "x3d_poisson_create": (I, [VP, ctypes.POINTER(VP), c_int_p] + [c_double_p] * 7),
    "x3d_poisson_destroy": (I, [VP]),
    "x3d_poisson_fft_forward": (I, [VP, VP]),
    "x3d_poisson_postprocess_000": (I, [VP]),
    "x3d_poisson_fft_backward": (I, [VP, VP]),
    "x3d_poisson_solve_000": (I, [VP, VP]),
    "x3d_poisson_enforce_periodicity_y": (I, [VP, VP, VP]),
    "x3d_poisson_undo_periodicity_y": (I, [VP, VP, VP]),
    "x3d_poisson_set_stretching": (I, [VP, ctypes.c_int, c_double_p, c_double_p]),
    "x3d_sfftz_spectrum": (I, [VP, ctypes.POINTER(VP), c_int_p, ctypes.POINTER(ctypes.c_long)]),
    "x3d_poisson_create_proxy": (I, [VP, ctypes.POINTER(VP), VP, I, ctypes.c_long, VP, VP]),
    "x3d_poisson_set_stretching_zfirst": (I, [VP, ctypes.c_int, c_double_p, c_double_p]),
    "x3d_poisson_postprocess_010": (I, [VP]),
    "x3d_poisson_solve_010": (I, [VP, VP, VP]),
    "x3d_poisson_solve_010_rows": (I, [VP, VP]),
    "x3d_poisson_solve_010_rows_zfirst": (I, [VP, VP]),
    "x3d_poisson_get_spectral": (I, [VP, c_double_p]),
    "x3d_poisson_set_spectral": (I, [VP, c_double_p]),
    "x3d_sfft_create": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I]),
    "x3d_sfft_create_parts": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I, I]),
    "x3d_sfft_fft_z_part": (I, [VP, VP, I, I]),
    "x3d_sfft_postprocess_000_part": (I, [VP, VP, I]),
    "x3d_sfft_destroy": (I, [VP]),
    "x3d_sfft_sizes": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_sfft_set_waves": (I, [VP] + [c_double_p] * 7),
    "x3d_sfft_forward_local": (I, [VP, VP, VP]),
    "x3d_sfft_fft_z": (I, [VP, VP, I]),
    "x3d_sfft_postprocess_000": (I, [VP, VP]),
    "x3d_sfft_backward_local": (I, [VP, VP, VP]),
    "x3d_sfft010_create": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I]),
    "x3d_sfft010_create_parts": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I, I]),
    "x3d_sfft010_fft_z_part": (I, [VP, VP, I, I]),
    "x3d_sfft010_postprocess_010_part": (I, [VP, VP, I]),
    "x3d_sfft010_destroy": (I, [VP]),
    "x3d_sfft010_sizes": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_sfft010_set_waves": (I, [VP] + [c_double_p] * 7),
    "x3d_sfft010_set_stretching": (I, [VP, I, c_double_p, c_double_p]),
    "x3d_sfft010_periodicity_y": (I, [VP, VP, VP, I]),
    "x3d_sfft010_forward_local": (I, [VP, VP, VP]),
    "x3d_sfft010_fft_z": (I, [VP, VP, I]),
    "x3d_sfft010_postprocess_010": (I, [VP, VP]),
    "x3d_sfft010_backward_local": (I, [VP, VP, VP]),
    "x3d_poisson_zfirst_ok": (I, [VP, c_int_p]),
    "x3d_poisson_zfirst_middle": (I, [VP]),
    "x3d_poisson_zfirst_forward": (I, [VP, VP]),
    "x3d_poisson_zfirst_backward": (I, [VP, VP]),
    "x3d_poisson_solve_000_zfirst": (I, [VP, VP]),
    "x3d_tds_pair_zfirst": (I, [VP, VP, I, VP, VP, VP, VP, VP, VP, c_int_p]),
    "x3d_sfftz_create": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I, I]),
    "x3d_sfftz_destroy": (I, [VP]),
    "x3d_sfftz_sizes": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_sfftz_set_waves": (I, [VP] + [c_double_p] * 7),
    "x3d_sfftz_tds_pair": (I, [VP, I, VP, VP, VP, VP, VP, VP, c_int_p]),
    "x3d_sfftz_z": (I, [VP, VP, I]),
    "x3d_sfftz_z_field": (I, [VP, VP, I]),
    "x3d_sfftz_x_forward": (I, [VP, VP, I]),
    "x3d_sfftz_y_stage": (I, [VP, VP, I, I]),
    "x3d_sfftz_x_backward": (I, [VP, VP, I]),
    "x3d_sfftz_tds_pair_rows": (I, [VP, I, VP, VP, VP, VP, VP, VP, I, I, c_int_p]),
    "x3d_sfftz_z_rows": (I, [VP, VP, I, I, I]),
    "x3d_sfftz_x_forward_rows": (I, [VP, VP, I, I, I]),
    "x3d_sfftz_x_backward_rows": (I, [VP, VP, I, I, I]),
    "x3d_pfft_create": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I, I, I]),
    "x3d_pfft_destroy": (I, [VP]),
    "x3d_pfft_sizes": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_pfft_set_waves": (I, [VP] + [c_double_p] * 7),
    "x3d_pfft_fwd_x": (I, [VP, VP]),
    "x3d_pfft_bwd_x": (I, [VP, VP]),
    "x3d_pfft_fft_y": (I, [VP, I]),
    "x3d_pfft_fft_z": (I, [VP, I]),
    "x3d_pfft_pack_xy": (I, [VP, VP]),
    "x3d_pfft_unpack_xy": (I, [VP, VP]),
    "x3d_pfft_pack_yx": (I, [VP, VP]),
    "x3d_pfft_unpack_yx": (I, [VP, VP]),
    "x3d_pfft_pack_yz": (I, [VP, VP]),
    "x3d_pfft_unpack_yz": (I, [VP, VP]),
    "x3d_pfft_pack_zy": (I, [VP, VP]),
    "x3d_pfft_unpack_zy": (I, [VP, VP]),
    "x3d_pfft_transpose_local": (I, [VP, I]),
    "x3d_pfft_own_chunk": (I, [VP, VP, I]),
    "x3d_pfft_postprocess_000": (I, [VP]),
    "x3d_pfft_create_parts": (I, [VP, ctypes.POINTER(VP), c_int_p, I, I, I, I, I]),
    "x3d_pfft_part_layout": (I, [VP, ctypes.POINTER(ctypes.c_long)]),
    "x3d_pfft_fwd_a_part": (I, [VP, VP, VP, I]),
    "x3d_pfft_fwd_b_part": (I, [VP, VP, VP, I]),
    "x3d_pfft_fwd_c_part": (I, [VP, VP, I]),
    "x3d_pfft_bwd_c_part": (I, [VP, VP, I]),
    "x3d_pfft_bwd_b_part": (I, [VP, VP, VP, I]),
    "x3d_pfft_bwd_a_part": (I, [VP, VP, VP, I]),
    "x3d_timer_start": (I, [VP]),
    "x3d_timer_stop_ms": (I, [VP, ctypes.POINTER(ctypes.c_float)]),
    "x3d_prof_enable": (I, [VP, I]),
    "x3d_prof_select": (I, [VP, ctypes.c_uint]),
    "x3d_prof_reset": (I, [VP]),
    "x3d_prof_get": (I, [VP, I, I, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def build(verbose=False):
    """hipcc-compile every HIP source for gfx950 into x3d2_amd/libx3d2_hip.so
    (cross-compiles without a GPU)."""
    for flavour in ([], ["SP=1"]):  # FP64 (libx3d2_hip.so) and FP32 (libx3d2_hip_sp.so: -DX3D_SINGLE_PREC)
        r = subprocess.run(["make", "-C", CSRC, "-j4"] + flavour, capture_output=True, text=True)
        if verbose or r.returncode != 0:
            print(r.stdout[-4000:], r.stderr[-4000:])
        if r.returncode != 0:
            raise X3dError("building libx3d2_hip%s.so failed" % ("_sp" if flavour else ""))
    return LIB_PATH


def load():
    """dlopen the HIP backend; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its wheel bundles its own HIP runtime and must be the one the
    # process initialises (loading /opt/rocm's copy first leaves two runtimes)
    import torch  # noqa: F401
    # rocFFT compiles its kernels at plan creation (seconds per new transform length on a fresh box); a cache file that
    # travels with the tree (built artefact, git-ignored like the .so) spares every process that compilation
    os.environ.setdefault("ROCFFT_RTC_CACHE_PATH", os.path.join(HERE, "rocfft_rtc_cache.db"))
    if not os.path.exists(LIB_PATH):
        raise X3dError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                       " (the HIP backend has no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing from the .so
        fn.restype = res
        fn.argtypes = args
    want = 4 if SINGLE else 8
    if lib.x3d_real_bytes() != want:  # (a stale or misnamed build: every pointer below would be misread)
        raise X3dError(f"{LIB_PATH} computes in {lib.x3d_real_bytes()}-byte reals, this process expects {want}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise X3dError(load().x3d_last_error().decode())


def ints(*v):
    return (ctypes.c_int * len(v))(*[int(x) for x in v])
