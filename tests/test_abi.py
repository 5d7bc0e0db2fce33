"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/x3d2_hip.h declares; the ctypes prototypes cover exactly that set.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "x3d2_hip.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(x3d_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_declares_the_operator_interface():
    names = declared_functions()
    # one entry point per deferred procedure of base_backend_t used on the hot path
    for required in ("x3d_transeq", "x3d_tds_solve", "x3d_reorder", "x3d_sum_intox", "x3d_veccopy",
                     "x3d_vecadd", "x3d_vecmult", "x3d_scalar_product", "x3d_field_max_sum",
                     "x3d_slice_max_sum", "x3d_field_scale", "x3d_field_shift", "x3d_field_volume_integral",
                     "x3d_field_set_face", "x3d_field_set_face_from_field", "x3d_set_field_data",
                     "x3d_get_field_data", "x3d_tdsops_create", "x3d_poisson_create",
                     "x3d_poisson_fft_forward", "x3d_poisson_postprocess_000", "x3d_poisson_fft_backward"):
        assert required in names, required


def test_library_exports_every_declared_symbol():
    from x3d2_amd import _lib
    lib = _lib.load()  # built by __graft_entry__.build(); loading needs no GPU
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in x3d2_hip.h but missing from libx3d2_hip.so"
    assert lib.x3d_abi_version() == 1


def test_single_precision_flavour_exports_the_same_symbols():
    """libx3d2_hip_sp.so (make SP=1: -DX3D_SINGLE_PREC, x3d_real = float) is the same ABI on 4-byte reals"""
    from x3d2_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libx3d2_hip_sp.so")
    assert os.path.exists(path), "build() compiles both flavours"
    import torch  # noqa: F401  (its HIP runtime first, as _lib.load does)
    lib = ctypes.CDLL(path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} missing from libx3d2_hip_sp.so"


def test_ctypes_prototypes_match_header():
    from x3d2_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_functions()


def test_no_compute_without_gpu_fails_loudly():
    """the product has no CPU fallback: creating a backend without a HIP device is an error"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from x3d2_amd.backend import HipBackend
    from x3d2_amd.common import X3dError
    from x3d2_amd.mesh import Mesh
    mesh = Mesh((16, 16, 16), (1, 1, 1), (1.0,) * 3, ("periodic",) * 2, ("periodic",) * 2, ("periodic",) * 2)
    with pytest.raises(X3dError):
        HipBackend(mesh)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under x3d2_amd/ may import, link or call it"""
    pkg = os.path.join(ROOT, "x3d2_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|x3d_oracle|libx3d_oracle)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), f"{f} references the oracle"
