"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
include/pea_hip.h declares (no compute call is made without a GPU), and the product path refuses
to run without the HIP extension or a device (no CPU fallback)."""
import ctypes
import os

import pytest

from pea_diffusion_amd import _lib


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_parses_and_all_symbols_exported(built):
    protos = _lib.parse_header()
    assert len(protos) >= 25
    for name in protos:
        assert hasattr(built, name), f"libpea_hip.so does not export {name}"
    for must in ["pea_op_gemm", "pea_op_conv3x3", "pea_op_attention_fwd", "pea_op_attention_bwd", "pea_op_kd_loss",
                 "pea_op_groupnorm_fwd", "pea_op_layernorm_bwd", "pea_last_error"]:
        assert must in protos


def test_library_has_gfx950_code_object():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"gemm_lcp_kernel" in data


def test_version_and_error_string(built):
    L = _lib.lib()
    assert L.pea_version() >= 100
    assert isinstance(L.pea_last_error(), bytes)


def test_shape_errors_are_reported_without_gpu():
    L = _lib.lib()
    # K not a multiple of 64 is rejected before any device work
    rc = L.pea_op_gemm(None, 8, None, 8, None, 8, 4, 4, 10, 1.0, None, None, 0, 1, 0, None, 0, None, 0, 0, 0, None)
    assert rc == -3 and b"K=10" in L.pea_last_error()
