"""CPU tests of the drop-in boundary: include/vo_hip.h, the built library and the product package.
No compute calls (there is no GPU here); on a GPU box the same checks run plus the loud-failure one."""
import ctypes as C
import pathlib
import re
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _header_symbols():
    text = (ROOT / "include" / "vo_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vo_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree(vo):
    assert _header_symbols() == sorted(vo.SYMBOLS)


def test_library_exports_every_declared_symbol(vo):
    lib = vo.lib()
    missing = [s for s in _header_symbols() if not hasattr(lib, s)]
    assert not missing
    out = subprocess.run(["nm", "-D", "--defined-only", str(vo.SO)], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (vo_[a-z0-9_]+)", out))
    assert set(_header_symbols()) <= exported
    assert exported <= set(_header_symbols()), f"exported but not declared in include/vo_hip.h: {sorted(exported - set(_header_symbols()))}"
    assert lib.vo_version().decode().startswith("vo_slam_test_amd")


def test_keypoint_layout_matches_cv_keypoint(vo):
    assert vo.KP_DTYPE.itemsize == 28
    assert [vo.KP_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]
    assert C.sizeof(vo.LmSummary) == 40


def test_fails_loudly_without_gpu(vo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = vo.lib().vo_orb_create(C.byref(h), 1000, C.c_float(1.2), 8, 20, 7)
    assert rc == -2 and b"no CPU fallback" in vo.lib().vo_last_error()
    with pytest.raises(vo.VoError):
        vo.OrbExtractor()
    with pytest.raises(vo.VoError):
        vo.hamming_matrix(np.zeros((2, 32), np.uint8), np.zeros((2, 32), np.uint8))


def test_invalid_arguments_are_rejected(vo):
    h = C.c_void_p()
    assert vo.lib().vo_orb_create(C.byref(h), 0, C.c_float(1.2), 8, 20, 7) == -1
    assert vo.lib().vo_orb_create(C.byref(h), 1000, C.c_float(1.0), 8, 20, 7) == -1
    assert vo.lib().vo_orb_create(C.byref(h), 1000, C.c_float(1.2), 17, 20, 7) == -1
    assert vo.lib().vo_hamming_matrix(None, -1, None, 0, None) == -1
    assert vo.lib().vo_hamming_matrix(None, 0, None, 0, None) == 0


def test_product_never_touches_the_oracle_or_reference():
    """the HIP path must not import, link or execute anything under oracle/ (or read /root/reference)"""
    for f in list((ROOT / "vo_slam_test_amd").rglob("*.py")) + list((ROOT / "vo_slam_test_amd" / "csrc").glob("*")) + \
            list((ROOT / "include").rglob("*.h")):
        if f.suffix in (".so", ".o"):
            continue
        t = f.read_text(errors="ignore")
        assert "oracle_lib" not in t and "liboracle" not in t and "oracle.h" not in t, f
        assert "/root/reference" not in t, f
    out = subprocess.run(["ldd", str(ROOT / "vo_slam_test_amd" / "libvo_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
