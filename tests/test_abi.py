"""The C-ABI library loads and exports every symbol include/qgcm_hip.h declares.
No compute calls (CPU only)."""
import ctypes
import os
import re

import pytest

from qgcm_hip import lib


def header_symbols(root):
    txt = open(os.path.join(root, "include", "qgcm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qgcm_hip_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(repo_root):
    assert header_symbols(repo_root) == sorted(lib.SYMBOLS)


def test_library_exports_all_symbols(repo_root):
    path = lib.library_path()
    if not os.path.exists(path):
        pytest.fail("libqgcm_hip.so not built - run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = ctypes.CDLL(path)
    for s in header_symbols(repo_root):
        assert hasattr(L, s), s
    L.qgcm_hip_abi_version.restype = ctypes.c_int
    assert L.qgcm_hip_abi_version() == 2  # 2: qgcm_hip_params.atmos + the atmosphere entry points


def test_params_struct_layout():
    # 4 ints + 7 doubles + 4*MAXL + 3*MAXL^2 + MAXL + 1 doubles + 3 ints (slab_g0, slab_g1, atmos) + 4 B tail padding
    n = lib.MAXL
    assert ctypes.sizeof(lib.Params) == 4 * 4 + 8 * (7 + 4 * n + 3 * n * n + n + 1) + 3 * 4 + 4
    assert lib.Params.atmos.offset == ctypes.sizeof(lib.Params) - 8


def test_oml_params_struct_layout(repo_root):
    # 9 doubles + 2 ints, field order as in include/qgcm_hip.h
    assert ctypes.sizeof(lib.OmlParams) == 9 * 8 + 2 * 4
    hdr = open(os.path.join(repo_root, "include", "qgcm_hip.h")).read()
    body = hdr[hdr.index("typedef struct qgcm_hip_oml_params {"):hdr.index("} qgcm_hip_oml_params;")]
    pos = [body.index(" %s" % f[0]) for f in lib.OmlParams._fields_]
    assert pos == sorted(pos)


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device the constructor raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qgcm_hip import OceanModel, QgcmHipError, preset
    with pytest.raises(QgcmHipError):
        OceanModel(preset("box_tiny"))
