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
    assert L.qgcm_hip_abi_version() == lib.ABI_VERSION == 3  # changelog: include/qgcm_hip.h
    hdr = open(os.path.join(repo_root, "include", "qgcm_hip.h")).read()
    assert "#define QGCM_HIP_ABI_VERSION %d" % lib.ABI_VERSION in hdr


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


def test_hot_kernels_do_not_spill(repo_root):
    """The compiler's resource report written next to the library at build time (q-gcm_amd/csrc/Makefile): the kernels
    of the 5 km step, of SOcn 5 km and of the atmospheric channel use no scratch.  k_thomas<16,0> sits at its 128-VGPR
    limit (1024-thread workgroups) and has been pushed into scratch by innocent-looking edits more than once - at 5 km
    that is +13 MB of traffic per launch and a visibly slower step."""
    path = os.path.join(repo_root, "q-gcm_amd", "lib", "kernel_resources.txt")
    if not os.path.exists(path):
        pytest.fail("kernel_resources.txt missing - rebuild with `make -C q-gcm_amd/csrc`")
    res, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            res[cur] = int(m.group(1))
    hot = ["_Z6k_tendILi3ELb0ELb", "_Z6k_tendILi3ELb1ELb", "_Z7k_dst64ILi15ELb0EEv", "_Z8k_thomasILi16ELi0ELb0E",
           "_Z8k_thomasILi10ELi0ELb1E", "_Z8k_thomasILi2ELi0ELb1E", "_Z8k_thomasILi20ELi", "_Z8k_thomasILi32ELi", "_Z14k_dst64_unpackILi15ELi3ELb1ELb0ELb1ELb",
           "_Z8k_rfft64ILi6ELb0EEv", "_Z15k_rfft64_unpackILi6ELi3ELb1ELb1EEv", "_Z10k_rfft_cycILb0E8Fft3PlanILi16ELi16ELi18EELi256EEv",
           "_Z10k_rfft_cycILb1E8Fft3PlanILi16ELi16ELi18EELi256EEv", "_Z14k_rfft3_unpackI8Fft3PlanILi16ELi16ELi18EELi3ELi256ELb", "_Z9k_dst_boxILb0ELi256E8Fft3PlanILi12ELi20ELi20EEEv", "_Z13k_thomas_corr", "_Z8k_thomasILi10ELi1ELb0E",
           "_Z10k_oml_step11QgOmlParams", "_Z11k_oml_entoc11QgOmlParams"]
    for h in hot:
        hits = [k for k in res if k.startswith(h)]
        assert hits, "kernel %s not in the report" % h
        for k in hits:
            assert res[k] == 0, "%s spills %d B per lane" % (k, res[k])
