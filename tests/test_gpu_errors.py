"""Error behaviour of the C ABI on the GPU box: every misuse returns a non-zero status with a message in
qgcm_hip_last_error() (the Python host raises QgcmHipError, the Fortran shim prints it and stops - the reference's
own convention, e.g. src/ocisubs.F:361-365); nothing falls back to another code path."""
import ctypes as C

import numpy as np
import pytest

from qgcm_hip import OceanModel, QgcmHipError, preset
from qgcm_hip.lib import Params, check, load_library

pytestmark = pytest.mark.gpu


def _params(cfg):
    m = OceanModel(cfg)
    p = Params.from_buffer_copy(m.params)
    m.close()
    return p


def test_unsupported_layer_counts_and_sizes():
    L = load_library()
    p = _params(preset("box_tiny"))
    h = C.c_void_p()
    p.nlo = 9      # the ABI's arrays hold QGCM_HIP_MAXL = 8 layers
    with pytest.raises(QgcmHipError, match="nlo"):
        check(L.qgcm_hip_create(C.byref(h), C.byref(p), -1))
    p.nlo = 1
    with pytest.raises(QgcmHipError, match="nlo"):
        check(L.qgcm_hip_create(C.byref(h), C.byref(p), -1))
    p.nlo = 3
    p.nxpo = 2
    with pytest.raises(QgcmHipError, match="grid too small"):
        check(L.qgcm_hip_create(C.byref(h), C.byref(p), -1))
    assert not h.value


def test_calls_out_of_order():
    L = load_library()
    p = _params(preset("box_tiny"))
    h = C.c_void_p()
    check(L.qgcm_hip_create(C.byref(h), C.byref(p), -1))
    try:
        with pytest.raises(QgcmHipError, match="qgcm_hip_set_grid has not been called"):
            check(L.qgcm_hip_qgostep(h))
        with pytest.raises(QgcmHipError, match="qgcm_hip_set_grid has not been called"):
            check(L.qgcm_hip_steps(h, 1, 1))
    finally:
        L.qgcm_hip_destroy(h)
    with pytest.raises(QgcmHipError, match="null handle"):
        check(L.qgcm_hip_qgostep(None))


def test_homogeneous_solutions_required_and_bad_ranges():
    cfg = preset("box_tiny")
    L = load_library()
    p = _params(cfg)
    h = C.c_void_p()
    check(L.qgcm_hip_create(C.byref(h), C.byref(p), -1))
    try:
        yp = np.ascontiguousarray(cfg.yporel())
        from qgcm_hip import hostinit
        _, bd2 = hostinit.bd2oc(cfg)
        dp = C.POINTER(C.c_double)
        check(L.qgcm_hip_set_grid(h, yp.ctypes.data_as(dp), np.ascontiguousarray(bd2).ctypes.data_as(dp), None))
        check(L.qgcm_hip_qgostep(h))                       # the tendency needs no homogeneous solutions
        with pytest.raises(QgcmHipError, match="homogeneous solutions not set"):
            check(L.qgcm_hip_ocinvq(h))
    finally:
        L.qgcm_hip_destroy(h)
    m = OceanModel(cfg)
    try:
        with pytest.raises(QgcmHipError, match="bad step range"):
            m.steps(3, s0=0)
        with pytest.raises(QgcmHipError, match="qgcm_hip_comm_init"):
            check(m.L.qgcm_hip_slab_steps(m.h, 1, 1))
    finally:
        m.close()


def test_slab_handles_refuse_whole_domain_entry_points():
    from qgcm_hip.slab import HipSlab, global_consts, partition
    from common import make_oracle
    cfg = preset("box_small")
    o = make_oracle(cfg)
    try:
        consts = global_consts(cfg, o.helmholtz)
    finally:
        o.close()
    (g0, g1), _ = partition(cfg.nypo, 2)
    sl = HipSlab(cfg, consts, g0, g1, 0, 2)
    try:
        with pytest.raises(QgcmHipError, match="y-slab"):
            check(sl.L.qgcm_hip_ocinvq(sl.h))
        with pytest.raises(QgcmHipError, match="y-slab"):
            check(sl.L.qgcm_hip_steps(sl.h, 1, 1))
        with pytest.raises(QgcmHipError, match="whole domain"):
            check(sl.L.qgcm_hip_valids(sl.h, None, None))
        with pytest.raises(QgcmHipError, match="stage must be 1..7"):
            sl.stage(8)
    finally:
        sl.close()
    # the split tendency launch (stages 4 / 5) needs three 16-row tile rows: a 20-row slab refuses
    (g0, g1) = partition(cfg.nypo, 4)[1]
    thin = HipSlab(cfg, consts, g0, g1, 1, 4)
    try:
        with pytest.raises(QgcmHipError, match="at least three tile rows"):
            thin.stage(4)
    finally:
        thin.close()


def test_mixed_layer_on_slabs_misuse():
    """y-slab handles: the one-call `oml` is for whole-domain handles, the slab stages 10 / 11 need qgcm_hip_oml_init,
    and the mixed layer must be switched on before the communicator sizes the messages."""
    import torch
    from qgcm_hip import oml_preset
    from qgcm_hip.slab import HipSlab, global_consts, partition, rccl_unique_id
    cfg = preset("box_small")
    consts = global_consts(cfg, lambda rhs, boc: np.zeros_like(rhs))  # any homogeneous solutions will do here
    (g0, g1), _ = partition(cfg.nypo, 2)
    sl = HipSlab(cfg, consts, g0, g1, 0, 2)
    try:
        buf = sl.new_buffer(64)
        torch.cuda.synchronize()
        with pytest.raises(QgcmHipError, match="qgcm_hip_oml_init has not been called"):
            sl.stage(10, buf)
        n0, h0 = sl.th_len, sl.halo_len
        sl.oml_init(oml_preset(cfg))
        assert sl.th_len == n0 + 3 and sl.halo_len == h0 + 3 * ((cfg.nxto + 15) // 16 * 16)
        with pytest.raises(QgcmHipError, match="y-slab"):
            check(sl.L.qgcm_hip_oml(sl.h))
    finally:
        sl.close()
    whole = HipSlab(cfg, consts, 1, cfg.nypo, 0, 1)
    try:
        whole.comm_init(rccl_unique_id())
        with pytest.raises(QgcmHipError, match="before qgcm_hip_comm_init"):
            whole.oml_init(oml_preset(cfg))
    finally:
        whole.close()
