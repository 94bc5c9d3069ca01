"""ctypes binding of the C ABI declared in include/qgcm_hip.h."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
MAXL = 8
ABI_VERSION = 3  # QGCM_HIP_ABI_VERSION of include/qgcm_hip.h


class QgcmHipError(RuntimeError):
    pass


def library_path():
    """In-tree build; QGCM_HIP_LIB overrides it (kernel experiments / ablation builds)."""
    return os.environ.get("QGCM_HIP_LIB") or os.path.normpath(os.path.join(HERE, "..", "..", "lib", "libqgcm_hip.so"))


class Params(C.Structure):
    """struct qgcm_hip_params (include/qgcm_hip.h)."""
    _fields_ = [
        ("nxpo", C.c_int), ("nypo", C.c_int), ("nlo", C.c_int), ("cyclic", C.c_int),
        ("fnot", C.c_double), ("beta", C.c_double), ("dxo", C.c_double), ("dyo", C.c_double),
        ("tdto", C.c_double), ("delek", C.c_double), ("bccooc", C.c_double),
        ("ah2oc", C.c_double * MAXL), ("ah4oc", C.c_double * MAXL),
        ("hoc", C.c_double * MAXL), ("gpoc", C.c_double * MAXL),
        ("amatoc", C.c_double * (MAXL * MAXL)), ("ctl2moc", C.c_double * (MAXL * MAXL)),
        ("ctm2loc", C.c_double * (MAXL * MAXL)), ("rdm2oc", C.c_double * MAXL),
        ("aoc", C.c_double),
        ("slab_g0", C.c_int), ("slab_g1", C.c_int),
        ("atmos", C.c_int),
    ]


class OmlParams(C.Structure):
    """struct qgcm_hip_oml_params (include/qgcm_hip.h)."""
    _fields_ = [
        ("hmoc", C.c_double), ("toc1", C.c_double), ("toc2", C.c_double), ("st2d", C.c_double), ("st4d", C.c_double),
        ("ycexp", C.c_double), ("rrcpoc", C.c_double), ("tsbdy", C.c_double), ("tnbdy", C.c_double),
        ("sb_hflux", C.c_int), ("nb_hflux", C.c_int),
    ]


# every symbol include/qgcm_hip.h declares
SYMBOLS = [
    "qgcm_hip_create", "qgcm_hip_destroy", "qgcm_hip_last_error", "qgcm_hip_abi_version",
    "qgcm_hip_set_grid", "qgcm_hip_set_geometry", "qgcm_hip_set_homog_box", "qgcm_hip_set_homog_cyc",
    "qgcm_hip_set_state", "qgcm_hip_get_state", "qgcm_hip_set_forcing", "qgcm_hip_set_cyc_forcing", "qgcm_hip_set_sponge",
    "qgcm_hip_set_scalars", "qgcm_hip_get_scalars", "qgcm_hip_get_inv_diag", "qgcm_hip_get_monitors",
    "qgcm_hip_qgostep", "qgcm_hip_ocinvq", "qgcm_hip_ocqbdy", "qgcm_hip_lf_average", "qgcm_hip_ocqbdy_host",
    "qgcm_hip_steps", "qgcm_hip_sync", "qgcm_hip_helmholtz",
    "qgcm_hip_qgastep", "qgcm_hip_atinvq", "qgcm_hip_atqzbd", "qgcm_hip_get_bsums", "qgcm_hip_coupled_steps", "qgcm_hip_set_cu_range",
    "qgcm_hip_wrk_fill", "qgcm_hip_wrk_get", "qgcm_hip_wrk_set", "qgcm_hip_area_integrals",
    "qgcm_hip_local_rows", "qgcm_hip_row_transform", "qgcm_hip_thomas_msg_len", "qgcm_hip_thomas_phase",
    "qgcm_hip_thomas_const_len", "qgcm_hip_thomas_consts", "qgcm_hip_set_thomas_consts",
    "qgcm_hip_constr", "qgcm_hip_unpack",
    "qgcm_hip_halo_msg_len", "qgcm_hip_halo_pack", "qgcm_hip_halo_unpack", "qgcm_hip_slab_stage", "qgcm_hip_oml_msg_len",
    "qgcm_hip_comm_unique_id", "qgcm_hip_comm_init", "qgcm_hip_slab_steps", "qgcm_hip_comm_set_halo_p2p", "qgcm_hip_comm_set_overlap", "qgcm_hip_comm_probe",
    "qgcm_hip_oml_init", "qgcm_hip_oml_set_state", "qgcm_hip_oml_get_state", "qgcm_hip_oml_set_forcing",
    "qgcm_hip_oml", "qgcm_hip_oml_get_diag", "qgcm_hip_set_dtopoc", "qgcm_hip_valids",
    "qgcm_hip_init_from_p", "qgcm_hip_wekpo_from_tau", "qgcm_hip_prsamp",
    "qgcm_hip_time_steps", "qgcm_hip_prepare_steps", "qgcm_hip_profile_steps", "qgcm_hip_copy_bandwidth", "qgcm_hip_stream_mix_bandwidth", "qgcm_hip_stream",
]

_lib = None


def load_library():
    """Load libqgcm_hip.so; raises QgcmHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise QgcmHipError("HIP library %s is missing - run __graft_entry__.build() "
                           "(make -C q-gcm_amd/csrc); there is no CPU fallback" % path)
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.qgcm_hip_last_error.restype = C.c_char_p
    if L.qgcm_hip_abi_version() != ABI_VERSION:
        raise QgcmHipError("%s has ABI version %d, this package binds version %d (include/qgcm_hip.h) - rebuild it"
                           % (path, L.qgcm_hip_abi_version(), ABI_VERSION))
    L.qgcm_hip_create.argtypes = [C.POINTER(vp), C.POINTER(Params), C.c_int]
    L.qgcm_hip_destroy.argtypes = [vp]
    L.qgcm_hip_set_grid.argtypes = [vp, dp, dp, dp]
    L.qgcm_hip_set_geometry.argtypes = [vp, dp, dp]
    L.qgcm_hip_set_homog_box.argtypes = [vp, dp, dp, dp]
    L.qgcm_hip_set_homog_cyc.argtypes = [vp] + [dp] * 8 + [C.c_double, C.c_double]
    L.qgcm_hip_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.qgcm_hip_get_state.argtypes = [vp, dp, dp, dp, dp]
    L.qgcm_hip_set_forcing.argtypes = [vp, dp, dp, dp]
    L.qgcm_hip_set_cyc_forcing.argtypes = [vp, C.c_double, C.c_double, dp, dp]
    L.qgcm_hip_set_sponge.argtypes = [vp, dp, C.c_double]
    L.qgcm_hip_set_scalars.argtypes = [vp, dp]
    L.qgcm_hip_get_scalars.argtypes = [vp, dp]
    L.qgcm_hip_get_inv_diag.argtypes = [vp, dp, dp]
    L.qgcm_hip_get_monitors.argtypes = [vp, dp, dp]
    for n in ("qgcm_hip_qgostep", "qgcm_hip_ocinvq", "qgcm_hip_ocqbdy", "qgcm_hip_lf_average", "qgcm_hip_sync"):
        getattr(L, n).argtypes = [vp]
    for n in ("qgcm_hip_qgastep", "qgcm_hip_atinvq", "qgcm_hip_atqzbd"):
        getattr(L, n).argtypes = [vp]
    L.qgcm_hip_get_bsums.argtypes = [vp, dp]
    L.qgcm_hip_coupled_steps.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int]
    L.qgcm_hip_set_cu_range.argtypes = [vp, C.c_int, C.c_int]
    L.qgcm_hip_ocqbdy_host.argtypes = [vp, dp, dp]
    L.qgcm_hip_steps.argtypes = [vp, C.c_int, C.c_int]
    L.qgcm_hip_helmholtz.argtypes = [vp, dp, dp]
    ip = C.POINTER(C.c_int)
    L.qgcm_hip_local_rows.argtypes = [vp, ip, ip, ip, ip]
    L.qgcm_hip_row_transform.argtypes = [vp, C.c_int]
    L.qgcm_hip_wrk_fill.argtypes = [vp, C.c_double]
    L.qgcm_hip_wrk_get.argtypes = [vp, dp]
    L.qgcm_hip_wrk_set.argtypes = [vp, dp]
    L.qgcm_hip_area_integrals.argtypes = [vp, dp]
    L.qgcm_hip_thomas_msg_len.argtypes = [vp]
    L.qgcm_hip_thomas_const_len.argtypes = [vp]
    L.qgcm_hip_thomas_consts.argtypes = [vp, vp]
    L.qgcm_hip_set_thomas_consts.argtypes = [vp, vp, C.c_int]
    L.qgcm_hip_thomas_phase.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int]
    L.qgcm_hip_constr.argtypes = [vp]
    L.qgcm_hip_unpack.argtypes = [vp, C.c_int]
    L.qgcm_hip_halo_msg_len.argtypes = [vp]
    L.qgcm_hip_oml_msg_len.argtypes = [vp]
    L.qgcm_hip_halo_pack.argtypes = [vp, vp, vp]
    L.qgcm_hip_halo_unpack.argtypes = [vp, vp, vp]
    L.qgcm_hip_slab_stage.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.qgcm_hip_comm_unique_id.argtypes = [C.c_char_p, C.c_int]
    L.qgcm_hip_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.qgcm_hip_slab_steps.argtypes = [vp, C.c_int, C.c_int]
    L.qgcm_hip_comm_set_halo_p2p.argtypes = [vp, C.c_int]
    L.qgcm_hip_comm_set_overlap.argtypes = [vp, C.c_int]
    L.qgcm_hip_comm_probe.argtypes = [vp, C.c_int, dp]
    L.qgcm_hip_oml_init.argtypes = [vp, C.POINTER(OmlParams)]
    L.qgcm_hip_oml_set_state.argtypes = [vp, dp, dp]
    L.qgcm_hip_oml_get_state.argtypes = [vp, dp, dp]
    L.qgcm_hip_oml_set_forcing.argtypes = [vp, dp, dp, dp, dp]
    L.qgcm_hip_oml.argtypes = [vp]
    L.qgcm_hip_oml_get_diag.argtypes = [vp, dp, dp]
    L.qgcm_hip_set_dtopoc.argtypes = [vp, dp]
    L.qgcm_hip_valids.argtypes = [vp, dp, C.POINTER(C.c_int)]
    L.qgcm_hip_init_from_p.argtypes = [vp]
    L.qgcm_hip_wekpo_from_tau.argtypes = [vp, dp, dp]
    L.qgcm_hip_prsamp.argtypes = [vp, dp]
    L.qgcm_hip_time_steps.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.qgcm_hip_prepare_steps.argtypes = [vp, C.c_int, C.c_int]
    L.qgcm_hip_profile_steps.argtypes = [vp, C.c_int, C.c_int, dp, C.POINTER(C.c_int),
                                         C.POINTER(C.c_char_p), C.POINTER(C.c_int)]
    L.qgcm_hip_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, dp]
    L.qgcm_hip_stream_mix_bandwidth.argtypes = [vp, C.c_int, C.c_int, C.c_size_t, C.c_int, dp]
    L.qgcm_hip_stream.argtypes = [vp]
    L.qgcm_hip_stream.restype = vp
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise QgcmHipError(load_library().qgcm_hip_last_error().decode())
