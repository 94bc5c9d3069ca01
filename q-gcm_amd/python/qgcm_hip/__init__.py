"""qgcm_hip - host side of the MI355X-native Q-GCM ocean PV-advance / inversion path.

The compute path is the C-ABI HIP library ``q-gcm_amd/lib/libqgcm_hip.so``
(include/qgcm_hip.h).  This package is the Python mirror of the reference's
operator interface for that path (``qgostep`` / ``ocinvq`` / ``ocqbdy`` acting
on the ``ocstate`` fields) plus the start-up arithmetic the reference main
program performs before the time loop.  There is no CPU fallback: importing
works anywhere, but constructing an :class:`OceanModel` without the built
library or without a HIP device raises.
"""
from .config import AtmosConfig, OceanConfig, OmlConfig, PRESETS, atmos_of, atmos_preset, oml_preset, preset  # noqa: F401
from .lib import QgcmHipError, check, load_library, library_path  # noqa: F401
from .model import AtmosModel, OceanModel, coupled_steps, share_gpu  # noqa: F401
from . import hostinit, synth  # noqa: F401
