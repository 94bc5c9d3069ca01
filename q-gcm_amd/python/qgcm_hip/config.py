"""Grid + physical parameters of one ocean configuration.

Mirrors what the reference takes from MODULE parameters (compile-time,
src/parameters_data.F:23-147) and from input.params (run-time, order fixed by
src/in_param.f:31-142); only the entries the ocean hot path reads are kept.
"""
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np


@dataclass(frozen=True)
class OceanConfig:
    name: str
    nxta: int
    nyta: int
    nxaooc: int
    nyaooc: int
    ndxr: int
    nlo: int
    fnot: float
    beta: float
    cyclic: bool
    dxo: float
    dta: float = 180.0
    nstr: int = 3
    delek: float = 2.0
    bccooc: float = 0.2
    ah2oc: Tuple[float, ...] = (0.0, 0.0, 0.0)
    ah4oc: Tuple[float, ...] = (2.0e9, 2.0e9, 2.0e9)
    hoc: Tuple[float, ...] = (350.0, 750.0, 2900.0)
    gpoc: Tuple[float, ...] = (0.015, 0.0075)
    # the fork's sponge layer (cpp option sponge_layer_k247; constants of src/parameters_data.F:140-144):
    # l_spl = 0 means the option is not defined (every BASELINE configuration)
    c1_spl: float = 0.0
    l_spl: float = 0.0
    spl_ns_only: bool = False  # cpp option nospl_in_ewbdy_k247: ramp at the zonal (N / S) boundaries only

    # derived grid parameters, src/parameters_data.F (nxto = ndxr*nxaooc, ...)
    @property
    def nxto(self):
        return self.ndxr * self.nxaooc

    @property
    def nyto(self):
        return self.ndxr * self.nyaooc

    @property
    def nxpo(self):
        return self.nxto + 1

    @property
    def nypo(self):
        return self.nyto + 1

    @property
    def dto(self):  # src/q-gcm.F:381
        return self.nstr * self.dta

    @property
    def tdto(self):  # src/q-gcm.F:440
        return 2.0 * self.dto

    @property
    def dyo(self):  # src/q-gcm.F:413
        return self.dxo

    @property
    def xlo(self):
        return self.nxto * self.dxo

    @property
    def ylo(self):
        return self.nyto * self.dyo

    def yporel(self):
        """src/q-gcm.F:380-427 - depends on the *atmosphere* dims even ocean-only."""
        dxa = self.ndxr * self.dxo
        dya = dxa
        yla = self.nyta * dya
        ny1 = 1 + (self.nyta - self.nyaooc) // 2
        ypo = (ny1 - 1) * dya + np.arange(self.nypo, dtype=np.float64) * self.dyo
        return ypo - 0.5 * yla

    def model_years_per_day(self, steps_per_s):
        """steps/s * dto / 365  (secsyr = 86400*365, src/timinfo_data.F:33-35)."""
        return steps_per_s * self.dto / 365.0


@dataclass(frozen=True)
class OmlConfig:
    """Run-time parameters of the ocean mixed layer (oml / omladf, src/omlsubs.F): input.params entries
    hmoc, st2d, st4d, ycexp, rhooc, cpoc (src/in_param.f), layer temperatures toc(1:2) (MODULE occonst) and
    the reference's compile-time boundary options sb_hflux / nb_hflux with tsbdy / tnbdy (MODULE intrfac)."""
    hmoc: float = 100.0
    toc: Tuple[float, float] = (15.0, 10.0)
    st2d: float = 100.0
    st4d: float = 2.0e9
    ycexp: float = 1.0
    rhooc: float = 1.0e3
    cpoc: float = 4.0e3
    sb_hflux: bool = False
    tsbdy: float = 0.0
    nb_hflux: bool = False
    tnbdy: float = 0.0

    @property
    def rrcpoc(self):  # src/q-gcm.F:438
        return 1.0 / (self.rhooc * self.cpoc)


def oml_preset(cfg, sb_hflux=False, nb_hflux=False):
    """Mixed-layer parameters of the examples (input.params.dg_oo:46,66-70) with the diffusivities
    scaled from the 5 km grid to cfg.dxo (same grid-scale damping rate)."""
    r = cfg.dxo / 5.0e3
    return OmlConfig(st2d=100.0 * r * r, st4d=2.0e9 * r ** 4, sb_hflux=sb_hflux, tsbdy=18.0 if sb_hflux else 0.0,
                     nb_hflux=nb_hflux, tnbdy=12.0 if nb_hflux else 0.0)


@dataclass(frozen=True)
class AtmosConfig:
    """The atmospheric channel of a coupled run: MODULE parameters nxta, nyta, nla, fnot, beta
    and the input.params entries dta, bccoat, ah4at, hat, gpat (src/in_param.f); dxa = ndxr*dxo (src/q-gcm.F:380).
    The ocean-named properties let the start-up arithmetic of hostinit (shared with the cyclic ocean) read it."""
    name: str
    nxta: int
    nyta: int
    fnot: float
    beta: float
    dxa: float
    nla: int = 3
    dta: float = 180.0
    bccoat: float = 1.0
    ah4at: Tuple[float, ...] = (1.5e14, 1.5e14, 1.5e14)
    hat: Tuple[float, ...] = (2000.0, 3000.0, 4000.0)
    gpat: Tuple[float, ...] = (1.2, 0.4)
    atmos = True
    cyclic = True
    delek = 0.0

    nxpa = property(lambda self: self.nxta + 1)
    nypa = property(lambda self: self.nyta + 1)
    tdta = property(lambda self: 2.0 * self.dta)  # src/q-gcm.F:441
    dya = property(lambda self: self.dxa)         # src/q-gcm.F:391
    xla = property(lambda self: self.nxta * self.dxa)
    yla = property(lambda self: self.nyta * self.dxa)
    # the same quantities under the names the ocean code uses
    nxto = property(lambda self: self.nxta)
    nyto = property(lambda self: self.nyta)
    nxpo = property(lambda self: self.nxta + 1)
    nypo = property(lambda self: self.nyta + 1)
    nlo = property(lambda self: self.nla)
    dxo = property(lambda self: self.dxa)
    dyo = property(lambda self: self.dxa)
    dto = property(lambda self: self.dta)
    tdto = property(lambda self: 2.0 * self.dta)
    xlo = property(lambda self: self.nxta * self.dxa)
    ylo = property(lambda self: self.nyta * self.dxa)
    bccooc = property(lambda self: self.bccoat)
    ah2oc = property(lambda self: (0.0,) * self.nla)
    ah4oc = property(lambda self: self.ah4at)
    hoc = property(lambda self: self.hat)
    gpoc = property(lambda self: self.gpat)

    def ypa(self):
        """src/q-gcm.F:401-402"""
        return np.arange(self.nypa, dtype=np.float64) * self.dya

    def yporel(self):
        """yparel, src/q-gcm.F:403"""
        return self.ypa() - 0.5 * self.yla


def atmos_of(cfg, **kw):
    """The atmosphere that goes with an ocean configuration: same nxta, nyta, fnot, beta; dxa = ndxr*dxo."""
    return AtmosConfig("atm_" + cfg.name, cfg.nxta, cfg.nyta, cfg.fnot, cfg.beta, cfg.ndxr * cfg.dxo, dta=cfg.dta, **kw)


_NATL = dict(fnot=9.37456e-05, beta=1.75360e-11, cyclic=False)
_SOCN = dict(fnot=-1.19467e-04, beta=1.31301e-11, cyclic=True)

PRESETS = {
    # parity-test grids (same dimension sets as oracle/ref_binding.CONFIGS)
    "box_tiny": OceanConfig("box_tiny", 8, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                            ah4oc=(3.2e12,) * 3, **_NATL),
    "box_small": OceanConfig("box_small", 12, 10, 6, 5, 16, 3, dxo=5.0e4, dta=360.0,
                             ah4oc=(2.0e11,) * 3, **_NATL),
    "box_tiny2": OceanConfig("box_tiny2", 8, 8, 5, 4, 6, 2, fnot=5.92e-05, beta=2.08e-11, cyclic=False,
                             dxo=1.0e5, dta=720.0, ah2oc=(0.0, 0.0), ah4oc=(3.2e12, 3.2e12),
                             hoc=(500.0, 3500.0), gpoc=(0.02,)),
    # five and six layers (the kernels are templates on nlo = 2 .. 8; own reference builds box_tiny5 / cyc_tiny6)
    "box_tiny5": OceanConfig("box_tiny5", 8, 8, 4, 3, 12, 5, dxo=1.0e5, dta=720.0, ah2oc=(0.0,) * 5, ah4oc=(3.2e12,) * 5,
                             hoc=(300.0, 400.0, 600.0, 1000.0, 1700.0), gpoc=(0.02, 0.012, 0.008, 0.005), **_NATL),
    "cyc_tiny6": OceanConfig("cyc_tiny6", 4, 8, 4, 3, 12, 6, dxo=1.0e5, dta=720.0, ah2oc=(0.0,) * 6, ah4oc=(3.2e12,) * 6,
                             hoc=(250.0, 350.0, 500.0, 700.0, 1000.0, 1200.0), gpoc=(0.02, 0.015, 0.01, 0.007, 0.004), **_SOCN),
    # ... and compiled with -Dsponge_layer_k247 (src/qgosubs.F:203-205): own reference builds box_tiny_spl / cyc_tiny_spl
    "box_tiny_spl": OceanConfig("box_tiny_spl", 8, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                                ah4oc=(3.2e12,) * 3, c1_spl=-2.5e-5, l_spl=4.0e5, **_NATL),
    "cyc_tiny_spl": OceanConfig("cyc_tiny_spl", 4, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                                ah4oc=(3.2e12,) * 3, c1_spl=-2.5e-5, l_spl=4.0e5, spl_ns_only=True, **_SOCN),
    # nxto = 192 = 64*3: exercises the wave-per-row-pair DST kernel at a size the oracle runs in seconds
    "box_med": OceanConfig("box_med", 16, 10, 12, 6, 16, 3, dxo=2.5e4, dta=240.0,
                           ah4oc=(1.2e10,) * 3, **_NATL),
    "cyc_tiny": OceanConfig("cyc_tiny", 4, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                            ah4oc=(3.2e12,) * 3, **_SOCN),
    "cyc_small": OceanConfig("cyc_small", 6, 10, 6, 4, 16, 3, dxo=5.0e4, dta=360.0,
                             ah4oc=(2.0e11,) * 3, **_SOCN),
    # the tiny grids with the Laplacian-viscosity (Del-4th of p) term switched on, a different value per layer
    # (src/qgosubs.F:375-377; ah2oc = 0 in every example of the reference: this is the only pin of that branch and of
    # the cyclic ap3soc / ap3noc sums, src/qgosubs.F:429-443).  Same reference builds as box_tiny / cyc_tiny.
    "box_tiny_ah2": OceanConfig("box_tiny_ah2", 8, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                                ah2oc=(240.0, 160.0, 80.0), ah4oc=(3.2e12,) * 3, **_NATL),
    "cyc_tiny_ah2": OceanConfig("cyc_tiny_ah2", 4, 8, 4, 3, 12, 3, dxo=1.0e5, dta=720.0,
                                ah2oc=(240.0, 160.0, 80.0), ah4oc=(3.2e12,) * 3, **_SOCN),
    # nxto = 192 = 64*3 and 960 = 64*15: exercise the wave-per-row-pair real-FFT kernels (k_rfft64.h) of the cyclic path
    "cyc_med": OceanConfig("cyc_med", 12, 10, 12, 4, 16, 3, dxo=2.5e4, dta=240.0, ah4oc=(1.2e10,) * 3, **_SOCN),
    "cyc_960": OceanConfig("cyc_960", 60, 12, 60, 3, 16, 3, dxo=5.0e3, **_SOCN),
    # long rows with a three-stage plan (nxto = 2880 = 12 * 15 * 16) on few rows: the fused inverse rows + unpack launch
    "cyc_2880": OceanConfig("cyc_2880", 180, 12, 180, 3, 16, 3, dxo=5.0e3, **_SOCN),
    # examples/double_gyre_ocean_only: parameters_data.F.dg_oo + input.params.dg_oo (BASELINE configs[0], [1])
    "natl5": OceanConfig("natl5", 384, 96, 60, 60, 16, 3, dxo=5.0e3, **_NATL),
    # examples/southern_ocean_ocean_only (BASELINE configs[2])
    "socn5": OceanConfig("socn5", 288, 108, 288, 36, 16, 3, dxo=5.0e3, **_SOCN),
    # src/parameters_data.F.NAtl.1km + src/input.params.NAtl.1km with dta=60,nstr=3 (SURVEY 8d caveat)
    "natl1": OceanConfig("natl1", 384, 96, 60, 60, 80, 3, dxo=1.0e3, dta=60.0, bccooc=0.1,
                         ah4oc=(5.0e7,) * 3, **_NATL),
}


# coupled grids (oracle/ref_binding.CONFIGS "cpl_*"): the ocean half; atmos_preset gives the atmosphere.
# cpl_natl5 = examples/double_gyre_coupled (BASELINE configs[3]): NAtl 5 km ocean under a 385 x 97 x 3 atmosphere.
PRESETS["cpl_tiny"] = OceanConfig("cpl_tiny", 16, 12, 4, 3, 12, 3, dxo=1.0e4, dta=360.0, ah4oc=(3.2e10,) * 3, **_NATL)
PRESETS["cpl_small"] = OceanConfig("cpl_small", 32, 20, 6, 5, 16, 3, dxo=5.0e3, **_NATL)
PRESETS["cpl_natl5"] = OceanConfig("cpl_natl5", 384, 96, 60, 60, 16, 3, dxo=5.0e3, **_NATL)


def preset(name):
    return PRESETS[name]


def atmos_preset(name):
    """Atmosphere of the coupled preset `name`; Del-6th coefficient scaled from the 80 km grid of
    examples/double_gyre_coupled (ah4at = 1.5e14) to the grid spacing (same grid-scale damping rate per second)."""
    oc = PRESETS[name]
    r = oc.ndxr * oc.dxo / 8.0e4
    return atmos_of(oc, ah4at=(1.5e14 * r ** 4,) * 3)
