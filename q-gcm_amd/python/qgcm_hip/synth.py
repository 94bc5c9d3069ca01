"""Deterministic synthetic inputs (the reference's forcing blobs are not in the
checkout): double-gyre / channel wind stress, Gaussian-eddy initial
pressure, and the ocean-only Ekman pumping derived from the stress."""
import numpy as np


def wind_stress(cfg):
    """tauxo, tauyo on the p-grid (m^2 s^-2), SURVEY 8d."""
    nx, ny = cfg.nxpo, cfg.nypo
    j = np.arange(ny, dtype=np.float64)
    if cfg.cyclic:
        prof = 1.0e-4 * np.sin(np.pi * j / (ny - 1)) ** 2
    else:
        prof = -1.0e-4 * np.cos(2.0 * np.pi * j / (ny - 1))
    taux = np.asfortranarray(np.broadcast_to(prof[None, :], (nx, ny)).copy())
    tauy = np.zeros((nx, ny), order="F")
    return taux, tauy


def wekpo_from_tau(cfg, tauxo, tauyo):
    """Ocean-only Ekman pumping: wekto at T points and its p-point average
    (src/xfosubs.F:138 hxofac, 566-645)."""
    nx, ny = cfg.nxpo, cfg.nypo
    hxofac = 0.5 * (1.0 / (cfg.dxo * cfg.fnot))
    tx, ty = tauxo, tauyo
    wt = hxofac * (ty[1:, 1:] + ty[1:, :-1] - (ty[:-1, 1:] + ty[:-1, :-1])
                   + tx[1:, :-1] + tx[:-1, :-1] - (tx[1:, 1:] + tx[:-1, 1:]))
    wp = np.zeros((nx, ny), order="F")
    wp[1:-1, 1:-1] = 0.25 * (wt[:-1, :-1] + wt[:-1, 1:] + wt[1:, :-1] + wt[1:, 1:])
    wp[1:-1, 0] = 0.5 * (wt[:-1, 0] + wt[1:, 0])
    wp[1:-1, -1] = 0.5 * (wt[:-1, -1] + wt[1:, -1])
    if cfg.cyclic:
        wp[0, 1:-1] = 0.25 * (wt[-1, :-1] + wt[-1, 1:] + wt[0, :-1] + wt[0, 1:])
        wp[0, 0] = 0.5 * (wt[-1, 0] + wt[0, 0])
        wp[0, -1] = 0.5 * (wt[-1, -1] + wt[0, -1])
        wp[-1, :] = wp[0, :]
    else:
        wp[0, 1:-1] = 0.5 * (wt[0, :-1] + wt[0, 1:])
        wp[-1, 1:-1] = 0.5 * (wt[-1, :-1] + wt[-1, 1:])
        wp[0, 0], wp[0, -1] = wt[0, 0], wt[0, -1]
        wp[-1, 0], wp[-1, -1] = wt[-1, 0], wt[-1, -1]
    return np.asfortranarray(wt), wp


def tau_line_integrals(cfg, tauxo):
    """txisoc, txinoc of the cyclic momentum constraints (xfosubs.F:655-671)."""
    def line(v):
        return 0.5 * v[0] + v[1:-1].sum() + 0.5 * v[-1]
    txs = line(tauxo[:, 0] + tauxo[:, 1])
    txn = line(tauxo[:, -2] + tauxo[:, -1])
    return 0.5 * cfg.dxo * txs, 0.5 * cfg.dxo * txn


def gaussian_eddy(cfg, amp=0.15, lfold=None, xc=0.4, yc=0.55, noise=0.0, seed=247):
    """IC B / IC C of SURVEY 8d: surface Gaussian eddy (formula of the fork's restart
    generator, src/k247_make_restart_q-gcm.F90:240-262), optional smoothed noise."""
    nx, ny, nl = cfg.nxpo, cfg.nypo, cfg.nlo
    if lfold is None:
        lfold = max(8.0e4, 6.0 * cfg.dxo)
    x = np.arange(nx)[:, None] * cfg.dxo
    y = np.arange(ny)[None, :] * cfg.dyo
    po = np.zeros((nx, ny, nl), order="F")
    po[:, :, 0] = 9.8 * amp * np.exp(-((x - xc * cfg.xlo) ** 2 + (y - yc * cfg.ylo) ** 2) / lfold ** 2)
    if cfg.cyclic:
        # make the field exactly periodic in x and add a zonal-wave component
        i = np.arange(nx)[:, None]
        j = np.arange(ny)[None, :]
        po[:, :, 0] += 0.3 * np.sin(2 * np.pi * i / (nx - 1)) * np.sin(np.pi * j / (ny - 1))
        po[:, :, 1] = 0.2 * po[:, :, 0]
        po[-1, :, :] = po[0, :, :]
    if noise > 0.0:
        rng = np.random.default_rng(seed)
        for k in range(nl):
            r = rng.uniform(-1.0, 1.0, size=(nx, ny))
            s = r.copy()
            s[1:-1, 1:-1] = 0.2 * (r[1:-1, 1:-1] + r[:-2, 1:-1] + r[2:, 1:-1] + r[1:-1, :-2] + r[1:-1, 2:])
            po[:, :, k] += noise * np.abs(po[:, :, 0]).max() * s
        if cfg.cyclic:
            po[-1, :, :] = po[0, :, :]
    return po


def mixed_layer_fields(cfg, oml, seed=None):
    """Deterministic synthetic inputs of the ocean mixed layer: sst, sstm, fnetoc on the
    T grid (nxto,nyto) and a wind stress with both components on the p grid.  sst straddles toc(1) so that
    the convective adjustment of eqn (7.13) (src/omlsubs.F:116-119) is active in the northern half."""
    nxt, nyt = cfg.nxto, cfg.nyto
    x = (np.arange(nxt)[:, None] + 0.5) / nxt
    y = (np.arange(nyt)[None, :] + 0.5) / nyt
    sst = oml.toc[0] + 2.0 * np.cos(np.pi * y) + 0.5 * np.sin(2.0 * np.pi * x) * np.sin(np.pi * y)
    sstm = sst - 0.05 * np.cos(2.0 * np.pi * x) * np.cos(np.pi * y)
    if seed is not None:
        rng = np.random.default_rng(seed)
        sst = sst + 0.05 * rng.uniform(-1.0, 1.0, sst.shape)
        sstm = sstm + 0.05 * rng.uniform(-1.0, 1.0, sst.shape)
    fnet = 40.0 * np.cos(np.pi * y) * (1.0 + 0.2 * np.sin(2.0 * np.pi * x))
    tx, ty = wind_stress(cfg)
    xp = np.arange(cfg.nxpo)[:, None] / (cfg.nxpo - 1.0)
    yp = np.arange(cfg.nypo)[None, :] / (cfg.nypo - 1.0)
    ty = np.asfortranarray(2.0e-5 * np.sin(2.0 * np.pi * xp) * np.sin(np.pi * yp))
    return (np.asfortranarray(sst), np.asfortranarray(sstm), np.asfortranarray(fnet), np.asfortranarray(tx), ty)


def atmos_fields(acfg, noise=1.0e-2, seed=385):
    """Deterministic synthetic inputs of the atmospheric channel, all exactly periodic in x:
    pa, pam (nxpa,nypa,nla): a zonal jet in thermal-wind balance with a wavenumber-2/4 planetary wave and smoothed
    noise; wekpa, entat (p grid); ddynat (a smooth "mountain" under layer 1); xan and the line integrals txisat,
    txinat, enisat, eninat that xforc / aml would supply.  Magnitudes follow a spun-up double_gyre_coupled run
    (pa ~ 1e3 m^2 s^-2, wekpa ~ 1e-3 m s^-1)."""
    nx, ny, nl = acfg.nxpa, acfg.nypa, acfg.nla
    x = np.arange(nx)[:, None] / (nx - 1.0)
    y = np.arange(ny)[None, :] / (ny - 1.0)
    rng = np.random.default_rng(seed)
    pa = np.zeros((nx, ny, nl), order="F")
    for k in range(nl):
        jet = -1500.0 * (1.0 + 0.5 * k) * np.tanh(4.0 * (y - 0.5))
        wave = 400.0 * (1.0 + 0.3 * k) * np.sin(2 * np.pi * 2 * x + 0.7 * k) * np.sin(np.pi * y) ** 2 \
            + 150.0 * np.cos(2 * np.pi * 4 * x - 0.4 * k) * np.sin(2 * np.pi * y)
        r = rng.uniform(-1.0, 1.0, size=(nx - 1, ny))  # one period; 5-point smoothing, periodic in x
        s = r.copy()
        s[:, 1:-1] = 0.2 * (r[:, 1:-1] + np.roll(r, 1, axis=0)[:, 1:-1] + np.roll(r, -1, axis=0)[:, 1:-1]
                            + r[:, :-2] + r[:, 2:])
        s = np.vstack([s, s[:1]])
        pa[:, :, k] = jet + wave + noise * 1500.0 * s
    pa[-1, :, :] = pa[0, :, :]
    pam = np.asfortranarray(0.985 * pa)
    wekpa = np.asfortranarray(1.0e-3 * np.sin(2 * np.pi * x) * np.sin(np.pi * y) + 2.0e-4 * np.cos(2 * np.pi * 3 * x) * y)
    entat = np.asfortranarray(3.0e-4 * np.cos(2 * np.pi * x) * np.sin(np.pi * y) ** 2)
    ddynat = np.asfortranarray(2.0e-6 * np.exp(-((x - 0.3) ** 2 + (y - 0.45) ** 2) / 0.02) + 0.0 * y)
    for a in (wekpa, entat, ddynat):
        a[-1, :] = a[0, :]
    area = acfg.xla * acfg.yla
    return dict(pa=pa, pam=pam, wekpa=wekpa, entat=entat, ddynat=ddynat,
                xan=np.array([2.0e-6 * area] + [0.0] * (nl - 2)), txis=3.0e2 * acfg.xla / 3.0e7,
                txin=-2.0e2 * acfg.xla / 3.0e7, enis=np.array([1.0e-5 * acfg.xla] + [0.0] * (nl - 2)),
                enin=np.array([-0.6e-5 * acfg.xla] + [0.0] * (nl - 2)))
