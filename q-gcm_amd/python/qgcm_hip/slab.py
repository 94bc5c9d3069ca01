"""y-slab decomposition of the ocean hot path across GPUs (one process per GPU).

The reference is a single-process OpenMP code; this decomposition is new design
(SURVEY 8e).  Rank r owns a contiguous range of rows j for all i and layers - x is
never split, so the tendency kernel, the row transforms and the unpack stay local.
Per ocean step the ranks exchange

  * the slab summaries of the tridiagonal sweeps along y and of the area integrals (ONE
    all-gather of 7*nlo*nk doubles: both sweeps and the column sums are linear in the
    values entering a slab, see k_thomas.h, instead of two all-to-all transposes of the
    whole work array and an all-reduce); every rank derives bit-identical constraint
    coefficients from it,
  * halo rows of the new po (3 rows: del-6 of the lagged field) and qo (1 row).

The exchanges are issued either from here (torch.distributed between the stage calls) or
by the library itself (`use_library_exchanges`: RCCL from C++, qgcm_hip_slab_steps).

`SlabOcean` is the orchestration; it is independent of where the slab kernels
run (`HipSlab`: the C ABI on a GPU) and of the transport (`LocalComm`: several
slabs in one process, used to test the decomposition on one GPU; `DistComm`:
torch.distributed, "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).
All message buffers are torch tensors.

Import order: torch bundles its own HIP runtime; a process that uses torch (these transports do) must import it
before the first OceanModel / HipSlab loads libqgcm_hip.so, otherwise torch reports "No HIP GPUs are available".
"""
import ctypes as C

import numpy as np

from . import hostinit
from .lib import Params, check, load_library

HALO = 3


def partition(nyg, nranks):
    """Global row ranges (1-based, inclusive) of the y-slabs: near-equal, contiguous."""
    base, rem = divmod(nyg, nranks)
    out, g = [], 1
    for r in range(nranks):
        n = base + (1 if r < rem else 0)
        out.append((g, g + n - 1))
        g += n
    return out


def local_rows(nyg, g0, g1):
    """(nyl, joff, jlo, jhi): local array rows, global = local + joff, owned local range."""
    hlo = HALO if g0 > 1 else 0
    hhi = HALO if g1 < nyg else 0
    return (g1 - g0 + 1) + hlo + hhi, g0 - hlo - 1, hlo + 1, hlo + (g1 - g0 + 1)


def slab_slice(nyg, g0, g1):
    """numpy slice (0-based) of the global rows held locally (owned + halos)."""
    nyl, joff, _, _ = local_rows(nyg, g0, g1)
    return slice(joff, joff + nyl)


# --------------------------------------------------------------------------
# transports
# --------------------------------------------------------------------------
class LocalComm:
    """All slabs live in this process (virtual ranks)."""

    def __init__(self, nranks, after=None):
        self.nranks = nranks
        self.local_ranks = list(range(nranks))
        self.after = after  # e.g. torch.cuda.synchronize when the buffers are device tensors

    def all_gather(self, gath, send):
        n = send[0].numel()
        for g in gath:
            for r, s in enumerate(send):
                g[r * n:(r + 1) * n].copy_(s)
        if self.after:
            self.after()

    def halo_exchange(self, to_lo, to_hi, from_lo, from_hi):
        for r in range(self.nranks):
            if r > 0:
                from_lo[r].copy_(to_hi[r - 1])
            if r < self.nranks - 1:
                from_hi[r].copy_(to_lo[r + 1])
        if self.after:
            self.after()


class DistComm:
    """One slab per process over torch.distributed (backend nccl = RCCL, or gloo)."""

    def __init__(self, group=None, halo_via_all_gather=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.local_ranks = [self.rank]
        # halo rows either as point-to-point send/recv with the two neighbours or as ONE
        # all-gather of everybody's edge rows (2 x 92 KB per rank at 5 km: cheaper than two
        # latency-bound p2p launches on xGMI, and it uses a single collective type)
        self.halo_via_all_gather = halo_via_all_gather
        self._hsend = self._hgath = None
        # gloo moves device tensors through the host and is not ordered on the slab's HIP stream (rehearsals of the
        # multi-rank path on one GPU, CPU tests): bracket its collectives with device synchronisations. nccl = RCCL
        # is stream-ordered and needs none.
        self.host_staged = dist.get_backend(group) != "nccl"

    def _fence(self):
        if self.host_staged:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    def _all_gather(self, gath, send):
        if self.host_staged and send.is_cuda:  # gloo: explicit host staging (its device-tensor path is not reliable)
            import torch
            h = torch.empty(gath.numel(), dtype=gath.dtype)
            self.dist.all_gather_into_tensor(h, send.cpu(), group=self.group)
            gath.copy_(h)
        else:
            self.dist.all_gather_into_tensor(gath, send, group=self.group)

    def all_gather(self, gath, send):
        self._fence()
        self._all_gather(gath[0], send[0])
        self._fence()

    def halo_exchange(self, to_lo, to_hi, from_lo, from_hi):
        self._fence()
        try:
            return self._halo_exchange(to_lo, to_hi, from_lo, from_hi)
        finally:
            self._fence()

    def _halo_exchange(self, to_lo, to_hi, from_lo, from_hi):
        if self.halo_via_all_gather:
            return self._halo_all_gather(to_lo, to_hi, from_lo, from_hi)
        d, r = self.dist, self.rank
        stage = self.host_staged and (to_lo[0] if to_lo[0] is not None else to_hi[0]).is_cuda
        sl = (to_lo[0].cpu() if stage else to_lo[0]) if r > 0 else None
        sh = (to_hi[0].cpu() if stage else to_hi[0]) if r < self.nranks - 1 else None
        rl = (from_lo[0].cpu() if stage else from_lo[0]) if r > 0 else None
        rh = (from_hi[0].cpu() if stage else from_hi[0]) if r < self.nranks - 1 else None
        ops = []
        if r > 0:
            ops += [d.P2POp(d.isend, sl, r - 1, self.group), d.P2POp(d.irecv, rl, r - 1, self.group)]
        if r < self.nranks - 1:
            ops += [d.P2POp(d.isend, sh, r + 1, self.group), d.P2POp(d.irecv, rh, r + 1, self.group)]
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        if stage:
            if r > 0:
                from_lo[0].copy_(rl)
            if r < self.nranks - 1:
                from_hi[0].copy_(rh)

    def _halo_all_gather(self, to_lo, to_hi, from_lo, from_hi):
        r, P = self.rank, self.nranks
        ref = to_lo[0] if to_lo[0] is not None else to_hi[0]
        n = ref.numel()
        if self._hsend is None:
            self._hsend = ref.new_zeros(2 * n)
            self._hgath = ref.new_zeros(2 * n * P)
            if ref.is_cuda:  # fills run on torch's current stream: settle before the slab's stream writes (SlabOcean._settle)
                import torch
                torch.cuda.synchronize()
        if to_lo[0] is not None:
            self._hsend[:n].copy_(to_lo[0])
        if to_hi[0] is not None:
            self._hsend[n:].copy_(to_hi[0])
        self._all_gather(self._hgath, self._hsend)
        if r > 0:  # what the lower neighbour sent upwards
            from_lo[0].copy_(self._hgath[(2 * (r - 1) + 1) * n:(2 * (r - 1) + 2) * n])
        if r < P - 1:  # what the upper neighbour sent downwards
            from_hi[0].copy_(self._hgath[(2 * (r + 1)) * n:(2 * (r + 1) + 1) * n])


def rccl_unique_id():
    """The rendezvous id of qgcm_hip_comm_unique_id (call on rank 0, hand to every rank)."""
    buf = C.create_string_buffer(128)
    check(load_library().qgcm_hip_comm_unique_id(buf, 128))
    return buf.raw


def broadcast_unique_id(dist, device, group=None):
    """rank 0 creates the id, torch.distributed carries it to the others."""
    import torch
    t = torch.zeros(128, dtype=torch.uint8, device=device)
    if dist.get_rank(group) == 0:
        t.copy_(torch.frombuffer(bytearray(rccl_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, src=0, group=group)
    return bytes(t.cpu().numpy().tobytes())


# --------------------------------------------------------------------------
# one slab on one GPU through the C ABI
# --------------------------------------------------------------------------
def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class HipSlab:
    """The slab kernels of include/qgcm_hip.h ("y-slab building blocks")."""

    def __init__(self, cfg, consts, g0, g1, rank, nranks, device=-1, sync_each_call=False):
        import torch
        self.torch = torch
        self.cfg, self.rank, self.nranks = cfg, rank, nranks
        self.g0, self.g1 = g0, g1
        self.L = load_library()
        nl, nyg = cfg.nlo, cfg.nypo
        self.nyl, self.joff, self.jlo, self.jhi = local_rows(nyg, g0, g1)
        sl = slab_slice(nyg, g0, g1)
        p = Params()
        p.nxpo, p.nypo, p.nlo, p.cyclic = cfg.nxpo, nyg, nl, int(cfg.cyclic)
        p.fnot, p.beta, p.dxo, p.dyo = cfg.fnot, cfg.beta, cfg.dxo, cfg.dyo
        p.tdto, p.delek, p.bccooc, p.aoc = cfg.tdto, cfg.delek, cfg.bccooc, consts["aoc"]
        p.slab_g0, p.slab_g1 = g0, g1
        for k in range(nl):
            p.ah2oc[k], p.ah4oc[k], p.hoc[k], p.rdm2oc[k] = cfg.ah2oc[k], cfg.ah4oc[k], cfg.hoc[k], consts["rdm2oc"][k]
        for k in range(nl - 1):
            p.gpoc[k] = cfg.gpoc[k]
        for name in ("amatoc", "ctl2moc", "ctm2loc"):
            arr = getattr(p, name)
            for i, v in enumerate(np.asarray(consts[name]).ravel(order="F")):
                arr[i] = v
        self.h = C.c_void_p()
        check(self.L.qgcm_hip_create(C.byref(self.h), C.byref(p), int(device)))
        yp = np.ascontiguousarray(consts["yporel"][sl])
        bd2 = np.ascontiguousarray(consts["bd2oc"])
        dd = np.asfortranarray(consts["ddynoc"][:, sl])
        check(self.L.qgcm_hip_set_grid(self.h, _dp(yp), _dp(bd2), _dp(dd)))
        self.consts = consts
        if cfg.cyclic:
            # the homogeneous solutions of a channel depend on y only (conhoms.F:376-543): local slices + global scalars
            hg = consts
            args = [np.asfortranarray(hg["pch1oc"][sl, :]), np.asfortranarray(hg["pch2oc"][sl, :]),
                    np.ascontiguousarray(hg["pbhoc"][sl]),
                    *[np.ascontiguousarray(hg[k], dtype=np.float64) for k in ("aipcho", "hc1soc", "hc2soc", "hc1noc", "hc2noc")]]
            check(self.L.qgcm_hip_set_homog_cyc(self.h, *[_dp(a) for a in args], float(hg["hbsioc"]), float(hg["aipbho"])))
        elif "ochom" in consts:  # global homogeneous solutions given; else SlabOcean.homsol() computes them on the slabs
            self.set_homog(np.asfortranarray(consts["ochom"][:, sl, :]), consts["cdiffo"], consts["cdhoc"])
        self.sync_each_call = sync_each_call
        self.device = torch.device("cuda", torch.cuda.current_device() if device < 0 else device)
        self.th_len = self.L.qgcm_hip_thomas_msg_len(self.h)
        self.cst_len = self.L.qgcm_hip_thomas_const_len(self.h)
        self.halo_len = self.L.qgcm_hip_halo_msg_len(self.h)
        self.stream_ptr = self.L.qgcm_hip_stream(self.h)
        self.oml_on = False

    def set_homog(self, ochom_local, cdiffo, cdhoc):
        oh, cd, ch = np.asfortranarray(ochom_local), np.asfortranarray(cdiffo), np.asfortranarray(cdhoc)
        check(self.L.qgcm_hip_set_homog_box(self.h, _dp(oh), _dp(cd), _dp(ch)))

    def wrk_fill(self, v):
        check(self.L.qgcm_hip_wrk_fill(self.h, float(v))); self._done()

    def wrk_get(self):
        w = np.zeros((self.cfg.nxpo, self.nyl, self.cfg.nlo), order="F")
        check(self.L.qgcm_hip_wrk_get(self.h, _dp(w)))
        return w

    def area_integrals(self):
        x = np.zeros(self.cfg.nlo)
        check(self.L.qgcm_hip_area_integrals(self.h, _dp(x)))
        return x

    # buffers -------------------------------------------------------------
    def new_buffer(self, n):
        return self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)

    @staticmethod
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _done(self):
        if self.sync_each_call:
            check(self.L.qgcm_hip_sync(self.h))

    def sync(self):
        check(self.L.qgcm_hip_sync(self.h))

    def close(self):
        if self.h:
            self.L.qgcm_hip_destroy(self.h)
            self.h = None

    # state (local blocks incl. halos) ---------------------------------------
    def set_state(self, po, pom, qo, qom):
        a = [np.asfortranarray(x, dtype=np.float64) for x in (po, pom, qo, qom)]
        check(self.L.qgcm_hip_set_state(self.h, *[_dp(x) for x in a]))

    def get_state(self):
        a = [np.zeros((self.cfg.nxpo, self.nyl, self.cfg.nlo), order="F") for _ in range(4)]
        check(self.L.qgcm_hip_get_state(self.h, *[_dp(x) for x in a]))
        return a

    def set_forcing(self, wekpo, entoc, xon):
        w, e = np.asfortranarray(wekpo, dtype=np.float64), np.asfortranarray(entoc, dtype=np.float64)
        x = np.ascontiguousarray(xon, dtype=np.float64)
        check(self.L.qgcm_hip_set_forcing(self.h, _dp(w), _dp(e), _dp(x)))

    def set_cyc_forcing(self, txisoc, txinoc, enisoc=None, eninoc=None):
        nl = self.cfg.nlo
        es = np.zeros(nl - 1) if enisoc is None else np.ascontiguousarray(enisoc, dtype=np.float64)
        en = np.zeros(nl - 1) if eninoc is None else np.ascontiguousarray(eninoc, dtype=np.float64)
        check(self.L.qgcm_hip_set_cyc_forcing(self.h, float(txisoc), float(txinoc), _dp(es), _dp(en)))

    # ocean mixed layer (T grid: row j between p rows j and j+1; local arrays hold rows 1..nyl-1 of the slab view) ------
    def oml_init(self, om):
        """Switch the mixed layer on for this slab (before SlabOcean sizes its message buffers)."""
        from .lib import OmlParams
        p = OmlParams()
        p.hmoc, p.toc1, p.toc2, p.st2d, p.st4d = om.hmoc, om.toc[0], om.toc[1], om.st2d, om.st4d
        p.ycexp, p.rrcpoc, p.tsbdy, p.tnbdy = om.ycexp, om.rrcpoc, om.tsbdy, om.tnbdy
        p.sb_hflux, p.nb_hflux = int(om.sb_hflux), int(om.nb_hflux)
        check(self.L.qgcm_hip_oml_init(self.h, C.byref(p)))
        self.oml_on = True
        self.th_len = self.L.qgcm_hip_thomas_msg_len(self.h)
        self.halo_len = self.L.qgcm_hip_halo_msg_len(self.h)
        self.oml_len = self.L.qgcm_hip_oml_msg_len(self.h)

    def t_slice(self):
        """Rows of a GLOBAL (nxto, nyto) T-grid array that this slab's local T array holds."""
        return slice(self.joff, self.joff + self.nyl - 1)

    def oml_set_state(self, sst, sstm):
        """sst, sstm: GLOBAL (nxto, nyto) arrays; the local rows (incl. halo rows) are cut out here."""
        a = [np.asfortranarray(x[:, self.t_slice()], dtype=np.float64) for x in (sst, sstm)]
        check(self.L.qgcm_hip_oml_set_state(self.h, *[_dp(x) for x in a]))

    def oml_get_state(self):
        """Local (nxto, nyl-1) sst, sstm; owned T rows are local rows jlo..(jhi or jhi-1 on the last rank)."""
        a = [np.zeros((self.cfg.nxto, self.nyl - 1), order="F") for _ in range(2)]
        check(self.L.qgcm_hip_oml_get_state(self.h, *[_dp(x) for x in a]))
        return a

    def oml_set_forcing(self, fnetoc, wekto, tauxo, tauyo):
        """GLOBAL arrays: fnetoc, wekto on the T grid (nxto, nyto); tauxo, tauyo on the p grid (nxpo, nypo)."""
        sl = slab_slice(self.cfg.nypo, self.g0, self.g1)
        a = [np.asfortranarray(fnetoc[:, self.t_slice()], dtype=np.float64), np.asfortranarray(wekto[:, self.t_slice()], dtype=np.float64),
             np.asfortranarray(tauxo[:, sl], dtype=np.float64), np.asfortranarray(tauyo[:, sl], dtype=np.float64)]
        check(self.L.qgcm_hip_oml_set_forcing(self.h, *[_dp(x) for x in a]))

    def set_scalars(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        check(self.L.qgcm_hip_set_scalars(self.h, _dp(s)))

    def get_scalars(self):
        s = np.zeros(2 * (self.cfg.nlo - 1) + 4 * self.cfg.nlo)
        check(self.L.qgcm_hip_get_scalars(self.h, _dp(s)))
        return s

    def set_sponge(self, r_spl, c1_spl):
        """The fork's sponge layer on this slab: the GLOBAL ramp r_spl(nxpo, nypo), of which the local rows (incl. halo
        rows) are sent; None switches the term off."""
        if r_spl is None:
            check(self.L.qgcm_hip_set_sponge(self.h, None, 0.0))
            return
        loc = np.asfortranarray(np.asarray(r_spl, dtype=np.float64)[:, self.joff:self.joff + self.nyl])
        check(self.L.qgcm_hip_set_sponge(self.h, _dp(loc), float(c1_spl)))

    def get_monitors(self):
        """(ermaso, emfroc) of the last constraint solve of a zonally cyclic slab (every rank holds the same numbers)."""
        e, f = np.zeros(self.cfg.nlo - 1), np.zeros(self.cfg.nlo - 1)
        check(self.L.qgcm_hip_get_monitors(self.h, _dp(e), _dp(f)))
        return e, f

    # slab kernels -------------------------------------------------------------
    def qgostep(self):
        check(self.L.qgcm_hip_qgostep(self.h)); self._done()

    def row_transform(self, inverse):
        check(self.L.qgcm_hip_row_transform(self.h, int(inverse))); self._done()

    def thomas_phase(self, phase, gath, send):
        check(self.L.qgcm_hip_thomas_phase(self.h, int(phase), self._ptr(gath), self._ptr(send), self.rank, self.nranks))
        self._done()

    def thomas_consts(self, dst):
        """This slab's right-hand-side independent summary constants -> dst (device buffer of cst_len doubles)."""
        check(self.L.qgcm_hip_thomas_consts(self.h, self._ptr(dst))); self._done()

    def set_thomas_consts(self, gath):
        """All ranks' constants (rank-major); once after set-up."""
        check(self.L.qgcm_hip_set_thomas_consts(self.h, self._ptr(gath), self.nranks)); self._done()

    def constr(self):
        check(self.L.qgcm_hip_constr(self.h)); self._done()

    def unpack(self, fuse_ocqbdy=True):
        check(self.L.qgcm_hip_unpack(self.h, int(fuse_ocqbdy))); self._done()

    def halo_pack(self, to_lo, to_hi):
        check(self.L.qgcm_hip_halo_pack(self.h, self._ptr(to_lo), self._ptr(to_hi))); self._done()

    def halo_unpack(self, from_lo, from_hi):
        check(self.L.qgcm_hip_halo_unpack(self.h, self._ptr(from_lo), self._ptr(from_hi))); self._done()

    def lf_average(self):
        check(self.L.qgcm_hip_lf_average(self.h)); self._done()

    # exchanges issued by the library itself (RCCL on its own stream) ---------------
    def comm_init(self, comm_id):
        """Collective over all ranks' handles; comm_id = rccl_unique_id() of rank 0."""
        check(self.L.qgcm_hip_comm_init(self.h, comm_id, len(comm_id), self.rank, self.nranks))
        self.has_comm = True

    def comm_probe(self, reps=200):
        """The step's exchanges back to back (collective): us per summaries all-gather, halo all-gather, halo send/recv."""
        us = (C.c_double * 3)()
        check(self.L.qgcm_hip_comm_probe(self.h, int(reps), us))
        return [us[0], us[1], us[2]]

    def set_halo_p2p(self, on):
        check(self.L.qgcm_hip_comm_set_halo_p2p(self.h, int(on)))

    def set_overlap(self, on):
        """Library-issued exchanges: halo exchange on a second stream under the next step's inner tendency tiles."""
        check(self.L.qgcm_hip_comm_set_overlap(self.h, int(on)))

    def slab_steps(self, s0, n):
        check(self.L.qgcm_hip_slab_steps(self.h, int(s0), int(n)))
        self._done()

    def stage(self, n, a=None, b=None, c=None, flags=0):
        """One C call per communication-free stage (qgcm_hip_slab_stage)."""
        check(self.L.qgcm_hip_slab_stage(self.h, int(n), self._ptr(a), self._ptr(b), self._ptr(c), self.rank, self.nranks, int(flags)))
        self._done()


# --------------------------------------------------------------------------
# orchestration
# --------------------------------------------------------------------------
def tend_tile_rows(cfg, g0, g1):
    """Tile rows of the tendency launch on the slab of global rows g0..g1 (k_tend.h tend_tiling: 16-row tiles; the
    northern wall row of a box basin is peeled off when it would open a tile row of its own)."""
    rows = g1 - g0 + 1
    peel = (not cfg.cyclic) and rows % 16 == 1 and rows > 1 and g1 == cfg.nypo
    return rows // 16 if peel else -(-rows // 16)


class SlabOcean:
    """The distributed ocean step.  `slabs` are the slab objects this process owns
    (one per local rank of `comm`); a slab object provides the methods of HipSlab."""

    def __init__(self, cfg, slabs, comm, stream_ctx=None):
        self.cfg, self.slabs, self.comm = cfg, slabs, comm
        self.P = comm.nranks
        nl = cfg.nlo
        self.stream_ctx = stream_ctx  # context manager factory that makes torch's current stream the slab's stream
        self.th_send = [s.new_buffer(s.th_len) for s in slabs]
        self.th_gath = [s.new_buffer(s.th_len * self.P) for s in slabs]
        self.h_to_lo = [s.new_buffer(s.halo_len) if s.rank > 0 else None for s in slabs]
        self.h_to_hi = [s.new_buffer(s.halo_len) if s.rank < self.P - 1 else None for s in slabs]
        self.h_from_lo = [s.new_buffer(s.halo_len) if s.rank > 0 else None for s in slabs]
        self.h_from_hi = [s.new_buffer(s.halo_len) if s.rank < self.P - 1 else None for s in slabs]
        # mixed layer (HipSlab.oml_init on every slab BEFORE this constructor): one more small all-gather per step
        self.oml_on = all(getattr(s, "oml_on", False) for s in slabs)
        if self.oml_on:
            self.om_send = [s.new_buffer(s.oml_len) for s in slabs]
            self.om_gath = [s.new_buffer(s.oml_len * self.P) for s in slabs]
        self._settle()
        self.step_index = 1
        # the right-hand-side independent part of the slab summaries is exchanged once
        if self.P > 1:
            cs_send = [s.new_buffer(s.cst_len) for s in slabs]
            cs_gath = [s.new_buffer(s.cst_len * self.P) for s in slabs]
            self._settle()
            for i, x in enumerate(slabs):
                x.thomas_consts(cs_send[i])
            self._comm(comm.all_gather, cs_gath, cs_send)
            for i, x in enumerate(slabs):
                x.set_thomas_consts(cs_gath[i])
            for x in slabs:
                x.sync()

    @staticmethod
    def _settle():
        """torch zero-fills a new buffer on ITS current stream - the null stream unless the caller made the slab's
        stream current - and the slabs' streams are non-blocking: not ordered against it.  A kernel that writes into a
        fresh buffer could be overtaken by the fill.  Wait for the fills before the buffers are used."""
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        except ImportError:  # numpy-backed slabs of the CPU tests
            pass

    def _comm(self, fn, *a):
        if self.stream_ctx is None:
            fn(*a)
        else:
            with self.stream_ctx():
                fn(*a)

    def step(self, s):
        S, cm = self.slabs, self.comm
        if self.oml_on:  # `call oml` precedes qgostep (src/q-gcm.F:1232): its mean entrainment is a basin-wide number
            for i, x in enumerate(S):
                x.stage(10, self.om_send[i])
            self._comm(cm.all_gather, self.om_gath, self.om_send)
            for i, x in enumerate(S):
                x.stage(11, self.om_gath[i])
        if getattr(self, "_halo_pending", False):
            # early_tend: the halo rows of the previous step have not been unpacked yet - the inner tile rows of the
            # tendency launch do not read them (stage 4); then halo rows in (stage 3) and the rest of stage 1 (stage 5)
            for x in S:
                x.stage(4)
            for i, x in enumerate(S):
                x.stage(3, self.h_from_lo[i], self.h_from_hi[i], None, 0)
            self._halo_pending = False
            for i, x in enumerate(S):
                x.stage(5, self.th_send[i])
        else:
            for i, x in enumerate(S):  # tendency, forward row transform, slab summary of the y sweeps
                x.stage(1, self.th_send[i])
        self._comm(cm.all_gather, self.th_gath, self.th_send)
        # both sweeps + basin-wide area integrals, constraints, inverse row transform,
        # modes -> layers (+ boundary PV), halo rows out
        for i, x in enumerate(S):
            x.stage(2, self.th_gath[i], self.h_to_lo[i], self.h_to_hi[i])
        if self.P > 1:
            self._comm(cm.halo_exchange, self.h_to_lo, self.h_to_hi, self.h_from_lo, self.h_from_hi)
        avg = 1 if (s - 1) % 25 == 0 else 0
        if (getattr(self, "early_tend", False) and self.P > 1 and not avg and not self.oml_on
                and all(tend_tile_rows(self.cfg, x.g0, x.g1) >= 3 for x in S)):
            self._halo_pending = True  # stage 3 of this step follows stage 4 of the next one (the order the library's
            return                     # overlapped exchange produces, qgcm_hip_comm_set_overlap)
        if self.P > 1 or avg:
            for i, x in enumerate(S):  # halo rows in, leapfrog averaging every 25th step
                x.stage(3, self.h_from_lo[i], self.h_from_hi[i], None, avg)

    def _join(self):
        if getattr(self, "_halo_pending", False):
            for i, x in enumerate(self.slabs):
                x.stage(3, self.h_from_lo[i], self.h_from_hi[i], None, 0)
            self._halo_pending = False

    def homsol(self):
        """homsol of the box ocean (src/conhoms.F:549-641) ON the slabs: the modal Helmholtz problems of a step are
        homsol's, so one distributed solve with right-hand side 1 gives every ochom(:,:,m); no host-side solver.
        Sets the homogeneous solutions on every local slab and returns the global products aipohs, cdiffo, cdhoc."""
        cfg, S = self.cfg, self.slabs
        nl = cfg.nlo
        rdm2, cm2l = S[0].consts["rdm2oc"], S[0].consts["ctm2loc"]
        for i, x in enumerate(S):
            x.wrk_fill(1.0)
            x.row_transform(0)
            x.thomas_phase(1, None, self.th_send[i])
        self._comm(self.comm.all_gather, self.th_gath, self.th_send)
        sols, xin = [], None
        for i, x in enumerate(S):
            x.thomas_phase(2, self.th_gath[i], None)
            xin = x.area_integrals()          # dxo*dyo*xintp(solution of mode m), basin-wide, the same on every rank
            x.row_transform(1)
            sols.append(x.wrk_get())
        dA = cfg.dxo * cfg.dyo
        aipohs = np.array([dA * cfg.nxto * cfg.nyto + rdm2[m + 1] * xin[m + 1] for m in range(nl - 1)])  # xintp(1) = nxto*nyto
        cdiffo = np.zeros((nl, nl - 1), order="F")
        cdhoc = np.zeros((nl - 1, nl - 1), order="F")
        for k in range(nl - 1):
            for m in range(nl):
                cdiffo[m, k] = cm2l[m, k + 1] - cm2l[m, k]
            for m in range(nl - 1):
                cdhoc[k, m] = (cm2l[m + 1, k + 1] - cm2l[m + 1, k]) * aipohs[m]
        for x, w in zip(S, sols):
            oh = np.zeros((cfg.nxpo, x.nyl, nl - 1), order="F")
            for m in range(nl - 1):
                oh[:, :, m] = 1.0 + rdm2[m + 1] * w[:, :, m + 1]
            # halo rows belong to the neighbours' solves; unpack only reads ochom on owned rows
            x.set_homog(oh, cdiffo, cdhoc)
            x.ochom_local = oh
        return dict(aipohs=aipohs, cdiffo=cdiffo, cdhoc=cdhoc)

    def use_library_exchanges(self, comm_id):
        """From now on steps() runs qgcm_hip_slab_steps: the library issues the RCCL exchanges
        itself, Python is out of the step loop.  One slab per process; collective."""
        assert len(self.slabs) == 1
        self.slabs[0].comm_init(comm_id)
        self.native = True

    def steps(self, n, s0=None):
        s0 = self.step_index if s0 is None else int(s0)
        if getattr(self, "native", False):
            self.slabs[0].slab_steps(s0, n)
        else:
            for s in range(s0, s0 + int(n)):
                self.step(s)
            self._join()
        self.step_index = s0 + int(n)

    # helpers to scatter / gather global arrays (host side, for tests and set-up) --
    def scatter_state(self, po, pom, qo, qom, wekpo, entoc, xon, scal):
        nyg = self.cfg.nypo
        for x in self.slabs:
            sl = slab_slice(nyg, x.g0, x.g1)
            x.set_state(po[:, sl, :], pom[:, sl, :], qo[:, sl, :], qom[:, sl, :])
            x.set_forcing(wekpo[:, sl], entoc[:, sl], xon)
            x.set_scalars(scal)

    def gather_local(self):
        """Owned rows of the local slabs: list of (g0, g1, [po, pom, qo, qom])."""
        out = []
        for x in self.slabs:
            st = x.get_state()
            out.append((x.g0, x.g1, [a[:, x.jlo - 1:x.jhi, :] for a in st]))
        return out


def global_consts(cfg, helmholtz=None):
    """Start-up constants on the GLOBAL grid (numpy, init-only), as in OceanModel.  helmholtz = None: without the
    homogeneous solutions - SlabOcean.homsol() then computes them on the slabs (no host-side solver)."""
    A, rdm2, cl2m, cm2l = hostinit.eigmod(cfg.gpoc, cfg.hoc, cfg.fnot)
    aoc, bd2 = hostinit.bd2oc(cfg)
    c = dict(amatoc=A, rdm2oc=rdm2, ctl2moc=cl2m, ctm2loc=cm2l, aoc=aoc, bd2oc=bd2, yporel=cfg.yporel(),
             ddynoc=np.zeros((cfg.nxpo, cfg.nypo), order="F"))
    if cfg.cyclic:
        # channel: the homogeneous solutions depend on y only - a tridiagonal solve per mode, no 2-D solver needed
        solve = helmholtz if helmholtz is not None else (lambda rhs, boc: hostinit.helmholtz_cyc_column(cfg, rhs, boc))
        c.update(hostinit.homsol_cyc(cfg, rdm2, bd2, c["yporel"], solve))
    elif helmholtz is not None:
        c.update(hostinit.homsol_box(cfg, rdm2, cm2l, bd2, helmholtz))
    return c
