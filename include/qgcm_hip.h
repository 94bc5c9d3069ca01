/* qgcm_hip.h - C ABI of the MI355X-native Q-GCM ocean PV-advance / inversion path.
 *
 * Drop-in boundary for the three argument-less calls the reference main
 * program makes once per ocean step (src/q-gcm.F:1243-1249):
 *
 *     call qgostep          -> qgcm_hip_qgostep()   (src/qgosubs.F:45-221 + ocadif 231-446)
 *     call ocinvq           -> qgcm_hip_ocinvq()    (src/ocisubs.F:64-407, hsbxoc 415-512, hscyoc 521-618)
 *     call ocqbdy (qo, po)  -> qgcm_hip_ocqbdy()    (src/vorsubs.F:245-388)
 *
 * plus the leapfrog time-level averaging block (src/q-gcm.F:1328-1366), the
 * Helmholtz solver that homsol calls at start-up (src/conhoms.F:454-455,572)
 * and the data movement that replaces the reference's shared module arrays
 * (ocstate: src/ocstate_data.F:39-42; occonst: src/occonst_data.F:36-44;
 * ochomog: src/ochomog_data.F:44-68; ocisubs: src/ocisubs.F:51-55).
 *
 * Conventions
 *  - plain C: raw double pointers + sizes, no Fortran descriptors, no torch types.
 *  - every host array is Fortran ordered exactly as the reference declares
 *    it, e.g. po(nxpo,nypo,nlo): element (i,j,k) 1-based at
 *    (i-1) + nxpo*((j-1) + nypo*(k-1)).  Host buffers are caller-owned and
 *    never retained.
 *  - the device owns the authoritative po,pom,qo,qom between set_state and
 *    get_state; all kernels run on one HIP stream owned by the handle.
 *  - every entry point returns 0 on success, non-zero on failure;
 *    qgcm_hip_last_error() returns a static description.  The reference's
 *    own convention is print + stop (e.g. src/ocisubs.F:361-365); the Fortran
 *    shim (q-gcm_amd/fortran) restores that behaviour.
 *  - one host thread per handle (the reference is called from its single
 *    main thread); calls are asynchronous until qgcm_hip_sync / a get_*.
 *  - there is NO CPU fallback: without a HIP device create() fails.
 */
#ifndef QGCM_HIP_H
#define QGCM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QGCM_HIP_MAXL 8 /* max number of QG layers supported: 2 <= nlo <= 8 (kernels instantiated for every count;
                           the fused single-launch forms of the inversion for nlo <= 4) */
/* 2: qgcm_hip_params.atmos + the atmosphere entry points
 * 3: qgcm_hip_get_monitors (required by the Fortran shim), qgcm_hip_prepare_steps, qgcm_hip_stream_mix_bandwidth,
 *    qgcm_hip_set_sponge, qgcm_hip_set_cu_range; halo rows of qgcm_hip_slab_steps default to neighbour send/recv */
#define QGCM_HIP_ABI_VERSION 3

typedef struct qgcm_hip_ctx *qgcm_hip_handle;

/* Scalars + small matrices of MODULE parameters / occonst / ochomog that the
 * path reads (src/parameters_data.F:23-147, src/occonst_data.F:36-44).
 * Matrices are Fortran ordered with leading dimension nlo, packed. */
typedef struct qgcm_hip_params {
  int nxpo, nypo, nlo; /* p-grid size, layers */
  int cyclic;          /* 1 = -Dcyclic_ocean (hscyoc path), 0 = box (hsbxoc) */
  double fnot, beta;   /* parameters_data.F: fnot, beta */
  double dxo, dyo;     /* occonst: grid spacing (dyo = dxo in the reference) */
  double tdto;         /* occonst: 2*dto */
  double delek;        /* bottom Ekman layer thickness */
  double bccooc;       /* mixed BC coefficient */
  double ah2oc[QGCM_HIP_MAXL];
  double ah4oc[QGCM_HIP_MAXL];
  double hoc[QGCM_HIP_MAXL];
  double gpoc[QGCM_HIP_MAXL];                        /* nlo-1 used */
  double amatoc[QGCM_HIP_MAXL * QGCM_HIP_MAXL];      /* amatoc(nlo,nlo)   eigmode.f:131-144 */
  double ctl2moc[QGCM_HIP_MAXL * QGCM_HIP_MAXL];     /* ctl2moc(nlo,nlo)  eigmode.f:420-428 */
  double ctm2loc[QGCM_HIP_MAXL * QGCM_HIP_MAXL];     /* ctm2loc(nlo,nlo) */
  double rdm2oc[QGCM_HIP_MAXL];                      /* 1/Rd^2 per mode */
  double aoc;                                        /* ocisubs: 1/dyo^2 (q-gcm.F:932) */
  /* y-slab decomposition (no counterpart in the reference, which is single-process):
   * this handle owns the global rows slab_g0..slab_g1 (1-based, inclusive) of the
   * nypo rows; 0,0 = the whole domain.  In slab mode every host array passed to
   * set_/get_* is the LOCAL block (nxpo, nyl, .) with nyl = owned rows + a 3-row
   * halo on each side that has a neighbour (qgcm_hip_local_rows). */
  int slab_g0, slab_g1;
  /* 1 = the handle is the ATMOSPHERIC channel of a coupled run: qgastep / atinvq / atqzbd
   * (src/qgasubs.F:45-317, src/atisubs.F:60-395, src/vorsubs.F:396-480).  Zonally periodic (cyclic must be 1);
   * the fields of this struct then carry MODULE atconst / parameters values under the ocean's names:
   * nxpo,nypo,nlo = nxpa,nypa,nla; dxo,dyo = dxa,dya; tdto = tdta; bccooc = bccoat; ah4oc = ah4at; hoc = hat;
   * gpoc = gpat; amatoc.. = amatat, ctl2mat, ctm2lat, rdm2at; aoc = aat; delek and ah2oc are ignored.
   * Differences from the cyclic ocean that the kernels honour: layer 1 is the bottom layer (topography term
   * in layer 1), forcing signs of src/qgasubs.F:128-131, no drag and no Del-4th term, constraint right-hand
   * sides of src/atisubs.F:177-196, dpiat = integral of pa(k)-pa(k+1), src/vorsubs.F:470 as written,
   * time levels averaged when mod(nt-1,100) == 0 (src/q-gcm.F:1370). */
  int atmos;
} qgcm_hip_params;

/* ---- life cycle -------------------------------------------------------- */
/* device < 0: use the current HIP device. */
int qgcm_hip_create(qgcm_hip_handle *h, const qgcm_hip_params *prm, int device);
int qgcm_hip_destroy(qgcm_hip_handle h);
const char *qgcm_hip_last_error(void);
int qgcm_hip_abi_version(void);

/* yporel(nypo), bd2oc(nxto) [reference/FFTPACK ordering, q-gcm.F:933-954],
 * ddynoc(nxpo,nypo).  Builds the device-side Thomas tables. */
int qgcm_hip_set_grid(qgcm_hip_handle h, const double *yporel, const double *bd2oc,
                      const double *ddynoc);
/* The part of set_grid that does not need bd2oc: yporel and ddynoc only.  The main program calls ocqbdy / atqzbd on
 * host arrays (src/q-gcm.F:724-725, 743-744) before it computes bd2oc (:932-972); qgcm_hip_ocqbdy_host needs no more
 * than this.  The stepping entry points still require qgcm_hip_set_grid. */
int qgcm_hip_set_geometry(qgcm_hip_handle h, const double *yporel, const double *ddynoc);

/* Products of homsol (src/conhoms.F:544-641 box / 376-543 cyclic).
 * box:    ochom(nxpo,nypo,nlo-1), cdiffo(nlo,nlo-1), cdhoc(nlo-1,nlo-1)
 *         (the LU factors cdhlu/ipivch are recomputed internally).
 * cyclic: pch1oc(nypo,nlo-1), pch2oc(nypo,nlo-1), pbhoc(nypo), aipcho(nlo-1),
 *         hc1soc, hc2soc, hc1noc, hc2noc (nlo-1 each), hbsioc, aipbho. */
int qgcm_hip_set_homog_box(qgcm_hip_handle h, const double *ochom, const double *cdiffo,
                           const double *cdhoc);
int qgcm_hip_set_homog_cyc(qgcm_hip_handle h, const double *pch1oc, const double *pch2oc,
                           const double *pbhoc, const double *aipcho, const double *hc1soc,
                           const double *hc2soc, const double *hc1noc, const double *hc2noc,
                           double hbsioc, double aipbho);

/* ---- state (MODULE ocstate) -------------------------------------------- */
/* Any pointer may be NULL to skip that field. */
int qgcm_hip_set_state(qgcm_hip_handle h, const double *po, const double *pom,
                       const double *qo, const double *qom);
int qgcm_hip_get_state(qgcm_hip_handle h, double *po, double *pom, double *qo, double *qom);
/* wekpo(nxpo,nypo), entoc(nxpo,nypo), xon(nlo-1)  (written by xforc / oml) */
int qgcm_hip_set_forcing(qgcm_hip_handle h, const double *wekpo, const double *entoc,
                         const double *xon);
/* The fork's sponge layer (cpp option sponge_layer_k247): the leapfrog step of qgostep gains
 *   + tdto*c1_spl*r_spl(i,j)*(qom(i,j,k) - beta*yporel(j))            src/qgosubs.F:203-205
 * r_spl(nxpo,nypo): the ramp of MODULE occonst (src/occonst_data.F:100-105) as the main program sets it
 * (src/q-gcm.F:1154-1168; a y-slab handle takes its local rows); c1_spl: src/parameters_data.F:144.
 * NULL switches the term off again (the default: no BASELINE configuration defines the option). */
int qgcm_hip_set_sponge(qgcm_hip_handle h, const double *r_spl, double c1_spl);
/* cyclic only: txisoc, txinoc (xforc), enisoc/eninoc(nlo-1) (oml) */
int qgcm_hip_set_cyc_forcing(qgcm_hip_handle h, double txisoc, double txinoc,
                             const double *enisoc, const double *eninoc);
/* constraint scalars of MODULE ochomog:
 *   scal[0 .. nlo-2]        dpioc
 *   scal[nlo-1 .. 2nlo-3]   dpiocp
 *   then (cyclic) ocncs, ocncn, ocncsp, ocncnp (nlo each); box: ignored/zero.
 * length 2*(nlo-1) + 4*nlo. */
int qgcm_hip_set_scalars(qgcm_hip_handle h, const double *scal);
int qgcm_hip_get_scalars(qgcm_hip_handle h, double *scal);
/* diagnostics of the last ocinvq: xinhom(nlo); coef = hclco(nlo-1) [box] or
 * c1(nlo-1), c2(nlo-1), c3 [cyclic] */
int qgcm_hip_get_inv_diag(qgcm_hip_handle h, double *xinhom, double *coef);
/* Continuity monitors of the last ocinvq / atinvq of a zonally cyclic handle (MODULE monitor: ermaso, emfroc,
 * src/ocisubs.F:268-283; ermasa, emfrat, src/atisubs.F:236-248): nlo-1 doubles each.  Synchronous. */
int qgcm_hip_get_monitors(qgcm_hip_handle h, double *ermas, double *emfr);

/* ---- the path ----------------------------------------------------------- */
int qgcm_hip_qgostep(qgcm_hip_handle h);    /* replaces "call qgostep"        q-gcm.F:1243 */
int qgcm_hip_ocinvq(qgcm_hip_handle h);     /* replaces "call ocinvq"         q-gcm.F:1246 */
int qgcm_hip_ocqbdy(qgcm_hip_handle h);     /* replaces "call ocqbdy (qo,po)" q-gcm.F:1249 */
int qgcm_hip_lf_average(qgcm_hip_handle h); /* ocean part of q-gcm.F:1328-1366 (incl. sst once qgcm_hip_oml_init was called) */
/* "call ocqbdy (q, p)" / "call atqzbd (q, p)" on HOST arrays, as the main program does at start-up for both time
 * levels before the device owns the state (src/q-gcm.F:724-725, 743-744): p(nxpo,nypo,nlo) is uploaded to scratch,
 * the boundary PV kernel runs, and the boundary ring of q(nxpo,nypo,nlo) is written back (interior untouched).
 * Does not touch the device-resident state. Synchronous. */
int qgcm_hip_ocqbdy_host(qgcm_hip_handle h, double *q, const double *p);
/* n whole ocean steps starting at 1-based ocean step index s0: qgostep, ocinvq,
 * ocqbdy and, when mod(s-1,25)==0, the averaging (nt = 1+(s-1)*nstr in
 * q-gcm.F:1222,1328).  Uses captured HIP graphs. */
int qgcm_hip_steps(qgcm_hip_handle h, int s0, int n);
int qgcm_hip_sync(qgcm_hip_handle h);

/* ---- the atmosphere path (handles created with atmos = 1) ------------------
 * One-for-one replacements of the three calls the main program makes every atmospheric step
 * (src/q-gcm.F:1262-1268).  State and inputs move through the calls above under the ocean's names:
 *   set_/get_state     pa, pam, qa, qam (nxpa,nypa,nla)                        MODULE atstate
 *   set_forcing        wekpa, entat (nxpa,nypa), xan(nla-1)                    atstate / athomog (written by xforc / aml)
 *   set_cyc_forcing    txisat, txinat, enisat(nla-1), eninat(nla-1)            athomog
 *   set_/get_scalars   dpiat, dpiatp (nla-1), atmcs, atmcn, atmcsp, atmcnp (nla)
 *   set_homog_cyc      pch1at, pch2at, pbhat, aipcha, hc1sat, hc2sat, hc1nat, hc2nat, hbsiat, aipbha
 *   set_grid           yparel(nypa), bd2at(nxta) [FFTPACK order, src/q-gcm.F:961-970], ddynat(nxpa,nypa)
 *   helmholtz          hscyat (src/atisubs.F:298-395) for homsol (src/conhoms.F:678-679)
 *   lf_average         atmospheric half of the averaging block, src/q-gcm.F:1370-1404
 *   steps(nt0, n)      n atmospheric steps nt = nt0.., averaging after the steps with mod(nt-1,100) == 0 */
int qgcm_hip_qgastep(qgcm_hip_handle h);    /* replaces "call qgastep"          q-gcm.F:1262 */
int qgcm_hip_atinvq(qgcm_hip_handle h);     /* replaces "call atinvq"           q-gcm.F:1265 */
int qgcm_hip_atqzbd(qgcm_hip_handle h);     /* replaces "call atqzbd (qa, pa)"  q-gcm.F:1268 */
/* boundary line sums of the last qgastep / cyclic qgostep, as the reference leaves them in MODULE athomog /
 * ochomog: b = ajis, ajin, ap5s, ap5n (nlo each).  They are formed by the constraint kernel, i.e. valid after
 * the atinvq / ocinvq that follows the step.  Synchronous. */
int qgcm_hip_get_bsums(qgcm_hip_handle h, double *b);
/* A coupled run with the forcing held between calls (xforc / oml / aml stay with the host):
 * atmospheric steps nt = nt0 .. nt0+n-1 on `atm`, and on `oc` one ocean step before every atmospheric step with
 * mod(nt,nstr) == 1 (src/q-gcm.F:1220-1268), each with its own averaging rule.  The two handles run on
 * their own HIP streams, so the small atmospheric kernels overlap the ocean's. Either handle may be NULL. */
int qgcm_hip_coupled_steps(qgcm_hip_handle oc, qgcm_hip_handle atm, int nt0, int n, int nstr);
/* Two handles that step side by side on one GPU (the ocean and the atmosphere under qgcm_hip_coupled_steps) can be
 * given disjoint ranges of compute units: the handle's HIP stream is replaced by one that may use the CUs
 * first .. first+count-1 only (count = 0: all of them again).  Without it the atmosphere's small dependent launches
 * queue behind the ocean's chip-filling ones.  Not for handles with a communicator.  Synchronous; no reference
 * counterpart (the reference runs both halves on the same OpenMP threads, src/q-gcm.F:1220-1268). */
int qgcm_hip_set_cu_range(qgcm_hip_handle h, int first, int count);

/* Helmholtz solve for homsol: wrk(nxpo,nypo) in/out, boc(nxto)
 * (replaces hsbxoc / hscyoc, src/ocisubs.F:415-618). Synchronous. */
int qgcm_hip_helmholtz(qgcm_hip_handle h, double *wrk, const double *boc);

/* ---- y-slab building blocks (multi-GPU; one handle per slab) -------------
 * A distributed step is: qgostep | row_transform(0) | thomas_phase 1, exchange,
 * 2 | constr | row_transform(1) | unpack | halo_pack, exchange, halo_unpack.
 * Two exchanges per step. All buffers named *_dev are DEVICE pointers owned by
 * the caller (e.g. torch tensors used with torch.distributed); every call is
 * asynchronous on the handle's stream. */
int qgcm_hip_local_rows(qgcm_hip_handle h, int *nyl, int *joff, int *jlo, int *jhi);
int qgcm_hip_row_transform(qgcm_hip_handle h, int inverse);
/* number of doubles of one per-step slab summary message: 3 * nlo * ldw (+ the boundary line sums of a cyclic ocean,
 * + 3 sums of the mixed layer once qgcm_hip_oml_init was called - query after it) */
int qgcm_hip_thomas_msg_len(qgcm_hip_handle h);
/* The right-hand-side independent part of the summaries (gains and unit-response sums, 4 * nlo * ldw doubles per
 * slab) is exchanged ONCE after qgcm_hip_set_grid: every rank copies its own with qgcm_hip_thomas_consts, the host
 * all-gathers them (rank-major) and hands the result to qgcm_hip_set_thomas_consts. qgcm_hip_comm_init does this
 * by itself for the library-issued exchanges; a lone slab (nranks = 1) needs nothing. */
int qgcm_hip_thomas_const_len(qgcm_hip_handle h);
int qgcm_hip_thomas_consts(qgcm_hip_handle h, double *dst_dev);
int qgcm_hip_set_thomas_consts(qgcm_hip_handle h, const double *gath_dev, int nranks);
/* phase 1: this slab's summary of the two y sweeps -> send_dev: per mode and wavenumber the zero-inflow
 *          end values of both sweeps and the zero-inflow column sum behind the area integrals
 *          (xintp, src/ocisubs.F:160) - 3 numbers.
 * phase 2: gath_dev = all ranks' phase-1 messages (rank-major) -> both sweeps finished, and the
 *          basin-wide area integrals known on every rank (bitwise the same). */
int qgcm_hip_thomas_phase(qgcm_hip_handle h, int phase, const double *gath_dev, double *send_dev,
                          int rank, int nranks);
/* mass-constraint solve (src/ocisubs.F:329-370) from the area integrals thomas_phase 2 left behind */
int qgcm_hip_constr(qgcm_hip_handle h);
int qgcm_hip_unpack(qgcm_hip_handle h, int fuse_ocqbdy);
/* halo messages: (3 rows of po + 1 row of qo) * nlo rows of ldx doubles each (+ 3 rows of sst with the mixed layer on) */
int qgcm_hip_halo_msg_len(qgcm_hip_handle h);
int qgcm_hip_halo_pack(qgcm_hip_handle h, double *to_lower_dev, double *to_upper_dev);
int qgcm_hip_halo_unpack(qgcm_hip_handle h, const double *from_lower_dev, const double *from_upper_dev);

/* homsol on y-slabs (src/conhoms.F:549-601 needs hsbxoc on the whole basin): the modal Helmholtz problems of a
 * step ARE homsol's (boc = bd2oc - rdm2oc(m)), so the same distributed solve serves - fill the work array with the
 * right-hand side 1 (qgcm_hip_wrk_fill), row_transform(0), thomas_phase 1 | exchange | 2, row_transform(1), read the
 * solutions of the local rows back (qgcm_hip_wrk_get: (nxpo, nyl, nlo) block, walls and halo rows zero) and their
 * basin-wide area integrals dxo*dyo*xintp(wrk_m) (qgcm_hip_area_integrals, from the spectral column sums of
 * thomas_phase 2 / a whole-domain sweep; nlo doubles, synchronous).  No host-side solver is involved. */
int qgcm_hip_wrk_fill(qgcm_hip_handle h, double value);
int qgcm_hip_wrk_get(qgcm_hip_handle h, double *wrk);
/* rows of a (nxpo, nyl, nlo) host block into the work array (the counterpart of qgcm_hip_wrk_get; walls ignored):
 * with qgcm_hip_row_transform this exposes the row transforms that replace FFTPACK's dsint / drfftf / drfftb
 * (src/ocisubs.F:461-463, 494-499, 566-568, 601-605) by themselves - forward: dsint resp. drfftf, unnormalised, the
 * cyclic spectrum in FFTPACK's half-complex order; inverse: dsint resp. drfftb.  Synchronous. */
int qgcm_hip_wrk_set(qgcm_hip_handle h, const double *wrk);
int qgcm_hip_area_integrals(qgcm_hip_handle h, double *xin);

/* One call per communication-free stage of a distributed step (fewer host round trips):
 *   stage 1: qgostep, row_transform(0), thomas_phase(1)                          a = summary send buffer
 *   stage 2: thomas_phase(2), constr, row_transform(1), unpack(+ocqbdy), halo_pack
 *                                                      a = summary gather buffer, b/c = halo to-lower/to-upper
 *   stage 3: halo_unpack, optional lf_average (flags & 1)                        a/b = halo from-lower/from-upper
 *   stage 4 + stage 5 = stage 1 in two parts, for drivers that overlap the halo exchange with compute: stage 4 is the
 *            part of the tendency launch that reads no halo row (all but the first and last 16-row tile row; it may
 *            run before stage 3 of the previous step), stage 5 the rest of stage 1 (a = summary send buffer).  Needs
 *            at least three tile rows per slab and the mixed layer off.
 *   stage 6 + stage 7 = stage 5 in two parts: stage 6 the outer tile rows + edge work of the tendency launch alone (it
 *            may run on another stream BESIDE stage 4 - disjoint tiles - as soon as the halo rows are in: what
 *            qgcm_hip_slab_steps does on the exchange's stream), stage 7 the forward rows and the summary sweep (after
 *            stages 4 and 6; a = summary send buffer in both).
 * With the ocean mixed layer on the device (qgcm_hip_oml_init on every slab, before the buffers are sized) `call oml`
 * (src/q-gcm.F:1232) runs before stage 1, in two halves around ONE more all-gather of qgcm_hip_oml_msg_len() doubles
 * per rank (the mean entrainment is a basin-wide number, src/omlsubs.F:153):
 *   stage 10: new sst, raw entrainment, this slab's sums                         a = send buffer (3 doubles)
 *   stage 11: entoc from entrainment minus mean, sst buffer rotation             a = gathered sums (3 * nranks)
 * xon(1) and the boundary line integrals of entoc then travel at the end of the stage-1 message, the edge rows of sst
 * at the end of the halo messages. */
int qgcm_hip_oml_msg_len(qgcm_hip_handle h);
int qgcm_hip_slab_stage(qgcm_hip_handle h, int stage, double *a_dev, double *b_dev, double *c_dev,
                        int rank, int nranks, int flags);

/* The same distributed step with the exchanges issued by the library itself: RCCL calls on the
 * handle's stream between the four stages, no host language in the loop. RCCL is bound at run
 * time (dlopen), a single-GPU process never loads it.
 *   qgcm_hip_comm_unique_id : rank 0 obtains the QGCM_HIP_COMM_ID_BYTES-byte rendezvous id and hands
 *                             it to the other ranks by whatever the host has (MPI_Bcast, torch.distributed)
 *   qgcm_hip_comm_init      : collective over the nranks handles (one process per GPU); rank r must own the
 *                             r-th slab (slab_g0/slab_g1 of qgcm_hip_params); allocates the exchange buffers
 *   qgcm_hip_slab_steps     : n whole steps from step s0 (collective). Per step: all-gather of the slab
 *                             summaries (3*nlo*ldw doubles) and the edge rows (3 of po + 1 of qo per layer) to
 *                             both neighbours as one all-gather, or as send/recv with QGCM_HIP_HALO_P2P=1.
 *                             QGCM_HIP_SLAB_GRAPH=1 replays 50-step HIP graphs that contain the collectives.
 * The communicator is released by qgcm_hip_destroy. */
#define QGCM_HIP_COMM_ID_BYTES 128
int qgcm_hip_comm_unique_id(char *id, int nbytes);
int qgcm_hip_comm_init(qgcm_hip_handle h, const char *id, int nbytes, int rank, int nranks);
int qgcm_hip_slab_steps(qgcm_hip_handle h, int s0, int n);
/* edge rows as grouped send/recv with the two neighbours (1) or as one all-gather (0); collective:
 * every rank must make the same choice */
int qgcm_hip_comm_set_halo_p2p(qgcm_hip_handle h, int on);
/* on = 1: the halo exchange of step s (and the halo unpack) run on a second stream while the handle's stream already
 * computes the tile rows of step s+1's tendency launch that need no halo row (stage 4); the outer tile rows follow
 * the halo rows on that second stream (stage 6), the handle's stream waits for them and goes on (stage 7).  Bitwise the same results.  Steps followed by a leapfrog averaging, the mixed layer and slabs of fewer
 * than three 16-row tile rows keep the plain order.  Collective choice, like the one above. */
int qgcm_hip_comm_set_overlap(qgcm_hip_handle h, int on);
/* measurement aid (collective): the step's exchanges back to back, microseconds each:
 * us[0] summaries all-gather, us[1] halo rows as all-gather, us[2] halo rows as send/recv */
int qgcm_hip_comm_probe(qgcm_hip_handle h, int reps, double *us);

/* ---- ocean mixed layer -----------------------------------
 * `call oml` (src/q-gcm.F:1232; body src/omlsubs.F:47-236 + omladf 244-763) on the device: steps the
 * mixed-layer temperature on the T grid (nxto,nyto) = (nxpo-1,nypo-1), and produces what the PV path
 * consumes - entoc on the p grid, xon(1) and (cyclic) enisoc(1)/eninoc(1) - without leaving the GPU.
 * Only for a handle that owns the whole domain. */
typedef struct qgcm_hip_oml_params {
  double hmoc;         /* mixed layer thickness                 (MODULE intrfac, input.params) */
  double toc1, toc2;   /* toc(1), toc(2)                        (MODULE occonst) */
  double st2d, st4d;   /* Del-sqd / Del-4th sst diffusivities   (MODULE intrfac) */
  double ycexp;        /* sst advection coupling coefficient    (MODULE occonst) */
  double rrcpoc;       /* 1/(rhooc*cpoc), src/q-gcm.F:438       (MODULE radiate) */
  double tsbdy, tnbdy; /* boundary temperatures of the options below */
  int sb_hflux;        /* the reference's cpp options sb_hflux / nb_hflux as run-time flags */
  int nb_hflux;
} qgcm_hip_oml_params;
/* allocates the mixed-layer state and switches it on: qgcm_hip_steps then runs oml before qgostep in
 * every step and averages sst with the other fields (src/q-gcm.F:1345-1351).  On a y-slab handle (arrays = local rows
 * incl. halos, T row j between p rows j and j+1) call it before qgcm_hip_comm_init / before sizing the message buffers;
 * qgcm_hip_slab_steps / the slab stages 10, 11 then step it. */
int qgcm_hip_oml_init(qgcm_hip_handle h, const qgcm_hip_oml_params *p);
/* sst, sstm (MODULE intrfac), dense (nxto,nyto) Fortran order; NULL = leave unchanged / do not fetch */
int qgcm_hip_oml_set_state(qgcm_hip_handle h, const double *sst, const double *sstm);
int qgcm_hip_oml_get_state(qgcm_hip_handle h, double *sst, double *sstm);
/* fnetoc(nxto,nyto) (intrfac), wekto(nxto,nyto) (ocstate), tauxo, tauyo(nxpo,nypo) (intrfac); NULL = unchanged */
int qgcm_hip_oml_set_forcing(qgcm_hip_handle h, const double *fnetoc, const double *wekto,
                             const double *tauxo, const double *tauyo);
int qgcm_hip_oml(qgcm_hip_handle h);          /* replaces "call oml", src/q-gcm.F:1232 */
/* entoc(nxpo,nypo) (or NULL) and diag[5] = xon(1), cfraoc, centoc, enisoc(1), eninoc(1); synchronous */
int qgcm_hip_oml_get_diag(qgcm_hip_handle h, double *entoc, double *diag);

/* ---- validity scan -----------------------------------------
 * Ocean part of "call valids (solnok)" (src/q-gcm.F:1278; src/valsubs.F:272-527) on the device: instead of
 * pulling po, qo (44 MB at 5 km) every valday, 14 + nlo doubles and the verdict come back.
 *   out[0..13]  min, max of po, qo, sst, wekto, full layer thickness top / intermediate / bottom
 *   out[14..]   hfbad(1..nlo): per cent of the basin where layer k is thinner than thkmin = 100 m
 *               (evaluated, as in the reference, only when some thickness is <= thkmin; else 0)
 *   *solnok     0 if |po| >= 1e4, |qo| >= 0.05, |sst| >= 75, |wekto| >= 1e-3 or hfbad(k) > 20 (the reference's
 *               limits, src/valsubs.F:78-97), else 1.  sst / wekto are scanned when the mixed layer is
 *               initialised (their entries stay at +/-1e30 otherwise).
 * Bitwise the reference's numbers (min / max / quarter-integer sums are order independent). The
 * neighbourhood print-out of a failing run stays on the host (pull the state, call the reference's valids).
 * qgcm_hip_set_dtopoc: bottom topography dtopoc(nxpo,nypo) of MODULE occonst (NULL = flat). Synchronous. */
int qgcm_hip_set_dtopoc(qgcm_hip_handle h, const double *dtopoc);
int qgcm_hip_valids(qgcm_hip_handle h, double *out, int *solnok);

/* ---- start-up / restart arithmetic and the progress sample on the device ------------
 * qgcm_hip_init_from_p: the start-up sequence of the main program (src/q-gcm.F:711-731; atmosphere :738-749) from
 *   the po, pom ALREADY on the device (qgcm_hip_set_state with qo = qom = NULL, e.g. after a restart read):
 *   constr (dpioc, dpiocp and, cyclic / atmosphere, ocncs, ocncn, ocncsp, ocncnp: src/conhoms.F:93-300), qcomp
 *   (src/vorsubs.F:49-138), ocqbdy / atqzbd, merqcy (src/vorsubs.F:142-239) for both time levels.  qcomp / merqcy /
 *   ocqbdy are bitwise the reference; the constraint integrals agree to rounding (parallel sums).
 * qgcm_hip_wekpo_from_tau: ocean-only Ekman pumping from the wind stress tauxo, tauyo (nxpo,nypo) - wekto on the
 *   T grid and its p-grid average wekpo (src/xfosubs.F:138, 566-645) - into the forcing the path reads (and
 *   into the mixed layer's wekto / stress once qgcm_hip_oml_init was called).  Synchronous.
 * qgcm_hip_prsamp: the ocean numbers of the progress print-out prsamp (src/q-gcm.F:1933-2066) without pulling the
 *   state: out = po(k), qo(k) at the basin centre ((nxpo+1)/2, (nypo+1)/2), the layer averages pavgoc(k), qavgoc(k)
 *   (src/monitor_diag.F:729-739), then min, max of sst (+-1e30 without the device mixed layer): 4*nlo + 2 doubles.
 *   Synchronous. */
int qgcm_hip_init_from_p(qgcm_hip_handle h);
int qgcm_hip_wekpo_from_tau(qgcm_hip_handle h, const double *tauxo, const double *tauyo);
int qgcm_hip_prsamp(qgcm_hip_handle h, double *out);

/* ---- measurement -------------------------------------------------------- */
/* Runs n steps like qgcm_hip_steps and returns the HIP-event time (ms) of
 * the whole region, measured on the handle's stream. */
int qgcm_hip_time_steps(qgcm_hip_handle h, int s0, int n, float *ms);
/* Captures, instantiates and uploads the HIP graphs qgcm_hip_steps(s0, n) will replay (50-step blocks + one block
 * for the even part of the remainder) without running a step, so that a caller timing a window with its own clock
 * keeps graph construction outside it.  Synchronous; the state is untouched. */
int qgcm_hip_prepare_steps(qgcm_hip_handle h, int s0, int n);
/* Runs n steps eagerly with HIP events around every kernel launch and
 * accumulates per-kernel totals: ms[i], launches[i] for i < *nk (in: capacity,
 * out: number of kernel slots).  names[i] points to static strings. */
int qgcm_hip_profile_steps(qgcm_hip_handle h, int s0, int n, double *ms, int *launches,
                           const char **names, int *nk);
/* device copy bandwidth probe (GB/s of read+write traffic) used as the
 * "measured peak" beside the nominal 8 TB/s. */
int qgcm_hip_copy_bandwidth(qgcm_hip_handle h, size_t bytes, int reps, double *gbps);
/* Rate (GB/s of read + written bytes) of a pure streaming kernel that reads nr fields and writes nw fields of
 * field_bytes each, 16 bytes per lane: the practical ceiling for a kernel of that read : write mix on buffers of
 * that size (15:6 = the tendency kernel, 3:3 = row transforms / Thomas sweep, 5:3 = fused inverse rows). */
int qgcm_hip_stream_mix_bandwidth(qgcm_hip_handle h, int nr, int nw, size_t field_bytes, int reps, double *gbps);
/* HIP stream of the handle as an opaque pointer (hipStream_t). */
void *qgcm_hip_stream(qgcm_hip_handle h);

#ifdef __cplusplus
}
#endif
#endif /* QGCM_HIP_H */
