// qgcm_hip.hip - C ABI (include/qgcm_hip.h) over the gfx950 kernels.
//
// One HIP stream per handle; state stays resident in HBM; the q and p time
// levels rotate between two buffers each instead of being copied
// (reference: qom<-qo in src/qgosubs.F:201-206, pom<-po in src/ocisubs.F:392).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>

#include "qgcm_dev.h"
#include "k_tend.h"
#include "k_dst.h"
#include "k_fft3.h"
// the row lengths with a three-stage plan: NAtl 1 km (4800), SOcn 5 km (4608), and two more for the tests
#ifndef FFT3_NT
#define FFT3_NT 256
#endif
// (4800 = 12 * 20 * 20: the cheap radix takes the stage with two rounds of butterflies - profiles/r4_fft_plan_4800.log)
#ifdef FFT3_4800_OLD // (A/B: the round-2 plan of the 4800-point rows)
#define QG_FFT3_PLANS(X) X(1, 15, 16, 20) X(2, 16, 16, 18) X(3, 16, 16, 16) X(4, 12, 15, 16)
#else
#define QG_FFT3_PLANS(X) X(1, 12, 20, 20) X(2, 16, 16, 18) X(3, 16, 16, 16) X(4, 12, 15, 16)
#endif
#include "k_dst64.h"
#include "k_thomas.h"
#include "k_misc.h"
#include "k_cyclic.h"
#include "k_rfft64.h"
#include "k_fft3_unpack.h"
#include "k_oml.h"
#include "k_valids.h"
#include "k_setup.h"
#include "slab_comm.h"

// kernels that are templates on the number of layers: instantiated for nlo = 2 .. QGCM_HIP_MAXL (8); the fused fast
// paths (k_dst64_unpack, k_rfft64_unpack, k_rfft3_unpack, the riding constraint solves) stay with nlo <= 4
static_assert(QG_MAXL == 8, "QG_SWITCH_NL lists the cases");
#define QG_SWITCH_NL(nl, X, what)                    \
  switch (nl) {                                      \
    case 2: X(2); break;                             \
    case 3: X(3); break;                             \
    case 4: X(4); break;                             \
    case 5: X(5); break;                             \
    case 6: X(6); break;                             \
    case 7: X(7); break;                             \
    case 8: X(8); break;                             \
    default: QG_FAIL(what ": unsupported nlo");      \
  }

static thread_local char g_err[512] = "";

#define QG_FAIL(...)                             \
  do {                                           \
    snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
    return 1;                                    \
  } while (0)

#define HIPCHECK(expr)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) QG_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

enum { KN_TEND = 0, KN_BSUMS, KN_DSTF, KN_THOMAS, KN_DSTI, KN_CONSTR, KN_UNPACK, KN_OCQBDY, KN_LFAVG, KN_OML, KN_OML_ENTOC, KN_NOOP, KN_NOOP_TRAIN, KN_COUNT };
// (k_oml = k_oml_step, the sst step + raw entrainment; k_oml_entoc = the entrainment on the p grid)
static const char *kKernelNames[KN_COUNT] = {"k_tend",   "k_cyc_bsums", "k_dst_fwd", "k_thomas", "k_dst_inv",
                                             "k_constr", "k_unpack",  "k_ocqbdy", "k_lf_average", "k_oml", "k_oml_entoc", "k_noop", "k_noop_train"};

// Device copy of the Thomas pivot tables of one set of diagonals (see QgThomasParams / build_pivots).
struct QgThomasTab {
  double *binf = nullptr, *ptab = nullptr;
  int *rcb = nullptr, *poff = nullptr;
  size_t binf_n = 0, ptab_n = 0, rcb_n = 0, poff_n = 0; // capacities (elements)
};

struct qgcm_hip_ctx {
  qgcm_hip_params prm;
  QgGeom g;
  int device;
  hipStream_t stream;
  double *p[2], *q[2];
  int ip, iq; // p[ip] = po, p[ip^1] = pom ; q[iq] = qo, q[iq^1] = qom
  double *wekpo, *entoc, *ddynoc, *ochom, *yporel;
  double *rspl = nullptr; // sponge-layer ramp r_spl (qgcm_hip_set_sponge), or nullptr
  double c1_spl = 0.0;
  double *wrk, *rowsum;
  double *ybnd = nullptr;                  // cyclic y-slabs: (2, nl) zonal-mean solution next to the zonal boundaries (k_thomas PHASE 2)
  const double *slab_gath = nullptr;       // cyclic y-slabs: the gathered step messages of the last thomas phase 2
  int slab_nranks = 1;
  const double *oml_gath = nullptr;        // y-slabs: the gathered sums of the mixed layer's stage 10 (3, nranks)
  double *bpart_out = nullptr;             // where k_tend's extra workgroups put the boundary line sums (bpart, or the tail of the step message)
  QgThomasTab tt, tt_tmp;                  // Thomas pivot tables of the modal solves / of the last qgcm_hip_helmholtz
  // y-slabs: the slab's responses to unit inflows from below / above, tabulated as far as they reach (k_thomas_corr)
  struct {
    double *ptab = nullptr, *qtab = nullptr;
    int *meta = nullptr; // np, nq, poff, qoff: (nblk, nl) each
    size_t prows = 0, qrows = 0;
  } corr;
  double *slabDE;                          // y-slab summary constants (D, E, SP, SQ) per (mode, wavenumber)
  double *th_cgath = nullptr;              // all ranks' slabDE (rank-major), exchanged once (qgcm_hip_set_thomas_consts)
  int th_cgath_ranks = 0;
  double *ksum, *wcot;                     // spectral column sums of the solution and their cot weights (k_thomas.h)
  double *bpart;                           // cyclic: partial boundary line sums (k_tend's extra workgroups -> constraint algebra)
  QgConstrParams *d_boxq = nullptr;        // box: device copy of the constraint parameters (generic inverse rows' extra workgroup)
  QgCycConstrParams *d_cycq = nullptr;     // cyclic: device copy of the constraint parameters (fused step path)
  int thR;                                 // rows per chunk of the Thomas kernel
  double *pch1, *pch2, *pbh;
  QgScalars *sc;
  double2 *twid;
  double *sintab;
  int fftN, nfac, fac[QG_MAXFAC];
  QgConstr cs;
  bool grid_set, homog_set;
  bool geom_set = false; // yporel / ddynoc are on the device (qgcm_hip_set_geometry or set_grid)
  bool whole; // the handle owns the whole domain (no y-slab neighbours)
  bool dst_single = false; // generic row kernels run single-buffer (in-place) stages
  int fft3 = 0;            // long rows: three-stage register-radix plan of k_fft3.h (0 = none; index into QG_FFT3_PLANS)
  size_t fft3_lds = 0;
  bool force_generic_dst; // QGCM_HIP_GENERIC_DST=1: use the generic Stockham row kernel (A/B + tests)
  bool no_fused_unpack;   // QGCM_HIP_NO_FUSED_UNPACK=1: separate inverse transform and unpack launches (A/B + tests)
  int cu_first = 0, cu_count = 0; // qgcm_hip_set_cu_range: the CUs this handle's stream may use (0: all)
  bool no_fused_avg;      // QGCM_HIP_NO_FUSED_AVG=1: leapfrog averaging as a launch of its own inside qgcm_hip_steps (A/B + tests)
  bool avg_now = false;   // one_step: this step's kernels store the averaged time level (k_tend, k_dst64_unpack)
  bool no_fused_constr;   // QGCM_HIP_NO_FUSED_CONSTR=1: keep the k_constr_box launch inside qgcm_hip_steps (A/B + tests)
  bool tend_wide;         // QGCM_HIP_TEND_WIDE=1: the tendency kernel's HBM-bound instantiation (32-wide tiles, plain stores) at any size (tests)
  std::vector<double> bd2oc;
  // profiling
  hipError_t timer_err = hipSuccess;
  int timer_err_at = 0;
  bool profiling;
  hipEvent_t ev0, ev1;
  std::vector<hipEvent_t> evpool; // pairs, drained once per step (no host sync between kernels)
  std::vector<int> evkid;
  double kms[KN_COUNT];
  int klaunch[KN_COUNT];
  // graphs keyed by (ip, iq, phase)
  struct GraphEntry { hipGraphExec_t exec; unsigned long used; }; // used: the steps_impl call that last replayed it
  std::map<long long, GraphEntry> graphs;
  unsigned long graph_call = 0;
  int avg_period = 25; // time levels are averaged after steps s with (s-1) mod avg_period == 0: ocean 25, atmosphere 100
  // ocean mixed layer (qgcm_hip_oml_init): three rotating sst buffers (is = sst, ism = sstm, spare = 3-is-ism)
  struct {
    bool on = false;
    qgcm_hip_oml_params prm;
    int ldt = 0, is = 0, ism = 1, nblkA = 0, nblkB = 0;
    double *sst[3] = {nullptr, nullptr, nullptr};
    double *fnet = nullptr, *wekto = nullptr, *xfo = nullptr, *taux = nullptr, *tauy = nullptr;
    double *partA = nullptr, *partB = nullptr, *diag = nullptr;
  } oml;
  // validity scan (qgcm_hip_valids): partials, results, optional bottom topography
  double *val_part = nullptr, *val_out = nullptr, *dtopoc = nullptr;
  double *area_part = nullptr, *area_out = nullptr; // trapezoid area integrals of po, pom, qo (k_setup.h)
  // y-slab exchanges over RCCL (qgcm_hip_comm_init); slab-step graphs keyed like `graphs`
  QgSlabComm *sc_comm = nullptr;
  std::map<int, hipGraphExec_t> slab_graphs;
  bool slab_graph_mode = false; // QGCM_HIP_SLAB_GRAPH=1: capture 50 distributed steps (collectives included)
  size_t dst_lds;
};

static const int kGraphBlock = 50; // y-slab step graphs (qgcm_hip_slab_steps)

extern "C" const char *qgcm_hip_last_error(void) { return g_err; }
extern "C" int qgcm_hip_abi_version(void) { return QGCM_HIP_ABI_VERSION; }

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Captured step graphs hold kernel parameters by value and raw device pointers: every call that changes
// either (tables, homogeneous solutions, mixed-layer parameters, slab constants) drops them.
// The stream is drained even when no graph exists: the callers go on to overwrite device tables with blocking copies
// on the null stream, which the handle's non-blocking stream does not wait for - eager steps still in flight would
// read half-updated tables (all callers are set-up calls: the wait costs nothing that matters).
static void drop_graphs(qgcm_hip_ctx *c) {
  (void)hipStreamSynchronize(c->stream);
  if (c->graphs.empty() && c->slab_graphs.empty()) return;
  for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second.exec);
  for (auto &kv : c->slab_graphs) hipGraphExecDestroy(kv.second);
  c->graphs.clear();
  c->slab_graphs.clear();
}

static int factorize(int n, int *fac) {
  int nf = 0;
  static const int tr[5] = {8, 4, 2, 3, 5}; // fewest passes through LDS (FFTPACK's own order is 4,2,3,5)
  for (int t = 0; t < 5; ++t)
    while (n % tr[t] == 0) {
      fac[nf++] = tr[t];
      n /= tr[t];
    }
  for (int p = 7; n > 1; p += 2)
    while (n % p == 0) {
      fac[nf++] = p;
      n /= p;
    }
  return nf;
}

// pitched host <-> device copies of (nx, rows) Fortran blocks
static int upload2d(qgcm_hip_ctx *c, double *dst, int ld, const double *src, int nx, long rows) {
  HIPCHECK(hipMemcpy2DAsync(dst, (size_t)ld * 8, src, (size_t)nx * 8, (size_t)nx * 8, (size_t)rows,
                            hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}
static int download2d(qgcm_hip_ctx *c, double *dst, const double *src, int ld, int nx, long rows) {
  HIPCHECK(hipMemcpy2DAsync(dst, (size_t)nx * 8, src, (size_t)ld * 8, (size_t)nx * 8, (size_t)rows,
                            hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

static int dalloc(double **p, size_t n) {
  HIPCHECK(hipMalloc((void **)p, n * sizeof(double)));
  // The zero fill runs on the NULL stream, and a device memset need not have finished when hipMemset returns; the
  // handle's stream is non-blocking (not ordered against the null stream), so work queued on it right after an
  // allocation - the copy in qgcm_hip_set_thomas_consts, the all-gather in qgcm_hip_comm_init - could be overtaken
  // by the fill and zeroed. Seen as wrong slab constants with three one-process-per-slab ranks. Wait for the fill.
  HIPCHECK(hipMemset(*p, 0, n * sizeof(double)));
  HIPCHECK(hipStreamSynchronize(nullptr));
  return 0;
}

extern "C" int qgcm_hip_create(qgcm_hip_handle *h, const qgcm_hip_params *prm, int device) {
  if (!h || !prm) QG_FAIL("qgcm_hip_create: null argument");
  *h = nullptr;
  if (prm->nlo < 2 || prm->nlo > QG_MAXL) QG_FAIL("qgcm_hip_create: nlo=%d outside 2..%d", prm->nlo, QG_MAXL);
  if (prm->nxpo < 4 || prm->nypo < 4) QG_FAIL("qgcm_hip_create: grid too small");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) QG_FAIL("qgcm_hip_create: no HIP device (%s) - there is no CPU fallback", hipGetErrorString(e));
  if (device >= 0) HIPCHECK(hipSetDevice(device));
  qgcm_hip_ctx *c = new qgcm_hip_ctx();
  memset(&c->prm, 0, sizeof(c->prm));
  c->prm = *prm;
  HIPCHECK(hipGetDevice(&c->device));
  QgGeom &g = c->g;
  g.nx = prm->nxpo; g.nl = prm->nlo; g.cyc = prm->cyclic;
  g.atm = prm->atmos ? 1 : 0;
  if (g.atm && !g.cyc) QG_FAIL("qgcm_hip_create: the atmosphere is a zonally periodic channel (atmos = 1 needs cyclic = 1)");
  if (g.atm) {
    c->avg_period = 100;   // src/q-gcm.F:1370
    c->prm.delek = 0.0;    // no drag layer and no Del-4th term in qgastep / atadif
    for (int k = 0; k < QG_MAXL; ++k) c->prm.ah2oc[k] = 0.0;
  }
  g.nyg = prm->nypo;
  {
    // y-slab view: owned global rows g0..g1 (+ 3 halo rows towards each neighbour)
    int g0 = prm->slab_g0, g1 = prm->slab_g1;
    if (g0 == 0 && g1 == 0) { g0 = 1; g1 = g.nyg; }
    if (g0 < 1 || g1 > g.nyg || g1 - g0 + 1 < 4) QG_FAIL("qgcm_hip_create: bad slab %d..%d of %d rows (need >= 4 rows)", g0, g1, g.nyg);
    const int hlo = g0 > 1 ? 3 : 0, hhi = g1 < g.nyg ? 3 : 0;
    g.ny = (g1 - g0 + 1) + hlo + hhi;
    g.joff = g0 - hlo - 1;
    g.jlo = hlo + 1;
    g.jhi = hlo + (g1 - g0 + 1);
    g.jr0 = (g0 == 1) ? g.jlo + 1 : g.jlo;       // rows 2..nyg-1 of the global grid that this slab owns
    g.jr1 = (g1 == g.nyg) ? g.jhi - 1 : g.jhi;
    c->whole = (g0 == 1 && g1 == g.nyg);
    if (prm->atmos && !c->whole) QG_FAIL("qgcm_hip_create: y-slabs are implemented for the oceans only");
  }
  g.nxt = g.nx - 1;
  g.nk = g.cyc ? g.nxt : g.nxt - 1;
  g.ldx = round_up(g.nx, 16);
  g.ldw = round_up(g.nxt + 1, 16);
  g.fstride = (long)g.ldx * g.ny;
  g.wstride = (long)g.ldw * g.ny;
  // Layout invariants of the 16-byte write-through pair stores (qg_store16_wt / qg_pair_store_wt, qgcm_dev.h): pairs
  // start on 16-byte boundaries (even pitches and strides), an odd row length leaves a padding column for the last
  // pair's second half, and the work array has one behind the last coefficient.  They hold for the pitches chosen
  // above; a later change of the layout must not break them silently.  (Row padding then holds garbage: nothing may
  // read or reduce over columns >= nx resp. >= nk.)
  if (g.ldx % 2 || g.ldw % 2 || g.fstride % 2 || g.wstride % 2 || (g.nx % 2 && g.ldx <= g.nx) || g.ldw <= g.nk)
    QG_FAIL("qgcm_hip_create: row pitches ldx=%d ldw=%d break the alignment / padding rules of the paired stores", g.ldx, g.ldw);
  if ((double)g.fstride * g.nl >= 2147483647.0 || (double)g.wstride * g.nl >= 2147483647.0)
    QG_FAIL("qgcm_hip_create: %ld x %d elements per field exceed the kernels' 32-bit element offsets", g.fstride, g.nl);
  HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  const size_t F = (size_t)g.fstride, W = (size_t)g.wstride;
  for (int i = 0; i < 2; ++i) {
    if (dalloc(&c->p[i], F * g.nl)) return 1;
    if (dalloc(&c->q[i], F * g.nl)) return 1;
  }
  c->ip = c->iq = 0;
  if (dalloc(&c->wekpo, F) || dalloc(&c->entoc, F) || dalloc(&c->ddynoc, F)) return 1;
  if (dalloc(&c->ochom, F * (g.nl - 1))) return 1;
  if (dalloc(&c->yporel, g.ny)) return 1;
  if (dalloc(&c->wrk, W * g.nl)) return 1;
  if (dalloc(&c->slabDE, (size_t)4 * g.ldw * g.nl)) return 1;
  if (g.cyc && dalloc(&c->ybnd, 2 * QG_MAXL)) return 1;
  if (dalloc(&c->ksum, (size_t)g.ldw * g.nl)) return 1;
  if (dalloc(&c->wcot, (size_t)g.ldw)) return 1;
  if (dalloc(&c->bpart, (size_t)5 * BSUM_NB * 2 * g.nl)) return 1;
  if (dalloc(&c->rowsum, (size_t)g.ny * g.nl)) return 1;
  if (dalloc(&c->pch1, (size_t)g.ny * g.nl) || dalloc(&c->pch2, (size_t)g.ny * g.nl) || dalloc(&c->pbh, g.ny)) return 1;
  HIPCHECK(hipMalloc((void **)&c->sc, sizeof(QgScalars)));
  HIPCHECK(hipMemset(c->sc, 0, sizeof(QgScalars)));
  c->twid = nullptr;
  c->sintab = nullptr;
  c->grid_set = c->homog_set = false;
  {
    const char *e = getenv("QGCM_HIP_GENERIC_DST");
    c->force_generic_dst = e && e[0] == '1';
    const char *f = getenv("QGCM_HIP_NO_FUSED_UNPACK");
    c->no_fused_unpack = f && f[0] == '1';
    const char *fc = getenv("QGCM_HIP_NO_FUSED_CONSTR");
    c->no_fused_constr = fc && fc[0] == '1';
    const char *tw = getenv("QGCM_HIP_TEND_WIDE");
    c->tend_wide = tw && tw[0] == '1';
    const char *fa = getenv("QGCM_HIP_NO_FUSED_AVG");
    c->no_fused_avg = fa && fa[0] == '1';
  }
  c->profiling = false;
  HIPCHECK(hipEventCreate(&c->ev0));
  HIPCHECK(hipEventCreate(&c->ev1));
  memset(&c->cs, 0, sizeof(c->cs));
  HIPCHECK(hipDeviceSynchronize()); // the zero fills above ran on the null stream
  *h = c;
  return 0;
}

extern "C" int qgcm_hip_destroy(qgcm_hip_handle c) {
  if (!c) return 0;
  hipStreamSynchronize(c->stream);
  for (auto &kv : c->graphs) hipGraphExecDestroy(kv.second.exec);
  for (auto &kv : c->slab_graphs) hipGraphExecDestroy(kv.second);
  if (c->sc_comm) {
    QgSlabComm *m = c->sc_comm;
    if (m->comm) m->api->CommDestroy(m->comm);
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    if (m->ev_halo) hipEventDestroy(m->ev_halo);
    if (m->cstream) hipStreamDestroy(m->cstream);
    double *cb[] = {m->th_send, m->th_gath, m->h_send, m->h_gath, m->oml_send, m->oml_gath};
    for (double *p : cb)
      if (p) hipFree(p);
    delete m;
  }
  double *vp[] = {c->val_part, c->val_out, c->dtopoc, c->th_cgath, c->area_part, c->area_out, c->ybnd, c->rspl};
  for (double *p : vp)
    if (p) hipFree(p);
  double *omp[] = {c->oml.sst[0], c->oml.sst[1], c->oml.sst[2], c->oml.fnet, c->oml.wekto, c->oml.xfo,
                   c->oml.taux, c->oml.tauy, c->oml.partA, c->oml.partB, c->oml.diag};
  for (double *p : omp)
    if (p) hipFree(p);
  double *ptrs[] = {c->p[0], c->p[1], c->q[0], c->q[1], c->wekpo, c->entoc, c->ddynoc, c->ochom, c->yporel,
                    c->wrk,  c->slabDE, c->ksum, c->wcot, c->bpart, c->rowsum, c->pch1, c->pch2, c->pbh, c->sintab};
  for (double *p : ptrs)
    if (p) hipFree(p);
  if (c->twid) hipFree(c->twid);
  hipFree(c->sc);
  if (c->d_cycq) hipFree(c->d_cycq);
  if (c->d_boxq) hipFree(c->d_boxq);
  if (c->corr.ptab) hipFree(c->corr.ptab);
  if (c->corr.qtab) hipFree(c->corr.qtab);
  if (c->corr.meta) hipFree(c->corr.meta);
  for (QgThomasTab *t : {&c->tt, &c->tt_tmp}) {
    if (t->binf) hipFree(t->binf);
    if (t->ptab) hipFree(t->ptab);
    if (t->rcb) hipFree(t->rcb);
    if (t->poff) hipFree(t->poff);
  }
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->evpool) hipEventDestroy(e);
  hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

static int thomas_rows_per_chunk(int nrows) {
  const int need = (nrows + TH_NC - 1) / TH_NC;
  for (int r : {1, 2, 4, 8, 10, 12, 16, 20, 24, 32}) // rows per thread; 10 / 20: the 600- and 1200-row slabs of NAtl 1 km
    if (need <= r) return r;
  return -1;
}

// per-step message of one slab: the summaries of the two y sweeps; a cyclic ocean appends its boundary line sums
static inline size_t slab_msg_len(const QgGeom &g) {
  return (size_t)TH_MSG * g.nl * g.ldw + (g.cyc ? (size_t)5 * BSUM_NB * 2 * g.nl : 0);
}
// ... and, with the mixed layer on the device, the three sums of its entoc pass at the very end
static inline size_t slab_msg_len_oml(const qgcm_hip_ctx *c) { return slab_msg_len(c->g) + (c->oml.on ? 3 : 0); }
// halo message per direction: 3 rows of po + 1 row of qo per layer, + 3 rows of the new sst with the mixed layer on
static inline size_t halo_msg_len(const qgcm_hip_ctx *c) {
  return (size_t)4 * c->g.nl * c->g.ldx + (c->oml.on ? (size_t)3 * c->oml.ldt : 0);
}

// Thomas pivots, exactly the recurrence of src/ocisubs.F:472-477 (box) / 577-582 (cyclic), run once on the host.
// The recurrence reaches a bitwise fixed point after a few rows for most spectral indices; per block of TH_KW indices
// the pivots of the local rows before the block's slowest index is stationary are tabulated (rows of TH_KW doubles),
// plus the stationary pivot per index.  Appends the tables of one layer (set of diagonals) to T.
struct ThomasTabHost {
  std::vector<double> binf, ptab;
  std::vector<int> rcb, poff;
};
static void build_pivots(const QgGeom &g, double aoc, const double *boc /* per spectral index */, ThomasTabHost &T) {
  const int nr = g.jr1 - g.jr0 + 1;         // rows of this slab
  const int rg0 = g.jr0 + g.joff - 2;       // global interior-row index of the slab's first row
  const int nblk = (g.nk + TH_KW - 1) / TH_KW;
  const size_t b0 = T.binf.size();
  T.binf.resize(b0 + g.ldw, 0.0);
  std::vector<double> piv((size_t)TH_KW * nr);
  for (int bx = 0; bx < nblk; ++bx) {
    int rcb = 1; // at least one row, so that the kernel's clamped table read is always in bounds
    std::fill(piv.begin(), piv.end(), 0.0);
    for (int kk = 0; kk < TH_KW; ++kk) {
      const int k = bx * TH_KW + kk;
      if (k >= g.nk) continue;
      double betinv = 1.0 / boc[k]; // global interior row 0
      int gconv = rg0 + nr;         // first global row whose pivot equals its predecessor's (bitwise)
      for (int rg = 0; rg < rg0 + nr; ++rg) {
        if (rg > 0) {
          double gam = aoc * betinv;
          double nb = 1.0 / (boc[k] - aoc * gam);
          if (nb == betinv && gconv == rg0 + nr) gconv = rg;
          betinv = nb;
        }
        if (rg >= rg0) piv[(size_t)(rg - rg0) * TH_KW + kk] = betinv;
      }
      T.binf[b0 + k] = betinv; // the stationary value (or, if never stationary, unused: rcb = nr then)
      int lc = gconv - rg0;
      lc = lc < 0 ? 0 : (lc > nr ? nr : lc);
      if (lc > rcb) rcb = lc;
    }
    T.rcb.push_back(rcb);
    T.poff.push_back((int)(T.ptab.size() / TH_KW));
    T.ptab.insert(T.ptab.end(), piv.begin(), piv.begin() + (size_t)rcb * TH_KW);
  }
}

// (re)allocates the device tables as needed and uploads T (blocking copies: T may die at the caller's scope exit)
static int upload_pivots(qgcm_hip_ctx *c, QgThomasTab &D, const ThomasTabHost &T) {
  auto fit = [&](void **p, size_t &cap, size_t n, size_t el) -> int {
    if (n <= cap && *p) return 0;
    if (*p) HIPCHECK(hipFree(*p));
    *p = nullptr;
    HIPCHECK(hipMalloc(p, n * el));
    cap = n;
    return 0;
  };
  HIPCHECK(hipStreamSynchronize(c->stream)); // no launch may still be reading the old tables
  if (fit((void **)&D.binf, D.binf_n, T.binf.size(), sizeof(double))) return 1;
  if (fit((void **)&D.ptab, D.ptab_n, T.ptab.size(), sizeof(double))) return 1;
  if (fit((void **)&D.rcb, D.rcb_n, T.rcb.size(), sizeof(int))) return 1;
  if (fit((void **)&D.poff, D.poff_n, T.poff.size(), sizeof(int))) return 1;
  HIPCHECK(hipMemcpy(D.binf, T.binf.data(), T.binf.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(D.ptab, T.ptab.data(), T.ptab.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(D.rcb, T.rcb.data(), T.rcb.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(D.poff, T.poff.data(), T.poff.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

static int launch_thomas(qgcm_hip_ctx *c, double *wrk, const QgThomasTab &tab, int nlayers, int phase,
                         const double *gath, double *send, int rank, int nranks, int layer0, hipStream_t st,
                         bool cyc_part_a = false);
static void fill_cyc_constr_params(qgcm_hip_ctx *c, QgCycConstrParams &Q);

static void fill_thomas_params(qgcm_hip_ctx *c, QgThomasParams &P, double *wrk, const QgThomasTab &tab, int nlayers, int layer0);

// Set-up of the y-slab solve (k_thomas.h): PHASE 4 / 5 give the slab's gain and response sums (slabDE) and leave the
// responses Pv / Qv to unit inflows in the work array; they are tabulated per block of TH_KW wavenumbers as far as
// they reach (above 1e-19 of the inflow) - towards a side without a neighbour nothing ever enters: no table.
static int build_slab_responses(qgcm_hip_ctx *c) {
  const QgGeom &g = c->g;
  const int nblk = (g.nk + TH_KW - 1) / TH_KW, nm = nblk * g.nl;
  const bool nb_lo = g.joff + g.jlo > 1, nb_hi = g.joff + g.jhi < g.nyg;
  if (!c->corr.meta) HIPCHECK(hipMalloc((void **)&c->corr.meta, sizeof(int) * 4 * nm));
  std::vector<int> meta((size_t)4 * nm, 0);
  QgThomasParams P;
  fill_thomas_params(c, P, c->wrk, c->tt, g.nl, 0);
  const double tol = 1.0e-19 * P.ftnorm; // three orders below the rounding of the inflow itself
  for (int side = 0; side < 2; ++side) { // 0: inflow from below (PHASE 4, Pv), 1: from above (PHASE 5, Qv)
    if (launch_thomas(c, c->wrk, c->tt, g.nl, 4 + side, nullptr, nullptr, 0, 1, 0, nullptr)) return 1;
    double *&tab = side ? c->corr.qtab : c->corr.ptab;
    size_t &rows = side ? c->corr.qrows : c->corr.prows;
    if (tab) HIPCHECK(hipFree(tab));
    tab = nullptr;
    rows = 0;
    int *d_reach = c->corr.meta + side * nm, *d_off = c->corr.meta + (2 + side) * nm;
    if (!(side ? nb_hi : nb_lo)) continue; // reach 0 everywhere
    hipLaunchKernelGGL(k_thomas_reach, dim3(nblk, g.nl), dim3(256), 0, c->stream, P, side, tol, d_reach);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(meta.data() + side * nm, d_reach, sizeof(int) * nm, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    size_t off = 0;
    for (int i = 0; i < nm; ++i) {
      meta[(2 + side) * nm + i] = (int)off;
      off += meta[side * nm + i];
    }
    rows = off;
    HIPCHECK(hipMalloc((void **)&tab, sizeof(double) * TH_KW * (off ? off : 1)));
    HIPCHECK(hipMemcpyAsync(d_off, meta.data() + (2 + side) * nm, sizeof(int) * nm, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_thomas_pack, dim3(nblk, g.nl), dim3(256), 0, c->stream, P, side, (const int *)d_reach, (const int *)d_off, tab);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(c->stream));
  }
  // (the sides without a neighbour: zero reach, zero offsets)
  HIPCHECK(hipMemcpyAsync(c->corr.meta, meta.data(), sizeof(int) * 4 * nm, hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_set_geometry(qgcm_hip_handle c, const double *yporel, const double *ddynoc) {
  if (!c || !yporel) QG_FAIL("qgcm_hip_set_geometry: null argument");
  const QgGeom &g = c->g;
  drop_graphs(c);
  HIPCHECK(hipMemcpy(c->yporel, yporel, sizeof(double) * g.ny, hipMemcpyHostToDevice));
  if (ddynoc) {
    if (upload2d(c, c->ddynoc, g.ldx, ddynoc, g.nx, g.ny)) return 1;
  }
  c->geom_set = true;
  return 0;
}

extern "C" int qgcm_hip_set_grid(qgcm_hip_handle c, const double *yporel, const double *bd2oc, const double *ddynoc) {
  if (!c || !yporel || !bd2oc) QG_FAIL("qgcm_hip_set_grid: null argument");
  const QgGeom &g = c->g;
  if (qgcm_hip_set_geometry(c, yporel, ddynoc)) return 1;
  c->bd2oc.assign(bd2oc, bd2oc + g.nxt);
  // Thomas diagonal + chunk-entry pivots per mode: boc = bd2oc - rdm2oc(m)   (src/ocisubs.F:148-150)
  c->thR = thomas_rows_per_chunk(g.jr1 - g.jr0 + 1);
  if (c->thR < 0) QG_FAIL("qgcm_hip_set_grid: %d rows per slab exceed the single-segment Thomas kernel (<= 2048)", g.jr1 - g.jr0 + 1);
  {
    ThomasTabHost T;
    std::vector<double> boc(g.nk);
    for (int m = 0; m < g.nl; ++m) {
      for (int k = 0; k < g.nk; ++k) boc[k] = bd2oc[k] - c->prm.rdm2oc[m];
      build_pivots(g, c->prm.aoc, boc.data(), T);
    }
    if (upload_pivots(c, c->tt, T)) return 1;
  }
  // FFT tables: complex length N = nxto (box DST-I of length nxto-1)
  const int N = g.nxt;
  c->fftN = N;
  c->nfac = factorize(N, c->fac);
  if (c->nfac > QG_MAXFAC) QG_FAIL("qgcm_hip_set_grid: too many FFT factors");
  std::vector<double2> tw(N);
  const double twopi = 6.28318530717958647692528676655900577;
  for (int t = 0; t < N; ++t) {
    double ang = -twopi * (double)t / (double)N;
    tw[t].x = cos(ang);
    tw[t].y = sin(ang);
  }
  std::vector<double> st(N / 2 + 2);
  const double pi = 3.14159265358979323846264338327950288;
  for (int k = 0; k <= N / 2 + 1; ++k) st[k] = 2.0 * sin((double)k * (pi / (double)N)); // dsinti.f:20-24
  if (c->twid) hipFree(c->twid);
  if (c->sintab) hipFree(c->sintab);
  HIPCHECK(hipMalloc((void **)&c->twid, sizeof(double2) * N));
  HIPCHECK(hipMalloc((void **)&c->sintab, sizeof(double) * st.size()));
  HIPCHECK(hipMemcpy(c->twid, tw.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(c->sintab, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice));
  // generic row kernels: two ping-pong buffers, or ONE buffer with in-place stages for long rows
  c->dst_single = false;
  if (N >= DST_SINGLE_MINN && N <= DST_SINGLE_MAXN && N % 2 == 0 && !getenv("QGCM_HIP_NO_SINGLE_BUFFER")) {
    c->dst_single = true;
    for (int f = 0; f < c->nfac; ++f)
      if (c->fac[f] != 2 && c->fac[f] != 3 && c->fac[f] != 4 && c->fac[f] != 5 && c->fac[f] != 8) c->dst_single = false;
  }
  c->dst_lds = (size_t)(c->dst_single ? 1 : 2) * N * sizeof(cplx) + 2 * (g.cyc ? RFFT_NT : (N >= DST_BIG_N ? DST_NT_BIG : DST_NT)) * sizeof(double);
  if (c->dst_single) c->dst_lds = (size_t)N * sizeof(cplx) + 2 * (RFFT_NT / 64) * sizeof(double); // RFFT_NT == DST_NT_BIG
  if (c->dst_lds > 160 * 1024) QG_FAIL("qgcm_hip_set_grid: nxto=%d needs %zu B of LDS per row pair (> 160 KiB)", N, c->dst_lds);
  HIPCHECK(hipFuncSetAttribute((const void *)k_dst_box<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dst_lds));
  HIPCHECK(hipFuncSetAttribute((const void *)k_dst_box<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dst_lds));
  HIPCHECK(hipFuncSetAttribute((const void *)k_dst_box<false, DST_NT_BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dst_lds));
  HIPCHECK(hipFuncSetAttribute((const void *)k_rfft_cyc<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dst_lds));
  HIPCHECK(hipFuncSetAttribute((const void *)k_rfft_cyc<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->dst_lds));
  // long rows: the three-stage register-radix plans of k_fft3.h (QGCM_HIP_NO_FFT3=1: the Stockham plan, for A/B + tests)
  c->fft3 = 0;
  if (!getenv("QGCM_HIP_NO_FFT3")) {
#define QG_FFT3_SETUP(ID, R1, R2, R3)                                                                                        \
    if (N == R1 * R2 * R3) {                                                                                                 \
      typedef Fft3Plan<R1, R2, R3> PL;                                                                                       \
      c->fft3 = ID;                                                                                                          \
      c->fft3_lds = (size_t)PL::LDS_CPLX * sizeof(cplx) + 2 * (FFT3_NT / 64) * sizeof(double);                               \
      HIPCHECK(hipFuncSetAttribute((const void *)k_dst_box<false, FFT3_NT, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fft3_lds)); \
      HIPCHECK(hipFuncSetAttribute((const void *)k_rfft_cyc<false, PL, FFT3_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fft3_lds)); \
      HIPCHECK(hipFuncSetAttribute((const void *)k_rfft_cyc<true, PL, FFT3_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fft3_lds)); \
      HIPCHECK(hipFuncSetAttribute((const void *)k_rfft3_unpack<PL, 3, FFT3_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->fft3_lds)); \
    }
    QG_FFT3_PLANS(QG_FFT3_SETUP)
#undef QG_FFT3_SETUP
  }
  c->grid_set = true;
  // slab summary constants (gain, backward image and column sums of the unit responses), once
  if (build_slab_responses(c)) return 1;
  {
    // weights of the spectral area integral (k_thomas.h): sum_{i=1}^{n-1} 2 sin(k i pi/n) = 2 cot(k pi/2n), k odd
    std::vector<double> wc((size_t)g.ldw, 0.0);
    if (!g.cyc)
      for (int k = 1; k <= g.nk; k += 2) {
        const double h = (double)k * (pi / (2.0 * (double)N));
        wc[k - 1] = 2.0 * cos(h) / sin(h);
      }
    HIPCHECK(hipMemcpyAsync(c->wcot, wc.data(), wc.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

static int lu_factor(int n, double *a, int *piv) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double mx = fabs(a[k + n * k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i + n * k]) > mx) {
        mx = fabs(a[i + n * k]);
        p = i;
      }
    piv[k] = p;
    if (mx == 0.0) return k + 1;
    if (p != k)
      for (int j = 0; j < n; ++j) {
        double t = a[k + n * j];
        a[k + n * j] = a[p + n * j];
        a[p + n * j] = t;
      }
    for (int i = k + 1; i < n; ++i) {
      a[i + n * k] /= a[k + n * k];
      for (int j = k + 1; j < n; ++j) a[i + n * j] -= a[i + n * k] * a[k + n * j];
    }
  }
  return 0;
}

static void fill_constr_params(qgcm_hip_ctx *c, QgConstrParams &P);

extern "C" int qgcm_hip_set_homog_box(qgcm_hip_handle c, const double *ochom, const double *cdiffo, const double *cdhoc) {
  if (!c || !ochom || !cdiffo || !cdhoc) QG_FAIL("qgcm_hip_set_homog_box: null argument");
  if (c->g.cyc) QG_FAIL("qgcm_hip_set_homog_box: handle is cyclic");
  drop_graphs(c);
  const QgGeom &g = c->g;
  const int nl = g.nl, n1 = nl - 1;
  if (upload2d(c, c->ochom, g.ldx, ochom, g.nx, (long)g.ny * n1)) return 1;
  memcpy(c->cs.cdiffo, cdiffo, sizeof(double) * nl * n1);
  memcpy(c->cs.cdhoc, cdhoc, sizeof(double) * n1 * n1);
  memcpy(c->cs.cdhlu, cdhoc, sizeof(double) * n1 * n1);
  int info = lu_factor(n1, c->cs.cdhlu, c->cs.ipiv); // DGETRF, src/conhoms.F:627
  if (info) QG_FAIL("qgcm_hip_set_homog_box: cdhoc is singular (info=%d)", info);
  c->homog_set = true;
  {
    // device copy of the constraint parameters for the extra workgroup of the generic inverse rows
    QgConstrParams Q;
    fill_constr_params(c, Q);
    if (!c->d_boxq) HIPCHECK(hipMalloc((void **)&c->d_boxq, sizeof(Q)));
    HIPCHECK(hipMemcpy(c->d_boxq, &Q, sizeof(Q), hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int qgcm_hip_set_homog_cyc(qgcm_hip_handle c, const double *pch1oc, const double *pch2oc, const double *pbhoc,
                                      const double *aipcho, const double *hc1soc, const double *hc2soc, const double *hc1noc,
                                      const double *hc2noc, double hbsioc, double aipbho) {
  if (!c || !pch1oc || !pch2oc || !pbhoc || !aipcho || !hc1soc || !hc2soc || !hc1noc || !hc2noc)
    QG_FAIL("qgcm_hip_set_homog_cyc: null argument");
  if (!c->g.cyc) QG_FAIL("qgcm_hip_set_homog_cyc: handle is a box ocean");
  drop_graphs(c);
  const QgGeom &g = c->g;
  const int n1 = g.nl - 1;
  HIPCHECK(hipMemcpy(c->pch1, pch1oc, sizeof(double) * g.ny * n1, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(c->pch2, pch2oc, sizeof(double) * g.ny * n1, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(c->pbh, pbhoc, sizeof(double) * g.ny, hipMemcpyHostToDevice));
  for (int m = 0; m < n1; ++m) {
    c->cs.aipcho[m] = aipcho[m];
    c->cs.hc1soc[m] = hc1soc[m];
    c->cs.hc2soc[m] = hc2soc[m];
    c->cs.hc1noc[m] = hc1noc[m];
    c->cs.hc2noc[m] = hc2noc[m];
  }
  c->cs.hbsioc = hbsioc;
  c->cs.aipbho = aipbho;
  c->homog_set = true;
  {
    // device copy of the constraint parameters for the fused step path (k_thomas's extra workgroup, k_rfft64_unpack)
    QgCycConstrParams Q;
    fill_cyc_constr_params(c, Q);
    if (!c->d_cycq) HIPCHECK(hipMalloc((void **)&c->d_cycq, sizeof(Q)));
    HIPCHECK(hipMemcpy(c->d_cycq, &Q, sizeof(Q), hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int qgcm_hip_set_state(qgcm_hip_handle c, const double *po, const double *pom, const double *qo, const double *qom) {
  if (!c) QG_FAIL("qgcm_hip_set_state: null handle");
  const QgGeom &g = c->g;
  const long rows = (long)g.ny * g.nl;
  if (po && upload2d(c, c->p[c->ip], g.ldx, po, g.nx, rows)) return 1;
  if (pom && upload2d(c, c->p[c->ip ^ 1], g.ldx, pom, g.nx, rows)) return 1;
  if (qo && upload2d(c, c->q[c->iq], g.ldx, qo, g.nx, rows)) return 1;
  if (qom && upload2d(c, c->q[c->iq ^ 1], g.ldx, qom, g.nx, rows)) return 1;
  return 0;
}

extern "C" int qgcm_hip_get_state(qgcm_hip_handle c, double *po, double *pom, double *qo, double *qom) {
  if (!c) QG_FAIL("qgcm_hip_get_state: null handle");
  const QgGeom &g = c->g;
  const long rows = (long)g.ny * g.nl;
  if (po && download2d(c, po, c->p[c->ip], g.ldx, g.nx, rows)) return 1;
  if (pom && download2d(c, pom, c->p[c->ip ^ 1], g.ldx, g.nx, rows)) return 1;
  if (qo && download2d(c, qo, c->q[c->iq], g.ldx, g.nx, rows)) return 1;
  if (qom && download2d(c, qom, c->q[c->iq ^ 1], g.ldx, g.nx, rows)) return 1;
  return 0;
}

extern "C" int qgcm_hip_set_forcing(qgcm_hip_handle c, const double *wekpo, const double *entoc, const double *xon) {
  if (!c) QG_FAIL("qgcm_hip_set_forcing: null handle");
  const QgGeom &g = c->g;
  if (wekpo && upload2d(c, c->wekpo, g.ldx, wekpo, g.nx, g.ny)) return 1;
  if (entoc && upload2d(c, c->entoc, g.ldx, entoc, g.nx, g.ny)) return 1;
  if (xon) {
    HIPCHECK(hipMemcpyAsync((char *)c->sc + offsetof(QgScalars, xon), xon, sizeof(double) * (g.nl - 1), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
  }
  return 0;
}

// The fork's sponge layer (-Dsponge_layer_k247): r_spl(nxpo,nypo) of MODULE occonst (set by the main program,
// src/q-gcm.F:1154-1168) and the compile-time constant c1_spl (src/parameters_data.F:144).  NULL switches the term off.
extern "C" int qgcm_hip_set_sponge(qgcm_hip_handle c, const double *r_spl, double c1_spl) {
  if (!c) QG_FAIL("qgcm_hip_set_sponge: null handle");
  if (c->g.atm) QG_FAIL("qgcm_hip_set_sponge: the atmosphere has no sponge term (src/qgasubs.F)");
  drop_graphs(c); // captured steps hold the kernel parameters by value
  const QgGeom &g = c->g;
  if (!r_spl) {
    if (c->rspl) HIPCHECK(hipFree(c->rspl));
    c->rspl = nullptr;
    c->c1_spl = 0.0;
    return 0;
  }
  if (g.cyc) {
    // A periodic channel needs a ramp that is periodic in x (the fork's option nospl_in_ewbdy_k247: N / S boundaries
    // only).  The reference steps the duplicate column nxpo by itself (src/qgosubs.F:181), so with the full ramp of
    // src/q-gcm.F:1164-1166, whose values at i = 1 and i = nxpo differ, its qo(nxpo,j) drifts away from qo(1,j); on
    // the device column nxpo IS column 1 - refuse instead of differing silently.
    for (int j = 0; j < g.ny; ++j)
      if (r_spl[(size_t)j * g.nx] != r_spl[(size_t)j * g.nx + g.nx - 1])
        QG_FAIL("qgcm_hip_set_sponge: r_spl(1,%d) != r_spl(nxpo,%d) - a zonally cyclic ocean needs a ramp that is periodic in x (build with -Dnospl_in_ewbdy_k247)", j + 1, j + 1);
  }
  if (!c->rspl && dalloc(&c->rspl, (size_t)g.fstride)) return 1;
  c->c1_spl = c1_spl;
  return upload2d(c, c->rspl, g.ldx, r_spl, g.nx, g.ny);
}

extern "C" int qgcm_hip_set_cyc_forcing(qgcm_hip_handle c, double txisoc, double txinoc, const double *enisoc,
                                        const double *eninoc) {
  if (!c) QG_FAIL("qgcm_hip_set_cyc_forcing: null handle");
  if (!c->g.cyc) QG_FAIL("qgcm_hip_set_cyc_forcing: handle is a box ocean");
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  h.txisoc = txisoc;
  h.txinoc = txinoc;
  for (int k = 0; k < c->g.nl - 1; ++k) {
    if (enisoc) h.enisoc[k] = enisoc[k];
    if (eninoc) h.eninoc[k] = eninoc[k];
  }
  HIPCHECK(hipMemcpyAsync(c->sc, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_set_scalars(qgcm_hip_handle c, const double *s) {
  if (!c || !s) QG_FAIL("qgcm_hip_set_scalars: null argument");
  const int nl = c->g.nl, o = 2 * (nl - 1);
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < nl - 1; ++k) {
    h.dpioc[k] = s[k];
    h.dpiocp[k] = s[nl - 1 + k];
  }
  for (int k = 0; k < nl; ++k) {
    h.ocncs[k] = s[o + k];
    h.ocncn[k] = s[o + nl + k];
    h.ocncsp[k] = s[o + 2 * nl + k];
    h.ocncnp[k] = s[o + 3 * nl + k];
  }
  HIPCHECK(hipMemcpyAsync(c->sc, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_get_scalars(qgcm_hip_handle c, double *s) {
  if (!c || !s) QG_FAIL("qgcm_hip_get_scalars: null argument");
  const int nl = c->g.nl, o = 2 * (nl - 1);
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < nl - 1; ++k) {
    s[k] = h.dpioc[k];
    s[nl - 1 + k] = h.dpiocp[k];
  }
  for (int k = 0; k < nl; ++k) {
    s[o + k] = c->g.cyc ? h.ocncs[k] : 0.0;
    s[o + nl + k] = c->g.cyc ? h.ocncn[k] : 0.0;
    s[o + 2 * nl + k] = c->g.cyc ? h.ocncsp[k] : 0.0;
    s[o + 3 * nl + k] = c->g.cyc ? h.ocncnp[k] : 0.0;
  }
  return 0;
}

extern "C" int qgcm_hip_get_inv_diag(qgcm_hip_handle c, double *xinhom, double *coef) {
  if (!c) QG_FAIL("qgcm_hip_get_inv_diag: null handle");
  const int nl = c->g.nl;
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  if (xinhom)
    for (int m = 0; m < nl; ++m) xinhom[m] = h.xinhom[m];
  if (coef) {
    if (c->g.cyc) {
      for (int m = 0; m < nl - 1; ++m) {
        coef[m] = h.c1[m];
        coef[nl - 1 + m] = h.c2[m];
      }
      coef[2 * (nl - 1)] = h.c3;
    } else {
      for (int m = 0; m < nl - 1; ++m) coef[m] = h.hclco[m];
    }
  }
  return 0;
}

extern "C" int qgcm_hip_get_monitors(qgcm_hip_handle c, double *ermas, double *emfr) {
  if (!c) QG_FAIL("qgcm_hip_get_monitors: null handle");
  if (!c->g.cyc) QG_FAIL("qgcm_hip_get_monitors: the box ocean has no continuity monitors (src/ocisubs.F:268-283 is cyclic_ocean code)");
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->g.nl - 1; ++k) {
    if (ermas) ermas[k] = h.ermas[k];
    if (emfr) emfr[k] = h.emfr[k];
  }
  return 0;
}

// ---------------------------------------------------------------------------
// kernel launches
// ---------------------------------------------------------------------------
// Profiling mode: a HIP event pair brackets every kernel launch on the stream; the
// pairs are drained once per step so the queue stays busy between kernels.
struct KTimer {
  qgcm_hip_ctx *c;
  size_t slot;
  hipStream_t st;
  KTimer(qgcm_hip_ctx *c_, int id, hipStream_t st_ = nullptr) : c(c_), slot(0), st(st_ ? st_ : c_->stream) {
    if (c->profiling) {
      slot = c->evkid.size();
      if (c->evpool.size() < 2 * (slot + 1)) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        c->evpool.push_back(a);
        c->evpool.push_back(b);
      }
      c->evkid.push_back(id);
      note(hipEventRecord(c->evpool[2 * slot], st), 0);
    }
  }
  ~KTimer() {
    if (c->profiling)
      note(hipEventRecord(c->evpool[2 * slot + 1], st), 1);
  }
  void note(hipError_t e, int which) {
    if (e != hipSuccess && c->timer_err == hipSuccess) {
      c->timer_err = e;
      c->timer_err_at = 2 * (int)slot + which;
    }
  }
};

static void drain_timers(qgcm_hip_ctx *c) {
  if (c->evkid.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (size_t i = 0; i < c->evkid.size(); ++i) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->evpool[2 * i], c->evpool[2 * i + 1]);
    c->kms[c->evkid[i]] += ms;
    c->klaunch[c->evkid[i]] += 1;
  }
  c->evkid.clear();
}

static void fill_oml_final(qgcm_hip_ctx *c, QgOmlFinal &F, bool on);

// part: TEND_ALL = the whole launch; TEND_INNER = the tile rows whose stencils stay inside the owned rows (they need no
// halo row: y-slabs run them while the halo exchange of the previous step is still under way) plus the one-thread /
// one-workgroup riders of workgroup 0; TEND_OUTER = the first and the last tile row plus the edge / line-sum workgroups.
// INNER followed by OUTER writes what ALL writes, bit for bit (tiles are independent).
enum { TEND_ALL = 0, TEND_INNER = 1, TEND_OUTER = 2 };
static bool tend_can_split(const qgcm_hip_ctx *c) {
  const TendTiling T = c->g.cyc ? tend_tiling<true>(c->g) : tend_tiling<false>(c->g);
  return T.gy >= 3;
}

// write-through pair stores of the new qo (and 16-wide tiles) while the fields fit the Infinity Cache; plain stores and
// 32-wide tiles beyond (k_tend.h)
static bool tend_wtq(const qgcm_hip_ctx *c) {
  return (double)c->g.fstride * 8.0 * (7 * c->g.nl - 1) < 200.0e6 && !c->tend_wide;
}

static int launch_tend(qgcm_hip_ctx *c, bool upd_dpi = false, bool oml_final = false, int part = TEND_ALL) {
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr = c->prm;
  QgTendParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.pom = c->p[c->ip ^ 1];
  P.po = c->p[c->ip];
  P.qo = c->q[c->iq];
  P.qnew = c->q[c->iq ^ 1];
  P.wekpo = c->wekpo; P.entoc = c->entoc; P.ddynoc = c->ddynoc; P.yporel = c->yporel;
  P.wrk = c->wrk; P.sc = c->sc; P.bsum = nullptr;
  // scalar prologue of qgostep, src/qgosubs.F:76-82, and ocadif, :276-277
  P.adfaco = 1.0 / (12.0 * pr.dxo * pr.dyo * pr.fnot);
  P.dxom2 = 1.0 / (pr.dxo * pr.dxo);
  P.bcfaco = pr.bccooc * P.dxom2 / (0.5 * pr.bccooc + 1.0);
  P.tdto = pr.tdto;
  P.bdrfac = 0.5 * (pr.fnot >= 0.0 ? 1.0 : -1.0) * pr.delek / pr.hoc[g.nl - 1];
  P.fnot = pr.fnot; P.beta = pr.beta;
  for (int k = 0; k < g.nl; ++k) {
    P.fohfac[k] = pr.fnot / pr.hoc[k];
    P.ah2fac[k] = pr.ah2oc[k] / pr.fnot;
    P.ah4fac[k] = pr.ah4oc[k] / pr.fnot;
  }
  for (int i = 0; i < g.nl * g.nl; ++i) P.ctl2m[i] = pr.ctl2moc[i];
  P.upd_dpi = upd_dpi ? 1 : 0;
  P.rspl = c->rspl;
  P.tdc1 = pr.tdto * c->c1_spl; // tdto*c1_spl, src/qgosubs.F:204
  for (int k = 0; k < g.nl; ++k) P.gpoc[k] = pr.gpoc[k];
  // write-through pair stores of the new qo and 16-wide tiles while the step's working set (~ 7 nl - 1 fields) stays in the
  // 256 MiB Infinity Cache (NAtl 5 km: 150 MB: -1 us per step; SOcn 5 km's 425 MB: +8 us); plain stores and 32-wide
  // tiles at the HBM-bound sizes (k_tend.h)
  const bool wtq = tend_wtq(c);
  if (c->avg_now && (part != TEND_ALL || (g.cyc ? g.nl != 3 : (!wtq || g.nl > 4)))) QG_FAIL("k_tend: the fused leapfrog averaging belongs to whole-domain steps of the fused inverse-row kernels");
  const TendTiling T = g.cyc ? (wtq ? tend_tiling<true, TEND_TX>(g) : tend_tiling<true, TEND_TX_WIDE>(g))
                             : (wtq ? tend_tiling<false, TEND_TX>(g) : tend_tiling<false, TEND_TX_WIDE>(g));
  if (part != TEND_ALL && T.gy < 3) QG_FAIL("k_tend: a slab of fewer than three tile rows cannot be split");
  if (part == TEND_INNER) { P.trow0 = 1; P.trows = T.gy - 2; P.tstride = 1; }
  if (part == TEND_OUTER) { P.trow0 = 0; P.trows = 2; P.tstride = T.gy - 1; P.upd_dpi = 0; }
  const int ntiles = T.gx * (part == TEND_ALL ? T.gy : P.trows);
  // cyclic: the boundary line sums for the momentum constraints (state before the step) ride as extra workgroups
  QgCycSumParams S;
  memset(&S, 0, sizeof(S));
  if (g.cyc) {
    S.g = g;
    S.pom = P.pom; S.po = P.po; S.qo = P.qo;
    S.part = c->bpart_out ? c->bpart_out : c->bpart;
    S.bcfaco = P.bcfaco; S.dxom2 = P.dxom2; S.adfaco = P.adfaco; S.fnot = pr.fnot;
    S.dxo = pr.dxo; S.dyo = pr.dyo;
  }
  QgOmlFinal F;
  fill_oml_final(c, F, oml_final && c->oml.on && part != TEND_OUTER);
  const int nextra = part == TEND_INNER ? 0 : (g.cyc ? g.nl * 2 * BSUM_NB : T.nedge);
  dim3 grid(8 * ((ntiles + 7) / 8) + nextra); // 1-D: the kernel maps blockIdx -> tile per XCD band, then edge / line-sum work
  KTimer t(c, KN_TEND);
#define QG_TEND(NLV)                                                                                       \
  if (g.cyc && wtq) hipLaunchKernelGGL((k_tend<NLV, true, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F);    \
  else if (g.cyc) hipLaunchKernelGGL((k_tend<NLV, true, false>), grid, dim3(TEND_NT), 0, c->stream, P, S, F);     \
  else if (wtq) hipLaunchKernelGGL((k_tend<NLV, false, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F);       \
  else hipLaunchKernelGGL((k_tend<NLV, false, false>), grid, dim3(TEND_NT), 0, c->stream, P, S, F)
  if (c->avg_now && g.cyc) { // (long-row cyclic oceans: k_rfft3_unpack<.., AVG> does the rest)
    if (wtq) hipLaunchKernelGGL((k_tend<3, true, true, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F);
    else hipLaunchKernelGGL((k_tend<3, true, false, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F);
  } else if (c->avg_now) { // the step before a leapfrog averaging stores the averaged qo itself (one_step)
    switch (g.nl) {
      case 2: hipLaunchKernelGGL((k_tend<2, false, true, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F); break;
      case 3: hipLaunchKernelGGL((k_tend<3, false, true, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F); break;
      default: hipLaunchKernelGGL((k_tend<4, false, true, true>), grid, dim3(TEND_NT), 0, c->stream, P, S, F); break;
    }
  } else
  switch (g.nl) {
    case 2: QG_TEND(2); break;
    case 3: QG_TEND(3); break;
    case 4: QG_TEND(4); break;
    case 5: QG_TEND(5); break;
    case 6: QG_TEND(6); break;
    case 7: QG_TEND(7); break;
    case 8: QG_TEND(8); break;
    default: QG_FAIL("k_tend: unsupported nlo");
  }
#undef QG_TEND
  HIPCHECK(hipGetLastError());
  return 0;
}

// box rows that run k_dst_box (not the wave-per-row-pair kernels of k_dst64.h)
static bool dst_box_generic(const qgcm_hip_ctx *c) {
  return !c->g.cyc && (c->force_generic_dst || !(c->fftN == 64 * 15 || c->fftN == 64 * 3));
}
// ... whose inverse launch can carry the box constraint solve as an extra workgroup (2 or 3 layers, see k_dst_box)
static bool dst_box_rides_constr(const qgcm_hip_ctx *c) { return dst_box_generic(c) && c->g.nl <= 3 && !c->no_fused_constr; }

static int launch_dst(qgcm_hip_ctx *c, double *wrk, int nlayers, bool inverse, int layer0 = 0, hipStream_t st = nullptr,
                      bool cyc_part_b = false, bool box_constr = false) {
  if (!st) st = c->stream;
  const QgGeom &g = c->g;
  QgDstParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.wrk = wrk;
  P.twid = c->twid;
  P.sintab = c->sintab;
  P.rowsum = nullptr; // area / line integrals come from k_thomas (ksum and the k = 0 column)
  P.N = c->fftN;
  P.nfac = c->nfac;
  for (int f = 0; f < c->nfac; ++f) P.fac[f] = c->fac[f];
  P.nlayers = nlayers;
  P.layer0 = layer0;
  P.single = c->dst_single ? 1 : 0;
  const int nrows = g.jr1 - g.jr0 + 1;
  const int npairs = (nrows + 1) / 2;
  dim3 grid(npairs, nlayers);
  dim3 grid64((npairs + D64_WAVES - 1) / D64_WAVES, nlayers);
  if (cyc_part_b) { // generic cyclic inverse rows: one extra workgroup runs part B of the constraint algebra
    if (!inverse || !g.cyc || !c->d_cycq || g.nl > 4) QG_FAIL("launch_dst: part B rides in the inverse rows of a cyclic ocean with homogeneous solutions, nlo <= 4");
    P.cycq = c->d_cycq;
    grid.x += 1;
  }
  if (box_constr) { // generic box inverse rows: one extra workgroup runs the constraint solve (k_constr_box's body)
    if (!inverse || g.cyc || !c->d_boxq || !dst_box_generic(c) || g.nl > 3) QG_FAIL("launch_dst: the box constraint solve rides in the generic inverse rows of a box ocean with homogeneous solutions");
    P.boxq = c->d_boxq;
    grid.x += 1;
  }
  KTimer t(c, inverse ? KN_DSTI : KN_DSTF, st);
  if (c->fft3 && !c->force_generic_dst) {
    // long rows: three in-place register-radix stages (k_fft3.h)
#define QG_FFT3_LAUNCH(ID, R1, R2, R3)                                                                                     \
    if (c->fft3 == ID) {                                                                                                   \
      typedef Fft3Plan<R1, R2, R3> PL;                                                                                     \
      if (!g.cyc) hipLaunchKernelGGL((k_dst_box<false, FFT3_NT, PL>), grid, dim3(FFT3_NT), c->fft3_lds, st, P);            \
      else if (inverse) hipLaunchKernelGGL((k_rfft_cyc<true, PL, FFT3_NT>), grid, dim3(FFT3_NT), c->fft3_lds, st, P);      \
      else hipLaunchKernelGGL((k_rfft_cyc<false, PL, FFT3_NT>), grid, dim3(FFT3_NT), c->fft3_lds, st, P);                  \
    }
    QG_FFT3_PLANS(QG_FFT3_LAUNCH)
#undef QG_FFT3_LAUNCH
    HIPCHECK(hipGetLastError());
    return 0;
  }
  if (g.cyc) {
    // wave-per-row-pair fast path when nxto = 64*M (k_rfft64.h); the generic Stockham kernel otherwise
    const int M64 = (!c->force_generic_dst && c->fftN % 64 == 0) ? c->fftN / 64 : 0;
#define QG_RF(MV)                                                                           \
  if (inverse) hipLaunchKernelGGL((k_rfft64<MV, true>), grid64, dim3(D64_NT), 0, st, P);    \
  else hipLaunchKernelGGL((k_rfft64<MV, false>), grid64, dim3(D64_NT), 0, st, P)
    if (M64 == 3) { QG_RF(3); }
    else if (M64 == 6) { QG_RF(6); }
    else if (M64 == 15) { QG_RF(15); }
    else if (inverse) hipLaunchKernelGGL((k_rfft_cyc<true>), grid, dim3(RFFT_NT), c->dst_lds, st, P);
    else hipLaunchKernelGGL((k_rfft_cyc<false>), grid, dim3(RFFT_NT), c->dst_lds, st, P);
#undef QG_RF
    HIPCHECK(hipGetLastError());
    return 0;
  }
  // wave-per-row-pair fast path when nxto = 64*M with an in-register M-point DFT available
  if (c->fftN == 64 * 15 && !c->force_generic_dst) {
    hipLaunchKernelGGL((k_dst64<15, false>), grid64, dim3(D64_NT), 0, st, P);
  } else if (c->fftN == 64 * 3 && !c->force_generic_dst) {
    hipLaunchKernelGGL((k_dst64<3, false>), grid64, dim3(D64_NT), 0, st, P);
  } else if (c->fftN >= DST_BIG_N) hipLaunchKernelGGL((k_dst_box<false, DST_NT_BIG>), grid, dim3(DST_NT_BIG), c->dst_lds, st, P);
  else hipLaunchKernelGGL((k_dst_box<false>), grid, dim3(DST_NT), c->dst_lds, st, P);
  HIPCHECK(hipGetLastError());
  return 0;
}

static void fill_thomas_params(qgcm_hip_ctx *c, QgThomasParams &P, double *wrk, const QgThomasTab &tab, int nlayers, int layer0) {
  const QgGeom &g = c->g;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.gath_stride = (long)slab_msg_len_oml(c);
  P.slabDE = c->slabDE;
  P.ksum = c->ksum;
  P.wrk = wrk;
  P.binf = tab.binf; P.ptab = tab.ptab; P.rcb = tab.rcb; P.poff = tab.poff;
  P.nblk = (g.nk + TH_KW - 1) / TH_KW;
  P.aoc = c->prm.aoc;
  P.ftnorm = g.cyc ? 1.0 / g.nxt : 0.5 / g.nxt; // src/ocisubs.F:440, 547
  P.nlayers = nlayers;
  P.layer0 = layer0;
  P.nranks = 1;
}

// phase 0: whole column; 1: the slab's zero-inflow solution + its summary (send); 2: after the exchange - the inflows
// composed from all ranks' summaries (gath) and their responses added (k_thomas_corr); 4 / 5: set-up (k_thomas.h)
static int launch_thomas(qgcm_hip_ctx *c, double *wrk, const QgThomasTab &tab, int nlayers, int phase,
                         const double *gath, double *send, int rank, int nranks, int layer0, hipStream_t st,
                         bool cyc_part_a) {
  if (!st) st = c->stream;
  const QgGeom &g = c->g;
  if (!tab.binf) QG_FAIL("k_thomas: the pivot tables have not been built (qgcm_hip_set_grid)");
  QgThomasParams P;
  fill_thomas_params(c, P, wrk, tab, nlayers, layer0);
  P.gath = gath; P.send = send; P.rank = rank; P.nranks = nranks;
  if (phase == 2) {
    if (g.cyc) P.ybnd = c->ybnd;
    c->slab_gath = gath; // (the cyclic constraint algebra reads the boundary line sums at the end of the step messages)
    c->slab_nranks = nranks;
    P.cgath = (nranks == 1) ? c->slabDE : c->th_cgath; // a lone slab is its own rank 0
    if (nranks > 1 && (!c->th_cgath || c->th_cgath_ranks != nranks))
      QG_FAIL("qgcm_hip_thomas_phase: the set-up constants of the %d slabs have not been exchanged (qgcm_hip_set_thomas_consts)", nranks);
    if (&tab != &c->tt || !c->corr.meta) QG_FAIL("k_thomas_corr: the response tables belong to the modal solves of qgcm_hip_set_grid");
    const int nm = P.nblk * g.nl;
    QgThomasCorr T;
    T.ptab = c->corr.ptab; T.qtab = c->corr.qtab;
    T.np = c->corr.meta; T.nq = c->corr.meta + nm; T.poff = c->corr.meta + 2 * nm; T.qoff = c->corr.meta + 3 * nm;
    KTimer t(c, KN_THOMAS, st);
    const size_t lds = sizeof(double) * THC_LDS * nranks;
    if (lds > 32 * 1024) // (more than 31 slabs: beyond the default dynamic-LDS limit of a launch)
      HIPCHECK(hipFuncSetAttribute((const void *)k_thomas_corr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int nrows = g.jr1 - g.jr0 + 1, per = THC_UNR * (THC_NT / 8); // rows per slice (blockIdx.z)
    hipLaunchKernelGGL(k_thomas_corr, dim3(P.nblk, nlayers, (nrows + per - 1) / per), dim3(THC_NT), lds, st, P, T);
    HIPCHECK(hipGetLastError());
    return 0;
  }
  if (phase == 0) {
    c->slab_gath = nullptr; // whole-column solve: the constraint algebra reads bpart
    if (g.cyc && nlayers == g.nl) P.ybnd = c->ybnd; // ... and the zonal-mean rows next to the boundaries from here
  }
  // zonally cyclic geometries use the CYCA instantiation for the whole-column solve: it fills ybnd, and its extra
  // workgroup runs part A of the constraint algebra when asked to (inside qgcm_hip_steps)
  const bool cyca = (phase == 0 && g.cyc);
  if (cyc_part_a && (phase != 0 || !c->d_cycq || g.nl > 4)) QG_FAIL("k_thomas: part A of the constraint algebra needs the homogeneous solutions and nlo <= 4");
  if (cyca) P.cycq = cyc_part_a ? c->d_cycq : nullptr; // (its extra workgroup is added to the grid below)
  KTimer t(c, KN_THOMAS, st);
#define QG_TH(RV, KWV)                                                                                                  \
  {                                                                                                                      \
    dim3 gridk((g.nk + KWV - 1) / KWV + (cyca ? 1 : 0), nlayers);                                                        \
    switch (phase) {                                                                                                     \
      case 0:                                                                                                            \
        if (cyca) hipLaunchKernelGGL((k_thomas<RV, 0, true, KWV>), gridk, dim3(KWV * TH_NC), 0, st, P);                  \
        else hipLaunchKernelGGL((k_thomas<RV, 0, false, KWV>), gridk, dim3(KWV * TH_NC), 0, st, P);                      \
        break;                                                                                                           \
      case 1: hipLaunchKernelGGL((k_thomas<RV, 1, false, KWV>), gridk, dim3(KWV * TH_NC), 0, st, P); break;              \
      case 4: hipLaunchKernelGGL((k_thomas<RV, 4, false, KWV>), gridk, dim3(KWV * TH_NC), 0, st, P); break;              \
      default: hipLaunchKernelGGL((k_thomas<RV, 5, false, KWV>), gridk, dim3(KWV * TH_NC), 0, st, P); break;             \
    }                                                                                                                    \
  }
  // 8 .. 16 rows per thread: 512-thread workgroups of 8 wavenumbers, two per CU (128 VGPRs), the pair that shares the
  // 128-byte lines of a 16-wavenumber block on ONE XCD (k_thomas.h; r3: NAtl 5 km 10.6 -> 9.9 us against 1024-thread
  // workgroups); long columns (>= 20 rows per thread): 512 threads, 256 VGPRs
  switch (c->thR) {
    // (short columns too: 1024-thread workgroups of 16 wavenumbers measured slower everywhere - atmosphere 22.1 -> 21.3 us
    //  per step, 27.9 -> 26.3 on the two XCDs of a coupled run: profiles/r4_coupled_cu_masks.log)
    case 1: QG_TH(1, 8); break;
    case 2: QG_TH(2, 8); break;
    case 4: QG_TH(4, 8); break;
    case 8: QG_TH(8, 8); break;
    case 10: QG_TH(10, 8); break;
    case 12: QG_TH(12, 8); break;
    case 16: QG_TH(16, 8); break;
    case 20: QG_TH(20, 8); break;
    case 24: QG_TH(24, 8); break;
    case 32: QG_TH(32, 8); break;
    default: QG_FAIL("k_thomas: too many rows for the single-segment kernel");
  }
#undef QG_TH
  HIPCHECK(hipGetLastError());
  return 0;
}

static void fill_constr_params(qgcm_hip_ctx *c, QgConstrParams &P);

static int launch_constr(qgcm_hip_ctx *c) {
  const QgGeom &g = c->g;
  QgConstrParams P;
  fill_constr_params(c, P);
  if (g.cyc) {
    QgCycConstrParams Q;
    fill_cyc_constr_params(c, Q);
    KTimer t(c, KN_CONSTR);
    switch (g.nl) {
      case 2: hipLaunchKernelGGL((k_constr_cyc<2>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 3: hipLaunchKernelGGL((k_constr_cyc<3>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 4: hipLaunchKernelGGL((k_constr_cyc<4>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 5: hipLaunchKernelGGL((k_constr_cyc<5>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 6: hipLaunchKernelGGL((k_constr_cyc<6>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 7: hipLaunchKernelGGL((k_constr_cyc<7>), dim3(1), dim3(64), 0, c->stream, Q); break;
      case 8: hipLaunchKernelGGL((k_constr_cyc<8>), dim3(1), dim3(64), 0, c->stream, Q); break;
      default: QG_FAIL("k_constr_cyc: unsupported nlo");
    }
    HIPCHECK(hipGetLastError());
    return 0;
  }
  KTimer t(c, KN_CONSTR);
  switch (g.nl) {
    case 2: hipLaunchKernelGGL((k_constr_box<2>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 3: hipLaunchKernelGGL((k_constr_box<3>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 4: hipLaunchKernelGGL((k_constr_box<4>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 5: hipLaunchKernelGGL((k_constr_box<5>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 6: hipLaunchKernelGGL((k_constr_box<6>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 7: hipLaunchKernelGGL((k_constr_box<7>), dim3(1), dim3(64), 0, c->stream, P); break;
    case 8: hipLaunchKernelGGL((k_constr_box<8>), dim3(1), dim3(64), 0, c->stream, P); break;
    default: QG_FAIL("k_constr: unsupported nlo");
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

static void fill_cyc_constr_params(qgcm_hip_ctx *c, QgCycConstrParams &Q) {
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr2 = c->prm;
  QgConstrParams P;
  fill_constr_params(c, P);
  memset(&Q, 0, sizeof(Q));
  Q.g = g; Q.ksum = c->ksum; Q.wrk = c->wrk; Q.sc = c->sc; Q.cs = c->cs;
  Q.bpart = Q.bpart_n = c->bpart;
  Q.ybnd = c->ybnd; // k_thomas (PHASE 0 and PHASE 2 alike) leaves the zonal-mean rows next to the boundaries there
  if (c->slab_gath) {
    // y-slab stages: the boundary line sums travel at the end of the step messages (rank 0 owns the southern boundary,
    // the last rank the northern one); the solution next to the boundaries comes from k_thomas PHASE 2
    const size_t off = (size_t)TH_MSG * g.nl * g.ldw;
    Q.bpart = c->slab_gath + off;
    Q.bpart_n = c->slab_gath + (size_t)(c->slab_nranks - 1) * slab_msg_len_oml(c) + off;
    Q.ybnd = c->ybnd;
  }
  Q.adfaco = 1.0 / (12.0 * pr2.dxo * pr2.dyo * pr2.fnot);
  Q.delek_sgn = 0.5 * (pr2.fnot >= 0.0 ? 1.0 : -1.0) * pr2.delek;
  for (int k = 0; k < g.nl; ++k) {
    Q.ah2oc[k] = pr2.ah2oc[k];
    Q.ah4oc[k] = pr2.ah4oc[k];
  }
  Q.dxo = P.dxo; Q.dyo = P.dyo; Q.tdto = P.tdto; Q.fnot = P.fnot;
  for (int k = 0; k < QG_MAXL; ++k) { Q.gpoc[k] = P.gpoc[k]; Q.hoc[k] = P.hoc[k]; }
  for (int i = 0; i < QG_MAXL * QG_MAXL; ++i) { Q.ctl2m[i] = P.ctl2m[i]; Q.ctm2l[i] = P.ctm2l[i]; }
}

static void fill_constr_params(qgcm_hip_ctx *c, QgConstrParams &P) {
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr = c->prm;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.rowsum = c->rowsum;
  P.ksum = c->ksum;
  P.wcot = c->wcot;
  P.wrk = c->wrk;
  P.sc = c->sc;
  P.cs = c->cs;
  P.dxo = pr.dxo; P.dyo = pr.dyo; P.tdto = pr.tdto; P.fnot = pr.fnot;
  for (int k = 0; k < g.nl; ++k) {
    P.gpoc[k] = pr.gpoc[k];
    P.hoc[k] = pr.hoc[k];
  }
  for (int i = 0; i < g.nl * g.nl; ++i) {
    P.ctl2m[i] = pr.ctl2moc[i];
    P.ctm2l[i] = pr.ctm2loc[i];
  }
}

static void fill_bdy_params(qgcm_hip_ctx *c, QgBdyParams &P);

static int launch_unpack(qgcm_hip_ctx *c, bool fuse_bdy, double *msg_lo = nullptr, double *msg_hi = nullptr) {
  const QgGeom &g = c->g;
  QgUnpackParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.wrk = c->wrk;
  P.ochom = c->ochom;
  P.pnew = c->p[c->ip ^ 1];
  P.sc = c->sc;
  P.pch1 = c->pch1; P.pch2 = c->pch2; P.pbh = c->pbh;
  for (int i = 0; i < g.nl * g.nl; ++i) P.ctm2l[i] = c->prm.ctm2loc[i];
  if ((msg_lo || msg_hi) && (!fuse_bdy || g.jhi - g.jlo + 1 < 3)) QG_FAIL("k_unpack: halo messages need the fused boundary PV and three owned rows");
  P.msg_lo = msg_lo; // y-slab halo messages written by the same threads (slab stage 2)
  P.msg_hi = msg_hi;
  QgBdyParams B;
  fill_bdy_params(c, B); // B.qo = current qo; B.po unused by the fused kernel
  dim3 grid((g.nx + 255) / 256, g.jhi - g.jlo + 1);
  KTimer t(c, KN_UNPACK);
#define QG_UNPACK(NLV)                                                                                   \
  if (g.cyc && fuse_bdy) hipLaunchKernelGGL((k_unpack_cyc<NLV, true>), grid, dim3(256), 0, c->stream, P, B);       \
  else if (g.cyc) hipLaunchKernelGGL((k_unpack_cyc<NLV, false>), grid, dim3(256), 0, c->stream, P, B);             \
  else if (fuse_bdy) hipLaunchKernelGGL((k_unpack_box<NLV, true>), grid, dim3(256), 0, c->stream, P, B); \
  else hipLaunchKernelGGL((k_unpack_box<NLV, false>), grid, dim3(256), 0, c->stream, P, B)
  switch (g.nl) {
    case 2: QG_UNPACK(2); break;
    case 3: QG_UNPACK(3); break;
    case 4: QG_UNPACK(4); break;
    case 5: QG_UNPACK(5); break;
    case 6: QG_UNPACK(6); break;
    case 7: QG_UNPACK(7); break;
    case 8: QG_UNPACK(8); break;
    default: QG_FAIL("k_unpack: unsupported nlo");
  }
#undef QG_UNPACK
  HIPCHECK(hipGetLastError());
  return 0;
}

// cyclic / atmosphere fast path: inverse rows + homogeneous corrections + modes -> layers (+ zonal-boundary PV) in
// one launch (k_rfft64_unpack), nxto = 64 * {3, 6, 15}
static bool can_fuse_rfft_unpack(const qgcm_hip_ctx *c) {
  return c->g.cyc && c->whole && !c->force_generic_dst && !c->no_fused_unpack &&
         (c->fftN == 64 * 3 || c->fftN == 64 * 6 || c->fftN == 64 * 15) && c->g.nl >= 2 && c->g.nl <= 4;
}

static int launch_rfft_unpack(qgcm_hip_ctx *c, bool fuse_bdy, bool constr) {
  const QgGeom &g = c->g;
  QgDstParams D;
  memset(&D, 0, sizeof(D));
  D.g = g;
  D.wrk = c->wrk;
  D.twid = c->twid;
  D.N = c->fftN;
  D.nlayers = g.nl;
  QgUnpackParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.wrk = c->wrk;
  P.pnew = c->p[c->ip ^ 1];
  P.sc = c->sc;
  P.pch1 = c->pch1; P.pch2 = c->pch2; P.pbh = c->pbh;
  for (int i = 0; i < g.nl * g.nl; ++i) P.ctm2l[i] = c->prm.ctm2loc[i];
  QgBdyParams B;
  fill_bdy_params(c, B);
  if (constr && !c->d_cycq) QG_FAIL("k_rfft64_unpack: homogeneous solutions not set");
  QgCycConstrParams Qv; // by value: kernel arguments (k_rfft64.h)
  fill_cyc_constr_params(c, Qv);
  const int nrows = g.jr1 - g.jr0 + 1;
  dim3 grid((nrows + 1) / 2);
  KTimer t(c, KN_DSTI);
#define QG_RU(MV, NLV)                                                                                                        \
  if (fuse_bdy && constr) hipLaunchKernelGGL((k_rfft64_unpack<MV, NLV, true, true>), grid, dim3(64 * (NLV + 1)), 0, c->stream, D, P, B, Qv); \
  else if (fuse_bdy) hipLaunchKernelGGL((k_rfft64_unpack<MV, NLV, true, false>), grid, dim3(64 * NLV), 0, c->stream, D, P, B, Qv); \
  else hipLaunchKernelGGL((k_rfft64_unpack<MV, NLV, false, false>), grid, dim3(64 * NLV), 0, c->stream, D, P, B, Qv)
#define QG_RU_NL(MV)                 \
  switch (g.nl) {                    \
    case 2: QG_RU(MV, 2); break;     \
    case 3: QG_RU(MV, 3); break;     \
    default: QG_RU(MV, 4); break;    \
  }
  if (c->fftN == 64 * 15) {
    QG_RU_NL(15)
  } else if (c->fftN == 64 * 6) {
    QG_RU_NL(6)
  } else {
    QG_RU_NL(3)
  }
#undef QG_RU_NL
#undef QG_RU
  HIPCHECK(hipGetLastError());
  return 0;
}

// long rows (three-stage plans): inverse rows + homogeneous corrections + modes -> layers + boundary PV in one launch
// that reads the solved modes once (k_fft3_unpack.h); nlo = 3, the boundary PV always fused
static bool can_fuse_fft3_unpack(const qgcm_hip_ctx *c) {
  return c->fft3 && !c->force_generic_dst && !c->no_fused_unpack && c->g.nl == 3 && c->g.ldx % 2 == 0 && c->g.cyc && !c->g.atm;
}

static int launch_fft3_unpack(qgcm_hip_ctx *c, bool own_constr, double *msg_lo = nullptr, double *msg_hi = nullptr) {
  const QgGeom &g = c->g;
  if (!can_fuse_fft3_unpack(c)) QG_FAIL("k_fft3_unpack: not a long-row configuration of three layers");
  if ((msg_lo || msg_hi) && g.jhi - g.jlo + 1 < 3) QG_FAIL("k_fft3_unpack: halo messages need three owned rows");
  QgDstParams D;
  memset(&D, 0, sizeof(D));
  D.g = g;
  D.wrk = c->wrk;
  D.twid = c->twid;
  D.sintab = c->sintab;
  D.N = c->fftN;
  D.nlayers = g.nl;
  QgUnpackParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.wrk = c->wrk;
  P.ochom = c->ochom;
  P.pnew = c->p[c->ip ^ 1];
  P.sc = c->sc;
  P.pch1 = c->pch1; P.pch2 = c->pch2; P.pbh = c->pbh;
  P.msg_lo = msg_lo;
  P.msg_hi = msg_hi;
  for (int i = 0; i < g.nl * g.nl; ++i) P.ctm2l[i] = c->prm.ctm2loc[i];
  if (c->avg_now) {
    if (msg_lo || msg_hi || !own_constr) QG_FAIL("k_rfft3_unpack: the fused leapfrog averaging belongs to whole-domain steps");
    P.pavg = c->p[c->ip];     // this step's po (the launch writes the old pom buffer)
    P.qavg = c->q[c->iq ^ 1]; // this step's qo (iq already points at the new qo)
  }
  QgBdyParams B;
  fill_bdy_params(c, B);
  const int npairs = (g.jr1 - g.jr0 + 2) / 2;
  dim3 grid(8 * ((npairs + 7) / 8) * g.nl);
  KTimer t(c, KN_DSTI);
  if (g.cyc) {
    QgCycConstrParams Q;
    fill_cyc_constr_params(c, Q);
#define QG_FFT3U_LAUNCH(ID, R1, R2, R3)                                                                                   \
    if (c->fft3 == ID) {                                                                                                  \
      typedef Fft3Plan<R1, R2, R3> PL;                                                                                    \
      if (c->avg_now) hipLaunchKernelGGL((k_rfft3_unpack<PL, 3, FFT3_NT, true>), grid, dim3(FFT3_NT), c->fft3_lds, c->stream, D, P, B, Q, 1); \
      else hipLaunchKernelGGL((k_rfft3_unpack<PL, 3, FFT3_NT>), grid, dim3(FFT3_NT), c->fft3_lds, c->stream, D, P, B, Q, own_constr ? 1 : 0); \
    }
    QG_FFT3_PLANS(QG_FFT3U_LAUNCH)
#undef QG_FFT3U_LAUNCH
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

// box fast path: inverse row transform + modes -> layers (+ boundary PV) in one launch (k_dst64_unpack)
static bool can_fuse_dst_unpack(const qgcm_hip_ctx *c) {
  return !c->g.cyc && !c->force_generic_dst && !c->no_fused_unpack && (c->fftN == 64 * 15 || c->fftN == 64 * 3) &&
         c->g.nl >= 2 && c->g.nl <= 4;
}

static int launch_dst_unpack(qgcm_hip_ctx *c, bool fuse_bdy, double *msg_lo = nullptr, double *msg_hi = nullptr,
                             bool constr = false) {
  const QgGeom &g = c->g;
  QgDstParams D;
  memset(&D, 0, sizeof(D));
  D.g = g;
  D.wrk = c->wrk;
  D.twid = c->twid;
  D.sintab = c->sintab;
  D.N = c->fftN;
  D.nlayers = g.nl;
  QgUnpackParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.wrk = c->wrk;
  P.ochom = c->ochom;
  P.pnew = c->p[c->ip ^ 1];
  P.sc = c->sc;
  P.msg_lo = msg_lo;
  P.msg_hi = msg_hi;
  if (c->avg_now) {
    if (msg_lo || msg_hi || !fuse_bdy || !constr) QG_FAIL("k_dst64_unpack: the fused leapfrog averaging belongs to whole-domain steps");
    P.pavg = c->p[c->ip];     // this step's po (the launch writes the old pom buffer)
    P.qavg = c->q[c->iq ^ 1]; // this step's qo (iq already points at the new qo)
  }
  for (int i = 0; i < g.nl * g.nl; ++i) P.ctm2l[i] = c->prm.ctm2loc[i];
  QgBdyParams B;
  fill_bdy_params(c, B);
  QgConstrLite C;
  {
    QgConstrParams F;
    fill_constr_params(c, F);
    memset(&C, 0, sizeof(C));
    C.g = F.g; C.ksum = F.ksum; C.wcot = F.wcot; C.sc = F.sc;
    C.dxo = F.dxo; C.dyo = F.dyo;
    for (int i = 0; i < 16; ++i) { C.cs.cdiffo[i] = F.cs.cdiffo[i]; C.cs.cdhoc[i] = F.cs.cdhoc[i]; C.cs.cdhlu[i] = F.cs.cdhlu[i]; }
    for (int i = 0; i < 4; ++i) C.cs.ipiv[i] = F.cs.ipiv[i];
  }
  const int nrows = g.jr1 - g.jr0 + 1;
  dim3 grid((nrows + 1) / 2);
  KTimer t(c, KN_DSTI);
#define QG_DU(MV, NLV)                                                                                                  \
  if (c->avg_now) hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, true, false, true, true>), grid, dim3(64 * (NLV + 1)), 0, c->stream, D, P, B, C); \
  else if ((msg_lo || msg_hi) && constr) hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, true, true, true>), grid, dim3(64 * (NLV + 1)), 0, c->stream, D, P, B, C); \
  else if (msg_lo || msg_hi) hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, true, true, false>), grid, dim3(64 * NLV), 0, c->stream, D, P, B, C); \
  else if (fuse_bdy && constr) hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, true, false, true>), grid, dim3(64 * (NLV + 1)), 0, c->stream, D, P, B, C); \
  else if (fuse_bdy) hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, true, false, false>), grid, dim3(64 * NLV), 0, c->stream, D, P, B, C);  \
  else hipLaunchKernelGGL((k_dst64_unpack<MV, NLV, false, false, false>), grid, dim3(64 * NLV), 0, c->stream, D, P, B, C)
#define QG_DU_NL(MV)                 \
  switch (g.nl) {                    \
    case 2: QG_DU(MV, 2); break;     \
    case 3: QG_DU(MV, 3); break;     \
    default: QG_DU(MV, 4); break;    \
  }
  if (c->fftN == 64 * 15) {
    QG_DU_NL(15)
  } else {
    QG_DU_NL(3)
  }
#undef QG_DU_NL
#undef QG_DU
  HIPCHECK(hipGetLastError());
  return 0;
}

static void fill_bdy_params(qgcm_hip_ctx *c, QgBdyParams &P) {
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr = c->prm;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.po = c->p[c->ip];
  P.qo = c->q[c->iq];
  P.ddynoc = c->ddynoc;
  P.yporel = c->yporel;
  const double dxom2 = 1.0 / (pr.dxo * pr.dxo);
  P.bcfaco_f0 = pr.bccooc * dxom2 / (0.5 * pr.bccooc + 1.0) / pr.fnot; // src/vorsubs.F:268
  P.beta = pr.beta;
  for (int k = 0; k < g.nl; ++k)
    for (int l = 0; l < g.nl; ++l) P.f0A[k + g.nl * l] = pr.fnot * pr.amatoc[k + g.nl * l];
}

static int launch_ocqbdy(qgcm_hip_ctx *c) {
  const QgGeom &g = c->g;
  QgBdyParams P;
  fill_bdy_params(c, P);
  const int nmax = g.nx > g.ny ? g.nx : g.ny;
  dim3 grid((nmax + 255) / 256, g.cyc ? 2 : 4, g.nl);
  KTimer t(c, KN_OCQBDY);
  hipLaunchKernelGGL(k_ocqbdy, grid, dim3(256), 0, c->stream, P);
  HIPCHECK(hipGetLastError());
  return 0;
}

static int launch_lfavg(qgcm_hip_ctx *c) {
  const QgGeom &g = c->g;
  const long n = g.fstride * g.nl;
  KTimer t(c, KN_LFAVG);
  hipLaunchKernelGGL(k_lf_average, dim3(2048), dim3(256), 0, c->stream, c->q[c->iq], c->q[c->iq ^ 1], c->p[c->ip],
                     c->p[c->ip ^ 1], n, c->sc, g.nl, g.cyc);
  HIPCHECK(hipGetLastError());
  return 0;
}

static int check_ready(qgcm_hip_ctx *c, const char *who) {
  if (!c) QG_FAIL("%s: null handle", who);
  if (!c->grid_set) QG_FAIL("%s: qgcm_hip_set_grid has not been called", who);
  return 0;
}

extern "C" int qgcm_hip_qgostep(qgcm_hip_handle c) {
  if (check_ready(c, "qgcm_hip_qgostep")) return 1;
  if (launch_tend(c)) return 1;
  c->iq ^= 1; // the old qom buffer now holds qo; the old qo is qom
  return 0;
}

// in_step: called from qgcm_hip_steps, where k_tend has already stepped dpioc / dpiocp (launch_tend(c, true)) and
// the box constraint solve can ride in the fused inverse-transform kernel
static int ocinvq_impl(qgcm_hip_ctx *c, bool fuse_bdy, bool in_step = false) {
  if (check_ready(c, "qgcm_hip_ocinvq")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_ocinvq: this handle is a y-slab; drive it with the slab building blocks");
  if (!c->homog_set) QG_FAIL("qgcm_hip_ocinvq: homogeneous solutions not set");
  // (A per-mode side-stream variant of this chain was measured slower - 140 vs 116 us/step at 5 km - and removed.)
  if (launch_dst(c, c->wrk, c->g.nl, false)) return 1;
  if (c->g.cyc && can_fuse_rfft_unpack(c)) {
    // cyclic / atmosphere, nxto = 64*M: inside qgcm_hip_steps part A of the constraint algebra rides in the Thomas
    // launch and part B in the fused inverse-transform kernel (3 launches after k_tend instead of 5)
    const bool fc = in_step && fuse_bdy && !c->no_fused_constr;
    if (launch_thomas(c, c->wrk, c->tt, c->g.nl, 0, nullptr, nullptr, 0, 1, 0, nullptr, fc)) return 1;
    if (!fc && launch_constr(c)) return 1;
    if (launch_rfft_unpack(c, fuse_bdy, fc)) return 1;
    c->ip ^= 1;
    return 0;
  }
  // (the wave-per-row-pair kernels of k_rfft64.h have no such extra workgroup: they are fused with the unpack step
  //  above unless that is switched off, and then keep the stand-alone constraint launch)
  const bool generic_rows = c->force_generic_dst || !(c->fftN == 64 * 3 || c->fftN == 64 * 6 || c->fftN == 64 * 15);
  if (c->g.cyc && in_step && !c->no_fused_constr && generic_rows && c->g.nl <= 4) { // (the riding parts A / B: nlo <= 4)
    // cyclic ocean with generic row sizes (SOcn 5 km), inside qgcm_hip_steps: part A of the constraint algebra rides in
    // the Thomas launch, part B in the inverse-row launch (it reads ksum and ybnd, not wrk): no launch of its own
    if (launch_thomas(c, c->wrk, c->tt, c->g.nl, 0, nullptr, nullptr, 0, 1, 0, nullptr, true)) return 1;
    if (fuse_bdy && can_fuse_fft3_unpack(c)) { // long rows: one launch, every workgroup evaluates part B for itself
      if (launch_fft3_unpack(c, true)) return 1;
      c->ip ^= 1;
      return 0;
    }
    if (launch_dst(c, c->wrk, c->g.nl, true, 0, nullptr, true)) return 1;
    if (launch_unpack(c, fuse_bdy)) return 1;
    c->ip ^= 1;
    return 0;
  }
  if (launch_thomas(c, c->wrk, c->tt, c->g.nl, 0, nullptr, nullptr, 0, 1, 0, nullptr)) return 1;
  const bool fused_constr = in_step && fuse_bdy && can_fuse_dst_unpack(c) && !c->no_fused_constr;
  // generic box rows: the constraint solve rides as an extra workgroup of the inverse-row launch
  const bool ride_constr = !fused_constr && dst_box_rides_constr(c);
  // area (and, cyclic, line) integrals are a by-product of the y sweeps: the constraints precede the inverse transform
  if (!fused_constr && !ride_constr && launch_constr(c)) return 1;
  if (can_fuse_dst_unpack(c)) {
    if (launch_dst_unpack(c, fuse_bdy, nullptr, nullptr, fused_constr)) return 1;
    c->ip ^= 1; // new po sits in the old pom buffer; the old po is pom
    return 0;
  }
  if (launch_dst(c, c->wrk, c->g.nl, true, 0, nullptr, false, ride_constr)) return 1;
  if (launch_unpack(c, fuse_bdy)) return 1;
  c->ip ^= 1; // new po sits in the old pom buffer; the old po is pom
  return 0;
}

extern "C" int qgcm_hip_ocinvq(qgcm_hip_handle c) { return ocinvq_impl(c, false); }

extern "C" int qgcm_hip_ocqbdy(qgcm_hip_handle c) {
  if (check_ready(c, "qgcm_hip_ocqbdy")) return 1;
  return launch_ocqbdy(c);
}

// ocqbdy / atqzbd on host arrays (start-up calls of the main program, src/q-gcm.F:724-725, 743-744)
extern "C" int qgcm_hip_ocqbdy_host(qgcm_hip_handle c, double *q, const double *p) {
  if (!c) QG_FAIL("qgcm_hip_ocqbdy_host: null handle");
  if (!c->geom_set) QG_FAIL("qgcm_hip_ocqbdy_host: neither qgcm_hip_set_geometry nor qgcm_hip_set_grid has been called");
  if (!q || !p) QG_FAIL("qgcm_hip_ocqbdy_host: null argument");
  if (!c->whole) QG_FAIL("qgcm_hip_ocqbdy_host: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  const size_t n = (size_t)g.fstride * g.nl;
  double *dp = nullptr, *dq = nullptr;
  if (dalloc(&dp, n) || dalloc(&dq, n)) return 1;
  const long rows = (long)g.ny * g.nl;
  int rc = upload2d(c, dp, g.ldx, p, g.nx, rows) || upload2d(c, dq, g.ldx, q, g.nx, rows);
  if (!rc) {
    QgBdyParams P;
    fill_bdy_params(c, P);
    P.po = dp;
    P.qo = dq;
    const int nmax = g.nx > g.ny ? g.nx : g.ny;
    hipLaunchKernelGGL(k_ocqbdy, dim3((nmax + 255) / 256, g.cyc ? 2 : 4, g.nl), dim3(256), 0, c->stream, P);
    rc = hipGetLastError() != hipSuccess;
    if (rc) snprintf(g_err, sizeof(g_err), "qgcm_hip_ocqbdy_host: launch failed");
  }
  if (!rc) rc = download2d(c, q, dq, g.ldx, g.nx, rows);
  hipFree(dp);
  hipFree(dq);
  return rc;
}

static int launch_oml_average(qgcm_hip_ctx *c);

// the whole ocean part of the averaging block src/q-gcm.F:1328-1366: po, qo, constraint scalars and - when the
// mixed layer lives on the device - sst
extern "C" int qgcm_hip_lf_average(qgcm_hip_handle c) {
  if (check_ready(c, "qgcm_hip_lf_average")) return 1;
  if (launch_lfavg(c)) return 1;
  if (c->oml.on && launch_oml_average(c)) return 1;
  return 0;
}

// ---------------------------------------------------------------------------
// ocean mixed layer
// ---------------------------------------------------------------------------
extern "C" int qgcm_hip_oml_init(qgcm_hip_handle c, const qgcm_hip_oml_params *p) {
  if (check_ready(c, "qgcm_hip_oml_init")) return 1;
  if (!p) QG_FAIL("qgcm_hip_oml_init: null parameters");
  if (c->sc_comm) QG_FAIL("qgcm_hip_oml_init: call before qgcm_hip_comm_init (the step and halo messages grow by the mixed layer's parts)");
  if (!(p->hmoc > 0.0) || p->toc1 == p->toc2) QG_FAIL("qgcm_hip_oml_init: need hmoc > 0 and toc(1) != toc(2)");
  const QgGeom &g = c->g;
  const int nxt = g.nxt, nyt = g.ny - 1;
  if (nxt < 3 || nyt < 3) QG_FAIL("qgcm_hip_oml_init: grid too small");
  auto &o = c->oml;
  drop_graphs(c);
  o.prm = *p;
  if (!o.sst[0]) {
    o.ldt = round_up(nxt, 16);
    const size_t nT = (size_t)o.ldt * nyt, nP = (size_t)g.ldx * g.ny;
    o.nblkA = ((nxt + OML_TX - 1) / OML_TX) * ((nyt + OML_SH - 1) / OML_SH);
    o.nblkB = ((g.nx + OML_TX - 1) / OML_TX) * ((g.ny + OML_TY * OML_RPT * OML_ERR - 1) / (OML_TY * OML_RPT * OML_ERR));
    struct { double **ptr; size_t n; } bufs[] = {{&o.sst[0], nT}, {&o.sst[1], nT}, {&o.sst[2], nT}, {&o.fnet, nT}, {&o.wekto, nT},
                                                 {&o.xfo, nT}, {&o.taux, nP}, {&o.tauy, nP}, {&o.partA, (size_t)3 * o.nblkA},
                                                 {&o.partB, (size_t)3 * o.nblkB}, {&o.diag, 2}};
    for (auto &b : bufs) {
      HIPCHECK(hipMalloc((void **)b.ptr, b.n * sizeof(double)));
      HIPCHECK(hipMemsetAsync(*b.ptr, 0, b.n * sizeof(double), c->stream));
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    o.is = 0;
    o.ism = 1;
  }
  o.on = true;
  return 0;
}

static int oml_ready(qgcm_hip_ctx *c, const char *who) {
  if (check_ready(c, who)) return 1;
  if (!c->oml.on) QG_FAIL("%s: qgcm_hip_oml_init has not been called", who);
  return 0;
}

extern "C" int qgcm_hip_oml_set_state(qgcm_hip_handle c, const double *sst, const double *sstm) {
  if (oml_ready(c, "qgcm_hip_oml_set_state")) return 1;
  const int nxt = c->g.nxt, nyt = c->g.ny - 1;
  if (sst && upload2d(c, c->oml.sst[c->oml.is], c->oml.ldt, sst, nxt, nyt)) return 1;
  if (sstm && upload2d(c, c->oml.sst[c->oml.ism], c->oml.ldt, sstm, nxt, nyt)) return 1;
  return 0;
}

extern "C" int qgcm_hip_oml_get_state(qgcm_hip_handle c, double *sst, double *sstm) {
  if (oml_ready(c, "qgcm_hip_oml_get_state")) return 1;
  const int nxt = c->g.nxt, nyt = c->g.ny - 1;
  if (sst && download2d(c, sst, c->oml.sst[c->oml.is], c->oml.ldt, nxt, nyt)) return 1;
  if (sstm && download2d(c, sstm, c->oml.sst[c->oml.ism], c->oml.ldt, nxt, nyt)) return 1;
  return 0;
}

extern "C" int qgcm_hip_oml_set_forcing(qgcm_hip_handle c, const double *fnetoc, const double *wekto, const double *tauxo,
                                        const double *tauyo) {
  if (oml_ready(c, "qgcm_hip_oml_set_forcing")) return 1;
  const QgGeom &g = c->g;
  const int nxt = g.nxt, nyt = g.ny - 1;
  if (fnetoc && upload2d(c, c->oml.fnet, c->oml.ldt, fnetoc, nxt, nyt)) return 1;
  if (wekto && upload2d(c, c->oml.wekto, c->oml.ldt, wekto, nxt, nyt)) return 1;
  if (tauxo && upload2d(c, c->oml.taux, g.ldx, tauxo, g.nx, g.ny)) return 1;
  if (tauyo && upload2d(c, c->oml.tauy, g.ldx, tauyo, g.nx, g.ny)) return 1;
  return 0;
}

static void fill_oml_final(qgcm_hip_ctx *c, QgOmlFinal &F, bool on) {
  const auto &o = c->oml;
  memset(&F, 0, sizeof(F));
  F.partA = o.partA; F.partB = o.partB; F.nblkA = o.nblkA; F.nblkB = o.nblkB; F.cyc = c->g.cyc; F.on = on ? 1 : 0;
  F.sc = c->sc; F.diag = o.diag;
  F.ocnorm = 1.0 / ((double)c->g.nxt * (double)(c->g.nyg - 1)); // src/parameters_data.F:88
  F.dxo = c->prm.dxo; F.dyo = c->prm.dyo;
}

static void fill_oml_params(qgcm_hip_ctx *c, QgOmlParams &P) {
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr = c->prm;
  auto &o = c->oml;
  const qgcm_hip_oml_params &q = o.prm;
  memset(&P, 0, sizeof(P));
  P.nxt = g.nxt; P.nyt = g.ny - 1; P.nx = g.nx; P.ny = g.ny; P.cyc = g.cyc; P.sb = q.sb_hflux; P.nb = q.nb_hflux;
  P.ldt = o.ldt; P.ldx = g.ldx;
  // y-slab view (k_oml.h): T row j lies between p rows j and j+1
  P.joff = g.joff; P.nyg = g.nyg; P.nytg = g.nyg - 1;
  P.jP0 = g.jlo; P.jP1 = g.jhi;
  P.jT0 = g.jlo; P.jT1 = (g.jhi + g.joff == g.nyg) ? g.jhi - 1 : g.jhi;
  P.jX0 = (g.jlo + g.joff > 1) ? P.jT0 - 1 : P.jT0;
  const int spare = 3 - o.is - o.ism;
  P.sst = o.sst[o.is]; P.sstm = o.sst[o.ism]; P.sstn = o.sst[spare];
  P.fnet = o.fnet; P.wekto = o.wekto; P.xfo = o.xfo;
  P.po1 = c->p[c->ip]; P.taux = o.taux; P.tauy = o.tauy;
  P.entoc = c->entoc;
  P.partA = o.partA; P.partB = o.partB;
  P.nblkA = ((P.nxt + OML_TX - 1) / OML_TX) * ((P.jT1 - P.jX0 + 1 + OML_SH - 1) / OML_SH);
  P.nblkB = ((P.nx + OML_TX - 1) / OML_TX) * ((P.jP1 - P.jP0 + 1 + OML_TY * OML_RPT * OML_ERR - 1) / (OML_TY * OML_RPT * OML_ERR));
  P.sc = c->sc; P.diag = o.diag;
  const double dxom2 = 1.0 / (pr.dxo * pr.dxo), rdxof0 = 1.0 / (pr.dxo * pr.fnot); // src/q-gcm.F:435
  P.uvgfac = q.ycexp * rdxof0;           // src/omlsubs.F:274-277
  P.rhf0hm = 0.5 / (pr.fnot * q.hmoc);
  P.d2tfac = q.st2d * dxom2;
  P.d4tfac = q.st4d * (dxom2 * dxom2);
  P.hdxom1 = 0.5 / pr.dxo;
  P.hmoinv = 1.0 / q.hmoc;               // src/omlsubs.F:78-80
  P.dtoinv = 1.0 / (q.toc1 - q.toc2);
  P.entfac = q.hmoc * P.dtoinv / pr.tdto;
  P.tdto = pr.tdto; P.rrcpoc = q.rrcpoc; P.toc1 = q.toc1; P.tsbdy = q.tsbdy; P.tnbdy = q.tnbdy;
  P.ocnorm = 1.0 / ((double)P.nxt * (double)P.nytg); // src/parameters_data.F:88
  P.dxo = pr.dxo; P.dyo = pr.dyo;
}

// first half of `oml`: new sst, raw entrainment, per-workgroup partial sums.  send3 (y-slabs): this slab's three sums
static int launch_oml_a(qgcm_hip_ctx *c, double *send3) {
  QgOmlParams P;
  fill_oml_params(c, P);
  dim3 gA((P.nxt + OML_TX - 1) / OML_TX, (P.jT1 - P.jX0 + 1 + OML_SH - 1) / OML_SH);
  hipLaunchKernelGGL(k_oml_step, gA, dim3(OML_NT), 0, c->stream, P);
  if (send3) hipLaunchKernelGGL(k_oml_sum3, dim3(1), dim3(OML_NT), 0, c->stream, (const double *)P.partA, P.nblkA, send3);
  HIPCHECK(hipGetLastError());
  return 0;
}

// second half: entoc from the entrainment minus its basin-wide mean (gath3 (3, nranks): every slab's sums, or nullptr
// for a handle that owns the whole domain), then the rotation of the sst buffers
static int launch_oml_b(qgcm_hip_ctx *c, const double *gath3, int nranks) {
  auto &o = c->oml;
  QgOmlParams P;
  fill_oml_params(c, P);
  P.mean_gath = gath3;
  P.nranks = nranks;
  dim3 gB((P.nx + OML_TX - 1) / OML_TX, (P.jP1 - P.jP0 + 1 + OML_TY * OML_RPT * OML_ERR - 1) / (OML_TY * OML_RPT * OML_ERR));
  hipLaunchKernelGGL(k_oml_entoc, gB, dim3(OML_NT), 0, c->stream, P);
  HIPCHECK(hipGetLastError());
  // rotation: sstm <- sst, sst <- new (src/omlsubs.F:125-126)
  const int spare = 3 - o.is - o.ism;
  o.ism = o.is;
  o.is = spare;
  return 0;
}

// with_final = false: the final reduction rides in workgroup 0 of the tendency launch that follows (one_step)
static int launch_oml(qgcm_hip_ctx *c, bool with_final = true) {
  {
    KTimer t(c, KN_OML);
    if (launch_oml_a(c, nullptr)) return 1;
  }
  {
    KTimer t(c, KN_OML_ENTOC);
    if (launch_oml_b(c, nullptr, 1)) return 1;
  }
  if (with_final) {
    QgOmlFinal F;
    fill_oml_final(c, F, true);
    hipLaunchKernelGGL(k_oml_final, dim3(1), dim3(OML_NT), 0, c->stream, F);
  }
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int qgcm_hip_oml(qgcm_hip_handle c) {
  if (oml_ready(c, "qgcm_hip_oml")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_oml: this handle is a y-slab; the mixed layer steps with the slab stages 10 / 11");
  return launch_oml(c);
}

static int launch_oml_average(qgcm_hip_ctx *c) {
  const long n = (long)c->oml.ldt * (c->g.ny - 1);
  hipLaunchKernelGGL(k_oml_average, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->oml.sst[c->oml.is],
                     (const double *)c->oml.sst[c->oml.ism], n);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int qgcm_hip_oml_get_diag(qgcm_hip_handle c, double *entoc, double *diag) {
  if (oml_ready(c, "qgcm_hip_oml_get_diag")) return 1;
  const QgGeom &g = c->g;
  if (entoc && download2d(c, entoc, c->entoc, g.ldx, g.nx, g.ny)) return 1;
  if (diag) {
    QgScalars h;
    double d[2];
    HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(d, c->oml.diag, sizeof(d), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    diag[0] = h.xon[0]; diag[1] = d[0]; diag[2] = d[1];
    diag[3] = g.cyc ? h.enisoc[0] : 0.0; diag[4] = g.cyc ? h.eninoc[0] : 0.0;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// validity scan
// ---------------------------------------------------------------------------
extern "C" int qgcm_hip_set_dtopoc(qgcm_hip_handle c, const double *dtopoc) {
  if (check_ready(c, "qgcm_hip_set_dtopoc")) return 1;
  const QgGeom &g = c->g;
  if (!dtopoc) {
    if (c->dtopoc) hipFree(c->dtopoc);
    c->dtopoc = nullptr;
    return 0;
  }
  if (!c->dtopoc && dalloc(&c->dtopoc, (size_t)g.ldx * g.ny)) return 1;
  return upload2d(c, c->dtopoc, g.ldx, dtopoc, g.nx, g.ny);
}

extern "C" int qgcm_hip_valids(qgcm_hip_handle c, double *out, int *solnok) {
  if (check_ready(c, "qgcm_hip_valids")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_valids: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  if (g.nl < 2 || g.nl > QG_MAXL) QG_FAIL("qgcm_hip_valids: unsupported nlo");
  const int nres = 2 * VAL_NMM + g.nl + 1;
  if (!c->val_part) {
    if (dalloc(&c->val_part, (size_t)(2 * VAL_NMM + QG_MAXL) * VAL_NB)) return 1;
    if (dalloc(&c->val_out, (size_t)2 * VAL_NMM + QG_MAXL + 1)) return 1;
  }
  QgValidsParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.po = c->p[c->ip];
  P.qo = c->q[c->iq];
  if (c->oml.on) {
    P.sst = c->oml.sst[c->oml.is];
    P.wekto = c->oml.wekto;
    P.ldt = c->oml.ldt;
  }
  P.dtopoc = c->dtopoc;
  for (int k = 0; k < g.nl - 1; ++k) P.rgpoc[k] = 1.0 / c->prm.gpoc[k]; // src/valsubs.F:390-392
  for (int k = 0; k < g.nl; ++k) P.hoc[k] = c->prm.hoc[k];
  P.part = c->val_part;
  P.out = c->val_out;
  P.ocnorm = 1.0 / ((double)g.nxt * (double)(g.ny - 1));
#define QG_VALIDS(NLV)                                                                       \
  hipLaunchKernelGGL((k_valids_scan<NLV>), dim3(VAL_NB), dim3(VAL_NT), 0, c->stream, P);      \
  hipLaunchKernelGGL((k_valids_final<NLV>), dim3(1), dim3(VAL_NT), 0, c->stream, P)
  QG_SWITCH_NL(g.nl, QG_VALIDS, "k_valids");
#undef QG_VALIDS
  HIPCHECK(hipGetLastError());
  double h[2 * VAL_NMM + QG_MAXL + 1];
  HIPCHECK(hipMemcpyAsync(h, c->val_out, sizeof(double) * nres, hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  if (out)
    for (int q = 0; q < nres - 1; ++q) out[q] = h[q];
  if (solnok) *solnok = h[nres - 1] > 0.5 ? 1 : 0;
  return 0;
}

// ---------------------------------------------------------------------------
// start-up / restart arithmetic and the progress sample on the device
// ---------------------------------------------------------------------------
static int launch_area_sums(qgcm_hip_ctx *c) {
  const QgGeom &g = c->g;
  if (!c->area_part) {
    if (dalloc(&c->area_part, (size_t)AREA_NB * 3 * QG_MAXL) || dalloc(&c->area_out, (size_t)3 * QG_MAXL)) return 1;
  }
  QgAreaParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  P.f[0] = c->p[c->ip]; P.f[1] = c->p[c->ip ^ 1]; P.f[2] = c->q[c->iq];
  P.part = c->area_part; P.out = c->area_out;
#define QG_AREA(NLV)                                                                          \
  hipLaunchKernelGGL((k_area_partial<NLV>), dim3(AREA_NB), dim3(AREA_NT), 0, c->stream, P);     \
  hipLaunchKernelGGL((k_area_final<NLV>), dim3(1), dim3(64), 0, c->stream, P)
  QG_SWITCH_NL(g.nl, QG_AREA, "k_area");
#undef QG_AREA
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int qgcm_hip_init_from_p(qgcm_hip_handle c) {
  if (check_ready(c, "qgcm_hip_init_from_p")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_init_from_p: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  const qgcm_hip_params &pr = c->prm;
  // constr: src/q-gcm.F:711
  if (launch_area_sums(c)) return 1;
  {
    QgConstrInitParams P;
    memset(&P, 0, sizeof(P));
    P.g = g;
    P.po = c->p[c->ip]; P.pom = c->p[c->ip ^ 1]; P.area = c->area_out; P.sc = c->sc;
    P.dxo = pr.dxo; P.dyo = pr.dyo; P.fnot = pr.fnot;
    for (int i = 0; i < g.nl * g.nl; ++i) P.amat[i] = pr.amatoc[i];
#define QG_CINIT(NLV) hipLaunchKernelGGL((k_constr_init<NLV>), dim3(1), dim3(64), 0, c->stream, P)
    QG_SWITCH_NL(g.nl, QG_CINIT, "k_constr_init");
#undef QG_CINIT
    HIPCHECK(hipGetLastError());
  }
  // qcomp, ocqbdy / atqzbd, merqcy for both time levels: src/q-gcm.F:719-731, 738-749
  for (int t = 0; t < 2; ++t) {
    QgQcompParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.g = g;
    Q.p = c->p[t ? c->ip ^ 1 : c->ip];
    Q.q = c->q[t ? c->iq ^ 1 : c->iq];
    Q.ddyn = c->ddynoc; Q.yporel = c->yporel;
    Q.dx2fac = (1.0 / (pr.dxo * pr.dxo)) / pr.fnot;
    Q.beta = pr.beta; Q.fnot = pr.fnot;
    for (int i = 0; i < g.nl * g.nl; ++i) Q.amat[i] = pr.amatoc[i];
    Q.ktopo = g.atm ? 0 : g.nl - 1;
    hipLaunchKernelGGL(k_qcomp, dim3((g.nx + 255) / 256, g.ny - 2, g.nl), dim3(256), 0, c->stream, Q);
    QgBdyParams B;
    fill_bdy_params(c, B);
    B.po = Q.p;
    B.qo = Q.q;
    const int nmax = g.nx > g.ny ? g.nx : g.ny;
    hipLaunchKernelGGL(k_ocqbdy, dim3((nmax + 255) / 256, g.cyc ? 2 : 4, g.nl), dim3(256), 0, c->stream, B);
    HIPCHECK(hipGetLastError());
  }
  return 0;
}

extern "C" int qgcm_hip_wekpo_from_tau(qgcm_hip_handle c, const double *tauxo, const double *tauyo) {
  if (check_ready(c, "qgcm_hip_wekpo_from_tau")) return 1;
  if (!tauxo || !tauyo) QG_FAIL("qgcm_hip_wekpo_from_tau: null argument");
  if (!c->whole) QG_FAIL("qgcm_hip_wekpo_from_tau: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  const int nyt = g.ny - 1;
  QgWekParams P;
  memset(&P, 0, sizeof(P));
  P.g = g;
  double *tx = nullptr, *ty = nullptr, *wt = nullptr;
  const bool own = !c->oml.on; // with the device mixed layer the stress and wekto live in its arrays
  if (own) {
    P.ldt = round_up(g.nxt, 16);
    if (dalloc(&tx, (size_t)g.ldx * g.ny) || dalloc(&ty, (size_t)g.ldx * g.ny) || dalloc(&wt, (size_t)P.ldt * nyt)) return 1;
  } else {
    tx = c->oml.taux; ty = c->oml.tauy; wt = c->oml.wekto;
    P.ldt = c->oml.ldt;
  }
  int rc = upload2d(c, tx, g.ldx, tauxo, g.nx, g.ny) || upload2d(c, ty, g.ldx, tauyo, g.nx, g.ny);
  if (!rc) {
    P.taux = tx; P.tauy = ty; P.wekto = wt; P.wekpo = c->wekpo;
    P.hxofac = 0.5 * (1.0 / (c->prm.dxo * c->prm.fnot)); // src/xfosubs.F:138 with rdxof0 of src/q-gcm.F:435
    hipLaunchKernelGGL(k_wekto, dim3((g.nxt + 255) / 256, nyt), dim3(256), 0, c->stream, P);
    hipLaunchKernelGGL(k_wekpo, dim3((g.nx + 255) / 256, g.ny), dim3(256), 0, c->stream, P);
    rc = hipGetLastError() != hipSuccess;
    if (rc) snprintf(g_err, sizeof(g_err), "qgcm_hip_wekpo_from_tau: launch failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = 1;
  }
  if (own) {
    hipFree(tx); hipFree(ty); hipFree(wt);
  }
  return rc;
}

extern "C" int qgcm_hip_prsamp(qgcm_hip_handle c, double *out) {
  if (check_ready(c, "qgcm_hip_prsamp")) return 1;
  if (!out) QG_FAIL("qgcm_hip_prsamp: null argument");
  if (!c->whole) QG_FAIL("qgcm_hip_prsamp: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  const int nl = g.nl;
  if (launch_area_sums(c)) return 1;
  double area[3 * QG_MAXL];
  HIPCHECK(hipMemcpyAsync(area, c->area_out, sizeof(double) * 3 * nl, hipMemcpyDeviceToHost, c->stream));
  const int nxco = (g.nx + 1) / 2, nyco = (g.ny + 1) / 2; // src/q-gcm.F:1974-1975
  const long oc = (long)(nyco - 1) * g.ldx + (nxco - 1);
  for (int k = 0; k < nl; ++k) {
    HIPCHECK(hipMemcpyAsync(out + k, c->p[c->ip] + g.fstride * k + oc, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipMemcpyAsync(out + nl + k, c->q[c->iq] + g.fstride * k + oc, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHECK(hipStreamSynchronize(c->stream));
  const double ocnorm = 1.0 / ((double)g.nxt * (double)(g.ny - 1)); // src/parameters_data.F:88
  for (int k = 0; k < nl; ++k) {
    out[2 * nl + k] = area[k] * ocnorm;          // pavgoc
    out[3 * nl + k] = area[2 * nl + k] * ocnorm; // qavgoc
  }
  out[4 * nl] = 1.0e30;
  out[4 * nl + 1] = -1.0e30;
  if (c->oml.on) {
    double v[14 + QG_MAXL];
    int ok = 0;
    if (qgcm_hip_valids(c, v, &ok)) return 1;
    out[4 * nl] = v[4];     // min, max of sst (layout of qgcm_hip_valids)
    out[4 * nl + 1] = v[5];
  }
  return 0;
}

static int one_step(qgcm_hip_ctx *c, int s) {
  if (c->oml.on && launch_oml(c, false)) return 1; // src/q-gcm.F:1232; its final reduction rides in launch_tend
  const bool fused_constr = !c->g.cyc && can_fuse_dst_unpack(c) && !c->no_fused_constr; // see ocinvq_impl
  if (check_ready(c, "qgcm_hip_steps")) return 1;
  // Leapfrog averaging (src/q-gcm.F:1345-1351) after this step: where the box ocean's fused kernels run, they store the
  // averaged level themselves - k_tend the interior qo (both levels are in its registers), k_dst64_unpack the new po and
  // the boundary qo (one extra read of this step's po) - instead of a pass of its own over six fields (133 MB at 5 km,
  // 21 us every 25 steps); the integrals dpioc follow in a one-thread launch.  Same expressions: bitwise the same fields.
  const bool avg = (s - 1) % c->avg_period == 0;
  const bool avg_box = fused_constr && c->g.nl <= 4 && tend_wtq(c);
  const bool avg_cyc = c->g.cyc && can_fuse_fft3_unpack(c) && !c->no_fused_constr; // (ocinvq_impl: launch_fft3_unpack(c, true))
  c->avg_now = avg && (avg_box || avg_cyc) && !c->no_fused_avg;
  const bool avg_fused = c->avg_now;
  int rc = launch_tend(c, fused_constr, c->oml.on);
  if (!rc) {
    c->iq ^= 1; // as qgcm_hip_qgostep
    rc = ocinvq_impl(c, true, true); // ocqbdy fused into the unpack kernel
  }
  c->avg_now = false;
  if (rc) return 1;
  if (avg_fused) {
    KTimer t(c, KN_LFAVG);
    hipLaunchKernelGGL(k_lf_average_scalars, dim3(1), dim3(64), 0, c->stream, c->sc, c->g.nl, c->g.cyc ? 1 : 0);
    HIPCHECK(hipGetLastError());
    if (c->oml.on && launch_oml_average(c)) return 1; // the mixed-layer temperature keeps its own (one-field) pass
  } else if (avg) {
    if (qgcm_hip_lf_average(c)) return 1; // incl. sst when the mixed layer is on
  }
  if (c->profiling) {
    // an empty launch bracketed like the kernels (see qgcm_hip_profile_steps)
    KTimer t(c, KN_NOOP);
    hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, c->stream);
  }
  if (c->evkid.size() > 4000) drain_timers(c); // bounds the event pool on long profiling runs
  return 0;
}

// the sst buffers rotate with period 3 (sst -> sstm -> spare -> sst), one position per step
static void oml_rotate(qgcm_hip_ctx *c, int nsteps) {
  for (int r = 0; r < nsteps % 3; ++r) {
    const int spare = 3 - c->oml.is - c->oml.ism;
    c->oml.ism = c->oml.is;
    c->oml.is = spare;
  }
}

// One captured block of B consecutive steps. B is even, so both buffer rotations are back where they started
// after the block; the key carries everything else a captured step depends on: the position in the averaging
// cycle and the sst rotation. 50-step blocks serve long runs (one graph for the ocean: 50 = 2 x 25); what is left
// (< 50 steps) goes out as ONE more block of the largest even length - a run of n steps is n / 50 + 1 graph
// launches and at most one eager step (round 2 cut the tail into 10-step blocks: two replays for the driver's
// 20-step window, each paying the ~10-16 us host-side floor of a replay).  The cache is bounded: a caller
// that asks for ever new block lengths / phases evicts everything once kMaxGraphs is reached.
static const int kGraphBlock50 = 50;
static const size_t kMaxGraphs = 96;

static int get_graph(qgcm_hip_ctx *c, int s0, int B, hipGraphExec_t *out) {
  const int phase = (s0 - 1) % c->avg_period;
  const int omk = c->oml.on ? 1 + 3 * c->oml.is + c->oml.ism : 0; // mixed layer on/off and its buffer rotation
  const long long key = ((long long)B << 32) | (omk << 24) | (c->ip << 16) | (c->iq << 8) | phase;
  auto it = c->graphs.find(key);
  if (it != c->graphs.end()) {
    it->second.used = c->graph_call;
    *out = it->second.exec;
    return 0;
  }
  if (c->graphs.size() >= kMaxGraphs) {
    // evict what the CURRENT call has not touched (a dry pass that builds the graphs of a timed call must not lose
    // them to its own later blocks); least recently used first, down to half the cache
    HIPCHECK(hipStreamSynchronize(c->stream)); // replays of the graphs about to be destroyed may still be queued
    std::vector<std::pair<unsigned long, long long>> old;
    for (auto &kv : c->graphs)
      if (kv.second.used != c->graph_call) old.push_back({kv.second.used, kv.first});
    std::sort(old.begin(), old.end());
    for (size_t i = 0; i < old.size() && c->graphs.size() > kMaxGraphs / 2; ++i) {
      hipGraphExecDestroy(c->graphs[old[i].second].exec);
      c->graphs.erase(old[i].second);
    }
  }
  hipGraph_t graph;
  const int ip0 = c->ip, iq0 = c->iq, is0 = c->oml.is, ism0 = c->oml.ism;
  HIPCHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int s = s0; s < s0 + B && !rc; ++s) rc = one_step(c, s);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  c->ip = ip0; // capture does not execute: restore the rotation state
  c->iq = iq0;
  c->oml.is = is0;
  c->oml.ism = ism0;
  if (rc) return 1;
  HIPCHECK(e);
  hipGraphExec_t exec;
  HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIPCHECK(hipGraphDestroy(graph));
  // (the first replay of a fresh executable graph otherwise pays for its upload inside the caller's window)
  if (hipGraphUpload(exec, c->stream) != hipSuccess) (void)hipGetLastError();
  c->graphs[key] = {exec, c->graph_call};
  *out = exec;
  return 0;
}

// dry = true only instantiates the graphs the run will replay (so that a timed region does not pay for it)
static int steps_impl(qgcm_hip_ctx *c, int s0, int n, bool dry) {
  int s = s0;
  if (!dry) c->graph_call++; // (a dry pass belongs to the call that follows it: qgcm_hip_time_steps, prepare + steps)
  const int is0 = c->oml.is, ism0 = c->oml.ism;
  while (!c->profiling && n >= 2) {
    const int B = n >= kGraphBlock50 ? kGraphBlock50 : (n & ~1);
    hipGraphExec_t ge;
    if (get_graph(c, s, B, &ge)) return 1;
    if (!dry) HIPCHECK(hipGraphLaunch(ge, c->stream));
    s += B;
    n -= B;
    if (c->oml.on) oml_rotate(c, B); // the p and q rotations are back where they started, sst has moved on
  }
  if (dry) {
    c->oml.is = is0;
    c->oml.ism = ism0;
    return 0;
  }
  for (; n > 0; --n, ++s)
    if (one_step(c, s)) return 1;
  return 0;
}

extern "C" int qgcm_hip_steps(qgcm_hip_handle c, int s0, int n) {
  if (check_ready(c, "qgcm_hip_steps")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_steps: this handle is a y-slab; drive it with the slab building blocks");
  if (s0 < 1 || n < 0) QG_FAIL("qgcm_hip_steps: bad step range");
  return steps_impl(c, s0, n, false);
}

// ---------------------------------------------------------------------------
// atmosphere: the same kernels under the reference's names
// ---------------------------------------------------------------------------
static int check_atm(qgcm_hip_ctx *c, const char *who) {
  if (check_ready(c, who)) return 1;
  if (!c->g.atm) QG_FAIL("%s: the handle was not created with qgcm_hip_params.atmos = 1", who);
  return 0;
}

extern "C" int qgcm_hip_qgastep(qgcm_hip_handle c) {
  if (check_atm(c, "qgcm_hip_qgastep")) return 1;
  return qgcm_hip_qgostep(c);
}

extern "C" int qgcm_hip_atinvq(qgcm_hip_handle c) {
  if (check_atm(c, "qgcm_hip_atinvq")) return 1;
  return ocinvq_impl(c, false);
}

extern "C" int qgcm_hip_atqzbd(qgcm_hip_handle c) {
  if (check_atm(c, "qgcm_hip_atqzbd")) return 1;
  return launch_ocqbdy(c);
}

extern "C" int qgcm_hip_get_bsums(qgcm_hip_handle c, double *b) {
  if (check_ready(c, "qgcm_hip_get_bsums")) return 1;
  if (!b) QG_FAIL("qgcm_hip_get_bsums: null argument");
  if (!c->g.cyc) QG_FAIL("qgcm_hip_get_bsums: only the zonally cyclic ocean and the atmosphere have boundary line sums");
  QgScalars h;
  HIPCHECK(hipMemcpyAsync(&h, c->sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  const int nl = c->g.nl;
  for (int k = 0; k < nl; ++k) {
    b[k] = h.ajisoc[k];
    b[nl + k] = h.ajinoc[k];
    b[2 * nl + k] = h.ap5soc[k];
    b[3 * nl + k] = h.ap5noc[k];
  }
  return 0;
}

// Coupled run with the forcing held between calls: the main loop of src/q-gcm.F:1220-1268 without xforc / oml / aml.
// The ocean steps of the window are queued on the ocean handle's stream and the atmospheric steps on the
// atmosphere's: with the forcing frozen the two halves do not exchange data inside the window, so the streams need
// no cross dependencies and the GPU overlaps the atmosphere's small kernels with the ocean's.
// Two handles that step side by side on one GPU (qgcm_hip_coupled_steps: the ocean and the atmosphere of a coupled run)
// queue for the same wave slots: the atmosphere's twelve small dependent launches per ocean step then advance only as
// slots of the ocean's chip-filling launches retire (107 us per ocean step at NAtl 5 km, where the ocean alone takes 68
// and the atmosphere's three steps alone 67).  Given disjoint CU ranges (hipExtStreamCreateWithCUMask) each half keeps
// its own pace: 94.6 us with 96 CUs for the atmosphere and 160 for the ocean (profiles/r4_coupled_cu_masks.log).
// The handle's stream is REPLACED (not a second stream beside it: the runtime multiplexes streams onto few hardware
// queues, and masked streams that share one ran four times slower); count = 0 gives the unrestricted stream back.
extern "C" int qgcm_hip_set_cu_range(qgcm_hip_handle c, int first, int count) {
  if (!c) QG_FAIL("qgcm_hip_set_cu_range: null handle");
  if (c->sc_comm) QG_FAIL("qgcm_hip_set_cu_range: not for a handle with a communicator (its exchanges are ordered on the stream it has)");
  int ncu = 0, dev = 0;
  HIPCHECK(hipGetDevice(&dev));
  HIPCHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  if (first < 0 || count < 0 || first + count > ncu || ncu > 1024) QG_FAIL("qgcm_hip_set_cu_range: CUs %d..%d outside 0..%d", first, first + count - 1, ncu - 1);
  if (first == c->cu_first && count == c->cu_count) return 0;
  HIPCHECK(hipStreamSynchronize(c->stream));
  drop_graphs(c); // (executable graphs are uploaded to the stream they replay on)
  hipStream_t ns = nullptr;
  if (count > 0) {
    uint32_t mask[32];
    memset(mask, 0, sizeof(mask));
    for (int i = first; i < first + count; ++i) mask[i / 32] |= 1u << (i % 32);
    HIPCHECK(hipExtStreamCreateWithCUMask(&ns, (uint32_t)((ncu + 31) / 32), mask));
  } else
    HIPCHECK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
  HIPCHECK(hipStreamDestroy(c->stream));
  c->stream = ns;
  c->cu_first = first;
  c->cu_count = count;
  return 0;
}

extern "C" int qgcm_hip_coupled_steps(qgcm_hip_handle oc, qgcm_hip_handle atm, int nt0, int n, int nstr) {
  if (nt0 < 1 || n < 0 || nstr < 1) QG_FAIL("qgcm_hip_coupled_steps: bad step range");
  if (oc && check_ready(oc, "qgcm_hip_coupled_steps")) return 1;
  if (atm && check_atm(atm, "qgcm_hip_coupled_steps")) return 1;
  if (oc && (oc->g.atm || !oc->whole)) QG_FAIL("qgcm_hip_coupled_steps: first handle must be a whole-domain ocean");
  if (oc && n > 0) {
    // ocean steps s with nt = 1 + (s-1)*nstr in [nt0, nt0+n-1]   (mod(nt,nstr) == 1, src/q-gcm.F:1222; nstr = 1 never
    // steps the ocean in the reference - SURVEY 8d caveat - and neither does this)
    if (nstr > 1) {
      const int sfirst = (nt0 - 1 + nstr - 1) / nstr + 1; // smallest s with 1+(s-1)*nstr >= nt0
      const int slast = (nt0 + n - 2) / nstr + 1;         // largest s with 1+(s-1)*nstr <= nt0+n-1
      if (slast >= sfirst && steps_impl(oc, sfirst, slast - sfirst + 1, false)) return 1;
    }
  }
  if (atm && n > 0 && steps_impl(atm, nt0, n, false)) return 1;
  return 0;
}

extern "C" int qgcm_hip_sync(qgcm_hip_handle c) {
  if (!c) QG_FAIL("qgcm_hip_sync: null handle");
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_helmholtz(qgcm_hip_handle c, double *wrk, const double *boc) {
  if (check_ready(c, "qgcm_hip_helmholtz")) return 1;
  if (!wrk || !boc) QG_FAIL("qgcm_hip_helmholtz: null argument");
  if (!c->whole) QG_FAIL("qgcm_hip_helmholtz: only for a handle that owns the whole domain");
  const QgGeom &g = c->g;
  // pivots for this boc (box: boc(i-1) multiplies sine wavenumber i-1, src/ocisubs.F:470-478)
  {
    ThomasTabHost T;
    build_pivots(g, c->prm.aoc, boc, T);
    if (upload_pivots(c, c->tt_tmp, T)) return 1;
  }
  // box: interior columns i=2..nx-1 of every row -> wrk(c = i-2, j); cyclic: columns 1..nxto -> wrk(c = i-1, j)
  const int coff = g.cyc ? 0 : 1;
  HIPCHECK(hipMemcpy2DAsync(c->wrk, (size_t)g.ldw * 8, wrk + coff, (size_t)g.nx * 8, (size_t)g.nk * 8, (size_t)g.ny,
                            hipMemcpyHostToDevice, c->stream));
  if (launch_dst(c, c->wrk, 1, false)) return 1;
  if (launch_thomas(c, c->wrk, c->tt_tmp, 1, 0, nullptr, nullptr, 0, 1, 0, nullptr)) return 1;
  if (launch_dst(c, c->wrk, 1, true)) return 1;
  HIPCHECK(hipMemcpy2DAsync(wrk + coff, (size_t)g.nx * 8, c->wrk, (size_t)g.ldw * 8, (size_t)g.nk * 8, (size_t)g.ny,
                            hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  // box: solid-boundary values are zero (src/ocisubs.F:496-509); cyclic: E = W (:604)
  for (int j = 0; j < g.ny; ++j) {
    if (g.cyc) {
      wrk[(size_t)j * g.nx + g.nx - 1] = wrk[(size_t)j * g.nx];
    } else {
      wrk[(size_t)j * g.nx] = 0.0;
      wrk[(size_t)j * g.nx + g.nx - 1] = 0.0;
    }
  }
  for (int i = 0; i < g.nx; ++i) {
    wrk[i] = 0.0;
    wrk[(size_t)(g.ny - 1) * g.nx + i] = 0.0;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// y-slab building blocks
// ---------------------------------------------------------------------------
extern "C" int qgcm_hip_local_rows(qgcm_hip_handle c, int *nyl, int *joff, int *jlo, int *jhi) {
  if (!c) QG_FAIL("qgcm_hip_local_rows: null handle");
  if (nyl) *nyl = c->g.ny;
  if (joff) *joff = c->g.joff;
  if (jlo) *jlo = c->g.jlo;
  if (jhi) *jhi = c->g.jhi;
  return 0;
}

extern "C" int qgcm_hip_row_transform(qgcm_hip_handle c, int inverse) {
  if (check_ready(c, "qgcm_hip_row_transform")) return 1;
  return launch_dst(c, c->wrk, c->g.nl, inverse != 0);
}

extern "C" int qgcm_hip_wrk_fill(qgcm_hip_handle c, double value) {
  if (check_ready(c, "qgcm_hip_wrk_fill")) return 1;
  const QgGeom &g = c->g;
  HIPCHECK(hipMemsetAsync(c->wrk, 0, sizeof(double) * g.wstride * g.nl, c->stream));
  hipLaunchKernelGGL(k_fill_rows, dim3((g.nk + 255) / 256, g.jr1 - g.jr0 + 1, g.nl), dim3(256), 0, c->stream, c->wrk,
                     g.wstride, g.ldw, g.nk, g.jr0, g.jr1, g.nl, value);
  HIPCHECK(hipGetLastError());
  return 0;
}

// counterpart of qgcm_hip_wrk_get: the rows jr0..jr1 of a (nxpo, nyl, nlo) host block into the work array (box: the
// interior columns 2..nxpo-1; cyclic: columns 1..nxto), everything else of the work array zero
extern "C" int qgcm_hip_wrk_set(qgcm_hip_handle c, const double *wrk) {
  if (check_ready(c, "qgcm_hip_wrk_set")) return 1;
  if (!wrk) QG_FAIL("qgcm_hip_wrk_set: null argument");
  const QgGeom &g = c->g;
  HIPCHECK(hipMemsetAsync(c->wrk, 0, sizeof(double) * g.wstride * g.nl, c->stream));
  const int coff = g.cyc ? 0 : 1;
  for (int m = 0; m < g.nl; ++m)
    HIPCHECK(hipMemcpy2DAsync(c->wrk + g.wstride * m + (size_t)(g.jr0 - 1) * g.ldw, (size_t)g.ldw * 8,
                              wrk + (size_t)g.nx * g.ny * m + (size_t)(g.jr0 - 1) * g.nx + coff, (size_t)g.nx * 8, (size_t)g.nk * 8,
                              (size_t)(g.jr1 - g.jr0 + 1), hipMemcpyHostToDevice, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_wrk_get(qgcm_hip_handle c, double *wrk) {
  if (check_ready(c, "qgcm_hip_wrk_get")) return 1;
  if (!wrk) QG_FAIL("qgcm_hip_wrk_get: null argument");
  const QgGeom &g = c->g;
  memset(wrk, 0, sizeof(double) * (size_t)g.nx * g.ny * g.nl);
  const int coff = g.cyc ? 0 : 1;
  for (int m = 0; m < g.nl; ++m)
    HIPCHECK(hipMemcpy2DAsync(wrk + (size_t)g.nx * g.ny * m + (size_t)(g.jr0 - 1) * g.nx + coff, (size_t)g.nx * 8,
                              c->wrk + g.wstride * m + (size_t)(g.jr0 - 1) * g.ldw, (size_t)g.ldw * 8, (size_t)g.nk * 8,
                              (size_t)(g.jr1 - g.jr0 + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHECK(hipStreamSynchronize(c->stream));
  if (g.cyc)
    for (int m = 0; m < g.nl; ++m)
      for (int j = 0; j < g.ny; ++j) wrk[((size_t)m * g.ny + j) * g.nx + g.nx - 1] = wrk[((size_t)m * g.ny + j) * g.nx];
  return 0;
}

extern "C" int qgcm_hip_area_integrals(qgcm_hip_handle c, double *xin) {
  if (check_ready(c, "qgcm_hip_area_integrals")) return 1;
  if (!xin) QG_FAIL("qgcm_hip_area_integrals: null argument");
  if (c->g.cyc) QG_FAIL("qgcm_hip_area_integrals: box ocean only (spectral area integrals of the sine series)");
  QgConstrParams P;
  fill_constr_params(c, P);
#define QG_XIN(NLV) hipLaunchKernelGGL((k_xin_only<NLV>), dim3(1), dim3(64), 0, c->stream, P)
  QG_SWITCH_NL(c->g.nl, QG_XIN, "k_xin_only");
#undef QG_XIN
  HIPCHECK(hipGetLastError());
  return qgcm_hip_get_inv_diag(c, xin, nullptr);
}

extern "C" int qgcm_hip_thomas_msg_len(qgcm_hip_handle c) { return c ? (int)slab_msg_len_oml(c) : 0; }
extern "C" int qgcm_hip_oml_msg_len(qgcm_hip_handle c) { return c ? 3 : 0; }

extern "C" int qgcm_hip_thomas_const_len(qgcm_hip_handle c) { return c ? TH_CST * c->g.nl * c->g.ldw : 0; }

extern "C" int qgcm_hip_thomas_consts(qgcm_hip_handle c, double *dst_dev) {
  if (check_ready(c, "qgcm_hip_thomas_consts")) return 1;
  if (!dst_dev) QG_FAIL("qgcm_hip_thomas_consts: null buffer");
  HIPCHECK(hipMemcpyAsync(dst_dev, c->slabDE, sizeof(double) * TH_CST * c->g.nl * c->g.ldw, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

extern "C" int qgcm_hip_set_thomas_consts(qgcm_hip_handle c, const double *gath_dev, int nranks) {
  if (check_ready(c, "qgcm_hip_set_thomas_consts")) return 1;
  if (!gath_dev || nranks < 1 || nranks > 64) QG_FAIL("qgcm_hip_set_thomas_consts: bad argument");
  const size_t n = (size_t)TH_CST * c->g.nl * c->g.ldw * nranks;
  drop_graphs(c);
  if (c->th_cgath_ranks != nranks) {
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (c->th_cgath) hipFree(c->th_cgath);
    c->th_cgath = nullptr;
    if (dalloc(&c->th_cgath, n)) return 1;
    c->th_cgath_ranks = nranks;
  }
  HIPCHECK(hipMemcpyAsync(c->th_cgath, gath_dev, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

extern "C" int qgcm_hip_thomas_phase(qgcm_hip_handle c, int phase, const double *gath_dev, double *send_dev, int rank,
                                     int nranks) {
  if (check_ready(c, "qgcm_hip_thomas_phase")) return 1;
  if (phase < 1 || phase > 2) QG_FAIL("qgcm_hip_thomas_phase: phase must be 1 or 2");
  if ((phase == 1 && !send_dev) || (phase == 2 && !gath_dev)) QG_FAIL("qgcm_hip_thomas_phase: missing buffer");
  if (nranks > 64) QG_FAIL("qgcm_hip_thomas_phase: at most 64 slabs");
  return launch_thomas(c, c->wrk, c->tt, c->g.nl, phase, gath_dev, send_dev, rank, nranks, 0, nullptr);
}

extern "C" int qgcm_hip_constr(qgcm_hip_handle c) {
  if (check_ready(c, "qgcm_hip_constr")) return 1;
  if (!c->homog_set) QG_FAIL("qgcm_hip_constr: homogeneous solutions not set");
  return launch_constr(c);
}

extern "C" int qgcm_hip_unpack(qgcm_hip_handle c, int fuse_ocqbdy) {
  if (check_ready(c, "qgcm_hip_unpack")) return 1;
  if (launch_unpack(c, fuse_ocqbdy != 0)) return 1;
  c->ip ^= 1;
  return 0;
}

extern "C" int qgcm_hip_halo_msg_len(qgcm_hip_handle c) { return c ? (int)halo_msg_len(c) : 0; }

extern "C" int qgcm_hip_halo_pack(qgcm_hip_handle c, double *to_lower_dev, double *to_upper_dev) {
  if (check_ready(c, "qgcm_hip_halo_pack")) return 1;
  const QgGeom &g = c->g;
  dim3 grid((g.ldx + 255) / 256, 4 * g.nl, 2);
  hipLaunchKernelGGL(k_halo_pack, grid, dim3(256), 0, c->stream, g, (const double *)c->p[c->ip],
                     (const double *)c->q[c->iq], to_lower_dev, to_upper_dev);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int qgcm_hip_halo_unpack(qgcm_hip_handle c, const double *from_lower_dev, const double *from_upper_dev) {
  if (check_ready(c, "qgcm_hip_halo_unpack")) return 1;
  const QgGeom &g = c->g;
  if ((from_lower_dev && g.jlo < 4) || (from_upper_dev && g.ny - g.jhi < 3)) QG_FAIL("qgcm_hip_halo_unpack: no halo rows on that side");
  dim3 grid((g.ldx + 255) / 256, 4 * g.nl, 2);
  hipLaunchKernelGGL(k_halo_unpack, grid, dim3(256), 0, c->stream, g, c->p[c->ip], c->q[c->iq], from_lower_dev,
                     from_upper_dev);
  HIPCHECK(hipGetLastError());
  return 0;
}

// edge rows of the new sst (first / last three owned T rows) appended to the halo messages
static int oml_halo_pack(qgcm_hip_ctx *c, double *to_lo, double *to_hi) {
  if (!c->oml.on || (!to_lo && !to_hi)) return 0;
  const size_t off = (size_t)4 * c->g.nl * c->g.ldx;
  const int jT1 = (c->g.jhi + c->g.joff == c->g.nyg) ? c->g.jhi - 1 : c->g.jhi;
  hipLaunchKernelGGL(k_oml_halo, dim3((c->g.nxt + 255) / 256, 3, 2), dim3(256), 0, c->stream, c->oml.sst[c->oml.is], c->oml.ldt, c->g.nxt,
                     c->g.ny - 1, c->g.jlo, jT1, to_lo ? to_lo + off : nullptr, to_hi ? to_hi + off : nullptr, (const double *)nullptr,
                     (const double *)nullptr, 0);
  HIPCHECK(hipGetLastError());
  return 0;
}

extern "C" int qgcm_hip_slab_stage(qgcm_hip_handle c, int stage, double *a, double *b, double *cc, int rank, int nranks,
                                   int flags) {
  switch (stage) {
    case 1:
    case 4:   // the part of stage 1 that needs no halo row: the inner tile rows of the tendency launch
    case 5:   // the rest of stage 1: outer tile rows + edge work, forward rows, summary sweep
    case 6:   // ... its first half alone: the outer tile rows + edge work of the tendency launch
    case 7: { // ... and its second half: forward rows, summary sweep (after stages 4 AND 6)
      if (stage != 4 && !a) QG_FAIL("qgcm_hip_slab_stage: stage %d needs the send buffer", stage);
      if (check_ready(c, "qgcm_hip_slab_stage")) return 1;
      if (stage != 1 && (c->oml.on || !tend_can_split(c)))
        QG_FAIL("qgcm_hip_slab_stage: stages 4 - 7 need a slab of at least three tile rows (%d rows each) and the mixed layer off", TEND_TY);
      // cyclic: the boundary line sums of the tendency launch go straight into the tail of the step message
      if (c->g.cyc && stage != 4 && stage != 7) c->bpart_out = a + (size_t)TH_MSG * c->g.nl * c->g.ldw;
      // box fast path: as in qgcm_hip_steps the leapfrog of dpioc is done by the tendency launch and the constraint
      // solve by the extra wave of the fused inverse-transform kernel of stage 2 (no k_constr_box launch)
      if (!c->homog_set) QG_FAIL("qgcm_hip_slab_stage: homogeneous solutions not set");
      // (with the mixed layer on, xon(1) is only complete after the all-gather: dpioc is stepped in stage 2 then)
      const bool fused_constr = can_fuse_dst_unpack(c) && !c->no_fused_constr && !c->oml.on; // can_fuse: box ocean only
      const int rc = stage == 7 ? 0 : launch_tend(c, fused_constr, false, stage == 1 ? TEND_ALL : stage == 4 ? TEND_INNER : TEND_OUTER);
      c->bpart_out = nullptr;
      if (rc) return 1;
      if (stage == 4 || stage == 6) return 0;
      c->iq ^= 1; // as qgcm_hip_qgostep
      if (qgcm_hip_row_transform(c, 0)) return 1;
      if (qgcm_hip_thomas_phase(c, 1, nullptr, a, rank, nranks)) return 1;
      if (c->oml.on) { // the three sums of this slab's entoc pass (stage 11) ride at the end of the step message
        QgOmlParams OP;
        fill_oml_params(c, OP);
        hipLaunchKernelGGL(k_oml_sum3, dim3(1), dim3(OML_NT), 0, c->stream, (const double *)OP.partB, OP.nblkB, a + slab_msg_len(c->g));
        HIPCHECK(hipGetLastError());
      }
      return 0;
    }
    case 2:
      if (qgcm_hip_thomas_phase(c, 2, a, nullptr, rank, nranks)) return 1;
      if (c->oml.on) { // xon(1), enisoc(1) / eninoc(1), monitors from every rank's sums, before the constraints use them
        if (!c->oml_gath) QG_FAIL("qgcm_hip_slab_stage: the mixed layer's stages 10 / 11 have not run this step");
        QgOmlFinal F;
        fill_oml_final(c, F, true);
        hipLaunchKernelGGL(k_oml_final_slab, dim3(1), dim3(64), 0, c->stream, c->oml_gath, (const double *)(a + slab_msg_len(c->g)),
                           (long)slab_msg_len_oml(c), nranks, F);
        HIPCHECK(hipGetLastError());
      }
      if (can_fuse_dst_unpack(c)) {
        const bool fused_constr = !c->no_fused_constr && !c->oml.on;
        if (!fused_constr && qgcm_hip_constr(c)) return 1; // area integrals of the whole basin came with the slab summaries
        // the fused kernel also writes the halo messages (first / last three owned rows of po, edge row of qo)
        if (launch_dst_unpack(c, true, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr, fused_constr)) return 1;
        c->ip ^= 1;
        return oml_halo_pack(c, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr);
      }
      if (can_fuse_fft3_unpack(c)) { // long rows: inverse rows + unpack + halo messages in one launch (k_fft3_unpack.h)
        if (qgcm_hip_constr(c)) return 1;
        if (launch_fft3_unpack(c, false, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr)) return 1;
        c->ip ^= 1;
        return oml_halo_pack(c, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr);
      }
      if (dst_box_rides_constr(c)) { // box: the constraint solve rides in the inverse-row launch
        if (launch_dst(c, c->wrk, c->g.nl, true, 0, nullptr, false, true)) return 1;
      } else {
        if (qgcm_hip_constr(c)) return 1;
        if (qgcm_hip_row_transform(c, 1)) return 1;
      }
      // the unpack launch also writes the halo messages (first / last three owned rows of po, edge row of qo)
      if (check_ready(c, "qgcm_hip_slab_stage") || launch_unpack(c, true, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr)) return 1;
      c->ip ^= 1;
      return oml_halo_pack(c, nranks > 1 ? b : nullptr, nranks > 1 ? cc : nullptr);
    case 3:
      if (nranks > 1 && qgcm_hip_halo_unpack(c, a, b)) return 1;
      if (nranks > 1 && c->oml.on) { // the neighbours' edge rows of the new sst sit at the end of the halo messages
        const size_t off = (size_t)4 * c->g.nl * c->g.ldx;
        const int jT1 = (c->g.jhi + c->g.joff == c->g.nyg) ? c->g.jhi - 1 : c->g.jhi;
        hipLaunchKernelGGL(k_oml_halo, dim3((c->g.nxt + 255) / 256, 3, 2), dim3(256), 0, c->stream, c->oml.sst[c->oml.is], c->oml.ldt,
                           c->g.nxt, c->g.ny - 1, c->g.jlo, jT1, (double *)nullptr, (double *)nullptr,
                           a ? (const double *)(a + off) : nullptr, b ? (const double *)(b + off) : nullptr, 1);
        HIPCHECK(hipGetLastError());
      }
      if (flags & 1) return qgcm_hip_lf_average(c);
      return 0;
    // ---- ocean mixed layer on y-slabs: `call oml` (src/q-gcm.F:1232) comes before qgostep and needs the basin-wide
    // mean entrainment half way through - one more (three numbers per rank) all-gather per step
    case 10: // new sst, raw entrainment; a = this slab's three sums (send buffer of the mixed layer's all-gather)
      if (oml_ready(c, "qgcm_hip_slab_stage")) return 1;
      if (!a) QG_FAIL("qgcm_hip_slab_stage: stage 10 needs the send buffer");
      return launch_oml_a(c, a);
    case 11: // a = the gathered sums (3, nranks): entoc, rotation of the sst buffers
      if (oml_ready(c, "qgcm_hip_slab_stage")) return 1;
      if (!a) QG_FAIL("qgcm_hip_slab_stage: stage 11 needs the gathered sums");
      c->oml_gath = a;
      return launch_oml_b(c, a, nranks);
    default: QG_FAIL("qgcm_hip_slab_stage: stage must be 1..7, 10 or 11");
  }
}

// ---------------------------------------------------------------------------
// y-slab steps with the exchanges issued from here (RCCL on the library's stream)
// ---------------------------------------------------------------------------
#define NCCLCHECK(m, expr)                                                                           \
  do {                                                                                               \
    ncclResult_t r_ = (expr);                                                                        \
    if (r_ != ncclSuccess) QG_FAIL("%s failed: %s (%s:%d)", #expr, (m)->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

extern "C" int qgcm_hip_comm_unique_id(char *id, int nbytes) {
  if (!id || nbytes < (int)sizeof(ncclUniqueId)) QG_FAIL("qgcm_hip_comm_unique_id: need a buffer of %d bytes", (int)sizeof(ncclUniqueId));
  QgRccl *api = qg_rccl(g_err, sizeof(g_err));
  if (!api) return 1;
  ncclUniqueId u;
  NCCLCHECK(api, api->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int qgcm_hip_comm_init(qgcm_hip_handle c, const char *id, int nbytes, int rank, int nranks) {
  if (check_ready(c, "qgcm_hip_comm_init")) return 1;
  if (!id || nbytes < (int)sizeof(ncclUniqueId)) QG_FAIL("qgcm_hip_comm_init: need the %d-byte id of qgcm_hip_comm_unique_id", (int)sizeof(ncclUniqueId));
  if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) QG_FAIL("qgcm_hip_comm_init: bad rank %d of %d (at most 64 slabs)", rank, nranks);
  if (c->sc_comm) QG_FAIL("qgcm_hip_comm_init: this handle already has a communicator");
  const QgGeom &g = c->g;
  // the slab must be the rank-th piece of the basin: first rank owns global row 1, last rank row nyg
  if ((rank == 0) != (g.joff + g.jlo == 1) || (rank == nranks - 1) != (g.joff + g.jhi == g.nyg))
    QG_FAIL("qgcm_hip_comm_init: slab rows %d..%d of %d do not fit rank %d of %d", g.joff + g.jlo, g.joff + g.jhi, g.nyg, rank, nranks);
  QgRccl *api = qg_rccl(g_err, sizeof(g_err));
  if (!api) return 1;
  HIPCHECK(hipSetDevice(c->device));
  QgSlabComm *m = new QgSlabComm;
  m->api = api;
  m->rank = rank;
  m->nranks = nranks;
  const char *hp = getenv("QGCM_HIP_HALO_P2P");
  m->halo_p2p = hp ? atoi(hp) != 0 : true; // default: the two neighbours only (an all-gather moves nranks x the bytes)
  m->th_len = slab_msg_len_oml(c);
  m->halo_len = halo_msg_len(c);
  c->sc_comm = m; // owned by the handle from here on (freed in qgcm_hip_destroy)
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  NCCLCHECK(api, api->CommInitRank(&m->comm, nranks, u, rank));
  struct { double **p; size_t n; } bufs[] = {{&m->th_send, m->th_len}, {&m->th_gath, m->th_len * nranks},
                                             {&m->h_send, 2 * m->halo_len}, {&m->h_gath, 2 * m->halo_len * nranks},
                                             {&m->oml_send, 4}, {&m->oml_gath, (size_t)4 * nranks}};
  for (auto &b : bufs) {
    HIPCHECK(hipMalloc((void **)b.p, b.n * sizeof(double)));
    HIPCHECK(hipMemsetAsync(*b.p, 0, b.n * sizeof(double), c->stream));
  }
  // the right-hand-side independent part of the slab summaries (D, E, SP, SQ) is exchanged once
  if (nranks > 1) {
    const size_t nc = (size_t)TH_CST * g.nl * g.ldw;
    if (c->th_cgath) hipFree(c->th_cgath);
    c->th_cgath = nullptr;
    if (dalloc(&c->th_cgath, nc * nranks)) return 1;
    c->th_cgath_ranks = nranks;
    NCCLCHECK(api, api->AllGather(c->slabDE, c->th_cgath, nc, ncclDouble, m->comm, c->stream));
  }
  HIPCHECK(hipStreamSynchronize(c->stream));
  const char *gm = getenv("QGCM_HIP_SLAB_GRAPH");
  c->slab_graph_mode = gm && atoi(gm) != 0;
  return 0;
}

extern "C" int qgcm_hip_comm_set_halo_p2p(qgcm_hip_handle c, int on) {
  if (!c || !c->sc_comm) QG_FAIL("qgcm_hip_comm_set_halo_p2p: no communicator (qgcm_hip_comm_init)");
  HIPCHECK(hipStreamSynchronize(c->stream));
  if (c->sc_comm->halo_p2p != (on != 0)) drop_graphs(c); // captured steps contain the old exchange
  c->sc_comm->halo_p2p = (on != 0);
  return 0;
}

// Halo exchange of step s overlapped with the inner tile rows of step s+1's tendency launch: the exchange and the halo
// unpack go to a second stream; the handle's stream runs k_tend's inner tile rows (stage 4), waits for the halo rows,
// and goes on with the outer tile rows (stage 5).  Same results, bit for bit.  Not with the mixed layer on (its first
// stage needs the neighbours' sst rows), not across a leapfrog averaging, not for slabs of fewer than three tile rows:
// those steps keep the plain order.
extern "C" int qgcm_hip_comm_set_overlap(qgcm_hip_handle c, int on) {
  if (!c || !c->sc_comm) QG_FAIL("qgcm_hip_comm_set_overlap: no communicator (qgcm_hip_comm_init)");
  QgSlabComm *m = c->sc_comm;
  HIPCHECK(hipStreamSynchronize(c->stream));
  if (m->overlap != (on != 0)) drop_graphs(c); // captured steps contain the old order
  if (on && !m->cstream) {
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamCreateWithFlags(&m->cstream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&m->ev_halo, hipEventDisableTiming));
  }
  m->overlap = (on != 0);
  return 0;
}

// Measurement aid: the step's collectives issued back to back on the handle's stream, HIP-event timed
// (collective over all ranks). us[0] = all-gather of the slab summaries, us[1] = halo rows as one all-gather,
// us[2] = halo rows as grouped send/recv with the two neighbours.
extern "C" int qgcm_hip_comm_probe(qgcm_hip_handle c, int reps, double *us) {
  if (!c || !c->sc_comm) QG_FAIL("qgcm_hip_comm_probe: no communicator (qgcm_hip_comm_init)");
  if (!us || reps < 1) QG_FAIL("qgcm_hip_comm_probe: bad argument");
  QgSlabComm *m = c->sc_comm;
  const int r = m->rank, P = m->nranks;
  const size_t n = m->halo_len;
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  for (int which = 0; which < 3; ++which) {
    for (int pass = 0; pass < 2; ++pass) { // pass 0 warms the algorithm / channel set-up
      const int nrep = pass ? reps : 3;
      HIPCHECK(hipEventRecord(a, c->stream));
      for (int i = 0; i < nrep; ++i) {
        if (which == 0) {
          NCCLCHECK(m->api, m->api->AllGather(m->th_send, m->th_gath, m->th_len, ncclDouble, m->comm, c->stream));
        } else if (which == 1) {
          NCCLCHECK(m->api, m->api->AllGather(m->h_send, m->h_gath, 2 * n, ncclDouble, m->comm, c->stream));
        } else if (P > 1) {
          double *rl = m->h_gath + (size_t)(2 * (r > 0 ? r - 1 : 0) + 1) * n, *rh = m->h_gath + (size_t)(2 * (r < P - 1 ? r + 1 : 0)) * n;
          NCCLCHECK(m->api, m->api->GroupStart());
          if (r > 0) {
            NCCLCHECK(m->api, m->api->Send(m->h_send, n, ncclDouble, r - 1, m->comm, c->stream));
            NCCLCHECK(m->api, m->api->Recv(rl, n, ncclDouble, r - 1, m->comm, c->stream));
          }
          if (r < P - 1) {
            NCCLCHECK(m->api, m->api->Send(m->h_send + n, n, ncclDouble, r + 1, m->comm, c->stream));
            NCCLCHECK(m->api, m->api->Recv(rh, n, ncclDouble, r + 1, m->comm, c->stream));
          }
          NCCLCHECK(m->api, m->api->GroupEnd());
        }
      }
      HIPCHECK(hipEventRecord(b, c->stream));
      HIPCHECK(hipEventSynchronize(b));
      float ms = 0.f;
      HIPCHECK(hipEventElapsedTime(&ms, a, b));
      if (pass) us[which] = 1e3 * ms / nrep;
    }
  }
  hipEventDestroy(a);
  hipEventDestroy(b);
  return 0;
}

// one distributed ocean step: three communication-free stages (the same calls SlabOcean.step makes through
// qgcm_hip_slab_stage) and two exchanges, all ordered on c->stream; no host synchronisation
// next_follows: step s + 1 is executed by the same qgcm_hip_slab_steps call / captured block (its tendency may be started)
static int slab_step(qgcm_hip_ctx *c, int s, bool next_follows) {
  QgSlabComm *m = c->sc_comm;
  const int r = m->rank, P = m->nranks;
  const size_t n = m->halo_len;
  if (m->th_len != slab_msg_len_oml(c) || n != halo_msg_len(c))
    QG_FAIL("qgcm_hip_slab_steps: the mixed layer was switched on after qgcm_hip_comm_init (message sizes differ)");
  // 0. mixed layer (`call oml` precedes qgostep): new sst + raw entrainment, every slab's sums, entoc
  if (c->oml.on) {
    if (qgcm_hip_slab_stage(c, 10, m->oml_send, nullptr, nullptr, r, P, 0)) return 1;
    NCCLCHECK(m->api, m->api->AllGather(m->oml_send, m->oml_gath, 3, ncclDouble, m->comm, c->stream));
    if (qgcm_hip_slab_stage(c, 11, m->oml_gath, nullptr, nullptr, r, P, 0)) return 1;
  }
  // 1. tendency, forward row transform, slab summary of the two y sweeps
  if (m->pending) {
    // the halo rows of the previous step are still on their way (second stream): inner tile rows first
    if (qgcm_hip_slab_stage(c, 4, nullptr, nullptr, nullptr, r, P, 0)) return 1;
    HIPCHECK(hipStreamWaitEvent(c->stream, m->ev_halo, 0));
    m->pending = false;
    // the outer tile rows have run on the exchange's stream right behind the halo rows, beside the inner ones (below)
    if (qgcm_hip_slab_stage(c, m->outer_done ? 7 : 5, m->th_send, nullptr, nullptr, r, P, 0)) return 1;
    m->outer_done = false;
  } else if (qgcm_hip_slab_stage(c, 1, m->th_send, nullptr, nullptr, r, P, 0)) return 1;
  NCCLCHECK(m->api, m->api->AllGather(m->th_send, m->th_gath, m->th_len, ncclDouble, m->comm, c->stream));
  // 2. both sweeps from the composed inflows (+ basin-wide area integrals), constraints, inverse row transform,
  //    modes -> layers + boundary PV, edge rows out
  double *to_lo = r > 0 ? m->h_send : nullptr, *to_hi = r < P - 1 ? m->h_send + n : nullptr;
  if (qgcm_hip_slab_stage(c, 2, m->th_gath, to_lo, to_hi, r, P, 0)) return 1;
  const int avg = (s - 1) % 25 == 0 ? 1 : 0; // leapfrog averaging after steps s with (s-1) mod 25 == 0
  // overlapped: the exchange and the halo unpack on the second stream; the next step's stage 4 does not wait for them
  const bool ov = m->overlap && !avg && !c->oml.on && tend_can_split(c);
  hipStream_t xs = c->stream;
  if (ov) {
    HIPCHECK(hipEventRecord(m->ev_fork, c->stream));
    HIPCHECK(hipStreamWaitEvent(m->cstream, m->ev_fork, 0));
    xs = m->cstream;
  }
  const double *from_lo = nullptr, *from_hi = nullptr;
  if (P > 1 || ov) { // (one rank, overlapped: the all-gather is a local copy - it keeps the one-GPU tests on this path)
    if (m->halo_p2p && P > 1) {
      // neighbours only; receive areas are the neighbour's slots of h_gath, as with the all-gather
      double *rl = m->h_gath + (size_t)(2 * (r > 0 ? r - 1 : 0) + 1) * n, *rh = m->h_gath + (size_t)(2 * (r < P - 1 ? r + 1 : 0)) * n;
      NCCLCHECK(m->api, m->api->GroupStart());
      if (r > 0) {
        NCCLCHECK(m->api, m->api->Send(to_lo, n, ncclDouble, r - 1, m->comm, xs));
        NCCLCHECK(m->api, m->api->Recv(rl, n, ncclDouble, r - 1, m->comm, xs));
      }
      if (r < P - 1) {
        NCCLCHECK(m->api, m->api->Send(to_hi, n, ncclDouble, r + 1, m->comm, xs));
        NCCLCHECK(m->api, m->api->Recv(rh, n, ncclDouble, r + 1, m->comm, xs));
      }
      NCCLCHECK(m->api, m->api->GroupEnd());
    } else {
      NCCLCHECK(m->api, m->api->AllGather(m->h_send, m->h_gath, 2 * n, ncclDouble, m->comm, xs));
    }
    if (r > 0) from_lo = m->h_gath + (size_t)(2 * (r - 1) + 1) * n;  // what the lower neighbour sent upwards
    if (r < P - 1) from_hi = m->h_gath + (size_t)(2 * (r + 1)) * n;  // what the upper neighbour sent downwards
  }
  // 3. edge rows in (+ leapfrog averaging)
  if (P > 1 || avg) {
    hipStream_t main_stream = c->stream;
    c->stream = xs; // (the stage's launches follow the exchange on its stream)
    const int rc = qgcm_hip_slab_stage(c, 3, (double *)from_lo, (double *)from_hi, nullptr, r, P, avg);
    c->stream = main_stream;
    if (rc) return 1;
  }
  if (ov) {
    // The outer tile rows of step s + 1's tendency launch go out on the exchange's stream, right behind the halo rows
    // they need: they run BESIDE the inner tile rows the handle's stream is computing (disjoint tiles of the same
    // launch) instead of as a launch of their own after them - which cost a whole extra single-generation kernel per
    // step (round 3: +23 us on one rank).  Only when step s + 1 belongs to this call: a caller that stops here may
    // read the state, and the outer rows write into the buffer that holds qom.
    m->outer_done = false;
    if (next_follows) {
      hipStream_t main_stream = c->stream;
      c->stream = xs;
      const int rc = qgcm_hip_slab_stage(c, 6, m->th_send, nullptr, nullptr, r, P, 0);
      c->stream = main_stream;
      if (rc) return 1;
      m->outer_done = true;
    }
    HIPCHECK(hipEventRecord(m->ev_halo, m->cstream));
    m->pending = true;
  }
  return 0;
}

// the second stream joins the handle's stream (end of a qgcm_hip_slab_steps call, end of a captured block)
static int slab_join(qgcm_hip_ctx *c) {
  QgSlabComm *m = c->sc_comm;
  if (m && m->pending) {
    HIPCHECK(hipStreamWaitEvent(c->stream, m->ev_halo, 0));
    m->pending = false;
  }
  return 0;
}

static int get_slab_graph(qgcm_hip_ctx *c, int s0, hipGraphExec_t *out) {
  const int phase = (s0 - 1) % 25;
  const int omk = c->oml.on ? 1 + 3 * c->oml.is + c->oml.ism : 0; // mixed layer on/off and its buffer rotation
  const int key = (omk << 20) | (c->ip << 16) | (c->iq << 8) | phase;
  auto it = c->slab_graphs.find(key);
  if (it != c->slab_graphs.end()) {
    *out = it->second;
    return 0;
  }
  const int ip0 = c->ip, iq0 = c->iq, is0 = c->oml.is, ism0 = c->oml.ism;
  hipGraph_t graph;
  HIPCHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int k = 0; k < kGraphBlock && !rc; ++k) rc = slab_step(c, s0 + k, k + 1 < kGraphBlock);
  if (!rc) rc = slab_join(c);
  hipError_t e = hipStreamEndCapture(c->stream, &graph);
  if (rc && c->sc_comm) c->sc_comm->pending = c->sc_comm->outer_done = false;
  c->ip = ip0; // nothing ran: the rotation state is that of the block's first step
  c->iq = iq0;
  c->oml.is = is0;
  c->oml.ism = ism0;
  if (rc) return 1;
  HIPCHECK(e);
  hipGraphExec_t exec;
  HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIPCHECK(hipGraphDestroy(graph));
  c->slab_graphs[key] = exec;
  *out = exec;
  return 0;
}

extern "C" int qgcm_hip_slab_steps(qgcm_hip_handle c, int s0, int n) {
  if (check_ready(c, "qgcm_hip_slab_steps")) return 1;
  if (!c->sc_comm) QG_FAIL("qgcm_hip_slab_steps: no communicator (qgcm_hip_comm_init)");
  if (!c->homog_set) QG_FAIL("qgcm_hip_slab_steps: homogeneous solutions not set");
  if (s0 < 1 || n < 0) QG_FAIL("qgcm_hip_slab_steps: bad step range");
  int s = s0;
  while (c->slab_graph_mode && n >= kGraphBlock) {
    hipGraphExec_t ge;
    if (get_slab_graph(c, s, &ge)) return 1;
    HIPCHECK(hipGraphLaunch(ge, c->stream));
    if (c->oml.on) oml_rotate(c, kGraphBlock); // the p and q rotations are back where they started, sst has moved on
    s += kGraphBlock; // 50 steps: both rotations are back where they started
    n -= kGraphBlock;
  }
  for (; n > 0; --n, ++s)
    if (slab_step(c, s, n > 1)) return 1;
  return slab_join(c);
}

// Instantiate (and upload) the graphs qgcm_hip_steps(s0, n) will replay, without running anything: a caller that
// times a window with its own clock calls this first, so that capture + instantiation stay outside the window.
extern "C" int qgcm_hip_prepare_steps(qgcm_hip_handle c, int s0, int n) {
  if (check_ready(c, "qgcm_hip_prepare_steps")) return 1;
  if (!c->whole) QG_FAIL("qgcm_hip_prepare_steps: this handle is a y-slab");
  if (s0 < 1 || n < 0) QG_FAIL("qgcm_hip_prepare_steps: bad step range");
  if (steps_impl(c, s0, n, true)) return 1;
  HIPCHECK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int qgcm_hip_time_steps(qgcm_hip_handle c, int s0, int n, float *ms) {
  if (check_ready(c, "qgcm_hip_time_steps")) return 1;
  hipEvent_t a, b;
  HIPCHECK(hipEventCreate(&a));
  HIPCHECK(hipEventCreate(&b));
  // instantiate the graphs the run replays outside the timed region
  if (steps_impl(c, s0, n, true)) return 1;
  HIPCHECK(hipEventRecord(a, c->stream));
  if (qgcm_hip_steps(c, s0, n)) return 1;
  HIPCHECK(hipEventRecord(b, c->stream));
  // (polling, not hipEventSynchronize: a caller that times the call with its own clock should not pay the wake-up
  //  latency of a blocking wait - tens of microseconds on a window of 20 steps)
  for (;;) {
    const hipError_t q = hipEventQuery(b);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) HIPCHECK(q);
  }
  (void)hipGetLastError(); // (hipErrorNotReady of the polls is not an error to report later)
  float t = 0.f;
  HIPCHECK(hipEventElapsedTime(&t, a, b));
  if (ms) *ms = t;
  hipEventDestroy(a);
  hipEventDestroy(b);
  return 0;
}

extern "C" int qgcm_hip_profile_steps(qgcm_hip_handle c, int s0, int n, double *ms, int *launches, const char **names, int *nk) {
  if (check_ready(c, "qgcm_hip_profile_steps")) return 1;
  if (!nk || *nk < KN_COUNT) QG_FAIL("qgcm_hip_profile_steps: need room for %d kernels", KN_COUNT);
  for (int i = 0; i < KN_COUNT; ++i) {
    c->kms[i] = 0.0;
    c->klaunch[i] = 0;
  }
  HIPCHECK(hipStreamSynchronize(c->stream));
  c->profiling = true;
  c->timer_err = hipSuccess;
  // Calibration of the brackets. An event record is a packet of its own on the stream, so a
  // bracket reads (kernel + its two records). "k_noop" is an empty launch bracketed like the
  // kernels; "k_noop_train" is a long train of empty launches inside ONE bracket, i.e. what an
  // empty launch costs on the stream by itself. Their difference is the cost of a bracket.
  const int ntrain = 512;
  {
    KTimer t(c, KN_NOOP_TRAIN);
    for (int i = 0; i < ntrain; ++i) hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, c->stream);
  }
  // all brackets of the n steps are queued without a host synchronisation in between (the
  // stream stays busy, as in the timed region) and read at the end
  int rc = qgcm_hip_steps(c, s0, n);
  drain_timers(c);
  c->profiling = false;
  if (rc) return 1;
  if (c->timer_err != hipSuccess) QG_FAIL("qgcm_hip_profile_steps: event record failed: %s", hipGetErrorName(c->timer_err));
  c->klaunch[KN_NOOP_TRAIN] = ntrain;
  for (int i = 0; i < KN_COUNT; ++i) {
    if (ms) ms[i] = c->kms[i];
    if (launches) launches[i] = c->klaunch[i];
    if (names) names[i] = kKernelNames[i];
  }
  *nk = KN_COUNT;
  return 0;
}

extern "C" int qgcm_hip_copy_bandwidth(qgcm_hip_handle c, size_t bytes, int reps, double *gbps) {
  if (!c || !gbps) QG_FAIL("qgcm_hip_copy_bandwidth: null argument");
  double2 *a = nullptr, *b = nullptr;
  const long n = (long)(bytes / sizeof(double2));
  HIPCHECK(hipMalloc((void **)&a, n * sizeof(double2)));
  HIPCHECK(hipMalloc((void **)&b, n * sizeof(double2)));
  HIPCHECK(hipMemset(a, 1, n * sizeof(double2)));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, c->stream, a, b, n);
  HIPCHECK(hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, c->stream, a, b, n);
  HIPCHECK(hipEventRecord(e1, c->stream));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  *gbps = 2.0 * (double)n * sizeof(double2) * reps / (ms * 1e-3) / 1e9;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(a);
  hipFree(b);
  return 0;
}

// Rate (GB/s of read + written bytes) of a pure streaming kernel that reads nr fields and writes nw fields of
// field_bytes each (the mixes of the hot kernels: 15:6 tendency, 3:3 rows / sweep, 5:3 fused inverse rows).
extern "C" int qgcm_hip_stream_mix_bandwidth(qgcm_hip_handle c, int nr, int nw, size_t field_bytes, int reps, double *gbps) {
  if (!c || !gbps) QG_FAIL("qgcm_hip_stream_mix_bandwidth: null argument");
  field_bytes = field_bytes / 4096 * 4096;
  if (field_bytes == 0 || reps < 1) QG_FAIL("qgcm_hip_stream_mix_bandwidth: bad size / repetitions");
  double2 *a = nullptr, *b = nullptr;
  const long nper = (long)(field_bytes / sizeof(double2));
  HIPCHECK(hipMalloc((void **)&a, field_bytes * nr));
  HIPCHECK(hipMalloc((void **)&b, field_bytes * (nw > 0 ? nw : 1)));
  HIPCHECK(hipMemsetAsync(a, 0, field_bytes * nr, c->stream));
  HIPCHECK(hipMemsetAsync(b, 0, field_bytes * (nw > 0 ? nw : 1), c->stream));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  auto launch = [&]() -> int {
    const dim3 grid(2048), block(256);
    if (nr == 15 && nw == 6) hipLaunchKernelGGL((k_stream_mix<15, 6>), grid, block, 0, c->stream, a, b, nper);
    else if (nr == 3 && nw == 3) hipLaunchKernelGGL((k_stream_mix<3, 3>), grid, block, 0, c->stream, a, b, nper);
    else if (nr == 5 && nw == 3) hipLaunchKernelGGL((k_stream_mix<5, 3>), grid, block, 0, c->stream, a, b, nper);
    else return 1;
    return 0;
  };
  int rc = 0;
  for (int w = 0; w < 3 && !rc; ++w) rc = launch();
  if (!rc) {
    HIPCHECK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) launch();
    HIPCHECK(hipEventRecord(e1, c->stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    *gbps = (double)field_bytes * (nr + nw) * reps / (ms * 1e-3) / 1e9;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(a);
  hipFree(b);
  if (rc) QG_FAIL("qgcm_hip_stream_mix_bandwidth: supported mixes are 15:6, 3:3 and 5:3");
  return 0;
}

extern "C" void *qgcm_hip_stream(qgcm_hip_handle c) { return c ? (void *)c->stream : nullptr; }

#ifdef QG_STAMPS
// development builds only (scratch/stamps.py): the phase stamps of the last launch of each instrumented kernel
extern "C" int qgcm_hip_debug_stamps(long long *out, size_t nbytes) {
  if (nbytes > sizeof(qg_stamps)) nbytes = sizeof(qg_stamps);
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(qg_stamps), nbytes, 0, hipMemcpyDeviceToHost));
  return 0;
}
#endif
