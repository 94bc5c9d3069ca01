// k_tend.h - K1: PV tendency + leapfrog + layer->mode projection, one pass.
//
// Replaces, per ocean step (reference file:line):
//   del2p = Del^2(pom) with mixed BCs            src/qgosubs.F:86-130
//   ocadif: d4p, d6p, Arakawa 9-pt J(q,p), dqdt  src/qgosubs.F:306-400
//   forcing / bottom drag / leapfrog             src/qgosubs.F:173-219
//   ocinvq projection onto modes                 src/ocisubs.F:117-139
//
// One workgroup owns a TX x TY tile of p-points for ALL layers.  Per layer
// the radius-3 halo of pom and the radius-1 halos of po, qo are staged in
// LDS, Del^2 / Del^4 are built in LDS with the reference's boundary rules,
// and each thread finishes Del^6 + Jacobian for its points in registers.
// The global loads of layer k+1 (and of the point-wise epilogue operands) are
// issued into registers before layer k is computed, so a workgroup always
// has loads in flight (software pipelining across layers).
// Workgroups are renumbered so that each XCD (blockIdx mod 8) sweeps a
// contiguous band of tile rows: the halo rows re-read by vertically adjacent
// tiles then hit that XCD's own L2 instead of being re-fetched.
// Tile shape: 16 x 16 points, 256 threads, one point per thread (72 VGPRs, 12 KB LDS: 7 workgroups per CU). The
// kernel is bound by its chain of barrier-separated phases, so resident workgroups matter more than halo
// re-reads: measured at 5 km 64x8 (2 points per thread, 5 per CU) 38.4 us, 32x8 35.0, 16x16 33.3, 16x8 35.1,
// 32x16/512 threads 35.3, 16x32/512 36.2.
// Expression association order is the reference's, and the library is built
// with -ffp-contract=off, so qgostep reproduces the CPU reference bit for bit.
//
// Algorithmic HBM traffic: read pom,po,qo,qom (4 nl) + wekpo,entoc,ddynoc (3),
// write qo (nl) + wrk (nl)  ->  (6 nl + 3) N doubles = 21 N for nl = 3
// (SURVEY 8d counts 24 N because the reference also rewrites qom; here the
// q buffers rotate instead).
#pragma once
#include "qgcm_dev.h"
#include "k_misc.h" // constr_dpi_update
#include "k_cyclic.h" // cyc_bsums_block
#include "k_oml.h" // oml_final_block

#ifndef TEND_TX
#define TEND_TX 16
#endif
#ifndef TEND_TY
#define TEND_TY 16
#endif
#ifndef TEND_NT
#define TEND_NT 256
#endif

// Tiles of the HBM-bound sizes (the instantiations without write-through stores, WTQ = false): twice as wide.  A row of a
// 32-wide tile is two whole 128-byte lines per field instead of one, the radius-3 halo columns weigh half as much, and
// at SOcn 5 km / the slabs of NAtl 1 km the kernel's reads come from HBM, not from the Infinity Cache: measured at
// SOcn 5 km (profiles/r4_tend_shapes_socn5.log) 16 x 16: 100.6 us, 32 x 16 (256 threads, two rows per thread): 94.7,
// 32 x 16 / 512 threads: 95.0, 64 x 8: 94.7, 64 x 4: 96.2, 32 x 8: 96.3, 16 x 32: 100.4 (+), 32 x 32: 101.2.  At NAtl 5 km
// (cache resident, write-through pairs) the 16 x 16 tile stays: resident workgroups matter more there (see above).
#ifndef TEND_TX_WIDE
#define TEND_TX_WIDE 32
#endif
static_assert(TEND_TX % 2 == 0 && TEND_TX_WIDE % 2 == 0, "tend_point<.., PAIR> stores the new qo in pairs of neighbouring columns: tiles start on odd columns");

template <bool CYC>
__device__ __forceinline__ int tend_wrap(int gi, int nxt) {
  if (CYC) {
    if (gi < 1) gi += nxt;
    else if (gi > nxt) gi -= nxt;
  }
  return gi;
}

// tiles cover i = 1..imax, local rows jlo..jmax; what is left (E wall column, N wall row) is edge work
struct TendTiling {
  int gx, gy, imax, jmax;
  int ecol, erow; // 1 if the E wall column / N wall row is peeled off
  int nedge;      // edge workgroups
};

template <bool CYC, int TX = TEND_TX>
__host__ __device__ __forceinline__ TendTiling tend_tiling(const QgGeom &g) {
  TendTiling T;
  const int rows = g.jhi - g.jlo + 1;
  T.ecol = (!CYC && g.nx % TX == 1 && g.nx > 1) ? 1 : 0;
  T.erow = (!CYC && rows % TEND_TY == 1 && rows > 1 && g.jhi + g.joff == g.nyg) ? 1 : 0;
  T.gx = T.ecol ? g.nx / TX : (g.nx + TX - 1) / TX;
  T.gy = T.erow ? rows / TEND_TY : (rows + TEND_TY - 1) / TEND_TY;
  T.imax = T.ecol ? g.nx - 1 : g.nx;
  T.jmax = T.erow ? g.jhi - 1 : g.jhi;
  const int npts = T.ecol * rows + T.erow * T.imax;
  T.nedge = (npts + TEND_NT - 1) / TEND_NT;
  return T;
}

// Point-wise part of the step for one p-point and all layers (qgosubs.F:173-219, ocisubs.F:117-139):
// dq = dqdt of the point (0 outside the interior), d2bot = Del^2(pom) of the bottom layer,
// qm / qo = old qom / qo of the point (qo: only read on rows that are not stepped, and with AVG).
// PAIR (the tile kernel's epilogue: EVERY lane of the wave calls, `valid` says whether its point exists): the new qo
// and the work array leave as 16-byte write-through stores of two neighbouring columns (qgcm_dev.h: the 44 MB this
// kernel writes no longer wait, dirty in L2, for the end-of-kernel flush); !PAIR: plain stores (edge workgroups).
// AVG: the leapfrog averaging that follows this step (src/q-gcm.F:1345-1351) is folded into the store of the new qo:
// 0.5*(new qo + qo); the projection keeps the un-averaged value.  A template flag: as a kernel argument the extra
// branches cost the other 24 steps of 25 more than the averaging pass had (measured: +0.5 us per step at 5 km).
template <int NL, bool CYC, bool PAIR = false, bool AVG = false>
__device__ __forceinline__ void tend_point(const QgTendParams &P, int gi, int gj, const double *dq, double d2bot,
                                           const double *qm, const double *qo, double wek, double ent, double ddy,
                                           bool valid = true) {
  const long fs = P.g.fstride;
  const long o = (long)(gj - 1) * P.g.ldx + (gi - 1);
  const bool wallrow = (gj + P.g.joff == 1 || gj + P.g.joff == P.g.nyg);
  if (!PAIR && wallrow) {
    // rows not stepped: the new-qo buffer keeps qo (qgosubs.F:214-219)
#pragma unroll
    for (int k = 0; k < NL; ++k) P.qnew[fs * k + o] = qo[k];
    return;
  }
  double qdot[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) qdot[k] = dq[k];
  if (CYC && P.g.atm) {
    // atmosphere (src/qgasubs.F:128-131): entrainment and Ekman pumping act on the bottom layer 1 with the
    // opposite sign, there is no drag term
    qdot[0] = dq[0] + P.fohfac[0] * (ent - wek);
    qdot[1] = dq[1] - P.fohfac[1] * ent;
  } else {
    qdot[0] = dq[0] + P.fohfac[0] * (wek - ent);
    qdot[1] = dq[1] + P.fohfac[1] * ent;
    qdot[NL - 1] = qdot[NL - 1] - P.bdrfac * d2bot;
  }
  double ql[NL];
  double betay = P.beta * P.yporel[gj - 1];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    double qn = qm[k] + P.tdto * qdot[k]; // qom + tdto*qdot
    // sponge layer of the k247 fork (src/qgosubs.F:203-205), association as written there; a wave-uniform branch on a
    // kernel argument, not taken in any BASELINE configuration
    if (P.rspl) qn = qn + P.tdc1 * P.rspl[valid ? o : 0] * (qm[k] - betay);
    const double qs = AVG ? 0.5 * (qn + qo[k]) : qn;
    if (PAIR) {
      // (rows not stepped keep qo, qgosubs.F:214-219; o is even on even lanes: the tile starts at an odd column)
      qg_pair_store_wt(P.qnew + fs * k + o, wallrow ? qo[k] : qs, valid);
    } else {
      P.qnew[fs * k + o] = qs;
    }
    ql[k] = qn - betay;
  }
  if (CYC && P.g.atm) ql[0] = ql[0] - ddy; // topography under layer 1, src/atisubs.F:117
  else ql[NL - 1] = ql[NL - 1] - ddy;
  int c = CYC ? gi - 1 : gi - 2;
  const bool cok = c >= 0 && c < P.g.nk && !wallrow;
  if (PAIR || cok) {
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      double qmm = 0.0;
#pragma unroll
      for (int k = 0; k < NL; ++k) qmm = qmm + P.ctl2m[k + NL * m] * ql[k];
      // (plain stores: the box ocean's column c = gi - 2 is even on ODD lanes, and pairs that start on odd lanes leave
      //  lanes 0 and 15 of every tile row with single elements.  Measured: those as 8-byte plain stores into lines the
      //  16-byte write-through stores keep dropping from L2: kernel 29.6 -> 82 us; as 8-byte sc1 stores: 33.3 us.)
      if (valid && cok) P.wrk[P.g.wstride * m + (long)(gj - 1) * P.g.ldw + c] = P.fnot * qmm;
    }
  }
}

// Edge work of the box ocean: the E wall column (dqdt = 0, qgosubs.F:371,397; Del^2 of the bottom layer by
// the wall rule :112,125) and the N wall row (not stepped), one point per thread.
template <int NL, bool AVG = false>
__device__ __forceinline__ void tend_edge(const QgTendParams &P, const TendTiling &T, int eblock) {
  const QgGeom &g = P.g;
  const int rows = g.jhi - g.jlo + 1;
  const int t = eblock * TEND_NT + (int)threadIdx.x;
  int gi, gj;
  if (t < T.ecol * rows) {
    gi = g.nx;
    gj = g.jlo + t;
  } else {
    const int u = t - T.ecol * rows;
    if (!T.erow || u >= T.imax) return;
    gi = 1 + u;
    gj = g.jhi;
  }
  const long fs = g.fstride;
  const long o = (long)(gj - 1) * g.ldx + (gi - 1);
  const bool wallrow = (gj + g.joff == 1 || gj + g.joff == g.nyg);
  double dq[NL], qm[NL], qo[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    dq[k] = 0.0;
    qm[k] = wallrow ? 0.0 : P.qnew[fs * k + o];
    qo[k] = (wallrow || AVG) ? P.qo[fs * k + o] : 0.0; // (AVG: tend_point stores 0.5*(new qo + qo))
  }
  double d2bot = 0.0, wek = 0.0, ent = 0.0, ddy = 0.0;
  if (!wallrow) { // then gi == nx: E wall column of a stepped row
    const double *pb = P.pom + fs * (NL - 1) + o;
    d2bot = P.bcfaco * (pb[-1] - pb[0]);
    wek = P.wekpo[o];
    ent = P.entoc[o];
    ddy = P.ddynoc[o];
  }
  tend_point<NL, false, false, AVG>(P, gi, gj, dq, d2bot, qm, qo, wek, ent, ddy);
}

// WTQ: the epilogue stores the new qo in write-through pairs (tend_point<.., PAIR>) - worth 1 us per step while the
// step's working set stays in the Infinity Cache (NAtl 5 km), but it costs 6 VGPRs (78: six waves per SIMD instead of
// seven) and at HBM-bound sizes, where L2's own full-line evictions already spread the writes over the kernel, it made
// the kernel slower (SOcn 5 km 95 -> 103 us, written through or not): the host picks the instantiation by size.
template <int NL, bool CYC, bool WTQ, bool AVG = false>
#ifndef TEND_WAVES_PER_EU
#define TEND_WAVES_PER_EU 4
#endif
#ifndef TEND_WAVES_WTQ
#define TEND_WAVES_WTQ 4
#endif
__global__ __launch_bounds__(TEND_NT, (WTQ && NL <= 4) ? TEND_WAVES_WTQ : TEND_WAVES_PER_EU) void k_tend(const QgTendParams P, const QgCycSumParams S,
                                                                     const QgOmlFinal F) {
#ifndef TEND_EXPERIMENT // (tile-shape A/B builds without the mixed layer: profiles/r4_tend_shapes_socn5.log)
  static_assert(TEND_NT == OML_NT, "oml_final_block runs in workgroup 0 of this kernel");
#endif
  constexpr int TX = WTQ ? TEND_TX : TEND_TX_WIDE, TY = TEND_TY;
  static_assert(TEND_NT % TX == 0 && TY % (TEND_NT / TX) == 0, "whole rows of threads, whole rows per thread");
  constexpr int W3 = TX + 6, H3 = TY + 6; // pom tile, halo 3
  constexpr int W2 = TX + 4, H2 = TY + 4; // d2 tile, halo 2
  constexpr int W1 = TX + 2, H1 = TY + 2; // d4 / po / qo tiles, halo 1
  constexpr int N3 = (H3 * W3 + TEND_NT - 1) / TEND_NT; // staged elements per thread
  constexpr int N1 = (H1 * W1 + TEND_NT - 1) / TEND_NT;
  __shared__ double sp[H3 * W3];
  __shared__ double sd2[H2 * W2];
  // Del^4 tile: lives in the pom tile's storage (dead once Del^2 is formed; refilled only after the layer's
  // last barrier), which keeps the workgroup at 4 x 30 KB (TY = 8) resp. 3 x 51 KB (TY = 20) per CU
  double *sd4 = sp;
  static_assert(H1 * W1 <= H3 * W3, "Del^4 tile must fit into the pom tile");
  __shared__ double spo[H1 * W1];
  __shared__ double sqo[H1 * W1];

  const int nx = P.g.nx, ny = P.g.ny, nxt = P.g.nxt, ldx = P.g.ldx;
  const int nyg = P.g.nyg, joff = P.g.joff, jlo = P.g.jlo, jhi = P.g.jhi; // slab view (global row = local + joff)
  const long fs = P.g.fstride;
  const int tid = threadIdx.x;
  // ---- XCD-aware tile numbering ------------------------------------------
  // Box grids have nx = 64*g + 1 columns and (whole basin) 8*h + 1 rows: the last column and the last
  // row are walls whose update is point-wise (no stencil), so they are peeled off into a few "edge"
  // workgroups instead of a whole extra column / row of nearly empty tiles (tend_edge below).
  const TendTiling T = tend_tiling<CYC, TX>(P.g);
  const int gx = T.gx;
  const int ntiles = gx * (P.trows ? P.trows : T.gy); // (a window of the tile rows, or all of them)
  const int per_xcd = (ntiles + 7) / 8;
  if (F.on && blockIdx.x == 0) {
    // mixed layer on, inside qgcm_hip_steps: the last reduction of `oml` (xon(1), enisoc(1) / eninoc(1), monitors) is
    // done here instead of in a one-workgroup launch; xon must be final before dpioc is stepped below
    __shared__ double redf[20];
    oml_final_block(F, redf, tid);
  }
  if (!CYC && P.upd_dpi && blockIdx.x == 0 && tid == 0) constr_dpi_update<NL>(P.sc, P.tdto, P.gpoc); // see QgTendParams
  if ((int)blockIdx.x >= 8 * per_xcd) {
    // cyclic / atmosphere: the boundary line sums for the momentum constraints (k_cyclic.h); box: the wall edges
    if (CYC) cyc_bsums_block(S, (int)blockIdx.x - 8 * per_xcd);
    else tend_edge<NL, AVG>(P, T, (int)blockIdx.x - 8 * per_xcd);
    return;
  }
  const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  QG_STAMP(3, 0);
  const int i0 = (tile % gx) * TX + 1; // first global i of the tile (1-based)
  const int trow = P.trows ? P.trow0 + (tile / gx) * P.tstride : tile / gx;
  const int j0 = trow * TY + jlo; // first local row of the tile
  const int tx = tid % TX;
  const int ty0 = tid / TX; // 0..3
  constexpr int RPT = TY / (TEND_NT / TX); // rows per thread
  const double bcf = P.bcfaco, dxom2 = P.dxom2;

  // ---- global offsets of the elements this thread stages (same for every layer), computed ONCE: 32-bit element
  // offsets (the host checks ldx*ny*nl < 2^31) clamped to 0 where the element lies outside the array, plus a
  // validity bit. Every staging load is then unconditional: a "load or zero" conditional makes hipcc branch around
  // each load (exec-mask save / restore, s_cbranch_execz: ~560 of the kernel's 1900 instructions were that
  // bookkeeping, and the kernel is instruction-issue bound). Elements outside the array receive element 0 of the
  // field - a finite value that no point inside the domain ever reads.
  int o3[N3], o1[N1];
  bool v3[N3], v1[N1];
#pragma unroll
  for (int e = 0; e < N3; ++e) {
    int idx = tid + e * TEND_NT;
    int lx = idx % W3, ly = idx / W3;
    int gi = i0 - 3 + lx, gj = j0 - 3 + ly;
    v3[e] = idx < H3 * W3 && gj >= 1 && gj <= ny && (CYC ? (gi >= -2 && gi <= nx + 3) : (gi >= 1 && gi <= nx));
    o3[e] = v3[e] ? (gj - 1) * ldx + (tend_wrap<CYC>(gi, nxt) - 1) : 0;
  }
#pragma unroll
  for (int e = 0; e < N1; ++e) {
    int idx = tid + e * TEND_NT;
    int lx = idx % W1, ly = idx / W1;
    int gi = i0 - 1 + lx, gj = j0 - 1 + ly;
    v1[e] = idx < H1 * W1 && gj >= 1 && gj <= ny && (CYC ? (gi >= 0 && gi <= nx + 1) : (gi >= 1 && gi <= nx));
    o1[e] = v1[e] ? (gj - 1) * ldx + (tend_wrap<CYC>(gi, nxt) - 1) : 0;
  }
  double r3[N3], rp[N1], rq[N1];
#pragma unroll
  for (int e = 0; e < N3; ++e) r3[e] = P.pom[o3[e]];
#pragma unroll
  for (int e = 0; e < N1; ++e) {
    rp[e] = P.po[o1[e]];
    rq[e] = P.qo[o1[e]];
  }
  // ---- the Del^2 / Del^4 passes of this thread (elements tid, tid + 256, ... of the halo-2 / halo-1 region): the LDS
  // read base, the position of the ONE inner neighbour the wall rule uses (N of a S-wall point, ...) and whether the
  // point is a wall point do not depend on the layer - computed once.  Points outside the domain are computed from
  // whatever the (clamped) loads delivered: no point inside the domain ever reads them (interior points have all five
  // neighbours inside, wall points read their inner neighbour only), so they need neither a predicate nor a select.
  constexpr int N2 = (H2 * W2 + TEND_NT - 1) / TEND_NT;
  int b2[N2], b4[N1], i2[N2], i4[N1];
  bool w2[N2], w4[N1];
#pragma unroll
  for (int e = 0; e < N2; ++e) {
    const int idx = tid + e * TEND_NT;
    const int lx = idx % W2, ly = idx / W2;
    const int gi = i0 - 2 + lx, gj = j0 - 2 + ly;
    const bool inx = CYC ? (gi >= -1 && gi <= nx + 2) : (gi >= 1 && gi <= nx);
    (void)inx;
    b2[e] = (idx < H2 * W2) ? (ly + 1) * W3 + (lx + 1) : W3 + 1; // clamped: the extra pass reads valid LDS, stores nothing
    const bool wS = (gj + joff == 1), wN = (gj + joff == nyg), wW = (!CYC && gi == 1), wE = (!CYC && gi == nx);
    w2[e] = wS || wN || wW || wE;
    i2[e] = b2[e] + (wS ? W3 : (wN ? -W3 : (wW ? 1 : -1)));
  }
#pragma unroll
  for (int e = 0; e < N1; ++e) {
    const int idx = tid + e * TEND_NT;
    const int lx = idx % W1, ly = idx / W1;
    const int gi = i0 - 1 + lx, gj = j0 - 1 + ly;
    const bool inx = CYC ? (gi >= 0 && gi <= nx + 1) : (gi >= 1 && gi <= nx);
    (void)inx;
    b4[e] = (idx < H1 * W1) ? (ly + 1) * W2 + (lx + 1) : W2 + 1;
    const bool wS = (gj + joff == 1), wN = (gj + joff == nyg), wW = (!CYC && gi == 1), wE = (!CYC && gi == nx);
    w4[e] = wS || wN || wW || wE;
    i4[e] = b4[e] + (wS ? W2 : (wN ? -W2 : (wW ? 1 : -1)));
  }
  // ---- epilogue operands of this thread's own points: requested while the LAST layer is computed (the
  // registers of the layer prefetch are free by then); the old qo of the wall rows is read in the epilogue
  double e_qm[NL][RPT], e_wek[RPT], e_ent[RPT], e_ddy[RPT];
  double e_qo[AVG ? NL : 1][RPT]; // AVG: this step's qo of the thread's own points

  double dq[NL][RPT];
  double d2bot[RPT];

#pragma unroll
  for (int k = 0; k < NL; ++k) {
    // ---- registers -> LDS tiles of layer k --------------------------------
#pragma unroll
    for (int e = 0; e < N3; ++e) {
      int idx = tid + e * TEND_NT;
      if (idx < H3 * W3) sp[idx] = r3[e];
    }
#pragma unroll
    for (int e = 0; e < N1; ++e) {
      int idx = tid + e * TEND_NT;
      if (idx < H1 * W1) {
        spo[idx] = rp[e];
        sqo[idx] = rq[e];
      }
    }
    __syncthreads();
    QG_STAMP(3, 1 + 2 * k);
    // ---- issue the loads of layer k+1 (they fly while layer k is computed)
    if (k + 1 < NL) {
      const double *pom = P.pom + fs * (k + 1);
      const double *po = P.po + fs * (k + 1);
      const double *qo = P.qo + fs * (k + 1);
#pragma unroll
      for (int e = 0; e < N3; ++e) r3[e] = pom[o3[e]];
#pragma unroll
      for (int e = 0; e < N1; ++e) {
        rp[e] = po[o1[e]];
        rq[e] = qo[o1[e]];
      }
    } else {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        int ly = ty0 + r * (TEND_NT / TX);
        int gi = i0 + tx, gj = j0 + ly;
        bool in = gi <= T.imax && gj <= T.jmax;
        const int o = in ? (gj - 1) * ldx + (gi - 1) : 0; // clamped: unconditional loads, values unused when !in
        e_wek[r] = P.wekpo[o];
        e_ent[r] = P.entoc[o];
        e_ddy[r] = P.ddynoc[o];
#pragma unroll
        for (int kk = 0; kk < NL; ++kk) e_qm[kk][r] = P.qnew[fs * kk + o];
        if (AVG) {
#pragma unroll
          for (int kk = 0; kk < NL; ++kk) e_qo[kk][r] = P.qo[fs * kk + o];
        }
      }
    }
    // ---- Del^2(pom) on the halo-2 region (qgosubs.F:94-127) ----------
#pragma unroll
    for (int e = 0; e < N2; ++e) {
      const int idx = tid + e * TEND_NT;
      const double *c = &sp[b2[e]];
      // branch-free: the wall rule picks ONE inner neighbour (N, S, E or W of a wall point), the interior rule is
      // evaluated alongside and the result selected - same expressions, same rounding, no exec-mask bookkeeping
      const double cS = c[-W3], cW = c[-1], c0 = c[0], cE = c[1], cN = c[W3];
      const double vw = bcf * (sp[i2[e]] - c0);
      const double vi = (cS + cW + cE + cN - 4.0 * c0) * dxom2;
      if (idx < H2 * W2) sd2[idx] = w2[e] ? vw : vi;
    }
    __syncthreads();
    // ---- Del^4 on the halo-1 region (qgosubs.F:310-341) ---------------
#pragma unroll
    for (int e = 0; e < N1; ++e) {
      const int idx = tid + e * TEND_NT;
      const double *c = &sd2[b4[e]];
      const double cS = c[-W2], cW = c[-1], c0 = c[0], cE = c[1], cN = c[W2];
      const double vw = bcf * (sd2[i4[e]] - c0);
      const double vi = dxom2 * (cS + cW + cE + cN - 4.0 * c0);
      if (idx < H1 * W1) sd4[idx] = w4[e] ? vw : vi;
    }
    __syncthreads();
    // ---- Del^6 + Jacobian at the tile's own points (qgosubs.F:349-399)
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      int ly = ty0 + r * (TEND_NT / TX);
      int gi = i0 + tx, gj = j0 + ly;
      double val = 0.0;
      bool interior = (gj <= jhi && gj + joff >= 2 && gj + joff <= nyg - 1) && (CYC ? (gi >= 1 && gi <= nx) : (gi >= 2 && gi <= nx - 1));
      if (interior) {
        const double *d4 = &sd4[(ly + 1) * W1 + (tx + 1)];
        const double *p = &spo[(ly + 1) * W1 + (tx + 1)];
        const double *q = &sqo[(ly + 1) * W1 + (tx + 1)];
        double d6p = dxom2 * (d4[-W1] + d4[-1] + d4[1] + d4[W1] - 4.0 * d4[0]);
        double diffus = P.ah2fac[k] * d4[0] - P.ah4fac[k] * d6p;
        double jac = (q[1] - q[-1]) * (p[W1] - p[-W1]) + (q[-W1] - q[W1]) * (p[1] - p[-1]) +
                     q[1] * (p[W1 + 1] - p[-W1 + 1]) - q[-1] * (p[W1 - 1] - p[-W1 - 1]) -
                     q[W1] * (p[W1 + 1] - p[W1 - 1]) + q[-W1] * (p[-W1 + 1] - p[-W1 - 1]) +
                     p[W1] * (q[W1 + 1] - q[W1 - 1]) - p[-W1] * (q[-W1 + 1] - q[-W1 - 1]) -
                     p[1] * (q[W1 + 1] - q[-W1 + 1]) + p[-1] * (q[W1 - 1] - q[-W1 - 1]);
        val = P.adfaco * jac + diffus;
      }
      dq[k][r] = val;
      if (k == NL - 1) d2bot[r] = sd2[(ly + 2) * W2 + (tx + 2)];
    }
    QG_STAMP(3, 2 + 2 * k);
    if (k + 1 < NL) __syncthreads();
  }

  // ---- forcing, bottom drag, leapfrog, projection ----------------------
  if (WTQ) {
    // (every lane goes through: the stores are pairs of neighbouring lanes; points outside the tile's part of the
    //  domain compute on clamped operands and store nothing)
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      int ly = ty0 + r * (TEND_NT / TX);
      const int gi0 = i0 + tx, gj0 = j0 + ly;
      const bool valid = gi0 <= T.imax && gj0 <= T.jmax;
      const int gi = valid ? gi0 : (gi0 <= T.imax ? gi0 : T.imax), gj = gj0 <= T.jmax ? gj0 : T.jmax;
      double dqp[NL], qmp[NL], qop[NL];
      const bool wallrow = (gj + joff == 1 || gj + joff == nyg);
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        dqp[k] = dq[k][r];
        qmp[k] = e_qm[k][r];
        qop[k] = AVG ? e_qo[AVG ? k : 0][r] : (wallrow ? P.qo[fs * k + (long)(gj - 1) * ldx + (gi - 1)] : 0.0);
      }
      tend_point<NL, CYC, true, AVG>(P, gi0, gj, dqp, d2bot[r], qmp, qop, e_wek[r], e_ent[r], e_ddy[r], valid);
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      int ly = ty0 + r * (TEND_NT / TX);
      int gi = i0 + tx, gj = j0 + ly;
      if (gi > T.imax || gj > T.jmax) continue;
      double dqp[NL], qmp[NL], qop[NL];
      const bool wallrow = (gj + joff == 1 || gj + joff == nyg);
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        dqp[k] = dq[k][r];
        qmp[k] = e_qm[k][r];
        qop[k] = AVG ? e_qo[AVG ? k : 0][r] : (wallrow ? P.qo[fs * k + (long)(gj - 1) * ldx + (gi - 1)] : 0.0);
      }
      tend_point<NL, CYC, false, AVG>(P, gi, gj, dqp, d2bot[r], qmp, qop, e_wek[r], e_ent[r], e_ddy[r]);
    }
  }
  QG_STAMP(3, 7);
  QG_STAMP_DRAIN();
  QG_STAMP(3, 8);
}
