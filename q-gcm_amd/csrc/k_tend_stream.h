// k_tend_stream.h - K1 as a register-streaming kernel: PV tendency + leapfrog + layer->mode projection.
//
// Same arithmetic as k_tend.h (reference: del2p src/qgosubs.F:86-130, ocadif :306-400, forcing / drag / leapfrog
// :173-219, projection src/ocisubs.F:117-139; every expression keeps the reference's association order, contraction
// off: qgostep stays bit for bit the CPU reference) with the data movement turned round.  k_tend.h stages a 22 x 22
// halo tile per layer in LDS and walks the Del^2 -> Del^4 -> Del^6 cascade through three barrier-separated LDS
// passes (1400 instructions per wave, 296 of them fp64 arithmetic: instruction-issue bound).  Here
//
//   * one WAVE owns 64 consecutive columns of ONE layer and marches north through TS_S rows: lane = column, the
//     three-row windows of pom, Del^2, Del^4, po and qo live in registers and advance by one row per tick;
//   * the east / west neighbours of the stencils come from DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1),
//     not from LDS: lanes 0-2 and 61-63 are halo lanes (the cascade has radius 3), lanes 3..60 own TS_SW = 58 columns;
//   * every row of every field is loaded once per strip (one 512-byte run per wave and load instruction), TS_PF rows
//     ahead of its use, so a wave always has loads in flight while it computes - no load / compute / store phases;
//   * the NL layer-waves of a strip meet ONCE, after the march: each leaves its rows of (q_new - beta*y [- ddyn]) in
//     LDS, then the waves share out the rows and project them onto the modes (the only LDS traffic, no barrier in
//     the march).
// A unit = (strip, tile row of TS_S rows, all layers); a workgroup = SPW units (256 threads for 2 and 4 layers, 384
// for 3: the riders of workgroup 0 and the edge / line-sum workgroups at the end of the grid keep their 256-thread
// reductions).  Units are numbered so that each XCD sweeps a contiguous band of tile rows (halo rows and columns
// re-read by neighbouring units hit that XCD's L2).
// Redundancy: 6 of 64 lanes, and the warm-up of the cascade (6 extra rows of pom, 2 of po / qo per 16-row unit).
//
// Algorithmic HBM traffic: as k_tend.h, (6 nl + 3) N doubles = 21 N for nl = 3.
#pragma once
#include "k_tend.h"

#define TS_SW 58        // columns owned by a strip (64 lanes - 2 x 3 halo lanes)
#define TS_S TEND_TY    // rows per unit (= the tile rows of the y-slab split, TEND_INNER / TEND_OUTER)
#ifndef TS_PF
#define TS_PF 2         // rows a wave loads ahead of the row it computes
#endif

template <int NL>
struct TsCfg {
  static constexpr int SPW = (NL == 4) ? 1 : 2; // units per workgroup
  static constexpr int NT = 64 * NL * SPW;      // 256, 384, 256 threads for 2, 3, 4 layers
};

struct TsTiling {
  int gx, gy, jmax, erow, nedge;
};

// strips cover i = 1..nx (the E wall column sits in the last strip); tile rows as tend_tiling: the N wall row of a box
// whose row count is 16 h + 1 is peeled off into edge workgroups (tend_edge) instead of a tile row of its own
template <bool CYC>
__host__ __device__ __forceinline__ TsTiling ts_tiling(const QgGeom &g) {
  TsTiling T;
  const int rows = g.jhi - g.jlo + 1;
  T.erow = (!CYC && rows % TS_S == 1 && rows > 1 && g.jhi + g.joff == g.nyg) ? 1 : 0;
  T.gx = (g.nx + TS_SW - 1) / TS_SW;
  T.gy = T.erow ? rows / TS_S : (rows + TS_S - 1) / TS_S;
  T.jmax = T.erow ? g.jhi - 1 : g.jhi;
  T.nedge = T.erow ? (g.nx + TEND_NT - 1) / TEND_NT : 0;
  return T;
}

// value of the lane to the west (lane - 1) / east (lane + 1); lanes 0 / 63 receive 0 (halo lanes; bound_ctrl, so
// that the move has no tied "old" operand: with one, every DPP move is preceded by a plain v_mov)
template <int CTRL>
__device__ __forceinline__ double ts_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#define TS_WEST(v) ts_dpp<0x138>(v) // wave_shr:1
#define TS_EAST(v) ts_dpp<0x130>(v) // wave_shl:1

// lane / row rules of one strip
struct TsRules {
  bool wl;       // wave-uniform: the strip holds a wall column (box only)
  bool isW, isE; // this lane is the W / E wall column
  int nyg, joff;
  double bcf, dxom2;
};

// Five-point operator of one row with the reference's mixed boundary rule (qgosubs.F:94-127 for Del^2, :310-341 for
// Del^4): interior points (cS + cW + cE + cN - 4 c0) * dxom2; a wall point takes bcf * (inner neighbour - itself),
// the inner neighbour being N on the S wall row, S on the N wall row, E on the W wall column, W on the E wall column
// (rows first: the corners follow the row rule, as k_tend.h's i2 / i4 offsets).  gj: local row of c0 (row-uniform).
// EDGE = false: a unit none of whose stencils touches a wall row or a wall column - the interior formula only.
template <bool CYC, bool EDGE>
__device__ __forceinline__ double ts_lap(const TsRules &R, double cS, double c0, double cN, int gj) {
  if (EDGE) {
    const int G = gj + R.joff;
    if (G == 1) return R.bcf * (cN - c0);
    if (G == R.nyg) return R.bcf * (cS - c0);
  }
  const double cW = TS_WEST(c0), cE = TS_EAST(c0);
  double v = (cS + cW + cE + cN - 4.0 * c0) * R.dxom2;
  if (EDGE && !CYC && R.wl) {
    const double vw = R.bcf * ((R.isW ? cE : cW) - c0);
    v = (R.isW || R.isE) ? vw : v;
  }
  return v;
}

// what a wave needs for its march (all wave-uniform except the lane's column offsets and flags)
struct TsWave {
  const double *pomk, *pok, *qok, *fwek, *fent, *fddy; // (fwek / fddy point at entoc for the layers that use neither)
  double *qnk;
  double *ql;   // LDS: this layer's rows of the unit, [TS_S][64]
  double ypl;   // lane t: yporel of row j0 + t
  int k, j0, jmax, ny, ldx;
  unsigned lcol8, ecl8; // byte offsets of the lane's stencil column (wrapped / clamped) and own column in a row
  bool outl, cint, atm;
  double adfaco, tdto, bdrfac, beta, ah2, ah4, foh;
};

// element at byte offset `off` (32-bit, per lane) of the row that starts at the wave-uniform pointer `row`: keeps the
// address in the "scalar base + 32-bit lane offset" form of global_load (no 64-bit address arithmetic in VGPRs)
__device__ __forceinline__ double ts_ld(const double *row, unsigned off) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(row) + off);
}

// The march of one wave through the TS_S rows of its unit (see the header comment).  EDGE = false: every row the
// unit reads exists, none of its rows is a wall row, the strip holds no wall column, all TS_S rows are stepped -
// no clamps, no rules, no per-row tests (all but the first / last tile rows and the first / last strips of a box).
// The layer of the wave is a run-time (wave-uniform) number: what depends on it is selected, not branched on.
template <int NL, bool CYC, bool EDGE>
__device__ __forceinline__ void ts_march(const TsWave &Wv, const TsRules &R) {
  constexpr int SR = TS_S, PF = TS_PF;
  const int k = Wv.k, j0 = Wv.j0, ny = Wv.ny, ldx = Wv.ldx;
  const unsigned lcol8 = Wv.lcol8, ecl8 = Wv.ecl8;
  const double dxom2 = R.dxom2;
  const bool atm = Wv.atm;
  const bool k0 = (k == 0), k1 = (k == 1), kdrag = (!atm && k == NL - 1), kddy = (k == (atm ? 0 : NL - 1));
  const double *__restrict__ pomk = Wv.pomk;
  const double *__restrict__ pok = Wv.pok;
  const double *__restrict__ qok = Wv.qok;
  double *__restrict__ qnk = Wv.qnk;
  // byte offset of a row inside a field layer (32-bit: the host checks ldx*ny*nl < 2^31 elements, a layer is < 4 GB):
  // every load is "uniform layer pointer + (row bytes + lane bytes)", one 32-bit add per distinct row
  auto rowoff = [&](int gj) {
    if (!EDGE) return (unsigned)((gj - 1) * ldx) * 8u;
    const int rj = gj < 1 ? 1 : (gj > ny ? ny : gj); // clamped: rows outside the local array are never used by a point inside
    return (unsigned)((rj - 1) * ldx) * 8u;
  };
  // ---- warm-up of the cascade: pom rows j0-3 .. j0+2, po / qo rows j0-1, j0 --------------------------------------
  double w[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) w[r] = ts_ld(pomk, rowoff(j0 - 3 + r) + lcol8);
  double pS = ts_ld(pok, rowoff(j0 - 1) + lcol8), pC = ts_ld(pok, rowoff(j0) + lcol8);
  double qS = ts_ld(qok, rowoff(j0 - 1) + lcol8), qC = ts_ld(qok, rowoff(j0) + lcol8);
  // ---- the row queue: tick t consumes pom(j0+t+3), po / qo(j0+t+1), qom(j0+t) and the forcing of row j0+t ----------
  double f_pm[PF], f_p[PF], f_q[PF], f_qm[PF], f_wek[PF], f_ent[PF], f_ddy[PF];
  auto issue = [&](int t, int s) {
    const int gj = j0 + t;
    f_pm[s] = ts_ld(pomk, rowoff(gj + 3) + lcol8);
    const unsigned o1 = rowoff(gj + 1) + lcol8, o0 = rowoff(gj) + ecl8;
    f_p[s] = ts_ld(pok, o1);
    f_q[s] = ts_ld(qok, o1);
    f_qm[s] = ts_ld(qnk, o0);
    f_wek[s] = ts_ld(Wv.fwek, o0);
    f_ent[s] = ts_ld(Wv.fent, o0);
    f_ddy[s] = ts_ld(Wv.fddy, o0);
  };
#pragma unroll
  for (int t = 0; t < PF; ++t) issue(t, t);
  const double d2m2 = ts_lap<CYC, EDGE>(R, w[0], w[1], w[2], j0 - 2);
  const double d2m1 = ts_lap<CYC, EDGE>(R, w[1], w[2], w[3], j0 - 1);
  double d2A = ts_lap<CYC, EDGE>(R, w[2], w[3], w[4], j0);     // Del^2 rows gj, gj+1
  double d2B = ts_lap<CYC, EDGE>(R, w[3], w[4], w[5], j0 + 1);
  double d4A = ts_lap<CYC, EDGE>(R, d2m2, d2m1, d2A, j0 - 1);  // Del^4 rows gj-1, gj
  double d4B = ts_lap<CYC, EDGE>(R, d2m1, d2A, d2B, j0);
  double pmA = w[4], pmB = w[5];                               // pom rows gj+1, gj+2
  double pSw = TS_WEST(pS), pSe = TS_EAST(pS), pCw = TS_WEST(pC), pCe = TS_EAST(pC);
  double qSw = TS_WEST(qS), qSe = TS_EAST(qS), qCw = TS_WEST(qC), qCe = TS_EAST(qC);
  QG_STAMP(3, 1);
#pragma unroll
  for (int t = 0; t < SR; ++t) {
    const int gj = j0 + t;
    if (EDGE && gj > Wv.jmax) break; // (wave-uniform)
    if (t == 4) QG_STAMP(3, 2);
    if (t == 8) QG_STAMP(3, 3);
    if (t == 12) QG_STAMP(3, 4);
    const int s = t % PF;
    const double pmC = f_pm[s], pN = f_p[s], qN = f_q[s], qm = f_qm[s];
    const double wek = f_wek[s], ent = f_ent[s], ddy = f_ddy[s];
    if (t + PF < SR) issue(t + PF, s);
    // cascade: Del^2 of row gj+2, Del^4 of row gj+1, Del^6 of row gj
    const double d2C = ts_lap<CYC, EDGE>(R, pmA, pmB, pmC, gj + 2);
    const double d4C = ts_lap<CYC, EDGE>(R, d2A, d2B, d2C, gj + 1);
    const double d4w = TS_WEST(d4B), d4e = TS_EAST(d4B);
    const double pNw = TS_WEST(pN), pNe = TS_EAST(pN), qNw = TS_WEST(qN), qNe = TS_EAST(qN);
    // Del^6 + Arakawa Jacobian (qgosubs.F:349-399); names: S / C / N rows gj-1 / gj / gj+1, w / e columns i-1 / i+1
    const double d6p = dxom2 * (d4A + d4w + d4e + d4C - 4.0 * d4B);
    const double diffus = Wv.ah2 * d4B - Wv.ah4 * d6p;
    const double jac = (qCe - qCw) * (pN - pS) + (qS - qN) * (pCe - pCw) +
                       qCe * (pNe - pSe) - qCw * (pNw - pSw) -
                       qN * (pNe - pNw) + qS * (pSe - pSw) +
                       pN * (qNe - qNw) - pS * (qSe - qSw) -
                       pCe * (qNe - qSe) + pCw * (qNw - qSw);
    double val = Wv.adfaco * jac + diffus;
    if (EDGE) val = Wv.cint ? val : 0.0; // dqdt = 0 on the wall columns (qgosubs.F:371,397)
    // forcing, bottom drag, leapfrog (qgosubs.F:173-219, src/qgasubs.F:128-131; tend_point of k_tend.h):
    //   ocean  layer 1: dq + foh*(wek - ent), layer 2: dq + foh*ent, bottom layer: ... - bdrfac*Del^2(pom)
    //   atmos. layer 1: dq + foh*(ent - wek), layer 2: dq - foh*ent
    const double fa = atm ? ent : wek, fb = atm ? wek : ent;
    const double fx = k0 ? fa - fb : ent;
    const double ft = Wv.foh * fx;
    const double fsum = val + ft, fdif = val - ft;
    double qdot = k0 ? fsum : (k1 ? (atm ? fdif : fsum) : val);
    const double qdrag = qdot - Wv.bdrfac * d2A;
    qdot = kdrag ? qdrag : qdot;
    double qn = qm + Wv.tdto * qdot;
    if (EDGE) {
      const int G = gj + R.joff;
      if (G == 1 || G == R.nyg) qn = qC; // rows not stepped: the new-qo buffer keeps qo (qgosubs.F:214-219)
    }
    if (Wv.outl) *reinterpret_cast<double *>(reinterpret_cast<char *>(qnk) + (rowoff(gj) + ecl8)) = qn;
    const double yp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(Wv.ypl), t),
                                       __builtin_amdgcn_readlane(__double2loint(Wv.ypl), t));
    const double ql0 = qn - Wv.beta * yp;
    const double ql1 = ql0 - ddy;
    Wv.ql[t * 64 + (int)(threadIdx.x & 63)] = kddy ? ql1 : ql0;
    // advance the windows by one row
    pmA = pmB; pmB = pmC;
    d2A = d2B; d2B = d2C;
    d4A = d4B; d4B = d4C;
    pS = pC; pSw = pCw; pSe = pCe; pC = pN; pCw = pNw; pCe = pNe;
    qS = qC; qSw = qCw; qSe = qCe; qC = qN; qCw = qNw; qCe = qNe;
  }
}

#ifndef TS_WAVES_PER_EU
#define TS_WAVES_PER_EU 1
#endif
template <int NL, bool CYC>
__global__ __launch_bounds__(TsCfg<NL>::NT, TS_WAVES_PER_EU) void k_tend_stream(const QgTendParams P, const QgCycSumParams S, const QgOmlFinal F) {
  constexpr int SPW = TsCfg<NL>::SPW, NT = TsCfg<NL>::NT, SR = TS_S, PF = TS_PF;
  static_assert(NT >= TEND_NT && NT >= OML_NT, "the riders and the edge / line-sum workgroups use the first 256 threads");
  __shared__ double qlbuf[SPW][NL][SR][64]; // (q_new - beta*y [- ddyn]) of the unit's rows, per layer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = wv / NL, k = wv % NL; // unit of the workgroup, layer of this wave (wave-uniform)
  const int nx = P.g.nx, ny = P.g.ny, nxt = P.g.nxt, ldx = P.g.ldx;
  const int nyg = P.g.nyg, joff = P.g.joff, jlo = P.g.jlo;
  const long fs = P.g.fstride;
  const TsTiling T = ts_tiling<CYC>(P.g);
  const int gx = T.gx;
  const int nunits = gx * (P.trows ? P.trows : T.gy); // (a window of the tile rows, or all of them)
  const int nwg = (nunits + SPW - 1) / SPW;
  const int per_xcd = (nwg + 7) / 8;
  if (F.on && blockIdx.x == 0) {
    // mixed layer on, inside qgcm_hip_steps: the last reduction of `oml` rides here (see k_tend.h)
    __shared__ double redf[20];
    oml_final_block(F, redf, tid);
  }
  if (!CYC && P.upd_dpi && blockIdx.x == 0 && tid == 0) constr_dpi_update<NL>(P.sc, P.tdto, P.gpoc); // see QgTendParams
  if ((int)blockIdx.x >= 8 * per_xcd) {
    // cyclic / atmosphere: the boundary line sums for the momentum constraints (k_cyclic.h); box: the N wall row
    if (CYC) {
      cyc_bsums_block(S, (int)blockIdx.x - 8 * per_xcd);
    } else if (tid < TEND_NT) {
      TendTiling Te;
      Te.gx = gx; Te.gy = T.gy; Te.imax = nx; Te.jmax = T.jmax; Te.ecol = 0; Te.erow = T.erow; Te.nedge = T.nedge;
      tend_edge<NL>(P, Te, (int)blockIdx.x - 8 * per_xcd);
    }
    return;
  }
  const int wgt = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); // XCD-aware numbering: a band of tile rows per XCD
  if (wgt >= nwg) return;
  QG_STAMP(3, 0);
  const int unit = wgt * SPW + sub;
  const bool live = unit < nunits; // (odd number of units: the last workgroup's second half only keeps the barrier)
  const int strip = live ? unit % gx : 0, seg = live ? unit / gx : 0;
  const int i0 = strip * TS_SW + 1;                                  // first column the strip owns (1-based)
  const int gi = i0 - 3 + lane;                                      // this lane's column
  const int trow = P.trows ? P.trow0 + seg * P.tstride : seg;
  const int j0 = trow * SR + jlo;                                    // first local row of the unit
  const bool outl = lane >= 3 && lane <= 60 && gi <= nx;             // lanes that own a column (gi >= 1 then)
  const bool cint = CYC ? (gi >= 1 && gi <= nx) : (gi >= 2 && gi <= nx - 1);
  const int wcol = CYC ? (nxt - 1) + 1 : nx;                         // columns that exist for the stencil loads
  // stencil loads: wrapped (channel) and clamped (lanes outside the basin: no point inside ever reads their values)
  int gw = tend_wrap<CYC>(gi, nxt);
  gw = gw < 1 ? 1 : (gw > wcol ? wcol : gw);
  const unsigned lcol = (unsigned)(gw - 1);
  const unsigned ecl = (unsigned)((gi < 1 ? 1 : (gi > nx ? nx : gi)) - 1); // the lane's own (unwrapped) column
  TsRules R;
  R.wl = !CYC && (i0 - 3 <= 1 || i0 + 60 >= nx);
  R.isW = !CYC && gi == 1;
  R.isE = !CYC && gi == nx;
  R.nyg = nyg; R.joff = joff; R.bcf = P.bcfaco; R.dxom2 = P.dxom2;
  const bool atm = CYC && P.g.atm;
  const int c = CYC ? gi - 1 : gi - 2; // spectral / work-array column of the lane

  if (live) {
    TsWave Wv;
    const bool need_wek = (k == 0), need_ddy = (k == (atm ? 0 : NL - 1));
    Wv.pomk = P.pom + fs * k; Wv.pok = P.po + fs * k; Wv.qok = P.qo + fs * k; Wv.qnk = P.qnew + fs * k;
    Wv.fent = P.entoc; Wv.fwek = need_wek ? P.wekpo : P.entoc; Wv.fddy = need_ddy ? P.ddynoc : P.entoc;
    Wv.ql = &qlbuf[sub][k][0][0];
    {
      const int jr = j0 + lane; // lane t: yporel of row j0 + t (clamped, lanes >= TS_S unused)
      Wv.ypl = P.yporel[(jr > ny ? ny : jr) - 1];
    }
    Wv.k = k; Wv.j0 = j0; Wv.jmax = T.jmax; Wv.ny = ny; Wv.ldx = ldx;
    Wv.lcol8 = lcol * 8u; Wv.ecl8 = ecl * 8u; Wv.outl = outl; Wv.cint = cint; Wv.atm = atm;
    Wv.adfaco = P.adfaco; Wv.tdto = P.tdto; Wv.bdrfac = P.bdrfac; Wv.beta = P.beta;
    Wv.ah2 = P.ah2fac[k]; Wv.ah4 = P.ah4fac[k]; Wv.foh = P.fohfac[k < 2 ? k : 0];
    // interior unit: rows j0-3 .. j0+SR+2 exist, the stencil rows j0-2 .. j0+SR+1 hold no wall row, every row is
    // stepped, no wall column in the strip
    const bool edge = R.wl || j0 - 3 < 1 || j0 + SR + 2 > ny || j0 - 2 + joff <= 1 || j0 + SR + 1 + joff >= nyg ||
                      j0 + SR - 1 > T.jmax;
    if (edge) ts_march<NL, CYC, true>(Wv, R);
    else ts_march<NL, CYC, false>(Wv, R);
  }
  QG_STAMP(3, 5);
  __syncthreads();
  QG_STAMP(3, 6);
  if (!live) return;
  // ---- projection onto the modes (ocisubs.F:117-139): the layer-waves share out the rows of the unit -------------
  const bool pv = outl && c >= 0 && c < P.g.nk;
#pragma unroll
  for (int t0 = 0; t0 < SR; t0 += NL) {
    const int t = t0 + k;
    const int gj = j0 + t;
    if (t >= SR || gj > T.jmax) break;
    const int G = gj + joff;
    if (G == 1 || G == nyg) continue;
    double ql[NL];
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) ql[kk] = qlbuf[sub][kk][t][lane];
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      double qmm = 0.0;
#pragma unroll
      for (int kk = 0; kk < NL; ++kk) qmm = qmm + P.ctl2m[kk + NL * m] * ql[kk];
      if (pv) P.wrk[P.g.wstride * m + (long)(gj - 1) * P.g.ldw + c] = P.fnot * qmm;
    }
  }
  QG_STAMP(3, 7);
  QG_STAMP_DRAIN();
  QG_STAMP(3, 8);
}
