// k_cyclic.h - zonally cyclic ocean (-Dcyclic_ocean): momentum-constraint pieces.
//
//  k_cyc_bsums   boundary line sums that qgostep/ocadif accumulate for the
//                momentum constraints: Jacobian sums ajisoc/ajinoc
//                (src/qgosubs.F:279-297, 404-423), third/fifth-derivative sums
//                ap3/ap5 s/n oc (:429-443) and the bottom-drag sums bdrins/bdrinn
//                (:150-163).  They use the state BEFORE the step.
//  k_constr_cyc  ocinvq's constraint algebra (src/ocisubs.F:169-294): line
//                integrals ayis/ayin of the new modal solutions, c1, c2, c3,
//                dpioc update, ocncs/ocncn leapfrog.  All it needs of the new
//                solution - the area integral and the sums of rows 2 and nypo-1 -
//                is the zonal-mean (k = 0) column of the Thomas sweeps: the sum of
//                a row over one period is nxto times its mean coefficient.  So the
//                kernel runs BEFORE the inverse row transform (no row-sum pass over
//                the transformed field) and the inverse transform can be fused with
//                the unpack step.
//  k_unpack_cyc  homogeneous corrections + modes->layers (src/ocisubs.F:300-327),
//                optionally fused with ocqbdy (zonal boundaries only).
#pragma once
#include "qgcm_dev.h"

struct QgCycSumParams {
  QgGeom g;
  const double *pom, *po, *qo; // state before the step
  double *part; // (5, BSUM_NB, 2, nl) partial line sums
  double bcfaco, dxom2, adfaco, fnot, dxo, dyo;
};

// periodic column index (1-based): columns 1..nxt are stored, column nx == column 1
__device__ __forceinline__ int cyc_col(int i, int nxt) {
  if (i < 1) i += nxt;
  else if (i > nxt) i -= nxt;
  return i;
}

#define BSUM_NB 16 // blocks along x per (layer, side)

// One block of 256 threads = (layer k, side, slice of columns): nl * 2 * BSUM_NB blocks in all, block index
// bidx = (k * 2 + north) * BSUM_NB + slice.  They ride at the end of the grid of k_tend<NL, true> (the sums use the
// state BEFORE the step, which k_tend only reads), so the cyclic step has no launch for them.  Fixed-order
// reduction (wave butterfly, then the four wave totals left to right); the BSUM_NB partial sums per quantity are
// added, in block order, by k_constr_cyc (deterministic).
__device__ __forceinline__ void cyc_bsums_block(const QgCycSumParams &P, int bidx) {
  __shared__ double redw[5][4];
  const int tid = threadIdx.x;
  const int slice = bidx % BSUM_NB, north = (bidx / BSUM_NB) & 1, k = bidx / (2 * BSUM_NB);
  const int nx = P.g.nx, nxt = P.g.nxt, ldx = P.g.ldx;
  const long fs = P.g.fstride;
  const double *pom = P.pom + fs * k, *p = P.po + fs * k, *q = P.qo + fs * k;
  const double bcf = P.bcfaco, dxom2 = P.dxom2;
  // y-slabs: the sums belong to the rank that owns the zonal boundary (global row 1 / nyg); the other ranks publish
  // zeros (the consumers read rank 0's southern and the last rank's northern sums)
  const bool own = north ? (P.g.jhi + P.g.joff == P.g.nyg) : (P.g.jlo + P.g.joff == 1);
  if (!own) {
    if (tid < 5) P.part[(long)bidx * 5 + tid] = 0.0;
    return;
  }
  // rows counted from the boundary inwards: r = 0 boundary row, 1, 2, 3
  auto row = [&](int r) { return north ? (P.g.jhi - 1 - r) : (P.g.jlo - 1 + r); }; // 0-based local row
  auto PM = [&](int i, int r) { return pom[(long)row(r) * ldx + (cyc_col(i, nxt) - 1)]; };
  // Del^2 of pom on rows r = 0 (boundary, mixed BC), 1, 2 (interior 5-point, cyclic in x)
  auto D2 = [&](int i, int r) {
    if (r == 0) return bcf * (PM(i, 1) - PM(i, 0));
    return (PM(i, r - 1) + PM(i - 1, r) + PM(i + 1, r) + PM(i, r + 1) - 4.0 * PM(i, r)) * dxom2;
  };
  auto D4 = [&](int i, int r) { // r = 0 or 1
    if (r == 0) return bcf * (D2(i, 1) - D2(i, 0));
    return dxom2 * (D2(i, 0) + D2(i - 1, 1) + D2(i + 1, 1) + D2(i, 2) - 4.0 * D2(i, 1));
  };
  // note: for the northern boundary "r-1" is the row to the north, "r+1" to the south; the
  // 5-point operator is symmetric so the value is the same, only the summation order of the
  // two meridional neighbours differs from the reference (rounding level).
  double s5 = 0.0, s9 = 0.0, s3 = 0.0, s5d = 0.0, sb = 0.0;
  const int per = (nx + BSUM_NB - 1) / BSUM_NB;
  const int ibeg = 1 + slice * per, iend = min(nx, ibeg + per - 1);
  for (int i = tid < 256 ? ibeg + tid : iend + 1; i <= iend; i += 256) {
    // Jacobian sums: weights 0.5 at i = 1 and i = nx (the same point), 1 inside
    const double wgt = (i == 1 || i == nx) ? 0.5 : 1.0;
    const int ic = cyc_col(i, nxt);
    const double dp = p[(long)row(1) * ldx + (cyc_col(ic + 1, nxt) - 1)] - p[(long)row(1) * ldx + (cyc_col(ic - 1, nxt) - 1)];
    s5 += wgt * q[(long)row(0) * ldx + (ic - 1)] * dp;
    s9 += wgt * q[(long)row(1) * ldx + (ic - 1)] * dp;
    if (i <= nx - 1) {
      s3 += D2(i, 1) - D2(i, 0);
      s5d += D4(i, 1) - D4(i, 0);
      if (k == P.g.nl - 1) sb += PM(i, 1) - PM(i, 0);
    }
  }
  double sv[5] = {s5, s9, s3, s5d, sb};
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int v = 0; v < 5; ++v) sv[v] += __shfl_xor(sv[v], off);
  }
  if ((tid & 63) == 0 && tid < 256) {
#pragma unroll
    for (int v = 0; v < 5; ++v) redw[v][tid >> 6] = sv[v];
  }
  __syncthreads();
  if (tid == 0) {
    double *o = P.part + (long)bidx * 5;
    for (int v = 0; v < 5; ++v) o[v] = ((redw[v][0] + redw[v][1]) + redw[v][2]) + redw[v][3];
  }
}

// ---------------------------------------------------------------------------
struct QgCycConstrParams {
  QgGeom g;
  const double *bpart; // partial boundary line sums of k_cyc_bsums (southern sums are read from here)
  const double *bpart_n; // y-slabs: the northern sums (the last rank's part of the gathered step message); else = bpart
  const double *ybnd;  // y-slabs: (2, nl) zonal-mean values of the solved modes at global rows 2 and nyg-1
                       // (k_thomas PHASE 2, from the slab summaries); nullptr: read them from wrk
  double adfaco, delek_sgn; // 1/(12 dxo dyo f0); 0.5*sign(f0)*delek
  double ah2oc[QG_MAXL], ah4oc[QG_MAXL];
  const double *ksum, *wrk; // spectral column sums (k_thomas PHASE 0) and the solved spectral rows
  QgScalars *sc;
  QgConstr cs;
  double dxo, dyo, tdto, fnot;
  double gpoc[QG_MAXL], hoc[QG_MAXL];
  double ctl2m[QG_MAXL * QG_MAXL], ctm2l[QG_MAXL * QG_MAXL];
};

// Part A (one wave; needs the boundary line sums of this step's tendency launch): ajis.., ap3.., ap5.., bdrins;
// right-hand sides of the momentum constraints and the leapfrog step of the constraint vectors
// (src/ocisubs.F:176-206, src/atisubs.F:177-214).  Lane 0 updates the state; every lane returns the new vectors.
template <int NL>
__device__ __forceinline__ void constr_cyc_partA(const QgCycConstrParams &P, int lane, double *ocsnew, double *ocnnew) {
  // boundary line sums of the previous qgostep: lane (5*(2k+side) + v) adds the BSUM_NB block
  // partials of quantity v in block order (src/qgosubs.F:150-163, 279-297, 404-443)
  // (10 NL sums on 64 lanes: a second round for seven and eight layers)
  constexpr int NR = (10 * NL + 63) / 64;
  double bs[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    bs[r] = 0.0;
    const int idx = lane + 64 * r;
    if (idx < 10 * NL) {
      const int ks = idx / 5, v = idx % 5;
      const double *src = (ks & 1) ? P.bpart_n : P.bpart;
      for (int blk = 0; blk < BSUM_NB; ++blk) bs[r] += src[((long)ks * BSUM_NB + blk) * 5 + v];
    }
  }
  double bq[2 * NL][5];
#pragma unroll
  for (int ks = 0; ks < 2 * NL; ++ks)
#pragma unroll
    for (int v = 0; v < 5; ++v) bq[ks][v] = __shfl(bs[(5 * ks + v) / 64], (5 * ks + v) % 64);
  QgScalars *sc = P.sc;
  const double fnot = P.fnot, tdto = P.tdto;
  const double entfac = 0.5 * P.dyo * fnot * fnot;
  double ajis[NL], ajin[NL], ap3s[NL], ap3n[NL], ap5s[NL], ap5n[NL], bdrins = 0.0, bdrinn = 0.0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    // south: sums as written; north: the reference's sums carry the opposite sign
    // (qgosubs.F:409-420 Jacobian with leading minus; :436,438 "boundary minus inner")
    const double *S = bq[2 * k], *Nn = bq[2 * k + 1];
    ajis[k] = P.dxo * P.dyo * (fnot * P.adfaco * (S[0] + 2.0 * S[1]));
    ajin[k] = P.dxo * P.dyo * (fnot * P.adfaco * (-Nn[0] + 2.0 * (-Nn[1])));
    ap3s[k] = P.ah2oc[k] * S[2];
    ap3n[k] = P.ah2oc[k] * (-Nn[2]);
    ap5s[k] = P.ah4oc[k] * S[3];
    ap5n[k] = P.ah4oc[k] * (-Nn[3]);
    if (k == NL - 1) {
      bdrins = P.delek_sgn * S[4];
      bdrinn = P.delek_sgn * (-Nn[4]);
    }
  }
  double enis[NL], enin[NL];
#pragma unroll
  for (int k = 0; k < NL - 1; ++k) {
    enis[k] = sc->enisoc[k];
    enin[k] = sc->eninoc[k];
  }
  const double txis = sc->txisoc, txin = sc->txinoc;
  double rhss[NL], rhsn[NL];
  if (P.g.atm) {
    // atmosphere, src/atisubs.F:177-196: layer 1 is the bottom layer (stress and entrainment enter with the
    // opposite sign), no Del-4th and no drag terms
    rhss[0] = -(entfac / P.hoc[0]) * enis[0] - (fnot / P.hoc[0]) * txis + ajis[0] + ap5s[0];
    rhsn[0] = -(entfac / P.hoc[0]) * enin[0] + (fnot / P.hoc[0]) * txin + ajin[0] - ap5n[0];
#pragma unroll
    for (int k = 1; k < NL - 1; ++k) {
      rhss[k] = -(entfac / P.hoc[k]) * (enis[k] - enis[k - 1]) + ajis[k] + ap5s[k];
      rhsn[k] = -(entfac / P.hoc[k]) * (enin[k] - enin[k - 1]) + ajin[k] - ap5n[k];
    }
    rhss[NL - 1] = (entfac / P.hoc[NL - 1]) * enis[NL - 2] + ajis[NL - 1] + ap5s[NL - 1];
    rhsn[NL - 1] = (entfac / P.hoc[NL - 1]) * enin[NL - 2] + ajin[NL - 1] - ap5n[NL - 1];
  } else {
    // ocisubs.F:176-193
    rhss[0] = (entfac / P.hoc[0]) * enis[0] + (fnot / P.hoc[0]) * txis + ajis[0] - ap3s[0] + ap5s[0];
    rhsn[0] = (entfac / P.hoc[0]) * enin[0] - (fnot / P.hoc[0]) * txin + ajin[0] + ap3n[0] - ap5n[0];
#pragma unroll
    for (int k = 1; k < NL - 1; ++k) {
      rhss[k] = (entfac / P.hoc[k]) * (enis[k] - enis[k - 1]) + ajis[k] - ap3s[k] + ap5s[k];
      rhsn[k] = (entfac / P.hoc[k]) * (enin[k] - enin[k - 1]) + ajin[k] + ap3n[k] - ap5n[k];
    }
    rhss[NL - 1] = -(entfac / P.hoc[NL - 1]) * enis[NL - 2] + ajis[NL - 1] - ap3s[NL - 1] + ap5s[NL - 1] +
                   (fnot / P.hoc[NL - 1]) * bdrins;
    rhsn[NL - 1] = -(entfac / P.hoc[NL - 1]) * enin[NL - 2] + ajin[NL - 1] + ap3n[NL - 1] - ap5n[NL - 1] -
                   (fnot / P.hoc[NL - 1]) * bdrinn;
  }
  // ocisubs.F:199-206
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    ocsnew[k] = sc->ocncsp[k] + tdto * rhss[k];
    ocnnew[k] = sc->ocncnp[k] + tdto * rhsn[k];
  }
  if (lane != 0) return;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    sc->ajisoc[k] = ajis[k]; sc->ajinoc[k] = ajin[k];
    sc->ap3soc[k] = ap3s[k]; sc->ap3noc[k] = ap3n[k];
    sc->ap5soc[k] = ap5s[k]; sc->ap5noc[k] = ap5n[k];
    sc->ocncsp[k] = sc->ocncs[k];
    sc->ocncnp[k] = sc->ocncn[k];
    sc->ocncs[k] = ocsnew[k];
    sc->ocncn[k] = ocnnew[k];
  }
  sc->bdrins = bdrins;
  sc->bdrinn = bdrinn;
}

// Part B (needs the Thomas sweeps): line and area integrals of the new modal solutions from the zonal-mean column,
// c1, c2, c3 (src/ocisubs.F:212-244), area integrals of the layer pressures and the dpioc / dpiat update (:246-294).
// ocsnew / ocnnew = the NEW constraint vectors (part A).  Every lane computes the same; `record`: lane 0 stores
// xinhom, c1, c2, c3 and steps dpioc / dpiocp.
template <int NL>
__device__ __forceinline__ void constr_cyc_partB(const QgCycConstrParams &P, int lane, bool record, const double *ocsnew,
                                                 const double *ocnnew, double *c1, double *c2, double &c3) {
  const int ny = P.g.ny;
  double s[NL], ys[NL], yn[NL];
  // Sums over one period in x are nxto times the zonal-mean coefficient (spectral index 0 of the half-complex row;
  // the Thomas sweep has already applied ftnorm = 1/nxto and the inverse transform is unnormalised):
  //   area integral (xintp, ocisubs.F:160; rows 1 and nypo vanish): nxto * ksum(m, 0) - k_thomas's column sum;
  //   line integrals along rows 2 and nypo-1 (ocisubs.F:216-225): 0.5 w(1) + sum_{2..nx-1} + 0.5 w(nx) with
  //   w(nx) = w(1) is the sum over the nxto stored columns = nxto * wrk(k = 0, row).
  const double xn = (double)P.g.nxt;
#pragma unroll
  for (int m = 0; m < NL; ++m) {
    s[m] = xn * P.ksum[(long)m * P.g.ldw];
    ys[m] = xn * (P.ybnd ? P.ybnd[2 * m] : P.wrk[P.g.wstride * m + (long)1 * P.g.ldw]);
    yn[m] = -(xn * (P.ybnd ? P.ybnd[2 * m + 1] : P.wrk[P.g.wstride * m + (long)(ny - 2) * P.g.ldw]));
  }
  double xin[NL], clhss[NL], clhsn[NL];
#pragma unroll
  for (int m = 0; m < NL; ++m) xin[m] = s[m] * P.dxo * P.dyo;
  // ocisubs.F:212-234
#pragma unroll
  for (int m = 0; m < NL; ++m) {
    double ayis = ys[m] * (P.dxo / P.dyo), ayin = yn[m] * (P.dxo / P.dyo);
    double cs = 0.0, cn = 0.0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      cs = cs + P.ctl2m[k + NL * m] * ocsnew[k];
      cn = cn + P.ctl2m[k + NL * m] * ocnnew[k];
    }
    clhss[m] = cs + ayis;
    clhsn[m] = cn - ayin;
  }
  c3 = clhss[0] * P.cs.hbsioc;
#pragma unroll
  for (int m = 0; m < NL - 1; ++m) {
    c1[m] = P.cs.hc2noc[m] * clhss[m + 1] - P.cs.hc2soc[m] * clhsn[m + 1];
    c2[m] = P.cs.hc1soc[m] * clhsn[m + 1] - P.cs.hc1noc[m] * clhss[m + 1];
  }
  if (!record || lane != 0) return;
  QgScalars *sc = P.sc;
  double aipmod[NL], aiplay[NL];
#pragma unroll
  for (int m = 0; m < NL; ++m) sc->xinhom[m] = xin[m];
#pragma unroll
  for (int m = 0; m < NL - 1; ++m) {
    sc->c1[m] = c1[m];
    sc->c2[m] = c2[m];
  }
  sc->c3 = c3;
  aipmod[0] = xin[0] + c3 * P.cs.aipbho;
#pragma unroll
  for (int m = 1; m < NL; ++m) aipmod[m] = xin[m] + (c1[m - 1] + c2[m - 1]) * P.cs.aipcho[m - 1];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    double pl = 0.0;
#pragma unroll
    for (int m = 0; m < NL; ++m) pl = pl + P.ctm2l[m + NL * k] * aipmod[m];
    aiplay[k] = pl;
  }
  // ocisubs.F:268-294 / atisubs.F:232-257: the continuity monitors ermaso, emfroc (ermasa, emfrat) - the new integral
  // of the interface displacement against the one the entrainment predicts - then the step of dpioc / dpiocp
  const double ecrit = 1.0e-13; // ocisubs.F:93, atisubs.F:86
  const double area = ((double)P.g.nxt * P.dxo) * ((double)(P.g.nyg - 1) * P.dyo); // xlo*ylo (xla*yla)
#pragma unroll
  for (int k = 0; k < NL - 1; ++k) {
    const double est1 = P.g.atm ? aiplay[k] - aiplay[k + 1] /* dpiat, src/atisubs.F:256 */ : aiplay[k + 1] - aiplay[k];
    const double est2 = sc->dpiocp[k] - P.tdto * P.gpoc[k] * sc->xon[k];
    const double edif = est1 - est2;
    const double esum = fabs(est1) + fabs(est2);
    sc->ermas[k] = edif;
    sc->emfr[k] = (esum > ecrit * area * P.tdto * P.gpoc[k]) ? 2.0 * edif / esum : 0.0;
    sc->dpiocp[k] = sc->dpioc[k];
    sc->dpioc[k] = est1;
  }
}

// part B alone, reading the new constraint vectors part A left in the scalars (k_rfft_cyc's extra workgroup)
__device__ __forceinline__ void rfft_cyc_constr_partB(const QgCycConstrParams *Q, int lane) {
  double c1[QG_MAXL], c2[QG_MAXL], c3, ocs[QG_MAXL], ocn[QG_MAXL];
#pragma unroll
  for (int k = 0; k < QG_MAXL; ++k) { // (all QG_MAXL slots exist in the scalars; compile-time indices keep them in registers)
    ocs[k] = Q->sc->ocncs[k];
    ocn[k] = Q->sc->ocncn[k];
  }
  switch (Q->g.nl) {
    case 2: constr_cyc_partB<2>(*Q, lane, true, ocs, ocn, c1, c2, c3); break;
    case 3: constr_cyc_partB<3>(*Q, lane, true, ocs, ocn, c1, c2, c3); break;
    default: constr_cyc_partB<4>(*Q, lane, true, ocs, ocn, c1, c2, c3); break;
  }
}

// both parts in one launch (stand-alone atinvq / ocinvq, generic row sizes)
template <int NL>
__global__ __launch_bounds__(64) void k_constr_cyc(const QgCycConstrParams P) {
  const int lane = threadIdx.x;
  double ocsnew[NL], ocnnew[NL], c1[NL], c2[NL], c3;
  constr_cyc_partA<NL>(P, lane, ocsnew, ocnnew);
  constr_cyc_partB<NL>(P, lane, true, ocsnew, ocnnew, c1, c2, c3);
}

// ---------------------------------------------------------------------------
// grid: (ceil(nx/256), ny)
template <int NL, bool BDY>
__global__ __launch_bounds__(256) void k_unpack_cyc(const QgUnpackParams P, const QgBdyParams B) {
  const int nx = P.g.nx, ny = P.g.ny, nxt = P.g.nxt;
  const int nyg = P.g.nyg, joff = P.g.joff;
  const int gi0 = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const int gj = blockIdx.y + P.g.jlo; // owned rows only (the halo rows of a y-slab come with the exchange)
  if (gj > P.g.jhi) return;
  const bool valid = gi0 <= nx;
  const int gi = valid ? gi0 : nx; // (lanes past the row keep the wave whole for the pair stores; they write nothing)
  const long o = (long)(gj - 1) * P.g.ldx + (gi0 - 1);
  const int ci = (gi > nxt) ? 0 : gi - 1; // column nx is column 1
  auto point = [&](int jrow, double *pl) {
    const bool inner = (jrow + joff >= 2 && jrow + joff <= nyg - 1);
    double pm[NL];
    pm[0] = (inner ? P.wrk[(long)(jrow - 1) * P.g.ldw + ci] : 0.0) + P.sc->c3 * P.pbh[jrow - 1];
#pragma unroll
    for (int m = 1; m < NL; ++m) {
      double wv = inner ? P.wrk[P.g.wstride * m + (long)(jrow - 1) * P.g.ldw + ci] : 0.0;
      double homcor = P.sc->c1[m - 1] * P.pch1[(jrow - 1) + (long)ny * (m - 1)] + P.sc->c2[m - 1] * P.pch2[(jrow - 1) + (long)ny * (m - 1)];
      pm[m] = wv + homcor;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; ++m) v = v + P.ctm2l[m + NL * k] * pm[m];
      pl[k] = v;
    }
  };
  double pl[NL];
  point(gj, pl);
#pragma unroll
  for (int k = 0; k < NL; ++k) qg_pair_store_wt(P.pnew + P.g.fstride * k + o, pl[k], valid); // (qgcm_dev.h)
  if (!valid) return;
  const int G = gj + joff;
  // y-slab halo messages (k_halo_pack's layout, k_misc.h), written here when the host passes the buffers (slab
  // stage 2); an edge row with a neighbour is never a zonal boundary row, so its q is what k_tend set
  if (BDY && P.msg_lo && gj - P.g.jlo < 3) {
#pragma unroll
    for (int k = 0; k < NL; ++k) P.msg_lo[((long)k * 3 + (gj - P.g.jlo)) * P.g.ldx + (gi - 1)] = pl[k];
    if (gj == P.g.jlo)
#pragma unroll
      for (int k = 0; k < NL; ++k) P.msg_lo[((long)NL * 3 + k) * P.g.ldx + (gi - 1)] = B.qo[P.g.fstride * k + o];
  }
  if (BDY && P.msg_hi && P.g.jhi - gj < 3) {
#pragma unroll
    for (int k = 0; k < NL; ++k) P.msg_hi[((long)k * 3 + (gj - (P.g.jhi - 2))) * P.g.ldx + (gi - 1)] = pl[k];
    if (gj == P.g.jhi)
#pragma unroll
      for (int k = 0; k < NL; ++k) P.msg_hi[((long)NL * 3 + k) * P.g.ldx + (gi - 1)] = B.qo[P.g.fstride * k + o];
  }
  if (BDY && (G == 1 || G == nyg)) {
    double pin[NL];
    point(G == 1 ? gj + 1 : gj - 1, pin);
    const double by = B.beta * B.yporel[gj - 1];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double ap;
      if (k == 0) ap = B.f0A[0] * pl[0] + B.f0A[NL] * pl[1];
      else if (k == NL - 1) ap = B.f0A[k + NL * (k - 1)] * pl[k - 1] + B.f0A[k + NL * k] * pl[k];
      else ap = B.f0A[k + NL * (k - 1)] * pl[k - 1] + B.f0A[k + NL * k] * pl[k] + B.f0A[k + NL * (k + 1)] * pl[k + 1];
      if (P.g.atm && k == NL - 1 && G == 1) // southern value of the top layer: src/vorsubs.F:470 reads row 2
        ap = B.f0A[k + NL * (k - 1)] * pl[k - 1] + B.f0A[k + NL * k] * pin[k];
      double q = B.bcfaco_f0 * (pin[k] - pl[k]) - ap + by;
      if (k == (P.g.atm ? 0 : NL - 1)) q = q + B.ddynoc[o]; // topography: ocean bottom layer nlo, atmosphere layer 1
      B.qo[P.g.fstride * k + o] = q;
    }
  }
}
