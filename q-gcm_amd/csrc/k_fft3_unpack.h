// k_fft3_unpack.h - long rows (three-stage plans of k_fft3.h): inverse row transform, homogeneous corrections, modes ->
// layers and the boundary PV in ONE launch, without the round trip of the solved modes through the work array.
//
// The step after the y sweeps was   inverse rows (read wrk, write wrk)  ->  unpack (read wrk, write po)
// (src/ocisubs.F:383-401 after :444-452 / :601-605): at nxto = 4608 / 4800 that is the work array read twice and
// written once more than needed.  Modes -> layers is a pointwise nl x nl mix and the row transform is linear along the
// row, so the two commute: the workgroup of (row pair, layer k) reads the nl spectral rows of its pair, mixes them
// with ctm2l(:,k) while they go into LDS, runs ONE inverse transform and stores the layer's pressure rows directly,
// adding the homogeneous part.  Per row pair the transform count is the same as before (nl), the nl workgroups of a
// pair are renumbered onto the same XCD so that the second and third read of the spectral rows hit its L2, and HBM
// sees wrk once (read) and po once (write).  The result differs from the two-launch path by rounding only (the mix is
// done before the transform instead of after it): the parity tests of the long-row configurations hold it to the
// reference within the same tolerances as before.
#pragma once
#include "k_fft3.h"

// blockIdx.x -> (row pair, layer): ids x, x + 8, x + 16 ... run on the same XCD one after another
__device__ __forceinline__ bool fft3u_block(int id, int nl, int nrp, int &rp, int &kl) {
  const int xcd = id & 7, t = id >> 3;
  kl = t % nl;
  rp = (t / nl) * 8 + xcd;
  return rp < nrp;
}

// boundary PV of layer kl on a zonal boundary row of the ocean (src/vorsubs.F:245-388): pl = the row's pressure in every
// layer (constants of the row here), pin = the inward neighbour's pressure in layer kl
template <int NL>
__device__ __forceinline__ double fft3u_bdy_q(const QgBdyParams &B, int kl, const double *pl, double pin, double by) {
  double ap = 0.0, plk = 0.0; // (0.0 + x is x: the terms add up in the order l = kl - 1, kl, kl + 1 of the reference)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    if (l == kl) plk = pl[l];
    if (l >= kl - 1 && l <= kl + 1) ap = ap + B.f0A[kl + NL * l] * pl[l];
  }
  return B.bcfaco_f0 * (pin - plk) - ap + by;
}

// ---------------------------------------------------------------------------
// zonally cyclic ocean: half-complex spectral rows -> layer pressures (k_rfft_cyc<true> + k_unpack_cyc<.., true>);
// not for the atmospheric channel (its rows are short: k_rfft64_unpack)
// grid: 8 * ceil(row pairs / 8) * NL workgroups of NT threads; dynamic LDS as k_rfft_cyc
// own_constr: part B of the constraint algebra (c1, c2, c3 from the zonal-mean column, k_cyclic.h) is evaluated by
// every workgroup for itself and recorded by the first; 0: read from the scalars (y-slabs: k_constr_cyc ran before)
// ---------------------------------------------------------------------------
// AVG: the leapfrog averaging that follows the step (src/q-gcm.F:1345-1351) folded into the stores - the new po leaves as
// 0.5*(po_new + U.pavg), the PV of the zonal boundary rows as 0.5*(q + U.qavg) (k_dst64_unpack<.., AVG> is the box twin);
// whole-domain steps only, instantiated beside the plain kernel, which the other 24 steps of 25 run unchanged.
template <class PLAN, int NL, int NT, bool AVG = false>
__global__ __launch_bounds__(NT) void k_rfft3_unpack(const QgDstParams D, const QgUnpackParams U, const QgBdyParams B,
                                                     const QgCycConstrParams Q, const int own_constr) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  constexpr int NC = PLAN::N, HC = NC / 2;
  const int tid = threadIdx.x;
  const int nrp = (D.g.jr1 - D.g.jr0 + 2) / 2;
  int rp, kl;
  if (!fft3u_block(blockIdx.x, NL, nrp, rp, kl)) return;
  const int ldw = D.g.ldw, ldx = D.g.ldx, ny = D.g.ny;
  const long ws = D.g.wstride, fs = D.g.fstride;
  const int ja = D.g.jr0 + 2 * rp; // local rows ja, ja + 1
  const bool has_b = ja + 1 <= D.g.jr1;
  typename PLAN::Tw tw3 = PLAN::template prefetch<NT>(D.twid, tid); // table values requested before the rows
  QG_STAMP(2, 0);

  // ---- constraint coefficients and the homogeneous part of the rows (constants along a row) ----------------------
  double c1[NL], c2[NL], c3;
  {
    double ocs[NL], ocn[NL], d1[NL], d2[NL], d3;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      ocs[k] = U.sc->ocncs[k];
      ocn[k] = U.sc->ocncn[k];
    }
    constr_cyc_partB<NL>(Q, tid, false, ocs, ocn, d1, d2, d3);
    c3 = own_constr ? d3 : U.sc->c3;
#pragma unroll
    for (int m = 0; m < NL - 1; ++m) {
      c1[m] = own_constr ? d1[m] : U.sc->c1[m];
      c2[m] = own_constr ? d2[m] : U.sc->c2[m];
    }
  }
  // sum_m ctm2l(m, l) * (homogeneous part of mode m on local row j)
  auto homrow = [&](int j, int l) -> double {
    double v = U.ctm2l[NL * l] * (c3 * U.pbh[j - 1]);
#pragma unroll
    for (int m = 1; m < NL; ++m)
      v = v + U.ctm2l[m + NL * l] * (c1[m - 1] * U.pch1[(j - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(j - 1) + (long)ny * (m - 1)]);
    return v;
  };
  double cm[NL];
#pragma unroll
  for (int m = 0; m < NL; ++m) cm[m] = U.ctm2l[m + NL * kl];
  const double ha = homrow(ja, kl), hb = homrow(has_b ? ja + 1 : ja, kl);

  // ---- the NL spectral rows of the pair, mixed on their way into LDS (half-complex rows -> conj(Z), as k_rfft_cyc) --
  {
    const double *ra[NL], *rb[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      ra[m] = D.wrk + (ws * m + (long)(ja - 1) * ldw);
      rb[m] = has_b ? ra[m] + ldw : ra[m];
    }
    const double bs = has_b ? 1.0 : 0.0;
    constexpr int NIT = (HC + 1 + NT - 1) / NT, CH = (NIT + 1) / 2;
#pragma unroll
    for (int c0 = 0; c0 < NIT; c0 += CH) {
      double2 va[NL][CH], vb[NL][CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = tid + (c0 + i) * NT;
        const int kc = (c0 + i < NIT && k <= HC) ? k : 0;
        const unsigned i1 = (kc == 0) ? 0 : (kc == HC ? NC - 1 : 2 * kc - 1), i2 = (kc == 0 || kc == HC) ? i1 : 2 * kc;
#pragma unroll
        for (int m = 0; m < NL; ++m) {
          va[m][i] = {ra[m][i1], ra[m][i2]};
          vb[m][i] = {rb[m][i1], rb[m][i2]};
        }
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = tid + (c0 + i) * NT;
        if (c0 + i < NIT && k <= HC) {
          double ax = cm[0] * va[0][i].x, ay = cm[0] * va[0][i].y, bx = cm[0] * vb[0][i].x, by = cm[0] * vb[0][i].y;
#pragma unroll
          for (int m = 1; m < NL; ++m) {
            ax = fma(cm[m], va[m][i].x, ax);
            ay = fma(cm[m], va[m][i].y, ay);
            bx = fma(cm[m], vb[m][i].x, bx);
            by = fma(cm[m], vb[m][i].y, by);
          }
          const bool edge = (k == 0 || k == HC);
          const double ar = ax, ai = edge ? 0.0 : ay;
          const double br = bs * bx, bi = edge ? 0.0 : bs * by;
          A[PLAN::pos_in(k)] = {ar - bi, -(ai + br)};
          if (!edge) A[PLAN::pos_in(NC - k)] = {ar + bi, -(br - ai)};
        }
      }
    }
  }
  __syncthreads();
  QG_STAMP(2, 1);
  // AVG: this step's po of the two rows, requested before the transform (in the store loop below the loads would sit
  // between the write-through stores, one exposed round trip per round)
  constexpr int NSTA = (NC / 2 + NT - 1) / NT;
  double2 pva[AVG ? NSTA : 1], pvb[AVG ? NSTA : 1];
  if (AVG) {
    const double *va0 = U.pavg + fs * kl + (long)(ja - 1) * ldx;
#pragma unroll
    for (int it = 0; it < NSTA; ++it) {
      const int t = tid + it * NT, tc = t < NC / 2 ? t : 0;
      pva[it] = *reinterpret_cast<const double2 *>(va0 + 2 * tc);
      pvb[it] = *reinterpret_cast<const double2 *>(va0 + (has_b ? ldx : 0) + 2 * tc);
    }
  }
  PLAN::template run<NT>(A, tw3, tid);
  QG_STAMP(2, 2);

  // ---- stores: the two pressure rows of layer kl, halo messages, zonal boundary rows + their PV -------------------
  const int jlo = D.g.jlo, jhi = D.g.jhi;
  double *pa = U.pnew + fs * kl + (long)(ja - 1) * ldx, *pb = pa + ldx;
  auto avg_p = [&](long idx, double v) { return AVG ? 0.5 * (v + U.pavg[fs * kl + idx]) : v; };
  auto avg_q = [&](long idx, double v) { return AVG ? 0.5 * (v + U.qavg[fs * kl + idx]) : v; };
  // this workgroup's rows next to a zonal boundary of the basin: it writes the boundary row of its layer as well
  const bool doS = rp == 0 && D.g.jr0 + D.g.joff == 2;
  const bool doN = rp == nrp - 1 && D.g.jr1 + D.g.joff == D.g.nyg - 1;
  const int jS = D.g.jr0 - 1, jN = D.g.jr1 + 1;
  double hS[NL], hN[NL], hSk = 0.0, hNk = 0.0;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    hS[l] = doS ? homrow(jS, l) : 0.0;
    hN[l] = doN ? homrow(jN, l) : 0.0;
    if (l == kl) {
      hSk = hS[l];
      hNk = hN[l];
    }
  }
  const bool topo = kl == NL - 1; // topography term: bottom layer nlo
  const double byS = doS ? B.beta * B.yporel[jS - 1] : 0.0, byN = doN ? B.beta * B.yporel[jN - 1] : 0.0;
  // halo messages (k_halo_pack's layout): p rows gj - jlo < 3 / jhi - gj < 3, q row gj == jlo / jhi (set by k_tend)
  auto msg_p = [&](int gj, int col, double v) {
    if (U.msg_lo && gj - jlo < 3) U.msg_lo[((long)kl * 3 + (gj - jlo)) * ldx + col] = v;
    if (U.msg_hi && jhi - gj < 3) U.msg_hi[((long)kl * 3 + (gj - (jhi - 2))) * ldx + col] = v;
  };
  auto msg_q = [&](int gj, int col) {
    const bool lo = U.msg_lo && gj == jlo, hi = U.msg_hi && gj == jhi;
    if (lo || hi) {
      const double q = B.qo[fs * kl + (long)(gj - 1) * ldx + col];
      if (lo) U.msg_lo[((long)NL * 3 + kl) * ldx + col] = q;
      if (hi) U.msg_hi[((long)NL * 3 + kl) * ldx + col] = q;
    }
  };
  // one column of a boundary row: its pressure is the row constant, its PV follows from the inward neighbour pin
  auto bdy_col = [&](bool south, int col, double pin) {
    const int jb = south ? jS : jN;
    const long o = (long)(jb - 1) * ldx + col;
    double q = fft3u_bdy_q<NL>(B, kl, south ? hS : hN, pin, south ? byS : byN);
    if (topo) q = q + B.ddynoc[o];
    B.qo[fs * kl + o] = avg_q(o, q);
  };
  // Every workgroup stores its two rows in the same plain loop (the kernel is bound by instruction issue).  The few with
  // more to do - halo messages, a zonal boundary row - come back for it in a second pass below.
  constexpr int NST = (NC / 2 + NT - 1) / NT;
  // pos_out(2 (t + NT)) - pos_out(2 t) is a constant when 2 NT is a multiple of R1 R2
  constexpr bool STEP = (2 * NT) % PLAN::R1R2 == 0;
  const int p0 = PLAN::pos_out(2 * tid), p1 = PLAN::pos_out(2 * tid + 1);
  double a00 = 0.0, b00 = 0.0; // column 1 of the two rows (thread 0)
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int t = tid + it * NT;
    if (t < NC / 2) {
      const cplx z0 = A[STEP ? p0 + it * (2 * NT / PLAN::R1R2) : PLAN::pos_out(2 * t)];
      const cplx z1 = A[STEP ? p1 + it * (2 * NT / PLAN::R1R2) : PLAN::pos_out(2 * t + 1)];
      const double a0 = z0.x + ha, a1 = z1.x + ha, b0 = hb - z0.y, b1 = hb - z1.y;
      if (AVG) {
        const double2 ca = pva[AVG ? it : 0], cb = pvb[AVG ? it : 0];
        qg_store16_wt(pa + 2 * t, 0.5 * (a0 + ca.x), 0.5 * (a1 + ca.y));
        if (has_b) qg_store16_wt(pb + 2 * t, 0.5 * (b0 + cb.x), 0.5 * (b1 + cb.y));
      } else {
        qg_store16_wt(pa + 2 * t, a0, a1);
        if (has_b) qg_store16_wt(pb + 2 * t, b0, b1);
      }
      if (it == 0) {
        a00 = a0;
        b00 = b0;
      }
    }
  }
  if (doS || doN || U.msg_lo || U.msg_hi) {
    // second pass of the workgroups at a slab edge / a zonal boundary (the transform's output is still in LDS): all they
    // read from global memory - the topography under the boundary rows, the q rows of the halo messages - is requested
    // for every round FIRST.  (As a loop with the loads inside, these workgroups - the last row pair's are dispatched
    // last - paid one memory round trip per round at the very end of the launch.)
    const bool mq_a = (U.msg_lo && ja == jlo) || (U.msg_hi && ja == jhi);
    const bool mq_b = has_b && ((U.msg_lo && ja + 1 == jlo) || (U.msg_hi && ja + 1 == jhi));
    auto msg_qv = [&](int gj, int col, double q) {
      if (U.msg_lo && gj == jlo) U.msg_lo[((long)NL * 3 + kl) * ldx + col] = q;
      if (U.msg_hi && gj == jhi) U.msg_hi[((long)NL * 3 + kl) * ldx + col] = q;
    };
    constexpr int CHK = 3; // rounds per batch of requests (all NST at once would not fit the registers)
#pragma unroll
    for (int c0 = 0; c0 < NST; c0 += CHK) {
      double2 ddS[CHK], ddN[CHK], qa[CHK], qb[CHK];
#pragma unroll
      for (int i = 0; i < CHK; ++i) {
        const int t = tid + (c0 + i) * NT, tc = (c0 + i < NST && t < NC / 2) ? t : 0;
        // (unconditional loads from rows that exist, the values selected afterwards: `cond ? *p : zero` made the
        //  compiler select between the ADDRESS p and a zero kept in scratch memory, and load through a flat pointer)
        const double2 vS = *reinterpret_cast<const double2 *>(B.ddynoc + (doS ? (long)(jS - 1) * ldx : 0) + 2 * tc);
        const double2 vN = *reinterpret_cast<const double2 *>(B.ddynoc + (doN ? (long)(jN - 1) * ldx : 0) + 2 * tc);
        qa[i] = *reinterpret_cast<const double2 *>(B.qo + fs * kl + (long)(ja - 1) * ldx + 2 * tc);
        qb[i] = *reinterpret_cast<const double2 *>(B.qo + fs * kl + (long)(has_b ? ja : ja - 1) * ldx + 2 * tc);
        ddS[i].x = (doS && topo) ? vS.x : 0.0;
        ddS[i].y = (doS && topo) ? vS.y : 0.0;
        ddN[i].x = (doN && topo) ? vN.x : 0.0;
        ddN[i].y = (doN && topo) ? vN.y : 0.0;
      }
#pragma unroll
      for (int i = 0; i < CHK; ++i) {
        const int t = tid + (c0 + i) * NT;
        if (c0 + i < NST && t < NC / 2) {
          const cplx z0 = A[PLAN::pos_out(2 * t)], z1 = A[PLAN::pos_out(2 * t + 1)];
          const double a0 = z0.x + ha, a1 = z1.x + ha, b0 = hb - z0.y, b1 = hb - z1.y;
          if (U.msg_lo || U.msg_hi) {
            msg_p(ja, 2 * t, a0); msg_p(ja, 2 * t + 1, a1);
            if (mq_a) { msg_qv(ja, 2 * t, qa[i].x); msg_qv(ja, 2 * t + 1, qa[i].y); }
            if (has_b) {
              msg_p(ja + 1, 2 * t, b0); msg_p(ja + 1, 2 * t + 1, b1);
              if (mq_b) { msg_qv(ja + 1, 2 * t, qb[i].x); msg_qv(ja + 1, 2 * t + 1, qb[i].y); }
            }
          }
          if (doS) {
            const long o = (long)(jS - 1) * ldx + 2 * t;
            *reinterpret_cast<double2 *>(U.pnew + fs * kl + o) = double2{avg_p(o, hSk), avg_p(o + 1, hSk)};
            const double q0 = fft3u_bdy_q<NL>(B, kl, hS, a0, byS) + ddS[i].x, q1 = fft3u_bdy_q<NL>(B, kl, hS, a1, byS) + ddS[i].y;
            *reinterpret_cast<double2 *>(B.qo + fs * kl + o) = double2{avg_q(o, q0), avg_q(o + 1, q1)};
          }
          if (doN) {
            const long o = (long)(jN - 1) * ldx + 2 * t;
            *reinterpret_cast<double2 *>(U.pnew + fs * kl + o) = double2{avg_p(o, hNk), avg_p(o + 1, hNk)};
            const double n0 = has_b ? b0 : a0, n1 = has_b ? b1 : a1;
            const double q0 = fft3u_bdy_q<NL>(B, kl, hN, n0, byN) + ddN[i].x, q1 = fft3u_bdy_q<NL>(B, kl, hN, n1, byN) + ddN[i].y;
            *reinterpret_cast<double2 *>(B.qo + fs * kl + o) = double2{avg_q(o, q0), avg_q(o + 1, q1)};
          }
        }
      }
    }
  }
  if (tid == 0) { // column nxpo is column 1 (src/ocisubs.F:604)
    pa[NC] = avg_p((long)(ja - 1) * ldx + NC, a00);
    if (has_b) pb[NC] = avg_p((long)ja * ldx + NC, b00);
    if (U.msg_lo || U.msg_hi) {
      msg_p(ja, NC, a00); msg_q(ja, NC);
      if (has_b) { msg_p(ja + 1, NC, b00); msg_q(ja + 1, NC); }
    }
    if (doS) {
      U.pnew[fs * kl + (long)(jS - 1) * ldx + NC] = avg_p((long)(jS - 1) * ldx + NC, hSk);
      bdy_col(true, NC, a00);
    }
    if (doN) {
      U.pnew[fs * kl + (long)(jN - 1) * ldx + NC] = avg_p((long)(jN - 1) * ldx + NC, hNk);
      bdy_col(false, NC, has_b ? b00 : a00);
    }
  }
  QG_STAMP(2, 3);
  QG_STAMP_DRAIN();
  QG_STAMP(2, 4);
  // the first workgroup records part B: xinhom, c1, c2, c3, the continuity monitors and the step of dpioc / dpiocp
  // (none of which the other workgroups read)
  if (own_constr && blockIdx.x == 0 && tid == 0) {
    double ocs[NL], ocn[NL], d1[NL], d2[NL], d3;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      ocs[k] = U.sc->ocncs[k];
      ocn[k] = U.sc->ocncn[k];
    }
    constr_cyc_partB<NL>(Q, 0, true, ocs, ocn, d1, d2, d3);
  }
}
