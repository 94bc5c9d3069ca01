// k_rfft64.h - zonally periodic rows of length N = 64*M (atmosphere: nxta = 384 = 64*6; cyclic oceans with
// nxto = 192, 384, 960): real FFT rows, one wavefront per row PAIR, no workgroup barriers - the cyclic twin
// of k_dst64.h.  Replaces FFTPACK drfftf / drfftb in hscyat / hscyoc (src/atisubs.F:343-345, 380-384;
// src/ocisubs.F:566-568, 601-605).
//
// Two real rows ride one complex FFT of length N (z = a + i b); the spectra are kept in FFTPACK's half-complex order
//     r(1) = X_0, r(2k) = Re X_k, r(2k+1) = Im X_k, r(N) = X_{N/2}
// exactly as the generic kernel k_rfft_cyc (k_dst.h) keeps them, so the reference's bd2at / bd2oc ordering
// (src/q-gcm.F:935-943, 961-970) applies unchanged and the two kernels are interchangeable.
// Complex FFT of length 64*M by one wave (as dst64_core): lane n2 holds z[64*n1 + n2], n1 = 0..M-1, in registers;
// M-point DFT over n1 in registers, twiddle W_N^(n2*k1), then the M 64-point FFTs over n2 through LDS as two radix-8
// passes on rows padded 1-per-8; Z[k], k = k1 + M*k2, k2 = c + 8d, ends up at F[k1*72 + 9c + d].
//
//   k_rfft64<M, INV>            forward / inverse rows in place (helmholtz, stand-alone entry points)
//   k_rfft64_unpack<M, NL, ..>  inverse rows of the NL modes of one row pair fused with the homogeneous corrections,
//                               modes -> layers (src/atisubs.F:262-288, src/ocisubs.F:300-327) and the zonal-boundary
//                               PV (atqzbd / ocqbdy); an extra wave carries the second half of the constraint algebra
#pragma once
#include "k_cyclic.h"
#include "k_dst64.h"

#pragma clang fp contract(fast)

// 6 = 2 x 3 Cooley-Tukey: n = 3*na + nb, k = ka + 2*kb
template <>
__device__ __forceinline__ void dftM<6>(cplx *a) {
  const double s3 = 0.86602540378443864676;
  const cplx w1 = {0.5, -s3}, w2 = {-0.5, -s3}; // W6^1, W6^2
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) { // radix-2 over na: t[ka][nb] at a[3*ka + nb]
    cplx t0 = cadd(a[nb], a[3 + nb]), t1 = csub(a[nb], a[3 + nb]);
    a[nb] = t0;
    a[3 + nb] = t1;
  }
  a[3 + 1] = cmul(a[3 + 1], w1);
  a[3 + 2] = cmul(a[3 + 2], w2);
  dft3(a[0], a[1], a[2]); // X[0 + 2*kb] at a[kb]
  dft3(a[3], a[4], a[5]); // X[1 + 2*kb] at a[3 + kb]
  cplx t[6];
#pragma unroll
  for (int ka = 0; ka < 2; ++ka)
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) t[ka + 2 * kb] = a[3 * ka + kb];
#pragma unroll
  for (int k = 0; k < 6; ++k) a[k] = t[k];
}

// LDS location of Z[k]
template <int M>
__device__ __forceinline__ int zloc64(int k) {
  const int k1 = k % M, k2 = k / M;
  return k1 * D64_ROW + 9 * (k2 & 7) + (k2 >> 3);
}

// forward complex FFT of length 64*M: in a[n1] = z[64*n1 + lane]; out Z in F (layout above). W64: 64 cplx of LDS.
template <int M>
__device__ __forceinline__ void fft64M(const QgDstParams &P, cplx *a, cplx *F, cplx *W64, int lane) {
  cplx tw1[M]; // W_N^(lane*k1) as powers of W_N^lane
  {
    double2 w = P.twid[lane];
    tw1[1 % M] = {w.x, w.y};
#pragma unroll
    for (int k1 = 2; k1 < M; ++k1) tw1[k1] = cmul(tw1[k1 / 2], tw1[k1 - k1 / 2]);
  }
  {
    double2 w = P.twid[M * lane];
    W64[lane] = {w.x, w.y};
  }
  dftM<M>(a);
#pragma unroll
  for (int k1 = 0; k1 < M; ++k1) {
    cplx v = (k1 == 0) ? a[0] : cmul(a[k1], tw1[k1]);
    F[k1 * D64_ROW + lane + (lane >> 3)] = v;
  }
  wave_lds_sync();
  constexpr int NBF = M * 8;
  // radix-8 over a (n2 = 8a + b), twiddle W64^(b*c), in place
#pragma unroll
  for (int rd = 0; rd < (NBF + 63) / 64; ++rd) {
    const int id = lane + 64 * rd;
    if (id < NBF) {
      const int k1 = id >> 3, b = id & 7;
      cplx *row = F + k1 * D64_ROW + b;
      cplx x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = row[9 * q];
      dft8(x);
#pragma unroll
      for (int c = 1; c < 8; ++c) x[c] = cmul(x[c], W64[b * c]);
#pragma unroll
      for (int c = 0; c < 8; ++c) row[9 * c] = x[c];
    }
  }
  wave_lds_sync();
  // radix-8 over b for fixed c: positions 9c + b -> 9c + d (k2 = c + 8d)
#pragma unroll
  for (int rd = 0; rd < (NBF + 63) / 64; ++rd) {
    const int id = lane + 64 * rd;
    if (id < NBF) {
      const int k1 = id >> 3, c = id & 7;
      cplx *row = F + k1 * D64_ROW + 9 * c;
      cplx x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = row[q];
      dft8(x);
#pragma unroll
      for (int d = 0; d < 8; ++d) row[d] = x[d];
    }
  }
  wave_lds_sync();
}

// half-complex rows a, b (global) -> a[n1] = conj(Z)[64*n1 + lane], Z_k = Xa_k + i Xb_k (as k_rfft_cyc<true>)
template <int M>
__device__ __forceinline__ void rfft64_load_spectrum(const double *rowa, const double *rowb, bool has_b, int lane, cplx *a) {
  constexpr int N = 64 * M, H = N / 2;
#pragma unroll
  for (int n1 = 0; n1 < M; ++n1) {
    const int j = 64 * n1 + lane;
    const int k = (j <= H) ? j : N - j;
    double ar, ai, br, bi;
    if (k == 0) {
      ar = rowa[0]; ai = 0.0;
      br = has_b ? rowb[0] : 0.0; bi = 0.0;
    } else if (k == H) {
      ar = rowa[N - 1]; ai = 0.0;
      br = has_b ? rowb[N - 1] : 0.0; bi = 0.0;
    } else {
      ar = rowa[2 * k - 1]; ai = rowa[2 * k];
      br = has_b ? rowb[2 * k - 1] : 0.0; bi = has_b ? rowb[2 * k] : 0.0;
    }
    a[n1] = (j <= H) ? cplx{ar - bi, -(ai + br)} : cplx{ar + bi, -(br - ai)};
  }
}

// grid: (ceil(npairs / D64_WAVES), nlayers), block 64*D64_WAVES = independent waves; rows transformed in place
template <int M, bool INV>
__global__ __launch_bounds__(D64_NT) void k_rfft64(const QgDstParams P) {
  constexpr int N = 64 * M, H = N / 2;
  __shared__ __align__(16) cplx Fsh[D64_WAVES][M * D64_ROW];
  __shared__ __align__(16) cplx W64sh[D64_WAVES][64];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // scalar: uniform row pointers (k_dst64.h)
  const int m = blockIdx.y + P.layer0;
  const int pair = blockIdx.x * D64_WAVES + wv;
  const int ja = P.g.jr0 + 2 * pair;
  if (ja > P.g.jr1) return; // whole wave leaves; no workgroup barrier is ever used
  const bool has_b = (ja + 1 <= P.g.jr1);
  double *rowa = P.wrk + P.g.wstride * m + (long)(ja - 1) * P.g.ldw;
  double *rowb = rowa + P.g.ldw;
  cplx *F = Fsh[wv];
  cplx a[M];
  if (!INV) {
#pragma unroll
    for (int n1 = 0; n1 < M; ++n1) a[n1] = {rowa[64 * n1 + lane], has_b ? rowb[64 * n1 + lane] : 0.0};
  } else {
    rfft64_load_spectrum<M>(rowa, rowb, has_b, lane, a);
  }
  fft64M<M>(P, a, F, W64sh[wv], lane);
  if (!INV) {
    // Xa_k = (Z_k + conj Z_{N-k})/2, Xb_k = (Z_k - conj Z_{N-k})/(2i)
#pragma unroll
    for (int t = 0; t < (H + 1 + 63) / 64; ++t) {
      const int k = lane + 64 * t;
      if (k <= H) {
        cplx z1 = F[zloc64<M>(k)], z2 = F[zloc64<M>((N - k) % N)];
        double ar = 0.5 * (z1.x + z2.x), ai = 0.5 * (z1.y - z2.y);
        double br = 0.5 * (z1.y + z2.y), bi = -0.5 * (z1.x - z2.x);
        if (k == 0) {
          rowa[0] = ar;
          if (has_b) rowb[0] = br;
        } else if (k == H) {
          rowa[N - 1] = ar;
          if (has_b) rowb[N - 1] = br;
        } else {
          rowa[2 * k - 1] = ar; rowa[2 * k] = ai;
          if (has_b) { rowb[2 * k - 1] = br; rowb[2 * k] = bi; }
        }
      }
    }
  } else {
#pragma unroll
    for (int n1 = 0; n1 < M; ++n1) {
      const int j = 64 * n1 + lane;
      cplx z = F[zloc64<M>(j)];
      rowa[j] = z.x;
      if (has_b) rowb[j] = -z.y;
    }
  }
}

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// Inverse rows FUSED with the rest of atinvq / cyclic ocinvq and atqzbd / ocqbdy: a workgroup = NL waves = the NL
// modes of one row pair (+ one constraint wave with CONSTR).  Each wave transforms its mode into LDS; after one
// barrier all waves combine the modes point by point with the expressions of k_unpack_cyc (contraction off: bitwise
// the separate launches), writing the new pa / po and, for the workgroups of the first / last interior row pair,
// the zonal boundary rows and their PV.  The transformed field never goes to HBM.
// CONSTR: the constraint wave runs constr_cyc_partB (c1, c2, c3 from the zonal-mean column the Thomas sweep left in
// wrk; part A - boundary sums, leapfrog of the constraint vectors - ran in the Thomas launch) redundantly in every
// workgroup, hidden behind the transforms; workgroup 0 records the scalars (dpiat / dpioc, c1, c2, c3, xinhom).
// Its parameters come BY VALUE (kernel arguments: scalar loads, pointers known to be global): read through a pointer to a
// device copy, every field was a flat load and every array behind it a second, dependent round trip - in the wave
// the other three wait for.
// grid: (npairs), block 64*(NL + CONSTR)
// ---------------------------------------------------------------------------
template <int M, int NL, bool BDY, bool CONSTR>
__global__ __launch_bounds__(64 * (NL + (CONSTR ? 1 : 0))) void k_rfft64_unpack(const QgDstParams P, const QgUnpackParams U,
                                                                                 const QgBdyParams B,
                                                                                 const QgCycConstrParams Q) {
  constexpr int N = 64 * M;
  __shared__ __align__(16) cplx Fsh[NL][M * D64_ROW];
  __shared__ __align__(16) cplx W64sh[NL][64];
  __shared__ double cc_sh[2 * NL + 1]; // c1(1..NL-1), c2(1..NL-1), c3
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // = mode (scalar: uniform row pointers)
  const int ny = U.g.ny, nx = U.g.nx, nxt = U.g.nxt;
  if (CONSTR && wv == NL) { // the constraint wave
    double c1[NL], c2[NL], c3, ocs[NL], ocn[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) { // the new constraint vectors: part A ran in the Thomas launch
      ocs[k] = Q.sc->ocncs[k];
      ocn[k] = Q.sc->ocncn[k];
    }
    constr_cyc_partB<NL>(Q, lane, blockIdx.x == 0, ocs, ocn, c1, c2, c3);
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < NL - 1; ++m) {
        cc_sh[m] = c1[m];
        cc_sh[NL + m] = c2[m];
      }
      cc_sh[2 * NL] = c3;
    }
    __syncthreads(); // the barrier the transforming waves reach after their transform
    return;
  }
  const int ja = P.g.jr0 + 2 * blockIdx.x; // local rows ja, ja+1
  const bool has_b = (ja + 1 <= P.g.jr1);
  {
    const double *rowa = P.wrk + P.g.wstride * wv + (long)(ja - 1) * P.g.ldw;
    cplx a[M];
    rfft64_load_spectrum<M>(rowa, rowa + P.g.ldw, has_b, lane, a);
    fft64M<M>(P, a, Fsh[wv], W64sh[wv], lane);
  }
  __syncthreads();
  double c1[NL], c2[NL], c3;
  if (CONSTR) {
#pragma unroll
    for (int m = 0; m < NL - 1; ++m) {
      c1[m] = cc_sh[m];
      c2[m] = cc_sh[NL + m];
    }
    c3 = cc_sh[2 * NL];
  } else {
#pragma unroll
    for (int m = 0; m < NL - 1; ++m) {
      c1[m] = U.sc->c1[m];
      c2[m] = U.sc->c2[m];
    }
    c3 = U.sc->c3;
  }
  const long fs = U.g.fstride;
  // modal values of point (gi, row): sel 0 row a, 1 row b, -1 zonal boundary row (inhomogeneous part vanishes)
  auto point = [&](int gi, int jrow, int sel, double *pl) {
    const int ci = (gi > nxt) ? 0 : gi - 1; // column nx is column 1
    double pm[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      double wvv = 0.0;
      if (sel >= 0) {
        const cplx z = Fsh[m][zloc64<M>(ci)];
        wvv = sel ? -z.y : z.x;
      }
      const double homcor = (m == 0) ? c3 * U.pbh[jrow - 1]
                                     : c1[m - 1] * U.pch1[(jrow - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(jrow - 1) + (long)ny * (m - 1)];
      pm[m] = wvv + homcor;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; ++m) v = v + U.ctm2l[m + NL * k] * pm[m];
      pl[k] = v;
    }
  };
  constexpr int NT = 64 * NL;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && !has_b) break;
    const int gj = ja + r;
    for (int gi = tid + 1; gi <= nx; gi += NT) {
      double pl[NL];
      point(gi, gj, r, pl);
      const long o = (long)(gj - 1) * U.g.ldx + (gi - 1);
#pragma unroll
      for (int k = 0; k < NL; ++k) U.pnew[fs * k + o] = pl[k];
    }
    // zonal boundary rows (1 and ny): done by the workgroup that holds the adjacent interior row
    const int wall = (gj == 2) ? 1 : (gj == ny - 1 ? ny : 0);
    if (!wall) continue;
    // (a fixed number of rounds with a predicate, the topography values of all rounds requested first: as a loop over gi -
    //  trip count depends on the thread, it stays rolled - the two workgroups of the boundary rows paid one memory round
    //  trip per round at the very end of the launch)
    constexpr int NRW = (64 * M + 1 + NT - 1) / NT;
    double ddw[NRW];
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
      const int gi = tid + 1 + it * NT;
      ddw[it] = BDY ? B.ddynoc[(long)(wall - 1) * U.g.ldx + ((gi <= nx ? gi : 1) - 1)] : 0.0;
    }
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
      const int gi = tid + 1 + it * NT;
      if (gi > nx) break;
      double pw[NL], pin[NL];
      point(gi, wall, -1, pw);
      const long ow = (long)(wall - 1) * U.g.ldx + (gi - 1);
#pragma unroll
      for (int k = 0; k < NL; ++k) U.pnew[fs * k + ow] = pw[k];
      if (BDY) {
        point(gi, gj, r, pin);
        const double by = B.beta * B.yporel[wall - 1];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          double ap;
          if (k == 0) ap = B.f0A[0] * pw[0] + B.f0A[NL] * pw[1];
          else if (k == NL - 1) ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k];
          else ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k] + B.f0A[k + NL * (k + 1)] * pw[k + 1];
          if (U.g.atm && k == NL - 1 && wall == 1) // southern value of the top layer: src/vorsubs.F:470 reads row 2
            ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pin[k];
          double q = B.bcfaco_f0 * (pin[k] - pw[k]) - ap + by;
          if (k == (U.g.atm ? 0 : NL - 1)) q = q + ddw[it]; // topography: ocean layer nlo, atmosphere layer 1
          B.qo[fs * k + ow] = q;
        }
      }
    }
  }
}
