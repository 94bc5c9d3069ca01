// k_row1.h - long rows (nxto = 4608, 4800), ONE row per workgroup: the real transform of length N = nxto as a complex
// FFT of length H = N/2 (even samples real part, odd samples imaginary part) plus the split step
//   X_k = ((Z_k + conj Z_{H-k}) - i W_N^k (Z_k - conj Z_{H-k})) / 2 .
//
// The two-rows-per-complex-FFT kernels of k_dst.h need N complex numbers of LDS per workgroup (80 KB at these
// lengths: two workgroups per CU, 864-900 workgroups of one generation and a half per launch - their loads, three
// stages and stores hardly overlap).  Half the length is 39-41 KB: three or four workgroups per CU, twice as many of
// them, and - the point - a workgroup can run the NL modes of one row one after the other with their transformed
// values kept in registers (18 doubles per mode and thread), so that the inverse transform is FUSED with the modes ->
// layers step of ocinvq: the transformed field never goes to HBM (one write and one read of wrk less per step, and
// the k_unpack launch).
//
// Reference: src/ocisubs.F:566-568, 601-605 (drfftf / drfftb in hscyoc), :300-327 (homogeneous corrections, modes ->
// layers), src/vorsubs.F:245-388 (ocqbdy, zonal boundaries).  Spectra stay in FFTPACK's half-complex order
//   r(1) = X_0, r(2k) = Re X_k, r(2k+1) = Im X_k, r(N) = X_{N/2}.
#pragma once
#include "k_fft3.h"
#include "k_cyclic.h"

#pragma clang fp contract(fast)

// The three in-place stages of an Fft3Plan of length H run by NT threads (any NT: butterflies beyond NT take further
// sweeps), twiddles from the table of W_N, N = 2H (index doubled).
template <class PL, int NT>
struct Row1Fft {
  static constexpr int R1 = PL::RA, R2 = PL::RB, R3 = PL::RC;
  static constexpr int I1 = R2 * R3, I2 = R1 * R3, I3 = R1 * R2;
  static constexpr int N1 = (I1 + NT - 1) / NT, N2 = (I2 + NT - 1) / NT, N3 = (I3 + NT - 1) / NT;
  struct Tw {
    double2 w1[N1], w2[N2];
  };
  // requested before the rows (a table load issued inside a stage waits behind the bulk traffic, k_fft3.h)
  static __device__ __forceinline__ Tw prefetch(const double2 *__restrict__ twid, int tid) {
    Tw t;
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int r = tid + it * NT;
      t.w1[it] = twid[2 * (r < I1 ? r : 0)];
    }
#pragma unroll
    for (int it = 0; it < N2; ++it) {
      const int id = tid + it * NT;
      t.w2[it] = twid[2 * (id < I2 ? R1 * (id % R3) : 0)];
    }
    return t;
  }
  // forward complex FFT of the H values at pos_in(0..H-1), result at pos_out(0..H-1); ends with a barrier
  static __device__ __forceinline__ void s1(cplx *A, const Tw &T, int tid) {
#pragma unroll
    for (int it = 0; it < N1; ++it) {
      const int r = tid + it * NT;
      if (r < I1) PL::stage1(A, r, T.w1[it]);
    }
    __syncthreads();
  }
  static __device__ __forceinline__ void s2(cplx *A, const Tw &T, int tid) {
#pragma unroll
    for (int it = 0; it < N2; ++it) {
      const int id = tid + it * NT;
      if (id < I2) PL::stage2(A, id, T.w2[it]);
    }
    __syncthreads();
  }
  static __device__ __forceinline__ void s3(cplx *A, int tid) {
#pragma unroll
    for (int it = 0; it < N3; ++it) {
      const int id = tid + it * NT;
      if (id < I3) PL::stage3(A, id);
    }
    __syncthreads();
  }
  static __device__ __forceinline__ void run(cplx *A, const Tw &T, int tid) {
    s1(A, T, tid);
    s2(A, T, tid);
    s3(A, tid);
  }
};

// lane i takes the value of lane i + 1 / i - 1 (full-wave DPP shifts: VALU only)
template <int CTRL>
__device__ __forceinline__ double row1_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#define ROW1_FROM_NEXT 0x130 // wave_shl:1
#define ROW1_FROM_PREV 0x138 // wave_shr:1

// The split step works on the coefficient pairs (k, H-k), k = 0..H/2: both come from the same two LDS values and one
// twiddle.  The half-complex row keeps (Re X_k, Im X_k) at the odd/even positions (2k-1, 2k), so its 16-byte aligned
// pairs (2t, 2t+1) = (Im X_t, Re X_{t+1}) straddle two coefficients: the lanes of a wave take 64 consecutive k and
// pass one value to the neighbouring lane (DPP), 63 of them store / produce - chunk c covers k = 63c .. 63c+63, the
// last (forward) or first (inverse) lane only provides.  W_N^k = W_N^lane * W_N^(63c): one per-lane table value, one
// wave-uniform one.
template <int H>
struct Row1Chunks {
  static constexpr int NCH = (H / 2 + 1 + 62) / 63;
};

// ---------------------------------------------------------------------------
// drfftf of one row.  grid: (nrows, nlayers); dynamic LDS: PL::LDS_CPLX complex numbers
// ---------------------------------------------------------------------------
template <class PL, int NT>
__global__ __launch_bounds__(NT) void k_rfft1_fwd(const QgDstParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef Row1Fft<PL, NT> F;
  constexpr int H = PL::N, NIT = (H + NT - 1) / NT;
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  const int tid = threadIdx.x;
  const int m = blockIdx.y + P.layer0;
  const int j = P.g.jr0 + blockIdx.x; // 1-based local row
  double *row = P.wrk + P.g.wstride * m + (long)(j - 1) * P.g.ldw;
  QG_STAMP(0, 0);
  const typename F::Tw tw = F::prefetch(P.twid, tid);
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double2 wl = P.twid[lane];
  double2 v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int jj = tid + it * NT;
    v[it] = *reinterpret_cast<const double2 *>(row + 2 * (jj < H ? jj : 0));
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int jj = tid + it * NT;
    if (jj < H) A[PL::pos_in(jj)] = {v[it].x, v[it].y};
  }
  __syncthreads();
  QG_STAMP(0, 1);
  F::s1(A, tw, tid);
  QG_STAMP(0, 2);
  F::s2(A, tw, tid);
  QG_STAMP(0, 3);
  F::s3(A, tid);
  QG_STAMP(0, 4);
  // X_k = ((Z_k + conj Z_{H-k}) - i W^k (Z_k - conj Z_{H-k})) / 2 and X_{H-k} from the same e, d, W^k d
  constexpr int NCH = Row1Chunks<H>::NCH, NW = NT / 64, NITC = (NCH + NW - 1) / NW;
#pragma unroll
  for (int it = 0; it < NITC; ++it) {
    const int c = wv + NW * it; // wave-uniform
    if (c < NCH) {
      const int k = 63 * c + lane, kc = k < H / 2 ? k : H / 2;
      const double2 ws = P.twid[63 * c];
      const cplx wt = (c == 0) ? cplx{wl.x, wl.y} : cmul(cplx{wl.x, wl.y}, cplx{ws.x, ws.y});
      const cplx za = A[PL::pos_out(kc)], zb = A[PL::pos_out((H - kc) % H)];
      const cplx e = {za.x + zb.x, za.y - zb.y}, d = {za.x - zb.x, za.y + zb.y};
      const cplx wd = cmul(wt, d);
      const double xk_re = 0.5 * (e.x + wd.y), xk_im = 0.5 * (e.y - wd.x);  // X_k
      const double xm_re = 0.5 * (e.x - wd.y), xm_im = -0.5 * (e.y + wd.x); // X_{H-k}  (k = 0: X_H = Re Z_0 - Im Z_0)
      const double nre = row1_dpp<ROW1_FROM_NEXT>(xk_re); // Re X_{k+1}
      const double nmi = row1_dpp<ROW1_FROM_NEXT>(xm_im); // Im X_{H-k-1}
      if (lane < 63 && k < H / 2) { // 16-byte write-through stores (qgcm_dev.h)
        qg_store16_wt(row + 2 * k, k == 0 ? xk_re : xk_im, nre);
        qg_store16_wt(row + 2 * (H - 1 - k), nmi, xm_re);
      }
    }
  }
  QG_STAMP(0, 5);
  QG_STAMP_DRAIN();
  QG_STAMP(0, 6);
}

// One mode's half-complex row (16-byte loads, all in flight together) ...
template <int H, int NT>
struct Row1Spec {
  static constexpr int NCH = Row1Chunks<H>::NCH, NW = NT / 64, NITC = (NCH + NW - 1) / NW;
  double2 p[NITC], q[NITC]; // P_k = (Im X_k, Re X_{k+1}),  Q_k = P_{H-k-1} = (Im X_{H-k-1}, Re X_{H-k})
  __device__ __forceinline__ void load(const double *__restrict__ row, int lane, int wv) {
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const int c = wv + NW * it;
      const int k = 63 * c + lane, kc = (c < NCH && k < H / 2) ? k : H / 2;
      p[it] = *reinterpret_cast<const double2 *>(row + 2 * kc);
      q[it] = *reinterpret_cast<const double2 *>(row + 2 * (H - kc - 1));
    }
  }
  // ... -> conj(Z'_k) at pos_in(k), Z'_k = (X_k + conj X_{H-k}) + i conj(W^k) (X_k - conj X_{H-k}): the forward FFT of
  // it is the conjugate of drfftb's (x[2j], x[2j+1])
  template <class PL>
  __device__ __forceinline__ void build(cplx *A, const double2 *__restrict__ twid, double2 wl, int lane, int wv) const {
#pragma unroll
    for (int it = 0; it < NITC; ++it) {
      const int c = wv + NW * it;
      if (c < NCH) {
        const int k = 63 * c + lane;
        const double2 ws = twid[63 * c];
        const cplx w = (c == 0) ? cplx{wl.x, wl.y} : cmul(cplx{wl.x, wl.y}, cplx{ws.x, ws.y});
        const double re_k = row1_dpp<ROW1_FROM_PREV>(p[it].y), im_m = row1_dpp<ROW1_FROM_PREV>(q[it].x);
        const double im_k = p[it].x, re_m = q[it].y;
        if (k == 0) { // X_0 = P_0.x, X_H = Q_0.y
          A[PL::pos_in(0)] = {im_k + re_m, -(im_k - re_m)};
        } else if (lane > 0 && k <= H / 2) {
          const cplx a = {re_k + re_m, im_k - im_m}, b = {re_k - re_m, im_k + im_m};
          const cplx u = {w.x * b.x + w.y * b.y, w.x * b.y - w.y * b.x}; // conj(W^k) * b
          A[PL::pos_in(k)] = {a.x - u.y, -(a.y + u.x)};
          if (k < H / 2) A[PL::pos_in(H - k)] = {a.x + u.y, a.y - u.x};
        }
      }
    }
  }
};

#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// Inverse rows of a cyclic ocean FUSED with the rest of ocinvq and the zonal-boundary part of ocqbdy
// (k_rfft_cyc<true> + part B of the constraint algebra + k_unpack_cyc<NL, true> in one launch).
// A workgroup = NL groups of NT threads = the NL modes of one row, each group with its own LDS buffer (NL * 39 KB: one
// workgroup per CU); after the transforms all threads combine the modes point by point with the expressions of
// k_unpack_cyc (contraction off) straight from LDS - the transformed field never goes to HBM - and the new po leaves
// in 16-byte write-through stores.  The workgroups are PERSISTENT (grid = one per CU): each takes rows
// blockIdx.x, blockIdx.x + gridDim.x, .. and requests the next row's spectrum before it transforms the current one,
// so that only the first row's loads are exposed.  The rows 2 and nyg-1 bring the zonal boundary rows and their PV.
// c1, c2, c3 (part B: a few dozen operations on the k = 0 column sums of k_thomas) are formed by every thread;
// workgroup 0 records the scalars.  Whole-domain handles only.
// (A one-row workgroup that ran the modes one after the other - 39 KB of LDS, three per CU - was measured at 72 us
// for SOcn 5 km against 55 us for the separate launches: 2.25 rows per CU leave every CU latency-bound.)
// ---------------------------------------------------------------------------
template <class PL, int NT, int NL, bool BDY>
__global__ __launch_bounds__(NT * NL) void k_rfft1_unpack(const QgDstParams P, const QgUnpackParams U, const QgBdyParams B,
                                                          const QgCycConstrParams *Qp) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef Row1Fft<PL, NT> F;
  constexpr int H = PL::N, N = 2 * H, NTA = NT * NL, NIT = (H + NTA - 1) / NTA;
  const int tid = threadIdx.x;
  const int md = __builtin_amdgcn_readfirstlane(tid / NT); // this thread's mode (wave-uniform)
  const int t = tid - md * NT;
  cplx *A0 = reinterpret_cast<cplx *>(smem_raw);
  cplx *A = A0 + md * PL::LDS_CPLX;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ny = U.g.ny, nrows = P.g.jr1 - P.g.jr0 + 1;
  const typename F::Tw tw = F::prefetch(P.twid, t);
  const double2 wl = P.twid[lane];
  const double *wm = P.wrk + P.g.wstride * md;
  Row1Spec<H, NT> sp;
  int r = blockIdx.x;
  if (r < nrows) sp.load(wm + (long)(P.g.jr0 + r - 1) * P.g.ldw, lane, wv);
  // part B of the constraint algebra
  double c1[NL], c2[NL], c3;
  {
    double ocs[NL], ocn[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      ocs[k] = Qp->sc->ocncs[k];
      ocn[k] = Qp->sc->ocncn[k];
    }
    constr_cyc_partB<NL>(*Qp, tid, blockIdx.x == 0, ocs, ocn, c1, c2, c3);
  }
  auto layers = [&](const double *pm, double *pl) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; ++m) v = v + U.ctm2l[m + NL * k] * pm[m];
      pl[k] = v;
    }
  };
  // boundary PV of a point of the zonal boundary row (k_unpack_cyc / k_ocqbdy): pw on the boundary, pin next to it
  auto bdy_q = [&](const double *pw, const double *pin, double by, double dd, double *q) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double ap;
      if (k == 0) ap = B.f0A[0] * pw[0] + B.f0A[NL] * pw[1];
      else if (k == NL - 1) ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k];
      else ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k] + B.f0A[k + NL * (k + 1)] * pw[k + 1];
      q[k] = B.bcfaco_f0 * (pin[k] - pw[k]) - ap + by;
      if (k == NL - 1) q[k] = q[k] + dd;
    }
  };
  const long fs = U.g.fstride;
  QG_STAMP(2, 0);
  int qg_it = 0;
  for (; r < nrows; r += gridDim.x, ++qg_it) {
    const int gj = P.g.jr0 + r; // 1-based local row = global row (whole domain)
    // (thread indices made opaque per row: hoisted out of this loop, the LDS addresses of all stages - loop invariants -
    //  cost ~100 VGPRs)
    int tl = t, ll = lane, tg = tid;
    asm volatile("" : "+v"(tl), "+v"(ll), "+v"(tg));
    double2 wll = wl;
    asm volatile("" : "+v"(wll.x), "+v"(wll.y));
    typename F::Tw twl = tw; // (likewise the twiddle bases: their powers, 44 doubles per thread, are formed per row)
#pragma unroll
    for (int i = 0; i < F::N1; ++i) asm volatile("" : "+v"(twl.w1[i].x), "+v"(twl.w1[i].y));
#pragma unroll
    for (int i = 0; i < F::N2; ++i) asm volatile("" : "+v"(twl.w2[i].x), "+v"(twl.w2[i].y));
    sp.template build<PL>(A, P.twid, wll, ll, wv);
    if (r + (int)gridDim.x < nrows) sp.load(wm + (long)(gj + (int)gridDim.x - 1) * P.g.ldw, lane, wv); // flies during the stages
    __syncthreads();
    if (qg_it < 3) QG_STAMP(2, 1 + 3 * qg_it);
    F::run(A, twl, tl);
    if (qg_it < 3) QG_STAMP(2, 2 + 3 * qg_it);
    // homogeneous corrections of this row (and of the adjacent zonal boundary row), modes -> layers
    const int wall = (gj == 2) ? 1 : (gj == ny - 1 ? ny : 0);
    double hom[NL], homw[NL];
    hom[0] = c3 * U.pbh[gj - 1];
    homw[0] = wall ? c3 * U.pbh[wall - 1] : 0.0;
#pragma unroll
    for (int m = 1; m < NL; ++m) {
      hom[m] = c1[m - 1] * U.pch1[(gj - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(gj - 1) + (long)ny * (m - 1)];
      homw[m] = wall ? c1[m - 1] * U.pch1[(wall - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(wall - 1) + (long)ny * (m - 1)] : 0.0;
    }
    double pw[NL];
    {
      double pmw[NL];
#pragma unroll
      for (int m = 0; m < NL; ++m) pmw[m] = 0.0 + homw[m];
      layers(pmw, pw);
    }
    const double by = (BDY && wall) ? B.beta * B.yporel[wall - 1] : 0.0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int jj = tg + it * NTA;
      if (jj >= H) break;
      const long o = (long)(gj - 1) * U.g.ldx + 2 * jj; // stored columns 2jj, 2jj+1 (gi = 2jj+1, 2jj+2)
      double pma[NL], pmb[NL], pla[NL], plb[NL];
#pragma unroll
      for (int m = 0; m < NL; ++m) {
        const cplx z = A0[m * PL::LDS_CPLX + PL::pos_out(jj)]; // conj of (x[2jj], x[2jj+1])
        pma[m] = z.x + hom[m];
        pmb[m] = -z.y + hom[m];
      }
      layers(pma, pla);
      layers(pmb, plb);
#pragma unroll
      for (int k = 0; k < NL; ++k) qg_store16_wt(U.pnew + fs * k + o, pla[k], plb[k]);
      if (jj == 0) { // column nx is column 1
#pragma unroll
        for (int k = 0; k < NL; ++k) U.pnew[fs * k + (long)(gj - 1) * U.g.ldx + N] = pla[k];
      }
      if (wall) {
        const long ow = (long)(wall - 1) * U.g.ldx + 2 * jj;
#pragma unroll
        for (int k = 0; k < NL; ++k) qg_store16_wt(U.pnew + fs * k + ow, pw[k], pw[k]);
        if (jj == 0) {
#pragma unroll
          for (int k = 0; k < NL; ++k) U.pnew[fs * k + (long)(wall - 1) * U.g.ldx + N] = pw[k];
        }
        if (BDY) {
          const double2 dd = *reinterpret_cast<const double2 *>(B.ddynoc + ow);
          double qa[NL], qb[NL];
          bdy_q(pw, pla, by, dd.x, qa);
          bdy_q(pw, plb, by, dd.y, qb);
#pragma unroll
          for (int k = 0; k < NL; ++k) qg_store16_wt(B.qo + fs * k + ow, qa[k], qb[k]);
          if (jj == 0) {
            double qn[NL];
            bdy_q(pw, pla, by, B.ddynoc[(long)(wall - 1) * U.g.ldx + N], qn);
#pragma unroll
            for (int k = 0; k < NL; ++k) B.qo[fs * k + (long)(wall - 1) * U.g.ldx + N] = qn[k];
          }
        }
      }
    }
    if (qg_it < 3) QG_STAMP(2, 3 + 3 * qg_it);
    __syncthreads(); // every thread is done with the LDS buffers before the next row is built into them
  }
  QG_STAMP_DRAIN();
}

// ---------------------------------------------------------------------------
// The same fused step with the modes of a row run ONE AFTER THE OTHER by one group of NT threads through one LDS
// buffer (39-41 KB: three workgroups per CU, all rows of SOcn 5 km resident at once, out of phase with each other).
// The transformed rows of the first NL-1 modes are parked in place in wrk (write-through stores; every thread reads
// back exactly what it wrote, from its XCD's L2), the last mode stays in registers; the next mode's spectrum is
// requested before the stages of the current one.  grid: (nrows).
// ---------------------------------------------------------------------------
template <class PL, int NT, int NL, bool BDY>
__global__ __launch_bounds__(NT, 3) void k_rfft1_unpack_seq(const QgDstParams P, const QgUnpackParams U, const QgBdyParams B,
                                                            const QgCycConstrParams *Qp) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  typedef Row1Fft<PL, NT> F;
  constexpr int H = PL::N, N = 2 * H, NIT = (H + NT - 1) / NT;
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gj = P.g.jr0 + blockIdx.x; // 1-based local row = global row (whole domain)
  const int ny = U.g.ny;
  const typename F::Tw tw = F::prefetch(P.twid, tid);
  const double2 wl = P.twid[lane];
  const long rowoff = (long)(gj - 1) * P.g.ldw;
  Row1Spec<H, NT> sp;
  sp.load(P.wrk + rowoff, lane, wv);
  double xr[NIT][2];
#pragma unroll
  for (int m = 0; m < NL; ++m) {
    // (thread indices and twiddle bases made opaque per mode: as common subexpressions of the NL unrolled copies the
    //  LDS addresses and the twiddle powers of all stages - 44 doubles per thread - stay live and spill)
    int tl = tid, ll = lane;
    asm volatile("" : "+v"(tl), "+v"(ll));
    double2 wll = wl;
    asm volatile("" : "+v"(wll.x), "+v"(wll.y));
    typename F::Tw twl = tw;
#pragma unroll
    for (int i = 0; i < F::N1; ++i) asm volatile("" : "+v"(twl.w1[i].x), "+v"(twl.w1[i].y));
#pragma unroll
    for (int i = 0; i < F::N2; ++i) asm volatile("" : "+v"(twl.w2[i].x), "+v"(twl.w2[i].y));
    sp.template build<PL>(A, P.twid, wll, ll, wv);
    if (m + 1 < NL) sp.load(P.wrk + P.g.wstride * (m + 1) + rowoff, lane, wv); // flies during the stages
    __syncthreads();
    F::run(A, twl, tl);
    double *rowm = P.wrk + P.g.wstride * m + rowoff;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int jj = tl + it * NT;
      const cplx z = A[PL::pos_out(jj < H ? jj : 0)];
      if (m + 1 < NL) {
        if (jj < H) qg_store16_wt(rowm + 2 * jj, z.x, -z.y);
      } else {
        xr[it][0] = z.x;
        xr[it][1] = -z.y;
      }
    }
    if (m + 1 < NL) __syncthreads();
  }
  // part B of the constraint algebra
  double c1[NL], c2[NL], c3;
  {
    double ocs[NL], ocn[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      ocs[k] = Qp->sc->ocncs[k];
      ocn[k] = Qp->sc->ocncn[k];
    }
    constr_cyc_partB<NL>(*Qp, tid, blockIdx.x == 0, ocs, ocn, c1, c2, c3);
  }
  const int wall = (gj == 2) ? 1 : (gj == ny - 1 ? ny : 0);
  double hom[NL], homw[NL];
  hom[0] = c3 * U.pbh[gj - 1];
  homw[0] = wall ? c3 * U.pbh[wall - 1] : 0.0;
#pragma unroll
  for (int m = 1; m < NL; ++m) {
    hom[m] = c1[m - 1] * U.pch1[(gj - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(gj - 1) + (long)ny * (m - 1)];
    homw[m] = wall ? c1[m - 1] * U.pch1[(wall - 1) + (long)ny * (m - 1)] + c2[m - 1] * U.pch2[(wall - 1) + (long)ny * (m - 1)] : 0.0;
  }
  auto layers = [&](const double *pm, double *pl) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; ++m) v = v + U.ctm2l[m + NL * k] * pm[m];
      pl[k] = v;
    }
  };
  auto bdy_q = [&](const double *pw, const double *pin, double by, double dd, double *q) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double ap;
      if (k == 0) ap = B.f0A[0] * pw[0] + B.f0A[NL] * pw[1];
      else if (k == NL - 1) ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k];
      else ap = B.f0A[k + NL * (k - 1)] * pw[k - 1] + B.f0A[k + NL * k] * pw[k] + B.f0A[k + NL * (k + 1)] * pw[k + 1];
      q[k] = B.bcfaco_f0 * (pin[k] - pw[k]) - ap + by;
      if (k == NL - 1) q[k] = q[k] + dd;
    }
  };
  const long fs = U.g.fstride;
  double pw[NL];
  {
    double pmw[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m) pmw[m] = 0.0 + homw[m];
    layers(pmw, pw);
  }
  const double by = (BDY && wall) ? B.beta * B.yporel[wall - 1] : 0.0;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int jj = tid + it * NT;
    if (jj >= H) break;
    if (it % 3 == 0) asm volatile("" ::: "memory"); // (three column pairs' loads in flight at a time: registers)
    const long o = (long)(gj - 1) * U.g.ldx + 2 * jj; // stored columns 2jj, 2jj+1 (gi = 2jj+1, 2jj+2)
    double pma[NL], pmb[NL], pla[NL], plb[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      double2 x = {xr[it][0], xr[it][1]};
      if (m + 1 < NL) x = *reinterpret_cast<const double2 *>(P.wrk + P.g.wstride * m + rowoff + 2 * jj);
      pma[m] = x.x + hom[m];
      pmb[m] = x.y + hom[m];
    }
    layers(pma, pla);
    layers(pmb, plb);
#pragma unroll
    for (int k = 0; k < NL; ++k) qg_store16_wt(U.pnew + fs * k + o, pla[k], plb[k]);
    if (jj == 0) { // column nx is column 1
#pragma unroll
      for (int k = 0; k < NL; ++k) U.pnew[fs * k + (long)(gj - 1) * U.g.ldx + N] = pla[k];
    }
    if (wall) {
      const long ow = (long)(wall - 1) * U.g.ldx + 2 * jj;
#pragma unroll
      for (int k = 0; k < NL; ++k) qg_store16_wt(U.pnew + fs * k + ow, pw[k], pw[k]);
      if (jj == 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) U.pnew[fs * k + (long)(wall - 1) * U.g.ldx + N] = pw[k];
      }
      if (BDY) {
        const double2 dd = *reinterpret_cast<const double2 *>(B.ddynoc + ow);
        double qa[NL], qb[NL];
        bdy_q(pw, pla, by, dd.x, qa);
        bdy_q(pw, plb, by, dd.y, qb);
#pragma unroll
        for (int k = 0; k < NL; ++k) qg_store16_wt(B.qo + fs * k + ow, qa[k], qb[k]);
        if (jj == 0) {
          double qn[NL];
          bdy_q(pw, pla, by, B.ddynoc[(long)(wall - 1) * U.g.ldx + N], qn);
#pragma unroll
          for (int k = 0; k < NL; ++k) B.qo[fs * k + (long)(wall - 1) * U.g.ldx + N] = qn[k];
        }
      }
    }
  }
}
