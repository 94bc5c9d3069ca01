// k_rfft3.h - long cyclic rows (nxto = 4608, 4800, ...): k_rfft_cyc's three-stage plans with the packing / split steps
// FOLDED INTO the first and last stage.
//
// Round 3 measured the long-row kernels fp64-VALU-issue bound: 2300-3500 instructions per thread and
// row pair, of which a third was index arithmetic (pos_in / pos_out: two divisions by constants per LDS position) and
// the LDS passes of a separate pre and post step.  Here
//   * the thread that owns stage-1 butterfly r takes its R1 inputs n = n1*R2*R3 + r STRAIGHT FROM GLOBAL MEMORY (the
//     rows, or - inverse - the half-complex spectra turned into conj(Z) on the fly): no staging pass, no barrier,
//     no position arithmetic (the LDS slot of (n1, r) is n1*BLOCK + p0(r));
//   * inverse: the thread that owns stage-3 row id = k1 + R1*k2 holds X[id + R1*R2*k3], k3 = 0..R3-1, in registers
//     after its butterfly - consecutive lanes = consecutive row elements - and stores them from there (lane pairs
//     swap one value so that each lane writes one 16-byte pair);
//   * forward: the split step walks the spectrum in the same order, t = id + R1*R2*k3: every LDS position is a
//     per-thread base plus a compile-time offset.
// LDS passes per row pair: 6 (forward) / 4 (inverse) instead of 8; barriers 3 / 2 instead of 4.
// Reference: src/ocisubs.F:566-568, 601-605 (drfftf / drfftb in hscyoc); half-complex order as in k_dst.h.
#pragma once
#include "k_fft3.h"

#pragma clang fp contract(fast)

template <class PL>
struct Rfft3 {
  static constexpr int R1 = PL::RA, R2 = PL::RB, R3 = PL::RC;
  static constexpr int N = PL::N, H = N / 2, I1 = R2 * R3, I3 = R1 * R2, BLOCK = PL::BLOCK, PITCH = PL::PITCH;
  static_assert(R3 % 2 == 0, "the split step takes k3 < R3/2");
  static_assert(I3 % 2 == 0, "lane pairs store aligned element pairs");

  // stage 1 of butterfly r on inputs already in registers (x[n1] = element n1*I1 + r)
  static __device__ __forceinline__ void stage1_regs(cplx *A, int r, double2 w, cplx *x) {
    cplx tw[R1];
    const int p0 = r + r / R3;
    Fft3Dft<R1>::run(x);
    fft3_powers<R1>(cplx{w.x, w.y}, tw);
    A[p0] = x[0];
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) A[k1 * BLOCK + p0] = cmul(x[k1], tw[k1]);
  }
  // stage 3 of row id, outputs left in registers: x[k3] = X[id + I3*k3]
  static __device__ __forceinline__ void stage3_regs(const cplx *A, int id, cplx *x) {
    const cplx *base = A + (id % R1) * BLOCK + (id / R1) * PITCH;
#pragma unroll
    for (int n3 = 0; n3 < R3; ++n3) x[n3] = base[n3];
    Fft3Dft<R3>::run(x);
  }
  // LDS slot of X[id + I3*k3]
  static __device__ __forceinline__ int base_of(int id) { return (id % R1) * BLOCK + (id / R1) * PITCH; }
};

// defined in k_cyclic.h
__device__ __forceinline__ void rfft_cyc_constr_partB(const struct QgCycConstrParams *Q, int lane);

// grid: (ceil(nrows/2) [+1: part B of the constraint algebra, inverse launch inside qgcm_hip_steps], nlayers);
// 256 threads; dynamic LDS: PL::LDS_CPLX complex numbers
template <bool INV, class PL>
__global__ __launch_bounds__(256) void k_rfft3_cyc(const QgDstParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int NT = 256;
  typedef Rfft3<PL> F;
  constexpr int R1 = F::R1, R3 = F::R3, N = F::N, H = F::H, I1 = F::I1, I3 = F::I3;
  static_assert(I3 <= NT, "one stage-3 row per thread");
  if (INV && P.cycq && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y == 0 && threadIdx.x < 64) rfft_cyc_constr_partB(P.cycq, threadIdx.x);
    return;
  }
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  const int tid = threadIdx.x;
  const int ldw = P.g.ldw;
  const int m = blockIdx.y + P.layer0;
  const int ja = P.g.jr0 + 2 * blockIdx.x;
  const bool has_b = (ja + 1 <= P.g.jr1);
  double *rowa = P.wrk + P.g.wstride * m + (long)(ja - 1) * ldw;
  double *rowb = rowa + ldw;
  const double *rb = has_b ? rowb : rowa; // the odd last row has no partner: loads redirected, values zeroed
  const double bs = has_b ? 1.0 : 0.0;
  const typename PL::Tw tw3 = PL::template prefetch<NT>(P.twid, tid);

  // ---- stage 1 straight from global memory ----------------------------------------------------------------------
  auto first = [&](int r, double2 w) {
    cplx x[R1];
    if (!INV) {
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) {
        const int n = n1 * I1 + r;
        x[n1] = {rowa[n], bs * rb[n]};
      }
    } else {
      // element n of conj(Z), Z_k = Xa_k + i Xb_k on the Hermitian-extended spectra: coefficient k = min(n, N - n);
      // half-complex order r(1) = X_0, r(2k) = Re X_k, r(2k+1) = Im X_k, r(N) = X_{N/2}
      double ar[R1], ai[R1], br[R1], bi[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) {
        const int n = n1 * I1 + r;
        const int k = n <= H ? n : N - n;
        const bool edge = (k == 0 || k == H);
        const int i1 = (k == 0) ? 0 : 2 * k - 1, i2 = edge ? i1 : 2 * k;
        ar[n1] = rowa[i1];
        ai[n1] = rowa[i2];
        br[n1] = rb[i1];
        bi[n1] = rb[i2];
      }
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) {
        const int n = n1 * I1 + r;
        const int k = n <= H ? n : N - n;
        const bool edge = (k == 0 || k == H);
        const double a_r = ar[n1], a_i = edge ? 0.0 : ai[n1];
        const double b_r = bs * br[n1], b_i = edge ? 0.0 : bs * bi[n1];
        // k = n: conj(Xa_k + i Xb_k); k = N - n: conj(conj(Xa_k) + i conj(Xb_k))
        x[n1] = (n <= H) ? cplx{a_r - b_i, -(a_i + b_r)} : cplx{a_r + b_i, -(b_r - a_i)};
      }
    }
    F::stage1_regs(A, r, w, x);
  };
  {
    const int e = PL::template extra_item<NT>(tid, 1, I1);
    if (tid < I1) first(tid, tw3.w1);
    if (e >= 0) first(e, tw3.w1x);
  }
  __syncthreads();
  {
    const int e = PL::template extra_item<NT>(tid, 2, R1 * R3);
    if (tid < R1 * R3) PL::stage2(A, tid, tw3.w2);
    if (e >= 0) PL::stage2(A, e, tw3.w2x);
  }
  __syncthreads();
  if (INV) {
    // ---- stage 3, rows written from registers: element j = id + I3*k3 is Re Z_j (row a), -Im Z_j (row b); the lanes
    // of a pair (j even, j + 1) swap one value: the even lane stores row a's pair, the odd lane row b's
    if (tid < I3) {
      cplx x[R3];
      F::stage3_regs(A, tid, x);
      const bool odd = (tid & 1) != 0;
      double *dst = (odd ? rowb : rowa) + (tid & ~1);
#pragma unroll
      for (int k3 = 0; k3 < R3; ++k3) {
        const double va = x[k3].x, vb = -x[k3].y;
        const double give = odd ? va : vb;
        int lo = __double2loint(give), hi = __double2hiint(give);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true);
        const double got = __hiloint2double(hi, lo);
        if (!odd || has_b) qg_store16_wt(dst + I3 * k3, odd ? got : va, odd ? vb : got); // (write-through: qgcm_dev.h)
      }
    }
    return;
  }
  {
    const int e = PL::template extra_item<NT>(tid, 3, I3);
    if (tid < I3) PL::stage3(A, tid);
    if (e >= 0) PL::stage3(A, e);
  }
  __syncthreads();
  // ---- forward split step: Xa_k = (Z_k + conj Z_{N-k})/2, Xb_k = (Z_k - conj Z_{N-k})/(2i); the aligned pair (2t, 2t+1)
  // of the half-complex row is (Im X_t, Re X_{t+1}), (X_0, Re X_1) for t = 0.  t = id + I3*k3, k3 < R3/2: X[t] sits at
  // base(id) + k3, X[N-t] at base(I3-id) + R3-1-k3 (id > 0) / base(0) + (R3-k3) % R3 (id = 0); t + 1 likewise with
  // id + 1 (wrapping into the next k3 at id = I3 - 1)
  if (tid < I3) {
    const int id = tid;
    const int idn = (id + 1 == I3) ? 0 : id + 1, cn = (id + 1 == I3) ? 1 : 0;
    const cplx *z1p = A + F::base_of(id), *z3p = A + F::base_of(idn) + cn;
    const cplx *z2p = A + ((id == 0) ? F::base_of(0) + R3 : F::base_of(I3 - id) + R3 - 1);             // - k3 (mod R3 at id = 0)
    const cplx *z4p = A + ((idn == 0) ? F::base_of(0) + R3 - cn : F::base_of(I3 - idn) + R3 - 1);      // - k3
#pragma unroll
    for (int k3 = 0; k3 < R3 / 2; ++k3) {
      const int t = id + I3 * k3;
      const cplx z1 = z1p[k3], z3 = z3p[k3];
      const cplx z2 = (id == 0 && k3 == 0) ? z1p[0] : z2p[-k3];
      const cplx z4 = z4p[-k3];
      const double ar = 0.5 * (z1.x + z2.x), ai = 0.5 * (z1.y - z2.y);
      const double br = 0.5 * (z1.y + z2.y), bi = -0.5 * (z1.x - z2.x);
      const double arn = 0.5 * (z3.x + z4.x), brn = 0.5 * (z3.y + z4.y);
      qg_store16_wt(rowa + 2 * t, t == 0 ? ar : ai, arn);
      if (has_b) qg_store16_wt(rowb + 2 * t, t == 0 ? br : bi, brn);
    }
  }
}

#pragma clang fp contract(off)
