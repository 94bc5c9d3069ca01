// k_thomas.h - K4: batched constant-coefficient tridiagonal solves along y.
//
// Reference: for every wavenumber i the system
//     aoc*u(j-1) + boc(i)*u(j) + aoc*u(j+1) = rhs(i,j),  j = 2..nypo-1
// is solved by the Thomas algorithm (src/ocisubs.F:470-488 box, 575-593
// cyclic) and scaled by ftnorm.  The pivots betinv(j) = 1/(boc - aoc*gam(j))
// depend only on (boc, j).  The host tabulates the pivot entering each chunk
// (same recurrence and rounding as the reference); every thread re-runs the
// recurrence for its R rows (IEEE divides, identical values) while its row
// loads are in flight, so no full-size pivot table is streamed.
//
// Parallel formulation: both sweeps are first-order linear recurrences
//     forward : u_r = (w_r - aoc*u_{r-1}) * bet_r
//     backward: v_r = u_r - aoc*bet_r * v_{r+1}
// The rows are cut into 64 chunks of R rows; lanes hold 16 consecutive
// wavenumbers (one 128-B line) x 4 chunks per wave, a workgroup of 1024
// threads covers a whole column block.  Each thread keeps its R rows in
// registers, computes the chunk's affine map (C, D) with zero inflow, the
// maps are composed through LDS, and the sweep is re-run from the true
// inflow - so each row is read once and written once.
//
// Algorithmic traffic: read w + write u (16 B per point).
#pragma once
#include "qgcm_dev.h"

#define TH_KW 16   // wavenumbers per workgroup (128-B line)
#define TH_NC 64   // chunks per column
#define TH_NT (TH_KW * TH_NC)
static_assert(TH_NC == 64 && TH_KW == 16, "the chunk scan maps 64 chunks to the lanes of 16 waves");

// grid: (ceil(nk/16), nlayers)
template <int R>
__global__ __launch_bounds__(TH_NT) void k_thomas(const QgThomasParams P) {
  __shared__ double sC[TH_NC][TH_KW];
  __shared__ double sD[TH_NC][TH_KW];
  __shared__ double sIn[TH_NC][TH_KW];
  const int tid = threadIdx.x;
  const int kk = tid % TH_KW;
  const int c = tid / TH_KW;
  const int k = blockIdx.x * TH_KW + kk;
  const int m = blockIdx.y;
  const int nr = P.g.ny - 2; // rows j=2..ny-1  <->  r = 0..nr-1
  const int ldw = P.g.ldw;
  const bool kok = k < P.g.nk;
  const double a = P.aoc;
  double *wcol = P.wrk + P.g.wstride * m + (long)ldw + k;      // row j=2
  const int r0 = c * R;

  double w[R], b[R];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    int r = r0 + t;
    bool ok = kok && r < nr;
    w[t] = ok ? wcol[(long)r * ldw] : 0.0;
  }
  // pivots of this chunk: betc = betinv of the row before the chunk (src/ocisubs.F:472-477)
  {
    const double boc = kok ? P.boc[(long)m * ldw + k] : 1.0;
    double betinv = kok ? P.betc[((long)m * TH_NC + c) * ldw + k] : 0.0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      int r = r0 + t;
      if (r == 0) {
        betinv = 1.0 / boc;
      } else {
        double gam = a * betinv;
        betinv = 1.0 / (boc - a * gam);
      }
      b[t] = (kok && r < nr) ? betinv : 0.0;
    }
  }
  // ---- forward: local affine map (zero inflow) ---------------------------
  double C = 0.0, D = 1.0;
#pragma unroll
  for (int t = 0; t < R; ++t) {
    C = (w[t] - a * C) * b[t];
    D = -a * b[t] * D;
  }
  sC[c][kk] = C;
  sD[c][kk] = D;
  __syncthreads();
  // Compose the 64 chunk maps of one wavenumber with a wave scan: wave w takes
  // wavenumber kk = w, lane = chunk.  After the inclusive scan Cs is the value
  // leaving each chunk (the inflow to chunk 0 is zero), so the inflow of chunk c
  // is the scanned value of chunk c-1.
  const int lane = tid & 63, wv = tid >> 6;
  {
    double Cs = sC[lane][wv], Ds = sD[lane][wv];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      double Cp = __shfl_up(Cs, off), Dp = __shfl_up(Ds, off);
      if (lane >= off) {
        Cs = Cs + Ds * Cp;
        Ds = Ds * Dp;
      }
    }
    double inflow = __shfl_up(Cs, 1);
    sIn[lane][wv] = (lane == 0) ? 0.0 : inflow;
  }
  __syncthreads();
  double u = sIn[c][kk];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    u = (w[t] - a * u) * b[t];
    w[t] = u;
  }
  // ---- backward: v_r = u_r - a*bet_r*v_{r+1} ------------------------------
  C = 0.0;
#pragma unroll
  for (int t = R - 1; t >= 0; --t) C = w[t] - a * b[t] * C;
  __syncthreads();
  sC[c][kk] = C; // D is the same product as in the forward sweep
  __syncthreads();
  {
    // same scan in the opposite direction: lane l stands for chunk 63-l
    const int cr = 63 - lane;
    double Cs = sC[cr][wv], Ds = sD[cr][wv];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      double Cp = __shfl_up(Cs, off), Dp = __shfl_up(Ds, off);
      if (lane >= off) {
        Cs = Cs + Ds * Cp;
        Ds = Ds * Dp;
      }
    }
    double inflow = __shfl_up(Cs, 1);
    sIn[cr][wv] = (lane == 0) ? 0.0 : inflow;
  }
  __syncthreads();
  double v = sIn[c][kk];
  const double ft = P.ftnorm;
#pragma unroll
  for (int t = R - 1; t >= 0; --t) {
    v = w[t] - a * b[t] * v;
    w[t] = v;
  }
#pragma unroll
  for (int t = 0; t < R; ++t) {
    int r = r0 + t;
    if (kok && r < nr) wcol[(long)r * ldw] = ft * w[t];
  }
}
