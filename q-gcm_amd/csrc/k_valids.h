// k_valids.h - validity scan on the device.
//
// Replaces the ocean part of `call valids (solnok)` (src/q-gcm.F:1278; body src/valsubs.F:272-527): extremes of
// po, qo, sst, wekto and of the full layer thicknesses (top / intermediate / bottom), the percentage of the basin
// where a layer is thinner than thkmin, and the verdict.  On the host this scan needs po, qo (44 MB at 5 km)
// every valday; here 20 doubles come back.  min / max and the weighted counts (multiples of 1/4) are exact in
// any order, so the result is bitwise the reference's.  The neighbourhood print-out of a failing run
// (scan2D / scan3D) stays with the host: after a negative verdict it pulls the state and calls the reference's
// own routine.
//   k_valids_scan   grid-stride over the p points (all layers) and the T points; per-workgroup partials
//   k_valids_final  one workgroup: reduces the partials, applies the criteria of src/valsubs.F:78-97
#pragma once
#include "qgcm_dev.h"

#define VAL_NT 256
#define VAL_NB 512          // workgroups of the scan
#define VAL_NMM 7           // po, qo, sst, wekto, hf top / intermediate / bottom

struct QgValidsParams {
  QgGeom g;
  const double *po, *qo;   // (ldx, ny, nl)
  const double *sst, *wekto; // T grid (ldt pitch) or nullptr (mixed layer not initialised)
  const double *dtopoc;    // (ldx, ny) or nullptr (flat)
  int ldt;
  double rgpoc[QG_MAXL], hoc[QG_MAXL];
  double *part;            // (2*VAL_NMM + QG_MAXL, VAL_NB) partial min / max / thin-point weights
  double *out;             // 14 + nl results, then solnok as a double
  double ocnorm;
};

// src/valsubs.F:78-82, 96-97
#define VAL_BIGNUM 1.0e30
#define VAL_WTOEXT 1.0e-3
#define VAL_SSTEXT 75.0
#define VAL_POCEXT 1.0e4
#define VAL_QOCEXT 0.05
#define VAL_THKMIN 100.0
#define VAL_CRITPC 20.0

template <int NL>
__global__ __launch_bounds__(VAL_NT) void k_valids_scan(const QgValidsParams P) {
  __shared__ double red[VAL_NT];
  const int tid = threadIdx.x;
  const int nx = P.g.nx, ny = P.g.ny, ldx = P.g.ldx;
  const long fs = P.g.fstride;
  double mn[VAL_NMM], mx[VAL_NMM], bad[NL];
#pragma unroll
  for (int q = 0; q < VAL_NMM; ++q) { mn[q] = VAL_BIGNUM; mx[q] = -VAL_BIGNUM; }
#pragma unroll
  for (int k = 0; k < NL; ++k) bad[k] = 0.0;
#define MM(q, v) do { const double v_ = (v); if (v_ < mn[q]) mn[q] = v_; if (v_ > mx[q]) mx[q] = v_; } while (0)
  const long npts = (long)nx * ny;
  for (long t = (long)blockIdx.x * VAL_NT + tid; t < npts; t += (long)VAL_NB * VAL_NT) {
    const int i = (int)(t % nx) + 1, j = (int)(t / nx) + 1;
    const long o = (long)(j - 1) * ldx + (i - 1);
    double p[NL], eta[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      p[k] = P.po[fs * k + o];
      MM(0, p[k]);
      MM(1, P.qo[fs * k + o]);
    }
    const double w = ((i == 1 || i == nx) ? 0.5 : 1.0) * ((j == 1 || j == ny) ? 0.5 : 1.0);
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) eta[k] = P.rgpoc[k] * (p[k + 1] - p[k]); // :409-411
    double hf = P.hoc[0] - eta[0];
    MM(4, hf);
    if (hf < VAL_THKMIN) bad[0] += w;
#pragma unroll
    for (int k = 1; k < NL - 1; ++k) {
      hf = P.hoc[k] - eta[k] + eta[k - 1];
      MM(5, hf);
      if (hf < VAL_THKMIN) bad[k] += w;
    }
    hf = P.hoc[NL - 1] + eta[NL - 2] - (P.dtopoc ? P.dtopoc[o] : 0.0);
    MM(6, hf);
    if (hf < VAL_THKMIN) bad[NL - 1] += w;
  }
  if (P.sst) {
    const int nxt = P.g.nxt, nyt = ny - 1;
    const long nT = (long)nxt * nyt;
    for (long t = (long)blockIdx.x * VAL_NT + tid; t < nT; t += (long)VAL_NB * VAL_NT) {
      const long o = (t / nxt) * P.ldt + (t % nxt);
      MM(2, P.sst[o]);
      MM(3, P.wekto[o]);
    }
  }
#undef MM
  // workgroup reduction (order-independent operations)
  auto bmin = [&](double v) {
    red[tid] = v;
    __syncthreads();
    for (int off = VAL_NT / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] = red[tid + off] < red[tid] ? red[tid + off] : red[tid];
      __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
  };
  auto bsum = [&](double v) {
    red[tid] = v;
    __syncthreads();
    for (int off = VAL_NT / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
  };
  const int b = blockIdx.x;
#pragma unroll
  for (int q = 0; q < VAL_NMM; ++q) {
    const double a = bmin(mn[q]), c = -bmin(-mx[q]);
    if (tid == 0) {
      P.part[(2 * q) * VAL_NB + b] = a;
      P.part[(2 * q + 1) * VAL_NB + b] = c;
    }
  }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const double s = bsum(bad[k]); // multiples of 1/4: exact
    if (tid == 0) P.part[(2 * VAL_NMM + k) * VAL_NB + b] = s;
  }
}

template <int NL>
__global__ __launch_bounds__(VAL_NT) void k_valids_final(const QgValidsParams P) {
  __shared__ double red[VAL_NT];
  const int tid = threadIdx.x;
  double res[2 * VAL_NMM + NL];
  for (int q = 0; q < 2 * VAL_NMM + NL; ++q) {
    const bool ismin = q < 2 * VAL_NMM && (q % 2 == 0), ismax = q < 2 * VAL_NMM && (q % 2 == 1);
    double v = ismin ? VAL_BIGNUM : (ismax ? -VAL_BIGNUM : 0.0);
    for (int b = tid; b < VAL_NB; b += VAL_NT) {
      const double x = P.part[q * VAL_NB + b];
      if (ismin) v = x < v ? x : v;
      else if (ismax) v = x > v ? x : v;
      else v += x;
    }
    red[tid] = v;
    __syncthreads();
    for (int off = VAL_NT / 2; off > 0; off >>= 1) {
      if (tid < off) {
        const double x = red[tid + off];
        if (ismin) red[tid] = x < red[tid] ? x : red[tid];
        else if (ismax) red[tid] = x > red[tid] ? x : red[tid];
        else red[tid] += x;
      }
      __syncthreads();
    }
    res[q] = red[0];
    __syncthreads();
  }
  if (tid != 0) return;
  // src/valsubs.F:433-436, 439-486: the fractions are only evaluated when some layer is at or below thkmin
  double hfmina = res[8] < res[10] ? res[8] : res[10];
  hfmina = res[12] < hfmina ? res[12] : hfmina;
  const bool hffail = hfmina <= VAL_THKMIN;
  bool ok = true;
  if (fabs(res[0]) >= VAL_POCEXT || fabs(res[1]) >= VAL_POCEXT) ok = false; // :312
  if (fabs(res[2]) >= VAL_QOCEXT || fabs(res[3]) >= VAL_QOCEXT) ok = false; // :327
  if (P.sst) {
    if (fabs(res[4]) >= VAL_SSTEXT || fabs(res[5]) >= VAL_SSTEXT) ok = false; // :342
    if (fabs(res[6]) >= VAL_WTOEXT || fabs(res[7]) >= VAL_WTOEXT) ok = false; // :357
  }
  for (int q = 0; q < 2 * VAL_NMM; ++q) P.out[q] = res[q];
  for (int k = 0; k < NL; ++k) {
    const double pc = hffail ? 100.0 * res[2 * VAL_NMM + k] * P.ocnorm : 0.0; // :482-485
    P.out[2 * VAL_NMM + k] = pc;
    if (pc > VAL_CRITPC) ok = false; // spfail = .false., :507-512
  }
  P.out[2 * VAL_NMM + NL] = ok ? 1.0 : 0.0;
}
