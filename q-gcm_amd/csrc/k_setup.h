// k_setup.h - start-up / restart arithmetic and the progress print-out's numbers on the device:
//
//   k_qcomp          interior PV from pressure, qcomp (src/vorsubs.F:49-138) and, zonally cyclic, the meridional
//                    edge columns, merqcy (src/vorsubs.F:142-239); the boundary ring follows with k_ocqbdy.
//                    Same association order, contraction off: bitwise the reference.
//   k_area_partial / k_area_final
//                    trapezoid area integrals (xintp, src/intsubs.f:78-133: weight 1 inside, 1/2 on the edges, 1/4 in
//                    the corners) of every layer of po, pom and qo in one pass, fixed summation order.  They give
//                    constr's dpioc / dpiocp (src/conhoms.F:93-123) and prsamp's layer averages pavgoc / qavgoc
//                    (src/q-gcm.F:2026-2027; src/monitor_diag.F:729-739).
//   k_constr_lines   cyclic: the line integrals of constr that start the momentum-constraint vectors ocncs, ocncn
//                    (+ previous time level), src/conhoms.F:131-193 / 243-300.
//   k_wekpo          ocean-only Ekman pumping from the wind stress: wekto on the T grid and its p-grid average
//                    wekpo (src/xfosubs.F:138, 566-645).
#pragma once
#include "qgcm_dev.h"
#include "k_cyclic.h" // cyc_col

struct QgQcompParams {
  QgGeom g;
  const double *p;
  double *q;
  const double *ddyn, *yporel;
  double dx2fac, beta, fnot; // dxm2/fnot
  double amat[QG_MAXL * QG_MAXL]; // (k,l) at k + nl*l
  int ktopo;                       // 0-based layer that feels the topography (ocean nlo-1, atmosphere 0)
};

// grid: (ceil(nx/256), ny-2, nl): rows j = 2..ny-1; box: columns 2..nx-1; cyclic: all columns (1 and nx via the wrap)
__global__ __launch_bounds__(256) void k_qcomp(const QgQcompParams P) {
  const int nx = P.g.nx, nl = P.g.nl, nxt = P.g.nxt, ldx = P.g.ldx;
  const int gi = blockIdx.x * blockDim.x + threadIdx.x + 1;
  const int gj = blockIdx.y + 2;
  const int k = blockIdx.z;
  if (gi > nx) return;
  if (!P.g.cyc && (gi < 2 || gi > nx - 1)) return;
  const long fs = P.g.fstride;
  const int ic = P.g.cyc ? cyc_col(gi, nxt) : gi;                       // column nx is column 1
  const int iw = P.g.cyc ? cyc_col(ic - 1, nxt) : gi - 1, ie = P.g.cyc ? cyc_col(ic + 1, nxt) : gi + 1;
  const double *pk = P.p + fs * k;
  const long r = (long)(gj - 1) * ldx;
  const double pc = pk[r + ic - 1];
  const double betay = P.beta * P.yporel[gj - 1];
  const double lap = P.dx2fac * (pk[r - ldx + ic - 1] + pk[r + iw - 1] + pk[r + ie - 1] + pk[r + ldx + ic - 1] - 4.0 * pc) + betay;
  double ap;
  if (k == 0) ap = P.amat[0] * pc + P.amat[nl] * P.p[fs + r + ic - 1];
  else if (k == nl - 1) ap = P.amat[k + nl * (k - 1)] * P.p[fs * (k - 1) + r + ic - 1] + P.amat[k + nl * k] * pc;
  else ap = P.amat[k + nl * (k - 1)] * P.p[fs * (k - 1) + r + ic - 1] + P.amat[k + nl * k] * pc + P.amat[k + nl * (k + 1)] * P.p[fs * (k + 1) + r + ic - 1];
  double q = lap - P.fnot * ap;
  if (k == P.ktopo) q = q + P.ddyn[r + ic - 1];
  P.q[fs * k + r + gi - 1] = q;
}

// ---------------------------------------------------------------------------
#define AREA_NB 128
#define AREA_NT 256
struct QgAreaParams {
  QgGeom g;
  const double *f[3]; // po, pom, qo
  double *part;       // (AREA_NB, 3*nl)
  double *out;        // (3*nl): xintp of po(1..nl), pom(1..nl), qo(1..nl)
};

// block b sums rows b, b + AREA_NB, ...; fixed order (thread-strided along the row, wave butterfly, waves left to right)
template <int NL>
__global__ __launch_bounds__(AREA_NT) void k_area_partial(const QgAreaParams P) {
  __shared__ double red[3 * NL][AREA_NT / 64];
  const int nx = P.g.nx, ny = P.g.ny, ldx = P.g.ldx, tid = threadIdx.x;
  double s[3 * NL];
#pragma unroll
  for (int v = 0; v < 3 * NL; ++v) s[v] = 0.0;
  for (int j = 1 + blockIdx.x; j <= ny; j += AREA_NB) {
    const double wy = (j == 1 || j == ny) ? 0.5 : 1.0;
    for (int i = 1 + tid; i <= nx; i += AREA_NT) {
      const double w = wy * ((i == 1 || i == nx) ? 0.5 : 1.0);
      const long o = (long)(j - 1) * ldx + (i - 1);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < NL; ++k) s[a * NL + k] += w * P.f[a][P.g.fstride * k + o];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int v = 0; v < 3 * NL; ++v) s[v] += __shfl_xor(s[v], off);
  if ((tid & 63) == 0)
#pragma unroll
    for (int v = 0; v < 3 * NL; ++v) red[v][tid >> 6] = s[v];
  __syncthreads();
  if (tid < 3 * NL) {
    double t = 0.0;
    for (int w = 0; w < AREA_NT / 64; ++w) t += red[tid][w];
    P.part[(long)blockIdx.x * 3 * NL + tid] = t;
  }
}

template <int NL>
__global__ __launch_bounds__(64) void k_area_final(const QgAreaParams P) {
  const int lane = threadIdx.x;
  if (lane >= 3 * NL) return;
  double t = 0.0;
  for (int b = 0; b < AREA_NB; ++b) t += P.part[(long)b * 3 * NL + lane];
  P.out[lane] = t;
}

// constr on the device: dpioc / dpiocp from the area integrals (src/conhoms.F:93-123; atmosphere: pa(k)-pa(k+1),
// :205-216) and, cyclic, the line integrals that start ocncs / ocncn / ocncsp / ocncnp (:131-193, 243-300).
struct QgConstrInitParams {
  QgGeom g;
  const double *po, *pom, *area; // area: output of k_area_final
  QgScalars *sc;
  double dxo, dyo, fnot;
  double amat[QG_MAXL * QG_MAXL];
};

template <int NL>
__global__ __launch_bounds__(64) void k_constr_init(const QgConstrInitParams P) {
  const int lane = threadIdx.x;
  const int nx = P.g.nx, ny = P.g.ny, ldx = P.g.ldx;
  const long fs = P.g.fstride;
  QgScalars *sc = P.sc;
  const double dA = P.dxo * P.dyo;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
      const double cur = P.area[k + 1] - P.area[k], prev = P.area[NL + k + 1] - P.area[NL + k];
      sc->dpioc[k] = (P.g.atm ? -cur : cur) * dA;
      sc->dpiocp[k] = (P.g.atm ? -prev : prev) * dA;
    }
  }
  if (!P.g.cyc) return;
  // trapezoid line sums along x of p on rows 1, ny and of the one-sided y differences, both time levels
  double pins[2][NL], pinn[2][NL], cs[2][NL], cn[2][NL];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const double *p = t ? P.pom : P.po;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
      const double *pk = p + fs * k;
      for (int i = 1 + lane; i <= nx; i += 64) {
        const double w = (i == 1 || i == nx) ? 0.5 : 1.0;
        const double p1 = pk[i - 1], p2 = pk[ldx + i - 1], pm = pk[(long)(ny - 2) * ldx + i - 1], pn = pk[(long)(ny - 1) * ldx + i - 1];
        a += w * p1;
        b += w * pn;
        c += w * (p2 - p1);
        d += w * (pn - pm);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off); b += __shfl_xor(b, off); c += __shfl_xor(c, off); d += __shfl_xor(d, off);
      }
      pins[t][k] = P.dxo * a;
      pinn[t][k] = P.dxo * b;
      cs[t][k] = c * (P.dxo / P.dyo);
      cn[t][k] = d * (P.dxo / P.dyo);
    }
  }
  if (lane != 0) return;
  const double f = 0.5 * P.dyo * P.fnot * P.fnot;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    double aps[2] = {0.0, 0.0}, apn[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        aps[t] += P.amat[k + NL * l] * pins[t][l];
        apn[t] += P.amat[k + NL * l] * pinn[t][l];
      }
    sc->ocncs[k] = -cs[0][k] + f * aps[0];
    sc->ocncn[k] = cn[0][k] + f * apn[0];
    sc->ocncsp[k] = -cs[1][k] + f * aps[1];
    sc->ocncnp[k] = cn[1][k] + f * apn[1];
  }
}

// ---------------------------------------------------------------------------
// ocean-only Ekman pumping (src/xfosubs.F:138, 566-645): wekto on the T grid from the stress, then its p-grid average
struct QgWekParams {
  QgGeom g;
  const double *taux, *tauy; // p grid, pitch ldx
  double *wekto;             // T grid (nxt, ny-1), pitch ldt
  double *wekpo;             // p grid, pitch ldx
  int ldt;
  double hxofac; // 0.5 / (dxo * fnot)
};

// grid: (ceil(nxt/256), ny-1)
__global__ __launch_bounds__(256) void k_wekto(const QgWekParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
  if (i > P.g.nxt) return;
  const int ldx = P.g.ldx;
  const double *tx = P.taux, *ty = P.tauy;
  const long o = (long)(j - 1) * ldx + (i - 1); // (i, j); (i+1, j) = o+1; (i, j+1) = o+ldx
  P.wekto[(long)(j - 1) * P.ldt + (i - 1)] =
      P.hxofac * (ty[o + ldx + 1] + ty[o + 1] - (ty[o + ldx] + ty[o]) + tx[o + 1] + tx[o] - (tx[o + ldx + 1] + tx[o + ldx]));
}

// grid: (ceil(nx/256), ny)
__global__ __launch_bounds__(256) void k_wekpo(const QgWekParams P) {
  const int io = blockIdx.x * blockDim.x + threadIdx.x + 1, jo = blockIdx.y + 1;
  const int nx = P.g.nx, ny = P.g.ny, nxt = P.g.nxt, nyt = ny - 1;
  if (io > nx) return;
  auto W = [&](int i, int j) { return P.wekto[(long)(j - 1) * P.ldt + (i - 1)]; };
  const bool cyc = P.g.cyc != 0;
  int ic = io;
  if (cyc && io == nx) ic = 1; // wekpo(nxpo, j) = wekpo(1, j)
  double v;
  if (jo >= 2 && jo <= ny - 1) {
    if (ic == 1) v = cyc ? 0.25 * (W(nxt, jo - 1) + W(nxt, jo) + W(1, jo - 1) + W(1, jo)) : 0.5 * (W(1, jo - 1) + W(1, jo));
    else if (ic == nx) v = 0.5 * (W(nxt, jo - 1) + W(nxt, jo));
    else v = 0.25 * (W(ic - 1, jo - 1) + W(ic - 1, jo) + W(ic, jo - 1) + W(ic, jo));
  } else {
    const int jt = (jo == 1) ? 1 : nyt;
    if (ic == 1) v = cyc ? 0.5 * (W(nxt, jt) + W(1, jt)) : W(1, jt);
    else if (ic == nx) v = W(nxt, jt);
    else v = 0.5 * (W(ic - 1, jt) + W(ic, jt));
  }
  P.wekpo[(long)(jo - 1) * P.g.ldx + (io - 1)] = v;
}
