// k_oml.h - ocean mixed layer on the device.
//
// Replaces `call oml` of the reference main program (src/q-gcm.F:1232):
//   omladf  advective + diffusive tendency of the mixed-layer temperature, second-order C-grid advection by the
//           geostrophic + Ekman velocity of the top layer, Del^2 and Del^4 diffusion of the lagged sst with
//           no-flux walls or specified boundary temperature (sb_hflux / nb_hflux)   src/omlsubs.F:244-763
//   oml     leapfrog step of sst (7.11), entrainment at T points (7.12), convective adjustment (7.13),
//           removal of the mean entrainment, averaging onto the p grid (entoc), xon(1) = area integral,
//           cyclic: boundary line integrals enisoc(1) / eninoc(1)                    src/omlsubs.F:47-236
// Without it the host would need po(:,:,1) down and entoc up over PCIe every step (2 x 7.4 MB at 5 km).
//
// Three launches:
//   k_oml_step   one thread per T point, operand fields staged through LDS: rhs (every expression in the reference's
//                operand order, contraction off => sst is bitwise the reference's), new sst into the spare buffer (the
//                sst buffers rotate: new -> sst, sst -> sstm), raw entrainment xfo, per-workgroup partial sums
//   k_oml_entoc  every workgroup re-reduces the partials in the same fixed order (no extra launch, same mean
//                everywhere), entoc = average of (xfo - mean) onto the p points, partials of xintp / line sums.
//                One generation of workgroups (OML_ERR tile rows each): the re-reduction is a ~2 us prologue that
//                every generation pays (round 1: 3.5 generations, 9.3 us for 15 MB of work)
//   k_oml_final  one workgroup: xon(1), enisoc(1), eninoc(1), monitors cfraoc / centoc.  Inside qgcm_hip_steps
//                this is not a launch: workgroup 0 of the tendency kernel that follows does it (oml_final_block),
//                before it steps dpioc with the new xon (a device-wide completion counter instead would need
//                device-scope fences - an L2 write-back per workgroup on this multi-XCD part: measured 84 us)
// The reference's sums run over j then i (and thread-dependent under OpenMP); here the order is fixed but
// different, so the mean entrainment agrees to rounding (tests: 1e-13 of max|xfo|), everything else bitwise.
// Algorithmic traffic: read sstm, sst, po(1), tauxo, tauyo, fnetoc, wekto (7) + write sst, xfo (2) in the first
// kernel, read xfo + write entoc (2) in the second: 11 N * 8 B = 81 MB at 5 km.
#pragma once
#include "qgcm_dev.h"

#define OML_TX 64
#define OML_TY 4
#define OML_NT (OML_TX * OML_TY)
#define OML_RPT 4 // rows per thread in k_oml_step
#define OML_ERR 1 // tile rows per workgroup in k_oml_entoc (4: one generation of fat workgroups - 17 us instead of 9)

struct QgOmlParams {
  int nxt, nyt, nx, ny, cyc, sb, nb; // nyt, ny: rows of the LOCAL T / p arrays
  int ldt, ldx;
  // y-slab view (whole domain: joff = 0, jX0 = jT0 = jP0 = 1, jT1 = nytg = nyt, jP1 = nyg = ny): global row = local +
  // joff; this handle owns T rows jT0..jT1 and p rows jP0..jP1.  k_oml_step also computes T row jX0 = jT0 - 1 when a
  // neighbour lies below: k_oml_entoc averages the entrainment of T rows j-1 and j onto p row j, and recomputing that
  // one row (its stencil fits the three halo rows) is cheaper than another exchange; its sums count owned rows only.
  int joff, nytg, nyg, jX0, jT0, jT1, jP0, jP1;
  // y-slabs: the basin-wide mean entrainment needs every rank's sum: (3, nranks) from the all-gather of
  // k_oml_ranksum's message; nullptr: the sums of this handle's own partials (whole domain)
  const double *mean_gath;
  int nranks;
  const double *sst, *sstm; // T grid, pitch ldt
  double *sstn;             // new sst (spare buffer)
  const double *fnet, *wekto;
  double *xfo;
  const double *po1, *taux, *tauy; // p grid, pitch ldx (top layer of po)
  double *entoc;
  double *partA; // (3, nblkA): xfo sum, cfrasm, centsm per workgroup of k_oml_step
  double *partB; // (3, nblkB): xintp sum, S / N line sums per workgroup of k_oml_entoc
  int nblkA, nblkB;
  QgScalars *sc;
  double *diag; // cfraoc, centoc
  double uvgfac, rhf0hm, hdxom1, d2tfac, d4tfac, hmoinv, dtoinv, entfac, tdto, rrcpoc, toc1, tsbdy, tnbdy, ocnorm, dxo, dyo;
};

// Sums of NV values over the 256 threads of a workgroup in a fixed order (xor butterfly inside each wave, then the
// four wave totals left to right); every thread returns with the totals. One barrier; sm is (NV x 4) doubles.
template <int NV>
__device__ __forceinline__ void oml_block_sums(double *v, double *sm, int tid) {
#pragma unroll
  for (int q = 0; q < NV; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
    if ((tid & 63) == 0 && tid < OML_NT) sm[q * 4 + (tid >> 6)] = v[q]; // (a caller may have more than OML_NT threads)
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = (sm[q * 4] + sm[q * 4 + 1]) + (sm[q * 4 + 2] + sm[q * 4 + 3]);
}

// grid: (ceil(nxt/64), ceil(rows/OML_SH)), block 256 = 64 x 4; thread rows j0 + ty + 4 r, r < OML_SH/4
// Round 4: every operand field goes through LDS.  The round-1 kernel read its stencils straight from global memory: five
// loads of the lagged sst per Del^2 value (five trips of a rolled loop) and ~20 per T point, i.e. ~9 dependent memory
// round trips per workgroup and 31 eight-byte loads per point through the L1 / TA path - 22.6 us for 66.5 MB of compulsory
// traffic (PMC: 89 MB; profiles/r4_natl5_oml_*).  Now: ONE round trip stages the tile of sstm (halo 2), sst (halo 1) and
// the p-grid tiles of po(1), tauxo, tauyo (columns i .. i+1, rows j .. j+1), all loads in flight together, zonal wrap /
// wall clamp applied while staging; Del^2, the face velocities and the advection read LDS.  Arithmetic unchanged (every
// expression in the reference's operand order): sst stays bitwise the reference's.
#ifndef OML_SH
#define OML_SH 8 // tile rows (31 KB of LDS: five workgroups per CU; 16 rows: 56 KB, two per CU)
#endif
__global__ __launch_bounds__(OML_NT) void k_oml_step(const QgOmlParams P) {
  constexpr int TH = OML_SH, RPT = TH / OML_TY;
  constexpr int MW = OML_TX + 4, MH = TH + 4; // sstm tile, halo 2: local (lx, ly) <-> T point (i0 - 2 + lx, j0 - 2 + ly)
  constexpr int SW = OML_TX + 2, SH = TH + 2; // sst tile, halo 1
  constexpr int PW = OML_TX + 1, PH = TH + 1; // p-grid tiles: p points (i0 + lx, j0 + ly)
  constexpr int DW = OML_TX + 2, DH = TH + 2; // del2t tile, halo 1
  __shared__ double sM[MH * MW], sS[SH * SW], sP[PH * PW], sX[PH * PW], sY[PH * PW], sD[DH * DW];
  __shared__ double red[12];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * OML_TX + 1, j0 = blockIdx.y * TH + P.jX0; // local rows jX0..jT1
  const int lx0 = tid % OML_TX, ly0 = tid / OML_TX;
  const int i = i0 + lx0;
  const int nxt = P.nxt, nyt = P.nyt, cyc = P.cyc;
  const long ldt = P.ldt, ldx = P.ldx;
  // T column that local column position gi stands for: zonal wrap (cyclic) or the wall column itself (box: the dummy
  // columns of src/omlsubs.F:349-357 copy the edge column; nothing else reads a clamped value)
  // (modulo, not one subtraction: the last tile of a narrow channel reaches more than one period past the edge - those
  //  positions are never used, but their addresses must stay inside the row)
  auto tcol = [&](int gi) { return cyc ? ((gi - 1) % nxt + nxt) % nxt + 1 : (gi < 1 ? 1 : (gi > nxt ? nxt : gi)); };
  auto trow = [&](int gj) { return gj < 1 ? 1 : (gj > nyt ? nyt : gj); }; // (rows outside the local array: address clamped, value unused)
  // ---- stage: all loads of a thread in flight together
  {
    constexpr int NM = (MH * MW + OML_NT - 1) / OML_NT, NS = (SH * SW + OML_NT - 1) / OML_NT, NP = (PH * PW + OML_NT - 1) / OML_NT;
    double vm[NM], vs[NS], vp[NP], vx[NP], vy[NP];
#pragma unroll
    for (int e = 0; e < NM; ++e) {
      const int idx = tid + e * OML_NT, lx = idx % MW, ly = idx / MW;
      vm[e] = P.sstm[(long)(trow(j0 - 2 + (ly < MH ? ly : 0)) - 1) * ldt + (tcol(i0 - 2 + lx) - 1)];
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int idx = tid + e * OML_NT, lx = idx % SW, ly = idx / SW;
      vs[e] = P.sst[(long)(trow(j0 - 1 + (ly < SH ? ly : 0)) - 1) * ldt + (tcol(i0 - 1 + lx) - 1)];
    }
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int idx = tid + e * OML_NT, lx = idx % PW, ly = idx / PW;
      const int pi = i0 + lx > P.nx ? P.nx : i0 + lx, pj0 = j0 + (ly < PH ? ly : 0), pj = pj0 > P.ny ? P.ny : pj0;
      const long o = (long)(pj - 1) * ldx + (pi - 1);
      vp[e] = P.po1[o];
      vx[e] = P.taux[o];
      vy[e] = P.tauy[o];
    }
#pragma unroll
    for (int e = 0; e < NM; ++e) {
      const int idx = tid + e * OML_NT;
      if (idx < MH * MW) sM[idx] = vm[e];
    }
#pragma unroll
    for (int e = 0; e < NS; ++e) {
      const int idx = tid + e * OML_NT;
      if (idx < SH * SW) sS[idx] = vs[e];
    }
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      const int idx = tid + e * OML_NT;
      if (idx < PH * PW) {
        sP[idx] = vp[e];
        sX[idx] = vx[e];
        sY[idx] = vy[e];
      }
    }
  }
  // pointwise operands of this thread's own points, requested before the barrier
  double e_fnet[RPT], e_wk[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int j = j0 + ly0 + OML_TY * r;
    const bool in = i <= nxt && j <= P.jT1;
    const long o = in ? (long)(j - 1) * ldt + (i - 1) : 0;
    e_fnet[r] = P.fnet[o];
    e_wk[r] = P.wekto[o];
  }
  __syncthreads();
  // ---- del2t of the tile and its halo, each value once, from the sstm tile.  Boundary variants of src/omlsubs.F:297-300
  // (W), 331-346 (E), 403-422 (S), 437-454 (N), 466-647 (corners), the operand order of every case the reference's; the
  // dummy columns by the wall / wrap rule (:349-357): a box evaluates the wall column again, a channel the wrapped one
  for (int idx = tid; idx < DH * DW; idx += OML_NT) {
    const int gi = i0 - 1 + idx % DW, gj = j0 - 1 + idx / DW;
    const int G = gj + P.joff; // global T row: the boundary variants belong to the basin's first and last row
    double val = 0.0;
    if (G >= 1 && G <= P.nytg && gj >= 1 && gj <= nyt && gi >= 0 && gi <= nxt + 1) {
      const int ie = tcol(gi);                    // the T column evaluated
      const int lx = (cyc ? gi : ie) - (i0 - 2);  // ... and where its neighbourhood sits in the tile
      const double *T = &sM[(gj - (j0 - 2)) * MW + lx];
      const bool hasW = (ie > 1) || cyc, hasE = (ie < nxt) || cyc;
      const double cc = T[0];
      const double w = hasW ? T[-1] : 0.0;
      const double e = hasE ? T[1] : 0.0;
      double acc, n;
      if (G == 1) { // W, E, N, tsbdy
        const double nn = T[MW];
        if (hasW) { acc = w; n = 1.0; if (hasE) { acc = acc + e; n = 2.0; } }
        else { acc = e; n = 1.0; } // a row has at least one x neighbour
        acc = acc + nn; n += 1.0;
        if (P.sb) { acc = acc + P.tsbdy; n += 1.0; }
        val = acc - n * cc;
      } else {
        const double s = T[-MW];
        if (G == P.nytg) {
          if (cyc && ie == nxt && P.nb) val = s + w + e - 4.0 * cc + P.tnbdy; // :630-631
          else {
            acc = s; n = 1.0; // S, W, tnbdy, E
            if (hasW) { acc = acc + w; n += 1.0; }
            if (P.nb) { acc = acc + P.tnbdy; n += 1.0; }
            if (hasE) { acc = acc + e; n += 1.0; }
            val = acc - n * cc;
          }
        } else {
          acc = s; n = 1.0; // S, W, E, N
          if (hasW) { acc = acc + w; n += 1.0; }
          if (hasE) { acc = acc + e; n += 1.0; }
          acc = acc + T[MW]; n += 1.0;
          val = acc - n * cc;
        }
      }
    }
    sD[idx] = val;
  }
  __syncthreads();
  const double uvgfac = P.uvgfac, rhf0hm = P.rhf0hm, hdxom1 = P.hdxom1;
  double sxfo = 0.0, scfr = 0.0, scen = 0.0;
  // local accessors: p points (ii, jj) with ii in i .. i+1, jj in j .. j+1; T points (ii, jj) within one of (i, j)
#define PO1(ii, jj) sP[((jj)-j0) * PW + ((ii)-i0)]
#define TXo(ii, jj) sX[((jj)-j0) * PW + ((ii)-i0)]
#define TYo(ii, jj) sY[((jj)-j0) * PW + ((ii)-i0)]
#define STL(di, jj) sS[((jj)-(j0 - 1)) * SW + (lx0 + 1 + (di))]
#define UF(ii, jj) (-uvgfac * (PO1(ii, (jj) + 1) - PO1(ii, jj)) + rhf0hm * (TYo(ii, (jj) + 1) + TYo(ii, jj)))
#define VF(ii, jj) (uvgfac * (PO1((ii) + 1, jj) - PO1(ii, jj)) - rhf0hm * (TXo((ii) + 1, jj) + TXo(ii, jj)))
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int ly = ly0 + OML_TY * r;
    const int j = j0 + ly;
    if (i > nxt || j > P.jT1) continue;
    const int G = j + P.joff; // global T row
    // ---- advection, src/omlsubs.F:281-346 (rows), 370-456 (S/N rows), 458-700 (corners) ----
    double um, tm, up, tp;
    if (i == 1 && !cyc) { um = 0.0; tm = 0.0; }
    else { um = UF(i, j); tm = STL(-1, j) + STL(0, j); }
    if (i == nxt && !cyc) { up = 0.0; tp = 0.0; }
    else { up = UF(i + 1, j); tp = STL(0, j) + STL(1, j); }
    const double hxadv = hdxom1 * (up * tp - um * tm);
    double hyadv;
    if (G == 1) {
      const double vp = VF(i, j + 1), tp2 = STL(0, j) + STL(0, j + 1);
      if (P.sb) {
        const double vm = -rhf0hm * (TXo(i + 1, j) + TXo(i, j)), tm2 = STL(0, j) + P.tsbdy;
        hyadv = hdxom1 * (vp * tp2 - vm * tm2);
      } else hyadv = hdxom1 * (vp * tp2);
    } else if (G == P.nytg) {
      const double vm = VF(i, j), tm2 = STL(0, j - 1) + STL(0, j);
      if (P.nb) {
        const double vp = -rhf0hm * (TXo(i + 1, j + 1) + TXo(i, j + 1)), tp2 = STL(0, j) + P.tnbdy;
        hyadv = hdxom1 * (vp * tp2 - vm * tm2);
      } else hyadv = hdxom1 * (-vm * tm2);
    } else {
      const double vm = VF(i, j), vp = VF(i, j + 1);
      hyadv = hdxom1 * (vp * (STL(0, j + 1) + STL(0, j)) - vm * (STL(0, j) + STL(0, j - 1)));
    }
    double rhs = -(hxadv + hyadv);
    // ---- diffusion, src/omlsubs.F:733-759 ----
    const double *d = &sD[(ly + 1) * DW + (lx0 + 1)];
    const double dc = d[0], dw = d[-1], de = d[1];
    if (G == 1) rhs = rhs + P.d2tfac * dc - P.d4tfac * (dw + de + d[DW] - 3.0 * dc);
    else if (G == P.nytg) rhs = rhs + P.d2tfac * dc - P.d4tfac * (d[-DW] + dw + de - 3.0 * dc);
    else rhs = rhs + P.d2tfac * dc - P.d4tfac * (d[-DW] + dw + de + d[DW] - 4.0 * dc);
    // ---- oml, src/omlsubs.F:101-128 ----
    const long o = (long)(j - 1) * ldt + (i - 1);
    const double sm = sM[(ly + 2) * MW + (lx0 + 2)], wk = e_wk[r];
    const double diabat = 0.5 * wk * (sm + P.toc1);
    double sstnew = sm + P.tdto * (rhs + P.hmoinv * (P.rrcpoc * e_fnet[r] + diabat));
    const double xfoent = -(0.5 * P.dtoinv) * wk * (sm - P.toc1);
    const double dtonew = P.toc1 - sstnew;
    const double coneno = P.entfac * fmax(0.0, dtonew);
    const double xf = xfoent - coneno;
    sstnew = sstnew + fmax(0.0, dtonew);
    P.xfo[o] = xf;
    if (j < P.jT0) continue; // the recomputed row below the slab: its sst and its sums belong to the neighbour
    P.sstn[o] = sstnew;
    sxfo += xf;
    scfr += (0.5 - copysign(0.5, -dtonew));
    scen -= coneno;
  }
#undef PO1
#undef TXo
#undef TYo
#undef STL
#undef UF
#undef VF
  const int b = blockIdx.y * gridDim.x + blockIdx.x;
  double t[3] = {sxfo, scfr, scen};
  oml_block_sums<3>(t, red, tid);
  if (tid == 0) {
    P.partA[b] = t[0];
    P.partA[P.nblkA + b] = t[1];
    P.partA[2 * P.nblkA + b] = t[2];
  }
}

// grid: (ceil(nx/64), ceil(ny/(16*OML_ERR))), block 256 = 64 x 4, thread rows j0 + ty + 4 r of OML_ERR tile rows
__global__ __launch_bounds__(OML_NT) void k_oml_entoc(const QgOmlParams P) {
  __shared__ double redm[4], red[12];
  const int tid = threadIdx.x;
  const int i = blockIdx.x * OML_TX + (tid % OML_TX) + 1;
  const int nx = P.nx, nxt = P.nxt, nyg = P.nyg;
  const long ldt = P.ldt;
  // The T cells around every p point of this thread, read BEFORE the mean is reduced: the reduction below ends in a
  // barrier, which no load crosses - issued after it, the cells' round trip came on top of the partials' (9.4 -> 7 us).
  // n = cells in use (4 interior, 2 on a wall, 1 in a box corner), in the reference's order of summation.
  constexpr int NP = OML_ERR * OML_RPT;
  double raw[NP][4];
  int ncell[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int rr = q / OML_RPT, r = q % OML_RPT;
    const int j = (blockIdx.y * OML_ERR + rr) * (OML_TY * OML_RPT) + (tid / OML_TX) + OML_TY * r + P.jP0; // owned p rows
    ncell[q] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) raw[q][e] = 0.0;
    if (i > nx || j > P.jP1) continue;
    const int G = j + P.joff; // global p row; T rows j-1 and j (local) lie below / above it
    const bool xin = (i >= 2 && i <= nx - 1), yin = (G >= 2 && G <= nyg - 1);
    const int jt = (G == 1) ? j : j - 1; // the basin's first / last T row (rows on a zonal wall)
    int ci[4], cj[4], n;
    if (xin && yin) { n = 4; ci[0] = i - 1; cj[0] = j - 1; ci[1] = i; cj[1] = j - 1; ci[2] = i - 1; cj[2] = j; ci[3] = i; cj[3] = j; } // :162-163
    else if (xin) { n = 2; ci[0] = i - 1; cj[0] = jt; ci[1] = i; cj[1] = jt; }                                                             // :171-172
    else if (P.cyc) { // :178-189: W column from the wrapped T cells, E column = W column
      if (yin) { n = 4; ci[0] = nxt; cj[0] = j - 1; ci[1] = 1; cj[1] = j - 1; ci[2] = nxt; cj[2] = j; ci[3] = 1; cj[3] = j; }
      else { n = 2; ci[0] = nxt; cj[0] = jt; ci[1] = 1; cj[1] = jt; }
    } else { // :194-202
      const int it = (i == 1) ? 1 : nxt;
      if (yin) { n = 2; ci[0] = it; cj[0] = j - 1; ci[1] = it; cj[1] = j; }
      else { n = 1; ci[0] = it; cj[0] = jt; }
    }
    ncell[q] = n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < n) raw[q][e] = P.xfo[(long)(cj[e] - 1) * ldt + (ci[e] - 1)];
  }
  // mean entrainment: the same fixed-order reduction of the partials in every workgroup
  double s[1] = {0.0};
  if (P.mean_gath) { // y-slabs: the ranks' sums in rank order (every workgroup alike)
    if (tid == 0)
      for (int r = 0; r < P.nranks; ++r) s[0] += P.mean_gath[3 * r];
  } else {
    for (int k = tid; k < P.nblkA; k += OML_NT) s[0] += P.partA[k];
  }
  oml_block_sums<1>(s, redm, tid);
  const double xmean = s[0] * P.ocnorm; // xfosum*ocnorm, src/omlsubs.F:153
  double t[3] = {0.0, 0.0, 0.0}; // xintp sum, S and N line sums
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int rr = q / OML_RPT, r = q % OML_RPT;
    const int j = (blockIdx.y * OML_ERR + rr) * (OML_TY * OML_RPT) + (tid / OML_TX) + OML_TY * r + P.jP0;
    if (i > nx || j > P.jP1) continue;
    const int G = j + P.joff;
    const double x0 = raw[q][0] - xmean, x1 = raw[q][1] - xmean, x2 = raw[q][2] - xmean, x3 = raw[q][3] - xmean;
    double en;
    if (ncell[q] == 4) en = 0.25 * (x0 + x1 + x2 + x3);
    else if (ncell[q] == 2) en = 0.5 * (x0 + x1);
    else en = x0;
    P.entoc[(long)(j - 1) * P.ldx + (i - 1)] = en;
    const double wx = (i == 1 || i == nx) ? 0.5 : 1.0, wy = (G == 1 || G == nyg) ? 0.5 : 1.0; // xintp, src/intsubs.f:78-133
    t[0] += wx * wy * en;
    if (G == 1) t[1] += wx * en;  // src/omlsubs.F:222-231
    if (G == nyg) t[2] += wx * en;
  }
  const int b = blockIdx.y * gridDim.x + blockIdx.x;
  oml_block_sums<3>(t, red, tid);
  if (tid == 0) {
    P.partB[b] = t[0];
    P.partB[P.nblkB + b] = t[1];
    P.partB[2 * P.nblkB + b] = t[2];
  }
}

// what the final reduction needs (also carried by the tendency kernel's parameters, see k_tend.h)
struct QgOmlFinal {
  const double *partA, *partB;
  int nblkA, nblkB, cyc, on;
  QgScalars *sc;
  double *diag;
  double ocnorm, dxo, dyo;
};

// one workgroup of OML_NT threads; red: 20 doubles of LDS
__device__ __forceinline__ void oml_final_block(const QgOmlFinal &P, double *red, int tid) {
  double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int first = tid < OML_NT ? tid : (1 << 30); // threads beyond OML_NT only keep the barrier
  for (int k = first; k < P.nblkB; k += OML_NT) {
    a[0] += P.partB[k];
    a[1] += P.partB[P.nblkB + k];
    a[2] += P.partB[2 * P.nblkB + k];
  }
  for (int k = first; k < P.nblkA; k += OML_NT) {
    a[3] += P.partA[P.nblkA + k];
    a[4] += P.partA[2 * P.nblkA + k];
  }
  double *t = a;
  oml_block_sums<5>(a, red, tid);
  if (tid == 0) {
    P.sc->xon[0] = t[0] * P.dxo * P.dyo; // src/omlsubs.F:215-216
    if (P.cyc) {
      P.sc->enisoc[0] = P.dxo * t[1]; // :232-233
      P.sc->eninoc[0] = P.dxo * t[2];
    }
    P.diag[0] = t[3] * P.ocnorm;        // cfraoc, :207
    P.diag[1] = t[4] * P.dxo * P.dyo;   // centoc, :208
  }
}

__global__ __launch_bounds__(OML_NT) void k_oml_final(const QgOmlFinal P) {
  __shared__ double red[20];
  oml_final_block(P, red, threadIdx.x);
}

// ---- y-slabs ---------------------------------------------------------------------------------------------------------
// The sums of a slab's per-workgroup partials (3, nblk) in the fixed order of oml_final_block: out[0..2].  One workgroup.
// k_oml_step's go into the mixed layer's own small all-gather (the mean entrainment must be known before entoc is
// formed), k_oml_entoc's ride at the end of the step message of the tridiagonal sweeps.
__global__ __launch_bounds__(OML_NT) void k_oml_sum3(const double *part, int nblk, double *out) {
  __shared__ double red[12];
  const int tid = threadIdx.x;
  double a[3] = {0.0, 0.0, 0.0};
  for (int k = tid; k < nblk; k += OML_NT) {
    a[0] += part[k];
    a[1] += part[nblk + k];
    a[2] += part[2 * nblk + k];
  }
  oml_block_sums<3>(a, red, tid);
  if (tid < 3) out[tid] = a[tid];
}

// xon(1), enisoc(1) / eninoc(1) and the monitors from every rank's sums, in rank order (the same on every rank):
// gA = (3, nranks) of the mixed layer's all-gather, gB = the three numbers at gB + r*strideB of the step messages.
__global__ void k_oml_final_slab(const double *gA, const double *gB, long strideB, int nranks, QgOmlFinal P) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int r = 0; r < nranks; ++r) {
    t[0] += gB[r * strideB];
    t[1] += gB[r * strideB + 1];
    t[2] += gB[r * strideB + 2];
    t[3] += gA[3 * r + 1];
    t[4] += gA[3 * r + 2];
  }
  P.sc->xon[0] = t[0] * P.dxo * P.dyo; // src/omlsubs.F:215-216
  if (P.cyc) {
    P.sc->enisoc[0] = P.dxo * t[1]; // :232-233
    P.sc->eninoc[0] = P.dxo * t[2];
  }
  P.diag[0] = t[3] * P.ocnorm;        // cfraoc, :207
  P.diag[1] = t[4] * P.dxo * P.dyo;   // centoc, :208
}

// halo rows of the new sst: the first / last three owned T rows go to the lower / upper neighbour (appended to the
// halo message of the step), three rows come back on each side.  grid: (ceil(nxt/256), 3, 2 sides)
// (the local T array ends two rows above the slab - all the stencils need there; the third row of that message is unused)
__global__ __launch_bounds__(256) void k_oml_halo(double *sst, int ldt, int nxt, int nyt, int jT0, int jT1, double *to_lo,
                                                  double *to_hi, const double *from_lo, const double *from_hi, int unpack) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y, hi = blockIdx.z;
  if (i >= nxt) return;
  if (!unpack) {
    double *dst = hi ? to_hi : to_lo;
    if (!dst) return;
    const int j = hi ? jT1 - 2 + r : jT0 + r; // local T row (1-based)
    dst[(long)r * ldt + i] = sst[(long)(j - 1) * ldt + i];
  } else {
    const double *src = hi ? from_hi : from_lo;
    if (!src) return;
    const int j = hi ? jT1 + 1 + r : jT0 - 3 + r;
    if (j < 1 || j > nyt) return;
    sst[(long)(j - 1) * ldt + i] = src[(long)r * ldt + i];
  }
}

// leapfrog averaging of the mixed-layer temperature, src/q-gcm.F:1345-1351
__global__ __launch_bounds__(256) void k_oml_average(double *sst, const double *sstm, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) sst[t] = 0.5 * (sst[t] + sstm[t]);
}
