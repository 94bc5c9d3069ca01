// k_dst64.h - K3 fast path: DST-I rows of length n = 64*M - 1, one wavefront per
// row PAIR, no workgroup barriers.
//
// Same mathematics as k_dst.h (FFTPACK dsint restated: pre-twiddle, length
// N = n+1 real FFT, running-sum post-process; src/fftpack/newbihar/dsint.f:16-40;
// called from hsbxoc, src/ocisubs.F:461-463,494-499) with the complex FFT of
// the packed row pair organised for a 64-lane wave:
//
//   N = 64*M,  n = 64*n1 + n2,  k = k1 + M*k2
//   (1) lane n2 holds z[64*n1 + n2], n1 = 0..M-1, in registers: M-point DFT
//       over n1 in registers (M = 15 as 3x5), twiddle W_N^(n2*k1);
//   (2) the M independent 64-point FFTs over n2 run through LDS as two
//       radix-8 passes (8 = contiguous / stride-8 accesses on a padded row);
//   (3) conjugate-symmetric split of the two real spectra, FFTPACK's running
//       sum as a wave scan (shuffles), results staged in LDS and stored with
//       16-byte coalesced writes.
// LDS traffic per row pair ~ 6 passes over 15 KB; everything is wave-synchronous
// (LDS operations of one wave execute in order), so a workgroup is just four
// independent waves.
#pragma once
#include "qgcm_dev.h"
#include "k_dst.h" // cplx helpers
#include "k_misc.h" // constraint solve (k_dst64_unpack<.., CONSTR>)

// The row transform is not required to be bitwise FFTPACK:
// let the compiler fuse multiply-adds in this file only.
#pragma clang fp contract(fast)

#ifndef D64_WAVES
#define D64_WAVES 2 // 2 waves per workgroup spread the 1440 row-pair waves more evenly over the 256 CUs than 4 (17.0 -> 16.2 us)
#endif
#define D64_NT (64 * D64_WAVES)
#define D64_ROW 72 // padded length of one 64-point row (pad 1 per 8)

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ cplx cscale(cplx a, double s) { return {a.x * s, a.y * s}; }
__device__ __forceinline__ cplx cmpi(cplx a) { return {-a.y, a.x}; } // * (+i)

// forward 8-point DFT, y[c] = sum_a x[a] exp(-2 pi i a c / 8), in place
__device__ __forceinline__ void dft8(cplx *x) {
  const double r2 = 0.70710678118654752440;
  cplx e0 = cadd(x[0], x[4]), o0 = csub(x[0], x[4]);
  cplx e1 = cadd(x[1], x[5]), o1 = csub(x[1], x[5]);
  cplx e2 = cadd(x[2], x[6]), o2 = csub(x[2], x[6]);
  cplx e3 = cadd(x[3], x[7]), o3 = csub(x[3], x[7]);
  // odd branch twiddles W8^a
  cplx p1 = {r2 * (o1.x + o1.y), r2 * (o1.y - o1.x)};   // o1 * (1 - i)/sqrt2
  cplx p2 = cmni(o2);                                   // o2 * (-i)
  cplx p3 = {r2 * (o3.y - o3.x), -r2 * (o3.x + o3.y)};  // o3 * (-1 - i)/sqrt2
  cplx t0 = cadd(e0, e2), t1 = csub(e0, e2), t2 = cadd(e1, e3), t3 = cmni(csub(e1, e3));
  x[0] = cadd(t0, t2);
  x[4] = csub(t0, t2);
  x[2] = cadd(t1, t3);
  x[6] = csub(t1, t3);
  cplx u0 = cadd(o0, p2), u1 = csub(o0, p2), u2 = cadd(p1, p3), u3 = cmni(csub(p1, p3));
  x[1] = cadd(u0, u2);
  x[5] = csub(u0, u2);
  x[3] = cadd(u1, u3);
  x[7] = csub(u1, u3);
}

__device__ __forceinline__ void dft3(cplx &a0, cplx &a1, cplx &a2) {
  const double s3 = 0.86602540378443864676;
  cplx t1 = cadd(a1, a2);
  cplx t2 = {a0.x - 0.5 * t1.x, a0.y - 0.5 * t1.y};
  cplx d = csub(a1, a2);
  cplx t3 = {s3 * d.y, -s3 * d.x};
  a0 = cadd(a0, t1);
  a1 = cadd(t2, t3);
  a2 = csub(t2, t3);
}

__device__ __forceinline__ void dft5(cplx &a0, cplx &a1, cplx &a2, cplx &a3, cplx &a4) {
  const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
  const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
  cplx t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
  cplx m1 = {a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y};
  cplx m2 = {a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y};
  cplx n1 = {s1 * t3.y + s2 * t4.y, -(s1 * t3.x + s2 * t4.x)};
  cplx n2 = {s2 * t3.y - s1 * t4.y, -(s2 * t3.x - s1 * t4.x)};
  a0 = {a0.x + t1.x + t2.x, a0.y + t1.y + t2.y};
  a1 = cadd(m1, n1);
  a4 = csub(m1, n1);
  a2 = cadd(m2, n2);
  a3 = csub(m2, n2);
}

// in-register forward DFT of length M over a[0..M-1] (natural order in and out)
template <int M>
__device__ __forceinline__ void dftM(cplx *a);

template <>
__device__ __forceinline__ void dftM<3>(cplx *a) {
  dft3(a[0], a[1], a[2]);
}

// 15 = 3 x 5 Cooley-Tukey: n = 5*na + nb, k = ka + 3*kb
template <>
__device__ __forceinline__ void dftM<15>(cplx *a) {
  // W15^e = exp(-2 pi i e / 15) for e = 1,2,3,4,6,8
  const cplx w1 = {0.9135454576426008955, -0.4067366430758002078};
  const cplx w2 = {0.6691306063588582138, -0.7431448254773942350};
  const cplx w3 = {0.3090169943749474241, -0.9510565162951535721};
  const cplx w4 = {-0.1045284632676534714, -0.9945218953682733369};
  const cplx w6 = {-0.8090169943749474241, -0.5877852522924731292};
  const cplx w8 = {-0.9781476007338056379, 0.2079116908177593371};
  // radix-3 over na for each nb: (a[nb], a[5+nb], a[10+nb]) -> t[ka][nb] stored at a[5*ka + nb]
#pragma unroll
  for (int nb = 0; nb < 5; ++nb) dft3(a[nb], a[5 + nb], a[10 + nb]);
  // twiddle W15^(nb*ka)
  a[5 + 1] = cmul(a[5 + 1], w1);
  a[5 + 2] = cmul(a[5 + 2], w2);
  a[5 + 3] = cmul(a[5 + 3], w3);
  a[5 + 4] = cmul(a[5 + 4], w4);
  a[10 + 1] = cmul(a[10 + 1], w2);
  a[10 + 2] = cmul(a[10 + 2], w4);
  a[10 + 3] = cmul(a[10 + 3], w6);
  a[10 + 4] = cmul(a[10 + 4], w8);
  // radix-5 over nb for each ka -> X[ka + 3*kb] at a[5*ka + kb]
#pragma unroll
  for (int ka = 0; ka < 3; ++ka) dft5(a[5 * ka], a[5 * ka + 1], a[5 * ka + 2], a[5 * ka + 3], a[5 * ka + 4]);
  // reorder to natural k = ka + 3*kb
  cplx t[15];
#pragma unroll
  for (int ka = 0; ka < 3; ++ka)
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) t[ka + 3 * kb] = a[5 * ka + kb];
#pragma unroll
  for (int k = 0; k < 15; ++k) a[k] = t[k];
}

// Inclusive sum scan over the 64 lanes of a wave for two values at once, by DPP moves (VALU only; the shuffle form
// costs two ds_bpermute per value and step on the LDS pipe, which is what bounds these kernels).  Within each row of 16
// lanes: shifts by 1, 2, 4, 8 (lanes without a source add 0); then lane 15 of rows 0 / 2 is added to rows 1 / 3, and
// lane 31 to rows 2 and 3.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double d64_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROWMASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROWMASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void d64_wave_scan2(double &a, double &b) {
  a += d64_dpp<0x111, 0xf>(a); b += d64_dpp<0x111, 0xf>(b); // row_shr:1
  a += d64_dpp<0x112, 0xf>(a); b += d64_dpp<0x112, 0xf>(b); // row_shr:2
  a += d64_dpp<0x114, 0xf>(a); b += d64_dpp<0x114, 0xf>(b); // row_shr:4
  a += d64_dpp<0x118, 0xf>(a); b += d64_dpp<0x118, 0xf>(b); // row_shr:8
  a += d64_dpp<0x142, 0xa>(a); b += d64_dpp<0x142, 0xa>(b); // row_bcast:15 into rows 1, 3
  a += d64_dpp<0x143, 0xc>(a); b += d64_dpp<0x143, 0xc>(b); // row_bcast:31 into rows 2, 3
}

// The transform of one row pair by one wave: reads rowa/rowb (global), leaves the two output rows
// in the wave's LDS buffer F viewed as doubles: row a at raw[pidx(i)], row b at raw[NP + pidx(i)],
// i = 0..N-2, pidx(i) = i + (i >> 4), NP = N + N/16.  rsa / rsb = sums of the output rows.
// Split in two so that a caller can issue its own prefetches BETWEEN the halves: the front half (row loads,
// pre-twiddle, M-point DFTs) peaks at ~230 VGPRs, all dead when it returns.
template <int M>
__device__ __forceinline__ void dst64_front(const QgDstParams &P, const double *rowa, const double *rowb, bool has_b,
                                            cplx *F, cplx *W64, int lane) {
  constexpr int N = 64 * M, n = N - 1, NS2 = n / 2; // n odd: NS2 = N/2 - 1 = K

  // Everything this lane will need from global tables is requested up front so
  // that it arrives together with the rows: W_N^(lane*k1), the dsint sine
  // factors, and this wave's copy of the 64-point twiddles.
  cplx tw1[M]; // W_N^(lane*k1) as powers of W_N^lane (one coalesced load; depth-4 product tree)
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

  // ---- (1) pre-twiddle (dsint.f:19-33) into registers: a[n1] = z[64*n1 + lane].  z_j and z_{N-j} are built from the
  // SAME two row elements x(k), x(n+1-k), k = j <= N/2:  z_j = t1 + t2,  z_{N-j} = t2 - t1.  Round 2 let every lane
  // load the pair of each of its M elements - every row element was requested twice, and a CU's rate of outstanding
  // requests is what bounds the load phase of this one-generation kernel (phase stamps, DESIGN 3.6: rows in after
  // 4.2 us against 3.0 us for the Thomas sweep's same bytes).  Now a lane loads the pairs of its elements of the LOWER
  // half only (j <= N/2: registers n1 <= M/2), keeps z_j and hands z_{N-j} to its owner - lane (64 - lane) mod 64,
  // register M-1-n1 (M-n1 for lane 0) - through the wave's LDS buffer, which is idle until the M-point DFTs are done:
  // 32 instead of 60 row loads per wave, 8 LDS writes + 8 reads more.
  constexpr int MH = M / 2; // registers n1 = 0..MH hold (some) elements of the lower half
  cplx a[M];
  {
    const double *rb = has_b ? rowb : rowa; // the odd last row has no partner: its loads are redirected and zeroed
    const double bsc = has_b ? 1.0 : 0.0;
    double snv[MH + 1], xav[MH + 1], xac[MH + 1], xbv[MH + 1], xbc[MH + 1];
#pragma unroll
    for (int n1 = 0; n1 <= MH; ++n1) {
      const int j = 64 * n1 + lane;
      const int k = (j <= NS2 + 1) ? j : 0;                    // (lanes past the middle of register MH: clamped, unused)
      const int i1 = k > 0 ? k - 1 : 0, i2 = k > 0 ? n - k : 0; // j = 0: z = 0, loads clamped
      snv[n1] = P.sintab[(k <= NS2) ? k : 0];
      xav[n1] = rowa[i1];
      xac[n1] = rowa[i2];
      xbv[n1] = rb[i1];
      xbc[n1] = rb[i2];
    }
    const int ml = (64 - lane) & 63; // owner of the mirrored elements
#pragma unroll
    for (int n1 = 0; n1 <= MH; ++n1) {
      const int j = 64 * n1 + lane;
      const double va = xav[n1], ca = xac[n1], vb = bsc * xbv[n1], cb = bsc * xbc[n1];
      const double sn = snv[n1];
      const double t1a = va - ca, t2a = sn * (va + ca);
      const double t1b = vb - cb, t2b = sn * (vb + cb);
      cplx z = {t1a + t2a, t1b + t2b};
      if (j == NS2 + 1) z = {4.0 * va, 4.0 * vb}; // k = N - j = j: both loads hit x(j)
      if (j == 0) z = {0.0, 0.0};
      a[n1] = z; // (register MH, lanes past the middle: overwritten from LDS below)
      if (j >= 1 && j <= NS2) {
        const int mn1 = (lane == 0) ? M - n1 : M - 1 - n1; // N - j = 64*mn1 + ml
        F[mn1 * D64_ROW + ml + (ml >> 3)] = {t2a - t1a, t2b - t1b};
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int n1 = MH; n1 < M; ++n1) {
      const int j = 64 * n1 + lane;
      const cplx zz = F[n1 * D64_ROW + lane + (lane >> 3)];
      if (j >= NS2 + 2) a[n1] = zz;
    }
    wave_lds_sync(); // the buffer is rewritten below
  }
  dftM<M>(a);
#pragma unroll
  for (int k1 = 0; k1 < M; ++k1) {
    cplx v = (k1 == 0) ? a[0] : cmul(a[k1], tw1[k1]);
    F[k1 * D64_ROW + lane + (lane >> 3)] = v;
  }
  wave_lds_sync();
}

template <int M>
__device__ __forceinline__ void dst64_back(cplx *F, const cplx *W64, int lane, double &rsa, double &rsb) {
  constexpr int N = 64 * M, n = N - 1, NS2 = n / 2;
  double *raw = reinterpret_cast<double *>(F); // output rows a, b (padded)
  // ---- (2a) radix-8 over a (n2 = 8a + b), twiddle W64^(b*c), in place -------
  // (A wave has few neighbours to hide its LDS round trips behind - 1.4 waves per SIMD at NAtl 5 km - so every stage
  //  first asks for ALL it reads, of both rounds, then computes, then writes: round 2 rewrote these loops after the
  //  ISA showed read x8 -> wait -> compute -> write x8 per round, each round trip fully exposed.)
  constexpr int NBF = M * 8;
  constexpr int NRD = (NBF + 63) / 64;
  {
    cplx x[NRD][8];
    cplx tw[NRD][7];
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int id = lane + 64 * rd;
      const int idc = id < NBF ? id : NBF - 1; // (clamped: idle lanes of the last round read valid rows, write nothing)
      const int k1 = idc >> 3, b = idc & 7;
      const cplx *row = F + k1 * D64_ROW + b;
#pragma unroll
      for (int q = 0; q < 8; ++q) x[rd][q] = row[9 * q];
#pragma unroll
      for (int c = 1; c < 8; ++c) tw[rd][c - 1] = W64[b * c];
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      dft8(x[rd]);
#pragma unroll
      for (int c = 1; c < 8; ++c) x[rd][c] = cmul(x[rd][c], tw[rd][c - 1]);
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int id = lane + 64 * rd;
      if (id < NBF) {
        cplx *row = F + (id >> 3) * D64_ROW + (id & 7);
#pragma unroll
        for (int c = 0; c < 8; ++c) row[9 * c] = x[rd][c];
      }
    }
  }
  wave_lds_sync();
  // ---- (2b) radix-8 over b for fixed c: positions 9c + b -> 9c + d (k2 = c + 8d)
  {
    cplx x[NRD][8];
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int id = lane + 64 * rd;
      const int idc = id < NBF ? id : NBF - 1;
      const cplx *row = F + (idc >> 3) * D64_ROW + 9 * (idc & 7);
#pragma unroll
      for (int q = 0; q < 8; ++q) x[rd][q] = row[q];
    }
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) dft8(x[rd]);
#pragma unroll
    for (int rd = 0; rd < NRD; ++rd) {
      const int id = lane + 64 * rd;
      if (id < NBF) {
        cplx *row = F + (id >> 3) * D64_ROW + 9 * (id & 7);
#pragma unroll
        for (int d = 0; d < 8; ++d) row[d] = x[rd][d];
      }
    }
  }
  wave_lds_sync();

  // ---- (3) split + FFTPACK post-process (dsint.f:37-44) ----------------------
  // Z[k], k = k1 + M*k2, k2 = c + 8d, lives at F[k1*72 + 9c + d]
  constexpr int K = NS2;              // odd outputs b[2k+1], k = 1..K
  constexpr int CH = (K + 63) / 64;   // k's per lane
  const int k0 = 1 + lane * CH;
  // (k1, k2) of k0 and of N-k0 once; then step k -> k+1 / N-k -> N-k-1 without divisions
  int ak1 = k0 % M, ak2 = k0 / M;
  int bk1 = (N - k0) % M, bk2 = (N - k0) / M;
  double rea[CH], ima[CH], reb[CH], imb[CH];
  double suma = 0.0, sumb = 0.0;
  {
    // all reads of the spectrum first (their LDS positions do not depend on data), then the arithmetic
    cplx z1[CH], z2[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int za = ak1 * D64_ROW + 9 * (ak2 & 7) + (ak2 >> 3);
      const int zb = bk1 * D64_ROW + 9 * (bk2 & 7) + (bk2 >> 3);
      if (++ak1 == M) { ak1 = 0; ++ak2; }
      if (--bk1 < 0) { bk1 = M - 1; --bk2; }
      const bool ok = k0 + t <= K;
      z1[t] = F[ok ? za : 0];
      z2[t] = F[ok ? zb : 0];
    }
#pragma unroll
    for (int t = 0; t < CH; ++t) {
      const int k = k0 + t;
      if (k <= K) {
        rea[t] = 0.5 * (z1[t].x + z2[t].x);
        ima[t] = 0.5 * (z1[t].y - z2[t].y);
        reb[t] = 0.5 * (z1[t].y + z2[t].y);
        imb[t] = -0.5 * (z1[t].x - z2[t].x);
        suma += rea[t];
        sumb += reb[t];
      } else {
        rea[t] = ima[t] = reb[t] = imb[t] = 0.0;
      }
    }
  }
  const cplx z0 = F[0];
  // inclusive wave scan of the per-lane partial sums (DPP moves: no LDS-pipe instructions)
  double inca = suma, incb = sumb;
  d64_wave_scan2(inca, incb);
  const double b1a = 0.5 * z0.x, b1b = 0.5 * z0.y;
  double runa = b1a + (inca - suma), runb = b1b + (incb - sumb);
  wave_lds_sync(); // all spectrum reads done: reuse the buffer for the output rows
  // Output staging: element i of a row lives at i + (i >> 4) (one pad per 16) so that
  // the per-lane runs of 16 consecutive outputs fall on distinct LDS banks.
  constexpr int NP = N + N / 16; // padded row length
  auto pidx = [](int i) { return i + (i >> 4); };
  rsa = 0.0;
  rsb = 0.0;
  if (lane == 0) {
    raw[0] = b1a;
    raw[NP] = b1b;
    raw[pidx(N - 1)] = 0.0; // padding slots (element n of each row is unused)
    raw[NP + pidx(N - 1)] = 0.0;
    rsa = b1a;
    rsb = b1b;
  }
#pragma unroll
  for (int t = 0; t < CH; ++t) {
    const int k = k0 + t;
    if (k <= K) {
      runa += rea[t];
      runb += reb[t];
      raw[pidx(2 * k - 1)] = -ima[t];
      raw[pidx(2 * k)] = runa;
      raw[NP + pidx(2 * k - 1)] = -imb[t];
      raw[NP + pidx(2 * k)] = runb;
      rsa += runa - ima[t];
      rsb += runb - imb[t];
    }
  }
  wave_lds_sync();
}

template <int M>
__device__ __forceinline__ void dst64_core(const QgDstParams &P, const double *rowa, const double *rowb, bool has_b,
                                           cplx *F, cplx *W64, int lane, double &rsa, double &rsb) {
  dst64_front<M>(P, rowa, rowb, has_b, F, W64, lane);
  dst64_back<M>(F, W64, lane, rsa, rsb);
}

// grid: (ceil(npairs / 4), nlayers), block 64*D64_WAVES = independent waves
template <int M, bool ROWSUM>
__global__ __launch_bounds__(D64_NT) void k_dst64(const QgDstParams P) {
  constexpr int N = 64 * M;
  __shared__ __align__(16) cplx Fsh[D64_WAVES][M * D64_ROW];
  __shared__ __align__(16) cplx W64sh[D64_WAVES][64]; // exp(-2 pi i t / 64), per wave copy
  const int lane = threadIdx.x & 63;
  // (the wave's number as a scalar: the row pointers below are then uniform - one SGPR pair plus 32-bit lane offsets
  //  per load / store instead of a 64-bit address computed in VGPRs for each)
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ny = P.g.ny, ldw = P.g.ldw;
  const int m = blockIdx.y + P.layer0;
  const int pair = blockIdx.x * D64_WAVES + wv;
  const int ja = P.g.jr0 + 2 * pair;
  if (ja > P.g.jr1) return; // whole wave leaves; no workgroup barrier is ever used
  const bool has_b = (ja + 1 <= P.g.jr1);
  double *rowa = P.wrk + P.g.wstride * m + (long)(ja - 1) * ldw;
  double *rowb = rowa + ldw;
  double rsa, rsb;
  QG_STAMP(0, 0);
  dst64_front<M>(P, rowa, rowb, has_b, Fsh[wv], W64sh[wv], lane);
  QG_STAMP(0, 1);
  dst64_back<M>(Fsh[wv], W64sh[wv], lane, rsa, rsb);
  QG_STAMP(0, 2);
  const double *raw = reinterpret_cast<const double *>(Fsh[wv]);
  constexpr int NP = N + N / 16;
  {
    constexpr int NU = (N / 2 + 63) / 64;
    // all LDS reads of the two output rows first, then the global stores (one exposed LDS round trip instead of 2 NU)
    double2 oa[NU], ob[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int t = lane + 64 * u;
      const int tc = t < N / 2 ? t : N / 2 - 1;
      const int i = 2 * tc + ((2 * tc) >> 4); // 2t and 2t+1 share a 16-group: consecutive after padding
      oa[u] = double2{raw[i], raw[i + 1]};
      ob[u] = double2{raw[NP + i], raw[NP + i + 1]};
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int t = lane + 64 * u;
      if (t < N / 2) { // (write-through: qgcm_dev.h)
        qg_store16_wt(rowa + 2 * t, oa[u].x, oa[u].y);
        if (has_b) qg_store16_wt(rowb + 2 * t, ob[u].x, ob[u].y);
      }
    }
  }
  QG_STAMP(0, 3);
  QG_STAMP_DRAIN();
  QG_STAMP(0, 4);
  if (ROWSUM) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      rsa += __shfl_down(rsa, off);
      rsb += __shfl_down(rsb, off);
    }
    if (lane == 0) {
      P.rowsum[(long)m * ny + (ja - 1)] = rsa;
      if (has_b) P.rowsum[(long)m * ny + ja] = rsb;
    }
  }
}

// back to the library default (-ffp-contract=off) for everything included after this file
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------
// Inverse row transform FUSED with the modes -> layers step (and, with BDY, the boundary PV):
// K3 (inverse) + K7 (+ K8) in one launch.  A workgroup = NL waves = the NL modes of one row pair;
// each wave transforms its mode into LDS as above, then all waves combine the modes point by
// point exactly as k_unpack_box does (same expressions, same order; contraction off), writing the
// new po - the transformed field never goes to HBM (saves 2 passes over wrk and a launch).
// The workgroups of the first / last interior row also write the wall rows (G = 1, nyg).
// Reference: src/ocisubs.F:494-499 (inverse dsint), 377-401 (unpack), src/vorsubs.F:245-388.
// grid: (npairs), block 64*NL.  HALO: also write the y-slab halo messages (whole-domain handles compile it out).
// CONSTR: the mass-constraint solve (src/ocisubs.F:349-370) runs here too instead of in a launch of its own
// (k_constr_box: 4-5 us of a 95 us step for 23 KB of work): the workgroup carries one EXTRA wave that forms the area
// integrals from the spectral column sums of k_thomas and solves for hclco - redundantly in every workgroup, hidden
// behind the transform of the other waves (the solve is a 2 us chain of dependent fp64 divides: done inside the
// transforming waves it cost what the launch had saved) - and hands the coefficients over through LDS at the
// barrier that follows the transform.  dpioc has been stepped by k_tend.  Workgroup 0 records xinhom / hclco.
// Same functions as k_constr_box, contraction off: bitwise the same coefficients.
// ---------------------------------------------------------------------------
// AVG: the leapfrog averaging that follows the step (src/q-gcm.F:1345-1351) folded into the stores (QgUnpackParams.pavg /
// qavg) - a template flag, instantiated for the whole-domain step only: the other 24 steps of 25 run the plain code.
template <int M, int NL, bool BDY, bool HALO, bool CONSTR, bool AVG = false>
__global__ __launch_bounds__(64 * (NL + (CONSTR ? 1 : 0))) void k_dst64_unpack(const QgDstParams P, const QgUnpackParams U,
                                                                                      const QgBdyParams B, const QgConstrLite C) {
  constexpr int N = 64 * M, NP = N + N / 16;
  __shared__ __align__(16) cplx Fsh[NL][M * D64_ROW];
  __shared__ __align__(16) cplx W64sh[NL][64];
  __shared__ double hc_sh[NL];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6); // = mode (NL: the constraint wave); scalar: uniform row pointers
  const int ldw = P.g.ldw;
  const int ja = P.g.jr0 + 2 * blockIdx.x;     // local rows ja, ja+1 (grid is exactly the pairs)
  const bool has_b = (ja + 1 <= P.g.jr1);
  const long fs = U.g.fstride;
  const int nx = U.g.nx, nyg = U.g.nyg, joff = U.g.joff;
  constexpr int NX = N + 1;
  double hc[NL];
  // y-slab halo messages (k_halo_pack's layout: [k][3 rows][ldx] of p, then [k][ldx] of q): the first / last three
  // owned rows go to the lower / upper neighbour straight from here (no separate pack launch)
  const int jlo = U.g.jlo, jhi = U.g.jhi, ldxm = U.g.ldx;
  auto msg_p = [&](int gi, int gj, const double *pl) {
    if (HALO && U.msg_lo && gj - jlo < 3) {
#pragma unroll
      for (int k = 0; k < NL; ++k) U.msg_lo[((long)k * 3 + (gj - jlo)) * ldxm + (gi - 1)] = pl[k];
    }
    if (HALO && U.msg_hi && jhi - gj < 3) {
#pragma unroll
      for (int k = 0; k < NL; ++k) U.msg_hi[((long)k * 3 + (gj - (jhi - 2))) * ldxm + (gi - 1)] = pl[k];
    }
  };
  auto msg_q = [&](int gi, int gj, int k, double q) {
    if (HALO && U.msg_lo && gj == jlo) U.msg_lo[((long)NL * 3 + k) * ldxm + (gi - 1)] = q;
    if (HALO && U.msg_hi && gj == jhi) U.msg_hi[((long)NL * 3 + k) * ldxm + (gi - 1)] = q;
  };
  // fused leapfrog averaging: the value stored for the new po at field offset idx
  auto avg_p = [&](long idx, double v) { return AVG ? 0.5 * (v + U.pavg[idx]) : v; };
  // unpack_point of k_misc.h with the transformed rows taken from LDS; sel: 0 row a, 1 row b, -1 wall row;
  // ocv: prefetched ochom values of the point, or nullptr (read them here)
  auto point = [&](int gi, int gj, int sel, const double *ocv, double *pl) {
    const long o = (long)(gj - 1) * U.g.ldx + (gi - 1);
    const bool inner = (sel >= 0 && gi >= 2 && gi <= nx - 1);
    const int ip = (gi - 2) + ((gi - 2) >> 4) + (sel > 0 ? NP : 0);
    double pm[NL];
#pragma unroll
    for (int m = 0; m < NL; ++m) {
      const double wvv = inner ? reinterpret_cast<const double *>(Fsh[m])[ip] : 0.0;
      pm[m] = (m == 0) ? wvv : wvv + hc[m] * (ocv ? ocv[m - 1] : U.ochom[fs * (m - 1) + o]);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double v = 0.0;
#pragma unroll
      for (int m = 0; m < NL; ++m) v = v + U.ctm2l[m + NL * k] * pm[m];
      pl[k] = v;
    }
  };
  // boundary PV of one wall point (k_unpack_box / k_ocqbdy): by = beta*yporel(j), dd = ddynoc(i,j)
  auto bdy_q = [&](long o, const double *pl, const double *pin, double by, double dd, int gi, int gj) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      double ap;
      if (k == 0) ap = B.f0A[0] * pl[0] + B.f0A[NL] * pl[1];
      else if (k == NL - 1) ap = B.f0A[k + NL * (k - 1)] * pl[k - 1] + B.f0A[k + NL * k] * pl[k];
      else ap = B.f0A[k + NL * (k - 1)] * pl[k - 1] + B.f0A[k + NL * k] * pl[k] + B.f0A[k + NL * (k + 1)] * pl[k + 1];
      double q = B.bcfaco_f0 * (pin[k] - pl[k]) - ap + by;
      if (k == NL - 1) q = q + dd;
      if (AVG) q = 0.5 * (q + U.qavg[fs * k + o]);
      B.qo[fs * k + o] = q;
      msg_q(gi, gj, k, q);
    }
  };
  // The two wall columns (W: side 0, E: side 1) of the row pair: one thread per side.  Everything they read from
  // global memory is requested BEFORE the transform (wall_prefetch), used after it (wall_columns).
  struct WallPre {
    double ocw[2][NL - 1], ocn[2][NL - 1], byw[2], ddw[2];
  };
  auto wall_prefetch = [&](bool mine, int side, WallPre &w) {
    const int gw = (side == 0) ? 1 : NX, gn = (side == 0) ? 2 : NX - 1; // wall column and its inward neighbour
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const bool ok = mine && (r == 0 || has_b);
      const long ow = (long)(ja + r - 1) * U.g.ldx + (gw - 1), on = (long)(ja + r - 1) * U.g.ldx + (gn - 1);
#pragma unroll
      for (int m = 1; m < NL; ++m) {
        w.ocw[r][m - 1] = ok ? U.ochom[fs * (m - 1) + ow] : 0.0;
        w.ocn[r][m - 1] = ok ? U.ochom[fs * (m - 1) + on] : 0.0;
      }
      w.byw[r] = (BDY && ok) ? B.beta * B.yporel[ja + r - 1] : 0.0;
      w.ddw[r] = (BDY && ok) ? B.ddynoc[ow] : 0.0;
    }
  };
  auto wall_columns = [&](int side, const WallPre &w) {
    const int gw = (side == 0) ? 1 : NX, gn = (side == 0) ? 2 : NX - 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && !has_b) break;
      const int gj = ja + r;
      const long o = (long)(gj - 1) * U.g.ldx + (gw - 1);
      double pl[NL];
      point(gw, gj, r, w.ocw[r], pl);
#pragma unroll
      for (int k = 0; k < NL; ++k) U.pnew[fs * k + o] = avg_p(fs * k + o, pl[k]);
      msg_p(gw, gj, pl);
      if (BDY || side == 0) {
        double pin[NL];
        point(gn, gj, r, w.ocn[r], pin);
        if (BDY) bdy_q(o, pl, pin, w.byw[r], w.ddw[r], gw, gj);
        if (side == 0) {
          // column 2: the interior columns are combined in 16-byte pairs (3,4) .. (nx-2, nx-1) - this one is left over
          const long on = (long)(gj - 1) * U.g.ldx + (gn - 1);
#pragma unroll
          for (int k = 0; k < NL; ++k) U.pnew[fs * k + on] = avg_p(fs * k + on, pin[k]);
          msg_p(gn, gj, pin);
          if (HALO && ((U.msg_lo && gj == jlo) || (U.msg_hi && gj == jhi))) {
#pragma unroll
            for (int k = 0; k < NL; ++k) msg_q(gn, gj, k, B.qo[fs * k + on]);
          }
        }
      }
    }
  };

  if (CONSTR && wv == NL) {
    // ---- the constraint wave: solve, hand hclco over through LDS, then (after the transform) the two wall columns -
    // a separate code path, so that the prefetched wall values do not occupy registers of the transforming waves
    WallPre w;
    wall_prefetch(lane < 2, lane, w);
    double xin[NL], dpn[NL - 1], x[NL - 1];
    constr_xin<NL>(C, lane, xin);
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) dpn[k] = C.sc->dpioc[k];
    constr_box_solve<NL>(C, xin, dpn, x);
    if (lane == 0) {
#pragma unroll
      for (int m = 1; m < NL; ++m) hc_sh[m] = x[m - 1];
      if (blockIdx.x == 0) {
#pragma unroll
        for (int m = 0; m < NL; ++m) C.sc->xinhom[m] = xin[m];
#pragma unroll
        for (int k = 0; k < NL - 1; ++k) C.sc->hclco[k] = x[k];
      }
    }
    __syncthreads(); // the barrier the transforming waves reach after the transform
#pragma unroll
    for (int m = 1; m < NL; ++m) hc[m] = x[m - 1];
    if (lane < 2) wall_columns(lane, w);
    return;
  }
  // Work split of the combine step: the nx-2 interior columns go round-robin over the threads (NIT
  // full sweeps, nx = N+1); the two wall columns are done by the constraint wave, or without one by thread 0 of the
  // waves of modes 0 (W) and 1 (E).
  // Everything the combine step reads from global memory is requested before the back half of the transform - after
  // the front half, whose row loads need the registers (one wave per mode), so that the kernel keeps two waves per SIMD.
  constexpr int NT = 64 * NL, NPAIR = (NX - 3) / 2, NIT = (NPAIR + NT - 1) / NT;
  const double *rowa = P.wrk + P.g.wstride * wv + (long)(ja - 1) * ldw;
  QG_STAMP(2, 0);
  dst64_front<M>(P, rowa, rowa + ldw, has_b, Fsh[wv], W64sh[wv], lane);
  QG_STAMP(2, 1);
  asm volatile("" ::: "memory"); // keep the prefetch below the front half
  // (column pairs (3,4), (5,6) .. (nx-2, nx-1): 16-byte aligned in the field arrays - 16-byte loads of ochom and
  //  16-byte write-through stores of the new po; column 2 goes with the W wall column)
  double2 oc[2][NIT][NL - 1];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = tid + it * NT;
      const int gi = 3 + 2 * q;
      const bool ok = q < NPAIR && (r == 0 || has_b);
      const long o = (long)(ja + r - 1) * U.g.ldx + (gi - 1);
#pragma unroll
      for (int m = 1; m < NL; ++m) oc[r][it][m - 1] = ok ? *reinterpret_cast<const double2 *>(U.ochom + fs * (m - 1) + o) : double2{0.0, 0.0};
    }
  // AVG: this step's po of the same column pairs, requested here as well (in the store loop the loads sat between the
  // write-through stores: one exposed round trip per round, +12 us on the averaging step's launch)
  double2 pcv[AVG ? 2 : 1][AVG ? NIT : 1][AVG ? NL : 1];
  if (AVG) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = tid + it * NT, qc = q < NPAIR ? q : 0;
        const long o = (long)(ja + ((r == 0 || has_b) ? r : 0) - 1) * U.g.ldx + (3 + 2 * qc - 1);
#pragma unroll
        for (int k = 0; k < NL; ++k) pcv[AVG ? r : 0][AVG ? it : 0][AVG ? k : 0] = *reinterpret_cast<const double2 *>(U.pavg + fs * k + o);
      }
  }
  const bool wallcol = !CONSTR && (lane == 0 && wv < 2);
  WallPre wpre;
  if (!CONSTR) wall_prefetch(wallcol, wv, wpre);
  if (!CONSTR) {
#pragma unroll
    for (int m = 1; m < NL; ++m) hc[m] = U.sc->hclco[m - 1];
  }
  {
    double rsa, rsb;
    dst64_back<M>(Fsh[wv], W64sh[wv], lane, rsa, rsb);
  }
  QG_STAMP(2, 2);
  __syncthreads();
  QG_STAMP(2, 3);
  if (CONSTR) {
#pragma unroll
    for (int m = 1; m < NL; ++m) hc[m] = hc_sh[m];
  }
  if (!CONSTR) {
    if (wallcol) wall_columns(wv, wpre);
  }
  // interior columns
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && !has_b) break;
    const int gj = ja + r;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = tid + it * NT;
      if (q >= NPAIR) break;
      const int gi = 3 + 2 * q;
      const long o = (long)(gj - 1) * U.g.ldx + (gi - 1);
      double oca[NL - 1], ocb[NL - 1], pla[NL], plb[NL];
#pragma unroll
      for (int m = 1; m < NL; ++m) {
        oca[m - 1] = oc[r][it][m - 1].x;
        ocb[m - 1] = oc[r][it][m - 1].y;
      }
      point(gi, gj, r, oca, pla);
      point(gi + 1, gj, r, ocb, plb);
      if (AVG) {
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          const double2 pc = pcv[AVG ? r : 0][AVG ? it : 0][AVG ? k : 0];
          qg_store16_wt(U.pnew + fs * k + o, 0.5 * (pla[k] + pc.x), 0.5 * (plb[k] + pc.y));
        }
      } else {
#pragma unroll
        for (int k = 0; k < NL; ++k) qg_store16_wt(U.pnew + fs * k + o, pla[k], plb[k]);
      }
      msg_p(gi, gj, pla);
      msg_p(gi + 1, gj, plb);
      if (HALO && ((U.msg_lo && gj == jlo) || (U.msg_hi && gj == jhi))) { // interior columns of the q row: set by k_tend
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          msg_q(gi, gj, k, B.qo[fs * k + o]);
          msg_q(gi + 1, gj, k, B.qo[fs * k + o + 1]);
        }
      }
    }
  }
  QG_STAMP(2, 4);
  QG_STAMP_DRAIN();
  QG_STAMP(2, 5);
  // wall rows of the basin (G = 1, nyg): done by the workgroup of the first / last interior row
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && !has_b) break;
    const int gj = ja + r, G = gj + joff;
    const int wall = (G == 2) ? gj - 1 : (G == nyg - 1 ? gj + 1 : 0);
    if (!wall) continue;
    // A fixed number of rounds with a predicate, everything a round reads from global memory requested first: as a loop
    // over gi (trip count depends on the thread: it stays rolled) this was one memory round trip per round, on the two
    // workgroups that finish last anyway.
    constexpr int NRW = (NX + NT - 1) / NT;
    double ocw[NRW][NL - 1], oci[NRW][NL - 1], ddw[NRW];
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
      const int gi = tid + 1 + it * NT, gic = gi <= nx ? gi : 1;
      const long ow = (long)(wall - 1) * U.g.ldx + (gic - 1), oi = (long)(gj - 1) * U.g.ldx + (gic - 1);
#pragma unroll
      for (int m = 1; m < NL; ++m) {
        ocw[it][m - 1] = U.ochom[fs * (m - 1) + ow];
        oci[it][m - 1] = BDY ? U.ochom[fs * (m - 1) + oi] : 0.0;
      }
      ddw[it] = BDY ? B.ddynoc[ow] : 0.0;
    }
    const double byw = BDY ? B.beta * B.yporel[wall - 1] : 0.0;
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
      const int gi = tid + 1 + it * NT;
      if (gi <= nx) {
        const long ow = (long)(wall - 1) * U.g.ldx + (gi - 1);
        double pw[NL], pl[NL];
        point(gi, wall, -1, ocw[it], pw);
#pragma unroll
        for (int k = 0; k < NL; ++k) U.pnew[fs * k + ow] = avg_p(fs * k + ow, pw[k]);
        if (BDY) {
          point(gi, gj, r, oci[it], pl);
          bdy_q(ow, pw, pl, byw, ddw[it], gi, wall);
        }
      }
    }
  }
}
