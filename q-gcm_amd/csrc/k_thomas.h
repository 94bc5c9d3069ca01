// k_thomas.h - K4: batched constant-coefficient tridiagonal solves along y.
//
// Reference: for every wavenumber i the system
//     aoc*u(j-1) + boc(i)*u(j) + aoc*u(j+1) = rhs(i,j),  j = 2..nypo-1
// is solved by the Thomas algorithm (src/ocisubs.F:470-488 box, 575-593
// cyclic) and scaled by ftnorm.  The pivots betinv(j) = 1/(boc - aoc*gam(j))
// depend only on (boc, j), and the recurrence reaches a bitwise fixed point after
// a few rows (median 15 at 5 km; the lowest wavenumbers of the barotropic mode
// never do).  The host runs the recurrence once (same operations and rounding as the
// reference) and tabulates, per block of 16 wavenumbers, the pivots of the rows before
// the block's last wavenumber has become stationary (< 1 MB at 5 km) plus the
// stationary value per wavenumber: no divide is left in the kernel.  (Round 1
// re-ran the recurrence per thread: the 16 IEEE divides of the never-stationary
// block made that workgroup - and with it the launch - 4 us longer than the rest.)
//
// Parallel formulation: both sweeps are first-order linear recurrences
//     forward : u_r = (w_r - aoc*u_{r-1}) * bet_r     = fma(-g_r, u_{r-1}, w_r*bet_r),  g_r = aoc*bet_r
//     backward: v_r = u_r - aoc*bet_r * v_{r+1}       = fma(-g_r, v_{r+1}, u_r)
// (one fused multiply-add per row and sweep: the chain of dependent operations per row is 1 instead of 3, 16 -> 10
// VALU instructions per row in all, and with w*bet and g held in place of w and bet the kernel needs 92 VGPRs
// instead of 124.  The reference's rounding sequence is not kept - no chunked evaluation can keep it - the
// difference stays at the rounding level: tests/test_gpu_parity.py compares with the reference at 1e-12.)
// The rows are cut into 64 chunks of R rows; lanes hold 16 consecutive
// wavenumbers (one 128-B line) x 4 chunks per wave, a workgroup of 1024
// threads covers a whole column block.  Each thread keeps its R rows in
// registers, computes the chunk's affine map (C, D) with zero inflow, the
// maps are composed by a wave scan, and the sweep is re-run from the true
// inflow - so each row is read once and written once (PHASE 0).
//
// y-slab decomposition: the same idea one level up, see the PHASE list below: each
// rank reduces its slab to four numbers per wavenumber, ONE all-gather of
// 4*nk*nl doubles replaces the all-to-all transposes of the whole array.
//
// Algorithmic traffic: read w + write u (16 B per point) for PHASE 0.
#pragma once
#include "qgcm_dev.h"
#include "k_cyclic.h" // constr_cyc_partA

#define TH_KW 16   // wavenumbers per workgroup (128-B line)
#define TH_NC 64   // chunks per column
#define TH_NT (TH_KW * TH_NC)
static_assert(TH_NC == 64, "the chunk scan maps the 64 chunks of a wavenumber to the lanes of one wave");

// Inclusive scan of affine maps over the 64 lanes of a wave (lane = position in
// sweep order).  On return (Cs, Ds) is the composition of positions 0..lane.
// DPP moves instead of ds_bpermute shuffles (VALU only, a few cycles instead of an LDS-pipe round trip per step):
// shifts by 1, 2, 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3, then lane 31 into rows
// 2 and 3.  The composition is not commutative: the map taken from the lower lanes is always applied first.
template <int CTRL>
__device__ __forceinline__ double th_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ void th_scan_step(double &Cs, double &Ds, bool take) {
  const double Cp = th_dpp<CTRL>(Cs), Dp = th_dpp<CTRL>(Ds);
  if (take) {
    Cs = Cs + Ds * Cp;
    Ds = Ds * Dp;
  }
}
__device__ __forceinline__ void affine_scan(double &Cs, double &Ds, int lane) {
  const int rl = lane & 15;
  th_scan_step<0x111>(Cs, Ds, rl >= 1);        // row_shr:1
  th_scan_step<0x112>(Cs, Ds, rl >= 2);        // row_shr:2
  th_scan_step<0x114>(Cs, Ds, rl >= 4);        // row_shr:4
  th_scan_step<0x118>(Cs, Ds, rl >= 8);        // row_shr:8
  th_scan_step<0x142>(Cs, Ds, (lane & 16) != 0); // row_bcast:15 -> rows 1 and 3
  th_scan_step<0x143>(Cs, Ds, lane >= 32);     // row_bcast:31 -> rows 2 and 3
}

// PHASE 0  whole column in this handle: zero inflow at both ends, 1 read + 1 write.
// y-slab decomposition (one exchange per step, ONE pass over the work array):
//   u = u0 + uin*P,  v = B(u) + vin*Q  are linear in the values uin / vin that enter
//   the slab from the ranks below / above, so a slab is summarised by
//     Cf  = last value of the zero-inflow forward sweep,
//     Cb  = first value of the zero-inflow backward sweep applied to u0,
//     D   = product of (-a*bet) over the slab (forward AND backward gain),
//     E   = first value of the backward sweep applied to the unit forward response P.
//   D and E do not depend on the right-hand side (PHASE 4 computes them once).
// Area integrals (xintp of the solution, src/ocisubs.F:160 + src/intsubs.f:78-133) are taken in
// spectral space: the solution vanishes on the four walls, so xintp is the plain sum over the
// interior, and the sum over i of sin(k i pi/n) is cot(k pi/2n) for odd k, 0 for even k. Hence
//     xintp(wrk_m) = sum_k wcot(k) * ksum(m,k),   ksum(m,k) = ftnorm * sum_j v_j(k),
// and ksum is a by-product of the backward sweep (no pass over the transformed field, and the
// constraint solve no longer waits for the inverse row transform).  For y-slabs the column sum is
// linear in the inflows too:  sum_j v_j = S0 + uin*SP + vin*SQ  (S0: zero-inflow sum, SP / SQ:
// sums of the unit responses, right-hand-side independent), so every rank forms the basin-wide
// ksum from the one all-gather of the slab summaries - bitwise the same on every rank.
// PHASE 1  the slab's ZERO-INFLOW solution, written in place (1 read + 1 write, the same work as
//          PHASE 0), and (Cf, Cb, S0) per wavenumber published.  Rounds 1-3 published the summary
//          WITHOUT writing and ran both sweeps again from the true inflows after the exchange (a
//          second read + write of the whole array: NAtl 1 km 21 + 47 us per slab for one solve).
//          Now what is left after the exchange is k_thomas_corr below: the full responses to the
//          inflows, uin*Pv_r + vin*Qv_r, are ADDED to the rows they reach - they decay like
//          lambda^distance from the slab's ends, so for all but the lowest wavenumbers that is a few
//          rows (tables of Pv / Qv up to where they drop below 1e-19 of the inflow, built once at
//          set-up from PHASE 4 / 5's own output, QgThomasParams.corr*).
// PHASE 4  set-up: D, E and SP of this slab (input column = 0, unit inflow from below); the
//          response Pv is left in the work array for the table build.
// PHASE 5  set-up: SQ (input column = 0, unit inflow from above); Qv left in the work array.
// per-step message of one rank: TH_MSG doubles (Cf, Cb, S0) at TH_MSG*(m*ldw + k); the right-hand-side independent
// part (D, E, SP, SQ: TH_CST doubles, the layout of slabDE) is exchanged ONCE after set-up (cgath)
// grid: (ceil(nk/16), nlayers)
#define TH_MSG 3
#define TH_CST 4
// CYCA: the instantiation of the zonally cyclic geometries: it leaves the zonal-mean solution next to the boundaries in
// ybnd, and its grid carries one extra workgroup for part A of the cyclic / atmospheric constraint algebra (a template
// flag, not a run-time test: inlined into the plain instantiation the extra code cost it 11 spilled VGPRs at R = 16)
// KW: wavenumbers per workgroup = waves per workgroup.  16 (a 128-byte line per row, 1024 threads, 128 VGPRs) up to
// 16 rows per thread; 8 (512 threads, 256 VGPRs) for the long columns - R >= 20 keeps 4R doubles of rows and pivots in
// registers and spilled 100-800 B per lane under the 128-VGPR limit.
template <int R, int PHASE, bool CYCA = false, int KW = TH_KW>
// (second launch bound = waves per SIMD: 512-thread workgroups of 8 wavenumbers share a CU in pairs at 128 VGPRs)
__global__ __launch_bounds__(KW * TH_NC, (KW == 8 && R <= 16) ? 4 : 1) void k_thomas(const QgThomasParams P) {
  static_assert(TH_KW % KW == 0, "a workgroup's wavenumbers lie inside one block of the pivot tables");
  static_assert(KW % 2 == 0, "the write-through stores pair the lanes of neighbouring wavenumbers");
  // pitch TH_KW + 1: the scans read / write these arrays transposed ([lane][wv]: 64 lanes at a stride of one row);
  // at a pitch of 16 doubles = 128 B every lane hit the same pair of banks (SQ_LDS_BANK_CONFLICT was 55 % of the
  // kernel's LDS cycles)
  __shared__ double sC[TH_NC][KW + 1];
  __shared__ double sD[TH_NC][KW + 1];
  __shared__ double sIn[TH_NC][KW + 1];
  const int tid = threadIdx.x;
  if (CYCA && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup: part A of the cyclic / atmospheric constraint algebra, one wave
    if (P.cycq && blockIdx.y == 0 && tid < 64) { // (cycq == nullptr: a stand-alone solve, part A runs in k_constr_cyc)
      double a4[QG_MAXL], b4[QG_MAXL];
      switch (P.g.nl) {
        case 2: constr_cyc_partA<2>(*P.cycq, tid, a4, b4); break;
        case 3: constr_cyc_partA<3>(*P.cycq, tid, a4, b4); break;
        default: constr_cyc_partA<4>(*P.cycq, tid, a4, b4); break;
      }
    }
    return;
  }
  QG_STAMP(1, 0);
  // KW = 8 (half a 128-byte line per row): the two workgroups that share the lines of a 16-wavenumber block go to the
  // SAME XCD - physical blocks b and b + 8 of every group of 16 (blocks are dealt round-robin over the 8 XCDs: the
  // grid's x extent is a multiple of 8 then) - so that one L2 fetches each line once.
  int bxl = (int)blockIdx.x;
  if (KW == 8) {
    const int nfull = ((int)gridDim.x - (CYCA ? 1 : 0)) / 16 * 16;
    if (bxl < nfull) bxl = (bxl & ~15) + 2 * (bxl & 7) + ((bxl >> 3) & 1);
  }
  const int kk = tid % KW;
  const int c = tid / KW;
  const int lane = tid & 63, wv = tid >> 6;
  const int k = bxl * KW + kk;
  const int kq = bxl * KW + wv; // wavenumber whose chunk maps this wave scans
  const int m = blockIdx.y + P.layer0;
  const int nr = P.g.jr1 - P.g.jr0 + 1; // local rows jr0..jr1  <->  r = 0..nr-1
  const int ldw = P.g.ldw;
  const bool kok = k < P.g.nk;
  const double a = P.aoc;
  const int r0 = c * R;
  const long mk = TH_MSG * ((long)m * ldw + kq);
  const double ft = P.ftnorm;

  // 32-bit element offsets from a uniform base (one scalar pointer + one VGPR per address)
  const double *wbase_c = P.wrk + P.g.wstride * m + (long)(P.g.jr0 - 1) * ldw + bxl * KW;
  double *wbase = const_cast<double *>(wbase_c);
  const unsigned off0 = (unsigned)(r0 * ldw + kk);
  // rows past the end of the slab: PHASE 0 (whole column, zero inflow at both ends) pads them with w = 0, b = 0 - the
  // forward values and the backward values of such rows are exactly 0, nothing of them is stored, and the sweeps need
  // no per-row predicate.  The slab phases publish the value LEAVING the slab, so there padded rows must be the
  // identity map: they keep the predicate.
  constexpr bool PRED = (PHASE != 0);
  double w[R], b[R];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    int r = r0 + t;
    const bool ok = kok && r < nr && PHASE != 4 && PHASE != 5;
    w[t] = ok ? wbase[off0 + (unsigned)(t * ldw)] : 0.0; // (unconditional loads at clamped offsets: 128 VGPRs + spills)
  }
  // pivots of this chunk (src/ocisubs.F:472-477, tabulated by the host): rows below rcb from the block's table,
  // the stationary value after that
  {
    const int tb = m * P.nblk + (bxl * KW) / TH_KW; // the tables are per block of TH_KW wavenumbers
    const int rcb = P.rcb[tb];
    const double binf = kok ? P.binf[(long)m * ldw + k] : 0.0;
#pragma unroll
    for (int t = 0; t < R; ++t) b[t] = binf;
    if (r0 < rcb) {
      const double *tab = P.ptab + (long)P.poff[tb] * TH_KW + (bxl * KW) % TH_KW + kk;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const int r = r0 + t;
        const int rc = r < rcb ? r : rcb - 1;
        const double v = tab[rc * TH_KW];
        b[t] = (r < rcb && kok) ? v : binf;
      }
    }
    if (r0 + R > nr) {
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (r0 + t >= nr) b[t] = 0.0;
    }
  }
  // values entering the slab: none (the inflows of the other ranks are added by k_thomas_corr), or the unit inflows
  // of the set-up phases
  const double uin = (PHASE == 4) ? 1.0 : 0.0, vin = (PHASE == 5) ? 1.0 : 0.0;
  // ---- forward: local affine maps (zero inflow); rows past the slab are the identity (PRED) or the zero map (PHASE 0)
  double C = 0.0, D = 1.0;
#pragma unroll
  for (int t = 0; t < R; ++t) {
    w[t] *= b[t];
    b[t] *= a;
  }
#pragma unroll
  for (int t = 0; t < R; ++t) {
    if (!PRED || r0 + t < nr) {
      C = __builtin_fma(-b[t], C, w[t]);
      D = -b[t] * D;
    }
  }
  QG_STAMP(1, 1);
  sC[c][kk] = C;
  sD[c][kk] = D;
  __syncthreads();
  QG_STAMP(1, 2);
  double Cs = sC[lane][wv], Ds = sD[lane][wv];
  affine_scan(Cs, Ds, lane);
  // lane 63 holds the composition of all chunks: zero-inflow end value and slab gain
  const double Cf_tot = __shfl(Cs, 63), D_tot = __shfl(Ds, 63);
  {
    // value entering chunk `lane` = scanned map of chunk lane-1 applied to uin
    double Cprev = th_dpp<0x138>(Cs), Dprev = th_dpp<0x138>(Ds); // wave_shr:1 (lane 0 is not used)
    sIn[lane][wv] = (lane == 0) ? uin : Cprev + Dprev * uin;
  }
  __syncthreads();
  QG_STAMP(1, 3);
  double u = sIn[c][kk];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    if (!PRED || r0 + t < nr) {
      u = __builtin_fma(-b[t], u, w[t]);
      w[t] = u;
    }
  }
  // ---- backward: v_r = u_r - a*bet_r*v_{r+1} ------------------------------
  // (the chunk gain is formed again, in descending order)
  C = 0.0;
  D = 1.0;
#pragma unroll
  for (int t = R - 1; t >= 0; --t) {
    if (!PRED || r0 + t < nr) {
      C = __builtin_fma(-b[t], C, w[t]);
      D = -b[t] * D;
    }
  }
  sC[c][kk] = C; // safe without a barrier: every wave read the forward maps in sC / sD before the barrier above
  sD[c][kk] = D;
  __syncthreads();
  QG_STAMP(1, 4);
  // sweep order is last chunk first: lane l stands for chunk 63-l
  const int cr = 63 - lane;
  Cs = sC[cr][wv];
  Ds = sD[cr][wv];
  affine_scan(Cs, Ds, lane);
  const double Cb_tot = __shfl(Cs, 63); // first value of the zero-inflow backward sweep
  {
    double Cprev = th_dpp<0x138>(Cs), Dprev = th_dpp<0x138>(Ds); // wave_shr:1 (lane 0 is not used)
    sIn[cr][wv] = (lane == 0) ? vin : Cprev + Dprev * vin;
  }
  __syncthreads();
  QG_STAMP(1, 5);
  double v = sIn[c][kk];
  double colsum = 0.0;
#pragma unroll
  for (int t = R - 1; t >= 0; --t) {
    if (!PRED || r0 + t < nr) {
      v = __builtin_fma(-b[t], v, w[t]);
      w[t] = v;
      colsum += v;
    }
  }
  QG_STAMP(1, 6);
  {
    if (R % 2 == 0) {
      // 16-byte write-through stores (qgcm_dev.h: all of this kernel's stores come at its very end): the lanes of two
      // neighbouring wavenumbers swap one value per pair of rows, the even lane then stores row t for both, the odd
      // lane row t + 1 (columns k - 1, k).  A column past nk is a padding column of the row (ldw = nk rounded up).
      const bool odd = (kk & 1) != 0;
      const bool pok = odd ? (k - 1 < P.g.nk) : kok; // the pair's first column exists
      double *cbase = wbase + (odd ? off0 - 1 : off0);
#pragma unroll
      for (int t = 0; t < R; t += 2) {
        const double mine0 = ft * w[t], mine1 = ft * w[t + 1];
        const double give = odd ? mine0 : mine1;
        int lo = __double2loint(give), hi = __double2hiint(give);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true); // quad_perm [1,0,3,2]
        hi = __builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true);
        const double got = __hiloint2double(hi, lo);
        const int r = r0 + t + (odd ? 1 : 0);
        if (pok && r < nr) qg_store16_wt(cbase + (unsigned)((t + (odd ? 1 : 0)) * ldw), odd ? got : mine0, odd ? mine1 : got);
      }
    } else {
#pragma unroll
      for (int t = 0; t < R; ++t) {
        int r = r0 + t;
        if (kok && r < nr) wbase[off0 + (unsigned)(t * ldw)] = ft * w[t];
      }
    }
  }
  QG_STAMP(1, 7);
  QG_STAMP_DRAIN();
  QG_STAMP(1, 8);
  if (CYCA && PHASE == 0 && P.ybnd && k == 0) { // (cyclic instantiation only: the box kernel has no register to spare)
    // cyclic constraints: the zonal-mean solution next to the two zonal boundaries (rows 2 and nypo-1) goes to a side
    // buffer, so that part B of the constraint algebra does not have to read wrk - which the in-place inverse rows of
    // the generic sizes overwrite - and can ride in that launch
#pragma unroll
    for (int t = 0; t < R; ++t) {
      if (r0 + t == 0) P.ybnd[2 * m] = ft * w[t];
      if (r0 + t == nr - 1) P.ybnd[2 * m + 1] = ft * w[t];
    }
  }
  // column sum of this slab: chunks in a fixed order (lanes of wave wv = chunks of wavenumber kq)
  sC[c][kk] = colsum;
  __syncthreads();
  double tot = sC[lane][wv];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
  if (lane != 0 || kq >= P.g.nk) return;
  double *cst = P.slabDE + 4 * ((long)m * ldw + kq); // D, E, SP, SQ of this slab
  if (PHASE == 0) {
    P.ksum[(long)m * ldw + kq] = ft * tot;
  } else if (PHASE == 1) {
    P.send[mk] = Cf_tot;
    P.send[mk + 1] = Cb_tot;
    P.send[mk + 2] = tot;
  } else if (PHASE == 4) {
    cst[0] = D_tot;
    cst[1] = Cb_tot;
    cst[2] = tot;
  } else {
    cst[3] = tot;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// y-slabs, after the exchange of the slab summaries: k_thomas_corr.
//   1. composes all ranks' summaries (lanes = ranks, two affine scans per wavenumber) into the values uin / vin that
//      enter THIS slab from below / above, the basin-wide column sums ksum (bitwise the same on every rank) and, for
//      the zonally cyclic oceans, the zonal-mean solution next to the two zonal boundaries (ybnd);
//   2. adds the slab's full responses to those inflows to the zero-inflow solution PHASE 1 left in the work array:
//          v_r += uin * Pv_r  (rows r < np of the block)  +  vin * Qv_r  (the last nq rows),
//      Pv / Qv from the tables built at set-up.  Everything is scaled by ftnorm, as the stored rows are.
// Workgroup = one block of TH_KW wavenumbers (a 128-byte line per row): 256 threads; in step 2 a thread owns a pair
// of neighbouring wavenumbers (16-byte loads / stores) in every 32nd row.  grid: (ceil(nk/16), nlayers).
// The rows are independent of one another here - no sweep, no scan, no divide: the pass is bound by the bytes it
// touches, 2 x 8 B per corrected element + 8 B of table (NAtl 1 km, 600-row slabs: ~ 20 % of the rows on average).
struct QgThomasCorr {
  const double *ptab, *qtab; // responses, TH_KW doubles per row: block b of layer m starts at row poff / qoff of its table
  const int *np, *nq;        // (nblk, nlayers) rows within reach of the lower / upper inflow (0: no neighbour on that side)
  const int *poff, *qoff;
};
#define THC_NT 256
#ifndef THC_UNR
#define THC_UNR 4 // rows in flight per thread and trip of the correction loop
#endif
#define THC_PER ((TH_MSG + TH_CST) * TH_KW) // doubles per rank and block: summaries + constants of 16 wavenumbers
#define THC_LDS (THC_PER + TH_KW + 1)       // ... + the inflow per wavenumber (+ 1: bank spread between ranks)
__global__ __launch_bounds__(THC_NT) void k_thomas_corr(const QgThomasParams P, const QgThomasCorr T) {
  // 256 threads: four workgroups share a CU (the pass is all memory latency).  Every thread first requests its first
  // rows and table entries, then the workgroup fetches all ranks' summaries of the block's 16 wavenumbers into LDS in
  // one round trip - the two latencies overlap instead of adding - and 16 lanes run the two short chains of
  // multiply-adds over the ranks.  (Measured, NAtl 1 km, 600-row slabs: 1024-thread workgroups
  // composing with one wave scan per wavenumber 20.8 us; 256 threads, four scans per wave one after the other: 24.5 us.)
  __shared__ double sU[TH_KW], sV[TH_KW];
  extern __shared__ double sG[]; // dynamic: nranks x (THC_PER doubles of summaries and constants + TH_KW inflows)
  const int tid = threadIdx.x;
  const int bx = blockIdx.x, m = blockIdx.y + P.layer0;
  const int ldw = P.g.ldw;
  const double ft = P.ftnorm;
  const int tb = m * P.nblk + bx;
  const int np = T.np[tb], nq = T.nq[tb];
  const int nr = P.g.jr1 - P.g.jr0 + 1;
  const int kp = tid & 7, c = tid >> 3; // pair of wavenumbers 2 kp, 2 kp + 1 of the block; rows c, c + 32, ...
  constexpr int STEP = THC_NT / 8;
  const bool pair_ok = bx * TH_KW + 2 * kp < P.g.nk; // (the second of a pair may be the row's padding column: ldw > nk)
  double *wb = P.wrk + P.g.wstride * m + (long)(P.g.jr0 - 1) * ldw + bx * TH_KW + 2 * kp;
  const double *pt = T.ptab + (long)T.poff[tb] * TH_KW + 2 * kp;
  const double *qt = T.qtab + (long)T.qoff[tb] * TH_KW + 2 * kp;
  const int q0 = nr - nq; // first row the upper inflow reaches (table row r - q0)
  // the rows to correct: [0, np) and [max(np, q0), nr), as one index space of n1 + n2 rows
  const int n1 = np, s2 = np > q0 ? np : q0, ntot = n1 + (nr - s2);
  double2 w[THC_UNR], a[THC_UNR], b[THC_UNR];
  auto row_of = [&](int x) { return x < n1 ? x : s2 + (x - n1); };
  auto request = [&](int x0) {
#pragma unroll
    for (int i = 0; i < THC_UNR; ++i) {
      const int x = x0 + i * STEP;
      const int r = row_of(x < ntot ? x : ntot - 1); // clamped: unconditional loads
      w[i] = *reinterpret_cast<const double2 *>(wb + (long)r * ldw);
      a[i] = double2{0.0, 0.0};
      b[i] = double2{0.0, 0.0};
      if (r < np) a[i] = *reinterpret_cast<const double2 *>(pt + (long)r * TH_KW);
      if (r >= q0 && nq > 0) b[i] = *reinterpret_cast<const double2 *>(qt + (long)(r - q0) * TH_KW);
    }
  };
  // blockIdx.z: slice of THC_UNR * STEP = 128 of those rows (one trip of the loop below per workgroup: the blocks of
  // the lowest wavenumbers, whose responses reach through the whole slab, otherwise run 5 trips = 5 memory round trips
  // one after the other while everybody else has long finished - measured 22.7 us per 600-row slab of NAtl 1 km)
  const int xa = blockIdx.z * (THC_UNR * STEP), xb0 = xa + THC_UNR * STEP, xb = xb0 < ntot ? xb0 : ntot;
  if (blockIdx.z > 0 && xa >= ntot) return; // (uniform; slice 0 always composes: it owns ksum / ybnd)
  const bool work = xa < ntot && pair_ok;
  if (work) request(xa + c);
  // every rank's summaries (Cf, Cb, S0) and constants (D, E, SP, SQ) of the block's 16 wavenumbers: 48 + 64 contiguous
  // doubles per rank, fetched by the whole workgroup in one round trip
  {
    const long mk0 = TH_MSG * ((long)m * ldw + bx * TH_KW), ck0 = TH_CST * ((long)m * ldw + bx * TH_KW);
    const long cstride = (long)TH_CST * P.g.nl * ldw;
    // (four loads in flight per trip: a rolled loop waits for every load before it requests the next - with eight ranks
    //  that was four memory round trips one after the other in every workgroup's latency chain, now one)
    const int total = P.nranks * THC_PER;
    for (int it0 = tid; it0 < total; it0 += 4 * THC_NT) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * THC_NT, itc = it < total ? it : 0;
        const int sr = itc / THC_PER, jj = itc - sr * THC_PER;
        v[u] = jj < TH_MSG * TH_KW ? P.gath[(long)sr * P.gath_stride + mk0 + jj] : P.cgath[(long)sr * cstride + ck0 + (jj - TH_MSG * TH_KW)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * THC_NT;
        if (it < total) {
          const int sr = it / THC_PER, jj = it - sr * THC_PER;
          sG[sr * THC_LDS + jj] = v[u];
        }
      }
    }
  }
  __syncthreads();
  if (tid < TH_KW) {
    // lane = wavenumber kq of the block; ranks s = 0 .. nranks-1 in turn (operands in LDS: two short chains)
    const int kq = bx * TH_KW + tid;
    const double *gk = sG + TH_MSG * tid, *ck = sG + TH_MSG * TH_KW + TH_CST * tid;
    // forward chain: value leaving rank s = Cf_s + D_s * (value entering rank s), nothing enters rank 0
    double u = 0.0, mine_u = 0.0, vlast = 0.0;
    for (int sr = 0; sr < P.nranks; ++sr) {
      sG[sr * THC_LDS + THC_PER + tid] = u;
      if (sr == P.rank) mine_u = u;
      u = __builtin_fma(ck[sr * THC_LDS], u, gk[sr * THC_LDS]);
      vlast = u;
    }
    // backward chain, last rank first: value leaving rank s downwards = Cb_s + E_s*uin_s + D_s * (value entering from above)
    double v = 0.0, mine_v = 0.0, term = 0.0, vfirst = 0.0;
    for (int sr = P.nranks - 1; sr >= 0; --sr) {
      const double *gs = gk + sr * THC_LDS, *gc = ck + sr * THC_LDS;
      const double us = sG[sr * THC_LDS + THC_PER + tid];
      if (sr == P.rank) mine_v = v;
      term += gs[2] + us * gc[2] + v * gc[3]; // column sum of rank s: S0 + uin*SP + vin*SQ
      v = __builtin_fma(gc[0], v, gs[1] + gc[1] * us);
      vfirst = v;
    }
    sU[tid] = ft * mine_u;
    sV[tid] = ft * mine_v;
    if (kq < P.g.nk && blockIdx.z == 0) P.ksum[(long)m * ldw + kq] = ft * term;
    if (P.ybnd && kq == 0 && blockIdx.z == 0) {
      // cyclic constraints: the zonal-mean solution next to the two zonal boundaries, on every rank, from the
      // summaries: the value leaving rank 0 downwards is the solution at its first row (global row 2); the forward
      // value leaving the last rank is the solution at its last row (global row nyg-1: nothing enters from above)
      P.ybnd[2 * m] = ft * vfirst;
      P.ybnd[2 * m + 1] = ft * vlast;
    }
  }
  if (xa >= ntot) return; // (uniform: no barrier is skipped by part of the workgroup)
  __syncthreads();
  if (!pair_ok) return;
  const double u0 = sU[2 * kp], u1 = sU[2 * kp + 1], v0 = sV[2 * kp], v1 = sV[2 * kp + 1];
  {
    const int x0 = xa + c;
#pragma unroll
    for (int i = 0; i < THC_UNR; ++i) {
      const int x = x0 + i * STEP;
      if (x < xb) {
        const double xx = __builtin_fma(v0, b[i].x, __builtin_fma(u0, a[i].x, w[i].x));
        const double yy = __builtin_fma(v1, b[i].y, __builtin_fma(u1, a[i].y, w[i].y));
        qg_store16_wt(wb + (long)row_of(x) * ldw, xx, yy);
      }
    }
  }
}

// set-up of the tables: how far a unit inflow reaches into the slab.  The work array holds PHASE 4's (from_top = 0:
// the response decays upwards from row 0) or PHASE 5's (from_top = 1: downwards from the last row) output, scaled by
// ftnorm; per block of TH_KW wavenumbers the number of rows up to the last one with an entry above tol.
// grid: (nblk, nlayers), 256 threads
__global__ __launch_bounds__(256) void k_thomas_reach(const QgThomasParams P, int from_top, double tol, int *reach) {
  __shared__ int smax[256];
  const int tid = threadIdx.x, kk = tid % TH_KW, c = tid / TH_KW;
  const int bx = blockIdx.x, m = blockIdx.y;
  const int nr = P.g.jr1 - P.g.jr0 + 1;
  const int k = bx * TH_KW + kk;
  const double *wb = P.wrk + P.g.wstride * m + (long)(P.g.jr0 - 1) * P.g.ldw + k;
  int far = 0; // rows counted from the end the inflow enters at
  if (k < P.g.nk)
    for (int r = c; r < nr; r += 256 / TH_KW) {
      const double v = wb[(long)r * P.g.ldw];
      if (!(fabs(v) <= tol)) { // (a NaN counts as "reaches")
        const int d = from_top ? nr - r : r + 1;
        far = d > far ? d : far;
      }
    }
  smax[tid] = far;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) smax[tid] = smax[tid] > smax[tid + s] ? smax[tid] : smax[tid + s];
    __syncthreads();
  }
  if (tid == 0) reach[m * P.nblk + bx] = smax[0];
}

// ... and the copy of those rows into the packed table, divided by ftnorm (the stored rows carry it, the tables do
// not: k_thomas_corr multiplies the inflows by it).  grid: (nblk, nlayers), 256 threads
__global__ __launch_bounds__(256) void k_thomas_pack(const QgThomasParams P, int from_top, const int *reach, const int *off,
                                                     double *tab) {
  const int tid = threadIdx.x, kk = tid % TH_KW, c = tid / TH_KW;
  const int bx = blockIdx.x, m = blockIdx.y;
  const int nr = P.g.jr1 - P.g.jr0 + 1;
  const int k = bx * TH_KW + kk;
  const int n = reach[m * P.nblk + bx];
  const int r0 = from_top ? nr - n : 0;
  const double *wb = P.wrk + P.g.wstride * m + (long)(P.g.jr0 - 1) * P.g.ldw + k;
  double *t = tab + (long)off[m * P.nblk + bx] * TH_KW + kk;
  for (int r = c; r < n; r += 256 / TH_KW) t[(long)r * TH_KW] = (k < P.g.nk) ? wb[(long)(r0 + r) * P.g.ldw] / P.ftnorm : 0.0;
}
