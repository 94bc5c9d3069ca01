// k_dst.h - K3: per-row DST-I (box ocean) in LDS, replacing FFTPACK dsint.
//
// Reference: hsbxoc calls dsint(nxto-1, wrk(2,j)) for every row j=2..nypo-1,
// once forward and once inverse per mode per step (src/ocisubs.F:461-463,
// 494-499); dsint = pre-twiddle + length-(n+1) real FFT + running-sum
// post-process (src/fftpack/newbihar/dsint.f:16-40, dsinti.f:15-25).
//
// Here one workgroup transforms TWO rows at once: the two pre-twiddled real
// sequences of length N = n+1 = nxto are packed as real/imaginary parts of one
// complex sequence, a Stockham mixed-radix (8,4,2,3,5) complex FFT of length N
// runs in LDS, the two real spectra are separated by conjugate symmetry and
// the FFTPACK post-process (including its running sum, done as a block scan)
// produces both sine transforms.  The transform is its own inverse up to
// 2N, absorbed by ftnorm in the Thomas kernel (ocisubs.F:440,486).
//
// Algorithmic traffic: one read + one write of the row (8 B each per point).
#pragma once
#include "qgcm_dev.h"

#define DST_NT 128
#define DST_NT_BIG 512
#define DST_BIG_N 2048 // rows at least this long use DST_NT_BIG threads
// single-buffer (in-place) stages for rows whose two ping-pong buffers would not let two workgroups share a CU:
// 2*N*16 B > 80 KB, N*16 B + 256 B <= 80 KB, and N / (R * 512) within the MAXIT bounds of the kernels
#define DST_SINGLE_MINN 2561
#define DST_SINGLE_MAXN 5104

struct cplx {
  double x, y;
};

// Where the FFT part of the row kernels keeps element n of its input / element k of its spectrum in LDS.  The
// Stockham stages below are autosort (natural order); the three-stage plans of k_fft3.h are padded and digit-reversed.
struct FftPlanNatural {
  static constexpr bool three_stage = false;
  static constexpr int N = 0; // run-time length (QgDstParams.N)
  struct Tw {};
  template <int NT>
  static __device__ __forceinline__ Tw prefetch(const double2 *, int) { return Tw{}; }
  static __device__ __forceinline__ int pos_in(int n) { return n; }
  static __device__ __forceinline__ int pos_out(int k) { return k; }
};
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
// (the row transforms are not required to be bitwise FFTPACK: the complex product is two multiplies and two fused
// multiply-adds, contracted by the front end - the same in every kernel it is inlined into - while the rest of the
// library stays at -ffp-contract=off)
#pragma clang fp contract(on)
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
#pragma clang fp contract(off)
// multiply by -i (forward quarter turn)
__device__ __forceinline__ cplx cmni(cplx a) { return {a.y, -a.x}; }

// Everything below is row-transform arithmetic (pre-twiddle, FFT stages, spectrum split, running-sum post-process):
// fused multiply-adds allowed, as in k_dst64.h / k_fft3.h / k_rfft64.h.
#pragma clang fp contract(fast)

// forward 8-point DFT, y[c] = sum_a x[a] exp(-2 pi i a c / 8), in place (radix-8 stage of the generic plan)
__device__ __forceinline__ void dft8_g(cplx *x) {
  const double r2 = 0.70710678118654752440;
  cplx e0 = cadd(x[0], x[4]), o0 = csub(x[0], x[4]);
  cplx e1 = cadd(x[1], x[5]), o1 = csub(x[1], x[5]);
  cplx e2 = cadd(x[2], x[6]), o2 = csub(x[2], x[6]);
  cplx e3 = cadd(x[3], x[7]), o3 = csub(x[3], x[7]);
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

// radix-R butterfly: o[u] = sum_r a[r] exp(-2 pi i r u / R)
template <int R>
__device__ __forceinline__ void dst_bfly(const cplx *a, cplx *o) {
  if (R == 2) {
    o[0] = cadd(a[0], a[1]);
    o[1] = csub(a[0], a[1]);
  } else if (R == 4) {
    cplx t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
    cplx t2 = cadd(a[1], a[3]), t3 = cmni(csub(a[1], a[3]));
    o[0] = cadd(t0, t2);
    o[2] = csub(t0, t2);
    o[1] = cadd(t1, t3);
    o[3] = csub(t1, t3);
  } else if (R == 8) {
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = a[r];
    dft8_g(o);
  } else if (R == 3) {
    const double s3 = 0.86602540378443864676;
    cplx t1 = cadd(a[1], a[2]);
    cplx t2 = {a[0].x - 0.5 * t1.x, a[0].y - 0.5 * t1.y};
    cplx d = csub(a[1], a[2]);
    cplx t3 = {s3 * d.y, -s3 * d.x}; // -i * s3 * d
    o[0] = cadd(a[0], t1);
    o[1] = cadd(t2, t3);
    o[2] = csub(t2, t3);
  } else if (R == 5) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;
    cplx t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
    cplx t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    o[0] = {a[0].x + t1.x + t2.x, a[0].y + t1.y + t2.y};
    cplx m1 = {a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y};
    cplx m2 = {a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y};
    // -i*(s1*t3 + s2*t4), -i*(s2*t3 - s1*t4)
    cplx n1 = {s1 * t3.y + s2 * t4.y, -(s1 * t3.x + s2 * t4.x)};
    cplx n2 = {s2 * t3.y - s1 * t4.y, -(s2 * t3.x - s1 * t4.x)};
    o[1] = cadd(m1, n1);
    o[4] = csub(m1, n1);
    o[2] = cadd(m2, n2);
    o[3] = csub(m2, n2);
  }
}

// One Stockham DIF stage of radix R over the whole length-N sequence:
//   a_r = in[q + s*(p + m*r)],  b_u = sum_r a_r w_R^{ru},  out[q + s*(R*p + u)] = b_u * w_len^{p*u}
template <int R, int NT = DST_NT>
__device__ __forceinline__ void dst_stage(const cplx *__restrict__ in, cplx *__restrict__ out, int N, int s, int m,
                                          const double2 *__restrict__ tw, int twstep, int tid) {
  const int nb = m * s;
  for (int b = tid; b < nb; b += NT) {
    int p = b / s, q = b - p * s;
    cplx a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = in[q + s * (p + m * r)];
    cplx o[R];
    dst_bfly<R>(a, o);
    out[q + s * (R * p)] = o[0];
#pragma unroll
    for (int u = 1; u < R; ++u) {
      double2 w = tw[p * u * twstep];
      out[q + s * (R * p + u)] = cmul(o[u], cplx{w.x, w.y});
    }
  }
}

// The same stage IN PLACE (single LDS buffer): every thread first takes all its butterflies into registers, the
// workgroup synchronises, then the results are written back. Costs one more barrier per stage and
// MAXIT*R complex registers, halves the LDS footprint - a 4608- or 4800-point row pair then leaves room for a
// second workgroup on the CU.  MAXIT >= ceil(N / (R * NT)).
template <int R, int NT, int MAXIT>
__device__ __forceinline__ void dst_stage_ip(cplx *__restrict__ buf, int N, int s, int m, const double2 *__restrict__ tw,
                                             int twstep, int tid) {
  const int nb = m * s;
  cplx o[MAXIT][R];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int b = tid + it * NT;
    if (b < nb) {
      int p = b / s, q = b - p * s;
      cplx a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = buf[q + s * (p + m * r)];
      dst_bfly<R>(a, o[it]);
#pragma unroll
      for (int u = 1; u < R; ++u) {
        double2 w = tw[p * u * twstep];
        o[it][u] = cmul(o[it][u], cplx{w.x, w.y});
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int b = tid + it * NT;
    if (b < nb) {
      int p = b / s, q = b - p * s;
#pragma unroll
      for (int u = 0; u < R; ++u) buf[q + s * (R * p + u)] = o[it][u];
    }
  }
}

// generic odd prime radix (rare: only for grids whose nxto has a factor > 5)
template <int NT = DST_NT>
__device__ __forceinline__ void dst_stage_generic(int R, const cplx *__restrict__ in, cplx *__restrict__ out, int N,
                                                  int s, int m, const double2 *__restrict__ tw, int twstep, int tid) {
  const int nb = m * s;
  const int rstep = N / R;
  for (int b = tid; b < nb; b += NT) {
    int p = b / s, q = b - p * s;
    for (int u = 0; u < R; ++u) {
      cplx acc = {0.0, 0.0};
      for (int r = 0; r < R; ++r) {
        double2 w = tw[((r * u) % R) * rstep];
        acc = cadd(acc, cmul(in[q + s * (p + m * r)], cplx{w.x, w.y}));
      }
      double2 w = tw[p * u * twstep];
      out[q + s * (R * p + u)] = cmul(acc, cplx{w.x, w.y});
    }
  }
}

// grid: (ceil(nrows/2), nlayers);  rows j = jr0..jr1 (owned, interior to the global domain)
// dynamic LDS: 2*N cplx + 2*NT doubles; NT = 128 threads for short rows, DST_NT_BIG for long ones (a long row
// fills the LDS of a CU by itself, so the one resident workgroup must bring enough waves)
template <int NL>
__device__ __forceinline__ void constr_box_all(const QgConstrParams &P, int lane); // k_misc.h

// (three-stage plans at 384 threads: 128 VGPRs, so that two workgroups of six waves fit a CU however their waves fall
// on the four SIMDs - at 134-146 VGPRs the second workgroup did not become resident: FFT3_WPE)
template <bool ROWSUM, int NT = DST_NT, class PLAN = FftPlanNatural>
__global__ __launch_bounds__(NT, (PLAN::three_stage && NT == 384) ? 4 : 1) void k_dst_box(const QgDstParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  if (P.boxq && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup of the inverse rows inside a step: the box constraint solve (one wave; nothing of it
    // depends on the rows, the unpack launch that follows reads its coefficients)
    if (blockIdx.y == 0 && threadIdx.x < 64) {
      switch (P.g.nl) {
        case 2: constr_box_all<2>(*P.boxq, threadIdx.x); break;
        default: constr_box_all<3>(*P.boxq, threadIdx.x); break; // (4 layers: the pivot search indexes at run time -
      }                                                          //  scratch; such oceans keep the k_constr_box launch)
    }
    return;
  }
  const int N = PLAN::three_stage ? PLAN::N : P.N, n = N - 1; // compile-time for the three-stage plans: loops unroll
  const bool single = PLAN::three_stage || P.single != 0; // one LDS buffer, in-place stages (N even, see DST_SINGLE_*)
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  cplx *B = single ? A : A + N;
  double *red; // 2 * (NT / 64) doubles behind the buffer(s)
  if constexpr (PLAN::three_stage) red = reinterpret_cast<double *>(A + PLAN::LDS_CPLX);
  else red = reinterpret_cast<double *>((single ? A : B) + N);
  const int tid = threadIdx.x;
  const int ny = P.g.ny, ldw = P.g.ldw;
  const int m = blockIdx.y + P.layer0;
  const int ja = P.g.jr0 + 2 * blockIdx.x; // first row (1-based local j)
  const bool has_b = (ja + 1 <= P.g.jr1);
  double *rowa = P.wrk + P.g.wstride * m + (long)(ja - 1) * ldw;
  double *rowb = rowa + ldw;
  typename PLAN::Tw tw3 = PLAN::template prefetch<NT>(P.twid, tid); // table values requested before the rows
  QG_STAMP(0, 0);

  // ---- pre-twiddle (dsint.f:19-33): element a[k], k=1..n lives at index k-1
  const int ns2 = n / 2;
  if (tid == 0) A[0] = {0.0, 0.0};
  const double *rbp = has_b ? rowb : rowa; // the odd last row has no partner: loads redirected, values zeroed
  const double bsc = has_b ? 1.0 : 0.0;
  if constexpr (PLAN::three_stage) {
    // compile-time length: a fixed number of rounds with a predicate instead of a loop whose trip count depends on the
    // thread - every load of a thread is in flight at once (the rolled loop paid one memory round trip per few rounds)
    constexpr int NS2 = (PLAN::N - 1) / 2, NIT = (NS2 + NT - 1) / NT;
    double xa[NIT], xac[NIT], xb[NIT], xbc[NIT], sn[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = 1 + tid + it * NT, kc = k <= NS2 ? k : 1;
      xa[it] = rowa[kc - 1];
      xac[it] = rowa[n - kc];
      xb[it] = rbp[kc - 1];
      xbc[it] = rbp[n - kc];
      sn[it] = P.sintab[kc];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int k = 1 + tid + it * NT;
      if (k <= NS2) {
        const double b0 = bsc * xb[it], b1 = bsc * xbc[it];
        const double t1a = xa[it] - xac[it], t2a = sn[it] * (xa[it] + xac[it]);
        const double t1b = b0 - b1, t2b = sn[it] * (b0 + b1);
        A[PLAN::pos_in(k)] = {t1a + t2a, t1b + t2b};
        A[PLAN::pos_in(N - k)] = {t2a - t1a, t2b - t1b};
      }
    }
  } else {
#pragma unroll
  for (int k = 1 + tid; k <= ns2; k += NT) {
    double xa = rowa[k - 1], xac = rowa[n - k];
    double xb = bsc * rbp[k - 1], xbc = bsc * rbp[n - k];
    double sn = P.sintab[k];
    double t1a = xa - xac, t2a = sn * (xa + xac);
    double t1b = xb - xbc, t2b = sn * (xb + xbc);
    A[PLAN::pos_in(k)] = {t1a + t2a, t1b + t2b};
    A[PLAN::pos_in(N - k)] = {t2a - t1a, t2b - t1b};
  }
  }
  if ((n & 1) && tid == 0) {
    int kc = ns2 + 1;
    double xa = rowa[kc - 1], xb = has_b ? rowb[kc - 1] : 0.0;
    A[PLAN::pos_in(kc)] = {4.0 * xa, 4.0 * xb};
  }
  __syncthreads();
  QG_STAMP(0, 1);

  // ---- complex FFT of length N, Stockham autosort ----------------------
  cplx *in = A, *out = B;
  int s = 1, len = N;
  if constexpr (PLAN::three_stage) PLAN::template run<NT>(A, tw3, tid);
  else
  for (int f = 0; f < P.nfac; ++f) {
    const int R = P.fac[f];
    const int mm = len / R;
    const int twstep = N / len;
    if (single) { // NT = DST_NT_BIG here; the MAXIT bounds cover N / (R * NT) for N <= DST_SINGLE_MAXN
      switch (R) {
        case 2: dst_stage_ip<2, NT, 5>(A, N, s, mm, P.twid, twstep, tid); break;
        case 3: dst_stage_ip<3, NT, 4>(A, N, s, mm, P.twid, twstep, tid); break;
        case 4: dst_stage_ip<4, NT, 3>(A, N, s, mm, P.twid, twstep, tid); break;
        case 5: dst_stage_ip<5, NT, 2>(A, N, s, mm, P.twid, twstep, tid); break;
        default: dst_stage_ip<8, NT, 2>(A, N, s, mm, P.twid, twstep, tid); break;
      }
    } else {
      switch (R) {
        case 2: dst_stage<2, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 3: dst_stage<3, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 4: dst_stage<4, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 5: dst_stage<5, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 8: dst_stage<8, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        default: dst_stage_generic<NT>(R, in, out, N, s, mm, P.twid, twstep, tid); break;
      }
    }
    __syncthreads();
    cplx *t = in;
    in = out;
    out = t;
    s *= R;
    len = mm;
  }
  cplx *Z = single ? A : in;
  QG_STAMP(0, 2);

  // ---- separate the two real spectra and post-process (dsint.f:37-44) ----
  //   Y_k  = (Z_k + conj Z_{N-k})/2      (row a)
  //   Y'_k = (Z_k - conj Z_{N-k})/(2i)   (row b)
  //   b[1] = 0.5 Re Y_0 ; b[2k] = -Im Y_k ; b[2k+1] = b[2k-1] + Re Y_k
  const int K = (n - 1) / 2;              // odd outputs b[2k+1], k=1..K
  const int chunk = (K + NT - 1) / NT;
  const int k0 = 1 + tid * chunk;
  const double b1a = 0.5 * Z[0].x, b1b = 0.5 * Z[0].y; // read before anything is staged over the spectrum
  double suma = 0.0, sumb = 0.0;
#pragma unroll
  for (int k = k0; k < k0 + chunk && k <= K; ++k) {
    cplx z1 = Z[PLAN::pos_out(k)], z2 = Z[PLAN::pos_out(N - k)];
    suma += 0.5 * (z1.x + z2.x);
    sumb += 0.5 * (z1.y + z2.y);
  }
  // exclusive prefix of the per-thread partial sums (FFTPACK's running sum): shuffles inside a wave, the wave
  // totals through LDS
  const int lane = tid & 63, wv = tid >> 6;
  double inca = suma, incb = sumb;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    double va = __shfl_up(inca, off), vb = __shfl_up(incb, off);
    if (lane >= off) {
      inca += va;
      incb += vb;
    }
  }
  if (lane == 63) {
    red[wv] = inca;
    red[NT / 64 + wv] = incb;
  }
  __syncthreads();
  double runa = b1a + (inca - suma), runb = b1b + (incb - sumb);
  for (int w = 0; w < wv; ++w) {
    runa += red[w];
    runb += red[NT / 64 + w];
  }
  // Output staging for one coalesced sweep to global memory (each thread produces runs of consecutive elements):
  //   two buffers : the free buffer, row a at oa[i], row b at ob[i]
  //   one buffer  : in place - the four outputs of index k take the four doubles of Z[k], Z[N-k], which only this
  //                 thread reads: row a: b[2k-1], b[2k] at A[k].x, A[k].y; row b at A[N-k].x, .y; b[0] at A[0]
  double *oa = reinterpret_cast<double *>(out), *ob = oa + N;
  double rsa = 0.0, rsb = 0.0; // row sums (inverse pass of the round-1a structure; kept for ROWSUM)
  if (tid == 0) {
    if (single) Z[0] = {b1a, b1b};
    else {
      oa[0] = b1a;
      ob[0] = b1b;
    }
    rsa += b1a;
    rsb += b1b;
  }
#pragma unroll
  for (int k = k0; k < k0 + chunk && k <= K; ++k) {
    cplx z1 = Z[PLAN::pos_out(k)], z2 = Z[PLAN::pos_out(N - k)];
    double rea = 0.5 * (z1.x + z2.x), ima = 0.5 * (z1.y - z2.y);
    double reb = 0.5 * (z1.y + z2.y), imb = -0.5 * (z1.x - z2.x);
    runa += rea;
    runb += reb;
    if (single) {
      Z[PLAN::pos_out(k)] = {-ima, runa};
      Z[PLAN::pos_out(N - k)] = {-imb, runb};
    } else {
      oa[2 * k - 1] = -ima; // b[2k]
      oa[2 * k] = runa;     // b[2k+1]
      ob[2 * k - 1] = -imb;
      ob[2 * k] = runb;
    }
    rsa += runa - ima;
    if (has_b) rsb += runb - imb;
  }
  // n even: the last even output b[n] = -Im Y_{n/2} has no odd partner (never with one buffer: N is even there)
  if (!(n & 1) && tid == NT - 1) {
    int k = n / 2;
    cplx z1 = Z[PLAN::pos_out(k)], z2 = Z[PLAN::pos_out(N - k)];
    double ima = 0.5 * (z1.y - z2.y), imb = -0.5 * (z1.x - z2.x);
    oa[n - 1] = -ima;
    rsa += -ima;
    ob[n - 1] = -imb;
    if (has_b) rsb += -imb;
  }
  __syncthreads();
  QG_STAMP(0, 3);
  if (single) {
    // elements 2t, 2t+1 as one 16-byte write-through store (qgcm_dev.h; N is even here, and element n = N - 1 of the
    // last pair is the padding column of the row): b[2t] = Z[t].y (Z[0].x for t = 0), b[2t+1] = Z[t+1].x, row b alike
    // from Z[N-t], Z[N-t-1]
    // (the store loops stay loops: as rounds with a predicate they gained nothing here and lost 2 us per step at SOcn 5 km)
#pragma unroll
    for (int t = tid; t < N / 2; t += NT) {
      const cplx za0 = Z[PLAN::pos_out(t)], za1 = Z[PLAN::pos_out(t + 1)];
      const cplx zb0 = Z[PLAN::pos_out((N - t) % N)], zb1 = Z[PLAN::pos_out(N - t - 1)];
      qg_store16_wt(rowa + 2 * t, t == 0 ? za0.x : za0.y, za1.x);
      if (has_b) qg_store16_wt(rowb + 2 * t, t == 0 ? za0.y : zb0.y, zb1.x);
    }
  } else {
    for (int i = tid; i < n; i += NT) {
      rowa[i] = oa[i];
      if (has_b) rowb[i] = ob[i];
    }
  }
  QG_STAMP(0, 4);
  QG_STAMP_DRAIN();
  QG_STAMP(0, 5);
  if (ROWSUM) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      rsa += __shfl_xor(rsa, off);
      rsb += __shfl_xor(rsb, off);
    }
    __syncthreads();
    if (lane == 0) {
      red[wv] = rsa;
      red[NT / 64 + wv] = rsb;
    }
    __syncthreads();
    if (tid == 0) {
      double ta = 0.0, tb = 0.0;
      for (int w = 0; w < NT / 64; ++w) {
        ta += red[w];
        tb += red[NT / 64 + w];
      }
      P.rowsum[(long)m * ny + (ja - 1)] = ta;
      if (has_b) P.rowsum[(long)m * ny + ja] = tb;
    }
  }
}

// ---------------------------------------------------------------------------
// Cyclic ocean: real FFT rows, replacing FFTPACK drfftf / drfftb in hscyoc
// (src/ocisubs.F:566-568, 601-605).  Two rows ride one complex FFT of length
// N = nxto; the spectra are kept in FFTPACK's half-complex order
//   r(1) = X_0, r(2k) = Re X_k, r(2k+1) = Im X_k, r(N) = X_{N/2}
// so the reference's bd2oc ordering (src/q-gcm.F:935-943) applies unchanged.
// INV = false: drfftf (forward, exp(-i..)).  INV = true: drfftb (unnormalised
// inverse) by conjugation, plus the per-row sums needed by xintp and the
// periodic copy wrk(nxpo,j) = wrk(1,j) is left to the consumers (column nxto
// is simply not stored: consumers read column 1).
// grid: (ceil(nrows/2), nlayers); dynamic LDS: 2*N cplx + 2*DST_NT doubles
// ---------------------------------------------------------------------------
#define RFFT_NT 512
// defined in k_cyclic.h (included after this file): part B of the cyclic constraint algebra by one wave, all layer counts
__device__ __forceinline__ void rfft_cyc_constr_partB(const struct QgCycConstrParams *Q, int lane);

template <bool INV, class PLAN = FftPlanNatural, int NT = RFFT_NT>
__global__ __launch_bounds__(NT, (PLAN::three_stage && NT == 384) ? 4 : 1) void k_rfft_cyc(const QgDstParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  if (INV && P.cycq && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup of the inverse launch inside qgcm_hip_steps: c1, c2, c3 and the dpioc step from the
    // zonal-mean column (ksum, ybnd - nothing of wrk), while the other workgroups transform the rows
    if (blockIdx.y == 0 && threadIdx.x < 64) rfft_cyc_constr_partB(P.cycq, threadIdx.x);
    return;
  }
  const int N = PLAN::three_stage ? PLAN::N : P.N, H = N / 2;
  const bool single = PLAN::three_stage || P.single != 0;
  cplx *A = reinterpret_cast<cplx *>(smem_raw);
  cplx *B = single ? A : A + N;
  double *red; // 2 * (NT / 64) doubles behind the buffer(s)
  if constexpr (PLAN::three_stage) red = reinterpret_cast<double *>(A + PLAN::LDS_CPLX);
  else red = reinterpret_cast<double *>((single ? A : B) + N);
  const int tid = threadIdx.x;
  const int ny = P.g.ny, ldw = P.g.ldw;
  const int m = blockIdx.y + P.layer0;
  const int ja = P.g.jr0 + 2 * blockIdx.x;
  const bool has_b = (ja + 1 <= P.g.jr1);
  double *rowa = P.wrk + P.g.wstride * m + (long)(ja - 1) * ldw;
  double *rowb = rowa + ldw;
  typename PLAN::Tw tw3 = PLAN::template prefetch<NT>(P.twid, tid); // table values requested before the rows

  // With a compile-time length (three-stage plans) the row loops are fully unrolled: all loads of a thread are in
  // flight together instead of one memory round trip per iteration of a rolled loop.
  constexpr int NC = PLAN::N;
  if (!INV) {
    if constexpr (PLAN::three_stage) {
      double2 v[(NC + NT - 1) / NT];
#pragma unroll
      for (int it = 0; it < (NC + NT - 1) / NT; ++it) {
        const int j = tid + it * NT, jc = j < NC ? j : 0;
        v[it] = {rowa[jc], has_b ? rowb[jc] : 0.0};
      }
#pragma unroll
      for (int it = 0; it < (NC + NT - 1) / NT; ++it) {
        const int j = tid + it * NT;
        if (j < NC) A[PLAN::pos_in(j)] = {v[it].x, v[it].y};
      }
    } else {
      for (int j = tid; j < N; j += NT) A[PLAN::pos_in(j)] = {rowa[j], has_b ? rowb[j] : 0.0};
    }
  } else {
    // half-complex rows -> conj(Z), Z_k = Xa_k + i Xb_k
    if constexpr (PLAN::three_stage) {
      // unconditional loads at clamped indices, all in flight together; the special coefficients k = 0, N/2 by selects
      constexpr int HC = NC / 2, NIT = (HC + 1 + NT - 1) / NT;
      double2 va[NIT], vb[NIT];
      const double *rb = has_b ? rowb : rowa;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * NT;
        const int kc = k <= HC ? k : 0;
        const int i1 = (kc == 0) ? 0 : (kc == HC ? NC - 1 : 2 * kc - 1), i2 = (kc == 0 || kc == HC) ? i1 : 2 * kc;
        va[it] = {rowa[i1], rowa[i2]};
        vb[it] = {rb[i1], rb[i2]};
      }
      const double bs = has_b ? 1.0 : 0.0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int k = tid + it * NT;
        if (k <= HC) {
          const bool edge = (k == 0 || k == HC);
          const double ar = va[it].x, ai = edge ? 0.0 : va[it].y;
          const double br = bs * vb[it].x, bi = edge ? 0.0 : bs * vb[it].y;
          A[PLAN::pos_in(k)] = {ar - bi, -(ai + br)};
          if (!edge) A[PLAN::pos_in(NC - k)] = {ar + bi, -(br - ai)};
        }
      }
    } else
    for (int k = tid; k <= H; k += NT) {
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
      A[PLAN::pos_in(k)] = {ar - bi, -(ai + br)};
      if (k > 0 && k < H) A[PLAN::pos_in(N - k)] = {ar + bi, -(br - ai)};
    }
  }
  __syncthreads();

  cplx *in = A, *out = B;
  int s = 1, len = N;
  if constexpr (PLAN::three_stage) PLAN::template run<NT>(A, tw3, tid);
  else
  for (int f = 0; f < P.nfac; ++f) {
    const int R = P.fac[f];
    const int mm = len / R;
    const int twstep = N / len;
    if (single) { // N <= DST_SINGLE_MAXN: the MAXIT bounds below cover N / (R * NT)
      switch (R) {
        case 2: dst_stage_ip<2, NT, 5>(A, N, s, mm, P.twid, twstep, tid); break;
        case 3: dst_stage_ip<3, NT, 4>(A, N, s, mm, P.twid, twstep, tid); break;
        case 4: dst_stage_ip<4, NT, 3>(A, N, s, mm, P.twid, twstep, tid); break;
        case 5: dst_stage_ip<5, NT, 2>(A, N, s, mm, P.twid, twstep, tid); break;
        default: dst_stage_ip<8, NT, 2>(A, N, s, mm, P.twid, twstep, tid); break;
      }
    } else {
      switch (R) {
        case 2: dst_stage<2, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 3: dst_stage<3, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 4: dst_stage<4, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 5: dst_stage<5, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        case 8: dst_stage<8, NT>(in, out, N, s, mm, P.twid, twstep, tid); break;
        default: dst_stage_generic<NT>(R, in, out, N, s, mm, P.twid, twstep, tid); break;
      }
    }
    __syncthreads();
    cplx *t = in;
    in = out;
    out = t;
    s *= R;
    len = mm;
  }
  const cplx *Z = in;

  if (!INV) {
    // Xa_k = (Z_k + conj Z_{N-k})/2, Xb_k = (Z_k - conj Z_{N-k})/(2i)
    // half-complex order r(1) = X_0, r(2k) = Re X_k, r(2k+1) = Im X_k, r(N) = X_{N/2}: the 16-byte aligned pair
    // (2t, 2t+1) of the row is (Im X_t, Re X_{t+1}), (X_0, Re X_1) for t = 0 - one write-through store (qgcm_dev.h)
    // (left as a loop over the thread's elements: fully unrolled - rounds with a predicate, as k_dst_box loads its rows -
    //  the stores leave in one burst at the end and SOcn 5 km was 2 us per step SLOWER, profiles/r4_row_loops_ab.log)
#pragma unroll
    for (int t = tid; t < (PLAN::three_stage ? NC / 2 : H); t += NT) {
      const cplx z1 = Z[PLAN::pos_out(t)], z2 = Z[PLAN::pos_out((N - t) % N)];
      const cplx z3 = Z[PLAN::pos_out(t + 1)], z4 = Z[PLAN::pos_out(N - t - 1)];
      const double ar = 0.5 * (z1.x + z2.x), ai = 0.5 * (z1.y - z2.y);
      const double br = 0.5 * (z1.y + z2.y), bi = -0.5 * (z1.x - z2.x);
      const double arn = 0.5 * (z3.x + z4.x), brn = 0.5 * (z3.y + z4.y);
      qg_store16_wt(rowa + 2 * t, t == 0 ? ar : ai, arn);
      if (has_b) qg_store16_wt(rowb + 2 * t, t == 0 ? br : bi, brn);
    }
  } else {
    double rsa = 0.0, rsb = 0.0;
#pragma unroll
    for (int t = tid; t < (PLAN::three_stage ? NC : N) / 2; t += NT) { // 16-byte write-through stores (qgcm_dev.h)
      const cplx z0 = Z[PLAN::pos_out(2 * t)], z1 = Z[PLAN::pos_out(2 * t + 1)];
      qg_store16_wt(rowa + 2 * t, z0.x, z1.x);
      rsa += z0.x;
      rsa += z1.x;
      if (has_b) {
        qg_store16_wt(rowb + 2 * t, -z0.y, -z1.y);
        rsb += -z0.y;
        rsb += -z1.y;
      }
    }
    if (P.rowsum) {
      // fixed order: xor butterfly inside each wave, then the wave totals left to right
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        rsa += __shfl_xor(rsa, off);
        rsb += __shfl_xor(rsb, off);
      }
      __syncthreads(); // every thread is done with the spectrum in A (red may share its tail in single mode)
      if ((tid & 63) == 0) {
        red[tid >> 6] = rsa;
        red[NT / 64 + (tid >> 6)] = rsb;
      }
      __syncthreads();
      if (tid == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < NT / 64; ++w) {
          ta += red[w];
          tb += red[NT / 64 + w];
        }
        P.rowsum[(long)m * ny + (ja - 1)] = ta;
        if (has_b) P.rowsum[(long)m * ny + ja] = tb;
      }
    }
  }
}

#pragma clang fp contract(off)
