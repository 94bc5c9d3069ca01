// k_fft3.h - row transforms of long rows (nxto = 4608, 4800, ...): complex FFT of length N = R1*R2*R3 in THREE
// in-place stages, each radix (8..20) done in registers by one thread.
//
// The generic Stockham plan of k_dst.h runs a 4608- or 4800-point row pair through five stages of radix <= 8, each
// with two workgroup barriers and a full pass through LDS; at those lengths the row kernels are bound by exactly
// that (NAtl 1 km: 64 us per direction and 600-row slab, 2.2 TB/s).  Here
//   n = n1*R2*R3 + n2*R3 + n3,   k = k1 + R1*k2 + R1*R2*k3
//   stage 1: DFT_R1 over n1 for every (n2, n3), twiddle W_N^(k1*(n2*R3+n3))
//   stage 2: DFT_R2 over n2 for every (k1, n3), twiddle W_N^(R1*k2*n3)
//   stage 3: DFT_R3 over n3 for every (k1, k2)
// every butterfly reads and writes its OWN R slots, so a stage is in place with one barrier after it (3 barriers, 3
// passes through LDS instead of 10 and 5); the output is left digit-reversed (X[k] at pos_out(k)) and the
// post-processing of the row kernels reads it through that map.  Rows of R3 complex numbers are padded by one
// (conflict-free 16-byte accesses in stage 3, where a thread owns a contiguous row).
// Composite radices (9, 10, 12, 15, 16, 18, 20) are Cooley-Tukey compositions of the radix-2/3/4/5/8 butterflies
// with compile-time twiddles.
#pragma once
#include "k_dst.h"

#pragma clang fp contract(fast)

template <int NN>
__device__ __forceinline__ cplx fft3_wc(int j); // exp(-2 pi i j / NN), j < NN, folded at compile time

template <>
__device__ __forceinline__ cplx fft3_wc<9>(int j) {
  constexpr double re[9] = {1.0, 0.766044443118978, 0.17364817766693041, -0.4999999999999998, -0.9396926207859083, -0.9396926207859084, -0.5000000000000004, 0.17364817766692997, 0.7660444431189778};
  constexpr double im[9] = {0.0, -0.6427876096865393, -0.984807753012208, -0.8660254037844387, -0.3420201433256689, 0.34202014332566866, 0.8660254037844384, 0.9848077530122081, 0.6427876096865396};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<10>(int j) {
  constexpr double re[10] = {1.0, 0.8090169943749475, 0.30901699437494745, -0.30901699437494734, -0.8090169943749473, -1.0, -0.8090169943749476, -0.30901699437494756, 0.30901699437494723, 0.8090169943749473};
  constexpr double im[10] = {0.0, -0.5877852522924731, -0.9510565162951535, -0.9510565162951536, -0.5877852522924732, -1.2246467991473532e-16, 0.587785252292473, 0.9510565162951535, 0.9510565162951536, 0.5877852522924734};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<12>(int j) {
  constexpr double re[12] = {1.0, 0.8660254037844387, 0.5000000000000001, 6.123233995736766e-17, -0.4999999999999998, -0.8660254037844387, -1.0, -0.8660254037844388, -0.5000000000000004, -1.8369701987210297e-16, 0.5000000000000001, 0.8660254037844384};
  constexpr double im[12] = {0.0, -0.49999999999999994, -0.8660254037844386, -1.0, -0.8660254037844387, -0.49999999999999994, -1.2246467991473532e-16, 0.4999999999999997, 0.8660254037844384, 1.0, 0.8660254037844386, 0.5000000000000004};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<15>(int j) {
  constexpr double re[15] = {1.0, 0.9135454576426009, 0.6691306063588582, 0.30901699437494745, -0.10452846326765333, -0.4999999999999998, -0.8090169943749473, -0.9781476007338057, -0.9781476007338057, -0.8090169943749476, -0.5000000000000004, -0.10452846326765423, 0.30901699437494723, 0.6691306063588585, 0.913545457642601};
  constexpr double im[15] = {0.0, -0.40673664307580015, -0.7431448254773941, -0.9510565162951535, -0.9945218953682734, -0.8660254037844387, -0.5877852522924732, -0.20791169081775931, 0.20791169081775907, 0.587785252292473, 0.8660254037844384, 0.9945218953682733, 0.9510565162951536, 0.743144825477394, 0.40673664307580015};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<16>(int j) {
  constexpr double re[16] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.38268343236508984, 6.123233995736766e-17, -0.3826834323650897, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112868, -0.7071067811865477, -0.38268343236509034, -1.8369701987210297e-16, 0.38268343236509, 0.7071067811865474, 0.9238795325112865};
  constexpr double im[16] = {0.0, -0.3826834323650898, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650899, -1.2246467991473532e-16, 0.38268343236508967, 0.7071067811865475, 0.9238795325112865, 1.0, 0.9238795325112866, 0.7071067811865477, 0.3826834323650904};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<18>(int j) {
  constexpr double re[18] = {1.0, 0.9396926207859084, 0.766044443118978, 0.5000000000000001, 0.17364817766693041, -0.1736481776669303, -0.4999999999999998, -0.7660444431189779, -0.9396926207859083, -1.0, -0.9396926207859084, -0.7660444431189783, -0.5000000000000004, -0.17364817766693033, 0.17364817766692997, 0.49999999999999933, 0.7660444431189778, 0.9396926207859084};
  constexpr double im[18] = {0.0, -0.3420201433256687, -0.6427876096865393, -0.8660254037844386, -0.984807753012208, -0.984807753012208, -0.8660254037844387, -0.6427876096865395, -0.3420201433256689, -1.2246467991473532e-16, 0.34202014332566866, 0.6427876096865389, 0.8660254037844384, 0.984807753012208, 0.9848077530122081, 0.866025403784439, 0.6427876096865396, 0.3420201433256686};
  return {re[j], im[j]};
}

template <>
__device__ __forceinline__ cplx fft3_wc<20>(int j) {
  constexpr double re[20] = {1.0, 0.9510565162951535, 0.8090169943749475, 0.5877852522924731, 0.30901699437494745, 6.123233995736766e-17, -0.30901699437494734, -0.587785252292473, -0.8090169943749473, -0.9510565162951535, -1.0, -0.9510565162951538, -0.8090169943749476, -0.5877852522924732, -0.30901699437494756, -1.8369701987210297e-16, 0.30901699437494723, 0.5877852522924729, 0.8090169943749473, 0.9510565162951535};
  constexpr double im[20] = {0.0, -0.3090169943749474, -0.5877852522924731, -0.8090169943749475, -0.9510565162951535, -1.0, -0.9510565162951536, -0.8090169943749475, -0.5877852522924732, -0.3090169943749475, -1.2246467991473532e-16, 0.3090169943749469, 0.587785252292473, 0.8090169943749473, 0.9510565162951535, 1.0, 0.9510565162951536, 0.8090169943749476, 0.5877852522924734, 0.3090169943749476};
  return {re[j], im[j]};
}


// in-register DFT of length NN, natural order in and out
template <int NN>
struct Fft3Dft;

template <int R>
struct Fft3Base {
  static __device__ __forceinline__ void run(cplx *a) {
    cplx o[R];
    dst_bfly<R>(a, o);
#pragma unroll
    for (int u = 0; u < R; ++u) a[u] = o[u];
  }
};
template <> struct Fft3Dft<2> : Fft3Base<2> {};
template <> struct Fft3Dft<3> : Fft3Base<3> {};
template <> struct Fft3Dft<4> : Fft3Base<4> {};
template <> struct Fft3Dft<5> : Fft3Base<5> {};
template <> struct Fft3Dft<8> : Fft3Base<8> {};

// Cooley-Tukey composition NN = A*B:  n = B*na + nb,  k = ka + A*kb
template <int A, int B>
struct Fft3CT {
  static __device__ __forceinline__ void run(cplx *a) {
    constexpr int NN = A * B;
    cplx t[NN];
#pragma unroll
    for (int nb = 0; nb < B; ++nb) {
      cplx x[A];
#pragma unroll
      for (int na = 0; na < A; ++na) x[na] = a[B * na + nb];
      Fft3Dft<A>::run(x);
#pragma unroll
      for (int ka = 0; ka < A; ++ka) t[B * ka + nb] = (ka * nb == 0) ? x[ka] : cmul(x[ka], fft3_wc<NN>((ka * nb) % NN));
    }
#pragma unroll
    for (int ka = 0; ka < A; ++ka) {
      cplx x[B];
#pragma unroll
      for (int nb = 0; nb < B; ++nb) x[nb] = t[B * ka + nb];
      Fft3Dft<B>::run(x);
#pragma unroll
      for (int kb = 0; kb < B; ++kb) a[ka + A * kb] = x[kb];
    }
  }
};
template <> struct Fft3Dft<9> : Fft3CT<3, 3> {};
template <> struct Fft3Dft<10> : Fft3CT<2, 5> {};
template <> struct Fft3Dft<12> : Fft3CT<3, 4> {};
template <> struct Fft3Dft<15> : Fft3CT<3, 5> {};
template <> struct Fft3Dft<16> : Fft3CT<4, 4> {};
template <> struct Fft3Dft<18> : Fft3CT<2, 9> {};
template <> struct Fft3Dft<20> : Fft3CT<4, 5> {};

// powers tw[k] = w^k, k = 1..R-1, by a product tree of depth log2(R) (rounding grows with the depth, not with k)
template <int R>
__device__ __forceinline__ void fft3_powers(cplx w, cplx *tw) {
  tw[1 % R] = w;
#pragma unroll
  for (int k = 2; k < R; ++k) tw[k] = cmul(tw[k / 2], tw[k - k / 2]);
}

template <int R1, int R2, int R3>
struct Fft3Plan {
  static constexpr bool three_stage = true;
  // rows of R3 are padded by one (stage 3: a thread owns a contiguous row), blocks of R2 rows by one more (the
  // post-processing reads X[k], k consecutive = consecutive k1 = consecutive blocks: without it every lane of a wave
  // hit the same LDS banks)
  static constexpr int N = R1 * R2 * R3, PITCH = R3 + 1, BLOCK = R2 * PITCH + 1, LDS_CPLX = R1 * BLOCK, R1R2 = R1 * R2;
  static __device__ __forceinline__ int pos_in(int n) { return (n / (R2 * R3)) * BLOCK + (n % (R2 * R3)) + (n % (R2 * R3)) / R3; }
  static __device__ __forceinline__ int pos_out(int k) { return (k % R1) * BLOCK + ((k / R1) % R2) * PITCH + k / (R1 * R2); }
  // The two twiddle bases of a thread (W_N^r for its stage-1 butterfly, W_N^(R1*n3) for its stage-2 butterfly) are
  // requested by the kernel BEFORE it loads the rows: a table load issued inside a stage waits 2-3 us behind the bulk
  // traffic of the other workgroups - per stage - which is what bounded the five-stage Stockham plan at these lengths.
  // Work split: NT = 256 threads = one wave per SIMD.  A stage has N/R butterflies (256..320): butterfly `tid` for
  // every thread, and the few beyond NT go to the first lanes of wave 1 in stage 1, wave 2 in stage 2, wave 3 in
  // stage 3 - so no SIMD carries a second wave's worth of work in every stage (a 320-thread workgroup puts two of its
  // five waves on one SIMD: measured VALU-bound on exactly that SIMD).
  // (a stage with more than NT + 64 butterflies - the cheap radix of an uneven plan, e.g. 12 of 12 * 20 * 20 - gives
  //  the second round to the first lanes of the workgroup: item NT + tid)
  template <int NT>
  static __device__ __forceinline__ int extra_item(int tid, int wave, int items) {
    if (items > NT + 64) return NT + tid < items ? NT + tid : -1;
    const int e = NT + tid - 64 * wave;
    return (tid >= 64 * wave && e < items) ? e : -1;
  }
  struct Tw {
    double2 w1, w2, w1x, w2x;
  };
  template <int NT>
  static __device__ __forceinline__ Tw prefetch(const double2 *__restrict__ twid, int tid) {
    static_assert(NT % 64 == 0 && NT >= 256, "whole waves; the extra butterflies go to waves 1 .. 3");
    static_assert(R2 * R3 <= 2 * NT && R1 * R3 <= 2 * NT && R1 * R2 <= 2 * NT, "at most one extra butterfly per lane");
    Tw t;
    const int e1 = extra_item<NT>(tid, 1, R2 * R3), e2 = extra_item<NT>(tid, 2, R1 * R3);
    t.w1 = twid[tid < R2 * R3 ? tid : 0];
    t.w2 = twid[tid < R1 * R3 ? R1 * (tid % R3) : 0];
    t.w1x = twid[e1 >= 0 ? e1 : 0];
    t.w2x = twid[e2 >= 0 ? R1 * (e2 % R3) : 0];
    return t;
  }
  static __device__ __forceinline__ void stage1(cplx *A, int r, double2 w) {
    cplx x[R1], tw[R1];
    const int p0 = r + r / R3; // pos_in(n1*R2*R3 + r) = n1*BLOCK + p0
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) x[n1] = A[n1 * BLOCK + p0];
    Fft3Dft<R1>::run(x);
    fft3_powers<R1>(cplx{w.x, w.y}, tw);
    A[p0] = x[0];
#pragma unroll
    for (int k1 = 1; k1 < R1; ++k1) A[k1 * BLOCK + p0] = cmul(x[k1], tw[k1]);
  }
  static __device__ __forceinline__ void stage2(cplx *A, int id, double2 w) {
    const int k1 = id / R3, n3 = id - k1 * R3;
    cplx x[R2], tw[R2];
    cplx *base = A + k1 * BLOCK + n3;
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) x[n2] = base[n2 * PITCH];
    Fft3Dft<R2>::run(x);
    fft3_powers<R2>(cplx{w.x, w.y}, tw);
    base[0] = x[0];
#pragma unroll
    for (int k2 = 1; k2 < R2; ++k2) base[k2 * PITCH] = cmul(x[k2], tw[k2]);
  }
  static __device__ __forceinline__ void stage3(cplx *A, int id) {
    cplx x[R3];
    cplx *base = A + (id / R2) * BLOCK + (id % R2) * PITCH;
#pragma unroll
    for (int n3 = 0; n3 < R3; ++n3) x[n3] = base[n3];
    Fft3Dft<R3>::run(x);
#pragma unroll
    for (int k3 = 0; k3 < R3; ++k3) base[k3] = x[k3];
  }
  // forward complex FFT (exp(-i..)) of the N values at pos_in(0..N-1), result at pos_out(0..N-1); ends with a barrier
  template <int NT>
  static __device__ __forceinline__ void run(cplx *A, const Tw &T, int tid) {
    {
      const int e = extra_item<NT>(tid, 1, R2 * R3);
      if (tid < R2 * R3) stage1(A, tid, T.w1);
      if (e >= 0) stage1(A, e, T.w1x);
    }
    __syncthreads();
    {
      const int e = extra_item<NT>(tid, 2, R1 * R3);
      if (tid < R1 * R3) stage2(A, tid, T.w2);
      if (e >= 0) stage2(A, e, T.w2x);
    }
    __syncthreads();
    {
      const int e = extra_item<NT>(tid, 3, R1 * R2);
      if (tid < R1 * R2) stage3(A, tid);
      if (e >= 0) stage3(A, e);
    }
    __syncthreads();
  }
};

#pragma clang fp contract(off)
