// Kernel chain A->B->A..: each kernel reads NR fields the previous one wrote and writes NW fields (R3 W3 of 7.5 MB =
// the solver kernels at 5 km).  What does the store flavour / memory type do to the time per kernel INCLUDING the
// boundary (end-of-kernel L2 write-back of dirty lines)?
#include <hip/hip_runtime.h>
#include <cstdio>
enum { PLAIN = 0, SC1 = 1, NT = 2, ASM16 = 3 };
template <int MODE>
__device__ __forceinline__ void st(double2 *p, double2 v) {
  if (MODE == PLAIN) *p = v;
  else if (MODE == NT) { typedef double v2d __attribute__((ext_vector_type(2))); v2d t = {v.x, v.y}; __builtin_nontemporal_store(t, (v2d *)p); }
  else if (MODE == SC1) {
    __hip_atomic_store(&p->x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&p->y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d t = {v.x, v.y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
  }
}
template <int MODE>
__device__ __forceinline__ void st(double *p, double v) {
  if (MODE == SC1 || MODE == ASM16) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
__device__ __forceinline__ double getx(double v) { return v; }
__device__ __forceinline__ double getx(double2 v) { return v.x; }
__device__ __forceinline__ void addx(double &v, double a) { v += a; }
__device__ __forceinline__ void addx(double2 &v, double a) { v.x += a; }
// LATE = 1: all loads first, a spin of `work` dependent FMAs, then all stores (the structure of the solver kernels)
template <typename T, int NR, int NW, int MODE>
__global__ __launch_bounds__(256) void k_mix(const T *__restrict__ src, T *__restrict__ dst, long nper, int work) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nper; i += stride) {
    T v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = src[r * nper + i];
    T acc = v[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) addx(acc, getx(v[r]));
    double x = getx(acc);
    for (int w = 0; w < work; ++w) x = __builtin_fma(x, 1.0000001, 1e-9);
    addx(acc, x);
#pragma unroll
    for (int w = 0; w < NW; ++w) { T o = acc; addx(o, (double)w); st<MODE>(&dst[w * nper + i], o); }
  }
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t fbytes = (size_t)(7.5e6) / 4096 * 4096;
  for (int unc = 0; unc < 2; ++unc) {
    char *a, *b;
    if (unc) { hipExtMallocWithFlags((void **)&a, fbytes * 3, hipDeviceMallocUncached); hipExtMallocWithFlags((void **)&b, fbytes * 3, hipDeviceMallocUncached); }
    else { hipMalloc(&a, fbytes * 3); hipMalloc(&b, fbytes * 3); }
    hipMemset(a, 0, fbytes * 3); hipMemset(b, 0, fbytes * 3);
    auto run = [&](const char *name, auto launch) {
      for (int w = 0; w < 4; ++w) launch(w & 1);
      hipEventRecord(e0);
      const int reps = 40;
      for (int r = 0; r < reps; ++r) launch(r & 1);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s %-44s %8.2f us per kernel (R3 W3 x 7.5 MB; streaming bound ~7.2)\n", unc ? "UNCACHED" : "cached  ", name, 1e3 * ms / reps);
    };
#define RUN(T, MODE, G, WORK, label)                                                                                      \
  run(label, [&](int odd) {                                                                                               \
    hipLaunchKernelGGL((k_mix<T, 3, 3, MODE>), dim3(G), dim3(256), 0, 0, (const T *)(odd ? b : a), (T *)(odd ? a : b), (long)(fbytes / sizeof(T)), WORK); });
    RUN(double2, PLAIN, 2048, 0, "16B plain grid 2048");
    RUN(double2, NT, 2048, 0, "16B nt grid 2048");
    RUN(double2, ASM16, 2048, 0, "16B asm sc1 grid 2048");
    RUN(double2, SC1, 2048, 0, "2x8B atomic sc1 grid 2048");
    RUN(double, PLAIN, 2048, 0, "8B plain grid 2048");
    RUN(double, SC1, 2048, 0, "8B sc1 grid 2048");
    RUN(double, NT, 2048, 0, "8B nt grid 2048");
    // one generation: every thread exactly one element (like the solver kernels): 7.5e6/16 = 468750 threads = 1832 blocks
    RUN(double2, PLAIN, 1832, 0, "16B plain one-generation");
    RUN(double2, ASM16, 1832, 0, "16B asm sc1 one-generation");
    RUN(double2, PLAIN, 1832, 2000, "16B plain one-gen + 2000 fma");
    RUN(double2, ASM16, 1832, 2000, "16B asm sc1 one-gen + 2000 fma");
    RUN(double, PLAIN, 3663, 2000, "8B plain one-gen + 2000 fma");
    RUN(double, SC1, 3663, 2000, "8B sc1 one-gen + 2000 fma");
    hipFree(a); hipFree(b);
  }
  return 0;
}
