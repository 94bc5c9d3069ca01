// The 5 km solver kernels reduced to their memory SHAPE (no arithmetic): what does the workgroup structure alone cost?
//   rows    : k_dst64's shape - one WAVE per row pair (2 x 960 doubles = 15.4 KB: 15 x 16 B per lane, all in flight),
//             wait `delay`, store 15.4 KB in place; 1443 waves in 361 workgroups of 256 threads.
//   columns : k_thomas's shape - 512-thread workgroups of 8 wavenumbers x 64 chunks, 16 rows per thread: 8-byte loads at
//             a stride of one row (64-byte segments), wait, two barriers, 8-byte stores; 360 workgroups (120 blocks x 3).
// Fields of 959 x 960 x 3 doubles (22 MB), Infinity-Cache resident as in the step.  delay in 10 ns ticks.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/shape5km profiles/tools/shape5km.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double qg_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wait_ticks(int delay) {
  if (delay) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(1);
  }
}
__global__ __launch_bounds__(256) void k_rows(double2 *w, int npairs_total, int delay) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wv;
  if (pair >= npairs_total) return;
  double2 *p = w + (long)pair * 960; // 2 rows x 960 doubles = 960 double2
  double2 v[15];
#pragma unroll
  for (int i = 0; i < 15; ++i) v[i] = p[lane + 64 * i];
  wait_ticks(delay);
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    const qg_v2d o = {v[i].x + 1.0, v[i].y};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + lane + 64 * i), "v"(o) : "memory");
  }
}
template <bool SHIPPED> // SHIPPED: the block pairing per XCD and the 16-byte write-through pair stores of k_thomas.h
__global__ __launch_bounds__(512) void k_cols(double *w, int nrows, int ldw, long mstride, int delay) {
  __shared__ double s[64][9];
  const int kk = threadIdx.x & 7, c = threadIdx.x >> 3;
  int bx = blockIdx.x;
  if (SHIPPED) {
    const int nfull = (int)gridDim.x / 16 * 16;
    if (bx < nfull) bx = (bx & ~15) + 2 * (bx & 7) + ((bx >> 3) & 1);
  }
  double *base = w + mstride * blockIdx.y + bx * 8 + kk;
  double v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int r = c * 16 + t;
    v[t] = r < nrows ? base[(long)r * ldw] : 0.0;
  }
  wait_ticks(delay / 2);
  s[c][kk] = v[0];
  __syncthreads();
  const double q = s[63 - c][kk];
  wait_ticks(delay / 2);
  __syncthreads();
  if (SHIPPED) {
    const bool odd = (kk & 1) != 0;
    double *cb = odd ? base - 1 : base;
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
      const double m0 = v[t] + q, m1 = v[t + 1] + q;
      const double give = odd ? m0 : m1;
      int lo = __double2loint(give), hi = __double2hiint(give);
      lo = __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true);
      hi = __builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true);
      const double got = __hiloint2double(hi, lo);
      const int r = c * 16 + t + (odd ? 1 : 0);
      const qg_v2d o = {odd ? got : m0, odd ? m1 : got};
      if (r < nrows) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(cb + (long)r * ldw), "v"(o) : "memory");
    }
  } else {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = c * 16 + t;
      if (r < nrows) base[(long)r * ldw] = v[t] + q;
    }
  }
}
// variant: 16 wavenumbers (a whole 128-byte line per row) per 512-thread workgroup, two wavenumbers = 16 bytes per lane
__global__ __launch_bounds__(512) void k_cols16(double *w, int nrows, int ldw, long mstride, int delay) {
  __shared__ double s[64][9];
  const int kk = threadIdx.x & 7, c = threadIdx.x >> 3;
  double *base = w + mstride * blockIdx.y + blockIdx.x * 16 + 2 * kk;
  double2 v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int r = c * 16 + t;
    v[t] = r < nrows ? *reinterpret_cast<const double2 *>(base + (long)r * ldw) : double2{0.0, 0.0};
  }
  wait_ticks(delay / 2);
  s[c][kk] = v[0].x;
  __syncthreads();
  const double q = s[63 - c][kk];
  wait_ticks(delay / 2);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int r = c * 16 + t;
    const qg_v2d o = {v[t].x + q, v[t].y + q};
    if (r < nrows) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(base + (long)r * ldw), "v"(o) : "memory");
  }
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nrows = 959, ldw = 960, nl = 3;
  const long mstride = (long)nrows * ldw;
  double *w; hipMalloc(&w, sizeof(double) * mstride * nl); hipMemset(w, 0, sizeof(double) * mstride * nl);
  auto timeit = [&](const char *name, auto launch) {
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e0);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-48s %7.2f us per launch (%.0f GB/s)\n", name, 1e3 * ms / reps, 2.0 * 8 * mstride * nl * reps / (ms * 1e-3) / 1e9);
  };
  const int npairs = (nrows * nl + 1) / 2;
  for (int delay : {0, 100, 200, 300, 400}) {
    char nm[80];
    snprintf(nm, 80, "rows    (k_dst64 shape)  wait %.1f us", delay / 100.0);
    timeit(nm, [&] { hipLaunchKernelGGL(k_rows, dim3((npairs + 3) / 4), dim3(256), 0, 0, (double2 *)w, npairs, delay); });
  }
  for (int delay : {0, 100, 200, 300, 400}) {
    char nm[80];
    snprintf(nm, 80, "columns (k_thomas shape, naive) wait %.1f us", delay / 100.0);
    timeit(nm, [&] { hipLaunchKernelGGL(k_cols<false>, dim3(ldw / 8, nl), dim3(512), 0, 0, w, nrows, ldw, mstride, delay); });
    snprintf(nm, 80, "columns (k_thomas shape, shipped) wait %.1f us", delay / 100.0);
    timeit(nm, [&] { hipLaunchKernelGGL(k_cols<true>, dim3(ldw / 8, nl), dim3(512), 0, 0, w, nrows, ldw, mstride, delay); });
    snprintf(nm, 80, "columns (16 wavenumbers, 16-B lanes) wait %.1f us", delay / 100.0);
    timeit(nm, [&] { hipLaunchKernelGGL(k_cols16, dim3(ldw / 16, nl), dim3(512), 0, 0, w, nrows, ldw, mstride, delay); });
  }
  return 0;
}
