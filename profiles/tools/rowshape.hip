// What does the SHAPE of the long-row kernels cost, transform aside?  864 workgroups of 256 threads; each loads one
// contiguous 73.7 KB chunk (two 4608-point rows: 18 x 16 B per thread, all in flight), waits `delay` (the stages),
// meets at a barrier, stores 73.7 KB.  LDS allocation sets the workgroups per CU (80 KB: two, as the shipped kernels;
// 40 KB: four; 1 KB: as many as the registers allow).  In place or into a second buffer; warm (same 64 MB every launch)
// or cold (six buffers in turn: 382 MB, beyond the Infinity Cache, as inside a step whose other kernels touch ~0.5 GB).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rowshape profiles/tools/rowshape.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double qg_v2d __attribute__((ext_vector_type(2)));
template <bool WT>
__global__ __launch_bounds__(256) void k_shape(const double2 *__restrict__ src, double2 *__restrict__ dst, int delay) {
  extern __shared__ char lds[];
  const long base = (long)blockIdx.x * (18 * 256);
  const int tid = threadIdx.x;
  double2 v[18];
#pragma unroll
  for (int it = 0; it < 18; ++it) v[it] = src[base + tid + it * 256];
  if (delay) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay) __builtin_amdgcn_s_sleep(2);
  }
  if (tid == 0) lds[0] = 1;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 18; ++it) {
    double2 *p = dst + base + tid + it * 256;
    if (WT) {
      const qg_v2d w = {v[it].x + 1.0, v[it].y};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
    } else {
      *p = double2{v[it].x + 1.0, v[it].y};
    }
  }
}
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nwg = 864;
  const size_t bytes = (size_t)nwg * 18 * 256 * 16;
  std::vector<double2 *> bufs(7);
  for (auto &b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
  for (size_t lds : {(size_t)80 * 1024, (size_t)40 * 1024, (size_t)1024}) {
    hipFuncSetAttribute((const void *)k_shape<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void *)k_shape<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int delay : {0, 300, 600, 900})
      for (int cold = 0; cold < 2; ++cold)
        for (int inplace = 0; inplace < 2; ++inplace)
          for (int wt = 0; wt < 2; ++wt) {
            auto launch = [&](int r) {
              double2 *s = bufs[cold ? r % 6 : 0], *d = inplace ? s : bufs[6];
              if (wt) hipLaunchKernelGGL(k_shape<true>, dim3(nwg), dim3(256), lds, 0, s, d, delay);
              else hipLaunchKernelGGL(k_shape<false>, dim3(nwg), dim3(256), lds, 0, s, d, delay);
            };
            for (int r = 0; r < 6; ++r) launch(r);
            hipEventRecord(e0);
            const int reps = 24;
            for (int r = 0; r < reps; ++r) launch(r);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("lds %3zu KB delay %4.1f us %s %s %s : %7.2f us per launch (%.0f GB/s)\n", lds / 1024, delay / 100.0,
                   cold ? "cold" : "warm", inplace ? "in-place " : "two-buffer", wt ? "sc1  " : "plain", 1e3 * ms / reps,
                   2.0 * bytes * reps / (ms * 1e-3) / 1e9);
          }
  }
  return 0;
}
