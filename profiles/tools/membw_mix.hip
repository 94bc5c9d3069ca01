// What can the memory system do for k_tend-shaped traffic? NR read streams + NW write streams of `field` bytes each,
// W bytes per lane (8 or 16), working set NR+NW fields (5 km: 7.5 MB fields: MALL-resident; SOcn: 21.3 MB: HBM)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T, int NR, int NW>
__global__ __launch_bounds__(256) void k_mix(const T *__restrict__ src, T *__restrict__ dst, long nper) {
  // each thread: one element index i of every field
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nper; i += stride) {
    T v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = src[r * nper + i];
    T acc = v[0];
#pragma unroll
    for (int r = 1; r < NR; ++r) { acc.x += v[r].x; }
#pragma unroll
    for (int w = 0; w < NW; ++w) { T o = acc; o.x += w; dst[w * nper + i] = o; }
    if (NW == 0 && acc.x == 1.2345e300) dst[0] = acc;
  }
}
struct d1 { double x; };
int main() {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (double fieldMB : {7.5, 21.3, 185.0}) {
    const size_t fbytes = (size_t)(fieldMB * 1e6) / 4096 * 4096;
    char *a, *b;
    hipMalloc(&a, fbytes * 15); hipMalloc(&b, fbytes * 6);
    hipMemset(a, 0, fbytes * 15); hipMemset(b, 0, fbytes * 6);
    auto run = [&](const char *name, auto launch, double bytes) {
      for (int w = 0; w < 3; ++w) launch();
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("field %6.1f MB %-34s %8.1f GB/s  %8.2f us\n", fieldMB, name, bytes * reps / (ms * 1e-3) / 1e9, 1e3 * ms / reps);
    };
    for (int g : {2048, 8192}) {
      char nm[96];
#define RUN(T, NR, NW, label)                                                                                             \
  snprintf(nm, 96, label " grid %d", g);                                                                                  \
  run(nm, [&] { hipLaunchKernelGGL((k_mix<T, NR, NW>), dim3(g), dim3(256), 0, 0, (const T *)a, (T *)b, (long)(fbytes / sizeof(T))); }, \
      (double)fbytes * (NR + NW));
      RUN(double2, 15, 6, "R15 W6 16B");
      RUN(d1, 15, 6, "R15 W6  8B");
      RUN(double2, 15, 0, "R15 W0 16B");
      RUN(d1, 15, 0, "R15 W0  8B");
      RUN(double2, 1, 6, "R1  W6 16B");
      RUN(d1, 1, 6, "R1  W6  8B");
      RUN(double2, 3, 3, "R3  W3 16B");
      RUN(d1, 3, 3, "R3  W3  8B");
      RUN(double2, 6, 6, "R6  W6 16B");
      RUN(double2, 5, 3, "R5  W3 16B");
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
