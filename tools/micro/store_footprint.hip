// Microbenchmark: the chip-wide rate of pure f32 stores as a function of the FOOTPRINT they cover, plain and non-temporal -- to
// reconcile the 5.2-5.4 TB/s at which any kernel writes the 2.64 GB correlation volume (store_rate.hip) with the 6.0-6.2 TB/s the
// MI355X guide quotes for the same store shape into 75-302 MB tables (which the 256 MB Infinity Cache absorbs).
// Shape: one dword per lane, a wave instruction = 256 contiguous bytes (or, mode 1, two 128-byte segments 8 rows apart like a 32x32
// accumulator register); 2048 workgroups x 256 threads sweep the footprint `passes` times so that every case writes ~16 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int NT>
__global__ __launch_bounds__(256) void wr(float* __restrict__ v, size_t n_dwords, int passes) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (int p = 0; p < passes; ++p)
    for (size_t i = tid; i < n_dwords; i += stride) {
      if (NT) __builtin_nontemporal_store((float)p, v + i);
      else v[i] = (float)p;
    }
}

int main() {
  const double mb[] = {64, 128, 192, 256, 384, 512, 1024, 2048, 2640, 4096};
  float* buf;
  if (hipMalloc(&buf, (size_t)4096 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("footprint_MB  plain_TBps  nt_TBps   (median of 5; ~16 GB written per measurement)\n");
  for (double m : mb) {
    const size_t n = (size_t)(m * (1 << 20) / 4);
    const int passes = std::max(1, (int)(16384.0 / m));
    double res[2];
    for (int nt = 0; nt < 2; ++nt) {
      std::vector<float> t;
      for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        if (nt) wr<1><<<2048, 256>>>(buf, n, passes); else wr<0><<<2048, 256>>>(buf, n, passes);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r) t.push_back(ms);
      }
      std::sort(t.begin(), t.end());
      res[nt] = (double)n * 4 * passes / (t[t.size() / 2] * 1e-3) / 1e12;
    }
    printf("%10.0f  %9.2f  %8.2f\n", m, res[0], res[1]);
    fflush(stdout);
  }
  return 0;
}
