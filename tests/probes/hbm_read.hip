// Probe: what a pure streaming read sustains on this MI355X (the ceiling the weight-gradient kernel's 5.3-5.7 TB/s is to
// be judged against): 2.4 GB read once per launch, 16-byte loads, 8 in flight per lane, by 256..2048 workgroups; and
// the same with a 5 % write stream beside it (the kernel's partial sums).
// build: hipcc --offload-arch=gfx950 -O3 tests/probes/hbm_read.hip -o hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WRITE>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ src, int64_t n16, f32x4* __restrict__ dst, float* sink) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
    if (WRITE && ((i / stride) % 160) == 0) __builtin_nontemporal_store(acc, dst + i);   // ~5 % of the bytes read
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) sink[0] = acc[0];
}

int main() {
  const int64_t bytes = (int64_t)2400 << 20, n16 = bytes / 16;
  f32x4 *src, *dst; float* sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 4);
  hipMemset(src, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int wr = 0; wr < 2; ++wr)
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (wr) k<1><<<blocks, 512>>>(src, n16, dst, sink); else k<0><<<blocks, 512>>>(src, n16, dst, sink);
        hipEventRecord(b); hipEventSynchronize(b);
      }
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%s  %4d workgroups x 512: %.3f ms  %.2f TB/s read\n", wr ? "read + 5%% write" : "read only     ", blocks, ms, bytes / ms / 1e9);
    }
  return 0;
}
