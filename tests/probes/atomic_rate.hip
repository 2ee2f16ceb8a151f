// Probe: throughput of device-scope fp32 atomic adds on MI355X as a function of the size of the region they fall into
// (does L2 / Infinity-Cache residency of the target matter?), of pairing (two adjacent floats per cell, as the hash
// table's two features) and of the alternative "read-modify-write without atomics" cost for reference.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tests/probes/atomic_rate.hip -o /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// every thread issues `per_thread` cell updates; a cell = 2 adjacent floats; cells drawn uniformly from [0, n_cells)
template <int MODE>   // 0: two fp32 atomics per cell, 1: one fp32 atomic per cell, 2: plain 8-byte load (gather cost)
__global__ void k(float* tab, uint32_t n_cells, int per_thread, float* sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  for (int i = 0; i < per_thread; ++i) {
    const uint32_t c = mix(gid * 977u + i * 0x9e3779b9u) % n_cells;
    if (MODE == 0) {
      __hip_atomic_fetch_add(tab + 2 * (size_t)c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(tab + 2 * (size_t)c + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 1) {
      __hip_atomic_fetch_add(tab + 2 * (size_t)c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 3) {        // lane pairs share a cell: even lane feature 0, odd lane feature 1 (one instruction, 32 cells)
      const uint32_t c2 = mix((gid >> 1) * 977u + i * 0x9e3779b9u) % n_cells;
      __hip_atomic_fetch_add(tab + 2 * (size_t)c2 + (gid & 1), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 4) {        // one 64-bit integer add per cell (two packed fixed-point features)
      __hip_atomic_fetch_add((unsigned long long*)tab + c, 0x0000000100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 5) {        // one 32-bit integer add per cell
      __hip_atomic_fetch_add((unsigned int*)tab + 2 * (size_t)c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 6) {        // 16 lanes share a 64-byte line (8 cells x 2 features), lines random
      const uint32_t l = mix((gid >> 4) * 977u + i * 0x9e3779b9u) % (n_cells / 8);
      __hip_atomic_fetch_add(tab + 16 * (size_t)l + (gid & 15), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const float2 v = *(const float2*)(tab + 2 * (size_t)c);
      acc += v.x + v.y;
    }
  }
  if (MODE == 2 && acc == 12345.678f) sink[0] = acc;
}

template <int MODE> float run(float* tab, uint32_t n_cells, int blocks, int per_thread, float* sink) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<blocks, 256>>>(tab, n_cells, per_thread, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 3; ++r) k<MODE><<<blocks, 256>>>(tab, n_cells, per_thread, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 3;
}

int main() {
  const size_t max_bytes = (size_t)1 << 30;
  float* tab; hipMalloc(&tab, max_bytes); hipMemset(tab, 0, max_bytes);
  float* sink; hipMalloc(&sink, 4);
  const int blocks = 4096, per_thread = 32;            // 33.5 M cell updates per launch
  const double cells = (double)blocks * 256 * per_thread;
  printf("region      | 2 atomics/cell: ms  Gcell/s | 1 atomic/cell: ms  Gcell/s | 8-byte gather: ms  Gcell/s\n");
  for (size_t bytes : {(size_t)1 << 16, (size_t)1 << 20, (size_t)1 << 22, (size_t)1 << 24, (size_t)56 << 20, (size_t)1 << 28, (size_t)1 << 30}) {
    const uint32_t n_cells = (uint32_t)(bytes / 8);
    const float t0 = run<0>(tab, n_cells, blocks, per_thread, sink);
    const float t1 = run<1>(tab, n_cells, blocks, per_thread, sink);
    const float t2 = run<2>(tab, n_cells, blocks, per_thread, sink);
    printf("%8.2f MB | %8.3f %8.2f | %8.3f %8.2f | %8.3f %8.2f\n", bytes / 1048576.0, t0, cells / t0 / 1e6, t1, cells / t1 / 1e6,
           t2, cells / t2 / 1e6);
    const float t3 = run<3>(tab, n_cells, blocks, per_thread, sink), t4 = run<4>(tab, n_cells, blocks, per_thread, sink);
    const float t5 = run<5>(tab, n_cells, blocks, per_thread, sink), t6 = run<6>(tab, n_cells, blocks, per_thread, sink);
    printf("            lane-atomics/s (G): paired lanes %.2f | u64 add %.2f | u32 add %.2f | 16 lanes per 64-B line %.2f\n",
           cells / t3 / 1e6, cells / t4 / 1e6, cells / t5 / 1e6, cells / t6 / 1e6);
  }
  return 0;
}
