// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (which lane receives which LDS element).
// Every lane l passes the address of 16-bit elements [4l, 4l+4); the kernel prints what came back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * l));
  for (int e = 0; e < 4; ++e) out[4 * l + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 512);
  probe<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  // hypothesis: out[l][e] = 64*(l>>4) + 16*e + (l&15)   (column l&15 of the group's 4x16 block)
  int ok = 1;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) ok &= (h[4*l+e] == 64*(l>>4) + 16*e + (l&15));
  printf("hypothesis %s\n", ok ? "HOLDS" : "FAILS");
  return 0;
}
