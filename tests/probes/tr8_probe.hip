// Probe for the next round's fp8 saved tensors: semantics of ds_read_b64_tr_b8 on gfx950 (which lane receives which
// LDS byte) and the k-order of v_mfma_f32_32x32x16_fp8_fp8's 8-byte operands.
// Every lane l passes the address of bytes [8l, 8l+8); the kernel prints what came back.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(unsigned char* out, float* mm) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (unsigned char)(i & 255);
  __syncthreads();
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + 8 * l));
  for (int e = 0; e < 8; ++e) out[8 * l + e] = (unsigned char)((e < 4 ? v[0] >> (8 * e) : v[1] >> (8 * (e - 4))) & 255);
  // MFMA k-order: A row i = lane & 31 holds 1.0 (fp8 e4m3 0x38) at byte e of lane-half g only for one (g, e) at a
  // time; B = column j all ones in every k -> C[i][j] = number of matching k: run 16 times, record which k each
  // (g, e) stands for by making B one-hot in k instead.
  for (int kk = 0; kk < 16; ++kk) {
    // B one-hot: lane (j = l & 31, g = l >> 5) byte e is 1.0 iff 8 * g + e == kk  (hypothesis: k = 8 g + e)
    long a = 0, b = 0;
    const int g = l >> 5;
    for (int e = 0; e < 8; ++e) {
      a |= (long)0x38 << (8 * e);                       // A = all ones
      if (8 * g + e == kk) b |= (long)0x38 << (8 * e);
    }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
    if (l == 0) mm[kk] = c[0];                          // expect 1.0 for every kk if k = 8 g + e on both sides
  }
}
int main() {
  unsigned char* d; float* m; hipMalloc(&d, 512); hipMalloc(&m, 64);
  probe<<<1, 64>>>(d, m);
  unsigned char h[512]; float hm[16];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 64, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 8; ++e) printf(" %3d", h[8 * l + e]);
    printf("\n");
  }
  printf("mfma one-hot k sums:"); for (int k = 0; k < 16; ++k) printf(" %.0f", hm[k]); printf("\n");
  return 0;
}
