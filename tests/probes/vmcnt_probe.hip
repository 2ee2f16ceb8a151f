// Probe: by how much does one global_load_lds_dwordx4 (LDS-DMA) increment the wave's VM_CNT?
// Reads HW_REG_IB_STS right after issuing n DMA instructions (vm_cnt = bits 3:0 | bits 23:22 << 4).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const char* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x;
  unsigned r[6];
  asm volatile("s_waitcnt vmcnt(0)");
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[0]));
  __builtin_amdgcn_global_load_lds(src + lane * 16, (__attribute__((address_space(3))) void*)(lds), 16, 0, 0);
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[1]));
  __builtin_amdgcn_global_load_lds(src + 1024 + lane * 16, (__attribute__((address_space(3))) void*)(lds + 1024), 16, 0, 0);
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[2]));
  __builtin_amdgcn_global_load_lds(src + 2048 + lane * 16, (__attribute__((address_space(3))) void*)(lds + 2048), 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src + 3072 + lane * 16, (__attribute__((address_space(3))) void*)(lds + 3072), 16, 0, 0);
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[3]));
  asm volatile("s_waitcnt vmcnt(0)");
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[4]));
  // a plain 16-byte global load for comparison
  float4 v = *(const float4*)(src + 8192 + lane * 16);
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(r[5]));
  if (lane == 0) for (int i = 0; i < 6; ++i) out[i] = r[i];
  if (v.x == 123.f) out[7] = 1;
}
int main() {
  char* src; unsigned* out; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&out, 64);
  probe<<<1, 64>>>(src, out);
  unsigned h[8]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  const char* what[6] = {"idle", "after 1 DMA", "after 2 DMA", "after 4 DMA", "after wait(0)", "after 1 plain load"};
  for (int i = 0; i < 6; ++i) printf("%-20s IB_STS=0x%08x vm_cnt=%u lgkm_cnt=%u\n", what[i], h[i], (h[i] & 0xf) | (((h[i] >> 22) & 3) << 4), (h[i] >> 8) & 0xf);
  return 0;
}
