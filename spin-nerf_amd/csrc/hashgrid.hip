// Hash-grid radiance network for gfx950: multiresolution hash encoding + spherical harmonics + the two small
// bias-free MLPs, forward and backward.
//
// Replaces NeRF_TCNN.forward and its autograd (DS_NeRF/run_nerf_helpers_tcnn.py:13-113, BASELINE config 5), whose
// arithmetic the reference delegates to tiny-cuda-nn (not in the reference tree: parity unpinned — the definition
// followed is restated with its sources in oracle/hashgrid_oracle.py).
//
//   forward   one wave per 32-sample tile.  Lane (sample s, half g) gathers the 8 table entries of each of ITS 8
//             levels (half g owns levels 4g..4g+3 and 8+4g..8+4g+3 — exactly the k-slots the bf16 MFMA B operand asks of
//             that lane half, so the encoding is born in operand layout), then both MLPs run as chained
//             v_mfma_f32_32x32x16_bf16 with the activations in registers (the C tile of a layer is the B operand of the
//             next: mlp_layout.h) and the 24 KiB of packed weights in LDS.  Training saves only the 32 encoded features.
//   backward  recomputes the MLP forward from the saved features (24 MFMAs), runs the dgrad chain, scatters the
//             gradient of the encoding into the table with hardware fp32 atomics, and writes the d z / activation
//             fragments; the weight gradients of the five small matrices are then the generic split-K pass of the big
//             MLP (mlp_wgrad.h) over those fragments.
// Roofline: gather / atomic traffic, not MFMA — 16 levels x 8 corners x 8 B random reads per sample forward and the same
// number of 4-byte atomics x 2 backward, against 20 KFLOP of matrix work per sample.
#include <type_traits>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"
#include "mlp_wgrad.h"
#include "hashgrid.h"

namespace snr {

using HFrag = Mma<kBF16>::Frag;

// packed fragment indices (1 KiB each): forward stages, then the transposed (dgrad) stages
enum { F_S1 = 0, F_S2 = 4, F_C1 = 8, F_C2 = 12, F_C3 = 20, F_C3T = 24, F_C2T = 26, F_C1T = 34, F_S2T = 38, F_S1T = 40,
       F_TOTAL = 44 };

// ------------------------------------------------------------------------------------------
// pack: thread per (frag, lane)
// ------------------------------------------------------------------------------------------
__global__ void hg_pack_kernel(const float* __restrict__ nets, char* __restrict__ blob) {
  // nets = [sigma_net.params (W1s 64x32, W2s 16x64) | color_net.params (W1c 64x32, W2c 64x64, W3c 16x64)]
  const float* W1s = nets;
  const float* W2s = nets + 2048;
  const float* W1c = nets + 3072;
  const float* W2c = W1c + 2048;
  const float* W3c = W2c + 4096;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int F = gid >> 6, lane = gid & 63;
  if (F >= F_TOTAL) return;
  const int i = lane & 31, g = lane >> 5;
  HFrag out = Mma<kBF16>::zero();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (F < F_S2) {                    // S1: tile t, step q; slot -> feature 16q + 8g + e
      const int t = (F - F_S1) >> 1, q = (F - F_S1) & 1;
      v = W1s[(32 * t + i) * 32 + 16 * q + 8 * g + e];
    } else if (F < F_C1) {             // S2: one tile (rows < 16), slot -> hidden neuron
      const int q = F - F_S2;
      if (i < 16) v = W2s[i * 64 + h_slot_neuron<kBF16>(q, g, e)];
    } else if (F < F_C2) {             // C1: step 0 = SH 8g+e, step 1 = [geo | pad]
      const int t = (F - F_C1) >> 1, q = (F - F_C1) & 1;
      const int col = q == 0 ? 8 * g + e : hg_inc_col(g, e);
      v = W1c[(32 * t + i) * 32 + col];
    } else if (F < F_C3) {             // C2
      const int t = (F - F_C2) >> 2, q = (F - F_C2) & 3;
      v = W2c[(32 * t + i) * 64 + h_slot_neuron<kBF16>(q, g, e)];
    } else if (F < F_C3T) {            // C3: rows < 16
      const int q = F - F_C3;
      if (i < 16) v = W3c[i * 64 + h_slot_neuron<kBF16>(q, g, e)];
    } else if (F < F_C2T) {            // C3^T: rows = hc2 neurons, slot -> output channel 8g+e
      const int t = F - F_C3T;
      v = W3c[(8 * g + e) * 64 + 32 * t + i];
    } else if (F < F_C1T) {            // C2^T: rows = hc1 neurons, slot -> hc2 neuron
      const int t = (F - F_C2T) >> 2, q = (F - F_C2T) & 3;
      v = W2c[h_slot_neuron<kBF16>(q, g, e) * 64 + 32 * t + i];
    } else if (F < F_S2T) {            // C1^T: row rho(r, g') with r < 8 -> sigma-output channel 8g'+r (channel 0: none)
      const int q = F - F_C1T;
      const int gr = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
      const int ch = 8 * gr + r;
      if (r < 8 && ch > 0) v = W1c[h_slot_neuron<kBF16>(q, g, e) * 32 + 15 + ch];
    } else if (F < F_S1T) {            // S2^T: rows = h1 neurons, slot -> sigma-output channel
      const int t = F - F_S2T;
      v = W2s[(8 * g + e) * 64 + 32 * t + i];
    } else {                           // S1^T: row rho(r, g') -> feature 8g' + r (+8 for r >= 8)
      const int q = F - F_S1T;
      const int gr = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
      const int f = 8 * gr + r + (r >= 8 ? 8 : 0);
      v = W1s[h_slot_neuron<kBF16>(q, g, e) * 32 + f];
    }
    out[e] = (__bf16)v;
  }
  *(HFrag*)(blob + ((int64_t)F * 64 + lane) * 16) = out;
}

// ------------------------------------------------------------------------------------------
// encoding
// ------------------------------------------------------------------------------------------
struct HgIn {
  const float* pts;        // [n,3] or null
  const float* rays;       // rows o(3) d(3) ...
  int ray_ld;
  const float* z_vals;     // [n_rays, S]
  const float* viewdirs;   // n_rays rows, leading dimension vd_ld
  int vd_ld;
  int64_t n_samples;
  int S;
};

__device__ __forceinline__ uint32_t hg_index(const HgLevel& L, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t idx;
  if (L.hashed) idx = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
  else idx = x + y * L.res + z * L.res * L.res;
  return idx % L.size;
}

// the 8 corners of level L around position p01: entry index and trilinear weight
__device__ __forceinline__ void hg_corners(const HgLevel& L, float x, float y, float z, uint32_t idx[8], float w[8]) {
  // one rounding, like tiny-cuda-nn's fmaf(scale, x, 0.5f): at the finest levels pos ~ 2e5 and its last bit is 1 % of a cell
  const float px = __builtin_fmaf(x, L.scale, 0.5f), py = __builtin_fmaf(y, L.scale, 0.5f), pz = __builtin_fmaf(z, L.scale, 0.5f);
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const float tx = px - fx, ty = py - fy, tz = pz - fz;
  const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ox = c & 1, oy = (c >> 1) & 1, oz = c >> 2;
    idx[c] = L.offset + hg_index(L, ix + ox, iy + oy, iz + oz);
    w[c] = (ox ? tx : 1.f - tx) * (oy ? ty : 1.f - ty) * (oz ? tz : 1.f - tz);
  }
}

__device__ __forceinline__ void hg_load_sample(const HgIn& a, int64_t m, bool valid, float p[3], float d[3]) {
  p[0] = p[1] = p[2] = 0.f; d[0] = d[1] = 0.f; d[2] = 1.f;
  if (!valid) return;
  const int64_t ray = m / a.S;
  if (a.pts) {
    p[0] = a.pts[3 * m]; p[1] = a.pts[3 * m + 1]; p[2] = a.pts[3 * m + 2];
  } else {
    const float* r = a.rays + ray * a.ray_ld;
    const float t = a.z_vals[m];
    p[0] = mul_add_unfused(r[3], t, r[0]);   // run_nerf.py:670-671, rounded like the reference's separate ops
    p[1] = mul_add_unfused(r[4], t, r[1]);
    p[2] = mul_add_unfused(r[5], t, r[2]);
  }
  const float* v = a.viewdirs + ray * a.vd_ld;
  d[0] = v[0]; d[1] = v[1]; d[2] = v[2];
}

// degree-4 spherical harmonics of direction (x, y, z): the 8 values k-slot half g asks for (g = 0: 0..7, g = 1: 8..15)
__device__ __forceinline__ HFrag hg_sh_frag(float x, float y, float z, int g) {
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  float o[8];
  if (g == 0) {
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
  } else {
    o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[2] = 2.8906114426405538f * xy * z;
    o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[4] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[5] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[6] = 1.4453057213202769f * z * (x2 - y2);
    o[7] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
  }
  HFrag f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (__bf16)o[e];
  return f;
}

// ------------------------------------------------------------------------------------------
// the two MLPs on one 32-sample tile (forward); everything the backward needs is returned in registers
// ------------------------------------------------------------------------------------------
struct HgActs {
  HFrag enc[2], h1[4], inc[2], hc1[4], hc2[4];
  float sigma;     // lanes g == 0
  f32x16 out_c;    // colour output tile: rows 0..2 in registers 0..2 of lanes g == 0
};

__device__ __forceinline__ HFrag relu_frag(const f32x16& acc, int H) {
  f32x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = fmaxf(acc[8 * H + e], 0.f);
  return __builtin_convertvector(t, HFrag);
}

__device__ __forceinline__ f32x16 zero16() { return f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }

__device__ __forceinline__ void hg_mlp_forward(const HFrag* W, int lane, HgActs& A, HFrag shf) {
  using M = Mma<kBF16>;
  const int g = lane >> 5;
  auto wf = [&](int F) { return W[F * 64 + lane]; };
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // sigma layer 1: 32 -> 64, relu
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 2; ++q) acc = M::mma(wf(F_S1 + 2 * t + q), A.enc[q], acc);
    A.h1[2 * t] = relu_frag(acc, 0); A.h1[2 * t + 1] = relu_frag(acc, 1);
  }
  {                               // sigma layer 2: 64 -> 16 (no activation): row 0 = sigma, rows 1..15 = geo features
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = M::mma(wf(F_S2 + q), A.h1[q], acc);
    A.sigma = acc[0];
    A.inc[0] = shf;
    A.inc[1] = M::from_acc<0>(acc);
    if (g == 0) A.inc[1][0] = (__bf16)1.0f;   // the slot of row 0 carries the constant-1 padding input instead of sigma
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // colour layer 1: [SH | geo | 1] -> 64, relu
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 2; ++q) acc = M::mma(wf(F_C1 + 2 * t + q), A.inc[q], acc);
    A.hc1[2 * t] = relu_frag(acc, 0); A.hc1[2 * t + 1] = relu_frag(acc, 1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // colour layer 2: 64 -> 64, relu
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = M::mma(wf(F_C2 + 4 * t + q), A.hc1[q], acc);
    A.hc2[2 * t] = relu_frag(acc, 0); A.hc2[2 * t + 1] = relu_frag(acc, 1);
  }
  {                               // colour layer 3: 64 -> 3 (of 16), no activation
    f32x16 acc = zero16();
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = M::mma(wf(F_C3 + q), A.hc2[q], acc);
    A.out_c = acc;
  }
}

// level 8q + 4g + j of lane half g, selected field by field from two compile-time indices (a run-time index into the
// kernel-argument table would move it to scratch memory)
__device__ __forceinline__ HgLevel hg_pick(const HgTable& T, int q, int j, int g) {
  const HgLevel& a = T.level[8 * q + j];
  const HgLevel& b = T.level[8 * q + 4 + j];
  HgLevel L;
  L.scale = g ? b.scale : a.scale; L.res = g ? b.res : a.res; L.size = g ? b.size : a.size;
  L.offset = g ? b.offset : a.offset; L.hashed = g ? b.hashed : a.hashed;
  return L;
}

// encoding of this lane's 8 levels into the two B fragments
__device__ __forceinline__ void hg_encode(const HgTable& T, const float* __restrict__ grid, const float p[3], int g,
                                          HFrag enc[2]) {
  // tcnn.py:93, with torch's rounding of a division by a scalar on the GPU: add, then multiply by the fp32 reciprocal
  // (one ulp of this value is 1 % of a cell at the finest level)
  const float x = (p[0] + T.bound) * T.inv_2bound, y = (p[1] + T.bound) * T.inv_2bound, z = (p[2] + T.bound) * T.inv_2bound;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const HgLevel L = hg_pick(T, q, j, g);
      uint32_t idx[8];
      float w[8];
      hg_corners(L, x, y, z, idx, w);
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float2 v = *(const float2*)(grid + 2 * (int64_t)idx[c]);
        f0 += w[c] * v.x; f1 += w[c] * v.y;
      }
      enc[q][2 * j] = (__bf16)f0; enc[q][2 * j + 1] = (__bf16)f1;
    }
  }
}

constexpr int kHgWaves = 4;

template <bool TRAIN>
__global__ __launch_bounds__(64 * kHgWaves) void hg_fwd_kernel(HgTable T, const float* __restrict__ grid,
                                                               const char* __restrict__ blob, HgIn a,
                                                               float* __restrict__ raw, char* __restrict__ act) {
  __shared__ __attribute__((aligned(16))) HFrag W[F_C3T * 64];   // forward fragments, 24 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < F_C3T * 64; i += 64 * kHgWaves) W[i] = ((const HFrag*)blob)[i];
  __syncthreads();
  const int sj = lane & 31, g = lane >> 5;
  const int64_t n_tiles = (a.n_samples + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * kHgWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kHgWaves) {
    const int64_t m = tile * 32 + sj;
    const bool valid = m < a.n_samples;
    float p[3], d[3];
    hg_load_sample(a, m, valid, p, d);
    HgActs A;
    hg_encode(T, grid, p, g, A.enc);
    if (!valid) { A.enc[0] = Mma<kBF16>::zero(); A.enc[1] = Mma<kBF16>::zero(); }
    // tcnn.py:100-101 maps d to [0,1] and the SH encoding maps it back to [-1,1]: the direction itself
    hg_mlp_forward(W, lane, A, hg_sh_frag(d[0], d[1], d[2], g));
    if (valid && g == 0) *(f32x4*)(raw + 4 * m) = f32x4{A.out_c[0], A.out_c[1], A.out_c[2], A.sigma};
    if constexpr (TRAIN) {   // the 32 encoded features: [tile][2 frags][64 lanes][16 B]
      HFrag* dst = (HFrag*)(act + tile * 2048);
      dst[lane] = A.enc[0]; dst[64 + lane] = A.enc[1];
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward: recompute, dgrad chain, table scatter, fragments for the weight-gradient pass
// ------------------------------------------------------------------------------------------
// ws sections, [n_tiles][ks KiB] each (the layout mlp_wgrad.h streams): B side then A side
struct HgWs {
  int64_t n_tiles;
  __host__ __device__ explicit HgWs(int64_t n_samples) : n_tiles((n_samples + 31) / 32) {}
  // KiB offsets per tile of each section
  static constexpr int K_ENC = 0, K_H1 = 2, K_INC = 6, K_HC1 = 8, K_HC2 = 12, K_DH1 = 16, K_DOUTS = 20, K_DHC1 = 21,
                       K_DHC2 = 25, K_DRGB = 29, K_TOTAL = 30;
  __host__ __device__ int64_t off(int k) const { return n_tiles * 1024 * k; }
  __host__ __device__ int64_t bytes() const { return n_tiles * 1024 * K_TOTAL; }
};

__device__ __forceinline__ HFrag mask_frag(const f32x16& acc, int H, const HFrag& h) {
  f32x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = ((float)h[e] > 0.f) ? acc[8 * H + e] : 0.f;
  return __builtin_convertvector(t, HFrag);
}

// value held by lane Q of this lane's quad: DPP quad_perm [Q,Q,Q,Q] (no LDS crossbar traffic)
template <int Q> __device__ __forceinline__ int quad_bcast(int x) {
  return __builtin_amdgcn_update_dpp(x, x, Q * 0x55, 0xf, 0xf, false);
}
template <int Q> __device__ __forceinline__ float quad_bcast(float x) {
  return __int_as_float(quad_bcast<Q>(__float_as_int(x)));
}
// One scatter instruction of the table gradient for the cells of quad lane Q's sample: the four lanes of the quad take
// (x-neighbour corner 2P | 2P+1) x (feature 0 | 1) — lane j: corner 2P + (j >> 1), feature j & 1.
template <int Q, int P>
__device__ __forceinline__ void hg_scatter_quad(float* __restrict__ ggrid, const uint32_t idx[8], const float val[16],
                                                bool active, int j) {
  const bool act = quad_bcast<Q>((int)active) != 0;
  const uint32_t i0 = (uint32_t)quad_bcast<Q>((int)idx[2 * P]), i1 = (uint32_t)quad_bcast<Q>((int)idx[2 * P + 1]);
  const float v0 = quad_bcast<Q>(val[4 * P]), v1 = quad_bcast<Q>(val[4 * P + 1]), v2 = quad_bcast<Q>(val[4 * P + 2]),
              v3 = quad_bcast<Q>(val[4 * P + 3]);
  const uint32_t cell = (j & 2) ? i1 : i0;
  const float v = (j & 2) ? ((j & 1) ? v3 : v2) : ((j & 1) ? v1 : v0);
  if (act) __hip_atomic_fetch_add(ggrid + 2 * (int64_t)cell + (j & 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int Q>
__device__ __forceinline__ void hg_scatter_sample(float* __restrict__ ggrid, const uint32_t idx[8], const float val[16],
                                                  bool active, int j) {
  hg_scatter_quad<Q, 0>(ggrid, idx, val, active, j); hg_scatter_quad<Q, 1>(ggrid, idx, val, active, j);
  hg_scatter_quad<Q, 2>(ggrid, idx, val, active, j); hg_scatter_quad<Q, 3>(ggrid, idx, val, active, j);
}

__global__ __launch_bounds__(64 * kHgWaves) void hg_bwd_kernel(HgTable T, const char* __restrict__ blob, HgIn a,
                                                               const float* __restrict__ d_raw,
                                                               const char* __restrict__ act, char* __restrict__ ws,
                                                               float* __restrict__ ggrid) {
  using M = Mma<kBF16>;
  __shared__ __attribute__((aligned(16))) HFrag W[F_TOTAL * 64];   // 44 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < F_TOTAL * 64; i += 64 * kHgWaves) W[i] = ((const HFrag*)blob)[i];
  __syncthreads();
  auto wf = [&](int F) { return W[F * 64 + lane]; };
  const int sj = lane & 31, g = lane >> 5;
  const HgWs WL(a.n_samples);
  // fragment f of a section tile: 1 KiB, lane's 16 bytes at row(sample) * 32 + g * 16; odd fragments swap the two
  // 4-sample groups of every 8 (act_row) — the layout the transposing LDS reads of the weight-gradient kernel expect
  const uint32_t lane_even = g * 16 + sj * 32, lane_odd = g * 16 + act_row<kBF16>(sj, 1) * 32;
  auto store = [&](int k_sec, int ks, int64_t tile, int f, const HFrag& v) {
    char* base = ws + WL.off(k_sec) + (tile * ks + f) * 1024;
    *(HFrag*)(base + ((f & 1) ? lane_odd : lane_even)) = v;
  };
  for (int64_t tile = (int64_t)blockIdx.x * kHgWaves + wave; tile < WL.n_tiles; tile += (int64_t)gridDim.x * kHgWaves) {
    const int64_t m = tile * 32 + sj;
    const bool valid = m < a.n_samples;
    float p[3], d[3];
    hg_load_sample(a, m, valid, p, d);
    HgActs A;
    const HFrag* src = (const HFrag*)(act + tile * 2048);
    A.enc[0] = src[lane]; A.enc[1] = src[64 + lane];
    hg_mlp_forward(W, lane, A, hg_sh_frag(d[0], d[1], d[2], g));

    // d raw -> d rgb fragment (channel 8g + e) and d sigma
    HFrag drgb = M::zero();
    float dsig = 0.f;
    if (valid) {
      const f32x4 dr = *(const f32x4*)(d_raw + 4 * m);
      if (g == 0) { drgb[0] = (__bf16)dr[0]; drgb[1] = (__bf16)dr[1]; drgb[2] = (__bf16)dr[2]; }
      dsig = dr[3];
    }
    HFrag dhc2[4], dhc1[4], dh1[4], douts;
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // d z(hc2) = relu'(hc2) * (W3c^T d rgb)
      f32x16 acc = M::mma(wf(F_C3T + t), drgb, zero16());
      dhc2[2 * t] = mask_frag(acc, 0, A.hc2[2 * t]); dhc2[2 * t + 1] = mask_frag(acc, 1, A.hc2[2 * t + 1]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // d z(hc1) = relu'(hc1) * (W2c^T d z(hc2))
      f32x16 acc = zero16();
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = M::mma(wf(F_C2T + 4 * t + q), dhc2[q], acc);
      dhc1[2 * t] = mask_frag(acc, 0, A.hc1[2 * t]); dhc1[2 * t + 1] = mask_frag(acc, 1, A.hc1[2 * t + 1]);
    }
    {                               // d (sigma-network output) = [d sigma | W1c[:, 16:31]^T d z(hc1)], no activation
      f32x16 acc = zero16();
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = M::mma(wf(F_C1T + q), dhc1[q], acc);
      if (g == 0) acc[0] = dsig;    // register 0 of half 0 = row 0 = sigma (its weight row is zero)
      douts = M::from_acc<0>(acc);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {   // d z(h1) = relu'(h1) * (W2s^T d out_s)
      f32x16 acc = M::mma(wf(F_S2T + t), douts, zero16());
      dh1[2 * t] = mask_frag(acc, 0, A.h1[2 * t]); dh1[2 * t + 1] = mask_frag(acc, 1, A.h1[2 * t + 1]);
    }
    f32x16 denc = zero16();         // d enc: register r <-> feature 8g + r (+8 for r >= 8): this lane's own levels
#pragma unroll
    for (int q = 0; q < 4; ++q) denc = M::mma(wf(F_S1T + q), dh1[q], denc);

#ifndef SNR_HG_ABLATE
#define SNR_HG_ABLATE 0   // timing experiments (results are garbage; tools/build_variant.py): 1 no table scatter at all, 2 none for the
#endif                    // dense levels, 4 no merge rounds, 8 no fragment stores, 16 no scatter for the hashed levels
    // ---- scatter into the table gradient ----
    // Device-scope fp32 atomics (the table gradient is shared by all XCDs: agent scope, performed at the memory side).
    // The 32 samples of a tile are consecutive samples of one ray, so at the coarse levels they fall into a handful of
    // cells — and the whole batch into a few dozen entries of level 0: unmerged, those addresses serialise millions of
    // atomics (measured 18 ms per 262 144-sample step).  So each half-wave first merges the lanes that share a cell
    // (up to kMerge distinct cells per level: leader's key broadcast, masked butterfly sums, the leader issues the 16
    // atomics); lanes left over after that issue their own.
    {
      const float x = (p[0] + T.bound) * T.inv_2bound, y = (p[1] + T.bound) * T.inv_2bound, z = (p[2] + T.bound) * T.inv_2bound;
      constexpr int kMerge = 4;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const HgLevel L = hg_pick(T, q, j, g);
          uint32_t idx[8];
          float w[8];
          hg_corners(L, x, y, z, idx, w);
          const float g0 = denc[8 * q + 2 * j], g1 = denc[8 * q + 2 * j + 1];
          float val[16];
#pragma unroll
          for (int c = 0; c < 8; ++c) { val[2 * c] = w[c] * g0; val[2 * c + 1] = w[c] * g1; }
          // cell key: the integer coordinates of corner 0 (21 bits each are plenty: scale < 2^18)
          const float fx = floorf(__builtin_fmaf(x, L.scale, 0.5f)), fy = floorf(__builtin_fmaf(y, L.scale, 0.5f)),
                      fz = floorf(__builtin_fmaf(z, L.scale, 0.5f));
          const uint32_t k_lo = (uint32_t)(int)fx | ((uint32_t)(int)fy << 21);
          const uint32_t k_hi = ((uint32_t)(int)fy >> 11) | ((uint32_t)(int)fz << 10);
          bool active = valid;
#if SNR_HG_ABLATE & 1
          active = false;
#endif
#if SNR_HG_ABLATE & 2
          if (!L.hashed) active = false;
#endif
#if SNR_HG_ABLATE & 16
          if (L.hashed) active = false;
#endif
          // consecutive samples of a ray are neighbouring lanes: when no lane shares its cell with the lane before it
          // (the fine levels), there is nothing to merge and the leader rounds would be wasted
          const bool dup = k_lo == (uint32_t)__shfl_up((int)k_lo, 1, 64) && k_hi == (uint32_t)__shfl_up((int)k_hi, 1, 64) &&
                           (lane & 31) != 0;
#if SNR_HG_ABLATE & 4
          const int rounds = 0;
#else
          const int rounds = __ballot(dup && active) ? kMerge : 0;
#endif
          for (int it = 0; it < rounds; ++it) {
            const unsigned long long bal = __ballot(active);
            if (bal == 0) break;                                   // wave-uniform
            const uint32_t half = g ? (uint32_t)(bal >> 32) : (uint32_t)bal;
            const int leader = half ? (__builtin_ctz(half) + 32 * g) : lane;   // an idle half follows itself: no match
            const uint32_t l_lo = __shfl(k_lo, leader, 64), l_hi = __shfl(k_hi, leader, 64);
            const bool match = active && half != 0 && k_lo == l_lo && k_hi == l_hi;
            // every lane of the half-wave ends up with the sums; the leader's quad issues them: lane j of it takes corner
            // 2P + (j >> 1), feature j & 1 of the leader's cell (see the request rule below)
            const bool writer = half != 0 && (lane | 3) == (leader | 3);
            const int jw = lane & 3;
            uint32_t l_idx[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) l_idx[c] = (uint32_t)__shfl((int)idx[c], leader, 64);
#pragma unroll
            for (int P = 0; P < 4; ++P) {
              float sm[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                sm[e] = match ? val[4 * P + e] : 0.f;
#pragma unroll
                for (int d = 1; d < 32; d <<= 1) sm[e] += __shfl_xor(sm[e], d, 64);
              }
              const uint32_t cell = (jw & 2) ? l_idx[2 * P + 1] : l_idx[2 * P];
              const float v = (jw & 2) ? ((jw & 1) ? sm[3] : sm[2]) : ((jw & 1) ? sm[1] : sm[0]);
              if (writer)
                __hip_atomic_fetch_add(ggrid + 2 * (int64_t)cell + (jw & 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            active = active && !match;
          }
          // The atomic unit's rate is per (instruction, 64-byte line) request, not per lane
          // (tests/probes/atomic_rate.hip: 21 G requests/s from 64 KB to 256 MB targets; lanes of one instruction that fall
          // into one line cost one request).  The two features of a cell are 8 bytes, and the x-neighbour of a corner with
          // even x is the next cell (dense levels: always; hashed levels: prime 1 on x keeps idx ^ 1).  So the four lanes of
          // a quad work together on one of their samples at a time: 16-24 requests per instruction instead of 64.
          {
            const int jq = lane & 3;
            hg_scatter_sample<0>(ggrid, idx, val, active, jq); hg_scatter_sample<1>(ggrid, idx, val, active, jq);
            hg_scatter_sample<2>(ggrid, idx, val, active, jq); hg_scatter_sample<3>(ggrid, idx, val, active, jq);
          }
        }
      }
    }

    // ---- fragments for the weight-gradient pass ----
#if !(SNR_HG_ABLATE & 8)
#pragma unroll
    for (int f = 0; f < 2; ++f) { store(HgWs::K_ENC, 2, tile, f, A.enc[f]); store(HgWs::K_INC, 2, tile, f, A.inc[f]); }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      store(HgWs::K_H1, 4, tile, f, A.h1[f]); store(HgWs::K_HC1, 4, tile, f, A.hc1[f]);
      store(HgWs::K_HC2, 4, tile, f, A.hc2[f]);
      store(HgWs::K_DH1, 4, tile, f, dh1[f]); store(HgWs::K_DHC1, 4, tile, f, dhc1[f]);
      store(HgWs::K_DHC2, 4, tile, f, dhc2[f]);
    }
    store(HgWs::K_DOUTS, 1, tile, 0, douts);
    store(HgWs::K_DRGB, 1, tile, 0, drgb);
#endif
  }
}

// weight-gradient jobs over the fragments hg_bwd_kernel wrote (both sides live in `ws`)
static WgradArgs hg_make_jobs(int64_t n_samples, int64_t net_off, int64_t* part_floats, int* total_splits) {
  const HgWs WL(n_samples);
  WgradArgs A{};
  A.n_tiles = WL.n_tiles;
  const WgradSec none{0, 0};
  int n = 0, no = 0;
  auto job = [&](int ka, int a_ks, int kb, int b_ks) {
    WgradJob& J = A.job[n];
    J.a[0] = WgradSec{WL.off(ka), a_ks}; J.a[1] = none; J.b[0] = WgradSec{WL.off(kb), b_ks}; J.b[1] = none;
    J.a_ks = a_ks; J.b_ks = b_ks;
    J.nta = (a_ks * 16 + 31) / 32; J.ntb = (b_ks * 16 + 31) / 32;
    return n++;
  };
  auto out = [&](int j, int rows, int a_kind, int cols, int b_kind, int64_t w_off, int ld, int rows_valid, int cols_valid) {
    WgradOut& O = A.out[no++];
    O.job = j; O.row0 = 0; O.rows = rows; O.a_kind = a_kind; O.col0 = 0; O.cols = cols; O.b_kind = b_kind; O.L = 0;
    O.w_off = (int)w_off; O.ld = ld; O.col_off = 0; O.row_off = 0; O.rows_valid = rows_valid; O.cols_valid = cols_valid;
    O.bias_off = -1; O.to_scratch = 0;
  };
  // parameter offsets inside the flat buffer (behind the table): W1s 64x32, W2s 16x64, W1c 64x32, W2c 64x64, W3c 16x64
  const int64_t w1s = net_off, w2s = w1s + 2048, w1c = w2s + 1024, w2c = w1c + 2048, w3c = w2c + 4096;
  out(job(HgWs::K_DH1, 4, HgWs::K_ENC, 2), 64, SRC_H, 32, SRC_NAT, w1s, 32, 64, 32);
  out(job(HgWs::K_DOUTS, 1, HgWs::K_H1, 4), 16, SRC_OUT, 64, SRC_H, w2s, 64, 16, 64);
  out(job(HgWs::K_DHC1, 4, HgWs::K_INC, 2), 64, SRC_H, 32, SRC_HG_INC, w1c, 32, 64, 32);
  out(job(HgWs::K_DHC2, 4, HgWs::K_HC1, 4), 64, SRC_H, 64, SRC_H, w2c, 64, 64, 64);
  out(job(HgWs::K_DRGB, 1, HgWs::K_HC2, 4), 16, SRC_OUT, 64, SRC_H, w3c, 64, 16, 64);
  A.n_jobs = n; A.n_outs = no;
  const int target = cu_count();
  int64_t cost = 0;
  for (int i = 0; i < n; ++i) cost += A.job[i].a_ks + A.job[i].b_ks;
  int sb = 0;
  int64_t po = 0;
  for (int i = 0; i < n; ++i) {
    WgradJob& J = A.job[i];
    int64_t s = (int64_t)target * (J.a_ks + J.b_ks) / cost;
    if (s < 1) s = 1;
    if (s > A.n_tiles) s = A.n_tiles;
    J.n_splits = (int)s; J.split_begin = sb; sb += (int)s;
    J.part_off = po; po += s * J.nta * 32 * J.ntb * 32;
    J.bias_part_off = po; po += s * J.nta * 32;
  }
  *part_floats = po;
  *total_splits = sb;
  return A;
}

}  // namespace snr

using namespace snr;

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static HgTable hg_table() {
  HgTable T{};
  T.bound = 100.f;                                                             // tcnn.py:22
  T.inv_2bound = (float)(1.0 / 200.0);
  // tcnn.py:34.  The level scales are formed in double precision and rounded once (tiny-cuda-nn evaluates
  // exp2f(level * log2f(per_level_scale)) * base - 1 in fp32, whose last bit depends on the libm at hand — and one ulp of
  // the finest scale is 2 % of a cell at the far corner of the cube; parity is unpinned either way, this is reproducible)
  const double log2_pls = log2(2048.0 * 100.0 / 16.0) / 15.0;
  uint32_t off = 0;
  for (int l = 0; l < kHgLevels; ++l) {
    const float scale = (float)(exp2((double)l * log2_pls) * 16.0 - 1.0);
    const uint32_t res = (uint32_t)ceilf(scale) + 1;
    uint64_t dense = (uint64_t)res * res * res;
    if (dense > 0xFFFFFFFFull) dense = 0xFFFFFFFFull;
    uint64_t n = (dense + 7) / 8 * 8;
    if (n > (1ull << 19)) n = 1ull << 19;
    T.level[l].scale = scale; T.level[l].res = res; T.level[l].size = (uint32_t)n; T.level[l].offset = off;
    T.level[l].hashed = (uint64_t)res * res * res > n;
    off += (uint32_t)n;
  }
  T.entries = off;
  return T;
}

extern "C" int64_t snr_hashgrid_table_entries(void) { return hg_table().entries; }
extern "C" int64_t snr_hashgrid_param_count(void) { return (int64_t)hg_table().entries * 2 + kHgNetParams; }
extern "C" int64_t snr_hashgrid_packed_bytes(void) { return (int64_t)F_TOTAL * 1024; }
extern "C" int64_t snr_hashgrid_act_bytes(int64_t n) { return n <= 0 ? SNR_ERR_SHAPE : (n + 31) / 32 * 2048; }
extern "C" int64_t snr_hashgrid_bwd_ws_bytes(int64_t n) {
  if (n <= 0) return SNR_ERR_SHAPE;
  int64_t pf; int ts;
  hg_make_jobs(n, 0, &pf, &ts);
  return HgWs(n).bytes() + pf * 4;
}

extern "C" int snr_hashgrid_pack(const float* params, void* packed, snr_stream_t stream) {
  SNR_CHECK_ARG(params && packed, SNR_ERR_NULL);
  const float* nets = params + (int64_t)hg_table().entries * 2;
  {
    ProfScope ps(K_HG_PACK, (hipStream_t)stream);
    hg_pack_kernel<<<dim3((F_TOTAL * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(nets, (char*)packed);
  }
  return launch_status();
}

static int hg_check_in(const float* pts, const float* rays, int ray_ld, const float* z_vals, const float* viewdirs,
                       int vd_ld, int64_t n, int S) {
  SNR_CHECK_ARG(pts || (rays && z_vals), SNR_ERR_NULL);
  SNR_CHECK_ARG(viewdirs && vd_ld >= 3, SNR_ERR_NULL);   // the network reads input[:, 3:] (tcnn.py:88): viewdirs are required
  SNR_CHECK_ARG(n > 0 && S > 0, SNR_ERR_SHAPE);
  SNR_CHECK_ARG(pts || ray_ld >= 6, SNR_ERR_SHAPE);
  return SNR_OK;
}

extern "C" int snr_hashgrid_forward(const float* params, const void* packed, const float* pts, const float* rays,
                                    int ray_ld, const float* z_vals, const float* viewdirs, int viewdirs_ld,
                                    int64_t n_samples, int samples_per_ray, float* raw, void* act, snr_stream_t stream) {
  SNR_CHECK_ARG(params && packed && raw, SNR_ERR_NULL);
  int st = hg_check_in(pts, rays, ray_ld, z_vals, viewdirs, viewdirs_ld, n_samples, samples_per_ray);
  if (st != SNR_OK) return st;
  const HgTable T = hg_table();
  HgIn a{pts, rays, ray_ld, z_vals, viewdirs, viewdirs_ld, n_samples, samples_per_ray};
  const int64_t n_tiles = (n_samples + 31) / 32;
  int64_t grid = (n_tiles + kHgWaves - 1) / kHgWaves;
  if (grid > 4096) grid = 4096;
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(K_HG_FWD, s);
    if (act) hg_fwd_kernel<true><<<dim3((unsigned)grid), dim3(64 * kHgWaves), 0, s>>>(T, params, (const char*)packed, a, raw, (char*)act);
    else hg_fwd_kernel<false><<<dim3((unsigned)grid), dim3(64 * kHgWaves), 0, s>>>(T, params, (const char*)packed, a, raw, nullptr);
  }
  return launch_status();
}

extern "C" int snr_hashgrid_backward(const float* params, const void* packed, const float* pts, const float* rays,
                                     int ray_ld, const float* z_vals, const float* viewdirs, int viewdirs_ld,
                                     const float* d_raw, int64_t n_samples, int samples_per_ray, const void* act, void* ws,
                                     float* grad, int accumulate, snr_stream_t stream) {
  SNR_CHECK_ARG(params && packed && d_raw && act && ws && grad, SNR_ERR_NULL);
  int st = hg_check_in(pts, rays, ray_ld, z_vals, viewdirs, viewdirs_ld, n_samples, samples_per_ray);
  if (st != SNR_OK) return st;
  const HgTable T = hg_table();
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) {   // the table gradient is accumulated with atomics: clear it; the MLP blocks are stored by the reduce
    hipError_t e = hipMemsetAsync(grad, 0, (size_t)T.entries * 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  HgIn a{pts, rays, ray_ld, z_vals, viewdirs, viewdirs_ld, n_samples, samples_per_ray};
  const int64_t n_tiles = (n_samples + 31) / 32;
  int64_t grid = (n_tiles + kHgWaves - 1) / kHgWaves;
  if (grid > 4096) grid = 4096;
  {
    ProfScope ps(K_HG_BWD, s);
    hg_bwd_kernel<<<dim3((unsigned)grid), dim3(64 * kHgWaves), 0, s>>>(T, (const char*)packed, a, d_raw, (const char*)act,
                                                                       (char*)ws, grad);
  }
  st = launch_status();
  if (st != SNR_OK) return st;
  int64_t pf; int total_splits;
  WgradArgs w = hg_make_jobs(n_samples, (int64_t)T.entries * 2, &pf, &total_splits);
  w.act = (const char*)ws;     // both operand sides were written into ws
  w.ws = (const char*)ws;
  w.part = (float*)((char*)ws + HgWs(n_samples).bytes());
  w.post = nullptr;
  return wgrad_launch_bf16(w, total_splits, grad, accumulate, s);
}
