// Fused positional-encoding + 8x256 NeRF MLP forward for gfx950, and the weight pack kernel.
//
// Replaces run_network()/NeRF.forward of the reference (DS_NeRF/run_nerf.py:56-71,
// DS_NeRF/run_nerf_helpers.py:22-70, 104-127).  See mlp_layout.h for the data layout and
// DESIGN.md for the roofline.
//
// Workgroup = ChainCfg waves (bf16: 8, two per SIMD; fp32: 4); each wave owns a tile of 32 samples and
// carries its activations in registers through all 11 linear layers (the C tile of layer i is, after
// bias/ReLU/convert, directly the B operand of layer i+1).  The packed weights stream global->LDS
// (global_load_lds_dwordx4) through a ring of 16 KiB blocks shared by the waves, several blocks ahead of the
// MFMAs that consume them (mlp_device.h: Pipe); fragment and bias reads are inline asm with counted waits.
#include <type_traits>

#include "snr_common.h"
#include "mlp_pack.h"
#include "mlp_device.h"

namespace snr {

// ------------------------------------------------------------------------------------------
// pack kernel: one thread per (frag, lane) = 16 bytes of the blob
// ------------------------------------------------------------------------------------------
template <int P>
__global__ void mlp_pack_kernel(PackTable T, const float* __restrict__ params, char* __restrict__ blob) {
  using Frag = typename Mma<P>::Frag;
  constexpr int EPF = Prec<P>::EPF;
  const int total_frags = T.fwd_frags + T.bwd_frags;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // the frag index is wave-uniform: keep the entry search on the scalar unit, all of its loads in flight at once (a
  // data-dependent while loop here chained up to 26 kernarg loads and was most of this kernel's 19 us)
  const int F = __builtin_amdgcn_readfirstlane((int)(gid >> 6)), lane = (int)(gid & 63);
  if (F < total_frags) {
    int ei = 0;
#pragma unroll
    for (int k = 1; k < kMaxPackEntries; ++k) ei += (k < T.n_entries && T.e[k].frag_begin <= F) ? 1 : 0;
    const PackEntry& E = T.e[ei];
    const int per_tile = E.src[0].ks + E.src[1].ks;
    const int local = F - E.frag_begin;
    const int tile = local / per_tile;
    int f = local % per_tile;
    const int s = f >= E.src[0].ks ? 1 : 0;
    if (s) f -= E.src[0].ks;
    const PackSrc& S = E.src[s];
    const int i = lane & 31, g = lane >> 5;
    const int row = 32 * tile + i;
    Frag out = Mma<P>::zero();
#pragma unroll
    for (int e = 0; e < EPF; ++e) {
      int slot;  // column (forward) or weight row (dgrad) this k-slot stands for; -1 = padding
      if (S.kind == SRC_ENC_PTS) slot = enc_slot_feature<P>(f, g, e, T.multires);
      else if (S.kind == SRC_ENC_DIR) slot = enc_slot_feature<P>(f, g, e, T.multires_views);
      else if (S.kind == SRC_H) slot = h_slot_neuron<P>(f, g, e);
      else {  // SRC_OUT: bf16 slot 8g+e, fp32 slot 2e+g  -> raw channel
        slot = (P == kBF16) ? 8 * g + e : 2 * e + g;
      }
      float v = 0.f;
      if (row < E.rows_valid && slot >= 0) {
        if (!E.transposed) {
          v = params[S.w_off + (int64_t)row * S.ld + S.col_off + slot];
        } else {
          const int n = slot - S.slot_off;
          if (n >= 0 && n < S.n_valid) v = params[S.w_off + (int64_t)n * S.ld + S.col_off + row];
        }
      }
      // (bf16 mode: the columns that multiply an encoding are fp16, like the encoding itself — mlp_layout.h: EncF16)
      if (EncF16<P>::value && (S.kind == SRC_ENC_PTS || S.kind == SRC_ENC_DIR)) Mma<P>::set_f16(out, e, v);
      else Mma<P>::set(out, e, v);
    }
    *(Frag*)(blob + ((int64_t)F * 64 + lane) * 16) = out;
  }
  // biases (fp32, true order, zero padded) live behind the frags
  float* bias = (float*)(blob + (int64_t)total_frags * 1024);
  if (gid < T.bias_floats) {
    float v = 0.f;
    for (int b = 0; b < T.n_bias; ++b) {
      const int r = (int)gid - T.b[b].dst;
      if (r >= 0 && r < T.b[b].count) v = r < T.b[b].n_valid ? params[T.b[b].src + r] : 0.f;
    }
    bias[gid] = v;
  }
}

struct FwdArgs {
  const char* blob;        // packed weights (forward section first)
  int fwd_blocks;          // 16 KiB blocks in the forward section (the cyclic DMA stream)
  const float* bias;       // bias block inside the blob
  int bias_floats;
  const float* pts;        // [n,3] or null
  const float* rays;       // rows o(3) d(3) ...
  int ray_ld;
  const float* z_vals;     // [n_rays, S]
  const float* viewdirs;   // n_rays rows of 3, leading dimension vd_ld
  int vd_ld;
  int64_t n_samples;
  int S;
  int multires, multires_views;
  int out_ch;
  float* raw;
  char* act;               // null = inference
  int save_even;           // 0: h0, h2, h4, h6 are not saved (the weight-gradient pass rebuilds them, mlp_wgrad_pair.h)
  int enc_generic;         // 1: the run-time encoding everywhere (SNR_ENC_GENERIC: the bit-identity test of encode_static)
};

template <int P, bool VD, bool TRAIN>
__global__ __launch_bounds__((64 * ChainCfg<P, TRAIN>::WAVES)) void mlp_fwd_kernel(FwdArgs a) {
  using B = Blob<P>;
  using M = Mma<P>;
  using Frag = typename M::Frag;
  constexpr int FPT = Prec<P>::FPT, NJ = ChainCfg<P, TRAIN>::NJ, WAVES = ChainCfg<P, TRAIN>::WAVES;
  constexpr int KS_H = B::KS_H, KS_PE = B::KS_PE, KS_DIR = B::KS_DIR, KS_H9 = B::KS_H9;
  constexpr int KS_DIRA = VD ? KS_DIR : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* bias_lds = (float*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sj = lane & 31, g = lane >> 5;   // sample within a 32-sample tile, lane half

  // ---- (round 5 experiment, SNR_IN_PREFETCH=1; off by default) a pass's per-sample inputs through LDS, one pass ahead -----
  // t = z_vals[m], the ray's origin and direction, its view direction: ten floats per sample behind a dependent chain
  // (sample -> ray -> row) of global loads at the top of every pass.  An ablation without those loads was 5 % (training
  // forward) / 8 % (frame) faster — but it also fed the network smooth positions, and the gain was the CLOCK of quieter
  // operands: with the real inputs prefetched as below the forward is 0.275 against 0.279 ms and the frame 42.9 against
  // 42.6 ms (profiles/r05_fwd_prologue_ab.txt).  The mechanism, for the record: each wave
  // fetches its next tile's ten fields with five LDS-DMA dwords per lane (lane (sample, half g), instruction i: field
  // 2 i + g — 0 = t, 1..6 = the ray's row, 7..9 = the view direction — lands at dword 64 i + lane of the wave's staging
  // area) while the current pass multiplies, and reads them back at the top of the next pass (field f of sample s =
  // dword 32 f + s).  The first pass's fetch is issued before the bias copy and the ring fill, under their latency.
  // Only the issuing wave reads its area: its own counted vmcnt wait orders the two.
  constexpr int kInBytes = 5 * 256;
  char* const in_stage = smem + kBiasLdsBytes + kRingBytes + wave * (NJ * kInBytes);
  const bool small_n = SNR_ENC_STATIC && a.n_samples <= 0x7fffffffll;   // 32-bit sample -> ray division (the 64-bit one: ~100 instructions)
  const bool staged = SNR_IN_PREFETCH && a.pts == nullptr;
  auto prefetch_inputs = [&](int64_t wgn) {
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
      int64_t mt = ((wgn * WAVES + wave) * NJ + jt) * 32 + sj;
      if (mt >= a.n_samples) mt = a.n_samples - 1;   // (a tile's tail beyond n_samples is masked by `valid`: any readable sample)
      uint32_t s32 = (uint32_t)a.S;
      asm volatile("" : "+s"(s32));
      const int64_t ray = small_n ? (int64_t)((uint32_t)mt / s32) : mt / a.S;
      const float* pr = a.rays + ray * a.ray_ld;
      const float* pv = VD ? a.viewdirs + ray * a.vd_ld : pr;
      const float* src[5] = {g ? pr : a.z_vals + mt, pr + 1 + g, pr + 3 + g, g ? pv : pr + 5, pv + 1 + g};
      char* dst = in_stage + jt * kInBytes;
#pragma unroll
      for (int i = 0; i < (VD ? 5 : 4); ++i) {
        asm volatile("s_nop 0");   // (no LDS read in the cycle in front of an LDS-DMA: mlp_device.h, Pipe::issue_one)
        __builtin_amdgcn_global_load_lds(src[i], SNR_LDS(dst + 256 * i), 4, 0, 0);
      }
    }
  };
  if (staged) prefetch_inputs(blockIdx.x);

  for (int i = tid; i < a.bias_floats; i += 64 * WAVES) bias_lds[i] = a.bias[i];

  Pipe<P, WAVES> pipe;
  pipe.init(smem + kBiasLdsBytes, a.blob, a.fwd_blocks, wave, lane);
  __syncthreads();  // bias block visible to all waves

  const ActLayout<P> AL(a.n_samples, VD);
  const int64_t n_wg = (AL.n_tiles + WAVES * NJ - 1) / (WAVES * NJ);   // tail tiles beyond n_samples are masked by `valid`
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IPE = std::integral_constant<int, KS_PE>;
  using IH = std::integral_constant<int, KS_H>;
  using IDIR = std::integral_constant<int, KS_DIR>;
  using IDIRA = std::integral_constant<int, KS_DIRA>;

  for (int64_t wg = blockIdx.x; wg < n_wg; wg += gridDim.x) {
    const int64_t tile0 = (wg * WAVES + wave) * NJ;   // this wave's first 32-sample tile
#ifdef SNR_TIMING
    const unsigned long long tp0 = SNR_T();
#endif
    int64_t m[NJ];
    bool valid[NJ];
    float px[NJ], py[NJ], pz[NJ], dx[NJ], dy[NJ], dz[NJ];
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) {
      m[jt] = (tile0 + jt) * 32 + sj;
      valid[jt] = m[jt] < a.n_samples;
      px[jt] = py[jt] = pz[jt] = dx[jt] = dy[jt] = dz[jt] = 0.f;
      if (staged) {
        // everything older than the youngest PIECES x kDepth vector-memory operations of this wave has completed: the fetch of
        // these fields is older (behind it: the ring fill / at least four blocks' pieces of the previous pass)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Pipe<P, WAVES>::PIECES * kDepth) : "memory");
        const uint32_t ia = lds_addr(in_stage + jt * kInBytes) + 4u * (uint32_t)sj;
        float f[10];
#define SNR_IN_RD(I) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[I]) : "v"(ia), "n"(128 * (I)))
        SNR_IN_RD(0); SNR_IN_RD(1); SNR_IN_RD(2); SNR_IN_RD(3); SNR_IN_RD(4); SNR_IN_RD(5); SNR_IN_RD(6);
        if constexpr (VD) { SNR_IN_RD(7); SNR_IN_RD(8); SNR_IN_RD(9); }
#undef SNR_IN_RD
        if constexpr (VD) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]));
        if (valid[jt]) {
          // run_nerf.py:670-671; separate multiply and add (no FMA) so pts round like the reference's
          px[jt] = mul_add_unfused(f[4], f[0], f[1]);
          py[jt] = mul_add_unfused(f[5], f[0], f[2]);
          pz[jt] = mul_add_unfused(f[6], f[0], f[3]);
          if constexpr (VD) { dx[jt] = f[7]; dy[jt] = f[8]; dz[jt] = f[9]; }
        }
        continue;
      }
      // (the ray of a sample: a 32-bit division where the launch allows it.  The divisor is an opaque copy: hoisted out of the
      //  pass loop, the division's reciprocal constant was spilled to scratch memory and re-loaded behind a vmcnt(0) at the top
      //  of every pass)
      uint32_t s32 = (uint32_t)a.S;
      asm volatile("" : "+s"(s32));
      const int64_t ray = !valid[jt] ? 0 : (small_n ? (int64_t)((uint32_t)m[jt] / s32) : m[jt] / a.S);
      if (valid[jt]) {
#if SNR_ABLATE & 256   // timing experiment (round 5): no global loads in the pass prologue (positions from the sample index)
        px[jt] = 1e-6f * (float)(int)m[jt]; py[jt] = 0.5f * px[jt]; pz[jt] = 1.f - px[jt];
        if constexpr (VD) { dx[jt] = px[jt]; dy[jt] = py[jt]; dz[jt] = pz[jt]; }
#else
        // (the view direction first: its load goes out with the ray's, not behind the wait for them — one round trip per pass)
        if constexpr (VD) {
          const float* v = a.viewdirs + ray * a.vd_ld;
          dx[jt] = v[0]; dy[jt] = v[1]; dz[jt] = v[2];
        }
        if (a.pts) {
          px[jt] = a.pts[3 * m[jt]]; py[jt] = a.pts[3 * m[jt] + 1]; pz[jt] = a.pts[3 * m[jt] + 2];
        } else {
          const float* r = a.rays + ray * a.ray_ld;
          const float t = a.z_vals[m[jt]];
          // run_nerf.py:670-671; separate multiply and add (no FMA) so pts round like the reference's
          px[jt] = mul_add_unfused(r[3], t, r[0]);
          py[jt] = mul_add_unfused(r[4], t, r[1]);
          pz[jt] = mul_add_unfused(r[5], t, r[2]);
        }
#endif
      }
    }
    // Encodings.  With two waves per SIMD (256 registers each) they are not kept across the trunk:
    // bf16 re-derives them where they are consumed (hardware sin/cos, ~200 instructions) — `fresh`
    // hides the inputs from common-subexpression elimination so the first copy really dies.
    constexpr bool kKeepEnc = P == kFP32;
    constexpr bool E16 = EncF16<P>::value;   // the encodings (and the weight columns they meet) are fp16 in bf16 mode
    Frag pe[NJ][KS_PE];
    Frag dir[NJ][KS_DIRA];
    auto make_pe = [&](bool fresh) {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) {
        float x = px[jt], y = py[jt], z = pz[jt];
        if (fresh) asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
        encode_auto<P, KS_PE, kMaxMultires, E16>(x, y, z, a.enc_generic ? -1 : a.multires, a.multires, g, pe[jt]);
      }
    };
    auto make_dir = [&](bool fresh) {
      if constexpr (VD) {
#pragma unroll
        for (int jt = 0; jt < NJ; ++jt) {
          float x = dx[jt], y = dy[jt], z = dz[jt];
          if (fresh) asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
          encode_auto<P, KS_DIR, kMaxMultiresViews, E16>(x, y, z, a.enc_generic ? -1 : a.multires_views, a.multires_views, g, dir[jt]);
        }
      }
    };
    make_pe(false);
    if (TRAIN || kKeepEnc) make_dir(false);

    // frags [n*nt/NT, n*(nt+1)/NT) of an n-frag section, for every sample tile of this wave: output tile
    // nt's share of the deferred stores.  Section layout [tile][frag][32 samples][32 B].
    // Addresses are (wave-uniform 64-bit base) + (32-bit lane offset): one SGPR pair and no vector address
    // arithmetic per store.
    const uint32_t lane_even = g * 16 + sj * 32, lane_odd = g * 16 + act_row<P>(sj, 1) * 32;
    auto sec_base = [&](int k_sec, int64_t tile, int n) {
#if SNR_ABLATE & 32   // timing experiment: every store hits the same 16 KiB per wave (no HBM traffic)
      return a.act + (int64_t)wave * 16384;
#endif
      return a.act + (AL.n_tiles * k_sec + tile * n) * 1024;
    };
    auto act_store_cvt = [&](auto CVT_, auto PAR_, int k_sec, auto N_, const Frag* src, auto STRIDE_, int nt, auto NT_) {
      constexpr int n = decltype(N_)::value, stride = decltype(STRIDE_)::value, NT = decltype(NT_)::value;
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt)
        store_tile_slice<P, n, NT, decltype(PAR_)::value, decltype(CVT_)::value>(sec_base(k_sec, tile0 + jt, n), src + jt * stride, nt, lane_even, lane_odd);
    };
    auto act_store_par = [&](auto PAR_, int k_sec, auto N_, const Frag* src, auto STRIDE_, int nt, auto NT_) {
      act_store_cvt(std::false_type{}, PAR_, k_sec, N_, src, STRIDE_, nt, NT_);
    };
    using PALL = std::integral_constant<int, -1>;
    // a saved hidden layer's even fragments are stored by the stage that produces them, the odd ones by the next stage (mlp_device.h)
    constexpr bool kSplit = SNR_STORE_SPLIT && P == kBF16;
    using PEVEN = std::integral_constant<int, kSplit ? 0 : -2>;   // -2: nothing from the producing stage
    using PODD = std::integral_constant<int, kSplit ? 1 : -1>;
    auto act_store = [&](int k_sec, auto N_, const Frag* src, auto STRIDE_, int nt, auto NT_) {
      act_store_par(PALL{}, k_sec, N_, src, STRIDE_, nt, NT_);
    };
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I9 = std::integral_constant<int, 9>;
    using IH9 = std::integral_constant<int, KS_H9>;
    auto mask_store = [&](int k_sec, const u32x4* mk) {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) store16_stream<0>(sec_base(k_sec, tile0 + jt, 1), lane * 16u, mk[jt]);
    };

    Frag hA[NJ][KS_H], hB[NJ][KS_H];
    u32x4 mask[NJ];

    // epilogue of one output tile: [relu] -> frags, flags of the positive outputs into mask[jt]
    // (bf16: packed, mlp_device.h; fp32: bit r of the tile's 16-bit field = C register r)
    auto tile_out = [&](auto RELU_, int nt, int jt, f32x16 acc, Frag* dst) {
      constexpr bool RELU = decltype(RELU_)::value;
      if constexpr (P == kBF16) {
        unsigned word = 0;
        finish_fwd_bf16<RELU, RELU && TRAIN>(acc, dst, word, 8 * (nt & 1));
        if constexpr (RELU && TRAIN) mask[jt][nt >> 1] |= word;
      } else {
        if constexpr (RELU) {
          unsigned bits = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            bits |= (acc[r] > 0.f ? 1u : 0u) << r;
            acc[r] = fmaxf(acc[r], 0.f);
          }
          if constexpr (TRAIN) mask[jt][nt >> 1] |= bits << (16 * (nt & 1));
        }
        acc_to_frags<P>(acc, dst);
      }
    };
    auto clear_masks = [&]() {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) mask[jt] = u32x4{0, 0, 0, 0};
    };
    // ---- generic 8-tile stage: dst = [relu](W [sa|sb] + b); sources / dst have per-sample-tile strides ----
    // (a stage whose FIRST source is an encoding — stages 0 and 5 — multiplies that segment in fp16: KA == KS_PE marks it)
    auto stage8 = [&](auto KA_, auto KB_, auto SA_, auto SB_, const Frag* sa, const Frag* sb, Frag* dst, int bias_off,
                      auto&& pre, int s_out) {
      constexpr int KA = decltype(KA_)::value, KB = decltype(KB_)::value;
      constexpr int SA = decltype(SA_)::value, SB = decltype(SB_)::value;
      constexpr bool A16 = E16 && KA == KS_PE;
      clear_masks();
      pipe.template run_tiles<KA, KB, 8, NJ, SA, SB, A16, false>(
          sa, sb, [&](int nt) { return bias_tile_addr(bias_lds, bias_off + 32 * nt, g); },
          [&](int nt, int jt, f32x16 acc) {
            tile_out(std::true_type{}, nt, jt, acc, dst + jt * KS_H + nt * FPT);
            if constexpr (TRAIN && PEVEN::value == 0) {   // the fragment pair just written: its even half leaves now
              if (jt == NJ - 1 && (a.save_even || (s_out & 1))) act_store_par(PEVEN{}, AL.k_h(s_out), IH{}, dst, IH{}, nt, I8{});
            }
          },
          pre);
    };

#ifdef SNR_TIMING
    pipe.t_phase[0] += SNR_T() - tp0;
#endif
    // stage 0: PE -> hA
    stage8(IPE{}, I0{}, IPE{}, IPE{}, &pe[0][0], &pe[0][0], &hA[0][0], bias_off_stage(0), [&](int nt) {
      if constexpr (TRAIN) {
        // (the saved encodings are bf16 whatever the forward multiplied with: the weight-gradient pass pairs them with d z)
        act_store_cvt(std::bool_constant<E16>{}, PALL{}, AL.k_pe(), IPE{}, &pe[0][0], IPE{}, nt, I8{});
        if constexpr (VD) act_store_cvt(std::bool_constant<E16>{}, PALL{}, AL.k_dir(), IDIR{}, &dir[0][0], IDIRA{}, nt, I8{});
      }
    }, 0);
    // stages 1..7 ping-pong hA/hB; stage 5 prepends the encoding (skip connection, helpers:110-111)
    u32x4 pmask[NJ];
    auto keep_masks = [&]() {
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) pmask[jt] = mask[jt];
    };
    auto pre_of = [&](int s_prev, const Frag* src) {
      return [&, s_prev, src](int nt) {
        if constexpr (TRAIN) {
          if (a.save_even || (s_prev & 1))
            act_store_par(PODD{}, AL.k_h(s_prev), IH{}, src, IH{}, nt, I8{});
          if (nt == 0) mask_store(AL.k_mask(s_prev), pmask);
        }
      };
    };
    for (int it = 0; it < 2; ++it) {  // stages (1,2), (3,4)
      const int s1 = 1 + 2 * it;
      keep_masks();
      stage8(I0{}, IH{}, IH{}, IH{}, &hA[0][0], &hA[0][0], &hB[0][0], bias_off_stage(s1), pre_of(s1 - 1, &hA[0][0]), s1);
      keep_masks();
      stage8(I0{}, IH{}, IH{}, IH{}, &hB[0][0], &hB[0][0], &hA[0][0], bias_off_stage(s1 + 1), pre_of(s1, &hB[0][0]), s1 + 1);
    }
    keep_masks();
#ifdef SNR_TIMING
    const unsigned long long tp1 = SNR_T();
#endif
    if (!kKeepEnc) make_pe(true);
#ifdef SNR_TIMING
    pipe.t_phase[1] += SNR_T() - tp1;
#endif
    stage8(IPE{}, IH{}, IPE{}, IH{}, &pe[0][0], &hA[0][0], &hB[0][0], bias_off_stage(5), pre_of(4, &hA[0][0]), 5);
    keep_masks();
    stage8(I0{}, IH{}, IH{}, IH{}, &hB[0][0], &hB[0][0], &hA[0][0], bias_off_stage(6), pre_of(5, &hB[0][0]), 6);
    keep_masks();
    if (staged && wg + gridDim.x < n_wg) prefetch_inputs(wg + gridDim.x);   // the next pass's inputs: a third of a pass ahead
    stage8(I0{}, IH{}, IH{}, IH{}, &hA[0][0], &hA[0][0], &hB[0][0], bias_off_stage(7), pre_of(6, &hA[0][0]), 7);
    keep_masks();   // masks of stage 7
    Frag* cur = &hB[0][0];   // h7
    Frag* nxt = &hA[0][0];
    if constexpr (VD) {
      // stage 8: feature (8 tiles, no relu) + the alpha tile (row 0 = alpha_linear)
      float alpha[NJ];
      pipe.template run_tiles<0, KS_H, 9, NJ, KS_H, KS_H>(
          cur, cur, [&](int nt) { return bias_tile_addr(bias_lds, kBiasFeat + 32 * nt, g); },
          [&](int nt, int jt, f32x16 acc) {
            if (nt < 8) tile_out(std::false_type{}, nt, jt, acc, nxt + jt * KS_H + nt * FPT);
            else alpha[jt] = acc[0];  // row 0 lives in register 0 of lanes 0..31
          },
          [&](int nt) {
            if constexpr (TRAIN) {
              act_store_par(PODD{}, AL.k_h(7), IH{}, cur, IH{}, nt, I9{});
              if (nt == 0) mask_store(AL.k_mask(7), pmask);
            }
          });
      // stage 9: views = relu(W [feat | dir] + b), 4 tiles -> h9 (in `cur` storage)
      Frag* feat = nxt;
      Frag* h9 = cur;
      clear_masks();
#ifdef SNR_TIMING
      const unsigned long long tp2 = SNR_T();
#endif
      if (!kKeepEnc) make_dir(true);
#ifdef SNR_TIMING
      pipe.t_phase[2] += SNR_T() - tp2;
#endif
      pipe.template run_tiles<KS_H, KS_DIR, 4, NJ, KS_H, KS_DIRA, false, E16>(
          feat, &dir[0][0], [&](int nt) { return bias_tile_addr(bias_lds, kBiasViews + 32 * nt, g); },
          [&](int nt, int jt, f32x16 acc) {
            tile_out(std::true_type{}, nt, jt, acc, h9 + jt * KS_H + nt * FPT);
            if constexpr (TRAIN && kSplit) {   // both fragments of the h9 tile leave with the tile (stage 10 is 8 MFMAs long)
              if (jt == NJ - 1) act_store(AL.k_h9(), IH9{}, h9, IH{}, nt, I4{});
            }
          },
          [&](int) {});   // feat is not saved (mlp_pack.h: ActLayout)
      // stage 10: rgb
      keep_masks();
      f32x16 acc_c[NJ];
      pipe.template run_tiles<0, KS_H9, 1, NJ, KS_H, KS_H>(
          h9, h9, [&](int) { return bias_tile_addr(bias_lds, kBiasRgb, g); },
          [&](int, int jt, f32x16 acc) { acc_c[jt] = acc; },
          [&](int) {
            if constexpr (TRAIN) { if constexpr (!kSplit) act_store(AL.k_h9(), IH9{}, h9, IH{}, 0, I1{}); mask_store(AL.k_mask9(), pmask); }
          });
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) {
        // the sample index is re-derived here (an opaque copy of the lane's sample number): kept from the top of the pass it
        // was a 64-bit value live across all eleven stages, and since round 5 (stores behind the epilogues) it went to scratch
        int sj2 = sj;
        asm volatile("" : "+v"(sj2));
        const int64_t mm = (tile0 + jt) * 32 + sj2;
        if (mm < a.n_samples && g == 0) *(f32x4*)(a.raw + 4 * mm) = f32x4{acc_c[jt][0], acc_c[jt][1], acc_c[jt][2], alpha[jt]};
      }
    } else {
      f32x16 acc_o[NJ];
      pipe.template run_tiles<0, KS_H, 1, NJ, KS_H, KS_H>(
          cur, cur, [&](int) { return bias_tile_addr(bias_lds, kBiasOut, g); },
          [&](int, int jt, f32x16 acc) { acc_o[jt] = acc; },
          [&](int) {
            if constexpr (TRAIN) { act_store_par(PODD{}, AL.k_h(7), IH{}, cur, IH{}, 0, I1{}); mask_store(AL.k_mask(7), pmask); }
          });
      // rows 0..3 = registers 0..3 of lane half 0; row 4 = register 0 of lane half 1
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) {
        if (valid[jt]) {
          float* o = a.raw + (int64_t)a.out_ch * m[jt];
          if (g == 0) { o[0] = acc_o[jt][0]; o[1] = acc_o[jt][1]; o[2] = acc_o[jt][2]; o[3] = acc_o[jt][3]; }
          else if (a.out_ch == 5) o[4] = acc_o[jt][0];
        }
      }
    }
#if SNR_ABLATE & 64
    pipe.dma_on = false;
#endif
  }
  pipe.drain();  // prefetched blocks still in flight must land before the LDS allocation is released
}


}  // namespace snr

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace snr;

static int check_cfg(const snr_mlp_config* c) {
  if (!c) return SNR_ERR_NULL;
  if (c->precision != SNR_PREC_BF16 && c->precision != SNR_PREC_FP32) return SNR_ERR_UNSUPPORTED;
  if (c->i_embed != 0 && c->i_embed != -1) return SNR_ERR_UNSUPPORTED;
  if (c->multires < 0 || c->multires > kMaxMultires) return SNR_ERR_UNSUPPORTED;
  if (c->multires_views < (c->use_viewdirs ? 0 : -1) || c->multires_views > kMaxMultiresViews)
    return SNR_ERR_UNSUPPORTED;
  if (c->use_viewdirs) { if (c->out_ch != 4) return SNR_ERR_UNSUPPORTED; }
  else if (c->out_ch != 4 && c->out_ch != 5) return SNR_ERR_UNSUPPORTED;
  return SNR_OK;
}

template <int P> static PackTable table_of(const snr_mlp_config* c) {
  return make_pack_table<P>(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1);
}

extern "C" int64_t snr_mlp_param_count(const snr_mlp_config* c) {
  if (check_cfg(c) != SNR_OK) return check_cfg(c);
  return make_param_layout(c->multires, c->multires_views, c->use_viewdirs, c->out_ch, c->i_embed == -1).total;
}

extern "C" int64_t snr_mlp_packed_bytes(const snr_mlp_config* c) {
  if (check_cfg(c) != SNR_OK) return check_cfg(c);
  const PackTable T = c->precision == SNR_PREC_BF16 ? table_of<kBF16>(c) : table_of<kFP32>(c);
  return (int64_t)(T.fwd_frags + T.bwd_frags) * 1024 + (int64_t)T.bias_floats * 4;
}

extern "C" int64_t snr_mlp_act_bytes(const snr_mlp_config* c, int64_t n) {
  if (check_cfg(c) != SNR_OK) return check_cfg(c);
  if (n <= 0) return SNR_ERR_SHAPE;
  return c->precision == SNR_PREC_BF16 ? ActLayout<kBF16>(n, c->use_viewdirs).bytes()
                                       : ActLayout<kFP32>(n, c->use_viewdirs).bytes();
}

extern "C" int snr_mlp_pack(const snr_mlp_config* c, const float* params, void* packed, snr_stream_t stream) {
  int st = check_cfg(c);
  if (st != SNR_OK) return st;
  SNR_CHECK_ARG(params && packed, SNR_ERR_NULL);
  hipStream_t s = (hipStream_t)stream;
  if (c->precision == SNR_PREC_BF16) {
    const PackTable T = table_of<kBF16>(c);
    const int64_t threads = (int64_t)(T.fwd_frags + T.bwd_frags) * 64;
    ProfScope ps(K_MLP_PACK, s);
    mlp_pack_kernel<kBF16><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(T, params, (char*)packed);
  } else {
    const PackTable T = table_of<kFP32>(c);
    const int64_t threads = (int64_t)(T.fwd_frags + T.bwd_frags) * 64;
    ProfScope ps(K_MLP_PACK, s);
    mlp_pack_kernel<kFP32><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(T, params, (char*)packed);
  }
  return launch_status();
}

template <int P, bool VD, bool TRAIN>
static int launch_fwd(const FwdArgs& a, hipStream_t s) {
  constexpr int per_wg = ChainCfg<P, TRAIN>::WAVES * ChainCfg<P, TRAIN>::NJ;
  const int64_t n_wg = (padded_tiles<P>(a.n_samples) + per_wg - 1) / per_wg;
  const int lds = kBiasLdsBytes + kRingBytes + per_wg * 5 * 256;   // bias block | weight ring | the waves' input staging areas
  if (int e = ensure_dynamic_lds<&mlp_fwd_kernel<P, VD, TRAIN>>(lds)) return e;
  // one workgroup per CU is resident; since the inputs of a pass are fetched a pass ahead (round 5) the grid is persistent:
  // a workgroup that loops keeps its bias block and its weight ring and finds its next inputs in LDS
  const int64_t cap = tunables().chain_grid > 0 ? tunables().chain_grid : (SNR_IN_PREFETCH ? cu_count() : 1024);
  const int64_t grid = n_wg < cap ? n_wg : cap;
  {
    ProfScope ps(K_MLP_FWD, s);
    mlp_fwd_kernel<P, VD, TRAIN><<<dim3((unsigned)grid), dim3(64 * ChainCfg<P, TRAIN>::WAVES), lds, s>>>(a);
  }
  return launch_status();
}

extern "C" int snr_mlp_forward(const snr_mlp_config* c, const void* packed, const float* pts, const float* rays,
                               int ray_ld, const float* z_vals, const float* viewdirs, int viewdirs_ld,
                               int64_t n_samples, int samples_per_ray, float* raw, void* act,
                               snr_stream_t stream) {
  int st = check_cfg(c);
  if (st != SNR_OK) return st;
  SNR_CHECK_ARG(packed && raw, SNR_ERR_NULL);
  SNR_CHECK_ARG(pts || (rays && z_vals), SNR_ERR_NULL);
  SNR_CHECK_ARG(!c->use_viewdirs || (viewdirs && viewdirs_ld >= 3), SNR_ERR_NULL);
  SNR_CHECK_ARG(n_samples > 0 && samples_per_ray > 0, SNR_ERR_SHAPE);
  SNR_CHECK_ARG(pts || ray_ld >= 6, SNR_ERR_SHAPE);
  const bool bf = c->precision == SNR_PREC_BF16;
  const PackTable T = bf ? table_of<kBF16>(c) : table_of<kFP32>(c);
  FwdArgs a{};
  a.blob = (const char*)packed;
  a.fwd_blocks = T.fwd_frags / kBlockFrags;
  a.bias = (const float*)((const char*)packed + (int64_t)(T.fwd_frags + T.bwd_frags) * 1024);
  a.bias_floats = T.bias_floats;
  a.pts = pts; a.rays = rays; a.ray_ld = ray_ld; a.z_vals = z_vals; a.viewdirs = viewdirs; a.vd_ld = viewdirs_ld;
  a.n_samples = n_samples; a.S = samples_per_ray;
  a.multires = T.multires; a.multires_views = T.multires_views; a.out_ch = c->out_ch;
  a.raw = raw; a.act = (char*)act;
  a.save_even = !(bf && recompute_enabled());
  a.enc_generic = tunables().enc_generic;
  hipStream_t s = (hipStream_t)stream;
  const bool vd = c->use_viewdirs, tr = act != nullptr;
#define SNR_FWD(P_) \
  (vd ? (tr ? launch_fwd<P_, true, true>(a, s) : launch_fwd<P_, true, false>(a, s)) \
      : (tr ? launch_fwd<P_, false, true>(a, s) : launch_fwd<P_, false, false>(a, s)))
  return bf ? SNR_FWD(kBF16) : SNR_FWD(kFP32);
#undef SNR_FWD
}

#ifdef SNR_TIMING
extern "C" int snr_debug_read_fwd(unsigned long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(snr::g_snr_dbg), 8 * sizeof(unsigned long long));
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(snr::g_snr_dbg), z, sizeof(z));
  return (int)e;
}
#endif
