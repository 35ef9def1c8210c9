// Backward of the fused field network for gfx950 (MI355X): what autograd runs for
// networks/dm_nerf.py:80-106 in the reference's training step (train_dmsr.py:32-64), as three kernels.
//
//   field_bwd_data_kernel   dL/d(pre-activation) of every layer, chained in registers exactly like the forward
//                           (transposed weights as the MFMA A operand, the previous gradient tile as B), ReLU
//                           masks from the forward's save record, every layer's gradient tile written once as a
//                           register image (field_layout.h "Training").  The instance branch stops at the trunk
//                           output (h.detach(), dm_nerf.py:95).
//   field_wgrad_kernel      dW = sum over points of G (x) X for the 17 (layer, input block) pairs, as split-K MFMA
//                           GEMMs over the saved images: rows of 32 points are transposed through LDS into the
//                           operand layout, each workgroup owns the whole [O x K] block for a slice of the points
//                           and writes one partial; bias gradients (row sums of G) fall out of the staging loads.
//   field_wgrad_reduce_kernel  sums the partials and scatters them into the flat parameter-gradient vector
//                           (state_dict order) through the host-built table (field_pack.cpp).
//
// All arithmetic is fp32 (v_mfma_f32_32x32x2_f32).
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/dmnerf_hip.h"
#include "field_common.h"
#include "field_stream.h"

using namespace dmn;

namespace {

struct BwdArgs {
  const float* g_raw;    // [M, C]  dL/d raw
  const float* save;     // forward save record
  const float* blob;     // backward-data blob
  float* grad;           // gradient record (out)
  long M;
  int ins_ch;            // ins_num + 1
};

#define SRC_GI(i) gi[(i) < 16 * T ? (i) : 0]

__device__ __forceinline__ bool mask_bit(const unsigned* m, int i) { return (m[i >> 5] >> (i & 31)) & 1u; }

template <int T>
__global__ __launch_bounds__(256, 1) void field_bwd_data_kernel(const BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* const cst = lds + 2 * kStageFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* const stream = a.blob + kConstFloats;
  constexpr int G = bwd_total_stages(T);

  for (int c = wave_s; c < kConstChunks; c += 4)
    glds16(a.blob + c * kChunkFloats + lane * 4, cst + c * kChunkFloats);
  load_stage(stream, 0, lds, wave_s, lane);

  const long tile = (long)blockIdx.x * 4 + wave_s;
  const long m = tile * 32 + p;
  const bool valid = m < a.M;
  const float* const srec = a.save + tile * kSaveTileFloats;
  float* const grec = a.grad + tile * grad_tile_floats(T);

  // ---- dL/d raw of this lane's point: rgb(3), sigma, instance logits in the ins_linear tile layout
  float gi[16 * T];
  float g_rgb[3] = {0.f, 0.f, 0.f}, g_sig = 0.f;
  {
    const int C = 4 + a.ins_ch;
    const float* row = a.g_raw + (valid ? m : 0) * C;
    if (valid) {
      g_rgb[0] = row[0]; g_rgb[1] = row[1]; g_rgb[2] = row[2]; g_sig = row[3];
    }
#pragma unroll
    for (int i = 0; i < 16 * T; ++i) {
      const int ch = feat_of(i, 0) + 4 * h;
      gi[i] = (valid && ch < a.ins_ch) ? row[4 + (ch < a.ins_ch ? ch : 0)] : 0.f;
      image_store(grec, kGradHead + i, lane, gi[i]);
    }
    image_store(grec, kGradHead + 16 * T + 0, lane, h ? g_rgb[1] : g_rgb[0]);
    image_store(grec, kGradHead + 16 * T + 1, lane, h ? g_sig : g_rgb[2]);
#pragma unroll
    for (int i = 2; i < 16; ++i) image_store(grec, kGradHead + 16 * T + i, lane, 0.f);
  }

  f32x16 acc[8];
  f32x16 acc2[4];
  float act[128];
  unsigned msk[4];
  int g = 0;
  const float* buf;

  // ---- instance branch: ins_linear^T, ReLU mask of ins_feature_linears.0, then its transpose
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
  for (int st = 0; st < bwd_inso_stages(T); ++st) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    constexpr int kAll = 16 * T;
    const int nk = kAll - 32 * st < 32 ? kAll - 32 * st : 32;
    if (nk == 32) { DMN_STAGE4(buf, SRC_GI, 32 * st, 32) }
    else if (nk == 16) { DMN_STAGE4(buf, SRC_GI, 32 * st, 16) }
    ++g;
  }
  msk[0] = __float_as_uint(image_load(srec, kSaveMaskI1 + 0, lane));
  msk[1] = __float_as_uint(image_load(srec, kSaveMaskI1 + 1, lane));
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const float v = mask_bit(msk, i) ? acc2[i >> 4][i & 15] : 0.f;
    acc2[i >> 4][i & 15] = v;
    image_store(grec, kGradI1 + i, lane, v);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE8(buf, SRC_ACC2, 16 * s)
    ++g;
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) image_store(grec, kGradInsF + i, lane, acc[i >> 4][i & 15]);

  // ---- colour branch: rgb_linear^T on the VALU, ReLU mask of rgb_feature_linears.0, its transpose (feature
  //      columns only: the view encoding needs no gradient), rgb_feature_linear^T
  msk[0] = __float_as_uint(image_load(srec, kSaveMaskR1 + 0, lane));
  msk[1] = __float_as_uint(image_load(srec, kSaveMaskR1 + 1, lane));
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) v = fmaf(g_rgb[c], cst[kConstRgbW + c * 128 + h * 64 + i], v);
    v = mask_bit(msk, i) ? v : 0.f;
    acc2[i >> 4][i & 15] = v;
    image_store(grec, kGradR1 + i, lane, v);
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE8(buf, SRC_ACC2, 16 * s)
    ++g;
  }
#pragma unroll
  for (int i = 0; i < 128; ++i) {
    act[i] = acc[i >> 4][i & 15];
    image_store(grec, kGradFeat + i, lane, act[i]);
  }

  // ---- trunk output: rgb_feature_linear^T + density_linear^T, then mlps.7 .. mlps.1 transposed
#pragma unroll 1
  for (int l = 8; l >= 1; --l) {
#pragma unroll
    for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(image_load(srec + (save_mask(1) + 4 * (l - 1)) * 64, w, lane));
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      buf = stage_begin(stream, g, G, lds, wave_s, lane);
      DMN_STAGE8(buf, SRC_ACT, 16 * s)
      ++g;
    }
    float* const out = grec + (kGradA + 128 * (l - 1)) * 64;
    const float sig = l == 8 ? g_sig : 0.f;          // density_linear reads h_8 (dm_nerf.py:101)
#pragma unroll
    for (int i = 0; i < 128; ++i) {
      float v = acc[i >> 4][i & 15];
      v = fmaf(sig, cst[kConstDensW + h * 128 + i], v);
      v = mask_bit(msk, i) ? v : 0.f;
      act[i] = v;
      image_store(out, i, lane, v);
    }
  }
}

// ================================================================================================
// weight gradients
struct WgArgs {
  const float* save;
  const float* grad;
  float* partial;          // this class's region: [splits][slice_len], element e at e - elem_base
  long n_tiles;            // 32-point tiles
  long grad_tile_floats;
  long slice_len, elem_base;
  long elem_base_small_len;   // slice length of the small-GEMM region
  int splits;
  WGemm gemm[kWGemms];
  int list[kWGemms];       // blockIdx.y -> gemm id (GEMMs of one shape class per launch)
};

// OT x KT tiles of 32 x 32; the four waves form a WR x WC grid, each owning (OT/WR) x (KT/WC) tiles
template <int OT, int KT, int WR, int WC>
__global__ __launch_bounds__(256, 1) void field_wgrad_kernel(const WgArgs a) {
  constexpr int R = 32 * (OT + KT);        // rows staged per point tile: G rows, then X rows
  constexpr int RS = R + 2;                // LDS row stride of the [point][row] tile (2 mod 16: conflict-free)
  constexpr int NP = OT + KT;              // 16-byte pieces per lane per tile (8 pieces per 128-byte row)
  constexpr int OTW = OT / WR, KTW = KT / WC;
  static_assert(WR * WC == 4 && OTW * WR == OT && KTW * WC == KT && OTW * KTW <= 16, "wave grid");
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * RS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, l = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const WGemm gm = a.gemm[a.list[blockIdx.y]];
  const long per = (a.n_tiles + a.splits - 1) / a.splits;
  const long t0 = (long)blockIdx.x * per;
  const long t1 = t0 + per < a.n_tiles ? t0 + per : a.n_tiles;
  float* const part = a.partial + (long)blockIdx.x * a.slice_len - a.elem_base;

  f32x16 acc[OTW * KTW];
#pragma unroll
  for (int n = 0; n < OTW * KTW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float bsum[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) bsum[i] = 0.f;

  // piece q = it*256 + tid: row q>>3 (G rows first), points 4*(q&7)..+3
  const int prow = tid >> 3, ppc = tid & 7;
  f32x4 stg[NP];
  auto fetch = [&](long tile) {
    const float* gsrc = a.grad + tile * a.grad_tile_floats + gm.g_reg * 64 + tid * 4;
    const float* xsrc = a.save + tile * kSaveTileFloats + gm.x_reg * 64 + tid * 4;
#pragma unroll
    for (int it = 0; it < OT; ++it) stg[it] = record_load4(gsrc + it * 1024);
#pragma unroll
    for (int it = 0; it < KT; ++it) stg[OT + it] = record_load4(xsrc + it * 1024);
  };
  auto stage = [&](float* buf) {
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      float* d = buf + (4 * ppc) * RS + it * 32 + prow;
      d[0 * RS] = stg[it].x;
      d[1 * RS] = stg[it].y;
      d[2 * RS] = stg[it].z;
      d[3 * RS] = stg[it].w;
      if (it < OT) bsum[it < OT ? it : 0] += (stg[it].x + stg[it].y) + (stg[it].z + stg[it].w);
    }
  };

  if (t0 < t1) {
    fetch(t0);
    stage(lds);
  }
  __syncthreads();
  for (long t = t0; t < t1; ++t) {
    const float* buf = lds + ((t - t0) & 1) * (32 * RS);
    if (t + 1 < t1) fetch(t + 1);
    // k-step ks: lane-half kh multiplies point ks + 16*kh (any bijection works: both operands use the same one)
    const float* pa = buf + (16 * kh) * RS + wr * OTW * 32 + l;
    const float* pb = buf + (16 * kh) * RS + OT * 32 + wc * KTW * 32 + l;
    float fa[OTW], fb[KTW];
#pragma unroll
    for (int i = 0; i < OTW; ++i) fa[i] = pa[i * 32];
#pragma unroll
    for (int j = 0; j < KTW; ++j) fb[j] = pb[j * 32];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      float na[OTW], nb[KTW];
#pragma unroll
      for (int i = 0; i < OTW; ++i) na[i] = ks + 1 < 16 ? pa[(ks + 1) * RS + i * 32] : 0.f;
#pragma unroll
      for (int j = 0; j < KTW; ++j) nb[j] = ks + 1 < 16 ? pb[(ks + 1) * RS + j * 32] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < OTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i * KTW + j] = DMN_MFMA(fa[i], fb[j], acc[i * KTW + j]);
#pragma unroll
      for (int i = 0; i < OTW; ++i) fa[i] = na[i];
#pragma unroll
      for (int j = 0; j < KTW; ++j) fb[j] = nb[j];
    }
    if (t + 1 < t1) stage(lds + ((t + 1 - t0) & 1) * (32 * RS));
    __syncthreads();
  }

  // ---- partial [O][K] block (MFMA C layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
  constexpr int K = KT * 32;
#pragma unroll
  for (int i = 0; i < OTW; ++i)
#pragma unroll
    for (int j = 0; j < KTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wr * OTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int k = (wc * KTW + j) * 32 + l;
        part[gm.index_off + o * K + k] = acc[i * KTW + j][r];
      }
  if (gm.bias_off >= 0) {
#pragma unroll
    for (int it = 0; it < OT; ++it) {
      float v = bsum[it];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      if (ppc == 0) part[gm.bias_off + it * 32 + prow] = v;
    }
  }
}

// Same GEMMs on v_mfma_f32_32x32x16_bf16 with the split product G*X ~= Gh*Xh + Gh*Xl + Gl*Xh (bf16 keeps the fp32
// exponent range, which gradients need; 2^-16 relative per product).  The [row][32 points] images are already in
// operand order for this instruction (8 consecutive points per lane), so staging is conversion only: no transpose.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// X6: three planes (hi, mid, lo: exact three-way splits) and six products, as the ring kernel below.
template <int OT, int KT, int WR, int WC, bool X6>
__global__ __launch_bounds__(256, 1) void field_wgrad_x3_kernel(const WgArgs a) {
  constexpr int R = 32 * (OT + KT);
  constexpr int RB = 80;                   // bytes per LDS row: 32 bf16 + 16 pad (16-byte reads of 16 lanes hit 64 banks)
  constexpr int NP = OT + KT;
  constexpr int OTW = OT / WR, KTW = KT / WC;
  static_assert(WR * WC == 4 && OTW * WR == OT && KTW * WC == KT && OTW * KTW <= 16, "wave grid");
  constexpr int NPL = X6 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char lds[NPL * R * RB];   // hi plane, lo plane (X6: hi, mid, lo)
  const int tid = threadIdx.x;
  const int lane = tid & 63, l = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const WGemm gm = a.gemm[a.list[blockIdx.y]];
  const long per = (a.n_tiles + a.splits - 1) / a.splits;
  const long t0 = (long)blockIdx.x * per;
  const long t1 = t0 + per < a.n_tiles ? t0 + per : a.n_tiles;
  float* const part = a.partial + (long)blockIdx.x * a.slice_len - a.elem_base;

  f32x16 acc[OTW * KTW];
#pragma unroll
  for (int n = 0; n < OTW * KTW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float bsum[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) bsum[i] = 0.f;

  const int prow = tid >> 3, ppc = tid & 7;
  f32x4 stg[NP];
  auto fetch = [&](long tile) {
    const float* gsrc = a.grad + tile * a.grad_tile_floats + gm.g_reg * 64 + tid * 4;
    const float* xsrc = a.save + tile * kSaveTileFloats + gm.x_reg * 64 + tid * 4;
#pragma unroll
    for (int it = 0; it < OT; ++it) stg[it] = record_load4(gsrc + it * 1024);
#pragma unroll
    for (int it = 0; it < KT; ++it) stg[OT + it] = record_load4(xsrc + it * 1024);
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const f32x4 v = stg[it];
      // packed conversions, and the rounded values back out of the packed words (shift / mask): scalar casts make
      // the compiler convert every value twice
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      auto back = [](const bf16x4& q) {
        const u32x2 w = __builtin_bit_cast(u32x2, q);
        return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                     __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
      };
      const bf16x4 hi = __builtin_convertvector(v, bf16x4);
      const f32x4 r1 = v - back(hi);
      char* d = lds + (it * 32 + prow) * RB + ppc * 8;
      *reinterpret_cast<bf16x4*>(d) = hi;
      if constexpr (X6) {
        const bf16x4 mid = __builtin_convertvector(r1, bf16x4);
        *reinterpret_cast<bf16x4*>(d + R * RB) = mid;
        *reinterpret_cast<bf16x4*>(d + 2 * R * RB) = __builtin_convertvector(r1 - back(mid), bf16x4);
      } else {
        *reinterpret_cast<bf16x4*>(d + R * RB) = __builtin_convertvector(r1, bf16x4);
      }
      if (it < OT) bsum[it < OT ? it : 0] += (v.x + v.y) + (v.z + v.w);
    }
  };

  // timing-only ablations (never in the product build): DMN_WG_ABL 1 = convert / stage only the first tile,
  // 2 = fetch only the first tile, 3 = both
#ifndef DMN_WG_ABL
#define DMN_WG_ABL 0
#endif
  if (t0 < t1) fetch(t0);
  for (long t = t0; t < t1; ++t) {
    __syncthreads();                       // everybody is done reading the previous tile
    if (!((DMN_WG_ABL & 1) && t > t0)) stage();
    __syncthreads();
    if (t + 1 < t1 && !(DMN_WG_ABL & 2)) fetch(t + 1);          // in flight under this tile's MFMAs
    const char* pa = lds + (wr * OTW * 32 + l) * RB + kh * 16;
    const char* pb = lds + (OT * 32 + wc * KTW * 32 + l) * RB + kh * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // planes: [0] hi, [1] lo (x3) / mid (x6), [2] lo (x6)
      bf16x8 fa[NPL][OTW], fb[NPL][KTW];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < OTW; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * RB + ks * 32 + pl * R * RB);
#pragma unroll
        for (int j = 0; j < KTW; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(pb + j * 32 * RB + ks * 32 + pl * R * RB);
      }
#pragma unroll
      for (int i = 0; i < OTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
          f32x16 c = acc[i * KTW + j];
          if constexpr (X6) {              // small terms first (field_x6.h)
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], c, 0, 0, 0);
          acc[i * KTW + j] = c;
        }
    }
  }

  constexpr int K = KT * 32;
#pragma unroll
  for (int i = 0; i < OTW; ++i)
#pragma unroll
    for (int j = 0; j < KTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wr * OTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int k = (wc * KTW + j) * 32 + l;
        part[gm.index_off + o * K + k] = acc[i * KTW + j][r];
      }
  if (gm.bias_off >= 0) {
#pragma unroll
    for (int it = 0; it < OT; ++it) {
      float v = bsum[it];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 4);
      if (ppc == 0) part[gm.bias_off + it * 32 + prow] = v;
    }
  }
}

// The nine 256 x 256 GEMMs again, fed by an LDS-DMA ring instead of staging registers and LDS planes.
// field_wgrad_x3_kernel has one 64 KB tile in flight per CU, and only while it multiplies (the registers of tile t+1
// are free for the fetch only after tile t is converted): 4.2 us per tile of which 2.1 us are waiting for HBM; and its
// conversion pass (registers -> bf16 hi / lo planes in LDS, two barriers) runs with the matrix pipe idle.  Here the
// 32-point tile is cut into two 16-point halves (one k-step each); a half arrives as raw fp32 rows (64 bytes per row) in
// one of four 32 KB ring slots by global_load_lds, three halves are in flight while one is multiplied, and every wave
// converts its own operand fragments from the raw rows (two 16-byte LDS reads per fragment, the same number the planes
// took) in the shadow of its MFMAs (hand-placed: see the loop): no staging registers, no planes, ONE barrier per half.  The G and X fragments a
// wave needs are converted by two waves each; that VALU work hides behind the 48 MFMAs of a half.
// Bank conflicts: rows of 64 bytes would put rows l, l+4, l+8, l+12 of a 16-byte read on the same banks; the ring holds
// piece c of row r at position c ^ ((r >> 2) & 3) instead (the DMA lane fetches the piece that belongs in ITS slot).
// Past the end of the slice the ring keeps fetching the last tile, so that the wait count is the same in every iteration.
// X6: every product from exact three-way bf16 splits of both operands (six MFMA products: the forward's bf16x6
// arithmetic, field_x6.h) instead of the two-way splits' three -- fp32-grade weight gradients.
template <bool X6>
__global__ __launch_bounds__(256, 1) void field_wgrad_ring_kernel(const WgArgs a) {
  constexpr int ROWS = 512;                 // 256 G rows, then 256 X rows
  constexpr int HB = ROWS * 64;             // bytes of a half: 16 points x 4 B per row
  constexpr int RING = 4;
  constexpr int OTW = 4, KTW = 4;           // 2 x 2 waves
  __shared__ __attribute__((aligned(16))) char lds[RING * HB];      // 128 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63, l = lane & 31, kh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const WGemm gm = a.gemm[a.list[blockIdx.y]];
  const long per = (a.n_tiles + a.splits - 1) / a.splits;
  const long t0 = (long)blockIdx.x * per;
  const long t1 = t0 + per < a.n_tiles ? t0 + per : a.n_tiles;
  float* const part = a.partial + (long)blockIdx.x * a.slice_len - a.elem_base;

  f32x16 acc[OTW * KTW];
#pragma unroll
  for (int n = 0; n < OTW * KTW; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float bsum[OTW] = {0.f, 0.f, 0.f, 0.f};

  // DMA instruction n (0..7) of this wave fills ring positions n*256 + wave*64 + lane: row n*64 + wave*16 + lane/4
  // (n < 4: G rows, else X rows), position lane%4 of the row = piece (lane%4) ^ ((row >> 2) & 3), and (row >> 2) & 3
  // = (lane >> 4) & 3 for every n and wave
  const unsigned lane_bytes = (unsigned)(lane >> 2) * 128u + (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) * 16u;
  auto issue = [&](long hidx) {
    long tile = t0 + (hidx >> 1);
    tile = tile < t1 ? tile : t1 - 1;
    const int hh = (int)(hidx & 1);
    const char* g = reinterpret_cast<const char*>(a.grad + tile * a.grad_tile_floats + gm.g_reg * 64) + wave * 16 * 128 + hh * 64;
    const char* x = reinterpret_cast<const char*>(a.save + tile * kSaveTileFloats + gm.x_reg * 64) + wave * 16 * 128 + hh * 64;
    char* dst = lds + (int)(hidx & (RING - 1)) * HB + wave * 1024;
#pragma unroll
    for (int n = 0; n < 4; ++n) glds16_u(g + n * 8192, lane_bytes, dst + n * 4096);
#pragma unroll
    for (int n = 0; n < 4; ++n) glds16_u(x + n * 8192, lane_bytes, dst + (4 + n) * 4096);
  };
  // operand fragment of one row: points 8*kh .. 8*kh+7 of the half = pieces 2*kh, 2*kh+1, split into bf16 hi / lo
  const int sw = (l >> 2) & 3;
  const int off0 = ((2 * kh) ^ sw) * 16, off1 = ((2 * kh + 1) ^ sw) * 16;
  struct Raw { f32x4 v0, v1; };
  auto fetch_row = [&](const char* row) -> Raw {
    return Raw{*reinterpret_cast<const f32x4*>(row + off0), *reinterpret_cast<const f32x4*>(row + off1)};
  };
  struct F { bf16x8 h, m, l; };            // operand fragment as bf16 planes (m: X6 only)
  auto split = [&](const Raw& r, F& f) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto back = [](const bf16x4& q) {
      const u32x2 w = __builtin_bit_cast(u32x2, q);
      return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                   __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
    };
    const bf16x4 h0 = __builtin_convertvector(r.v0, bf16x4), h1 = __builtin_convertvector(r.v1, bf16x4);
    const f32x4 r0 = r.v0 - back(h0), r1 = r.v1 - back(h1);
    f.h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    if constexpr (X6) {
      const bf16x4 m0 = __builtin_convertvector(r0, bf16x4), m1 = __builtin_convertvector(r1, bf16x4);
      const bf16x4 l0 = __builtin_convertvector(r0 - back(m0), bf16x4), l1 = __builtin_convertvector(r1 - back(m1), bf16x4);
      f.m = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
      f.l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
      const bf16x4 l0 = __builtin_convertvector(r0, bf16x4), l1 = __builtin_convertvector(r1, bf16x4);
      f.l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto row_sum = [](const Raw& r) { return ((r.v0.x + r.v0.y) + (r.v0.z + r.v0.w)) + ((r.v1.x + r.v1.y) + (r.v1.z + r.v1.w)); };
  auto triple = [&](int i, int j, const F& A, const F& B) {
    f32x16 c = acc[i * KTW + j];
    if constexpr (X6) {                    // small terms first (field_x6.h)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.m, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.h, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.m, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h, c, 0, 0, 0);
    acc[i * KTW + j] = c;
  };
#define DMN_FENCE() __builtin_amdgcn_sched_barrier(0)

  if (t0 < t1) {
    const long H = 2 * (t1 - t0);
    issue(0);
    issue(1);
    issue(2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // half 0
    __syncthreads();
    if constexpr (X6) {
      // Six products and three-way conversions: a whole fragment's conversion (52 vector instructions) no longer fits into
      // the shadow of the product it follows (6 MFMAs hide ~30), and in that arrangement the kernel ran at half its MFMA
      // rate (6.6 ms per step for the nine big GEMMs).  Here every product carries HALF a conversion (4 of a fragment's 8
      // values: 22 vector instructions + one 16-byte LDS read, fetched one product ahead), and the order of the products
      // gives every fragment two products of lead: carried into a half are A0, A1, B0, B1 (converted under the second
      // half of the half before), products (0,0) (1,0) (0,1) (1,1) convert A2, A3, products (2,0) (3,0) (2,1) (3,1)
      // convert B2, B3, and the last eight convert the next half's A0, A1, B0, B1 behind the ring hand-over.
      F fa[OTW], fb[4], nfa[2];
      auto fetch_half = [&](const char* row, int half) { return *reinterpret_cast<const f32x4*>(row + (half ? off1 : off0)); };
      auto half_sum = [](const f32x4& v) { return (v.x + v.y) + (v.z + v.w); };
      auto split_half = [&](const f32x4& v, int half, F& f) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        auto back = [](const bf16x4& q) {
          const u32x2 w = __builtin_bit_cast(u32x2, q);
          return f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
        };
        const bf16x4 hh = __builtin_convertvector(v, bf16x4);
        const f32x4 r1 = v - back(hh);
        const bf16x4 mm = __builtin_convertvector(r1, bf16x4);
        const bf16x4 ll = __builtin_convertvector(r1 - back(mm), bf16x4);
        if (half == 0) {
          f.h = __builtin_shufflevector(hh, hh, 0, 1, 2, 3, 0, 1, 2, 3);
          f.m = __builtin_shufflevector(mm, mm, 0, 1, 2, 3, 0, 1, 2, 3);
          f.l = __builtin_shufflevector(ll, ll, 0, 1, 2, 3, 0, 1, 2, 3);
        } else {
          const bf16x8 eh = __builtin_shufflevector(hh, hh, 0, 1, 2, 3, 0, 1, 2, 3);
          const bf16x8 em = __builtin_shufflevector(mm, mm, 0, 1, 2, 3, 0, 1, 2, 3);
          const bf16x8 el = __builtin_shufflevector(ll, ll, 0, 1, 2, 3, 0, 1, 2, 3);
          f.h = __builtin_shufflevector(f.h, eh, 0, 1, 2, 3, 8, 9, 10, 11);
          f.m = __builtin_shufflevector(f.m, em, 0, 1, 2, 3, 8, 9, 10, 11);
          f.l = __builtin_shufflevector(f.l, el, 0, 1, 2, 3, 8, 9, 10, 11);
        }
      };
      const int arow = (wr * OTW * 32 + l) * 64, brow = (256 + wc * KTW * 32 + l) * 64;
      {                                                      // half 0: A0, A1, B0, B1 up front
        const char* pa = lds + arow;
        const char* pb = lds + brow;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const f32x4 va = fetch_half(pa + i * 32 * 64, hf), vb = fetch_half(pb + i * 32 * 64, hf);
            split_half(va, hf, fa[i]);
            split_half(vb, hf, fb[i]);
            bsum[i] += half_sum(va);
          }
      }
      for (long h = 0; h < H; ++h) {
        const char* pa = lds + (int)(h & (RING - 1)) * HB + arow;
        const char* pb = lds + (int)(h & (RING - 1)) * HB + brow;
        const char* na = lds + (int)((h + 1) & (RING - 1)) * HB + arow;
        const char* nb = lds + (int)((h + 1) & (RING - 1)) * HB + brow;
        const bool more = h + 1 < H;
        f32x4 r = fetch_half(pa + 2 * 32 * 64, 0), rn;
        // one product: the next conversion's raw values are fetched above the fence, the product's six MFMAs and the
        // conversion of the values fetched one product earlier share the region below it
#define DMN_STEP(I, J, FETCH, CONVERT) \
        rn = FETCH;                      \
        DMN_FENCE();                     \
        triple(I, J, fa[I], fb[J]);      \
        CONVERT;                         \
        r = rn;
        DMN_STEP(0, 0, fetch_half(pa + 2 * 32 * 64, 1), (split_half(r, 0, fa[2]), bsum[2] += half_sum(r)))
        DMN_STEP(1, 0, fetch_half(pa + 3 * 32 * 64, 0), (split_half(r, 1, fa[2]), bsum[2] += half_sum(r)))
        DMN_STEP(0, 1, fetch_half(pa + 3 * 32 * 64, 1), (split_half(r, 0, fa[3]), bsum[3] += half_sum(r)))
        DMN_STEP(1, 1, fetch_half(pb + 2 * 32 * 64, 0), (split_half(r, 1, fa[3]), bsum[3] += half_sum(r)))
        DMN_STEP(2, 0, fetch_half(pb + 2 * 32 * 64, 1), split_half(r, 0, fb[2]))
        DMN_STEP(3, 0, fetch_half(pb + 3 * 32 * 64, 0), split_half(r, 1, fb[2]))
        DMN_STEP(2, 1, fetch_half(pb + 3 * 32 * 64, 1), split_half(r, 0, fb[3]))
        DMN_FENCE();
        triple(3, 1, fa[3], fb[1]);
        split_half(r, 1, fb[3]);
        DMN_FENCE();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // this wave's pieces of half h+1 (half h+2 stays in flight)
        __syncthreads();                                     // ... everybody's; and everybody is past half h-1
        issue(h + 3);                                        // into the slot of half h-1
        r = fetch_half(na, 0);
        DMN_STEP(0, 2, fetch_half(na, 1), (split_half(r, 0, nfa[0]), bsum[0] += more ? half_sum(r) : 0.f))
        DMN_STEP(1, 2, fetch_half(na + 32 * 64, 0), (split_half(r, 1, nfa[0]), bsum[0] += more ? half_sum(r) : 0.f))
        DMN_STEP(2, 2, fetch_half(na + 32 * 64, 1), (split_half(r, 0, nfa[1]), bsum[1] += more ? half_sum(r) : 0.f))
        DMN_STEP(3, 2, fetch_half(nb, 0), (split_half(r, 1, nfa[1]), bsum[1] += more ? half_sum(r) : 0.f))
        DMN_STEP(0, 3, fetch_half(nb, 1), split_half(r, 0, fb[0]))          // B0, B1 of this half are done
        DMN_STEP(1, 3, fetch_half(nb + 32 * 64, 0), split_half(r, 1, fb[0]))
        DMN_STEP(2, 3, fetch_half(nb + 32 * 64, 1), split_half(r, 0, fb[1]))
        DMN_FENCE();
        triple(3, 3, fa[3], fb[3]);
        split_half(r, 1, fb[1]);
        DMN_FENCE();
#undef DMN_STEP
        fa[0] = nfa[0];
        fa[1] = nfa[1];
      }
    } else {
    // A0 / B0 of the half about to be multiplied are converted at the end of the half before (here: up front)
      F fa[OTW], fb[2], nfa, nfb;
      {
        const Raw ra = fetch_row(lds + (wr * OTW * 32 + l) * 64), rb = fetch_row(lds + (256 + wc * KTW * 32 + l) * 64);
        split(ra, nfa);
        split(rb, nfb);
        bsum[0] += row_sum(ra);
      }
      for (long h = 0; h < H; ++h) {
        const char* pa = lds + (int)(h & (RING - 1)) * HB + (wr * OTW * 32 + l) * 64;
        const char* pb = lds + (int)(h & (RING - 1)) * HB + (256 + wc * KTW * 32 + l) * 64;
        const char* na = lds + (int)((h + 1) & (RING - 1)) * HB + (wr * OTW * 32 + l) * 64;
        const char* nb = lds + (int)((h + 1) & (RING - 1)) * HB + (256 + wc * KTW * 32 + l) * 64;
        fa[0] = nfa; fb[0] = nfb;
        // the schedule: every MFMA triple carries the conversion of a fragment needed later, raw rows are read one step
        // before they are converted; the ring hand-over (wait, barrier, re-issue) sits in the middle of the half
        Raw r = fetch_row(pa + 1 * 32 * 64);
        DMN_FENCE();
        triple(0, 0, fa[0], fb[0]);
        split(r, fa[1]); bsum[1] += row_sum(r);
        r = fetch_row(pa + 2 * 32 * 64);
        DMN_FENCE();
        triple(1, 0, fa[1], fb[0]);
        split(r, fa[2]); bsum[2] += row_sum(r);
        r = fetch_row(pa + 3 * 32 * 64);
        DMN_FENCE();
        triple(2, 0, fa[2], fb[0]);
        split(r, fa[3]); bsum[3] += row_sum(r);
        r = fetch_row(pb + 1 * 32 * 64);
        DMN_FENCE();
        triple(3, 0, fa[3], fb[0]);
        split(r, fb[1]);
        r = fetch_row(pb + 2 * 32 * 64);
        DMN_FENCE();
        triple(0, 1, fa[0], fb[1]);
        split(r, fb[0]);                               // B2 (B0 is done)
        r = fetch_row(pb + 3 * 32 * 64);
        DMN_FENCE();
        triple(1, 1, fa[1], fb[1]);
        DMN_FENCE();
        triple(2, 1, fa[2], fb[1]);
        DMN_FENCE();
        triple(3, 1, fa[3], fb[1]);
        DMN_FENCE();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // this wave's pieces of half h+1 (half h+2 stays in flight)
        __syncthreads();                                     // ... everybody's; and everybody is past half h-1
        issue(h + 3);                                        // into the slot of half h-1
        DMN_FENCE();
        triple(0, 2, fa[0], fb[0]);
        split(r, fb[1]);                               // B3 (B1 is done)
        Raw rn = fetch_row(na);
        DMN_FENCE();
        triple(1, 2, fa[1], fb[0]);
        split(rn, nfa); bsum[0] += h + 1 < H ? row_sum(rn) : 0.f;
        rn = fetch_row(nb);
        DMN_FENCE();
        triple(2, 2, fa[2], fb[0]);
        split(rn, nfb);
        DMN_FENCE();
        triple(3, 2, fa[3], fb[0]);
        DMN_FENCE();
        triple(0, 3, fa[0], fb[1]);
        DMN_FENCE();
        triple(1, 3, fa[1], fb[1]);
        DMN_FENCE();
        triple(2, 3, fa[2], fb[1]);
        DMN_FENCE();
        triple(3, 3, fa[3], fb[1]);
        DMN_FENCE();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring's last (unused) fetches must land before the LDS goes
  }
#undef DMN_FENCE

  constexpr int K = 256;
#pragma unroll
  for (int i = 0; i < OTW; ++i)
#pragma unroll
    for (int j = 0; j < KTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (wr * OTW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int k = (wc * KTW + j) * 32 + l;
        part[gm.index_off + o * K + k] = acc[i * KTW + j][r];
      }
  if (gm.bias_off >= 0 && wc == 0) {                       // both waves of a row block hold the same row sums
#pragma unroll
    for (int i = 0; i < OTW; ++i) {
      const float v = bsum[i] + __shfl_xor(bsum[i], 32);   // the two 8-point groups of a half
      if (kh == 0) part[gm.bias_off + (wr * OTW + i) * 32 + l] = v;
    }
  }
}

__global__ void field_wgrad_reduce_kernel(const float* partial, const int32_t* index, long index_len, int splits,
                                          float* grads, int accumulate) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= index_len) return;
  const int dst = index[e];
  if (dst < 0) return;
  const bool big = e < kWgradBigLen;
  const long len = big ? kWgradBigLen : index_len - kWgradBigLen;
  const float* p = big ? partial + e : partial + (long)splits * kWgradBigLen + (e - kWgradBigLen);
  const int n = big ? splits : splits * kWgradSmallSplitFactor;
  // eight loads in flight per thread (the slices are 2.4 MB apart: a dependent chain of single loads ran at 1.2 TB/s);
  // fixed association, so the sum is deterministic
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= n; s += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += __builtin_nontemporal_load(p + (long)(s + j) * len);
  }
  for (int j = 0; s < n; ++s, ++j) v[j] += __builtin_nontemporal_load(p + (long)s * len);
  const float sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  grads[dst] = accumulate ? grads[dst] + sum : sum;
}

// mode: 0 exact fp32 MFMA, 1 bf16x3 (three products per product), 2 bf16x6 (six, exact three-way splits)
template <int OT, int KT, int WR, int WC>
void launch_class(WgArgs a, const int* ids, int n, int mode, hipStream_t st) {
  for (int i = 0; i < n; ++i) a.list[i] = ids[i];
  if (!(OT == 8 && KT == 8)) {             // small GEMMs: more, shorter point slices in their own workspace region
    a.partial += (long)a.splits * kWgradBigLen;
    a.slice_len = a.elem_base_small_len;
    a.elem_base = kWgradBigLen;
    a.splits *= kWgradSmallSplitFactor;
  }
  if (mode == 2 && OT == 8 && KT == 8)
    hipLaunchKernelGGL(field_wgrad_ring_kernel<true>, dim3(a.splits, n), dim3(256), 0, st, a);
  else if (mode == 1 && OT == 8 && KT == 8)
    hipLaunchKernelGGL(field_wgrad_ring_kernel<false>, dim3(a.splits, n), dim3(256), 0, st, a);
  else if (mode == 2)
    hipLaunchKernelGGL((field_wgrad_x3_kernel<OT, KT, WR, WC, true>), dim3(a.splits, n), dim3(256), 0, st, a);
  else if (mode == 1)
    hipLaunchKernelGGL((field_wgrad_x3_kernel<OT, KT, WR, WC, false>), dim3(a.splits, n), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((field_wgrad_kernel<OT, KT, WR, WC>), dim3(a.splits, n), dim3(256), 0, st, a);
}

}  // namespace

extern "C" long dmn_field_grad_floats(long n_points, int ins_num) {
  if (n_points < 0 || ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_BAD_ARG;
  return ((n_points + 127) / 128) * 4 * grad_tile_floats(ins_tiles(ins_num));
}

extern "C" int dmn_field_bwd_data(const float* g_raw, const float* save, const float* bwd_blob, long n_points,
                                  int ins_num, float* grad_record, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!g_raw || !save || !bwd_blob || !grad_record) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  if (T > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  const long blocks = (n_points + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const BwdArgs a{g_raw, save, bwd_blob, grad_record, n_points, ins_num + 1};
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (T) {
    case 1: hipLaunchKernelGGL(field_bwd_data_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(field_bwd_data_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(field_bwd_data_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(field_bwd_data_kernel<4>, grid, block, 0, st, a); break;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

namespace {
// A second stream for the smallest weight-gradient GEMMs (25 KB of LDS, < 100 registers per wave): they fit into a CU
// beside a workgroup of the ring kernel (128 KB of LDS, one wave per SIMD) and run in its shadow instead of behind it.
// Fork / join by events; created once per device on first use.  The stream and its event pair are shared by every caller
// on the device: `mutex` serialises creation and each fork .. join sequence (record / wait pairs of two host threads must
// not interleave); callers on different streams of one thread are ordered by the events themselves.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ready = false;
  std::mutex mutex;
};
// call with s.mutex held by nobody; returns with the device's entry created (or null: no side stream, run in line)
SideStream* side_stream() {
  static SideStream per_device[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideStream& s = per_device[dev];
  std::lock_guard<std::mutex> lock(s.mutex);
  if (!s.ready) {
    if (s.stream == nullptr && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
      s.stream = nullptr;
      return nullptr;
    }
    if (s.fork == nullptr && hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess) { s.fork = nullptr; return nullptr; }
    if (s.join == nullptr && hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) { s.join = nullptr; return nullptr; }
    s.ready = true;
  }
  return &s;
}
}  // namespace

extern "C" long dmn_field_wgrad_workspace_floats(int ins_num, int splits) {
  if (ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles || splits <= 0) return DMN_ERR_BAD_ARG;
  const long small = wgrad_index_len(ins_tiles(ins_num)) - kWgradBigLen;
  return (long)splits * (kWgradBigLen + kWgradSmallSplitFactor * small);
}

extern "C" int dmn_field_bwd_weights(const float* save, const float* grad_record, long n_points, int ins_num,
                                     const int32_t* wgrad_index, int splits, int precision, float* workspace,
                                     float* grads_flat, int accumulate, void* stream) {
  if (n_points < 0 || ins_num < 0 || splits <= 0 || splits > 65535) return DMN_ERR_BAD_ARG;
  if (precision != DMN_PREC_FP32 && precision != DMN_PREC_BF16X3 && precision != DMN_PREC_BF16X6)
    return DMN_ERR_BAD_ARG;                                                                     // fp16 underflows on gradients
  const int x3 = precision == DMN_PREC_BF16X3 ? 1 : (precision == DMN_PREC_BF16X6 ? 2 : 0);     // launch_class mode
  if (n_points == 0) return DMN_OK;
  if (!save || !grad_record || !wgrad_index || !workspace || !grads_flat) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  if (T > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  WgArgs a{};
  a.save = save;
  a.grad = grad_record;
  a.partial = workspace;
  a.n_tiles = ((n_points + 127) / 128) * 4;
  a.grad_tile_floats = grad_tile_floats(T);
  a.slice_len = kWgradBigLen;
  a.elem_base = 0;
  a.elem_base_small_len = wgrad_index_len(T) - kWgradBigLen;
  a.splits = splits;
  wgemm_table(T, a.gemm);
  {
#ifndef DMN_WG_SIDE
#define DMN_WG_SIDE 1
#endif
    SideStream* side = x3 && DMN_WG_SIDE ? side_stream() : nullptr;
    std::unique_lock<std::mutex> fork_join;              // held from fork to join (one caller at a time per device)
    if (side != nullptr) fork_join = std::unique_lock<std::mutex>(side->mutex);
    hipStream_t st2 = st;
    if (side != nullptr && hipEventRecord(side->fork, st) == hipSuccess &&
        hipStreamWaitEvent(side->stream, side->fork, 0) == hipSuccess)
      st2 = side->stream;
    // the smallest GEMMs first, on the side stream: they trickle into the CUs beside the ring kernel's workgroups
    const int view[] = {12};
    launch_class<4, 1, 4, 1>(a, view, 1, x3, st2);
    const int rgbo[] = {15};
    launch_class<1, 4, 1, 4>(a, rgbo, 1, x3, st2);
    const int inso[] = {14};
    switch (T) {
      case 1: launch_class<1, 4, 1, 4>(a, inso, 1, x3, st2); break;
      case 2: launch_class<2, 4, 2, 2>(a, inso, 1, x3, st2); break;
      case 3: launch_class<3, 4, 1, 4>(a, inso, 1, x3, st2); break;
      case 4: launch_class<4, 4, 2, 2>(a, inso, 1, x3, st2); break;
    }
    const int big[] = {1, 2, 3, 4, 5, 7, 8, 9, 10};
    launch_class<8, 8, 2, 2>(a, big, 9, x3, st);
    const int pe[] = {0, 6};
    launch_class<8, 2, 4, 1>(a, pe, 2, x3, st);
    const int half[] = {11, 13};
    launch_class<4, 8, 2, 2>(a, half, 2, x3, st);
    const int dens[] = {16};
    launch_class<1, 8, 1, 4>(a, dens, 1, x3, st);
    if (st2 != st) {                                     // join: the reduction reads every partial sum
      if (hipEventRecord(side->join, st2) != hipSuccess || hipStreamWaitEvent(st, side->join, 0) != hipSuccess) {
        hipStreamSynchronize(st2);                       // the side kernels still write the caller's workspace
        return DMN_ERR_LAUNCH;
      }
    }
  }
  const long n = wgrad_index_len(T);
  hipLaunchKernelGGL(field_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace,
                     wgrad_index, n, splits, grads_flat, accumulate);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
