// Inference field kernel of the full-precision bf16 arithmetic ("bf16x6", field_x6.h) on v_mfma_f32_16x16x32_bf16:
// encodings -> the whole network -> raw rows (networks/render.py:37-38,49-61,71-83, networks/dm_nerf.py:80-106), the
// same six-product arithmetic on exact three-way bf16 splits, the same weight bytes and stage map as
// field_fwd_x6_kernel -- on 16 x 16 tiles with 32-deep k-steps (field_layout.h "x6w").
//
// Why the other shape: the kernel runs at the board's power limit, where throughput is power / energy per point.  The
// 16x16x32 instruction moves half the accumulator traffic per FLOP of 32x32x16 (a quarter of the C/D elements per
// instruction at half the FLOP), and on this structure -- same LDS bytes, same vector work per group -- the chip holds
// a 12 % higher clock for 10 % more cycles (tools/mfma_x6_shapes.hip, profiles/r04c_mfma_x6_shapes.txt: +6.5 %).
//
// Mapping: a wave's 32 points are two 16-point tiles p = 0, 1; lane l = (n = l & 15, q = l >> 4) holds, for point 16 p + n,
// features 16 t + 4 q + r (r = 0..3) of every feature tile t.  A layer is H'^T = W H^T as before: weights are the A operand
// (one [hi, mid, lo] chunk triple per (k-step, feature tile) group feeds BOTH point tiles: 12 MFMAs per triple), the layer
// before's accumulators are the B operand: the 8 k-slots of lane group q in k-step ks are the four registers of feature
// tiles 2 ks and 2 ks + 1 -- a k permutation the weight packing absorbs (field_pack.cpp, layer_x6w).  Two accumulator
// sets (2 x 128 registers) alternate as input and output, a k-step's operands are read, activated and split one
// k-step ahead in the MFMA shadow, exactly as in field_x6.h.  The saving (training) forward and the fused kernel stay on
// the 32x32x16 kernel: their save record and compositing slab are tied to its lane map.
//
// Launch: ONE workgroup per CU (163 KB of LDS: a CU holds one) that walks the launch's 128-point tiles round robin
// (x6w_tiles).  The weight stream wraps from a tile's last stage into stage 0 of the next, the const section is loaded once,
// and the dispatcher starts 256 workgroups instead of ~100 000: -6.7 % per launch against one tile per workgroup, where
// every tile paid dispatch + const section + the latency of its first stage with nothing of its CU's to hide behind.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_x6.h"

using namespace dmn;

namespace {


__device__ __forceinline__ float acc_read1(float v) {      // one register of a FINISHED accumulator tile (field_x3.h acc_read)
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
  return x;
}
__device__ __forceinline__ f32x4 mfma_w(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- a layer's input: value of k-slot j (0..7) of k-step ks for point tile p ------------------------------------------
template <int NT_IN>
struct SrcTilesW {                       // the accumulators of the layer before
  const f32x4 (*a)[NT_IN];
  __device__ __forceinline__ float operator()(int p, int ks, int j) const { return acc_read1(a[p][2 * ks + (j >> 2)][j & 3]); }
};
// The encodings, read where they are consumed out of the exchange area in LDS (see the kernel): channel 32 ks + 8 q + j of
// the CHANNELS channels that start at exchange row CH0, for point 16 p + n; `lane_base` = exchange row 8 q, column n of
// this wave.  Channels past the end are zero padding (never the neighbouring rows' contents).
constexpr int kXchRow = 33;                            // floats per channel row of the exchange: 32 points + 1 (banks)
template <int CH0, int CHANNELS>
struct SrcXch {
  const float* lane_base;
  int q;
  __device__ __forceinline__ float operator()(int p, int ks, int j) const {
    const float x = lane_base[(CH0 + 32 * ks + j) * kXchRow + 16 * p];
    return 32 * ks + j + 24 >= CHANNELS ? ((32 * ks + j + 8 * q < CHANNELS) ? x : 0.f) : x;     // only q = 3 can run past the end
  }
};
// h8 = relu(mlps.7) for its second reader (ins_feature_linear): feature tiles 0 .. kKeepTo-1 activated in vector registers
// (kept by rgb_feature_linear's operand split, see x6w_network), tiles kKeepTo .. 15 still in their accumulators.
// kKeepTo = 0: all of h8 stays in its accumulators and is activated again where it is read again.  The register file is
// full to the last tuple either way and the allocator's answer to one more live value is scratch, whose traffic evicts the
// weight stream from the L2 (FETCH_SIZE 5.7 GB per launch against 0.85 with 164 B of it), so this is measured, not reasoned:
// as one tile per workgroup the kernel was free of scratch with the FIRST ten tiles kept (rgb_feature_linear consumes its
// input in ascending order; its 128 output registers then meet at most 96 of h8) and spilled with none; as a loop over tiles
// (x6w_tiles) it spills 17 / 53 / 81 / 124 registers with 4 / 8 / 10 / 14 tiles kept and none with 0 or 2 (equal in time).
#ifndef DMN_X6W_KEEP
#define DMN_X6W_KEEP 0
#endif
constexpr int kKeepTo = DMN_X6W_KEEP;
constexpr int kKeepRegs = kKeepTo ? 4 * kKeepTo : 1;
struct SrcH8W {
  const f32x4 (*a)[16];
  const float (*keep)[kKeepRegs];
  __device__ __forceinline__ float operator()(int p, int ks, int j) const {
    const int t = 2 * ks + (j >> 2);
    return t < kKeepTo ? keep[p][t < kKeepTo ? t * 4 + (j & 3) : 0] : relu_bits(acc_read1(a[p][t < kKeepTo ? 15 : t][j & 3]));
  }
};

// One layer, k-steps outer: out[p][t] (t < NT feature tiles of 16, p = 0, 1) = bias + sum over the NKS k-steps of `src`
// (ReLU first if RELU) and the NKX k-steps of `ext` (encodings) of the six-term product with the stream's chunk triples.
//   bias   LDS address of this lane group's bias floats of tile 0 (natural feature order: + 16 t per tile)
//   KEEP   the activated values of source tiles 0 .. kKeepTo-1 stay behind in keep[p][4 t + r]
//   LAST   the layer ends the weight stream: its last stage issues LDS-DMA (stage 0 again) only if `more` (another tile follows)
template <int LID, int NT, int NKS, int NKX, bool RELU, bool KEEP, bool LAST, class Src, class Ext>
__device__ __forceinline__ void layer_w(Ctx6& c, f32x4 (*out)[NT], const float* bias, const Src& src, const Ext& ext,
                                        float (*keep)[kKeepRegs] = nullptr, bool more = false) {
  constexpr int NK = NKS + NKX;
  constexpr int KPS = x6w_kps(NT);              // k-steps per 16-group stage
  constexpr bool NTP = ((DMN_X6_NTMASK >> LID) & 1) != 0;
  static_assert(KPS >= 1 && KPS * NT * 3 <= kX6StageChunks, "a stage holds three chunks per group");
  // bias = initial accumulator of both point tiles: handed to the first MFMA of a tile's chain in k-step 0, read from LDS one
  // group ahead.  (All 16 bias tuples loaded up front sit in accumulator registers beside the two full accumulator sets:
  // the allocator then moves finished tiles to vector registers and, a layer later, to scratch.)
  f32x4 bias_next = *reinterpret_cast<const f32x4*>(bias);
  // value pair pr (0..7) of k-step ks: point tile pr >> 2, k-slots 2 (pr & 3), 2 (pr & 3) + 1
  auto build_pair = [&](int ks, int pr, frag* fh, frag* fm, frag* fl) {
    const int p = pr >> 2, jp = pr & 3;
    float x0, x1;
    if (ks < NKS) {
      x0 = src(p, ks, 2 * jp);
      x1 = src(p, ks, 2 * jp + 1);
      if (RELU) {
        x0 = relu_bits(x0);
        x1 = relu_bits(x1);
      }
      const int t = 2 * ks + (jp >> 1);         // source feature tile
      if (KEEP && t < kKeepTo) {
        keep[p][t * 4 + 2 * (jp & 1)] = x0;
        keep[p][t * 4 + 2 * (jp & 1) + 1] = x1;
      }
    } else {
      x0 = ext(p, ks - NKS, 2 * jp);
      x1 = ext(p, ks - NKS, 2 * jp + 1);
    }
    split3_pair(x0, x1, fh[p], fm[p], fl[p], 2 * jp);
  };
  frag bh[2], bm[2], bl[2], nh[2], nm[2], nl[2];
#pragma unroll
  for (int pr = 0; pr < 8; ++pr) build_pair(0, pr, bh, bm, bl);
  frag ah[2], am[2], al[2];                     // A fragments of the running group and of the next one
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    constexpr int kLastStage = (NK - 1) / KPS;
    const int stage = ks / KPS;
    const int groups = (stage == kLastStage ? NK - stage * KPS : KPS) * NT;
    const bool dma = !(LAST && stage == kLastStage) || more;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int gi = (ks % KPS) * NT + t;       // group inside its stage
      const int ar = (ks * NT + t) & 1;
      if (gi == 0) {
        x6_stage_begin(c);
        ah[ar] = *reinterpret_cast<const frag*>(c.buf);
        am[ar] = *reinterpret_cast<const frag*>(c.buf + 1024);
        al[ar] = *reinterpret_cast<const frag*>(c.buf + 2048);
      }
      if (gi + 1 < groups) {                    // next group's fragments fly under this group's MFMAs
        ah[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1)) * 1024);
        am[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 1) * 1024);
        al[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 2) * 1024);
      }
      f32x4 bias_cur = bias_next;
      if (ks == 0 && t + 1 < NT) bias_next = *reinterpret_cast<const f32x4*>(bias + 16 * (t + 1));
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch above this group's MFMAs
#pragma unroll
      for (int p = 0; p < 2; ++p) {             // small terms first (field_x6.h)
        out[p][t] = mfma_w(al[ar], bh[p], ks == 0 ? bias_cur : out[p][t]);
        out[p][t] = mfma_w(ah[ar], bl[p], out[p][t]);
        out[p][t] = mfma_w(am[ar], bm[p], out[p][t]);
        out[p][t] = mfma_w(am[ar], bh[p], out[p][t]);
        out[p][t] = mfma_w(ah[ar], bm[p], out[p][t]);
        out[p][t] = mfma_w(ah[ar], bh[p], out[p][t]);
      }
      if (dma) {
#if DMN_X6W_EARLY_DMA       // experiment: two pieces per group in the first six groups (longer lead before the stage barrier)
        if (2 * gi < kX6PiecesPerWave) { x6_piece<NTP>(c, 2 * gi); x6_piece<NTP>(c, 2 * gi + 1); }
        if (gi + 1 == groups) {
#pragma unroll
          for (int qq = 2 * (gi + 1); qq < kX6PiecesPerWave; ++qq) x6_piece<NTP>(c, qq);
        }
#else
        if (gi < kX6PiecesPerWave) x6_piece<NTP>(c, gi);
        if (gi + 1 == groups) {                 // a short stage still owes the rest of its pieces
#pragma unroll
          for (int qq = gi + 1; qq < kX6PiecesPerWave; ++qq) x6_piece<NTP>(c, qq);
        }
#endif
      }
      if (ks + 1 < NK) {                        // the next k-step's operand fragments, spread over the tiles
#pragma unroll
        for (int pr = 0; pr < 8; ++pr)
          if ((pr * NT) / 8 == t) build_pair(ks + 1, pr, nh, nm, nl);
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bh[p] = nh[p];
      bm[p] = nm[p];
      bl[p] = nl[p];
    }
  }
}

// sum over the four lane groups of a point (q = lane >> 4)
__device__ __forceinline__ float sum_groups(float x) {
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// The whole network on this wave's 32 points (networks/dm_nerf.py:80-106).  xch_lane: this lane's base in the encoding
// exchange (SrcXch).  Outputs per point tile (valid in all four lane groups): sigma[p], rgb[p][3];
// acc3[p][t][r] = instance logit of channel 16 t + 4 q + r.  With sigma_only the network stops behind its density head
// and returns true (see x6_network).
template <int T16>
__device__ __forceinline__ bool x6w_network(Ctx6& c, const float* cst, const float* xch_lane, int q, bool sigma_only,
                                            bool more, float (&sigma)[2], float (&rgb)[2][3], f32x4 (&acc3)[2][T16]) {
  f32x4 accA[2][16], accB[2][16];
  const SrcXch<0, kPosCh> s_pe{xch_lane, q};
  const SrcXch<kPosCh, kDirCh> s_qe{xch_lane, q};
  const SrcTilesW<16> s_a{accA}, s_b{accB};
  const float* const b8 = cst + kConstBias8 + 4 * q;

  // ---- mlps.0: position encoding (2 k-steps) -> A
  layer_w<0, 16, 2, 0, false, false, false>(c, accA, b8, s_pe, s_pe);
  // ---- mlps.1 .. mlps.4 in pairs A -> B -> A (rolled)
#pragma unroll 1
  for (int l = 1; l < 5; l += 2) {
    layer_w<1, 16, 8, 0, true, false, false>(c, accB, b8 + l * 256, s_a, s_pe);
    layer_w<1, 16, 8, 0, true, false, false>(c, accA, b8 + (l + 1) * 256, s_b, s_pe);
  }
  // ---- mlps.5 on cat([h, input_pts]) (dm_nerf.py:86-87), mlps.6, mlps.7: A -> B -> A -> B   (c.g: see x6_network)
  c.g = kX6StageL5;
  layer_w<2, 16, 8, 2, true, false, false>(c, accB, b8 + 5 * 256, s_a, s_pe);
  c.g = kX6StageL6;
  layer_w<3, 16, 8, 0, true, false, false>(c, accA, b8 + 6 * 256, s_b, s_pe);
  c.g = kX6StageL6 + 8;
  layer_w<4, 16, 8, 0, true, false, false>(c, accB, b8 + 7 * 256, s_a, s_pe);

  // ---- density_linear on h8 = relu(B) (dm_nerf.py:101): 256 -> 1 on the VALU, 64 terms per lane and point.  h8 is read
  //      again by the next two layers, so it is read here as they read it (acc_read1: the tiles stay where they are; given
  //      plain reads the compiler keeps vector-register copies of ten tiles alive across rgb_feature_linear and spills
  //      them).  Inline asm gets no MFMA -> VALU wait states from the compiler: tiles in ascending order put >= 12 MFMAs
  //      between a tile's last MFMA and its read, except for the last ones, which the s_nops cover.
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {               // (a tile at a time: left free, all 64 weight reads are hoisted together)
      const f32x4 w = *reinterpret_cast<const f32x4*>(cst + kConstDensW + 16 * t + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) s = fmaf(relu_bits(acc_read1(accB[p][t][r])), w[r], s);
      __builtin_amdgcn_sched_barrier(0);
    }
    sigma[p] = sum_groups(s) + cst[kConstDensB];
  }
  if (sigma_only) return true;      // (the stage in flight is stage 0 of the next tile: Ctx6::wrap)

  // ---- rgb_feature_linear on h8 (no activation behind it, dm_nerf.py:89) -> A; feature tiles 0 .. kKeepTo-1 of h8 stay
  //      behind activated in vector registers, their accumulators take the 8-tile layers' output
  float keep[2][kKeepRegs];
  f32x4 acc2[2][8];
  c.g = kX6StageRgbF;
  layer_w<5, 16, 8, 0, true, true, false>(c, accA, b8 + 8 * 256, s_b, s_pe, keep);
  // ---- rgb_feature_linears.0 on cat([feature, dirs]) -> 128, ReLU (dm_nerf.py:90-93); rgb_linear on the VALU (:102)
  c.g = kX6StageRgbH;
  layer_w<6, 8, 8, 1, false, false, false>(c, acc2, cst + kConstBias4 + 4 * q, s_a, s_qe);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(cst + kConstRgbW + k * 128 + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[k] = fmaf(relu_bits(acc2[p][t][r]), w[r], s[k]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[p][k] = sum_groups(s[k]) + cst[kConstRgbB + k];
  }
  // ---- ins_feature_linear on the same trunk output (h.detach(), dm_nerf.py:95-96), no activation -> A;
  //      ins_feature_linears.0 -> 128, ReLU (:97-99); ins_linear -> ins_num + 1 (:103)
  c.g = kX6StageInsF;
  layer_w<7, 16, 8, 0, false, false, false>(c, accA, b8 + 9 * 256, SrcH8W{accB, keep}, s_pe);
  c.g = kX6StageInsH;
  layer_w<8, 8, 8, 0, false, false, false>(c, acc2, cst + kConstBias4 + 128 + 4 * q, s_a, s_pe);
  c.g = kX6StageInsO;
  layer_w<9, T16, 4, 0, true, false, true>(c, acc3, cst + kConstBiasO + 4 * q, SrcTilesW<8>{acc2}, s_pe, nullptr, more);
  return false;
}

// (store_rows_coalesced: field_fwd_x6.hip; repeated here, both are file-local)
__device__ __forceinline__ void store_rows_coalesced_w(const float* stage, float* dst, int n_floats, int tid) {
  const int nvec = n_floats >> 2;
  for (int i = tid; i < nvec; i += 256)
    __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(stage)[i], reinterpret_cast<f32x4*>(dst) + i);
  const int rest = n_floats & 3;
  if (tid < rest) __builtin_nontemporal_store(stage[4 * nvec + tid], dst + 4 * nvec + tid);
}

// Encodings of point mc into this wave's exchange column `col` (= xch + point): row ch * kXchRow holds channel ch
// (0..62 position, 63..89 view direction; networks/dm_nerf.py:17-32 order: x y z, then per band sin xyz, cos xyz).
// The two lanes of a point (h = 0, 1) share its six coordinates -- h = 0: position x, position y, direction z; h = 1:
// position z, direction x, direction y -- and each writes BOTH the sin and the cos rows of its three: one fp64 sincos
// and the double-angle chain (trig.h) per coordinate, i.e. half of what the shared prologue (point_encodings: every lane
// all six, keeping sin or cos) spends, with the same doubles rounded to the same floats.
__device__ __forceinline__ void encodings_to_exchange(const FieldArgs& a, long mc, int h, float* col) {
  if (a.emb != nullptr) {                       // embedded mode: copy the row (45 channels per lane)
    const float* row = a.emb + mc * (kPosCh + kDirCh) + 45 * h;
#pragma unroll
    for (int j = 0; j < 45; ++j) col[(45 * h + j) * kXchRow] = row[j];
    return;
  }
  const long n = mc / a.S;
  const int s = (int)(mc - n * a.S);
  const float ox = a.rays_o[3 * n + 0], oy = a.rays_o[3 * n + 1], oz = a.rays_o[3 * n + 2];
  const float dx = a.rays_d[3 * n + 0], dy = a.rays_d[3 * n + 1], dz = a.rays_d[3 * n + 2];
  const float zv = a.z[n * a.z_stride + s];
  // pts = rays_o + rays_d * z, viewdirs = rays_d / ||rays_d||: separately rounded, as torch (field_common.h point_encodings)
  const float px = __fadd_rn(ox, __fmul_rn(dx, zv));
  const float py = __fadd_rn(oy, __fmul_rn(dy, zv));
  const float pz = __fadd_rn(oz, __fmul_rn(dz, zv));
  const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float vx = __fdiv_rn(dx, nrm), vy = __fdiv_rn(dy, nrm), vz = __fdiv_rn(dz, nrm);
  // chain A: position x | z (10 bands); chain B: position y (10 bands) | direction x (4); chain C: direction z | y (4)
  const float ca = h ? pz : px, cb = h ? vx : py, cc = h ? vy : vz;
  float* const ra = col + (h ? 2 : 0) * kXchRow;                 // row of the identity channel of each chain
  float* const rb = col + (h ? kPosCh + 0 : 1) * kXchRow;
  float* const rc = col + (kPosCh + (h ? 1 : 2)) * kXchRow;
  ra[0] = ca; rb[0] = cb; rc[0] = cc;
  double sa, cas, sb, cbs, sc, ccs;
  sincos_f64(ca, sa, cas);
  sincos_f64(cb, sb, cbs);
  sincos_f64(cc, sc, ccs);
#pragma unroll
  for (int f = 0; f < kPosFreqs; ++f) {                           // sin row of band f: +3 + 6 f, cos row: +6 + 6 f
    ra[(3 + 6 * f) * kXchRow] = (float)sa;
    ra[(6 + 6 * f) * kXchRow] = (float)cas;
    if (f < kDirFreqs || h == 0) {                                // chain B is a direction (4 bands) in half 1
      rb[(3 + 6 * f) * kXchRow] = (float)sb;
      rb[(6 + 6 * f) * kXchRow] = (float)cbs;
    }
    if (f < kDirFreqs) {
      rc[(3 + 6 * f) * kXchRow] = (float)sc;
      rc[(6 + 6 * f) * kXchRow] = (float)ccs;
    }
    if (f + 1 < kPosFreqs) {
      double_angle(sa, cas);
      double_angle(sb, cbs);
      if (f + 1 < kDirFreqs) double_angle(sc, ccs);
    }
  }
}

constexpr int kXchWave = (kPosCh + kDirCh) * kXchRow;  // 2,970 floats per wave
constexpr int kXchBytes = 4 * kXchWave * 4;            // 47,520 B behind the stage slots and the const section
static_assert(kX6LdsBytes + kXchBytes <= 160 * 1024, "LDS budget of one CU");

// The point tiles of one workgroup (tile = 128 points; tiles blockIdx.x, + gridDim.x, ...: the launch has one workgroup per
// CU, which is what the LDS lets a CU hold).  SIGMA: the density-only pass, a loop of its own (field_fwd_x6.hip).
// The weight stream wraps from a tile's last stage to stage 0 of the next tile, so that only the first tile of a workgroup
// waits for the const section and for its first stage, and the dispatch of a workgroup is paid once per CU.
template <int T16, bool SIGMA>
__device__ __forceinline__ void x6w_tiles(Ctx6& c, char* lds, const float* cst) {
  float* const rows = reinterpret_cast<float*>(lds + kX6LdsBytes);    // the exchange area; a tile's raw rows once it is through
  int tile = blockIdx.x;
  for (;;) {
    // What a tile needs of the launch's arguments and of the lane's identity is this iteration's: re-read from the kernel
    // argument segment, re-derived from an opaque lane id.  (Left loop-invariant, ~40 scalar and ~80 vector registers of
    // hoisted addresses and arguments live across the whole network, which has none to spare: 336 B of scratch.)
    typedef const __attribute__((address_space(4))) FieldArgs* ArgPtr;
    ArgPtr ap = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap), "+s"(c.stream), "+s"(c.wave_s), "+s"(c.par));
    const FieldArgs a{ap->rays_o, ap->rays_d, ap->z, ap->z_stride, ap->emb, nullptr, ap->raw, nullptr, ap->M, ap->S, ap->ins_ch,
                      nullptr, 0};
    const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));     // (no register carries it over)
    const int tid = c.wave_s * 64 + lane, n = lane & 15, q = lane >> 4;
    c.lane = lane;
    // ---- encodings: computed by the two lanes of a point (lane = point + 32 * half: three coordinates each,
    //      encodings_to_exchange) and left in LDS for the lanes that need them as k-slots: lane (n, q) of point tile p
    //      takes channels 32 e + 8 q + j of point 16 p + n, where the layers consume them (SrcXch: 80 reads per pass; held
    //      in registers they cost 48 for most of the kernel).  Rows of 33 floats keep the 32 points of a channel on 32 banks.
    float* const xch = rows + c.wave_s * kXchWave;
    const float* const xch_lane = xch + 8 * q * kXchRow + n;
    const int C = 4 + a.ins_ch;
    const int n_tiles = (int)((a.M + 127) / 128);
    const bool more = tile + (int)gridDim.x < n_tiles;
    {
      const int pt = lane & 31, h = lane >> 5;
      const long m = ((long)tile * 4 + c.wave_s) * 32 + pt;
      encodings_to_exchange(a, m < a.M ? m : a.M - 1, h, xch + pt);
      __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): written before this wave's first read (same wave: in order anyway)
    }
    float sigma[2], rgb[2][3];
    f32x4 acc3[2][T16];
    c.g = 0;
    x6w_network<T16>(c, cst, xch_lane, q, SIGMA, more, sigma, rgb, acc3);
    // ---- raw = cat([rgb, density, ins]) (dm_nerf.py:105): gathered in LDS, stored as whole lines (field_fwd_x6.hip).  In
    //      the exchange area: every wave read its last encoding at least thirteen stage barriers ago (the density-only pass:
    //      sixteen), and the stage slots already receive the next tile's stage 0.
    //      Rows wider than 92 floats (128 of them no longer fit the exchange's 47.5 KB) leave in two halves of 64.
    const bool halves = 128 * C * 4 > kXchBytes;
    const long first = (long)tile * 128;
    auto write_rows = [&](int slot) {            // this wave's 32 rows into row slots 32 slot .. 32 slot + 31 of the area
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float* const row = rows + (slot * 32 + 16 * p + n) * C;
        if (q == 0) {
          row[0] = SIGMA ? 0.f : rgb[p][0]; row[1] = SIGMA ? 0.f : rgb[p][1]; row[2] = SIGMA ? 0.f : rgb[p][2];
          row[3] = sigma[p];
        }
#pragma unroll
        for (int t = 0; t < T16; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ch = 16 * t + 4 * q + r;
            if (ch < a.ins_ch) row[4 + ch] = SIGMA ? 0.f : acc3[p][t][r];       // density only: raw = [0, 0, 0, sigma, 0 ...]
          }
      }
    };
    auto store_rows = [&](long r0, int cap) {    // rows r0 .. of the launch (at most cap of them) out of the area
      const long left = a.M - r0;
      store_rows_coalesced_w(rows, a.raw + r0 * C, (int)(left < 0 ? 0 : (left < cap ? left : cap)) * C, tid);
    };
    if (!halves) {
      write_rows(c.wave_s);
      __syncthreads();
      store_rows(first, 128);
    } else {
      if (c.wave_s < 2) write_rows(c.wave_s);
      __syncthreads();
      store_rows(first, 64);
      __syncthreads();                            // the first half is read: the second overwrites it
      if (c.wave_s >= 2) write_rows(c.wave_s - 2);
      __syncthreads();
      store_rows(first + 64, 64);
    }
    if (!more) break;
    __syncthreads();                              // the rows are read; the next tile's encodings overwrite them
    c.par = (c.par + c.G) % kX6Slots;
    tile += (int)gridDim.x;
  }
}

template <int T16>
__global__ __launch_bounds__(256, 1) void field_fwd_x6w_kernel(const FieldArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kX6LdsBytes + kXchBytes];
  float* const cst = reinterpret_cast<float*>(lds + kX6Slots * kX6StageBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const blob = static_cast<const char*>(a.blob);
  // the density-only pass ends behind the trunk: its stream wraps there
  Ctx6 c{blob + kConstFloats * 4, lds, nullptr, 0, a.sigma_only ? kX6StageRgbF : x6_total_stages(ins_tiles(a.ins_ch - 1)),
         wave_s, lane, 0, true};

  for (int k = wave_s; k < kConstChunks; k += 4)
    glds16(blob + k * 1024 + lane * 16, lds + kX6Slots * kX6StageBytes + k * 1024);
  x6_load_stage(c, 0);
  const_section_ready();      // mlps.0 reads its bias before its first stage barrier (once per workgroup)
  if (a.sigma_only) x6w_tiles<T16, true>(c, lds, cst);
  else x6w_tiles<T16, false>(c, lds, cst);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // a density-only tile's last stage prefetches past its end: landed
                                                        // before the LDS is let go of
}

// one workgroup per CU (the LDS holds one): the tiles are dealt round robin
int x6w_workgroups(long tiles) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return tiles < cus ? (int)tiles : cus;
}

int launch_x6w(const FieldArgs& a, int ins_num, hipStream_t st) {
  if (ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  if (a.M <= 0) return DMN_OK;
  const dim3 grid((unsigned)x6w_workgroups((a.M + 127) / 128)), block(256);
  switch (x6w_ins_tiles(ins_num)) {
    case 1: hipLaunchKernelGGL(field_fwd_x6w_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(field_fwd_x6w_kernel<2>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(field_fwd_x6w_kernel<4>, grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL(field_fwd_x6w_kernel<6>, grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL(field_fwd_x6w_kernel<8>, grid, block, 0, st, a); break;
    default: return DMN_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

}  // namespace

extern "C" int dmn_field_fwd_rays_x6w(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                      long n_rays, int n_samples, const void* blob, int ins_num, int sigma_only,
                                      float* raw, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, nullptr, sigma_only ? 1 : 0};
  return launch_x6w(a, ins_num, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_embedded_x6w(const float* embedded, long n_points, const void* blob, int ins_num,
                                          float* raw, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!embedded || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{nullptr, nullptr, nullptr, 0, embedded, blob, raw, nullptr, n_points, 1, ins_num + 1, nullptr, 0};
  return launch_x6w(a, ins_num, (hipStream_t)stream);
}
