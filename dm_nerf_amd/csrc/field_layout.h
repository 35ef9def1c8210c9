// Packed-weight layout of one DM-NeRF field network for the fused gfx950 field kernel.
//
// The kernel keeps the activations of 32 sample points per wave in registers as the
// *transposed* tile H^T[feature][point] produced by v_mfma_f32_32x32x2_f32 (C/D map:
// column = lane&31 = point, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = feature within a
// 32-row tile).  A layer is H'^T = W * H^T: the weight matrix is the MFMA A operand
// (row = output feature), the previous layer's accumulator registers are fed straight back
// as the B operand (k = feature, n = point) with no LDS round trip.  Because register i of
// lane-half h holds feature feat_of(i,h), k-step i of the next layer multiplies
// W[:, feat_of(i,0)] (lanes 0-31) and W[:, feat_of(i,1)] (lanes 32-63): the k order is
// permuted, which costs nothing since the weights are packed once into exactly the order
// the kernel consumes them (this file).
//
// Blob = [const section | weight stream].  The stream is a sequence of 1 KiB chunks (one
// wave-wide 16-byte LDS-DMA each); 32 chunks form a 32 KiB stage, and every layer is padded
// to whole stages so the kernel walks the stream with a single running stage counter.
//
// Reference shapes: networks/dm_nerf.py:59-78 (layers), :80-106 (dataflow).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define DMN_HD __host__ __device__
#else
#define DMN_HD
#endif

namespace dmn {

constexpr int kPosCh = 63;          // 3 + 3*2*10  (config.py:56, dm_nerf.py:45-52)
constexpr int kDirCh = 27;          // 3 + 3*2*4   (config.py:58)
constexpr int kPosFreqs = 10;
constexpr int kDirFreqs = 4;
constexpr int kWidth = 256;         // config.py:25
constexpr int kHalf = 128;
constexpr int kDepth = 8;           // config.py:23
constexpr int kSkipLayer = 5;       // mlps[5] takes cat([h, input_pts]) (dm_nerf.py:65-67,86-87)

constexpr int kChunkFloats = 256;   // 64 lanes x float4
constexpr int kStageChunks = 32;
constexpr int kStageFloats = kChunkFloats * kStageChunks;   // 8192 floats = 32 KiB
constexpr int kMaxInsTiles = 4;     // ins_num + 1 <= 128

// ---- which input feature a lane-half holds -----------------------------------------
// activation register i (0..127 for 256-wide, 0..63 for 128-wide) of lane-half h
DMN_HD constexpr int feat_of(int i, int h) { return 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * h; }

// positional-encoding register j (0..31) of lane-half h -> column of the 63-vector, or -1.
// half 0: x, y, then sin terms; half 1: z, (pad), then cos terms.  (dm_nerf.py:17-32 ordering)
DMN_HD constexpr int pos_slot(int j, int h) {
  if (j == 0) return h == 0 ? 0 : 2;
  if (j == 1) return h == 0 ? 1 : -1;
  return 3 + 6 * ((j - 2) / 3) + 3 * h + (j - 2) % 3;
}
// view-direction register j (0..15) of lane-half h -> column of the 27-vector, or -1.
DMN_HD constexpr int dir_slot(int j, int h) {
  if (j == 0) return h == 0 ? 0 : 2;
  if (j == 1) return h == 0 ? 1 : -1;
  if (j >= 14) return -1;
  return 3 + 6 * ((j - 2) / 3) + 3 * h + (j - 2) % 3;
}

// ---- flat parameter order (= reference state_dict order) -------
// mlps.{0..7}.{weight,bias}, rgb_feature_linear, ins_feature_linear, rgb_feature_linears.0,
// ins_feature_linears.0, density_linear, ins_linear, rgb_linear   (each weight then bias)
struct ParamOffsets {
  int mlp_w[kDepth], mlp_b[kDepth];
  int rgbf_w, rgbf_b, insf_w, insf_b, rgbh_w, rgbh_b, insh_w, insh_b, dens_w, dens_b, inso_w, inso_b, rgbo_w, rgbo_b;
  int total;
};
DMN_HD constexpr int mlp_fan_in(int l) { return l == 0 ? kPosCh : (l == kSkipLayer ? kWidth + kPosCh : kWidth); }
DMN_HD constexpr ParamOffsets param_offsets(int ins_num) {
  ParamOffsets o{};
  int p = 0;
  for (int l = 0; l < kDepth; ++l) {
    o.mlp_w[l] = p; p += kWidth * mlp_fan_in(l);
    o.mlp_b[l] = p; p += kWidth;
  }
  o.rgbf_w = p; p += kWidth * kWidth;            o.rgbf_b = p; p += kWidth;
  o.insf_w = p; p += kWidth * kWidth;            o.insf_b = p; p += kWidth;
  o.rgbh_w = p; p += kHalf * (kWidth + kDirCh);  o.rgbh_b = p; p += kHalf;
  o.insh_w = p; p += kHalf * kWidth;             o.insh_b = p; p += kHalf;
  o.dens_w = p; p += kWidth;                     o.dens_b = p; p += 1;
  o.inso_w = p; p += (ins_num + 1) * kHalf;      o.inso_b = p; p += ins_num + 1;
  o.rgbo_w = p; p += 3 * kHalf;                  o.rgbo_b = p; p += 3;
  o.total = p;
  return o;
}

// ---- const section (biases and the two VALU heads), floats --------------------------
// bias of an 8-tile layer, slot s (0..7 = mlps, 8 = rgb_feature_linear, 9 = ins_feature_linear):
//   [kConstBias8 + s*256 + h*128 + i] = b[feat_of(i,h)]
constexpr int kConstBias8 = 0;
// bias of a 4-tile layer, slot s (0 = rgb_feature_linears.0, 1 = ins_feature_linears.0):
//   [kConstBias4 + s*128 + h*64 + i] = b[feat_of(i,h)]
constexpr int kConstBias4 = 2560;
// ins_linear bias: [kConstBiasO + h*64 + i] = b[feat_of(i,h)] or 0 past ins_num   (i < 16*T)
constexpr int kConstBiasO = 2816;
// density_linear: [kConstDensW + h*128 + i] = w[feat_of(i,h)]; bias at kConstDensB
constexpr int kConstDensW = 2944;
constexpr int kConstDensB = 3200;
// rgb_linear: [kConstRgbW + c*128 + h*64 + i] = w[c][feat_of(i,h)]; biases at kConstRgbB + c
constexpr int kConstRgbW = 3204;
constexpr int kConstRgbB = 3588;
constexpr int kConstChunks = 15;
constexpr int kConstFloats = kConstChunks * kChunkFloats;    // 3840 >= 3592

// ---- weight stream: stage index where each layer starts ------------------------------
// 8-tile layers: k-step j = chunks (2j, 2j+1); chunk (2j+q)[lane][e] = W[32*(4q+e) + (lane&31)][col(j, lane>>5)]
// 4-tile layers: k-step j = chunk j;           chunk j[lane][e]      = W[32*e + (lane&31)][col(j, lane>>5)]
// ins_linear (T tiles): group jg of 4 k-steps, tile t = chunk jg*T + t; [lane][e] = W[32*t + (lane&31)][feat_of(4*jg+e, lane>>5)]
DMN_HD constexpr int ins_tiles(int ins_num) { return (ins_num + 1 + 31) / 32; }
DMN_HD constexpr int inso_groups_per_stage(int T) { return T <= 2 ? 16 : 8; }
DMN_HD constexpr int inso_stages(int T) { return 16 / inso_groups_per_stage(T); }

constexpr int kStageL0 = 0;                 // 32 k-steps  -> 2 stages
constexpr int kStageL1 = 2;                 // mlps.1..4: 8 stages each
constexpr int kStageL5 = 34;                // 128 + 32 k-steps -> 10 stages
constexpr int kStageL6 = 44;                // mlps.6, mlps.7
constexpr int kStageRgbF = 60;              // rgb_feature_linear      8
constexpr int kStageRgbH = 68;              // rgb_feature_linears.0   128+16 k-steps, 4 tiles -> 5 stages (padded)
constexpr int kStageInsF = 73;              // ins_feature_linear      8
constexpr int kStageInsH = 81;              // ins_feature_linears.0   128 k-steps, 4 tiles -> 4 stages
constexpr int kStageInsO = 85;              // ins_linear              1 or 2 stages
DMN_HD constexpr int total_stages(int T) { return kStageInsO + inso_stages(T); }
DMN_HD constexpr long blob_floats(int ins_num) {
  return (long)kConstFloats + (long)total_stages(ins_tiles(ins_num)) * kStageFloats;
}
DMN_HD constexpr int mlp_stage(int l) {
  return l == 0 ? kStageL0 : (l <= 4 ? kStageL1 + 8 * (l - 1) : (l == 5 ? kStageL5 : kStageL6 + 8 * (l - 6)));
}

// =====================================================================================
// Split-precision ("x3") blob: same const section (fp32), weight stream in 16-bit hi / lo halves
// for the 3-MFMA product  W*X ~= Whi*Xhi + Whi*Xlo + Wlo*Xhi  on v_mfma_f32_32x32x16_{f16,bf16}.
//
// A k-step now covers 16 input features: element j (0..7) of lane-half h in k-step ks is activation
// register 8*ks + j, i.e. feature feat_of(8*ks + j, h) (same map as above, see field_fwd_x3.hip).
// The kernel walks output tiles in the OUTER loop: for tile t, for k-step ks: [hi chunk, lo chunk];
// chunk[lane][j] = W[32*t + (lane&31)][col(8*ks + j, lane>>5)], 64 lanes x 8 halves = 1 KiB.
// 64 chunks = one 64 KiB stage; every layer is padded to whole stages.
// =====================================================================================
constexpr int kX3ChunkHalfs = 512;
constexpr int kX3StageChunks = 64;
constexpr int kX3StageBytes = kX3StageChunks * 1024;
constexpr int kX3StageL0 = 0;      // 8 tiles x 4 k-steps            ->  64 chunks, 1 stage
constexpr int kX3StageL1 = 1;      // mlps.1..4: 8 x 16 -> 256 chunks, 4 stages each
constexpr int kX3StageL5 = 17;     // 8 x (16 + 4) -> 320 chunks, 5 stages
constexpr int kX3StageL6 = 22;     // mlps.6, mlps.7
constexpr int kX3StageRgbF = 30;
constexpr int kX3StageRgbH = 34;   // 4 x (16 + 2) -> 144 chunks, 3 stages (padded)
constexpr int kX3StageInsF = 37;
constexpr int kX3StageInsH = 41;   // 4 x 16 -> 128 chunks, 2 stages
constexpr int kX3StageInsO = 43;   // T x 8 -> 16 T chunks, 1 stage
constexpr int kX3Stages = 44;
DMN_HD constexpr int x3_mlp_stage(int l) {
  return l == 0 ? kX3StageL0 : (l <= 4 ? kX3StageL1 + 4 * (l - 1) : (l == 5 ? kX3StageL5 : kX3StageL6 + 4 * (l - 6)));
}
constexpr long kX3StreamHalfs = (long)kX3Stages * kX3StageChunks * kX3ChunkHalfs;
constexpr long kX3BlobBytes = (long)kConstFloats * 4 + kX3StreamHalfs * 2;
constexpr long kX3IndexLen = (long)kConstFloats + kX3StreamHalfs;   // one gather entry per float / half

// =====================================================================================
// Full-precision bf16 ("x6") blob: same const section (fp32); every weight as THREE bf16 planes
//   w = hi + mid + lo exactly   (hi = bf16(w), mid = bf16(w - hi), lo = w - hi - mid: 8 + 8 + 8 significand bits and
//   the signs of the residuals cover fp32's 24),
// for the 6-MFMA product  W*X = Whi*Xhi + Whi*Xmid + Wmid*Xhi + Wmid*Xmid + Whi*Xlo + Wlo*Xhi  on
// v_mfma_f32_32x32x16_bf16 with X split the same way in the kernel (the three dropped terms are each below 2^-24 of the
// product).  field_fwd_x6.hip walks k-steps in the OUTER loop (all output tiles of a layer accumulate at once, as the
// exact-fp32 kernel does), so per layer: for k-step ks: for tile t: [hi, mid, lo] chunks;
// chunk[lane][j] = W_plane[32*t + (lane&31)][col(8*ks + j, lane>>5)], 64 lanes x 8 bf16 = 1 KiB (the x3 chunk).
// 48 chunks = one 48 KiB stage = 2 k-steps of an 8-tile layer / 4 k-steps of a 4-tile layer / 8 or 4 k-steps of
// ins_linear's T tiles; every layer is padded to whole stages.
// =====================================================================================
constexpr int kX6StageChunks = 48;
constexpr int kX6StageBytes = kX6StageChunks * 1024;
DMN_HD constexpr int x6_ksteps_per_stage(int n_tiles) { return kX6StageChunks / (3 * n_tiles); }   // 8 tiles: 2, 4: 4
DMN_HD constexpr int x6_inso_ksteps_per_stage(int T) { return T <= 2 ? 8 : 4; }
DMN_HD constexpr int x6_inso_stages(int T) { return 8 / x6_inso_ksteps_per_stage(T); }
constexpr int kX6StageL0 = 0;      // 4 k-steps x 8 tiles              -> 2 stages
constexpr int kX6StageL1 = 2;      // mlps.1..4: 16 k-steps            -> 8 stages each
constexpr int kX6StageL5 = 34;     // 16 + 4 k-steps                   -> 10 stages
constexpr int kX6StageL6 = 44;     // mlps.6, mlps.7
constexpr int kX6StageRgbF = 60;   // rgb_feature_linear               8
constexpr int kX6StageRgbH = 68;   // 4 tiles x (16 + 2) k-steps       -> 5 stages (the last one half used)
constexpr int kX6StageInsF = 73;   // ins_feature_linear               8
constexpr int kX6StageInsH = 81;   // 4 tiles x 16 k-steps             -> 4 stages
constexpr int kX6StageInsO = 85;   // T tiles x 8 k-steps              -> 1 or 2 stages
DMN_HD constexpr int x6_total_stages(int T) { return kX6StageInsO + x6_inso_stages(T); }
DMN_HD constexpr int x6_mlp_stage(int l) {
  return l == 0 ? kX6StageL0 : (l <= 4 ? kX6StageL1 + 8 * (l - 1) : (l == 5 ? kX6StageL5 : kX6StageL6 + 8 * (l - 6)));
}
DMN_HD constexpr long x6_stream_halfs(int T) { return (long)x6_total_stages(T) * kX6StageChunks * kX3ChunkHalfs; }
DMN_HD constexpr long x6_blob_bytes(int ins_num) { return (long)kConstFloats * 4 + x6_stream_halfs(ins_tiles(ins_num)) * 2; }
DMN_HD constexpr long x6_index_len(int ins_num) { return (long)kConstFloats + x6_stream_halfs(ins_tiles(ins_num)); }

// =====================================================================================
// "x6w": the full-precision bf16 arithmetic of "x6" on v_mfma_f32_16x16x32_bf16 (inference kernel, field_fwd_x6w.hip).
// Same bytes, same stage map (kX6Stage*, x6_total_stages), same [hi, mid, lo] chunk triples per (k-step, tile) group
// -- but a tile is 16 features x 16 points and a k-step covers 32 k-values:
//   lane l: n = l & 15 (point of a 16-point tile; a wave's 32 points are point tiles p = 0, 1), q = l >> 4
//   accumulator register r (0..3) of feature tile t holds feature 16 t + 4 q + r        (C/D map of the instruction)
//   A chunk (k-step ks, feature tile t): [lane][j] = W[16 t + (lane & 15)][col(ks, lane >> 4, j)], 64 lanes x 8 halves
//   B operand: lane group q supplies k-slots 8 q .. 8 q + 7 of the k-step.  Read out of the accumulators of the layer
//   before, the eight slots are the four registers of feature tiles 2 ks and 2 ks + 1: slot j is feature
//   x6w_feat(ks, q, j); encodings are laid out directly: slot j of group q in k-step e is channel 32 e + 8 q + j.
// The const section has the SAME offsets (kConstBias8 ...) but natural feature order: [kConstBias8 + s*256 + f],
// [kConstBias4 + s*128 + f], [kConstBiasO + f], [kConstDensW + f], [kConstRgbW + c*128 + f].
// 16 groups per 48-chunk stage: a layer of n feature tiles fits 16 / n k-steps into a stage.
// =====================================================================================
DMN_HD constexpr int x6w_feat(int ks, int q, int j) { return 32 * ks + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4)); }
DMN_HD constexpr int x6w_kps(int n_tiles) { return 16 / n_tiles; }                 // k-steps per stage (n_tiles <= 16)
// feature tiles of the ins_linear layer: the kernel is instantiated for 1, 2, 4, 6, 8 tiles of 16 channels
DMN_HD constexpr int x6w_ins_tiles(int ins_num) {
  const int need = (ins_num + 1 + 15) / 16;
  return need <= 2 ? need : (need <= 4 ? 4 : (need <= 6 ? 6 : 8));
}

// =====================================================================================
// Training (backward of the field; train_dmsr.py:32-64 drives it through autograd).
//
// Per-point state crosses HBM as "register images": what a wave of the field kernels holds for its 32
// points, register by register -- image[reg i][lane] (64 lanes x 4 B = one 256-byte row), lane = point +
// 32*half, and register i of half h is feature feat_of(i, h) (pos_slot / dir_slot for the encodings).
// Seen as a matrix an image is [2*regs rows][32 points] with row r = 2*i + h.  Every store / load of the
// producing kernels is one fully coalesced 256-byte wave access; the weight-gradient kernel reads rows.
//
// Save record of the forward (per 32-point tile, floats = kSaveRegs * 64):
constexpr int kSavePe = 0;          // 32 regs: position encoding (GEMM input of mlps.0 and of the skip)
constexpr int kSaveQe = 32;         // 16 regs: view encoding
constexpr int kSaveH = 48;          // h_1 .. h_8 (post-ReLU trunk activations), 128 regs each
constexpr int kSaveFeat = kSaveH + 8 * 128;      // rgb_feature_linear output (no activation), 128
constexpr int kSaveR1 = kSaveFeat + 128;         // rgb_feature_linears.0 output (post-ReLU), 64
constexpr int kSaveInsF = kSaveR1 + 64;          // ins_feature_linear output (no activation), 128
constexpr int kSaveI1 = kSaveInsF + 128;         // ins_feature_linears.0 output (post-ReLU), 64
constexpr int kSaveMask = kSaveI1 + 64;          // ReLU masks as bits: 4 regs per trunk layer (bit b of reg w
                                                 // = register 32*w + b is positive), then 2 for r1, 2 for i1
constexpr int kSaveRegs = kSaveMask + 8 * 4 + 2 + 2;          // 1492
constexpr long kSaveTileFloats = (long)kSaveRegs * 64;
DMN_HD constexpr int save_h(int l) { return kSaveH + 128 * (l - 1); }          // h_l, l = 1..8
DMN_HD constexpr int save_mask(int l) { return kSaveMask + 4 * (l - 1); }      // mask of h_l
constexpr int kSaveMaskR1 = kSaveMask + 32;
constexpr int kSaveMaskI1 = kSaveMask + 34;

// Gradient record written by the backward-data kernel (dL/d pre-activation of every layer), per tile:
constexpr int kGradA = 0;           // G_0 .. G_7 (trunk), 128 regs each
constexpr int kGradFeat = 8 * 128;  // dL/d rgb feature, 128
constexpr int kGradR1 = kGradFeat + 128;         // 64
constexpr int kGradInsF = kGradR1 + 64;          // 128
constexpr int kGradI1 = kGradInsF + 128;         // 64
constexpr int kGradHead = kGradI1 + 64;          // 16*T regs: dL/d ins logits (register 16t + r <-> channel
                                                 // feat_of(16t + r, h)), then one 16-register tile whose
                                                 // registers 0, 1 hold (g_rgb0 | g_rgb1) and (g_rgb2 | g_sigma)
DMN_HD constexpr int grad_regs(int T) { return kGradHead + 16 * (T + 1); }
DMN_HD constexpr long grad_tile_floats(int T) { return (long)grad_regs(T) * 64; }

// Backward-data weight stream (exact fp32, same chunk / stage geometry as the forward stream, weights
// transposed: a layer computes  G_in^T = W^T * G_out^T).  Const section = the forward's (the two VALU heads).
DMN_HD constexpr int bwd_inso_stages(int T) { return T <= 2 ? 1 : 2; }   // W_io^T: 4 tiles x 16T k-steps, 1 chunk each
DMN_HD constexpr int bwd_stage_insh(int T) { return bwd_inso_stages(T); }            // W_i1^T: 8 tiles x 64 k-steps -> 4
DMN_HD constexpr int bwd_stage_rgbh(int T) { return bwd_stage_insh(T) + 4; }         // W_r1^T[:256]: 4 stages
DMN_HD constexpr int bwd_stage_rgbf(int T) { return bwd_stage_rgbh(T) + 4; }         // W_rf^T: 8 stages
DMN_HD constexpr int bwd_stage_mlp(int T, int l) { return bwd_stage_rgbf(T) + 8 + 8 * (7 - l); }   // l = 7..1
DMN_HD constexpr int bwd_total_stages(int T) { return bwd_stage_rgbf(T) + 8 + 7 * 8; }
DMN_HD constexpr long bwd_blob_floats(int ins_num) {
  return (long)kConstFloats + (long)bwd_total_stages(ins_tiles(ins_num)) * kStageFloats;
}

// Backward-data stream in the split-precision format (64-chunk stages, tiles outer, [hi, lo] chunk pairs):
constexpr int kBwdX3StageInsO = 0;     // ins_linear^T:            4 tiles x 2T k-steps -> 1 stage
constexpr int kBwdX3StageInsH = 1;     // ins_feature_linears.0^T: 8 x 8  -> 128 chunks, 2 stages
constexpr int kBwdX3StageRgbH = 3;     // rgb_feature_linears.0^T (feature columns): 8 x 8 -> 2 stages
constexpr int kBwdX3StageRgbF = 5;     // rgb_feature_linear^T:    8 x 16 -> 4 stages
DMN_HD constexpr int bwd_x3_stage_mlp(int l) { return kBwdX3StageRgbF + 4 + 4 * (7 - l); }   // l = 7..1
constexpr int kBwdX3Stages = kBwdX3StageRgbF + 4 + 7 * 4;   // 37
constexpr long kBwdX3StreamHalfs = (long)kBwdX3Stages * kX3StageChunks * kX3ChunkHalfs;
constexpr long kBwdX3BlobBytes = (long)kConstFloats * 4 + kBwdX3StreamHalfs * 2;
constexpr long kBwdX3IndexLen = (long)kConstFloats + kBwdX3StreamHalfs;
// Backward-data stream in the full-precision bf16 format ("x6": 48-chunk stages, k-steps outer, [hi, mid, lo] chunk
// triples per (k-step, tile) group; field_bwd_x6.hip):
DMN_HD constexpr int bwd_x6_inso_stages(int T) { return (2 * T + x6_ksteps_per_stage(4) - 1) / x6_ksteps_per_stage(4); }   // 4 tiles x 2T k-steps
DMN_HD constexpr int bwd_x6_stage_insh(int T) { return bwd_x6_inso_stages(T); }      // ins_feature_linears.0^T: 8 x 8  -> 4 stages
DMN_HD constexpr int bwd_x6_stage_rgbh(int T) { return bwd_x6_stage_insh(T) + 4; }   // rgb_feature_linears.0^T:  8 x 8  -> 4
DMN_HD constexpr int bwd_x6_stage_rgbf(int T) { return bwd_x6_stage_rgbh(T) + 4; }   // rgb_feature_linear^T:     8 x 16 -> 8
DMN_HD constexpr int bwd_x6_stage_mlp(int T, int l) { return bwd_x6_stage_rgbf(T) + 8 + 8 * (7 - l); }   // l = 7..1
DMN_HD constexpr int bwd_x6_total_stages(int T) { return bwd_x6_stage_rgbf(T) + 8 + 7 * 8; }             // 73 (T <= 2) / 74
DMN_HD constexpr long bwd_x6_stream_halfs(int T) { return (long)bwd_x6_total_stages(T) * kX6StageChunks * kX3ChunkHalfs; }
DMN_HD constexpr long bwd_x6_blob_bytes(int ins_num) { return (long)kConstFloats * 4 + bwd_x6_stream_halfs(ins_tiles(ins_num)) * 2; }
DMN_HD constexpr long bwd_x6_index_len(int ins_num) { return (long)kConstFloats + bwd_x6_stream_halfs(ins_tiles(ins_num)); }
constexpr int kConstZeros = 3592;      // const floats 3592..3839 are zero padding: initial accumulators of the backward

// Weight-gradient GEMMs: dW[o][k] = sum over points of G[o][point] * X[k][point], G rows from the gradient
// record, X rows from the save record (or, for the heads, G from the head tile).  Shapes in 32-row tiles.
struct WGemm {
  int g_reg, g_tiles;      // gradient-record register offset, row tiles (rows / 32 = regs / 16)
  int x_reg, x_tiles;      // save-record register offset, row tiles
  int index_off;           // offset of this GEMM's [g_tiles*32][x_tiles*32] scatter table (flat-parameter index or -1)
  int bias_off;            // offset of its [g_tiles*32] bias scatter table, or -1
};
constexpr int kWGemms = 17;
DMN_HD constexpr bool wgemm_big(const WGemm& g) { return g.g_tiles == 8 && g.x_tiles == 8; }
constexpr long kWgradBigLen = 9L * (256 * 256 + 256);      // the nine 256x256 GEMMs and their biases come first
constexpr int kWgradSmallSplitFactor = 3;                  // the small GEMMs are cut into 3x as many point slices
// The list (index_off / bias_off are running sums in list order; bias only where a layer's G appears first).
inline void wgemm_table(int T, WGemm* g) {
  int n = 0;
  auto add = [&](int g_reg, int g_tiles, int x_reg, int x_tiles, bool bias) {
    g[n++] = WGemm{g_reg, g_tiles, x_reg, x_tiles, 0, bias ? 0 : -1};
  };
  add(kGradA, 8, kSavePe, 2, true);                                        // 0  mlps.0
  for (int l = 1; l <= 4; ++l) add(kGradA + 128 * l, 8, save_h(l), 8, true);   // 1-4
  add(kGradA + 128 * 5, 8, save_h(5), 8, true);                            // 5  mlps.5, columns of h
  add(kGradA + 128 * 5, 8, kSavePe, 2, false);                             // 6  mlps.5, columns of input_pts
  add(kGradA + 128 * 6, 8, save_h(6), 8, true);                            // 7
  add(kGradA + 128 * 7, 8, save_h(7), 8, true);                            // 8
  add(kGradFeat, 8, save_h(8), 8, true);                                   // 9  rgb_feature_linear
  add(kGradInsF, 8, save_h(8), 8, true);                                   // 10 ins_feature_linear
  add(kGradR1, 4, kSaveFeat, 8, true);                                     // 11 rgb_feature_linears.0, feature columns
  add(kGradR1, 4, kSaveQe, 1, false);                                      // 12 rgb_feature_linears.0, view columns
  add(kGradI1, 4, kSaveInsF, 8, true);                                     // 13 ins_feature_linears.0
  add(kGradHead, T, kSaveI1, 4, true);                                     // 14 ins_linear
  add(kGradHead + 16 * T, 1, kSaveR1, 4, true);                            // 15 rgb_linear (rows rgb0..2)
  add(kGradHead + 16 * T, 1, save_h(8), 8, true);                          // 16 density_linear (row sigma)
  // scatter-table / partial-sum element space: [tables of the 8x8 GEMMs][their biases][other tables][other biases];
  // the two regions are reduced over different numbers of point slices (wgrad_big_len)
  int off = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < kWGemms; ++i)
      if (wgemm_big(g[i]) == (pass == 0)) { g[i].index_off = off; off += g[i].g_tiles * 32 * g[i].x_tiles * 32; }
    for (int i = 0; i < kWGemms; ++i)
      if (wgemm_big(g[i]) == (pass == 0) && g[i].bias_off == 0) { g[i].bias_off = off; off += g[i].g_tiles * 32; }
  }
}
inline long wgrad_index_len(int T) {
  WGemm g[kWGemms];
  wgemm_table(T, g);
  long n = 0;
  for (int i = 0; i < kWGemms; ++i) {
    n += (long)g[i].g_tiles * 32 * g[i].x_tiles * 32;
    if (g[i].bias_off >= 0) n += g[i].g_tiles * 32;
  }
  return n;
}

}  // namespace dmn
