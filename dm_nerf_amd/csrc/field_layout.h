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

}  // namespace dmn
