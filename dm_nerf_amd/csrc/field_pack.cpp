// Host side of the packed-weight layouts: builds the gather tables that map every element of a
// kernel blob (const section + weight stream, see field_layout.h) to an index into the flat
// parameter vector (reference state_dict order), or -1 for zero padding.  Pure host code: it is
// exercised on CPU by tests/test_layout.py against lane-exact models of the MFMA instructions.
#include "field_layout.h"
#include "../../include/dmnerf_hip.h"

#include <algorithm>

using namespace dmn;

namespace {

bool supported(int ins_num) { return ins_num >= 0 && ins_tiles(ins_num) <= kMaxInsTiles; }

// biases and the two VALU heads (identical for both blob flavours)
void fill_const(int32_t* cst, const ParamOffsets& P, int ins_num) {
  const int T = ins_tiles(ins_num);
  for (int slot = 0; slot < 10; ++slot) {
    int bias = slot < 8 ? P.mlp_b[slot] : (slot == 8 ? P.rgbf_b : P.insf_b);
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < 128; ++i) cst[kConstBias8 + slot * 256 + h * 128 + i] = bias + feat_of(i, h);
  }
  for (int s = 0; s < 2; ++s) {
    int bias = s == 0 ? P.rgbh_b : P.insh_b;
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < 64; ++i) cst[kConstBias4 + s * 128 + h * 64 + i] = bias + feat_of(i, h);
  }
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < 16 * T; ++i) {
      int f = feat_of(i, h);
      cst[kConstBiasO + h * 64 + i] = f <= ins_num ? P.inso_b + f : -1;
    }
  for (int h = 0; h < 2; ++h)
    for (int i = 0; i < 128; ++i) cst[kConstDensW + h * 128 + i] = P.dens_w + feat_of(i, h);
  cst[kConstDensB] = P.dens_b;
  for (int c = 0; c < 3; ++c) {
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < 64; ++i) cst[kConstRgbW + c * 128 + h * 64 + i] = P.rgbo_w + c * kHalf + feat_of(i, h);
    cst[kConstRgbB + c] = P.rgbo_b + c;
  }
}

// ---- exact-fp32 stream (v_mfma_f32_32x32x2_f32) ---------------------------------------------
struct Builder {
  int32_t* stream;

  // one 8-tile k-step: two chunks
  template <class Col>
  void kstep8(int stage0, int j, int w_off, int fan_in, Col col) {
    for (int q = 0; q < 2; ++q) {
      int32_t* c = stream + ((long)stage0 * kStageChunks + 2 * j + q) * kChunkFloats;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
          int row = 32 * (4 * q + e) + (lane & 31);
          int k = col(lane >> 5);
          c[lane * 4 + e] = k < 0 ? -1 : w_off + row * fan_in + k;
        }
    }
  }
  // one 4-tile k-step: one chunk
  template <class Col>
  void kstep4(int stage0, int j, int w_off, int fan_in, Col col) {
    int32_t* c = stream + ((long)stage0 * kStageChunks + j) * kChunkFloats;
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 4; ++e) {
        int row = 32 * e + (lane & 31);
        int k = col(lane >> 5);
        c[lane * 4 + e] = k < 0 ? -1 : w_off + row * fan_in + k;
      }
  }
};

// ---- split-precision stream (v_mfma_f32_32x32x16_{f16,bf16}) ---------------------------------
// One layer: tiles in the outer loop, k-steps inner, [hi, lo] chunk pair per (tile, k-step).
// col(i, h): input column held by element i = 8*ks + j of lane-half h, or -1.
template <class Col>
void layer_x3(int32_t* stream, int stage0, int n_tiles, int n_ksteps, int w_off, int fan_in, int n_rows, Col col) {
  for (int t = 0; t < n_tiles; ++t)
    for (int ks = 0; ks < n_ksteps; ++ks)
      for (int part = 0; part < 2; ++part) {
        long chunk = (long)stage0 * kX3StageChunks + (long)(t * n_ksteps + ks) * 2 + part;
        int32_t* c = stream + chunk * kX3ChunkHalfs;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int row = 32 * t + (lane & 31);
            int k = col(8 * ks + j, lane >> 5);
            c[lane * 8 + j] = (k < 0 || row >= n_rows) ? -1 : w_off + row * fan_in + k;
          }
      }
}

}  // namespace

extern "C" int dmn_field_param_floats(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return param_offsets(ins_num).total;
}

extern "C" long dmn_field_blob_floats(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return blob_floats(ins_num);
}

extern "C" int dmn_field_pack_index(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  const long n = blob_floats(ins_num);
  std::fill(index_out, index_out + n, -1);
  fill_const(index_out, P, ins_num);
  Builder b{index_out + kConstFloats};

  for (int j = 0; j < 32; ++j)                                   // mlps.0: 63 -> 256
    b.kstep8(kStageL0, j, P.mlp_w[0], kPosCh, [&](int h) { return pos_slot(j, h); });
  for (int l = 1; l < kDepth; ++l) {                             // mlps.1..7
    const int fan = mlp_fan_in(l);
    for (int j = 0; j < 128; ++j)
      b.kstep8(mlp_stage(l), j, P.mlp_w[l], fan, [&](int h) { return feat_of(j, h); });
    if (l == kSkipLayer)                                         // cat([h, input_pts]) (dm_nerf.py:87)
      for (int j = 0; j < 32; ++j)
        b.kstep8(mlp_stage(l), 128 + j, P.mlp_w[l], fan, [&](int h) {
          int s = pos_slot(j, h);
          return s < 0 ? -1 : kWidth + s;
        });
  }
  for (int j = 0; j < 128; ++j) {
    b.kstep8(kStageRgbF, j, P.rgbf_w, kWidth, [&](int h) { return feat_of(j, h); });
    b.kstep8(kStageInsF, j, P.insf_w, kWidth, [&](int h) { return feat_of(j, h); });
    b.kstep4(kStageRgbH, j, P.rgbh_w, kWidth + kDirCh, [&](int h) { return feat_of(j, h); });
    b.kstep4(kStageInsH, j, P.insh_w, kWidth, [&](int h) { return feat_of(j, h); });
  }
  for (int j = 0; j < 16; ++j)                                   // cat([feature, dirs]) (dm_nerf.py:90)
    b.kstep4(kStageRgbH, 128 + j, P.rgbh_w, kWidth + kDirCh, [&](int h) {
      int s = dir_slot(j, h);
      return s < 0 ? -1 : kWidth + s;
    });
  const int JS = inso_groups_per_stage(T);
  for (int jg = 0; jg < 16; ++jg)                                // ins_linear: 128 -> ins_num+1
    for (int t = 0; t < T; ++t) {
      long chunk = (long)(kStageInsO + jg / JS) * kStageChunks + (jg % JS) * T + t;
      int32_t* c = b.stream + chunk * kChunkFloats;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
          int row = 32 * t + (lane & 31);
          c[lane * 4 + e] = row <= ins_num ? P.inso_w + row * kHalf + feat_of(4 * jg + e, lane >> 5) : -1;
        }
    }
  return DMN_OK;
}

extern "C" long dmn_field_blob_x3_bytes(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return kX3BlobBytes;
}

extern "C" long dmn_field_pack_index_x3_len(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return kX3IndexLen;
}

extern "C" int dmn_field_pack_index_x3(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  std::fill(index_out, index_out + kX3IndexLen, -1);
  fill_const(index_out, P, ins_num);
  int32_t* s = index_out + kConstFloats;
  auto act = [](int i, int h) { return feat_of(i, h); };

  layer_x3(s, kX3StageL0, 8, 4, P.mlp_w[0], kPosCh, kWidth, [](int i, int h) { return pos_slot(i, h); });
  for (int l = 1; l < kDepth; ++l) {
    if (l == kSkipLayer)
      layer_x3(s, x3_mlp_stage(l), 8, 20, P.mlp_w[l], mlp_fan_in(l), kWidth, [](int i, int h) {
        if (i < 128) return feat_of(i, h);
        int p = pos_slot(i - 128, h);
        return p < 0 ? -1 : kWidth + p;
      });
    else
      layer_x3(s, x3_mlp_stage(l), 8, 16, P.mlp_w[l], kWidth, kWidth, act);
  }
  layer_x3(s, kX3StageRgbF, 8, 16, P.rgbf_w, kWidth, kWidth, act);
  layer_x3(s, kX3StageRgbH, 4, 18, P.rgbh_w, kWidth + kDirCh, kHalf, [](int i, int h) {
    if (i < 128) return feat_of(i, h);
    int p = dir_slot(i - 128, h);
    return p < 0 ? -1 : kWidth + p;
  });
  layer_x3(s, kX3StageInsF, 8, 16, P.insf_w, kWidth, kWidth, act);
  layer_x3(s, kX3StageInsH, 4, 16, P.insh_w, kWidth, kHalf, act);
  layer_x3(s, kX3StageInsO, T, 8, P.inso_w, kHalf, ins_num + 1, act);
  return DMN_OK;
}
