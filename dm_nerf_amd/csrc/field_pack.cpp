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
  // transposed variants for the backward-data stream: A[row][k] = W[kf][row], kf = feature held by k-step j
  template <class Col>
  void kstep8T(int stage0, int j, int w_off, int fan_in, Col kfeat) {
    for (int q = 0; q < 2; ++q) {
      int32_t* c = stream + ((long)stage0 * kStageChunks + 2 * j + q) * kChunkFloats;
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
          int row = 32 * (4 * q + e) + (lane & 31);
          int kf = kfeat(lane >> 5);
          c[lane * 4 + e] = kf < 0 ? -1 : w_off + kf * fan_in + row;
        }
    }
  }
  template <class Col>
  void kstep4T(int stage0, int j, int w_off, int fan_in, Col kfeat) {
    int32_t* c = stream + ((long)stage0 * kStageChunks + j) * kChunkFloats;
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 4; ++e) {
        int row = 32 * e + (lane & 31);
        int kf = kfeat(lane >> 5);
        c[lane * 4 + e] = kf < 0 ? -1 : w_off + kf * fan_in + row;
      }
  }
};

// ---- split-precision stream (v_mfma_f32_32x32x16_{f16,bf16}) ---------------------------------
// One layer: tiles in the outer loop, k-steps inner, [hi, lo] chunk pair per (tile, k-step).
// col(i, h): input column held by element i = 8*ks + j of lane-half h, or -1.
// transposed: the layer multiplies by W^T (backward data): row indexes W's columns, the k feature its rows.
template <class Col>
void layer_x3(int32_t* stream, int stage0, int n_tiles, int n_ksteps, int w_off, int fan_in, int n_rows, Col col,
              bool transposed = false) {
  for (int t = 0; t < n_tiles; ++t)
    for (int ks = 0; ks < n_ksteps; ++ks)
      for (int part = 0; part < 2; ++part) {
        long chunk = (long)stage0 * kX3StageChunks + (long)(t * n_ksteps + ks) * 2 + part;
        int32_t* c = stream + chunk * kX3ChunkHalfs;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int row = 32 * t + (lane & 31);
            int k = col(8 * ks + j, lane >> 5);
            c[lane * 8 + j] = (k < 0 || row >= n_rows) ? -1 : (transposed ? w_off + k * fan_in + row : w_off + row * fan_in + k);
          }
      }
}

// ---- full-precision bf16 stream ("x6": three bf16 planes per weight, k-steps in the outer loop) ----------------
// One layer: for k-step ks, for tile t: [hi, mid, lo] chunks; kps k-steps per 48-chunk stage (field_layout.h).
// All three chunks of a (k-step, tile) carry the same gather entries: the plane is the chunk's position modulo 3.
// transposed: the layer multiplies by W^T (backward data): row indexes W's columns, the k feature its rows.
template <class Col>
void layer_x6(int32_t* stream, int stage0, int n_tiles, int n_ksteps, int kps, int w_off, int fan_in, int n_rows, Col col,
              bool transposed = false) {
  for (int ks = 0; ks < n_ksteps; ++ks)
    for (int t = 0; t < n_tiles; ++t)
      for (int plane = 0; plane < 3; ++plane) {
        long chunk = (long)(stage0 + ks / kps) * kX6StageChunks + (long)((ks % kps) * n_tiles + t) * 3 + plane;
        int32_t* c = stream + chunk * kX3ChunkHalfs;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int row = 32 * t + (lane & 31);
            int k = col(8 * ks + j, lane >> 5);
            c[lane * 8 + j] = (k < 0 || row >= n_rows) ? -1 : (transposed ? w_off + k * fan_in + row : w_off + row * fan_in + k);
          }
      }
}

// ---- "x6w" stream: the x6 planes for v_mfma_f32_16x16x32_bf16 (field_layout.h "x6w") ---------------------------------
// col(ks, q, j): input column of k-slot j of lane group q in k-step ks, or -1.
template <class Col>
void layer_x6w(int32_t* stream, int stage0, int n_tiles, int n_ksteps, int w_off, int fan_in, int n_rows, Col col) {
  const int kps = x6w_kps(n_tiles);
  for (int ks = 0; ks < n_ksteps; ++ks)
    for (int t = 0; t < n_tiles; ++t)
      for (int plane = 0; plane < 3; ++plane) {
        long chunk = (long)(stage0 + ks / kps) * kX6StageChunks + (long)((ks % kps) * n_tiles + t) * 3 + plane;
        int32_t* c = stream + chunk * kX3ChunkHalfs;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            int row = 16 * t + (lane & 15);
            int k = col(ks, lane >> 4, j);
            c[lane * 8 + j] = (k < 0 || row >= n_rows) ? -1 : w_off + row * fan_in + k;
          }
      }
}

// const section in natural feature order
void fill_const_x6w(int32_t* cst, const ParamOffsets& P, int ins_num) {
  for (int slot = 0; slot < 10; ++slot) {
    int bias = slot < 8 ? P.mlp_b[slot] : (slot == 8 ? P.rgbf_b : P.insf_b);
    for (int f = 0; f < 256; ++f) cst[kConstBias8 + slot * 256 + f] = bias + f;
  }
  for (int f = 0; f < 128; ++f) {
    cst[kConstBias4 + f] = P.rgbh_b + f;
    cst[kConstBias4 + 128 + f] = P.insh_b + f;
    cst[kConstBiasO + f] = f <= ins_num ? P.inso_b + f : -1;
  }
  for (int f = 0; f < 256; ++f) cst[kConstDensW + f] = P.dens_w + f;
  cst[kConstDensB] = P.dens_b;
  for (int c = 0; c < 3; ++c) {
    for (int f = 0; f < 128; ++f) cst[kConstRgbW + c * 128 + f] = P.rgbo_w + c * kHalf + f;
    cst[kConstRgbB + c] = P.rgbo_b + c;
  }
}

}  // namespace

extern "C" int dmn_field_pack_index_x6w(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T16 = x6w_ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  std::fill(index_out, index_out + x6_index_len(ins_num), -1);
  fill_const_x6w(index_out, P, ins_num);
  int32_t* s = index_out + kConstFloats;
  auto act = [](int ks, int q, int j) { return x6w_feat(ks, q, j); };
  auto enc = [](int e, int q, int j, int channels) { int c = 32 * e + 8 * q + j; return c < channels ? c : -1; };

  layer_x6w(s, kX6StageL0, 16, 2, P.mlp_w[0], kPosCh, kWidth, [&](int ks, int q, int j) { return enc(ks, q, j, kPosCh); });
  for (int l = 1; l < kDepth; ++l) {
    if (l == kSkipLayer)                                           // cat([h, input_pts]) (dm_nerf.py:87)
      layer_x6w(s, x6_mlp_stage(l), 16, 10, P.mlp_w[l], mlp_fan_in(l), kWidth, [&](int ks, int q, int j) {
        if (ks < 8) return x6w_feat(ks, q, j);
        int c = enc(ks - 8, q, j, kPosCh);
        return c < 0 ? -1 : kWidth + c;
      });
    else
      layer_x6w(s, x6_mlp_stage(l), 16, 8, P.mlp_w[l], kWidth, kWidth, act);
  }
  layer_x6w(s, kX6StageRgbF, 16, 8, P.rgbf_w, kWidth, kWidth, act);
  layer_x6w(s, kX6StageRgbH, 8, 9, P.rgbh_w, kWidth + kDirCh, kHalf, [&](int ks, int q, int j) {   // cat([feature, dirs])
    if (ks < 8) return x6w_feat(ks, q, j);
    int c = enc(0, q, j, kDirCh);
    return c < 0 ? -1 : kWidth + c;
  });
  layer_x6w(s, kX6StageInsF, 16, 8, P.insf_w, kWidth, kWidth, act);
  layer_x6w(s, kX6StageInsH, 8, 8, P.insh_w, kWidth, kHalf, act);
  layer_x6w(s, kX6StageInsO, T16, 4, P.inso_w, kHalf, ins_num + 1, act);
  return DMN_OK;
}

extern "C" long dmn_field_blob_x6_bytes(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return x6_blob_bytes(ins_num);
}

extern "C" long dmn_field_pack_index_x6_len(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return x6_index_len(ins_num);
}

extern "C" int dmn_field_pack_index_x6(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  std::fill(index_out, index_out + x6_index_len(ins_num), -1);
  fill_const(index_out, P, ins_num);
  int32_t* s = index_out + kConstFloats;
  auto act = [](int i, int h) { return feat_of(i, h); };
  const int k8 = x6_ksteps_per_stage(8), k4 = x6_ksteps_per_stage(4);

  layer_x6(s, kX6StageL0, 8, 4, k8, P.mlp_w[0], kPosCh, kWidth, [](int i, int h) { return pos_slot(i, h); });
  for (int l = 1; l < kDepth; ++l) {
    if (l == kSkipLayer)                                           // cat([h, input_pts]) (dm_nerf.py:87)
      layer_x6(s, x6_mlp_stage(l), 8, 20, k8, P.mlp_w[l], mlp_fan_in(l), kWidth, [](int i, int h) {
        if (i < 128) return feat_of(i, h);
        int p = pos_slot(i - 128, h);
        return p < 0 ? -1 : kWidth + p;
      });
    else
      layer_x6(s, x6_mlp_stage(l), 8, 16, k8, P.mlp_w[l], kWidth, kWidth, act);
  }
  layer_x6(s, kX6StageRgbF, 8, 16, k8, P.rgbf_w, kWidth, kWidth, act);
  layer_x6(s, kX6StageRgbH, 4, 18, k4, P.rgbh_w, kWidth + kDirCh, kHalf, [](int i, int h) {   // cat([feature, dirs])
    if (i < 128) return feat_of(i, h);
    int p = dir_slot(i - 128, h);
    return p < 0 ? -1 : kWidth + p;
  });
  layer_x6(s, kX6StageInsF, 8, 16, k8, P.insf_w, kWidth, kWidth, act);
  layer_x6(s, kX6StageInsH, 4, 16, k4, P.insh_w, kWidth, kHalf, act);
  layer_x6(s, kX6StageInsO, T, 8, x6_inso_ksteps_per_stage(T), P.inso_w, kHalf, ins_num + 1, act);
  return DMN_OK;
}

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

// ---- training: backward-data stream and weight-gradient scatter tables -------------------------------
extern "C" long dmn_field_bwd_blob_floats(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return bwd_blob_floats(ins_num);
}

extern "C" int dmn_field_bwd_pack_index(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  const long n = bwd_blob_floats(ins_num);
  std::fill(index_out, index_out + n, -1);
  fill_const(index_out, P, ins_num);
  Builder b{index_out + kConstFloats};
  for (int j = 0; j < 16 * T; ++j)                               // ins_linear^T: channels -> 128
    b.kstep4T(0, j, P.inso_w, kHalf, [&](int h) { int ch = feat_of(j, h); return ch <= ins_num ? ch : -1; });
  for (int j = 0; j < 64; ++j) {
    b.kstep8T(bwd_stage_insh(T), j, P.insh_w, kWidth, [&](int h) { return feat_of(j, h); });          // [128 x 256]^T
    b.kstep8T(bwd_stage_rgbh(T), j, P.rgbh_w, kWidth + kDirCh, [&](int h) { return feat_of(j, h); }); // first 256 columns
  }
  for (int j = 0; j < 128; ++j) {
    b.kstep8T(bwd_stage_rgbf(T), j, P.rgbf_w, kWidth, [&](int h) { return feat_of(j, h); });
    for (int l = 7; l >= 1; --l)
      b.kstep8T(bwd_stage_mlp(T, l), j, P.mlp_w[l], mlp_fan_in(l), [&](int h) { return feat_of(j, h); });
  }
  return DMN_OK;
}

extern "C" long dmn_field_bwd_blob_x3_bytes(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return kBwdX3BlobBytes;
}

extern "C" long dmn_field_bwd_pack_index_x3_len(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return kBwdX3IndexLen;
}

extern "C" int dmn_field_bwd_pack_index_x3(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  std::fill(index_out, index_out + kBwdX3IndexLen, -1);
  fill_const(index_out, P, ins_num);
  int32_t* s = index_out + kConstFloats;
  auto act = [](int i, int h) { return feat_of(i, h); };
  layer_x3(s, kBwdX3StageInsO, 4, 2 * T, P.inso_w, kHalf, kHalf,
           [&](int i, int h) { int ch = feat_of(i, h); return ch <= ins_num ? ch : -1; }, true);
  layer_x3(s, kBwdX3StageInsH, 8, 8, P.insh_w, kWidth, kWidth, act, true);
  layer_x3(s, kBwdX3StageRgbH, 8, 8, P.rgbh_w, kWidth + kDirCh, kWidth, act, true);
  layer_x3(s, kBwdX3StageRgbF, 8, 16, P.rgbf_w, kWidth, kWidth, act, true);
  for (int l = 7; l >= 1; --l) layer_x3(s, bwd_x3_stage_mlp(l), 8, 16, P.mlp_w[l], mlp_fan_in(l), kWidth, act, true);
  return DMN_OK;
}

extern "C" long dmn_field_bwd_blob_x6_bytes(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return bwd_x6_blob_bytes(ins_num);
}

extern "C" long dmn_field_bwd_pack_index_x6_len(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return bwd_x6_index_len(ins_num);
}

extern "C" int dmn_field_bwd_pack_index_x6(int ins_num, int32_t* index_out) {
  if (!index_out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  std::fill(index_out, index_out + bwd_x6_index_len(ins_num), -1);
  fill_const(index_out, P, ins_num);
  int32_t* s = index_out + kConstFloats;
  auto act = [](int i, int h) { return feat_of(i, h); };
  const int k8 = x6_ksteps_per_stage(8), k4 = x6_ksteps_per_stage(4);
  layer_x6(s, 0, 4, 2 * T, k4, P.inso_w, kHalf, kHalf,
           [&](int i, int h) { int ch = feat_of(i, h); return ch <= ins_num ? ch : -1; }, true);
  layer_x6(s, bwd_x6_stage_insh(T), 8, 8, k8, P.insh_w, kWidth, kWidth, act, true);
  layer_x6(s, bwd_x6_stage_rgbh(T), 8, 8, k8, P.rgbh_w, kWidth + kDirCh, kWidth, act, true);
  layer_x6(s, bwd_x6_stage_rgbf(T), 8, 16, k8, P.rgbf_w, kWidth, kWidth, act, true);
  for (int l = 7; l >= 1; --l) layer_x6(s, bwd_x6_stage_mlp(T, l), 8, 16, k8, P.mlp_w[l], mlp_fan_in(l), kWidth, act, true);
  return DMN_OK;
}

extern "C" long dmn_field_wgrad_index_len(int ins_num) {
  if (!supported(ins_num)) return DMN_ERR_BAD_ARG;
  return wgrad_index_len(ins_tiles(ins_num));
}

// Row r = 2*i + h of a register image -> column / row of the parameter it belongs to
extern "C" int dmn_field_wgrad_index(int ins_num, int32_t* out) {
  if (!out || !supported(ins_num)) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  const ParamOffsets P = param_offsets(ins_num);
  WGemm g[kWGemms];
  wgemm_table(T, g);
  auto act = [](int r) { return feat_of(r >> 1, r & 1); };
  auto pos = [](int r) { return pos_slot(r >> 1, r & 1); };
  auto dir = [](int r) { return dir_slot(r >> 1, r & 1); };
  auto chan = [&](int r) { int c = feat_of(r >> 1, r & 1); return c <= ins_num ? c : -1; };
  auto rgbrow = [](int r) { return r < 3 ? r : -1; };            // head tile rows: rgb0, rgb1, rgb2, sigma, 0...
  auto sigrow = [](int r) { return r == 3 ? 0 : -1; };
  // weight offset, fan-in, row map, column map (+ column base), bias offset
  auto fill = [&](int gi, int w_off, int fan_in, int col0, auto rowmap, auto colmap, int b_off) {
    const WGemm& G = g[gi];
    const int O = G.g_tiles * 32, K = G.x_tiles * 32;
    for (int o = 0; o < O; ++o) {
      const int ro = rowmap(o);
      for (int k = 0; k < K; ++k) {
        const int ck = colmap(k);
        out[G.index_off + o * K + k] = (ro < 0 || ck < 0) ? -1 : w_off + ro * fan_in + col0 + ck;
      }
      if (G.bias_off >= 0) out[G.bias_off + o] = ro < 0 ? -1 : b_off + ro;
    }
  };
  fill(0, P.mlp_w[0], kPosCh, 0, act, pos, P.mlp_b[0]);
  for (int l = 1; l <= 4; ++l) fill(l, P.mlp_w[l], kWidth, 0, act, act, P.mlp_b[l]);
  fill(5, P.mlp_w[5], kWidth + kPosCh, 0, act, act, P.mlp_b[5]);
  fill(6, P.mlp_w[5], kWidth + kPosCh, kWidth, act, pos, -1);
  fill(7, P.mlp_w[6], kWidth, 0, act, act, P.mlp_b[6]);
  fill(8, P.mlp_w[7], kWidth, 0, act, act, P.mlp_b[7]);
  fill(9, P.rgbf_w, kWidth, 0, act, act, P.rgbf_b);
  fill(10, P.insf_w, kWidth, 0, act, act, P.insf_b);
  fill(11, P.rgbh_w, kWidth + kDirCh, 0, act, act, P.rgbh_b);
  fill(12, P.rgbh_w, kWidth + kDirCh, kWidth, act, dir, -1);
  fill(13, P.insh_w, kWidth, 0, act, act, P.insh_b);
  fill(14, P.inso_w, kHalf, 0, chan, act, P.inso_b);
  fill(15, P.rgbo_w, kHalf, 0, rgbrow, act, P.rgbo_b);
  fill(16, P.dens_w, kWidth, 0, sigrow, act, P.dens_b);
  return DMN_OK;
}
