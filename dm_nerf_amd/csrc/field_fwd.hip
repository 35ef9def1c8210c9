// Fused field evaluation for gfx950 (MI355X): sample points -> positional / view encodings ->
// the whole DM-NeRF network (8x256 trunk with skip, density head, view-dependent colour branch,
// instance-code branch) in ONE kernel, on v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma
// chain).  Replaces networks/render.py:37-38,49-61 (+ :71-83) and networks/dm_nerf.py:80-106 of
// the reference; no intermediate ([M,90] encodings, [M,256] activations) ever touches HBM.
//
// Mapping (see field_layout.h for the packed-weight side):
//   * workgroup = 4 waves = 128 consecutive sample points; wave = 32 points; one wave per SIMD
//     (the kernel uses the whole 512-entry register file: 128 activation + 128 accumulator
//     registers per lane plus the encodings).
//   * activations live in registers as H^T[feature][point] in the MFMA C/D layout and are fed
//     back as the B operand of the next layer; weights are the A operand.
//   * weights stream L2 -> LDS in 32 KiB stages by LDS-DMA (global_load_lds_dwordx4), double
//     buffered, one barrier per stage (= per 128 MFMAs/wave, 8192 MFMA cycles); the 4 waves
//     share each stage.  Algorithmic intensity vs L2 is ~64 flop/B per workgroup.
//   * biases are the initial accumulator value; ReLU is applied when an accumulator becomes the
//     next B operand; density (256->1) and colour (128->3) heads are VALU dot products.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_common.h"
#ifndef DMN_F32_SPREAD
#define DMN_F32_SPREAD 1
#endif
#include "field_stream.h"

using namespace dmn;

namespace {

// training: store N post-ReLU registers as an image plus their sign bits (bit b of word w <-> register 32w + b)
template <int N>
__device__ __forceinline__ void save_relu(float* img, float* msk, int lane, const float* v) {
#pragma unroll
  for (int i = 0; i < N; ++i) image_store(img, i, lane, v[i]);
#pragma unroll
  for (int w = 0; w < N / 32; ++w) {
    unsigned bits = 0;
#pragma unroll
    for (int b = 0; b < 32; ++b) bits |= (v[32 * w + b] > 0.f ? 1u : 0u) << b;
    image_store(msk, w, lane, __uint_as_float(bits));
  }
}

// SAVE: also write the save record the backward kernels read (field_layout.h "Training")
template <int T, bool SAVE>
__global__ __launch_bounds__(256, 1) void field_fwd_kernel(const FieldArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  float* const cst = lds + 2 * kStageFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* const blob = static_cast<const float*>(a.blob);
  const float* const stream = blob + kConstFloats;
  constexpr int G = total_stages(T);

  // const section + stage 0 start streaming while the encodings are computed
  for (int c = wave_s; c < kConstChunks; c += 4)
    glds16(blob + c * kChunkFloats + lane * 4, cst + c * kChunkFloats);
  load_stage(stream, 0, lds, wave_s, lane);

  const long m = ((long)blockIdx.x * 4 + wave_s) * 32 + p;
  const bool valid = m < a.M;
  const long mc = valid ? m : a.M - 1;

  float pe[32], qe[16];
  point_encodings(a, mc, h, pe, qe);
  float* const rec = SAVE ? a.save + ((long)blockIdx.x * 4 + wave_s) * kSaveTileFloats : nullptr;
  if (SAVE) {
#pragma unroll
    for (int j = 0; j < 32; ++j) image_store(rec, kSavePe + j, lane, pe[j]);
#pragma unroll
    for (int j = 0; j < 16; ++j) image_store(rec, kSaveQe + j, lane, qe[j]);
  }

  f32x16 acc[8];
  float act[128];
  int g = 0;
  const float* buf;

  // ---- mlps.0: 63 -> 256 (32 k-steps from the position encoding)
  buf = stage_begin(stream, g, G, lds, wave_s, lane);
  bias_tiles8(acc, cst + kConstBias8 + h * 128);
  DMN_STAGE8(buf, SRC_PE, 0)
  ++g;
  buf = stage_begin(stream, g, G, lds, wave_s, lane);
  DMN_STAGE8(buf, SRC_PE, 16)
  ++g;
#pragma unroll
  for (int i = 0; i < 128; ++i) act[i] = fmaxf(acc[i >> 4][i & 15], 0.f);
  if (SAVE) save_relu<128>(rec + save_h(1) * 64, rec + save_mask(1) * 64, lane, act);

  // ---- mlps.1 .. mlps.7 (mlps.5 also consumes the position encoding: skip connection)
#pragma unroll 1
  for (int l = 1; l < kDepth; ++l) {
    bias_tiles8(acc, cst + kConstBias8 + l * 256 + h * 128);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      buf = stage_begin(stream, g, G, lds, wave_s, lane);
      DMN_STAGE8(buf, SRC_ACT, 16 * s)
      ++g;
    }
    if (l == kSkipLayer) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        buf = stage_begin(stream, g, G, lds, wave_s, lane);
        DMN_STAGE8(buf, SRC_PE, 16 * s)
        ++g;
      }
    }
#pragma unroll
    for (int i = 0; i < 128; ++i) act[i] = fmaxf(acc[i >> 4][i & 15], 0.f);
    if (SAVE) save_relu<128>(rec + (save_h(1) + 128 * l) * 64, rec + (save_mask(1) + 4 * l) * 64, lane, act);
  }

  // ---- density_linear: 256 -> 1 on the VALU (dm_nerf.py:101)
  float sigma = 0.f;
  {
    const float* w = cst + kConstDensW + h * 128;
#pragma unroll
    for (int i = 0; i < 128; ++i) sigma = fmaf(act[i], w[i], sigma);
    sigma += __shfl_xor(sigma, 32);
    sigma += cst[kConstDensB];
  }

  // ---- rgb_feature_linear: 256 -> 256, no activation (dm_nerf.py:89)
  bias_tiles8(acc, cst + kConstBias8 + 8 * 256 + h * 128);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE8(buf, SRC_ACT, 16 * s)
    ++g;
  }
  if (SAVE) {
#pragma unroll
    for (int i = 0; i < 128; ++i) image_store(rec, kSaveFeat + i, lane, acc[i >> 4][i & 15]);
  }
  // ---- rgb_feature_linears.0: cat([feature, dirs]) 283 -> 128, ReLU (dm_nerf.py:90-93)
  f32x16 acc2[4];
  bias_tiles<4>(acc2, cst + kConstBias4 + h * 64);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE4(buf, SRC_ACC, 32 * s, 32)
    ++g;
  }
  buf = stage_begin(stream, g, G, lds, wave_s, lane);
  DMN_STAGE4(buf, SRC_QE, 0, 16)
  ++g;
  if (SAVE) {
    float r1[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) r1[i] = fmaxf(acc2[i >> 4][i & 15], 0.f);
    save_relu<64>(rec + kSaveR1 * 64, rec + kSaveMaskR1 * 64, lane, r1);
  }
  // ---- rgb_linear: 128 -> 3 on the VALU (dm_nerf.py:102)
  float rgb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* w = cst + kConstRgbW + c * 128 + h * 64;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) v = fmaf(fmaxf(acc2[i >> 4][i & 15], 0.f), w[i], v);
    v += __shfl_xor(v, 32);
    rgb[c] = v + cst[kConstRgbB + c];
  }

  // ---- ins_feature_linear on the same trunk output (h.detach(), dm_nerf.py:95-96), no activation
  bias_tiles8(acc, cst + kConstBias8 + 9 * 256 + h * 128);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE8(buf, SRC_ACT, 16 * s)
    ++g;
  }
  if (SAVE) {
#pragma unroll
    for (int i = 0; i < 128; ++i) image_store(rec, kSaveInsF + i, lane, acc[i >> 4][i & 15]);
  }
  // ---- ins_feature_linears.0: 256 -> 128, ReLU (dm_nerf.py:97-99)
  bias_tiles<4>(acc2, cst + kConstBias4 + 128 + h * 64);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane);
    DMN_STAGE4(buf, SRC_ACC, 32 * s, 32)
    ++g;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[t][r] = fmaxf(acc2[t][r], 0.f);
  if (SAVE) {
    float i1[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) i1[i] = acc2[i >> 4][i & 15];
    save_relu<64>(rec + kSaveI1 * 64, rec + kSaveMaskI1 * 64, lane, i1);
  }
  // ---- ins_linear: 128 -> ins_num + 1 (dm_nerf.py:103), T output tiles
  f32x16 acc3[T];
  bias_tiles<T>(acc3, cst + kConstBiasO + h * 64);
  constexpr int JS = inso_groups_per_stage(T);
#pragma unroll
  for (int st = 0; st < inso_stages(T); ++st) {
    buf = stage_begin(stream, g, G, lds, wave_s, lane, true);      // short stages: their successor in a burst
#pragma unroll
    for (int jl = 0; jl < JS; ++jl) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(buf + (jl * T + t) * kChunkFloats);
        const int k0 = 4 * (st * JS + jl);
        acc3[t] = DMN_MFMA(a0.x, SRC_ACC2(k0 + 0), acc3[t]);
        acc3[t] = DMN_MFMA(a0.y, SRC_ACC2(k0 + 1), acc3[t]);
        acc3[t] = DMN_MFMA(a0.z, SRC_ACC2(k0 + 2), acc3[t]);
        acc3[t] = DMN_MFMA(a0.w, SRC_ACC2(k0 + 3), acc3[t]);
      }
    }
    ++g;
  }

  // ---- raw = cat([rgb, density, ins]) (dm_nerf.py:105)
  if (valid) {
    const int C = 4 + a.ins_ch;
    float* out = a.raw + m * C;
    if (h == 0) {
      out[0] = rgb[0];
      out[1] = rgb[1];
      out[2] = rgb[2];
      out[3] = sigma;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (ch < a.ins_ch) out[4 + ch] = acc3[t][r];
      }
  }
}

template <bool SAVE>
int launch_field_t(const FieldArgs& a, int T, dim3 grid, hipStream_t st) {
  const dim3 block(256);
  switch (T) {
    case 1: hipLaunchKernelGGL((field_fwd_kernel<1, SAVE>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((field_fwd_kernel<2, SAVE>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((field_fwd_kernel<3, SAVE>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((field_fwd_kernel<4, SAVE>), grid, block, 0, st, a); break;
    default: return DMN_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

int launch_field(const FieldArgs& a, int ins_num, hipStream_t st) {
  if (a.M <= 0) return DMN_OK;
  const long blocks = (a.M + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks);
  return a.save ? launch_field_t<true>(a, ins_tiles(ins_num), grid, st)
                : launch_field_t<false>(a, ins_tiles(ins_num), grid, st);
}

}  // namespace

extern "C" int dmn_field_fwd_rays(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                  long n_rays, int n_samples, const float* blob, int ins_num, float* raw,
                                  void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw) return DMN_ERR_BAD_ARG;
  if (ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, nullptr};
  return launch_field(a, ins_num, (hipStream_t)stream);
}

extern "C" long dmn_field_save_floats(long n_points) {
  return n_points < 0 ? DMN_ERR_BAD_ARG : ((n_points + 127) / 128) * 4 * kSaveTileFloats;
}

extern "C" int dmn_field_fwd_rays_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                       long n_rays, int n_samples, const float* blob, int ins_num, float* raw,
                                       float* save, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw || !save) return DMN_ERR_BAD_ARG;
  if (ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, save};
  return launch_field(a, ins_num, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_embedded(const float* embedded, long n_points, const float* blob, int ins_num,
                                      float* raw, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!embedded || !blob || !raw) return DMN_ERR_BAD_ARG;
  if (ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  FieldArgs a{nullptr, nullptr, nullptr, 0, embedded, blob, raw, nullptr, n_points, 1, ins_num + 1};
  return launch_field(a, ins_num, (hipStream_t)stream);
}
