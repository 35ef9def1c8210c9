// Backward-data kernel of the field on split-precision MFMA (three bf16 products per fp32 product, fp32
// accumulate): the same chain as field_bwd_data_kernel (field_bwd.hip) -- dL/d pre-activation of every layer,
// transposed weights as the A operand, the previous gradient tile as B, ReLU masks from the forward's save record,
// every gradient tile written once as a register image -- on the tile-outer machinery of field_x3.h.
// bf16 only: gradients span the fp32 exponent range, fp16 would flush them.  Masks come from the save record, so
// no ReLU decision is re-made here: the error is the 2^-16 relative rounding of the products.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_common.h"
#include "field_x3.h"

using namespace dmn;

namespace {

struct BwdX3Args {
  const float* g_raw;
  const float* save;
  const char* blob;
  float* grad;
  long M;
  int ins_ch;
};

typedef PrecBF16 P;
typedef P::frag frag;

// piece q (0..7) of tile t: accumulator registers 2q, 2q+1 = image registers 16t + 2q, 16t + 2q + 1
template <bool MASK, bool SPLIT, bool SIG = false>
struct EpiGrad {
  frag* oh;
  frag* ol;
  float* img;              // gradient image of this layer (lane offset included)
  const unsigned* msk;     // sign bits of the layer's ReLU (MASK)
  float sig;               // dL/d sigma, folded into the trunk-output gradient (0 elsewhere)
  const float* dw;         // density weights of this lane-half
  __device__ __forceinline__ void operator()(int t, int q, float x0, float x1) const {
    const int i = 16 * t + 2 * q;
    if (SIG) {
      x0 = fmaf(sig, dw[i], x0);
      x1 = fmaf(sig, dw[i + 1], x1);
    }
    if (MASK) {
      const unsigned w = msk[i >> 5];
      x0 = (w >> (i & 31)) & 1u ? x0 : 0.f;
      x1 = (w >> ((i + 1) & 31)) & 1u ? x1 : 0.f;
    }
    record_store(img + i * 64, x0);
    record_store(img + (i + 1) * 64, x1);
    if (SPLIT) split_pair<P, true>(x0, x1, oh[2 * t + (q >> 2)], ol[2 * t + (q >> 2)], (2 * q) & 7);
  }
};

template <int T>
__global__ __launch_bounds__(256, 1) void field_bwd_data_x3_kernel(const BwdX3Args a) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  float* const cst = reinterpret_cast<float*>(lds + 2 * kX3StageBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  Ctx c{a.blob + kConstFloats * 4, lds, nullptr, 0, kBwdX3Stages, wave_s, lane};
  for (int k = wave_s; k < kConstChunks; k += 4) glds16(a.blob + k * 1024 + lane * 16, lds + 2 * kX3StageBytes + k * 1024);
  load_stage(c, 0);

  const long tile = (long)blockIdx.x * 4 + wave_s;
  const long m = tile * 32 + p;
  const bool valid = m < a.M;
  const float* const srec = a.save + tile * kSaveTileFloats + lane;
  float* const grec = a.grad + tile * grad_tile_floats(T) + lane;
  const float* const zero = cst + kConstZeros;              // initial accumulators
  const float* const dw = cst + kConstDensW + h * 128;

  // ---- dL/d raw of this lane's point
  frag gi_h[2 * T], gi_l[2 * T];
  float g_rgb[3] = {0.f, 0.f, 0.f}, g_sig = 0.f;
  {
    const int C = 4 + a.ins_ch;
    const float* row = a.g_raw + (valid ? m : 0) * C;
    if (valid) {
      g_rgb[0] = row[0]; g_rgb[1] = row[1]; g_rgb[2] = row[2]; g_sig = row[3];
    }
    float gi[16 * T];
#pragma unroll
    for (int i = 0; i < 16 * T; ++i) {
      const int ch = feat_of(i, 0) + 4 * h;
      gi[i] = (valid && ch < a.ins_ch) ? row[4 + (ch < a.ins_ch ? ch : 0)] : 0.f;
      record_store(grec + (kGradHead + i) * 64, gi[i]);
    }
#pragma unroll
    for (int i = 0; i < 16 * T; i += 2) split_pair<P>(gi[i], gi[i + 1], gi_h[i >> 3], gi_l[i >> 3], i & 7);
    record_store(grec + (kGradHead + 16 * T + 0) * 64, h ? g_rgb[1] : g_rgb[0]);
    record_store(grec + (kGradHead + 16 * T + 1) * 64, h ? g_sig : g_rgb[2]);
#pragma unroll
    for (int i = 2; i < 16; ++i) record_store(grec + (kGradHead + 16 * T + i) * 64, 0.f);
  }

  frag act_h[16], act_l[16], nxt_h[16], nxt_l[16];
  unsigned msk[4];
  const_section_ready();      // the first layer reads its initial accumulators before its first stage barrier

  // ---- instance branch: ins_linear^T -> mask of ins_feature_linears.0 -> its transpose
  msk[0] = __float_as_uint(srec[(kSaveMaskI1 + 0) * 64]);
  msk[1] = __float_as_uint(srec[(kSaveMaskI1 + 1) * 64]);
  layer_x3<P, 4, 2 * T, 0>(c, zero, gi_h, gi_l, gi_h, gi_l,
                           EpiGrad<true, true>{act_h, act_l, grec + kGradI1 * 64, msk, 0.f, dw});
  layer_x3<P, 8, 8, 0>(c, zero, act_h, act_l, act_h, act_l,
                       EpiGrad<false, false>{nxt_h, nxt_l, grec + kGradInsF * 64, msk, 0.f, dw});

  // ---- colour branch: rgb_linear^T on the VALU, mask of rgb_feature_linears.0, its transpose, rgb_feature_linear^T
  msk[0] = __float_as_uint(srec[(kSaveMaskR1 + 0) * 64]);
  msk[1] = __float_as_uint(srec[(kSaveMaskR1 + 1) * 64]);
#pragma unroll
  for (int i = 0; i < 64; i += 2) {
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float x = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) x = fmaf(g_rgb[ch], cst[kConstRgbW + ch * 128 + h * 64 + i + e], x);
      v[e] = (msk[(i + e) >> 5] >> ((i + e) & 31)) & 1u ? x : 0.f;
      record_store(grec + (kGradR1 + i + e) * 64, v[e]);
    }
    split_pair<P>(v[0], v[1], nxt_h[i >> 3], nxt_l[i >> 3], i & 7);
  }
  layer_x3<P, 8, 8, 0>(c, zero, nxt_h, nxt_l, nxt_h, nxt_l,
                       EpiGrad<false, true>{act_h, act_l, grec + kGradFeat * 64, msk, 0.f, dw});

  // ---- trunk: rgb_feature_linear^T (+ density_linear^T) masked by h_8, then mlps.7 .. mlps.1 transposed
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(8) + w) * 64]);
  layer_x3<P, 8, 16, 0>(c, zero, act_h, act_l, act_h, act_l,
                        EpiGrad<true, true, true>{nxt_h, nxt_l, grec + (kGradA + 128 * 7) * 64, msk, g_sig, dw});
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(7) + w) * 64]);
  layer_x3<P, 8, 16, 0>(c, zero, nxt_h, nxt_l, nxt_h, nxt_l,
                        EpiGrad<true, true>{act_h, act_l, grec + (kGradA + 128 * 6) * 64, msk, 0.f, dw});
#pragma unroll 1
  for (int l = 6; l >= 2; l -= 2) {      // W_l^T G_l -> G_{l-1}, W_{l-1}^T G_{l-1} -> G_{l-2}
#pragma unroll
    for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(1) + 4 * (l - 1) + w) * 64]);
    layer_x3<P, 8, 16, 0>(c, zero, act_h, act_l, act_h, act_l,
                          EpiGrad<true, true>{nxt_h, nxt_l, grec + (kGradA + 128 * (l - 1)) * 64, msk, 0.f, dw});
#pragma unroll
    for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(1) + 4 * (l - 2) + w) * 64]);
    layer_x3<P, 8, 16, 0>(c, zero, nxt_h, nxt_l, nxt_h, nxt_l,
                          EpiGrad<true, true>{act_h, act_l, grec + (kGradA + 128 * (l - 2)) * 64, msk, 0.f, dw});
  }
}

}  // namespace

extern "C" int dmn_field_bwd_data_x3(const float* g_raw, const float* save, const void* bwd_blob_x3, long n_points,
                                     int ins_num, float* grad_record, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!g_raw || !save || !bwd_blob_x3 || !grad_record) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  if (T > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  const long blocks = (n_points + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const BwdX3Args a{g_raw, save, static_cast<const char*>(bwd_blob_x3), grad_record, n_points, ins_num + 1};
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (T) {
    case 1: hipLaunchKernelGGL(field_bwd_data_x3_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(field_bwd_data_x3_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(field_bwd_data_x3_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(field_bwd_data_x3_kernel<4>, grid, block, 0, st, a); break;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
