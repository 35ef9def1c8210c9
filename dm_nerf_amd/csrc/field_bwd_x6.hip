// Backward-data kernel of the field in the full-precision bf16 arithmetic ("bf16x6", field_x6.h): the same chain as
// field_bwd_data_kernel (field_bwd.hip) and field_bwd_data_x3_kernel -- dL/d pre-activation of every layer,
// G_in^T = W^T G_out^T with the transposed weights as the MFMA A operand and the previous gradient tile as B, ReLU
// masks from the forward's save record, every gradient tile written once as a register image -- with every fp32 product
// evaluated from EXACT three-way bf16 splits of both operands (six bf16 MFMA products, fp32 accumulation): the gradient
// arithmetic of the reference (autograd in fp32, train_dmsr.py:62-64 through networks/dm_nerf.py:80-106), where the x3
// kernel keeps 16 operand bits.  k-steps outer on two alternating accumulator sets (layer_x6): a layer's output gradient
// stays in the accumulators that produced it; the consuming layer reads it, adds the density head's contribution (h8
// only), applies the mask, stores the row of the gradient record and splits it -- once per value, in the MFMA shadow.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_x6.h"

using namespace dmn;

namespace {

struct BwdX6Args {
  const float* g_raw;
  const float* save;
  const char* blob;
  float* grad;
  long M;
  int ins_ch;
};

// Gradient w.r.t. a layer's output as the next (earlier) layer's operand: accumulator register i of the finished set,
// plus sig * density weight (SIG: the trunk output feeds density_linear as well, dm_nerf.py:101), times the ReLU mask
// bit of register i (MASK: the forward's sign bits, so no ReLU decision is re-made here).
template <bool MASK, bool SIG>
struct SrcGrad {
  const f32x16* a;
  const unsigned* msk;
  float sig;
  const float* dw;
  __device__ __forceinline__ float operator()(int i) const {
    float x = acc_read(a[i >> 4], i & 15);
    if (SIG) x = fmaf(sig, dw[i], x);
    if (MASK) x = (msk[i >> 5] >> (i & 31)) & 1u ? x : 0.f;
    return x;
  }
};

template <int T>
__global__ __launch_bounds__(256, 1) void field_bwd_data_x6_kernel(const BwdX6Args a) {
  __shared__ __attribute__((aligned(16))) char lds[kX6LdsBytes];
  float* const cst = reinterpret_cast<float*>(lds + kX6Slots * kX6StageBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  Ctx6 c{a.blob + kConstFloats * 4, lds, nullptr, 0, bwd_x6_total_stages(T), wave_s, lane};
  for (int k = wave_s; k < kConstChunks; k += 4)
    glds16(a.blob + k * 1024 + lane * 16, lds + kX6Slots * kX6StageBytes + k * 1024);
  x6_load_stage(c, 0);
  if (kX6Slots == 3) x6_load_stage(c, 1);

  const long tile = (long)blockIdx.x * 4 + wave_s;
  const long m = tile * 32 + p;
  const bool valid = m < a.M;
  const float* const srec = a.save + tile * kSaveTileFloats + lane;
  float* const grec = a.grad + tile * grad_tile_floats(T) + lane;
  const float* const zero = cst + kConstZeros;              // initial accumulators
  const float* const dw = cst + kConstDensW + h * 128;

  // ---- dL/d raw of this lane's point
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
      record_store(grec + (kGradHead + i) * 64, gi[i]);
    }
    record_store(grec + (kGradHead + 16 * T + 0) * 64, h ? g_rgb[1] : g_rgb[0]);
    record_store(grec + (kGradHead + 16 * T + 1) * 64, h ? g_sig : g_rgb[2]);
#pragma unroll
    for (int i = 2; i < 16; ++i) record_store(grec + (kGradHead + 16 * T + i) * 64, 0.f);
  }

  const_section_ready();
  unsigned msk[4] = {0u, 0u, 0u, 0u};
  f32x16 accA[8], accB[8], acc2[4];
  constexpr int K8 = x6_ksteps_per_stage(8), K4 = x6_ksteps_per_stage(4);
  constexpr int LID = 31;                                   // no non-temporal DMA: the 3.4 MiB stream fits the L2
  const SrcRegs s_gi{gi};
  const SrcRegs s_none{gi};

  // ---- instance branch: ins_linear^T -> mask of ins_feature_linears.0 -> its transpose (the branch stops there:
  //      h.detach(), dm_nerf.py:95)
  layer_x6<LID, 4, 2 * T, 0, K4, false, false, false>(c, acc2, zero, s_gi, s_none);
  msk[0] = __float_as_uint(srec[(kSaveMaskI1 + 0) * 64]);
  msk[1] = __float_as_uint(srec[(kSaveMaskI1 + 1) * 64]);
  c.g = bwd_x6_stage_insh(T);
  layer_x6<LID, 8, 8, 0, K8, false, false, false>(c, accA, zero, SrcGrad<true, false>{acc2, msk, 0.f, dw}, s_none, nullptr,
                                                  SaveRows<false>{grec + kGradI1 * 64, msk});
#pragma unroll
  for (int i = 0; i < 128; ++i) record_store(grec + (kGradInsF + i) * 64, accA[i >> 4][i & 15]);

  // ---- colour branch: rgb_linear^T on the VALU, mask of rgb_feature_linears.0, its transpose, rgb_feature_linear^T
  float v[64];
  {
    const unsigned m0 = __float_as_uint(srec[(kSaveMaskR1 + 0) * 64]), m1 = __float_as_uint(srec[(kSaveMaskR1 + 1) * 64]);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      float x = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) x = fmaf(g_rgb[ch], cst[kConstRgbW + ch * 128 + h * 64 + i], x);
      v[i] = ((i < 32 ? m0 : m1) >> (i & 31)) & 1u ? x : 0.f;
      record_store(grec + (kGradR1 + i) * 64, v[i]);
    }
  }
  c.g = bwd_x6_stage_rgbh(T);
  layer_x6<LID, 8, 8, 0, K8, false, false, false>(c, accB, zero, SrcRegs{v}, s_none);
  c.g = bwd_x6_stage_rgbf(T);
  layer_x6<LID, 8, 16, 0, K8, false, false, false>(c, accA, zero, SrcGrad<false, false>{accB, msk, 0.f, dw}, s_none, nullptr,
                                                   SaveRows<false>{grec + kGradFeat * 64, msk});

  // ---- trunk: (rgb_feature_linear^T + density_linear^T) masked by h_8 = G_7, then mlps.7 .. mlps.1 transposed
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(8) + w) * 64]);
  c.g = bwd_x6_stage_mlp(T, 7);
  layer_x6<LID, 8, 16, 0, K8, false, false, false>(c, accB, zero, SrcGrad<true, true>{accA, msk, g_sig, dw}, s_none, nullptr,
                                                   SaveRows<false>{grec + (kGradA + 128 * 7) * 64, msk});
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(7) + w) * 64]);
  c.g = bwd_x6_stage_mlp(T, 6);
  layer_x6<LID, 8, 16, 0, K8, false, false, false>(c, accA, zero, SrcGrad<true, false>{accB, msk, 0.f, dw}, s_none, nullptr,
                                                   SaveRows<false>{grec + (kGradA + 128 * 6) * 64, msk});
#pragma unroll 1
  for (int l = 5; l >= 3; l -= 2) {      // mlps.l^T: G_l (mask of h_(l+1)) -> accB;  mlps.(l-1)^T: G_(l-1) -> accA
#pragma unroll
    for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(l + 1) + w) * 64]);
    layer_x6<LID, 8, 16, 0, K8, false, false, false>(c, accB, zero, SrcGrad<true, false>{accA, msk, 0.f, dw}, s_none, nullptr,
                                                     SaveRows<false>{grec + (kGradA + 128 * l) * 64, msk});
#pragma unroll
    for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(l) + w) * 64]);
    layer_x6<LID, 8, 16, 0, K8, false, false, false>(c, accA, zero, SrcGrad<true, false>{accB, msk, 0.f, dw}, s_none, nullptr,
                                                     SaveRows<false>{grec + (kGradA + 128 * (l - 1)) * 64, msk});
  }
  // mlps.1^T: G_1 (mask of h_2) -> accB = dL/d h_1; masked by h_1: G_0
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(2) + w) * 64]);
  c.g = bwd_x6_stage_mlp(T, 1);
  layer_x6<LID, 8, 16, 0, K8, false, false, true>(c, accB, zero, SrcGrad<true, false>{accA, msk, 0.f, dw}, s_none, nullptr,
                                                  SaveRows<false>{grec + (kGradA + 128 * 1) * 64, msk});
#pragma unroll
  for (int w = 0; w < 4; ++w) msk[w] = __float_as_uint(srec[(save_mask(1) + w) * 64]);
#pragma unroll
  for (int i = 0; i < 128; ++i)
    record_store(grec + (kGradA + i) * 64, (msk[i >> 5] >> (i & 31)) & 1u ? accB[i >> 4][i & 15] : 0.f);
  if (kX6Slots == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int dmn_field_bwd_data_x6(const float* g_raw, const float* save, const void* bwd_blob_x6, long n_points,
                                     int ins_num, float* grad_record, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!g_raw || !save || !bwd_blob_x6 || !grad_record) return DMN_ERR_BAD_ARG;
  const int T = ins_tiles(ins_num);
  if (T > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  const long blocks = (n_points + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const BwdX6Args a{g_raw, save, static_cast<const char*>(bwd_blob_x6), grad_record, n_points, ins_num + 1};
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (T) {
    case 1: hipLaunchKernelGGL(field_bwd_data_x6_kernel<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(field_bwd_data_x6_kernel<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(field_bwd_data_x6_kernel<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(field_bwd_data_x6_kernel<4>, grid, block, 0, st, a); break;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
