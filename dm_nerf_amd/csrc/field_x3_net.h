// The inference network of the split-precision field kernels as one device function: position / view encodings in,
// [rgb, sigma, instance logits] of the wave's 32 points out in registers.  Shared by field_fwd_x3.hip (writes raw rows)
// and field_fused_x3.hip (composites them in the kernel); the training (saving) forward keeps its own copy with the
// save-record epilogues (field_fwd_x3.hip).  networks/dm_nerf.py:80-106.
#pragma once
#include "field_x3.h"

namespace dmn {

// Inference epilogues: DENS is compile-time -- as a run-time flag every epilogue piece carried a branch that cut the MFMA
// shadow it is scheduled into (63 branches per kernel).
template <class P, bool RELU, bool DENS = false>
struct EpiSplit {
  typename P::frag* oh;
  typename P::frag* ol;
  float* amax;
  float* sigma;          // density accumulation (DENS)
  const float* dw;       // density weights of this lane-half
  __device__ __forceinline__ void operator()(int t, int q, float x0, float x1) const {
    if (RELU) {
      x0 = relu_bits(x0);
      x1 = relu_bits(x1);
    }
    if (DENS) *sigma = fmaf(x1, dw[16 * t + 2 * q + 1], fmaf(x0, dw[16 * t + 2 * q], *sigma));
    if (P::kCheckRange) {
      *amax = fmaxf(*amax, fmaxf(fabsf(x0), fabsf(x1)));
      // materialise the running maximum here: left free, LLVM reassociates the kernel-long max chain into one
      // tree at the end of the kernel and keeps every converted value alive until then
      asm volatile("" : "+v"(*amax));
    }
    split_pair<P, true>(x0, x1, oh[2 * t + (q >> 2)], ol[2 * t + (q >> 2)], (2 * q) & 7);
  }
};
// -> rgb_linear on the VALU (ReLU first; dm_nerf.py:91-93,102)
struct EpiRgb {
  float* rgb;            // [3] partial sums of this lane-half
  const float* rw;       // cst + kConstRgbW + h*64
  __device__ __forceinline__ void operator()(int t, int q, float a0, float a1) const {
    const float x0 = relu_bits(a0), x1 = relu_bits(a1);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rgb[c] = fmaf(x1, rw[c * 128 + 16 * t + 2 * q + 1], fmaf(x0, rw[c * 128 + 16 * t + 2 * q], rgb[c]));
  }
};
// -> registers: value of accumulator register r of tile t lands in keep[16 t + r] (channel feat_of(16 t + r, 0) + 4 h)
struct EpiKeep {
  float* keep;
  __device__ __forceinline__ void operator()(int t, int q, float x0, float x1) const {
    keep[16 * t + 2 * q] = x0;
    keep[16 * t + 2 * q + 1] = x1;
  }
};

// Runs the whole network on this wave's 32 points (one per lane pair; lane-half h = lane >> 5).
//   c      weight stream context (stage 0 already requested by the caller: load_stage(c, 0))
//   cst    constant section in LDS
//   pe_*, qe_*   hi / lo operand fragments of the position and view encodings
// Outputs (valid in both lane halves unless noted): sigma, rgb[3], and insv[i] = instance logit of channel
// feat_of(i, 0) + 4 h (i < 16 T) of this lane's point.  With sigma_only the colour and instance branches (29 % of the
// network) are skipped and rgb / insv come back as zeros.  `amax` tracks |activation| for the fp16 range flag.
template <class P, int T>
__device__ __forceinline__ void x3_network(Ctx& c, const float* cst, const typename P::frag* pe_h,
                                           const typename P::frag* pe_l, const typename P::frag* qe_h,
                                           const typename P::frag* qe_l, int h, bool sigma_only, float& amax,
                                           float& sigma, float (&rgb)[3], float (&insv)[16 * T]) {
  typedef typename P::frag frag;
  frag act_h[16], act_l[16], nxt_h[16], nxt_l[16];
  sigma = 0.f;
  const float* const dw = cst + kConstDensW + h * 128;
  typedef EpiSplit<P, true, false> Relu;
  typedef EpiSplit<P, true, true> ReluDens;
  typedef EpiSplit<P, false, false> Lin;

  const_section_ready();      // mlps.0 reads its bias before its first stage barrier
  // ---- mlps.0: position encoding (4 k-steps) -> act
  layer_x3<P, 8, 4, 0>(c, cst + kConstBias8 + h * 128, pe_h, pe_l, pe_h, pe_l, Relu{act_h, act_l, &amax, &sigma, dw});
  // ---- mlps.1 .. mlps.5 in ping-pong pairs (rolled: a layer pair is ~3000 instructions); mlps.5 (skip connection)
  //      also consumes the position encoding
  int l = 1;
#pragma unroll 1
  for (int it = 0; it < 2; ++it) {
    layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + l * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                          Relu{nxt_h, nxt_l, &amax, &sigma, dw});
    ++l;
    layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + l * 256 + h * 128, nxt_h, nxt_l, pe_h, pe_l,
                          Relu{act_h, act_l, &amax, &sigma, dw});
    ++l;
    if (it == 1) {
      layer_x3<P, 8, 16, 4>(c, cst + kConstBias8 + l * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                            Relu{nxt_h, nxt_l, &amax, &sigma, dw});
      ++l;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        act_h[k] = nxt_h[k];
        act_l[k] = nxt_l[k];
      }
    }
  }
  // ---- mlps.6, and mlps.7 with the density head's partial sums (its own instantiation: DENS is compile-time)
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 6 * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                        Relu{nxt_h, nxt_l, &amax, &sigma, dw});
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 7 * 256 + h * 128, nxt_h, nxt_l, pe_h, pe_l,
                        ReluDens{act_h, act_l, &amax, &sigma, dw});
  // ---- density_linear (dm_nerf.py:101): fp32 partial sums were taken while mlps.7 was converted
  sigma += __shfl_xor(sigma, 32);
  sigma += cst[kConstDensB];
  rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;
#pragma unroll
  for (int i = 0; i < 16 * T; ++i) insv[i] = 0.f;
  if (sigma_only) return;      // density is all the caller wants (the coarse pass of an image loop feeds the resampler)

  // ---- rgb_feature_linear (no activation) -> nxt; rgb_feature_linears.0 on [feature, dirs] -> rgb_linear
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 8 * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                        Lin{nxt_h, nxt_l, &amax, &sigma, dw});
  layer_x3<P, 4, 16, 2>(c, cst + kConstBias4 + h * 64, nxt_h, nxt_l, qe_h, qe_l, EpiRgb{rgb, cst + kConstRgbW + h * 64});
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rgb[k] += __shfl_xor(rgb[k], 32);
    rgb[k] += cst[kConstRgbB + k];
  }
  // ---- ins_feature_linear on the trunk output (no activation) -> nxt; ins_feature_linears.0 (ReLU) -> act[0..7]
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 9 * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                        Lin{nxt_h, nxt_l, &amax, &sigma, dw});
  layer_x3<P, 4, 16, 0>(c, cst + kConstBias4 + 128 + h * 64, nxt_h, nxt_l, pe_h, pe_l,
                        Relu{act_h, act_l, &amax, &sigma, dw});
  // ---- ins_linear: 128 -> ins_num + 1, T tiles, kept in registers
  layer_x3<P, T, 8, 0>(c, cst + kConstBiasO + h * 64, act_h, act_l, pe_h, pe_l, EpiKeep{insv});
}

// pts = o + d z, viewdirs, encodings and their hi / lo operand fragments for point mc (render.py:37-38,49-58)
template <class P>
__device__ __forceinline__ void x3_point_fragments(const FieldArgs& a, long mc, int h, float& amax,
                                                   typename P::frag* pe_h, typename P::frag* pe_l,
                                                   typename P::frag* qe_h, typename P::frag* qe_l) {
  float pe[32], qe[16];
  point_encodings(a, mc, h, pe, qe);
#pragma unroll
  for (int j = 0; j < 32; j += 2) {
    if (P::kCheckRange) amax = fmaxf(amax, fmaxf(fabsf(pe[j]), fabsf(pe[j + 1])));
    split_pair<P>(pe[j], pe[j + 1], pe_h[j >> 3], pe_l[j >> 3], j & 7);
  }
#pragma unroll
  for (int j = 0; j < 16; j += 2) split_pair<P>(qe[j], qe[j + 1], qe_h[j >> 3], qe_l[j >> 3], j & 7);
}

}  // namespace dmn
