// Split-precision field kernel for gfx950: the same fused network as field_fwd.hip, with every
// fp32 product W*X evaluated as  Whi*Xhi + Whi*Xlo + Wlo*Xhi  on v_mfma_f32_32x32x16_{f16,bf16}
// (hi = 16-bit rounding of the value, lo = 16-bit rounding of the residual; fp32 accumulation).
// Three 32-cycle MFMAs cover 16 k-values where the exact kernel needs eight 64-cycle ones: 5.3x the
// MFMA throughput.  Measured accuracy against the fp32 reference path: fp16 split ~1e-6 relative on
// the raw outputs (two 11-bit halves = 22 bits), bf16 split ~1e-5 (two 8-bit halves).  The fp16
// flavour needs |activation| < 65504; the kernel tracks the running maximum and raises a sticky
// status bit instead of returning silently wrong numbers (the host then re-renders exactly).
//
// Structure (differences from field_fwd.hip):
//   * output tiles in the OUTER loop, k-steps inner: one 32x32 accumulator (plus one being
//     converted) instead of eight, so the converted activations of the layer being produced
//     (128 registers of packed hi/lo halves) and of the layer being consumed both fit;
//   * the epilogue of tile t-1 (ReLU, hi/lo split, range tracking) is spread over the k-steps of
//     tile t so its VALU work runs in the shadow of the MFMAs;
//   * 64 KiB stages (32 k-step/tile pairs x [hi, lo] chunk), double buffered; the LDS-DMA of the next stage
//     is issued one 1 KiB piece per step; A fragments prefetched two steps ahead inside a stage.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_common.h"
#include "field_x3.h"
#include "field_x3_net.h"

using namespace dmn;

namespace {

// ---- tile epilogues: piece q (0..7) handles accumulator registers 2q, 2q+1 of tile t -----------
// -> next layer's operand fragments (register 16t + r is element (r & 7) of k-step 2t + (r >> 3))
// ---- helpers of the saving epilogues: one store instruction per value and no address arithmetic on the vector unit --
// base = the tile's 16 image rows (wave-uniform: an SGPR pair advanced by scalar adds), per-lane offset lane * 4 in one
// VGPR, the row as the instruction's immediate; sign bit of a post-ReLU value (+0 or positive) as min(bits, 1) shifted
// in: two instructions instead of compare + select + shift + or.  Inline asm so that the forms survive the optimiser.
template <int ROW>
__device__ __forceinline__ void image_row_store(const float* tile_rows, unsigned lane_bytes, float x) {
  asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" : : "v"(lane_bytes), "v"(x), "s"(tile_rows), "n"(ROW * 256));
}
__device__ __forceinline__ void image_pair_store(const float* tile_rows, unsigned lane_bytes, int q, float x0, float x1) {
  switch (q) {      // q is a compile-time constant where this is inlined (unrolled epilogue pieces)
    case 0: image_row_store<0>(tile_rows, lane_bytes, x0); image_row_store<1>(tile_rows, lane_bytes, x1); break;
    case 1: image_row_store<2>(tile_rows, lane_bytes, x0); image_row_store<3>(tile_rows, lane_bytes, x1); break;
    case 2: image_row_store<4>(tile_rows, lane_bytes, x0); image_row_store<5>(tile_rows, lane_bytes, x1); break;
    case 3: image_row_store<6>(tile_rows, lane_bytes, x0); image_row_store<7>(tile_rows, lane_bytes, x1); break;
    case 4: image_row_store<8>(tile_rows, lane_bytes, x0); image_row_store<9>(tile_rows, lane_bytes, x1); break;
    case 5: image_row_store<10>(tile_rows, lane_bytes, x0); image_row_store<11>(tile_rows, lane_bytes, x1); break;
    case 6: image_row_store<12>(tile_rows, lane_bytes, x0); image_row_store<13>(tile_rows, lane_bytes, x1); break;
    default: image_row_store<14>(tile_rows, lane_bytes, x0); image_row_store<15>(tile_rows, lane_bytes, x1); break;
  }
}
__device__ __forceinline__ void sign_bit_or(unsigned& w, float x, int bit) {
  unsigned b;
  asm volatile("v_min_u32 %0, 1, %1" : "=v"(b) : "v"(__float_as_uint(x)));
  asm volatile("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(w) : "v"(b), "v"(bit));
}

// Training (saving) epilogues: run-time flags on purpose.  With compile-time flags (no branches) the compiler moves the
// 128 image stores and sign-bit updates of a layer freely and the kernel needs 512 registers + 1.3 KB of spills; the
// branches of this form keep every piece where it is written (407 registers, no scratch).
template <class P, bool RELU>
struct EpiSplitSave {
  typename P::frag* oh;
  typename P::frag* ol;
  float* amax;
  float* sigma;          // density accumulation (only when dens)
  const float* dw;       // density weights of this lane-half
  bool dens;
  float* img = nullptr;  // training: register image of this layer's output (field_layout.h), lane offset included
  unsigned* msk = nullptr;   // training: sign bits of the ReLU layer, 32 registers per word
  const float* imgu = nullptr;   // the same image, wave-uniform (no lane offset), and the lane's byte offset
  unsigned lane_bytes = 0;
  __device__ __forceinline__ void operator()(int t, int q, float x0, float x1) const {
    if (RELU) {
      x0 = relu_bits(x0);
      x1 = relu_bits(x1);
    }
    if (imgu != nullptr) {     // always taken in the saving kernel; the run-time (scalar) test keeps the piece a unit (see above)
      image_pair_store(imgu + 16 * t * 64, lane_bytes, q, x0, x1);
      if (RELU) {
        unsigned& w = msk[(16 * t + 2 * q) >> 5];
        sign_bit_or(w, x0, (16 * t + 2 * q) & 31);
        sign_bit_or(w, x1, (16 * t + 2 * q + 1) & 31);
      }
    }
    if (dens) *sigma = fmaf(x1, dw[16 * t + 2 * q + 1], fmaf(x0, dw[16 * t + 2 * q], *sigma));
    if (P::kCheckRange) {
      *amax = fmaxf(*amax, fmaxf(fabsf(x0), fabsf(x1)));
      // materialise the running maximum here: left free, LLVM reassociates the kernel-long max chain into one
      // tree at the end of the kernel and keeps every converted value alive until then
      asm volatile("" : "+v"(*amax));
    }
    split_pair<P, true>(x0, x1, oh[2 * t + (q >> 2)], ol[2 * t + (q >> 2)], (2 * q) & 7);
  }
};
struct EpiRgbSave {
  float* rgb;            // [3] partial sums of this lane-half
  const float* rw;       // cst + kConstRgbW + h*64
  float* img = nullptr;  // training: image of the post-ReLU layer output, and its sign bits
  unsigned* msk = nullptr;
  const float* imgu = nullptr;
  unsigned lane_bytes = 0;
  __device__ __forceinline__ void operator()(int t, int q, float a0, float a1) const {
    const float x0 = relu_bits(a0), x1 = relu_bits(a1);
    if (imgu != nullptr) {
      image_pair_store(imgu + 16 * t * 64, lane_bytes, q, x0, x1);
      unsigned& w = msk[(16 * t + 2 * q) >> 5];
      sign_bit_or(w, x0, (16 * t + 2 * q) & 31);
      sign_bit_or(w, x1, (16 * t + 2 * q + 1) & 31);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      rgb[c] = fmaf(x1, rw[c * 128 + 16 * t + 2 * q + 1], fmaf(x0, rw[c * 128 + 16 * t + 2 * q], rgb[c]));
  }
};
// -> instance logits straight to global memory (dm_nerf.py:103,105)
struct EpiStore {
  float* out;            // raw row of this lane's point, or null
  int ins_ch, h;
  __device__ __forceinline__ void operator()(int t, int q, float x0, float x1) const {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = 2 * q + e;
      const int ch = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (out != nullptr && ch < ins_ch) out[4 + ch] = e ? x1 : x0;
    }
  }
};

// training: write and clear the sign-bit words a ReLU layer's epilogues collected
template <int N>
__device__ __forceinline__ void flush_mask(float* dst, unsigned* msk) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    record_store(dst + w * 64, __uint_as_float(msk[w]));
    msk[w] = 0u;
  }
}

template <class P, int T, bool SAVE>
__global__ __launch_bounds__(256, 1) void field_fwd_x3_kernel(const FieldArgs a) {
  typedef typename P::frag frag;
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  float* const cst = reinterpret_cast<float*>(lds + 2 * kX3StageBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const blob = static_cast<const char*>(a.blob);
  Ctx c{blob + kConstFloats * 4, lds, nullptr, 0, kX3Stages, wave_s, lane};
#ifdef DMN_STAMPS
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
  const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif

  for (int k = wave_s; k < kConstChunks; k += 4) glds16(blob + k * 1024 + lane * 16, lds + 2 * kX3StageBytes + k * 1024);
  load_stage(c, 0);

  const long m = ((long)blockIdx.x * 4 + wave_s) * 32 + p;
  const bool valid = m < a.M;
  const long mc = valid ? m : a.M - 1;
  float amax = 0.f;
  // training: this wave's save record, lane offset included (null otherwise)
  float* const rec = SAVE ? a.save + ((long)blockIdx.x * 4 + wave_s) * kSaveTileFloats + lane : nullptr;
  unsigned msk[4] = {0u, 0u, 0u, 0u};
  frag pe_h[4], pe_l[4], qe_h[2], qe_l[2];
  {
    float pe[32], qe[16];
    point_encodings(a, mc, h, pe, qe);
    if (SAVE) {
#pragma unroll
      for (int j = 0; j < 32; ++j) record_store(rec + (kSavePe + j) * 64, pe[j]);
#pragma unroll
      for (int j = 0; j < 16; ++j) record_store(rec + (kSaveQe + j) * 64, qe[j]);
    }
#pragma unroll
    for (int j = 0; j < 32; j += 2) {
      if (P::kCheckRange) amax = fmaxf(amax, fmaxf(fabsf(pe[j]), fabsf(pe[j + 1])));
      split_pair<P>(pe[j], pe[j + 1], pe_h[j >> 3], pe_l[j >> 3], j & 7);
    }
#pragma unroll
    for (int j = 0; j < 16; j += 2) split_pair<P>(qe[j], qe[j + 1], qe_h[j >> 3], qe_l[j >> 3], j & 7);
  }

  if constexpr (SAVE) {
  const float* const recu = a.save + ((long)blockIdx.x * 4 + wave_s) * kSaveTileFloats;      // wave-uniform save record
  const unsigned lb = (unsigned)lane * 4u;
  frag act_h[16], act_l[16], nxt_h[16], nxt_l[16];
  float sigma = 0.f;
  const float* const dw = cst + kConstDensW + h * 128;
  typedef EpiSplitSave<P, true> Relu;
  typedef EpiSplitSave<P, false> Lin;

  const_section_ready();      // mlps.0 reads its bias before its first stage barrier
  // ---- mlps.0: position encoding (4 k-steps) -> act
  layer_x3<P, 8, 4, 0>(c, cst + kConstBias8 + h * 128, pe_h, pe_l, pe_h, pe_l,
                       Relu{act_h, act_l, &amax, &sigma, dw, false, rec + save_h(1) * 64, msk, recu + save_h(1) * 64, lb});
  if (SAVE) flush_mask<4>(rec + save_mask(1) * 64, msk);
  // ---- mlps.1 .. mlps.7 in ping-pong pairs; mlps.5 (skip connection) also consumes the position encoding
  int l = 1;
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + l * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                          Relu{nxt_h, nxt_l, &amax, &sigma, dw, false, rec + (save_h(1) + 128 * l) * 64, msk, recu + (save_h(1) + 128 * l) * 64, lb});
    if (SAVE) flush_mask<4>(rec + (save_mask(1) + 4 * l) * 64, msk);
    ++l;
    layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + l * 256 + h * 128, nxt_h, nxt_l, pe_h, pe_l,
                          Relu{act_h, act_l, &amax, &sigma, dw, l == kDepth - 1,    // density from mlps.7 output
                               rec + (save_h(1) + 128 * l) * 64, msk, recu + (save_h(1) + 128 * l) * 64, lb});
    if (SAVE) flush_mask<4>(rec + (save_mask(1) + 4 * l) * 64, msk);
    ++l;
    if (it == 1) {
      layer_x3<P, 8, 16, 4>(c, cst + kConstBias8 + l * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                            Relu{nxt_h, nxt_l, &amax, &sigma, dw, false, rec + (save_h(1) + 128 * l) * 64, msk, recu + (save_h(1) + 128 * l) * 64, lb});
      if (SAVE) flush_mask<4>(rec + (save_mask(1) + 4 * l) * 64, msk);
      ++l;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        act_h[k] = nxt_h[k];
        act_l[k] = nxt_l[k];
      }
    }
  }
  // ---- density_linear (dm_nerf.py:101): fp32 partial sums were taken while mlps.7 was converted
  sigma += __shfl_xor(sigma, 32);
  sigma += cst[kConstDensB];

  // ---- rgb_feature_linear (no activation) -> nxt; rgb_feature_linears.0 on [feature, dirs] -> rgb_linear
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 8 * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                        Lin{nxt_h, nxt_l, &amax, &sigma, dw, false, rec + kSaveFeat * 64, msk, recu + kSaveFeat * 64, lb});
  float rgb[3] = {0.f, 0.f, 0.f};
  layer_x3<P, 4, 16, 2>(c, cst + kConstBias4 + h * 64, nxt_h, nxt_l, qe_h, qe_l,
                        EpiRgbSave{rgb, cst + kConstRgbW + h * 64, rec + kSaveR1 * 64, msk, recu + kSaveR1 * 64, lb});
  if (SAVE) flush_mask<2>(rec + kSaveMaskR1 * 64, msk);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rgb[k] += __shfl_xor(rgb[k], 32);
    rgb[k] += cst[kConstRgbB + k];
  }
  float* const out = valid ? a.raw + m * (4 + a.ins_ch) : nullptr;
  if (out != nullptr && h == 0) {
    out[0] = rgb[0];
    out[1] = rgb[1];
    out[2] = rgb[2];
    out[3] = sigma;
  }

  // ---- ins_feature_linear on the trunk output (no activation) -> nxt; ins_feature_linears.0 (ReLU) -> act[0..7]
  layer_x3<P, 8, 16, 0>(c, cst + kConstBias8 + 9 * 256 + h * 128, act_h, act_l, pe_h, pe_l,
                        Lin{nxt_h, nxt_l, &amax, &sigma, dw, false, rec + kSaveInsF * 64, msk, recu + kSaveInsF * 64, lb});
  layer_x3<P, 4, 16, 0>(c, cst + kConstBias4 + 128 + h * 64, nxt_h, nxt_l, pe_h, pe_l,
                        Relu{act_h, act_l, &amax, &sigma, dw, false, rec + kSaveI1 * 64, msk, recu + kSaveI1 * 64, lb});
  if (SAVE) flush_mask<2>(rec + kSaveMaskI1 * 64, msk);
  // ---- ins_linear: 128 -> ins_num + 1, T tiles, stored straight from the accumulators
  layer_x3<P, T, 8, 0>(c, cst + kConstBiasO + h * 64, act_h, act_l, pe_h, pe_l, EpiStore{out, a.ins_ch, h});

  } else {
    float sigma, rgb[3], insv[16 * T];
    x3_network<P, T>(c, cst, pe_h, pe_l, qe_h, qe_l, h, a.sigma_only != 0, amax, sigma, rgb, insv);
    // raw row of this lane's point: [rgb, sigma] by lane-half 0, the instance logits by both (dm_nerf.py:105)
    if (valid) {
      float* row = a.raw + m * (4 + a.ins_ch);
      if (h == 0) {
        row[0] = rgb[0]; row[1] = rgb[1]; row[2] = rgb[2];
        row[3] = sigma;
      }
#pragma unroll
      for (int i = 0; i < 16 * T; ++i) {
        const int ch = feat_of(i, 0) + 4 * h;
        if (ch < a.ins_ch) row[4 + ch] = insv[i];
      }
    }
    if (a.sigma_only) {
      if (P::kCheckRange && a.status != nullptr && !(amax < 60000.f)) atomicOr(a.status, DMN_STATUS_FP16_RANGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage's LDS-DMA must land before this LDS is released
      return;
    }
  }
#ifdef DMN_STAMPS
  {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (a.status != nullptr && lane == 0 && blockIdx.x % 97 == 0) {   // a sample of waves; status doubles as debug sink
      atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 1, t_end - t_begin);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 2, c.t_dma);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 3, c.t_bar);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 4, 1ull);
      atomicAdd(reinterpret_cast<unsigned long long*>(a.status) + 5, r_end - r_begin);
    }
  }
#else
  if (P::kCheckRange && a.status != nullptr && !(amax < 60000.f)) atomicOr(a.status, DMN_STATUS_FP16_RANGE);
#endif
}

template <class P, bool SAVE>
int launch_x3(const FieldArgs& a, int ins_num, hipStream_t st) {
  if (a.M <= 0) return DMN_OK;
  const long blocks = (a.M + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), block(256);
  switch (ins_tiles(ins_num)) {
    case 1: hipLaunchKernelGGL((field_fwd_x3_kernel<P, 1, SAVE>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((field_fwd_x3_kernel<P, 2, SAVE>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((field_fwd_x3_kernel<P, 3, SAVE>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((field_fwd_x3_kernel<P, 4, SAVE>), grid, block, 0, st, a); break;
    default: return DMN_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

int launch_prec(const FieldArgs& a, int ins_num, int precision, hipStream_t st) {
  if (ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  if (a.save != nullptr)      // training: bf16 only (no range guard to consult between forward and backward)
    return precision == DMN_PREC_BF16X3 ? launch_x3<PrecBF16, true>(a, ins_num, st) : DMN_ERR_BAD_ARG;
  if (precision == DMN_PREC_FP16X3) return launch_x3<PrecF16, false>(a, ins_num, st);
  if (precision == DMN_PREC_BF16X3) return launch_x3<PrecBF16, false>(a, ins_num, st);
  return DMN_ERR_BAD_ARG;
}

// blob gather with the hi / lo split; entry i < kConstFloats is an fp32 constant, the rest are halves
template <class E>
__global__ void pack_x3_kernel(const float* __restrict__ params, const int32_t* __restrict__ index, long index_len,
                               char* __restrict__ blob) {
  float* cst = reinterpret_cast<float*>(blob);
  E* stream = reinterpret_cast<E*>(blob + kConstFloats * 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < index_len; i += (long)gridDim.x * blockDim.x) {
    const int32_t k = index[i];
    const float v = k < 0 ? 0.f : params[k];
    if (i < kConstFloats) {
      cst[i] = v;
    } else {
      const long e = i - kConstFloats;
      const bool lo = ((e / kX3ChunkHalfs) & 1) != 0;      // chunks alternate hi, lo
      const E hi = (E)v;
      stream[e] = lo ? (E)(v - (float)hi) : hi;
    }
  }
}

}  // namespace

extern "C" int dmn_field_pack_x3_n(const float* params, const int32_t* index, long index_len, int precision,
                                   void* blob, void* stream) {
  if (!params || !index || !blob || index_len < kConstFloats) return DMN_ERR_BAD_ARG;
  const dim3 grid(2048), block(256);
  if (precision == DMN_PREC_FP16X3)
    hipLaunchKernelGGL(pack_x3_kernel<_Float16>, grid, block, 0, (hipStream_t)stream, params, index, index_len,
                       (char*)blob);
  else if (precision == DMN_PREC_BF16X3)
    hipLaunchKernelGGL(pack_x3_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, params, index, index_len,
                       (char*)blob);
  else
    return DMN_ERR_BAD_ARG;
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_field_pack_x3(const float* params, const int32_t* index, int precision, void* blob, void* stream) {
  return dmn_field_pack_x3_n(params, index, kX3IndexLen, precision, blob, stream);
}

extern "C" int dmn_field_fwd_rays_x3(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                     long n_rays, int n_samples, const void* blob, int ins_num, int precision,
                                     float* raw, int* status, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, status, n_rays * (long)n_samples, n_samples,
              ins_num + 1, nullptr};
  return launch_prec(a, ins_num, precision, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_rays_x3_sigma(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                           long n_rays, int n_samples, const void* blob, int ins_num, int precision,
                                           float* raw, int* status, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, status, n_rays * (long)n_samples, n_samples,
              ins_num + 1, nullptr, 1};
  return launch_prec(a, ins_num, precision, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_rays_x3_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                          long n_rays, int n_samples, const void* blob, int ins_num, int precision,
                                          float* raw, float* save, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw || !save) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, save};
  return launch_prec(a, ins_num, precision, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_embedded_x3(const float* embedded, long n_points, const void* blob, int ins_num,
                                         int precision, float* raw, int* status, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!embedded || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{nullptr, nullptr, nullptr, 0, embedded, blob, raw, status, n_points, 1, ins_num + 1};
  return launch_prec(a, ins_num, precision, (hipStream_t)stream);
}
