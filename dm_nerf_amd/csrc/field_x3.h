// Split-precision MFMA machinery shared by the forward field kernel (field_fwd_x3.hip) and the backward-data
// kernel (field_bwd_x3.hip): precision traits, the 64 KiB-stage weight stream with its spread-out LDS-DMA, the
// hi/lo split of a value pair, and the tile-outer layer loop  acc(tile) = sum over k-steps of
// Ahi*Blo + Alo*Bhi + Ahi*Bhi  with the previous tile's epilogue running in the MFMA shadow.
#pragma once
#include "field_common.h"

// Timing-only ablation switches (never defined in the product build; results are wrong with them):
//   DMN_ABL_NODMA  the weight stream stops after the first two stages (LDS keeps realistic data)
//   DMN_ABL_NOLDS  A fragments are read from LDS only in the first stage of a kernel, then reused
#ifndef DMN_ABL_NODMA
#define DMN_ABL_NODMA 0
#endif
#ifndef DMN_ABL_NOLDS
#define DMN_ABL_NOLDS 0
#endif
#ifndef DMN_X3_PREFETCH
#define DMN_X3_PREFETCH 2
#endif
#ifndef DMN_X3_SCALAR_SUB
#define DMN_X3_SCALAR_SUB 0
#endif

namespace dmn {

struct PrecF16 {
  typedef _Float16 elem;
  typedef _Float16 frag __attribute__((ext_vector_type(8)));
  static constexpr bool kCheckRange = true;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  typedef _Float16 pair __attribute__((ext_vector_type(2)));
  typedef float float2v __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ float2v widen(pair p) { return __builtin_convertvector(p, float2v); }
};
struct PrecBF16 {
  typedef __bf16 elem;
  typedef __bf16 frag __attribute__((ext_vector_type(8)));
  static constexpr bool kCheckRange = false;
  static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  typedef __bf16 pair __attribute__((ext_vector_type(2)));
  typedef float float2v __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ float2v widen(pair p) {   // bf16 -> f32 is a shift / a mask of the packed register
    const unsigned u = __builtin_bit_cast(unsigned, p);
    return (float2v){__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
  }
};

constexpr int kLdsBytes = 2 * kX3StageBytes + kConstFloats * 4;   // 146,432 B

struct Ctx {
  const char* stream;   // weight stream in global memory
  char* lds;
  const char* buf;      // this lane's read base inside the current stage
  int g, G;             // next stage to begin, number of stages
  int wave_s, lane;
#ifdef DMN_STAMPS
  unsigned long long t_dma = 0, t_bar = 0;
#endif
};

// each wave DMA-copies 16 of the stage's 64 chunks
__device__ __forceinline__ void load_stage(const Ctx& c, int g) {
  const char* src = c.stream + (long)g * kX3StageBytes + (c.wave_s * 16) * 1024;
  char* dst = c.lds + (g & 1) * kX3StageBytes + (c.wave_s * 16) * 1024;
#pragma unroll
  for (int q = 0; q < 16; ++q) glds16_u(src + q * 1024, (unsigned)c.lane * 16u, dst + q * 1024);
}

// Barrier that retires stage g (its DMA has landed, every wave is done with stage g-1).  The DMA of stage
// g+1 into the other buffer is NOT issued here: 4 waves x 16 pieces right behind the barrier queue up on the
// CU's one address unit (16 cycles per 1 KiB piece) while the MFMAs wait in order behind them -- a quarter of
// the kernel when measured.  load_piece() below spreads them one per step over the first half of the stage.
__device__ __forceinline__ void stage_begin(Ctx& c) {
#ifdef DMN_STAMPS   // diagnostic build only (tools/stamps.sh): cycles spent waiting at the stage barrier
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  c.t_dma += t1 - t0;
  c.t_bar += t2 - t1;
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces (inline asm: not in the compiler's count)
  __syncthreads();
#endif
  c.buf = c.lds + (c.g & 1) * kX3StageBytes + c.lane * 16;
  ++c.g;
}

// piece q (0..15) of this wave's share of the NEXT stage (index c.g after stage_begin).  Past the end of the
// stream the source clamps to the last stage and the copy lands in the buffer nobody reads any more.  (Skipping those
// copies with a wave-uniform `if (c.g >= c.G) return;` was tried: where the stage counter is not a compile-time
// constant it becomes 557 branches in the step loop, the kernels need 512 registers + 600 B of spills, -7 %.)
__device__ __forceinline__ void load_piece(const Ctx& c, int q) {
  if (DMN_ABL_NODMA && c.g >= 2) return;
  const int g = c.g < c.G ? c.g : c.G - 1;
  const char* src = c.stream + (long)g * kX3StageBytes + (c.wave_s * 16 + q) * 1024;
  char* dst = c.lds + (c.g & 1) * kX3StageBytes + (c.wave_s * 16 + q) * 1024;
  glds16_u(src, (unsigned)c.lane * 16u, dst);
}

// PIN keeps the conversion where it is written (in the MFMA shadow of the next tile): without it the compiler
// sinks all 128 conversions of a layer to their first use in the next layer and keeps 8 accumulators alive.
template <class P, bool PIN = false>
__device__ __forceinline__ void split_pair(float x0, float x1, typename P::frag& hi, typename P::frag& lo, int j) {
  typedef typename P::elem E;
  typedef E pair __attribute__((ext_vector_type(2)));
  typedef float float2v __attribute__((ext_vector_type(2)));
  // one packed conversion per pair (v_cvt_pk_{f16,bf16}_f32), and the rounded values back from the packed register:
  // written with scalar casts the compiler converts every value twice
  const float2v x = {x0, x1};
  pair hp = __builtin_convertvector(x, pair);
  if (PIN) asm volatile("" : "+v"(hp));
#if DMN_X3_SCALAR_SUB
  // two single subtractions: written as a vector difference the compiler emits v_pk_add_f32, which beside MFMAs costs
  // ~13 cycles more than the pair of scalar instructions (MI355X_MICROARCH.md, packed-f32 fillers)
  const float2v wide = P::widen(hp);
  float r0, r1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(wide[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(wide[1]));
  const float2v res = {r0, r1};
  pair lp = __builtin_convertvector(res, pair);
#else
  pair lp = __builtin_convertvector(x - P::widen(hp), pair);          // v_pk_add_f32 with negated second operand
#endif
  if (PIN) asm volatile("" : "+v"(lp));
  hi[j] = hp[0];
  hi[j + 1] = hp[1];
  lo[j] = lp[0];
  lo[j + 1] = lp[1];
}

// ReLU as a signed-integer max on the bit pattern: one v_max_i32, where fmaxf / v_med3 cost a canonicalising
// v_max_f32 x, x first.  Negative values (and -0, negative NaNs) give +0, everything else is unchanged.
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// Element r of a FINISHED accumulator tile, read out of its AGPR where it is used (only behind at least one full
// k-step of MFMAs on the other tile, see layer_x3): left to itself the compiler copies the
// whole finished tile out behind the tile's last MFMA (s_nop for the result latency + 16 v_accvgpr_read with the
// matrix pipe idle) and reuses the same AGPRs for the next tile; the "a" constraint keeps the finished tile where
// it is, the next tile accumulates in the other tuple, and the copies ride in the MFMA shadow two per step.
__device__ __forceinline__ float acc_read(const f32x16& acc, int r) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[r]));
  return x;
}

// One layer: NT output tiles x (NKS + NKX) k-steps of 16; xh/xl: main operand fragments,
// eh/el: extra fragments (position / direction encodings) appended to the k range.
template <class P, int NT, int NKS, int NKX, class Epi>
__device__ __forceinline__ void layer_x3(Ctx& c, const float* bias, const typename P::frag* xh,
                                         const typename P::frag* xl, const typename P::frag* eh,
                                         const typename P::frag* el, const Epi& epi) {
  typedef typename P::frag frag;
  constexpr int NK = NKS + NKX;
  constexpr int STEPS = NT * NK;
  static_assert(NK >= 2, "the epilogue of a tile is spread over k-steps 1.. of the next one");
  // epilogue pieces per k-step, over k-steps 1 .. NK-1 of the next tile: k-step 0 (three MFMAs on the OTHER
  // accumulator, >= 96 cycles) covers the result latency of the finished tile's last MFMA.  acc_read() is inline
  // asm, for which the compiler inserts no MFMA -> VALU hazard nops; reading earlier returns a stale accumulator.
  constexpr int PPS = (8 + NK - 2) / (NK - 1);
  f32x16 acc[2];
  constexpr int PF = DMN_X3_PREFETCH;       // A-fragment prefetch distance in steps; ring of PF + 1
  frag ah[PF + 1], al[PF + 1];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {          // bias = initial accumulator
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + t * 16 + v * 4);
      acc[t & 1][4 * v + 0] = b.x; acc[t & 1][4 * v + 1] = b.y; acc[t & 1][4 * v + 2] = b.z; acc[t & 1][4 * v + 3] = b.w;
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int step = t * NK + ks;
      if ((step & 31) == 0) {              // 32 steps = 64 chunks = one stage
        stage_begin(c);
#pragma unroll
        for (int d = 0; d < PF; ++d)
          if (step + d < STEPS) {
            ah[(step + d) % (PF + 1)] = *reinterpret_cast<const frag*>(c.buf + ((2 * (step + d)) & 63) * 1024);
            al[(step + d) % (PF + 1)] = *reinterpret_cast<const frag*>(c.buf + ((2 * (step + d) + 1) & 63) * 1024);
          }
      }
      if (step + PF < STEPS && ((step + PF) >> 5) == (step >> 5) && !(DMN_ABL_NOLDS && c.g > 1)) {   // PF steps ahead, same stage
        ah[(step + PF) % (PF + 1)] = *reinterpret_cast<const frag*>(c.buf + ((2 * (step + PF)) & 63) * 1024);
        al[(step + PF) % (PF + 1)] = *reinterpret_cast<const frag*>(c.buf + ((2 * (step + PF) + 1) & 63) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above this step's MFMAs
      if (t > 0 && ks > 0) {
#pragma unroll
        for (int q = (ks - 1) * PPS; q < ks * PPS && q < 8; ++q)
          epi(t - 1, q, acc_read(acc[(t - 1) & 1], 2 * q), acc_read(acc[(t - 1) & 1], 2 * q + 1));
      }
      const frag bh = ks < NKS ? xh[ks < NKS ? ks : 0] : eh[ks < NKS ? 0 : ks - NKS];
      const frag bl = ks < NKS ? xl[ks < NKS ? ks : 0] : el[ks < NKS ? 0 : ks - NKS];
      acc[t & 1] = P::mfma(ah[step % (PF + 1)], bl, acc[t & 1]);
      acc[t & 1] = P::mfma(al[step % (PF + 1)], bh, acc[t & 1]);
      acc[t & 1] = P::mfma(ah[step % (PF + 1)], bh, acc[t & 1]);
      if ((step & 31) < 16) load_piece(c, step & 31);     // one DMA piece per step, behind this step's MFMAs
    }
  }
  if ((STEPS & 31) != 0 && (STEPS & 31) < 16) {          // a short last stage still owes the rest of its pieces
#pragma unroll
    for (int q = (STEPS & 31); q < 16; ++q) load_piece(c, q);
  }
  // the last tile has no successor to hide behind: plain register reads, so that the compiler sees the MFMA -> VALU
  // dependency and inserts the wait states itself
#pragma unroll
  for (int q = 0; q < 8; ++q) epi(NT - 1, q, acc[(NT - 1) & 1][2 * q], acc[(NT - 1) & 1][2 * q + 1]);
}


}  // namespace dmn
