// Exact-fp32 MFMA machinery shared by the forward field kernel and the backward-data kernel: the weight
// stream (1 KiB chunks, 32 KiB stages, LDS-DMA double buffering, one barrier per stage) and the k-step macros
// that feed v_mfma_f32_32x32x2_f32 with A = weights from LDS and B = a register of the point tile.
#pragma once
#include "field_common.h"

#define DMN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace dmn {

constexpr int kLdsFloats = 2 * kStageFloats + kConstFloats;   // 80,896 B

// each wave DMA-copies 8 of the stage's 32 chunks
__device__ __forceinline__ void load_stage(const float* stream, int g, float* buf, int wave_s, int lane) {
  const float* src = stream + (long)g * kStageFloats + wave_s * 8 * kChunkFloats;      // wave-uniform
  float* dst = buf + wave_s * 8 * kChunkFloats;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    glds16_u(reinterpret_cast<const char*>(src + q * kChunkFloats), (unsigned)lane * 16u, dst + q * kChunkFloats);
}

// DMN_F32_SPREAD (set by a translation unit before including this file): the 8 pieces of the next stage are not issued
// in a burst behind the stage barrier (where the matrix pipe waits for them: ~4 % of a stage) but one per two k-steps,
// each right behind that k-step's MFMAs (DMN_STAGE8 / DMN_STAGE4 below).
#ifndef DMN_F32_SPREAD
#define DMN_F32_SPREAD 0
#endif

// Barrier that retires stage g (DMA landed, everybody done with stage g-1), then -- `burst` -- prefetch of
// stage g+1 into the other buffer.  Returns this lane's read base inside stage g.
__device__ __forceinline__ const float* stage_begin(const float* stream, int g, int G, float* lds, int wave_s,
                                                    int lane, bool burst = !DMN_F32_SPREAD) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of stage g (inline asm: not in the compiler's count)
  __syncthreads();
  if (burst && g + 1 < G) load_stage(stream, g + 1, lds + ((g + 1) & 1) * kStageFloats, wave_s, lane);
  return lds + (g & 1) * kStageFloats + lane * 4;
}

// piece q (0..7) of this wave's share of stage g_next; past the end of the stream the source clamps to the last stage
// and the copy lands in the buffer nobody reads any more (a select, not a branch: see field_x3.h load_piece)
__device__ __forceinline__ void load_piece32(const float* stream, int g_next, int G, float* lds, int wave_s, int lane,
                                             int q) {
  const int gs = g_next < G ? g_next : G - 1;
  const float* src = stream + (long)gs * kStageFloats + (wave_s * 8 + q) * kChunkFloats;
  float* dst = lds + (g_next & 1) * kStageFloats + (wave_s * 8 + q) * kChunkFloats;
  glds16_u(reinterpret_cast<const char*>(src), (unsigned)lane * 16u, dst);
}
#define DMN_PIECE(Q)                                                              \
  do {                                                                            \
    if (DMN_F32_SPREAD) load_piece32(stream, g + 1, G, lds, wave_s, lane, (Q));   \
  } while (0)

__device__ __forceinline__ void bias_tiles8(f32x16* acc, const float* b) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(b + t * 16 + v * 4);
      acc[t][4 * v + 0] = x.x; acc[t][4 * v + 1] = x.y; acc[t][4 * v + 2] = x.z; acc[t][4 * v + 3] = x.w;
    }
}
template <int NT>
__device__ __forceinline__ void bias_tiles(f32x16* acc, const float* b) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(b + t * 16 + v * 4);
      acc[t][4 * v + 0] = x.x; acc[t][4 * v + 1] = x.y; acc[t][4 * v + 2] = x.z; acc[t][4 * v + 3] = x.w;
    }
}

// 16 k-steps of an 8-tile layer out of one stage; B operand SRC(i) for k-step i
#define DMN_STAGE8(BUF, SRC, BASE)                                                        \
  {                                                                                       \
    f32x4 a0 = *reinterpret_cast<const f32x4*>((BUF));                                    \
    f32x4 a1 = *reinterpret_cast<const f32x4*>((BUF) + kChunkFloats);                     \
    _Pragma("unroll") for (int jj = 0; jj < 16; ++jj) {                                   \
      f32x4 n0 = a0, n1 = a1;                                                             \
      if (jj + 1 < 16) { /* LDS reads of the next k-step fly under this k-step's MFMAs */ \
        n0 = *reinterpret_cast<const f32x4*>((BUF) + (2 * jj + 2) * kChunkFloats);        \
        n1 = *reinterpret_cast<const f32x4*>((BUF) + (2 * jj + 3) * kChunkFloats);        \
      }                                                                                   \
      __builtin_amdgcn_sched_barrier(0); /* keep the prefetch above the MFMAs */          \
      const float bv = SRC((BASE) + jj);                                                  \
      acc[0] = DMN_MFMA(a0.x, bv, acc[0]);                                                \
      acc[1] = DMN_MFMA(a0.y, bv, acc[1]);                                                \
      acc[2] = DMN_MFMA(a0.z, bv, acc[2]);                                                \
      acc[3] = DMN_MFMA(a0.w, bv, acc[3]);                                                \
      acc[4] = DMN_MFMA(a1.x, bv, acc[4]);                                                \
      acc[5] = DMN_MFMA(a1.y, bv, acc[5]);                                                \
      acc[6] = DMN_MFMA(a1.z, bv, acc[6]);                                                \
      acc[7] = DMN_MFMA(a1.w, bv, acc[7]);                                                \
      if ((jj & 1) == 0) DMN_PIECE(jj >> 1);                                              \
      a0 = n0;                                                                            \
      a1 = n1;                                                                            \
    }                                                                                     \
  }
// NK k-steps of a 4-tile layer out of one stage
#define DMN_STAGE4(BUF, SRC, BASE, NK)                                                    \
  {                                                                                       \
    f32x4 a0 = *reinterpret_cast<const f32x4*>((BUF));                                    \
    f32x4 a1 = *reinterpret_cast<const f32x4*>((BUF) + kChunkFloats);                     \
    _Pragma("unroll") for (int jj = 0; jj < (NK); jj += 2) {                              \
      f32x4 n0 = a0, n1 = a1;                                                             \
      if (jj + 2 < (NK)) {                                                                \
        n0 = *reinterpret_cast<const f32x4*>((BUF) + (jj + 2) * kChunkFloats);            \
        n1 = *reinterpret_cast<const f32x4*>((BUF) + (jj + 3) * kChunkFloats);            \
      }                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                  \
      const float b0 = SRC((BASE) + jj);                                                  \
      const float b1 = SRC((BASE) + jj + 1);                                              \
      acc2[0] = DMN_MFMA(a0.x, b0, acc2[0]);                                              \
      acc2[1] = DMN_MFMA(a0.y, b0, acc2[1]);                                              \
      acc2[2] = DMN_MFMA(a0.z, b0, acc2[2]);                                              \
      acc2[3] = DMN_MFMA(a0.w, b0, acc2[3]);                                              \
      acc2[0] = DMN_MFMA(a1.x, b1, acc2[0]);                                              \
      acc2[1] = DMN_MFMA(a1.y, b1, acc2[1]);                                              \
      acc2[2] = DMN_MFMA(a1.z, b1, acc2[2]);                                              \
      acc2[3] = DMN_MFMA(a1.w, b1, acc2[3]);                                              \
      if (((jj >> 1) % ((NK) / 16)) == 0) DMN_PIECE((jj >> 1) / ((NK) / 16));             \
      a0 = n0;                                                                            \
      a1 = n1;                                                                            \
    }                                                                                     \
  }

#define SRC_ACT(i) act[(i)]
#define SRC_PE(i) pe[(i)]
#define SRC_QE(i) qe[(i)]
#define SRC_ACC(i) acc[(i) >> 4][(i) & 15]
#define SRC_ACC2(i) acc2[(i) >> 4][(i) & 15]


}  // namespace dmn
