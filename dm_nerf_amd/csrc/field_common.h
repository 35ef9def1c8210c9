// Shared by the exact (fp32 MFMA) and the split-precision (3 x 16-bit MFMA) field kernels:
// kernel arguments, LDS-DMA helper, and the per-lane prologue that turns a sample point into its
// encoding registers (pts = o + d*z, viewdirs = d/|d|, sin/cos bands; networks/render.py:37-38,49-58,
// networks/dm_nerf.py:17-32).
#pragma once
#include <hip/hip_runtime.h>

#include "field_layout.h"
#include "trig.h"

namespace dmn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FieldArgs {
  const float* rays_o;   // [N,3]   (rays mode)
  const float* rays_d;   // [N,3]
  const float* z;        // z[n*z_stride + s]
  long z_stride;
  const float* emb;      // [M,90]  (embedded mode; rays_* unused)
  const void* blob;
  float* raw;            // [M, C]
  int* status;           // sticky flags (bit 0: fp16 range exceeded); may be null
  long M;                // number of points
  int S;                 // samples per ray (rays mode)
  int ins_ch;            // ins_num + 1
  float* save;           // training: save record per 32-point tile (field_layout.h), or null
  int sigma_only;        // x3 kernels: stop behind the trunk, raw = [0, 0, 0, sigma, 0...] (coarse pass of an image loop)
};

// Record stores are non-temporal (`global_store_dword ... nt`): a training step streams 12 GB of save record and 12 GB
// of gradient record through each XCD's 4 MB L2, where plain stores evict the weight stream every workgroup re-reads
// (saving forward 5.45 -> 4.45 ms, backward-data 4.85 -> 3.95 ms per step; sc1 / sc0 sc1 stores and 16-byte stores
// at a quarter of the instruction count changed nothing: experiments/README.md).
__device__ __forceinline__ void record_store(float* p, float v) { __builtin_nontemporal_store(v, p); }
// register image access: register `reg` of this wave's tile record (wave-uniform base, one 256-byte row)
__device__ __forceinline__ void image_store(float* rec, int reg, int lane, float v) { record_store(rec + reg * 64 + lane, v); }
// 16 bytes of a record that is read once (the weight-gradient kernels): non-temporal as well
__device__ __forceinline__ f32x4 record_load4(const float* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
__device__ __forceinline__ float image_load(const float* rec, int reg, int lane) { return rec[reg * 64 + lane]; }

__device__ __forceinline__ void glds16(const void* g, void* l) {
  // LDS destination is wave-uniform base + lane*16 (hardware rule); g is per lane.
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// The constant section (biases, VALU head weights) is LDS-DMA'd by all four waves at kernel start; a wave's own vmcnt
// covers only its own chunks, so before anybody READS the section every wave waits for its pieces and the workgroup
// meets once.  (The stage barriers of the layer loops come after the first bias reads.)
__device__ __forceinline__ void const_section_ready() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
// The same with the global address as wave-uniform base (SGPR pair) + 32-bit lane offset (one VGPR, the same for every
// piece): `global_load_lds_dwordx4 v_off, s[base:base+1]`.  The base advances by scalar adds; the C builtin always takes
// the per-lane 64-bit address form (a v_lshl_add_u64 per piece).  Inline asm: the compiler does not count these loads
// in vmcnt, so whoever consumes the LDS bytes waits explicitly (stage_begin does: s_waitcnt vmcnt(0) before its barrier).
#ifndef DMN_GLDS_ASM
#define DMN_GLDS_ASM 1
#endif
__device__ __forceinline__ void glds16_u(const char* base_uniform, unsigned lane_bytes, void* l) {
#if DMN_GLDS_ASM
  const unsigned lds_addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)l;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               : : "s"(lds_addr), "v"(lane_bytes), "s"(base_uniform) : "m0");
#else
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)(base_uniform + (unsigned long)lane_bytes),
      (__attribute__((address_space(3))) void*)l, 16, 0, 0);
#endif
}

// The same for a piece at a compile-time byte offset OFF inside a block whose source (global, wave-uniform) and
// destination (LDS, wave-uniform) bases are given: the offset is added to both INSIDE the statement (scalar adds into a
// fixed SGPR pair and M0), so that per piece the compiler sees no address arithmetic at all -- given one expression per
// piece it computes hundreds of them at the top of a basic block and parks them in vector-register lanes.
// NT: the load carries the non-temporal cache policy (streamed lines are the L2's first candidates for replacement).
template <int OFF, bool NT = false>
__device__ __forceinline__ void glds16_at(const char* src_base_uniform, unsigned lane_bytes, unsigned lds_base_uniform) {
  const unsigned long long a = (unsigned long long)src_base_uniform;
  if (NT)
    asm volatile("s_add_i32 m0, %2, %4\n\ts_add_u32 s98, %0, %4\n\ts_addc_u32 s99, %1, 0\n\tglobal_load_lds_dwordx4 %3, s[98:99] nt"
                 : : "s"((unsigned)a), "s"((unsigned)(a >> 32)), "s"(lds_base_uniform), "v"(lane_bytes), "n"(OFF)
                 : "m0", "s98", "s99", "scc");
  else
    asm volatile("s_add_i32 m0, %2, %4\n\ts_add_u32 s98, %0, %4\n\ts_addc_u32 s99, %1, 0\n\tglobal_load_lds_dwordx4 %3, s[98:99]"
                 : : "s"((unsigned)a), "s"((unsigned)(a >> 32)), "s"(lds_base_uniform), "v"(lane_bytes), "n"(OFF)
                 : "m0", "s98", "s99", "scc");
}

// slots 2.. of an encoding register file: sin (half 0) / cos (half 1) of 2^f * coordinate, f = 0 .. NF-1, by the
// double-angle recurrence of trig.h from one fp64 sincos per coordinate
template <int NF>
__device__ __forceinline__ void encode_regs(float* e, float x, float y, float zc, int h) {
  e[0] = h ? zc : x;
  e[1] = h ? 0.f : y;
  double sn[3], cs[3];
  sincos_f64(x, sn[0], cs[0]);
  sincos_f64(y, sn[1], cs[1]);
  sincos_f64(zc, sn[2], cs[2]);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      e[2 + 3 * f + k] = (float)(h ? cs[k] : sn[k]);
      if (f + 1 < NF) double_angle(sn[k], cs[k]);
    }
  }
}

// Encoding registers of point mc for lane-half h: pe[32] (position, slots of pos_slot) and
// qe[16] (view direction, slots of dir_slot).
__device__ __forceinline__ void point_encodings(const FieldArgs& a, long mc, int h, float* pe, float* qe) {
  if (a.emb != nullptr) {
    const float* row = a.emb + mc * (kPosCh + kDirCh);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int s0 = pos_slot(j, 0), s1 = pos_slot(j, 1);
      const int s = h ? s1 : s0;
      pe[j] = s >= 0 ? row[s >= 0 ? s : 0] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int s0 = dir_slot(j, 0), s1 = dir_slot(j, 1);
      const int s = h ? s1 : s0;
      qe[j] = s >= 0 ? row[kPosCh + (s >= 0 ? s : 0)] : 0.f;
    }
  } else {
    const long n = mc / a.S;
    const int s = (int)(mc - n * a.S);
    const float ox = a.rays_o[3 * n + 0], oy = a.rays_o[3 * n + 1], oz = a.rays_o[3 * n + 2];
    const float dx = a.rays_d[3 * n + 0], dy = a.rays_d[3 * n + 1], dz = a.rays_d[3 * n + 2];
    const float zv = a.z[n * a.z_stride + s];
    // pts = rays_o + rays_d * z: separately rounded multiply and add, as torch does (render.py:49).
    // The 2^9 band magnifies a 1-ulp difference in pts to ~5e-4 rad, so no fma contraction here.
    const float px = __fadd_rn(ox, __fmul_rn(dx, zv));
    const float py = __fadd_rn(oy, __fmul_rn(dy, zv));
    const float pz = __fadd_rn(oz, __fmul_rn(dz, zv));
    // viewdirs = rays_d / ||rays_d|| (render.py:37)
    const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    const float vx = __fdiv_rn(dx, nrm), vy = __fdiv_rn(dy, nrm), vz = __fdiv_rn(dz, nrm);
    encode_regs<kPosFreqs>(pe, px, py, pz, h);
    encode_regs<kDirFreqs>(qe, vx, vy, vz, h);
    qe[14] = 0.f;
    qe[15] = 0.f;
  }
}

}  // namespace dmn
