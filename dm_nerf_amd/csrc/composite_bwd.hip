// Backward of the alpha compositing (networks/render.py:6-28 under autograd; manipulator variant
// networks/manipulator.py:86-105) for gfx950: d(loss)/d(raw) from the gradients of rgb_map, depth_map,
// ins_map and (optionally) weights.  One wavefront per ray, same slab walk as the forward kernel.
//
//   w_i = a_i T_i,  T_i = prod_{j<i} f_j,  f_j = 1 - a_j + 1e-10,  a_i = 1 - exp(-relu(s_i) d_i)
//   G_i = dL/dw_i = sum_c g_rgb_c sigmoid(raw_ic) + g_depth z_i + g_w_i      (ins_map uses w.detach(), render.py:22-23)
//   dL/da_i = G_i T_i - (sum_{j>i} G_j w_j) / f_i                            -> reverse scan across the wavefront
//   dL/ds_i = [s_i > 0] d_i (1 - a_i) dL/da_i
//   dL/draw_ic   = g_rgb_c w_i sig(1 - sig)                   c < 3
//   dL/draw_i,4+k = g_ins_k ins_k (1 - ins_k) w_i             k < K  (K = C-5, or C-4 with keep_last)
//
// Pass 1 (forward over the slabs) recomputes a_i and T_i with the forward kernel's fp64 product scan and keeps
// them in LDS; pass 2 (backward over the slabs) stages each raw slab in LDS, turns it into the gradient slab
// in place and writes it back with wave-wide contiguous stores.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"

namespace {

constexpr int kWaves = 4;

template <bool PRODUCT>
__device__ __forceinline__ double scan_up(double v, int lane) {      // inclusive, lanes 0 -> 63
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_up(v, d);
    if (lane >= d) v = PRODUCT ? v * t : v + t;
  }
  return v;
}
__device__ __forceinline__ double scan_down_sum(double v, int lane) {  // inclusive suffix sum, lanes 63 -> 0
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_down(v, d);
    if (lane + d < 64) v += t;
  }
  return v;
}

struct BwdArgs {
  const float* raw;
  const float* z;
  long zs;
  const float* rays_d;
  const float* ins;        // forward ins_map [N,K] (post-sigmoid), may be null when g_ins is null
  const float* g_rgb;      // [N,3] or null
  const float* g_depth;    // [N] or null
  const float* g_ins;      // [N,K] or null
  const float* g_w;        // [N,S] or null
  long n_rays;
  int S, C, keep_last;
  float* g_raw;            // [N,S,C]
};

__global__ __launch_bounds__(64 * kWaves) void composite_bwd_kernel(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C, S = a.S;
  const int K = a.keep_last ? C - 4 : C - 5;
  // per wave: slab[64*C] | alpha[S] | trans[S] | gi[C]
  const int per_wave = 64 * C + 2 * S + C;
  float* slab = smem + (long)wave * per_wave;
  float* al = slab + 64 * C;
  float* tr = al + S;
  float* gi = tr + S;
  const long n = (long)blockIdx.x * kWaves + wave;
  const bool ray_ok = n < a.n_rays;
  const long nc = ray_ok ? n : a.n_rays - 1;

  const float dx = a.rays_d[3 * nc + 0], dy = a.rays_d[3 * nc + 1], dz = a.rays_d[3 * nc + 2];
  const float norm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zr = a.z + nc * a.zs;
  const float* rawr = a.raw + nc * (long)S * C;
  const int n_slabs = (S + 63) / 64;

  // ---- pass 1: alpha and transmittance of every sample (identical arithmetic to the forward kernel)
  double carry = 1.0;
  for (int b = 0; b < n_slabs; ++b) {
    const int s = b * 64 + lane;
    const bool ok = s < S;
    const float zc = ok ? zr[s] : 0.f;
    float delta = 1e10f;
    if (ok && s + 1 < S) delta = __fsub_rn(zr[s + 1], zc);
    delta = __fmul_rn(delta, norm);
    const float sigma = ok ? rawr[(long)s * C + 3] : 0.f;
    const float alpha = ok ? __fsub_rn(1.f, expf(-__fmul_rn(fmaxf(sigma, 0.f), delta))) : 0.f;
    const double f = ok ? (double)__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f) : 1.0;
    const double incl = scan_up<true>(f, lane);
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0;
    if (ok) {
      al[s] = alpha;
      tr[s] = (float)(carry * excl);
    }
    carry = carry * __shfl(incl, 63);
  }
  // per-channel ray constants of the instance head: g_ins_k * ins_k * (1 - ins_k)
  for (int c = lane; c < C; c += 64) {
    float v = 0.f;
    const int k = c - 4;
    if (a.g_ins && k >= 0 && k < K) {
      const float p = a.ins[nc * K + k];
      v = a.g_ins[nc * K + k] * p * (1.f - p);
    }
    gi[c] = v;
  }
  const float gr0 = a.g_rgb ? a.g_rgb[3 * nc + 0] : 0.f;
  const float gr1 = a.g_rgb ? a.g_rgb[3 * nc + 1] : 0.f;
  const float gr2 = a.g_rgb ? a.g_rgb[3 * nc + 2] : 0.f;
  const float gd = a.g_depth ? a.g_depth[nc] : 0.f;

  // ---- pass 2: slabs in reverse, suffix sums of G_j w_j carried from the far end of the ray
  double tail = 0.0;
  for (int b = n_slabs - 1; b >= 0; --b) {
    const int s0 = b * 64;
    const int cnt = min(64, S - s0);
    const int nfl = cnt * C;
    const float* src = rawr + (long)s0 * C;
    __syncthreads();                       // previous slab written out (uniform trip count across the block)
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int nv = nfl >> 2;
      for (int i = lane; i < nv; i += 64) reinterpret_cast<float4*>(slab)[i] = reinterpret_cast<const float4*>(src)[i];
      for (int i = (nv << 2) + lane; i < nfl; i += 64) slab[i] = src[i];
    } else {
      for (int i = lane; i < nfl; i += 64) slab[i] = src[i];
    }
    __syncthreads();

    const int s = s0 + lane;
    const bool ok = lane < cnt;
    float* row = slab + lane * C;
    const float zc = ok ? zr[s] : 0.f;
    float delta = 1e10f;
    if (ok && s + 1 < S) delta = __fsub_rn(zr[s + 1], zc);
    delta = __fmul_rn(delta, norm);
    const float alpha = ok ? al[s] : 0.f;
    const float T = ok ? tr[s] : 0.f;
    const float w = alpha * T;
    const float sigma = ok ? row[3] : 0.f;
    float sg[3] = {0.f, 0.f, 0.f};
    if (ok) {
#pragma unroll
      for (int c = 0; c < 3; ++c) sg[c] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-row[c])));
    }
    float G = gr0 * sg[0] + gr1 * sg[1] + gr2 * sg[2] + gd * zc;
    if (a.g_w && ok) G += a.g_w[nc * S + s];
    const double u = ok ? (double)G * (double)w : 0.0;
    const double incl = scan_down_sum(u, lane);              // sum_{j >= i in slab}
    const double after = tail + (incl - u);                  // sum_{j > i} over the rest of the ray
    tail += __shfl(incl, 0);
    const float f = __fadd_rn(__fsub_rn(1.f, alpha), 1e-10f);
    const float d_alpha = (float)((double)G * (double)T - after / (double)f);
    const float d_sigma = sigma > 0.f ? d_alpha * delta * (1.f - alpha) : 0.f;
    if (ok) {
      row[0] = gr0 * w * sg[0] * (1.f - sg[0]);
      row[1] = gr1 * w * sg[1] * (1.f - sg[1]);
      row[2] = gr2 * w * sg[2] * (1.f - sg[2]);
      row[3] = d_sigma;
      for (int c = 4; c < C; ++c) row[c] = gi[c] * w;
    }
    __syncthreads();
    if (ray_ok) {
      float* dst = a.g_raw + (nc * (long)S + s0) * C;
      if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int nv = nfl >> 2;
        for (int i = lane; i < nv; i += 64) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(slab)[i];
        for (int i = (nv << 2) + lane; i < nfl; i += 64) dst[i] = slab[i];
      } else {
        for (int i = lane; i < nfl; i += 64) dst[i] = slab[i];
      }
    }
  }
}

}  // namespace

extern "C" int dmn_composite_bwd(const float* raw, const float* z, long z_ray_stride, const float* rays_d,
                                 const float* ins, const float* g_rgb, const float* g_depth, const float* g_ins,
                                 const float* g_weights, long n_rays, int n_samples, int channels, int keep_last,
                                 float* g_raw, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || channels < 5 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!raw || !z || !rays_d || !g_raw || (g_ins && !ins)) return DMN_ERR_BAD_ARG;
  if (channels > 192) return DMN_ERR_UNSUPPORTED;
  const size_t lds = (size_t)kWaves * (64 * channels + 2 * n_samples + channels) * sizeof(float);
  if (lds > 160 * 1024) return DMN_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)composite_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return DMN_ERR_LAUNCH;
  BwdArgs a{raw, z, z_ray_stride, rays_d, ins, g_rgb, g_depth, g_ins, g_weights, n_rays, n_samples, channels,
            keep_last, g_raw};
  const long blocks = (n_rays + kWaves - 1) / kWaves;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)blocks), dim3(64 * kWaves), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
