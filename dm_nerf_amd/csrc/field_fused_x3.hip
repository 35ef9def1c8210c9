// Field evaluation + alpha compositing (+ inverse-CDF resampling) in ONE kernel for the image loop
// (networks/tester.py:62-76 keeps only rgb_fine / ins_fine of every chunk): `raw` never reaches HBM.
//
//   coarse pass:  field(o + d z_coarse) -> composite -> weights -> sample_pdf + sort -> z_fine        (render.py:49-70)
//   fine pass:    field(o + d z_fine)   -> composite -> rgb_map, depth_map, ins_map                   (render.py:71-86)
//
// A workgroup owns TWO rays.  It walks them in passes of 64 samples per ray: the 4 waves evaluate the network on
// 2 x 64 = 128 points exactly as the stand-alone field kernel of the same arithmetic does -- field_fwd_x3_kernel /
// x3_network() for the two-way splits, field_fwd_x6_kernel / x6_network() for bf16x6: same weight stream, same device
// function, so the same bits --,
// park the 128 raw rows in LDS -- in the weight stage buffers, which are idle between the last layer of a pass and
// the first stage of the next -- and the ray's two waves composite its 64-sample slab with composite_slab(), the
// function the stand-alone compositing kernels run (half of the channels each).  Transmittance and channel sums are
// carried in registers from pass to pass, so a ray of 192 samples is three passes of one workgroup; the weights of 128 points
// are streamed once per pass, as in the unfused kernel (no extra L2 -> LDS traffic).  Results are bit-identical to
// dmn_field_fwd_rays_x3 + dmn_composite_fwd (+ dmn_resample_merge): tests/test_gpu_parity.py.
//
// Saved per ray at C = 18 (DM-SR): 64 x 18 x 4 B written + read back per pass, two kernel launches per chunk; at
// C = 98 (Replica) the raw rows are 25 KB per ray and pass and the scattered row stores of the unfused kernel go away.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "composite_common.h"
#include "field_common.h"
#include "field_x3.h"
#include "field_x3_net.h"
#include "field_x6.h"

using namespace dmn;

// timing-only ablations (never in the product build; results are wrong with them): 1 = no compositing arithmetic,
// 2 = nothing between the network and the next pass (no slab, no barriers)
#ifndef DMN_FUSED_ABL
#define DMN_FUSED_ABL 0
#endif

namespace {

constexpr int kFusedExtraBytes = 16 * 1024;     // weights of the ray + resampling scratch, outside the stage buffers

// Which field arithmetic a fused kernel runs: its weight-stream context, its LDS geometry ([stage slots | constants],
// the slots double as the parking area of the raw rows between two passes) and the network on a wave's 32 points.
template <class P>
struct NetX3 {
  typedef Ctx ctx_t;
  static constexpr int kLds = kLdsBytes, kStageArea = 2 * kX3StageBytes;
  static constexpr bool kCheckRange = P::kCheckRange;
  template <int T>
  static __device__ __forceinline__ ctx_t begin(const char* blob, char* lds, int wave_s, int lane) {
    Ctx c{blob + kConstFloats * 4, lds, nullptr, 0, kX3Stages, wave_s, lane};
    load_stage(c, 0);
    return c;
  }
  template <int T>
  static __device__ __forceinline__ void run(ctx_t& c, const FieldArgs& f, long mc, int h, const float* cst, float& amax,
                                             float& sigma, float (&rgb)[3], float (&insv)[16 * T]) {
    typename P::frag pe_h[4], pe_l[4], qe_h[2], qe_l[2];
    x3_point_fragments<P>(f, mc, h, amax, pe_h, pe_l, qe_h, qe_l);
    x3_network<P, T>(c, cst, pe_h, pe_l, qe_h, qe_l, h, f.sigma_only != 0, amax, sigma, rgb, insv);
  }
};
struct NetX6 {
  typedef Ctx6 ctx_t;
  static constexpr int kLds = kX6LdsBytes, kStageArea = kX6Slots * kX6StageBytes;
  static constexpr bool kCheckRange = false;
  template <int T>
  static __device__ __forceinline__ ctx_t begin(const char* blob, char* lds, int wave_s, int lane) {
    Ctx6 c{blob + kConstFloats * 4, lds, nullptr, 0, x6_total_stages(T), wave_s, lane};
    x6_load_stage(c, 0);
    if (kX6Slots == 3) x6_load_stage(c, 1);
    return c;
  }
  template <int T>
  static __device__ __forceinline__ void run(ctx_t& c, const FieldArgs& f, long mc, int h, const float* cst, float& amax,
                                             float& sigma, float (&rgb)[3], float (&insv)[16 * T]) {
    float pe[32], qe[16];
    point_encodings(f, mc, h, pe, qe);
    if (x6_network<T, false>(c, cst, pe, qe, h, f.sigma_only != 0, nullptr, sigma, rgb, insv)) {
      rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;
#pragma unroll
      for (int i = 0; i < 16 * T; ++i) insv[i] = 0.f;
    }
  }
};

struct FusedArgs {
  FieldArgs f;           // rays, depths, blob, status, S, ins_ch, sigma_only (raw / emb / save unused)
  long n_rays;
  float* rgb;            // [N, 3]      any of the four may be null
  float* depth;          // [N]
  float* ins;            // [N, ins_num]   (render_train drops the "air" channel, render.py:26)
  float* weights;        // [N, S]
  const float* u;        // resampling (z_fine != null): draws u[n * u_stride + k] or null (linspace)
  long u_stride;
  int n_new;
  float* z_fine;         // [N, S + n_new]
};

template <class Net, int T>
__global__ __launch_bounds__(256, 1) void field_fused_x3_kernel(const FusedArgs a) {
  static_assert(Net::kLds + kFusedExtraBytes <= 160 * 1024, "LDS budget of one CU");
  __shared__ __attribute__((aligned(16))) char lds[Net::kLds + kFusedExtraBytes];
  float* const cst = reinterpret_cast<float*>(lds + Net::kStageArea);
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave_s >> 1, part = wave_s & 1;        // ray of the workgroup, half of its 64-sample slab
  const char* const blob = static_cast<const char*>(a.f.blob);
  for (int k = wave_s; k < kConstChunks; k += 4) glds16(blob + k * 1024 + lane * 16, lds + Net::kStageArea + k * 1024);

  const int S = a.f.S, C = 4 + a.f.ins_ch;
  const long ray = (long)blockIdx.x * 2 + pair;
  const bool ray_ok = ray < a.n_rays;
  const long rayc = ray_ok ? ray : a.n_rays - 1;
  const float norm = ray_norm(a.f.rays_d + 3 * rayc);
  const float* const zr = a.f.z + rayc * a.f.z_stride;
  float* const slab = reinterpret_cast<float*>(lds) + (long)pair * 64 * C;            // in the stage buffers
  float* const extra = reinterpret_cast<float*>(lds + Net::kLds) + (long)pair * (kFusedExtraBytes / 8);
  float* const wts = extra;                                                            // [S] weights of this ray
  const bool maps = a.rgb != nullptr || a.depth != nullptr || a.ins != nullptr;

  // Both waves of a ray composite its slab: each runs the (cheap) transmittance scan and takes half of the channels of
  // the lane-per-sample form (every channel sum is its own chain: no bit changes); the lane-per-channel form of wide
  // rows is left to wave 0 of the pair, its sums run over the samples in order.
  const bool wide = C > kWideChannels;
  const int c_lo = wide ? 0 : (part == 0 ? 0 : (C + 1) / 2), c_hi = wide ? C : (part == 0 ? (C + 1) / 2 : C);
  const bool composites = !wide || part == 0;
  RayComposite st;
  float amax = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    typename Net::ctx_t c = Net::template begin<T>(blob, lds, wave_s, lane);
    // depths of the compositing lane's sample and of the next one: asked for now, used after the network
    const float zc = zr[s0 + lane];
    const float zn = s0 + lane + 1 < S ? zr[s0 + lane + 1] : 0.f;
    const long mc = rayc * S + s0 + 32 * part + p;
    float sigma, rgb[3], insv[16 * T];
    Net::template run<T>(c, a.f, mc, h, cst, amax, sigma, rgb, insv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA has landed
#if DMN_FUSED_ABL == 2
    if (sigma + rgb[0] + insv[0] == 12345.f) slab[lane] = sigma;      // keep the network alive
    continue;
#endif
    __syncthreads();                                    // every wave is done with the weights: the stage buffers are free
    {
      float* row = slab + (32 * part + p) * C;
      if (h == 0) {
        row[0] = rgb[0]; row[1] = rgb[1]; row[2] = rgb[2];
        row[3] = sigma;
      }
#pragma unroll
      for (int i = 0; i < 16 * T; ++i) {
        const int ch = feat_of(i, 0) + 4 * h;
        if (ch < a.f.ins_ch) row[4 + ch] = insv[i];
      }
    }
    __syncthreads();
    if (composites && DMN_FUSED_ABL != 1) {
      const float w = wide ? composite_slab<true>(st, slab, zc, zn, s0, 64, S, C, norm, lane, maps)
                           : composite_slab<false>(st, slab, zc, zn, s0, 64, S, C, norm, lane, maps, c_lo, c_hi);
      if (part == 0) {
        if (ray_ok && a.weights != nullptr) a.weights[ray * S + s0 + lane] = w;
        if (a.z_fine != nullptr) wts[s0 + lane] = w;
      }
    }
    __syncthreads();                                    // slab consumed: the next pass may stream weights over it
  }
  if (Net::kCheckRange && a.f.status != nullptr && !(amax < 60000.f)) atomicOr(a.f.status, DMN_STATUS_FP16_RANGE);
  if (!ray_ok) return;
  if (composites && maps) composite_outputs(st, ray, C, 0, lane, a.rgb, a.depth, a.ins, c_lo, c_hi);
  if (part == 0 && a.z_fine != nullptr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int L = S + a.n_new;
    float* all = extra + S;                              // [L], then cdf [S], bins [S]
    resample_merge_ray(zr, wts, a.u ? a.u + ray * a.u_stride : nullptr, S, a.n_new, all, all + L, all + L + S,
                       a.z_fine + ray * L, lane);
  }
}

template <class P>
int launch_fused(const FusedArgs& a, int ins_num, hipStream_t st) {
  if (2L * 64 * (5 + ins_num) * 4 > P::kStageArea) return DMN_ERR_UNSUPPORTED;      // the two rays' raw rows of a pass
  const long blocks = (a.n_rays + 1) / 2;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), block(256);
  switch (ins_tiles(ins_num)) {
    case 1: hipLaunchKernelGGL((field_fused_x3_kernel<P, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((field_fused_x3_kernel<P, 2>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((field_fused_x3_kernel<P, 3>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((field_fused_x3_kernel<P, 4>), grid, block, 0, st, a); break;
    default: return DMN_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

}  // namespace

extern "C" int dmn_field_composite_x3(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                      long n_rays, int n_samples, const void* blob, int ins_num, int precision,
                                      int sigma_only, float* rgb, float* depth, float* ins, float* weights,
                                      const float* u, long u_ray_stride, int n_new, float* z_fine, int* status,
                                      void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0 || u_ray_stride < 0 || n_new < 0)
    return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob) return DMN_ERR_BAD_ARG;
  if (n_samples % 64 != 0) return DMN_ERR_UNSUPPORTED;                 // a workgroup walks a ray in 64-sample passes
  if (ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  if (z_fine != nullptr && (n_samples < 3 || (4L * n_samples + n_new) * 4 > kFusedExtraBytes / 2))
    return DMN_ERR_UNSUPPORTED;
  FusedArgs a{};
  a.f = FieldArgs{rays_o, rays_d, z, z_ray_stride, nullptr, blob, nullptr, status, n_rays * (long)n_samples, n_samples,
                  ins_num + 1, nullptr, sigma_only ? 1 : 0};
  a.n_rays = n_rays;
  a.rgb = rgb; a.depth = depth; a.ins = ins; a.weights = weights;
  a.u = u; a.u_stride = u_ray_stride; a.n_new = n_new; a.z_fine = z_fine;
  hipStream_t st = (hipStream_t)stream;
  if (precision == DMN_PREC_FP16X3) return launch_fused<NetX3<PrecF16>>(a, ins_num, st);
  if (precision == DMN_PREC_BF16X3) return launch_fused<NetX3<PrecBF16>>(a, ins_num, st);
  if (precision == DMN_PREC_BF16X6) return launch_fused<NetX6>(a, ins_num, st);      // blob: the x6 blob
  return DMN_ERR_BAD_ARG;
}

// One chunk of the image loop in two launches: coarse field + compositing + resampling, fine field + compositing.
// `z_fine` [n_rays, n_samples + n_importance] is written by the first launch and read by the second (the caller owns
// it: workspace or result, it is z_vals_fine of networks/render.py:70).  Coarse maps are optional (null: not computed;
// with coarse_sigma_only the coarse network stops behind its density head, they would be composites of zeros).
extern "C" int dmn_render_fused(const float* rays_o, const float* rays_d, const float* z_coarse, long z_ray_stride,
                                const float* u, long u_ray_stride, long n_rays, int n_samples, int n_importance,
                                const void* blob_coarse, const void* blob_fine, int ins_num, int precision,
                                int coarse_sigma_only, float* z_fine, float* rgb_fine, float* depth_fine,
                                float* ins_fine, float* rgb_coarse, float* depth_coarse, float* ins_coarse, int* status,
                                void* stream) {
  if (!z_fine || n_importance < 0) return DMN_ERR_BAD_ARG;
  if ((n_samples + n_importance) % 64 != 0) return DMN_ERR_UNSUPPORTED;
  int rc = dmn_field_composite_x3(rays_o, rays_d, z_coarse, z_ray_stride, n_rays, n_samples, blob_coarse, ins_num,
                                  precision, coarse_sigma_only, rgb_coarse, depth_coarse, ins_coarse, nullptr, u,
                                  u_ray_stride, n_importance, z_fine, status, stream);
  if (rc != DMN_OK) return rc;
  return dmn_field_composite_x3(rays_o, rays_d, z_fine, n_samples + n_importance, n_rays, n_samples + n_importance,
                                blob_fine, ins_num, precision, 0, rgb_fine, depth_fine, ins_fine, nullptr, nullptr, 0,
                                0, nullptr, status, stream);
}
