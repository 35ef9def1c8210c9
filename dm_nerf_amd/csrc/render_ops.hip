// The non-MLP stages of the DM-NeRF render path for gfx950: ray generation, stratified jitter,
// standalone positional encoder, alpha compositing (LDS-staged sample blocks + wave-level
// product scan), inverse-CDF resampling, depth merge, weight-blob gather.
// Wavefront = 64 lanes throughout; one wavefront owns one ray in the per-ray kernels.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "composite_common.h"
#include "field_layout.h"
#include "trig.h"

using namespace dmn;

namespace {

constexpr int kWavesPerBlock = 4;

inline int launch_status() { return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH; }

// ------------------------------------------------------------------------------------------
// blob gather
// ------------------------------------------------------------------------------------------
__global__ void pack_kernel(const float* __restrict__ params, const int32_t* __restrict__ index, long n,
                            float* __restrict__ blob) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int32_t k = index[i];
    blob[i] = k < 0 ? 0.f : params[k];
  }
}

// ------------------------------------------------------------------------------------------
// get_rays_k (networks/helpers.py:50-61)
// ------------------------------------------------------------------------------------------
struct Camera {
  float k00, k02, k11, k12, k22;
  float r[9];
  float t[3];
};

__global__ void raygen_kernel(int H, int W, Camera cam, float* __restrict__ rays_o, float* __restrict__ rays_d) {
  const long total = (long)H * W;
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
    const int row = (int)(pix / W), col = (int)(pix - (long)row * W);
    // dirs = [(i - K02)/K00, (j - K12)/K11, K22] with i = column, j = row (helpers.py:51-55)
    const float c0 = __fdiv_rn(__fsub_rn((float)col, cam.k02), cam.k00);
    const float c1 = __fdiv_rn(__fsub_rn((float)row, cam.k12), cam.k11);
    const float c2 = cam.k22;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // sum(dirs[..., None, :] * c2w[:3,:3], -1): separately rounded products, left-to-right sum
      const float v = __fadd_rn(__fadd_rn(__fmul_rn(c0, cam.r[3 * a + 0]), __fmul_rn(c1, cam.r[3 * a + 1])),
                                __fmul_rn(c2, cam.r[3 * a + 2]));
      rays_d[3 * pix + a] = v;
      rays_o[3 * pix + a] = cam.t[a];
    }
  }
}

// rays of selected pixels only (training: get_select_full / get_select_crop, networks/helpers.py:64-111,
// which build all H*W rays every step and then index them)
__global__ void raygen_select_kernel(int W, Camera cam, const long* __restrict__ pixel, long n,
                                     float* __restrict__ rays_o, float* __restrict__ rays_d) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long pix = pixel[i];
    const int row = (int)(pix / W), col = (int)(pix - (long)row * W);
    const float c0 = __fdiv_rn(__fsub_rn((float)col, cam.k02), cam.k00);
    const float c1 = __fdiv_rn(__fsub_rn((float)row, cam.k12), cam.k11);
    const float c2 = cam.k22;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rays_d[3 * i + a] = __fadd_rn(__fadd_rn(__fmul_rn(c0, cam.r[3 * a + 0]), __fmul_rn(c1, cam.r[3 * a + 1])),
                                    __fmul_rn(c2, cam.r[3 * a + 2]));
      rays_o[3 * i + a] = cam.t[a];
    }
  }
}

// ------------------------------------------------------------------------------------------
// stratified jitter (networks/render.py:40-47)
// ------------------------------------------------------------------------------------------
__global__ void jitter_kernel(const float* __restrict__ z, long zs, const float* __restrict__ t_rand, long n_rays,
                              int S, float* __restrict__ out) {
  const long total = n_rays * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / S;
    const int s = (int)(i - n * S);
    const float* zr = z + n * zs;
    const float zc = zr[s];
    const float lower = s > 0 ? __fmul_rn(0.5f, __fadd_rn(zc, zr[s - 1])) : zc;
    const float upper = s < S - 1 ? __fmul_rn(0.5f, __fadd_rn(zr[s + 1], zc)) : zc;
    out[i] = __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t_rand[i]));
  }
}

// ------------------------------------------------------------------------------------------
// Embedder.embed (networks/dm_nerf.py:13-38)
// ------------------------------------------------------------------------------------------
__global__ void encode_kernel(const float* __restrict__ x, long n_points, int n_freqs, float* __restrict__ out) {
  const int width = 3 + 6 * n_freqs;
  const long total = n_points * width;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / width;
    const int col = (int)(i - m * width);
    float v;
    if (col < 3) {
      v = x[3 * m + col];
    } else {
      const int e = col - 3, f = e / 6, r = e - 6 * f;
      const int is_cos = r >= 3, c = is_cos ? r - 3 : r;
      v = trig_sel(x[3 * m + c] * (float)(1 << f), is_cos);
    }
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// alpha compositing: render_train (networks/render.py:6-28), manipulator_render
// (networks/manipulator.py:86-105).  One wavefront per ray; the per-slab arithmetic lives in
// composite_common.h (shared with the fused field + compositing kernel, which must give the same bits).
//
// composite_kernel (rows of up to kWideChannels floats): the ray's raw block is walked in 64-sample slabs, a slab
// (64 x C contiguous floats) is copied global -> LDS with wave-wide contiguous loads (every HBM byte is fetched once,
// fully coalesced), lane i then owns sample i of the slab.
// (A whole-ray variant -- one wave per workgroup, all slabs loaded up front, one reduction per channel per
// ray -- measured 12 % slower on the same device at the render loop's 4096-ray launches: 2.16 vs 2.47 TB/s.)
// ------------------------------------------------------------------------------------------
struct CompositeArgs {
  const float* raw;
  const float* z;
  long zs;
  const float* rays_d;
  long n_rays;
  int S, C, keep_last;
  float* rgb;
  float* weights;
  float* depth;
  float* ins;
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_kernel(CompositeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C, S = a.S;
  float* slab = smem + (long)wave * 64 * C;
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  const bool ray_ok = n < a.n_rays;
  const long nc = ray_ok ? n : a.n_rays - 1;
  const float norm = ray_norm(a.rays_d + 3 * nc);
  const float* zr = a.z + nc * a.zs;
  const float* rawr = a.raw + nc * (long)S * C;

  RayComposite st;
  const int n_slabs = (S + 63) / 64;
  for (int b = 0; b < n_slabs; ++b) {
    const int s0 = b * 64;
    const int cnt = min(64, S - s0);
    const int nfl = cnt * C;
    const float* src = rawr + (long)s0 * C;
    // The slab is this wave's own: a wave's LDS operations execute in order, so what the copy below and the reads of
    // composite_slab() need is that the compiler keeps them in program order (wave-scope fence) -- not a workgroup
    // barrier, which would tie the four rays of a block to the slowest of them twice per slab.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int nv = nfl >> 2;
      for (int i = lane; i < nv; i += 64)
        reinterpret_cast<float4*>(slab)[i] = reinterpret_cast<const float4*>(src)[i];
      for (int i = (nv << 2) + lane; i < nfl; i += 64) slab[i] = src[i];
    } else {
      for (int i = lane; i < nfl; i += 64) slab[i] = src[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float w = composite_slab<false>(st, slab, zr, s0, cnt, S, C, norm, lane);
    if (lane < cnt && ray_ok && a.weights) a.weights[n * S + s0 + lane] = w;
  }
  if (!ray_ok) return;
  composite_outputs(st, n, C, a.keep_last, lane, a.rgb, a.depth, a.ins);
}

// Compositing for wide rows (C > kWideChannels; Replica / ScanNet carry ~100 channels): no LDS slab.  Lane = sample
// reads only the density column for the transmittance scan; then the lanes turn to channels (lane l owns channels
// l, 64 + l, 128 + l) and walk the slab's samples with rows read straight from global memory -- every row is one
// contiguous wave access, each raw byte comes from HBM once (the density column's lines are met again in L2), and
// without a 25 KB slab per ray the occupancy is no longer set by LDS.
__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_wide_kernel(CompositeArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C, S = a.S;
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  if (n >= a.n_rays) return;                        // no block-level synchronisation in this kernel
  const float norm = ray_norm(a.rays_d + 3 * n);
  const float* zr = a.z + n * a.zs;
  const float* rawr = a.raw + n * (long)S * C;
  RayComposite st;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int cnt = min(64, S - s0);
    const float w = composite_slab<true>(st, rawr + (long)s0 * C, zr, s0, cnt, S, C, norm, lane);
    if (lane < cnt && a.weights) a.weights[n * S + s0 + lane] = w;
  }
  composite_outputs(st, n, C, a.keep_last, lane, a.rgb, a.depth, a.ins);
}

// ------------------------------------------------------------------------------------------
// sample_pdf (networks/helpers.py:123-155) and the resample + merge step of dm_nerf
// (networks/render.py:66-70).  One wavefront per ray, pdf/cdf and candidates in LDS.
// ------------------------------------------------------------------------------------------

struct SampleArgs {
  const float* bins; long bins_stride;
  const float* weights; long w_stride;
  const float* u;         // u[n*u_stride + k] or null
  long u_stride;
  long n_rays;
  int B, n_new;
  float* out;
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void sample_pdf_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  if (n >= a.n_rays) return;                 // no block-level barrier below
  float* cdf = smem + (long)wave * 2 * a.B;
  float* bins = cdf + a.B;
  for (int j = lane; j < a.B; j += 64) bins[j] = a.bins[n * a.bins_stride + j];
  build_cdf(a.weights + n * a.w_stride, a.B - 1, cdf, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int k = lane; k < a.n_new; k += 64) {
    const float u = a.u ? a.u[n * a.u_stride + k] : linspace01(k, a.n_new);
    a.out[n * a.n_new + k] = invert_cdf(cdf, bins, a.B, u);
  }
}

struct ResampleArgs {
  const float* z; long zs;
  const float* weights;   // [N, S]
  const float* u;         // u[n*u_stride + k] or null
  long u_stride;
  long n_rays;
  int S, n_new;
  float* z_fine;          // [N, S + n_new]
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void resample_merge_kernel(ResampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  if (n >= a.n_rays) return;
  const int S = a.S, L = S + a.n_new;
  // per-wave LDS: all[L] (coarse then new depths), cdf[S], bins[S]
  float* all = smem + (long)wave * (L + 2 * S);
  resample_merge_ray(a.z + n * a.zs, a.weights + n * S, a.u ? a.u + n * a.u_stride : nullptr, S, a.n_new, all, all + L,
                     all + L + S, a.z_fine + n * L, lane);
}

struct MergeArgs {
  const float* a; long as; int na;
  const float* b; long bs; int nb;
  long n_rays;
  float* out;
};

__global__ __launch_bounds__(64 * kWavesPerBlock) void merge_kernel(MergeArgs m) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  if (n >= m.n_rays) return;
  const int L = m.na + m.nb;
  float* all = smem + (long)wave * L;
  for (int j = lane; j < m.na; j += 64) all[j] = m.a[n * m.as + j];
  for (int j = lane; j < m.nb; j += 64) all[m.na + j] = m.b[n * m.bs + j];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  merge_or_rank_sort(all, m.na, L, m.out + n * L, lane);
}

inline unsigned ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" int dmn_abi_version(void) { return 1; }

extern "C" const char* dmn_error_string(int code) {
  switch (code) {
    case DMN_OK: return "ok";
    case DMN_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size or unsupported shape)";
    case DMN_ERR_UNSUPPORTED: return "unsupported configuration for the gfx950 kernels";
    case DMN_ERR_LAUNCH: return "HIP kernel launch failed";
    case DMN_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

extern "C" int dmn_device_name(char* buf, int buf_len) {
  if (!buf || buf_len <= 0) return DMN_ERR_BAD_ARG;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    buf[0] = 0;
    (void)hipGetLastError();
    return DMN_ERR_NO_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return DMN_ERR_NO_DEVICE;
  snprintf(buf, buf_len, "%s", prop.gcnArchName);
  return DMN_OK;
}

extern "C" int dmn_field_pack(const float* params, const int32_t* index, long blob_floats, float* blob,
                              void* stream) {
  if (!params || !index || !blob || blob_floats <= 0) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_kernel, dim3(ew_blocks(blob_floats)), dim3(256), 0, (hipStream_t)stream, params, index,
                     blob_floats, blob);
  return launch_status();
}

extern "C" int dmn_raygen(int H, int W, float k00, float k02, float k11, float k12, float k22,
                          const float* c2w, float* rays_o, float* rays_d, void* stream) {
  if (H <= 0 || W <= 0 || !c2w || !rays_o || !rays_d) return DMN_ERR_BAD_ARG;
  Camera cam{k00, k02, k11, k12, k22, {}, {}};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) cam.r[3 * r + c] = c2w[4 * r + c];
    cam.t[r] = c2w[4 * r + 3];
  }
  hipLaunchKernelGGL(raygen_kernel, dim3(ew_blocks((long)H * W)), dim3(256), 0, (hipStream_t)stream, H, W, cam,
                     rays_o, rays_d);
  return launch_status();
}

extern "C" int dmn_raygen_select(int H, int W, float k00, float k02, float k11, float k12, float k22,
                                 const float* c2w, const long* pixel_index, long n, float* rays_o, float* rays_d,
                                 void* stream) {
  if (H <= 0 || W <= 0 || n < 0 || !c2w) return DMN_ERR_BAD_ARG;
  if (n == 0) return DMN_OK;
  if (!pixel_index || !rays_o || !rays_d) return DMN_ERR_BAD_ARG;
  Camera cam{k00, k02, k11, k12, k22, {}, {}};
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) cam.r[3 * r + c] = c2w[4 * r + c];
    cam.t[r] = c2w[4 * r + 3];
  }
  hipLaunchKernelGGL(raygen_select_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, W, cam, pixel_index,
                     n, rays_o, rays_d);
  return launch_status();
}

extern "C" int dmn_jitter_depths(const float* z, long z_ray_stride, const float* t_rand, long n_rays,
                                 int n_samples, float* z_out, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!z || !t_rand || !z_out) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(jitter_kernel, dim3(ew_blocks(n_rays * n_samples)), dim3(256), 0, (hipStream_t)stream, z,
                     z_ray_stride, t_rand, n_rays, n_samples, z_out);
  return launch_status();
}

extern "C" int dmn_encode(const float* x, long n_points, int n_freqs, float* out, void* stream) {
  if (n_points < 0 || n_freqs < 0 || n_freqs > 16) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!x || !out) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(encode_kernel, dim3(ew_blocks(n_points * (3 + 6 * n_freqs))), dim3(256), 0,
                     (hipStream_t)stream, x, n_points, n_freqs, out);
  return launch_status();
}

extern "C" int dmn_composite_fwd(const float* raw, const float* z, long z_ray_stride, const float* rays_d,
                                 long n_rays, int n_samples, int channels, int keep_last, float* rgb,
                                 float* weights, float* depth, float* ins, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || channels < 5 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!raw || !z || !rays_d) return DMN_ERR_BAD_ARG;
  if (channels > 192) return DMN_ERR_UNSUPPORTED;
  CompositeArgs a{raw, z, z_ray_stride, rays_d, n_rays, n_samples, channels, keep_last, rgb, weights, depth, ins};
  const long blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
  if (channels > kWideChannels) {          // wide rows: straight from global memory, no LDS slab
    hipLaunchKernelGGL(composite_wide_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), 0,
                       (hipStream_t)stream, a);
    return launch_status();
  }
  const size_t lds = (size_t)kWavesPerBlock * 64 * channels * sizeof(float);
  hipLaunchKernelGGL(composite_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, (hipStream_t)stream,
                     a);
  return launch_status();
}

extern "C" int dmn_sample_pdf(const float* bins, long bins_ray_stride, const float* weights, long w_ray_stride,
                              const float* u, long u_ray_stride, long n_rays, int n_bins, int n_new, float* out,
                              void* stream) {
  if (n_rays < 0 || n_bins < 2 || n_new < 0 || bins_ray_stride < 0 || w_ray_stride < 0 || u_ray_stride < 0)
    return DMN_ERR_BAD_ARG;
  if (n_rays == 0 || n_new == 0) return DMN_OK;
  if (!bins || !weights || !out) return DMN_ERR_BAD_ARG;
  const size_t lds = (size_t)kWavesPerBlock * 2 * n_bins * sizeof(float);
  if (lds > 64 * 1024) return DMN_ERR_UNSUPPORTED;
  SampleArgs a{bins, bins_ray_stride, weights, w_ray_stride, u, u_ray_stride, n_rays, n_bins, n_new, out};
  const long blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, (hipStream_t)stream,
                     a);
  return launch_status();
}

extern "C" int dmn_resample_merge(const float* z_coarse, long z_ray_stride, const float* weights, const float* u,
                                  long u_ray_stride, long n_rays, int n_samples, int n_new, float* z_fine,
                                  void* stream) {
  if (n_rays < 0 || n_samples < 3 || n_new < 0 || z_ray_stride < 0 || u_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!z_coarse || !z_fine || (n_new > 0 && !weights)) return DMN_ERR_BAD_ARG;
  const size_t lds = (size_t)kWavesPerBlock * (3 * n_samples + n_new) * sizeof(float);
  if (lds > 64 * 1024) return DMN_ERR_UNSUPPORTED;
  ResampleArgs a{z_coarse, z_ray_stride, weights, u, u_ray_stride, n_rays, n_samples, n_new, z_fine};
  const long blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(resample_merge_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds,
                     (hipStream_t)stream, a);
  return launch_status();
}

extern "C" int dmn_merge_depths(const float* a, long a_ray_stride, int n_a, const float* b, long b_ray_stride,
                                int n_b, long n_rays, float* out, void* stream) {
  if (n_rays < 0 || n_a < 0 || n_b < 0 || n_a + n_b <= 0 || a_ray_stride < 0 || b_ray_stride < 0)
    return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if ((n_a > 0 && !a) || (n_b > 0 && !b) || !out) return DMN_ERR_BAD_ARG;
  const size_t lds = (size_t)kWavesPerBlock * (n_a + n_b) * sizeof(float);
  if (lds > 64 * 1024) return DMN_ERR_UNSUPPORTED;
  MergeArgs m{a, a_ray_stride, n_a, b, b_ray_stride, n_b, n_rays, out};
  const long blocks = (n_rays + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, (hipStream_t)stream, m);
  return launch_status();
}
