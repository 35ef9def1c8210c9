// Emptiness penalizer of the reference's training step (networks/penalizer.py:5-62) as two elementwise +
// reduction kernels over the field's raw output: forward sums, and the gradient w.r.t. the instance logits.
// One thread per sample point; the ins_num + 1 logits of a point are contiguous.  HBM-bound:
// reads raw[..., 4:] once (forward) / once more (backward), writes g_raw once.
//
// Per sample, with n = |rays_d|, p = z*n, D = depth*n (products rounded separately, as the reference computes them):
//   gauss = exp(-(D - p)^2 / (2 w^2)) / (0.4 sqrt(2 pi)) + 1e-8
//   before = p < (depth - tol)*n,  after = p > (depth + tol)*n,  middle = 1 - (after + before)
//   L_before = sum_c bce_c * (1 - gauss) * before     bce_c = -log(1 - s_c + 1e-8), last channel: -log(s_c + 1e-8)
//   L_middle = -log(1 - s_last + 1e-8) * gauss * middle
//   loss = sum L_before / ((ins_num + 1) * max(sum before, 1e-8)) + sum L_middle / max(sum middle, 1e-8)
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"

namespace {

struct PenArgs {
  const float* raw;      // [N, S, C]
  const float* z;        // z[n*z_stride + s]
  long z_stride;
  const float* depth;    // [N]
  const float* rays_d;   // [N, 3]
  long n_rays;
  int S, C;
  float tolerance, two_w2, height;       // 2*w*w and 0.4*sqrt(2 pi) in fp32, rounded as torch rounds them
  double* sums;          // [4]: sum L_before, sum before, sum L_middle, sum middle
};

struct Weights {
  float gauss, before, middle;
};

__device__ __forceinline__ Weights sample_weights(const PenArgs& a, long n, int s) {
  const float dx = a.rays_d[3 * n], dy = a.rays_d[3 * n + 1], dz = a.rays_d[3 * n + 2];
  const float nrm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float depth = a.depth[n];
  const float p = __fmul_rn(a.z[n * a.z_stride + s], nrm);
  const float delta = __fsub_rn(__fmul_rn(depth, nrm), p);
  Weights w;
  w.gauss = __fadd_rn(__fdiv_rn(expf(-__fdiv_rn(__fmul_rn(delta, delta), a.two_w2)), a.height), 1e-8f);
  w.before = p < __fmul_rn(__fsub_rn(depth, a.tolerance), nrm) ? 1.f : 0.f;
  const float after = p > __fmul_rn(__fadd_rn(depth, a.tolerance), nrm) ? 1.f : 0.f;
  w.middle = 1.f - (after + w.before);
  return w;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ double block_sum(double v, double* scratch) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ __launch_bounds__(256) void penalizer_fwd_kernel(const PenArgs a) {
  __shared__ double scratch[4];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double lb = 0.0, nb = 0.0, lm = 0.0, nm = 0.0;
  if (i < a.n_rays * a.S) {
    const long n = i / a.S;
    const int s = (int)(i - n * a.S);
    const Weights w = sample_weights(a, n, s);
    const float* logits = a.raw + i * a.C + 4;
    const int K = a.C - 4;
    nb = w.before;
    nm = w.middle;
    if (w.before != 0.f) {
      float acc = 0.f;
      for (int c = 0; c < K - 1; ++c) acc += -logf(1.f - sigmoidf(logits[c]) + 1e-8f);
      acc += -logf(sigmoidf(logits[K - 1]) + 1e-8f);
      lb = (double)(acc * (1.f - w.gauss));
    }
    if (w.middle != 0.f) lm = (double)(-logf(1.f - sigmoidf(logits[K - 1]) + 1e-8f) * w.gauss * w.middle);
  }
  const double v[4] = {block_sum(lb, scratch), block_sum(nb, scratch), block_sum(lm, scratch), block_sum(nm, scratch)};
  if (threadIdx.x < 4 && v[threadIdx.x] != 0.0) atomicAdd(a.sums + threadIdx.x, v[threadIdx.x]);
}

// g_raw[n,s,4+c] = g_loss * d loss / d logit; channels 0..3 get 0.  One thread per ELEMENT of raw (the sample weights are
// recomputed by the C threads of a point): with one thread per point every access walked its own 72-byte row and the
// kernel moved its 113 MB at 0.6 TB/s.
__global__ __launch_bounds__(256) void penalizer_bwd_kernel(const PenArgs a, const float* g_loss, float* g_raw) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n_rays * a.S * a.C) return;
  const long i = e / a.C;
  const int ch = (int)(e - i * a.C);
  if (ch < 4) {
    g_raw[e] = 0.f;
    return;
  }
  const int c = ch - 4;
  const long n = i / a.S;
  const int s = (int)(i - n * a.S);
  const Weights w = sample_weights(a, n, s);
  const int K = a.C - 4;
  const float g = g_loss[0];
  const float kb = g * (1.f - w.gauss) * w.before / ((float)K * fmaxf((float)a.sums[1], 1e-8f));
  const float km = g * w.gauss * w.middle / fmaxf((float)a.sums[3], 1e-8f);
  const float sg = sigmoidf(a.raw[e]);
  const float ds = sg * (1.f - sg);
  float v;
  if (c < K - 1) {
    v = kb * ds / (1.f - sg + 1e-8f);                                    // d/dx -log(1 - s + 1e-8)
  } else {
    v = -kb * ds / (sg + 1e-8f) + km * ds / (1.f - sg + 1e-8f);          // -log(s + 1e-8), and the middle term
  }
  g_raw[e] = v;
}

// loss = sum L_before / ((ins_num + 1) max(sum before, 1e-8)) + sum L_middle / max(sum middle, 1e-8) in fp64, as the
// host-side tensor formula computes it (one launch instead of eight small ones)
__global__ void penalizer_loss_kernel(const double* sums, int K, float* loss) {
  loss[0] = (float)(sums[0] / ((double)K * fmax(sums[1], 1e-8)) + sums[2] / fmax(sums[3], 1e-8));
}

PenArgs make_args(const float* raw, const float* z, long z_stride, const float* depth, const float* rays_d, long n,
                  int S, int C, float tolerance, float deta_w, double* sums) {
  const float w2 = deta_w * deta_w;                       // torch: delta_W ** 2 on an fp32 tensor
  const float height = 0.4f * sqrtf(6.283185307179586f);  // torch.Tensor([0.4]) * sqrt(torch.Tensor([2 pi]))
  return PenArgs{raw, z, z_stride, depth, rays_d, n, S, C, tolerance, 2.f * w2, height, sums};
}

}  // namespace

extern "C" int dmn_penalizer_fwd(const float* raw, const float* z, long z_ray_stride, const float* depth,
                                 const float* rays_d, long n_rays, int n_samples, int channels, float tolerance,
                                 float deta_w, double* sums4, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || channels < 5 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (!sums4) return DMN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sums4, 0, 4 * sizeof(double), st) != hipSuccess) return DMN_ERR_LAUNCH;
  if (n_rays == 0) return DMN_OK;
  if (!raw || !z || !depth || !rays_d) return DMN_ERR_BAD_ARG;
  const long total = n_rays * n_samples;
  const PenArgs a = make_args(raw, z, z_ray_stride, depth, rays_d, n_rays, n_samples, channels, tolerance, deta_w, sums4);
  hipLaunchKernelGGL(penalizer_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_penalizer_loss(const double* sums4, int channels, float* loss, void* stream) {
  if (!sums4 || !loss || channels < 5) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(penalizer_loss_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sums4, channels - 4, loss);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_penalizer_bwd(const float* raw, const float* z, long z_ray_stride, const float* depth,
                                 const float* rays_d, long n_rays, int n_samples, int channels, float tolerance,
                                 float deta_w, const double* sums4, const float* g_loss, float* g_raw, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || channels < 5 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!raw || !z || !depth || !rays_d || !sums4 || !g_loss || !g_raw) return DMN_ERR_BAD_ARG;
  const long total = n_rays * n_samples * channels;
  if ((total + 255) / 256 > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const PenArgs a = make_args(raw, z, z_ray_stride, depth, rays_d, n_rays, n_samples, channels, tolerance, deta_w,
                              const_cast<double*>(sums4));
  hipLaunchKernelGGL(penalizer_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     g_loss, g_raw);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
