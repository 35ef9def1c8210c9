// Class-wise column statistics behind the instance-matching loss (networks/evaluator.py:19-74).  The reference
// broadcasts prediction [N, ins] against one-hot ground truth into [ins, ins, N] tensors three times; everything
// its cost matrices need is, per ground-truth class v and prediction column p,
//     S1[v,p] = sum over rays of class v of log(pred + 1e-8)
//     S0[v,p] = sum over rays of class v of log(1 - pred + 1e-8)
//     TP[v,p] = sum over rays of class v of pred
// (cost_ce = -(S1 + sum_v' S0[v'] - S0) / N;  FP = sum_v' TP[v'] - TP;  FN = count(v) - TP).
// Rays arrive grouped by class (order[] + segment offsets); a workgroup owns one class, its four waves stride over
// the class's rays, a lane owns columns p = lane, lane + 64; each prediction row is read once, contiguously.
// The backward is the matching gather: g_pred[n,p] = gS1[c,p]/(pred+1e-8) - gS0[c,p]/(1-pred+1e-8) + gTP[c,p], c = class(n).
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"

namespace {

constexpr int kMaxCols = 128;

__global__ __launch_bounds__(256) void ins_stats_fwd_kernel(const float* pred, const long* order, const long* seg,
                                                            int K, float* stats /* [3][V][K] */, int V) {
  __shared__ double part[4][3][kMaxCols];
  const int v = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long begin = seg[v], end = seg[v + 1];
  double s1[2] = {0.0, 0.0}, s0[2] = {0.0, 0.0}, tp[2] = {0.0, 0.0};
  for (long i = begin + wave; i < end; i += 4) {
    const float* row = pred + order[i] * K;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int p = lane + 64 * c;
      if (p < K) {
        const float x = row[p];
        s1[c] += (double)logf(x + 1e-8f);
        s0[c] += (double)logf(1.f - x + 1e-8f);
        tp[c] += (double)x;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    part[wave][0][lane + 64 * c] = s1[c];
    part[wave][1][lane + 64 * c] = s0[c];
    part[wave][2][lane + 64 * c] = tp[c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 3 * K; e += 256) {
    const int q = e / K, p = e - q * K;
    stats[((long)q * V + v) * K + p] = (float)(part[0][q][p] + part[1][q][p] + part[2][q][p] + part[3][q][p]);
  }
}

__global__ void ins_stats_bwd_kernel(const float* pred, const long* cls, long N, int K, int V, const float* g_stats,
                                     float* g_pred) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * K) return;
  const long n = e / K;
  const int p = (int)(e - n * K);
  const long c = cls[n];
  const float x = pred[e];
  const float g1 = g_stats[((long)0 * V + c) * K + p], g0 = g_stats[((long)1 * V + c) * K + p],
              gt = g_stats[((long)2 * V + c) * K + p];
  g_pred[e] = g1 / (x + 1e-8f) - g0 / (1.f - x + 1e-8f) + gt;
}

}  // namespace

extern "C" int dmn_ins_stats_fwd(const float* pred, const long* order, const long* segments, long n_rays, int n_cols,
                                 int n_classes, float* stats, void* stream) {
  if (n_rays < 0 || n_cols <= 0 || n_classes <= 0) return DMN_ERR_BAD_ARG;
  if (n_cols > kMaxCols) return DMN_ERR_UNSUPPORTED;
  if (!pred || !order || !segments || !stats) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_stats_fwd_kernel, dim3(n_classes), dim3(256), 0, (hipStream_t)stream, pred, order, segments,
                     n_cols, stats, n_classes);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_stats_bwd(const float* pred, const long* classes, long n_rays, int n_cols, int n_classes,
                                 const float* g_stats, float* g_pred, void* stream) {
  if (n_rays < 0 || n_cols <= 0 || n_classes <= 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!pred || !classes || !g_stats || !g_pred) return DMN_ERR_BAD_ARG;
  const long total = n_rays * n_cols;
  hipLaunchKernelGGL(ins_stats_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pred, classes, n_rays, n_cols, n_classes, g_stats, g_pred);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
