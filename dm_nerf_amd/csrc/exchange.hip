// Label surgery of the object manipulator (networks/manipulator.py:18-83, `exchanger`): per sample point, decide from
// the predicted instance label of the original scene, of the transformed (target) scene and of the two rays'
// composited labels whether the point keeps its raw row, takes the target's, or is emptied.  Integer logic on
// [N, S] labels plus row copies of raw [N, S, C]; one thread per sample point, one launch per moved object (the
// reference loops over `move_labels` and carries the original labels from one object to the next).
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"

namespace {

// argmax over sigmoid(x_c), first index among equal values -- torch.argmax(torch.sigmoid(x), -1).  The sigmoid is kept
// (it saturates: several large logits become equal and the first one wins).
__device__ __forceinline__ int label_of(const float* x, int k) {
  int best = 0;
  float top = 1.f / (1.f + expf(-x[0]));
  for (int c = 1; c < k; ++c) {
    const float s = 1.f / (1.f + expf(-x[c]));
    if (s > top) {
      top = s;
      best = c;
    }
  }
  return best;
}

__global__ void label_kernel(const float* x, long rows, long row_stride, int k, int* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) out[i] = label_of(x + i * row_stride, k);
}

__global__ void exchange_kernel(float* ori_raw, const float* tar_raw, int* ori_label, const int* ori_ray_label,
                                const int* tar_ray_label, int* tar_label_out, long n_rays, int S, int C, int move) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rays * S) return;
  const long n = i / S;
  float* o = ori_raw + i * C;
  const float* t = tar_raw + i * C;
  const int oal = ori_ray_label[n], tal = tar_ray_label[n];
  int opl = ori_label[i];
  if (opl == move && oal != move) opl = oal;            // :33-36  an occluding point inherits the ray's label
  ori_label[i] = opl;
  const bool filling = oal == move && opl != move;      // :40-42
  int tpl = label_of(t + 4, C - 4);                     // :45-48
  if (tpl == move && tal != move) tpl = tal;            // :57-60
  tar_label_out[i] = tpl;
  const bool ori_is = opl == move, tar_is = tpl == move;
  // :64-76  neither: keep; target only or both: take the target's row; original only: empty the point
  const bool take = filling || tar_is;                  // fillings are copied first, then the exchange mask (:78-81)
  const bool wipe = ori_is && !tar_is;
  if (take)
    for (int c = 0; c < C; ++c) o[c] = t[c];
  if (wipe)
    for (int c = 0; c < C; ++c) o[c] = o[c] * 0.f;       // `ori_raw * 0` of the reference (keeps NaN / sign of zero)
}

inline unsigned blocks(long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int dmn_sigmoid_argmax(const float* x, long rows, long row_stride, int n_channels, int* labels,
                                  void* stream) {
  if (rows < 0 || row_stride < n_channels || n_channels <= 0) return DMN_ERR_BAD_ARG;
  if (rows == 0) return DMN_OK;
  if (!x || !labels) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(label_kernel, dim3(blocks(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, row_stride,
                     n_channels, labels);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_exchange(float* ori_raw, const float* tar_raw, int* ori_labels, const int* ori_ray_labels,
                            const int* tar_ray_labels, int* tar_labels, long n_rays, int n_samples, int channels,
                            int move_label, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || channels < 5) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!ori_raw || !tar_raw || !ori_labels || !ori_ray_labels || !tar_ray_labels || !tar_labels) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(exchange_kernel, dim3(blocks(n_rays * n_samples)), dim3(256), 0, (hipStream_t)stream, ori_raw,
                     tar_raw, ori_labels, ori_ray_labels, tar_ray_labels, tar_labels, n_rays, n_samples, channels,
                     move_label);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}
