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

// ---- the whole loss head in five launches (evaluator.py:19-74 without its ~100 small tensor operations) -------
// labels -> classes: histogram of the labels in LDS (integer atomics), the labels present in increasing order get the
// class indices 0 .. V-1 (what torch.unique's inverse gives), count[v] = rays of class v.  meta = {V, error flag}.
constexpr int kMaxLabel = 1024;

// Ray-sharded training (bins > 0): the label set is the whole batch's.  hist_out != null: only this shard's histogram
// over labels [0, bins) leaves (as floats, to be summed over the shards); hist_in != null: the classes come from that
// summed histogram instead of this shard's own.
__global__ __launch_bounds__(1024) void ins_label_prep_kernel(const long* labels, long N, int Vmax, long* classes,
                                                              float* count, int* meta, const float* hist_in,
                                                              float* hist_out, int bins) {
  __shared__ int hist[kMaxLabel];
  __shared__ int slot[kMaxLabel];
  __shared__ int bad;
  const int tid = threadIdx.x;
  const int limit = bins > 0 ? bins : kMaxLabel;
  hist[tid] = hist_in != nullptr && tid < bins ? (int)hist_in[tid] : 0;
  if (tid == 0) bad = 0;
  __syncthreads();
  for (long i = tid; i < N; i += 1024) {
    const long l = labels[i];
    if (l < 0 || l >= limit) bad = 1; else if (hist_in == nullptr) atomicAdd(&hist[l], 1);
  }
  __syncthreads();
  if (hist_out != nullptr) {
    if (tid < bins) hist_out[tid] = (float)hist[tid];
    if (tid == 0) {
      meta[0] = 0;
      meta[1] = bad;
    }
    return;
  }
  // exclusive prefix count of the non-empty bins (Hillis-Steele over 1024 flags)
  const int mine = hist[tid] > 0 ? 1 : 0;
  slot[tid] = mine;
  __syncthreads();
  for (int d = 1; d < kMaxLabel; d <<= 1) {
    const int add = tid >= d ? slot[tid - d] : 0;
    __syncthreads();
    slot[tid] += add;
    __syncthreads();
  }
  const int V = slot[kMaxLabel - 1];
  const int my_slot = slot[tid] - mine;
  __syncthreads();
  slot[tid] = mine ? my_slot : -1;
  if (mine && my_slot < Vmax) count[my_slot] = (float)hist[tid];
  if (tid >= V && tid < Vmax) count[tid] = 0.f;
  __syncthreads();
  const bool fail = bad != 0 || V > Vmax;
  if (tid == 0) {
    meta[0] = V;
    meta[1] = fail ? 1 : 0;
  }
  for (long i = tid; i < N; i += 1024) {
    const long l = labels[i];
    classes[i] = (fail || l < 0 || l >= limit) ? 0 : slot[l];
  }
}

// the statistics without sorting the rays: workgroup (v, chunk) walks the class indices of its chunk of the rays and adds up
// the rows of class v in ray order (wave w owns the w-th quarter of the chunk; partial sums are added in wave order, the
// chunks' fp64 partials in chunk order by ins_cost_kernel): deterministic
__global__ __launch_bounds__(256) void ins_stats_scan_kernel(const float* pred, const long* cls, long N, int K,
                                                             double* partial /* [chunks][3][V][K] */, int V) {
  __shared__ double part[4][3][kMaxCols];
  const int v = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long per = ((N + 256L * gridDim.y - 1) / (256L * gridDim.y)) * 64;          // rays per wave, whole groups of 64
  const long begin = ((long)blockIdx.y * 4 + wave) * per, end = begin + per < N ? begin + per : N;
  double s1[2] = {0.0, 0.0}, s0[2] = {0.0, 0.0}, tp[2] = {0.0, 0.0};
  for (long i0 = begin; i0 < end; i0 += 64) {
    const long i = i0 + lane;
    unsigned long long hit = __ballot(i < end && cls[i] == v);
    while (hit) {
      const int j = __ffsll((long long)hit) - 1;
      hit &= hit - 1;
      const float* row = pred + (i0 + j) * K;
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
    partial[(((long)blockIdx.y * 3 + q) * V + v) * K + p] = part[0][q][p] + part[1][q][p] + part[2][q][p] + part[3][q][p];
  }
}

__global__ void ins_stats_sum_kernel(const double* partial, int chunks, long elems, float* stats) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  double v = 0.0;
  for (int c = 0; c < chunks; ++c) v += partial[(long)c * elems + e];
  stats[e] = (float)v;
}

// cost matrices (evaluator.py:60-70) from the statistics; host_blob = {V, flag, cost_ce + cost_siou [Vmax][K]} is what
// the host's Hungarian step reads back in one copy
// partial == null: stats is complete already (ray-sharded training: summed over the shards in between); n_dev != null: the
// ray count of the whole batch, on the device
__global__ __launch_bounds__(256) void ins_cost_kernel(const double* partial, int chunks, float* stats, const float* count,
                                                       const int* meta, long N, const float* n_dev, int K, int Vmax,
                                                       float* cost /* [2][Vmax][K] */, float* colsum /* [2][K] */,
                                                       float* host_blob) {
  __shared__ float s0sum[kMaxCols], tpsum[kMaxCols];
  const int V = meta[0];
  if (partial != nullptr)
    for (int e = threadIdx.x; e < 3 * Vmax * K; e += 256) {     // stats = sum of the chunks' partials, in chunk order
      double v = 0.0;
      for (int c = 0; c < chunks; ++c) v += partial[(long)c * 3 * Vmax * K + e];
      stats[e] = (float)v;
    }
  __syncthreads();
  const float* s1 = stats;
  const float* s0 = stats + (long)Vmax * K;
  const float* tp = stats + 2L * Vmax * K;
  for (int k = threadIdx.x; k < K; k += 256) {
    float a = 0.f, b = 0.f;
    for (int v = 0; v < V; ++v) {
      a += s0[v * K + k];
      b += tp[v * K + k];
    }
    s0sum[k] = a;
    tpsum[k] = b;
    colsum[k] = a;
    colsum[K + k] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    host_blob[0] = (float)V;
    host_blob[1] = (float)meta[1];
  }
  const float n = n_dev != nullptr ? *n_dev : (float)N;
  for (int e = threadIdx.x; e < Vmax * K; e += 256) {
    const int v = e / K, k = e - v * K;
    float ce = 0.f, siou = 0.f;
    if (v < V) {
      ce = -(s1[e] + (s0sum[k] - s0[e])) / n;                         // :60
      const float t = tp[e];
      const float fp = tpsum[k] - t, fn = count[v] - t;               // :63-65
      siou = 1.0f - t / (t + fp + fn + 1e-6f);                        // :66-67
    }
    cost[e] = ce;
    cost[(long)Vmax * K + e] = siou;
    host_blob[2 + e] = ce + siou;
  }
}

// The assignment itself on the device (opt-in: removes the training step's only wait for the device): rectangular linear
// sum assignment of the V x K matrix cost_ce + cost_siou (V <= K), the shortest-augmenting-path algorithm scipy's
// linear_sum_assignment runs (Crouse 2016), in fp64 like scipy, with scipy's scan order reproduced in the one place where
// it decides between equal candidates (its `remaining` array: columns in reverse order, a removed entry replaced by the
// last one; among equally short paths the LAST unassigned column in that order wins, else the FIRST column).
// One wave; lane l owns columns l and l + 64.  match = {col_of [Vmax], class_of [K]}, -1 where unmatched.
__global__ __launch_bounds__(64) void ins_assign_kernel(const float* host_blob, int* meta, int K, int Vmax, int* match) {
  __shared__ double u[kMaxCols], v[kMaxCols], spc[kMaxCols];
  __shared__ int path[kMaxCols], col4row[kMaxCols], row4col[kMaxCols], remaining[kMaxCols], pos[kMaxCols];
  __shared__ unsigned char sr[kMaxCols], sc[kMaxCols];
  const int lane = threadIdx.x;
  const int V = meta[0];
  const float* cost = host_blob + 2;
  const double inf = __builtin_huge_val();
  for (int j = lane; j < kMaxCols; j += 64) {
    u[j] = 0.0; v[j] = 0.0;
    col4row[j] = -1; row4col[j] = -1;
  }
  __syncthreads();
  bool failed = meta[1] != 0 || V <= 0 || V > K;
  for (int cur = 0; cur < V && !failed; ++cur) {
    for (int j = lane; j < kMaxCols; j += 64) {
      sr[j] = 0; sc[j] = 0; spc[j] = inf;
      if (j < K) { remaining[j] = K - 1 - j; pos[j] = K - 1 - j; }
    }
    __syncthreads();
    int num_remaining = K, sink = -1, i = cur;
    double min_val = 0.0;
    while (sink < 0) {
      if (lane == 0) sr[i] = 1;
      // relax the paths through row i; then the shortest over the remaining columns with scipy's choice among equals
      double best = inf;
      for (int j = lane; j < K; j += 64)
        if (!sc[j]) {
          const double r = min_val + (double)cost[(long)i * K + j] - u[i] - v[j];
          if (r < spc[j]) { path[j] = i; spc[j] = r; }
          best = spc[j] < best ? spc[j] : best;
        }
      for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(best, d);
        best = o < best ? o : best;
      }
      if (!(best < inf)) { failed = true; break; }
      int last_free = -1, first_any = 1 << 30;          // positions in `remaining` among the columns at `best`
      for (int j = lane; j < K; j += 64)
        if (!sc[j] && spc[j] == best) {
          if (row4col[j] < 0) last_free = pos[j] > last_free ? pos[j] : last_free;
          first_any = pos[j] < first_any ? pos[j] : first_any;
        }
      for (int d = 32; d >= 1; d >>= 1) {
        const int a = __shfl_xor(last_free, d), b = __shfl_xor(first_any, d);
        last_free = a > last_free ? a : last_free;
        first_any = b < first_any ? b : first_any;
      }
      const int index = last_free >= 0 ? last_free : first_any;
      const int j = remaining[index];
      min_val = best;
      if (row4col[j] < 0) sink = j; else i = row4col[j];
      __syncthreads();
      if (lane == 0) {
        sc[j] = 1;
        --num_remaining;
        const int moved = remaining[num_remaining];
        remaining[index] = moved;
        pos[moved] = index;
      }
      num_remaining = __shfl(num_remaining, 0);
      __syncthreads();
    }
    if (failed) break;
    // dual variables (scipy's order of operations), then the augmentation along the path
    for (int r = lane; r < V; r += 64)
      if (r == cur) u[r] += min_val;
      else if (sr[r]) u[r] += min_val - spc[col4row[r]];
    for (int j = lane; j < K; j += 64)
      if (sc[j]) v[j] -= min_val - spc[j];
    __syncthreads();
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r];
        col4row[r] = j;
        j = t;
        if (r == cur) break;
      }
    }
    __syncthreads();
  }
  if (lane == 0 && failed) meta[1] = 1;
  for (int r = lane; r < Vmax; r += 64) match[r] = (!failed && r < V) ? col4row[r] : -1;
  for (int j = lane; j < K; j += 64) match[Vmax + j] = failed ? -1 : row4col[j];
}

// loss terms for a matching: col_of[v] = column matched to class v (v < V), class_of[k] = class matched to column k or -1
// out = {valid_ce + invalid_ce + valid_siou, valid_ce, invalid_ce, valid_siou} (evaluator.py:27-37)
__global__ __launch_bounds__(64) void ins_loss_kernel(const float* cost, const float* colsum, const int* col_of,
                                                      const int* class_of, const int* meta, long N, const float* n_dev,
                                                      int K, int Vmax, float* out) {
  const int V = meta[0];
  const int lane = threadIdx.x;
  if (meta[1] != 0 || V <= 0 || V > Vmax) {          // no assignment (device-side matching on labels the kernels do not take)
    if (lane < 4) out[lane] = __builtin_nanf("");
    return;
  }
  float ce = 0.f, siou = 0.f, inv = 0.f;
  int unmatched = 0;
  for (int v = lane; v < V; v += 64) {
    const int c = col_of[v];
    if (c >= 0) {
      ce += cost[v * K + c];
      siou += cost[(long)Vmax * K + v * K + c];
    }
  }
  for (int k = lane; k < K; k += 64)
    if (class_of[k] < 0) {
      inv += colsum[K + k];
      ++unmatched;
    }
  for (int d = 32; d >= 1; d >>= 1) {
    ce += __shfl_xor(ce, d);
    siou += __shfl_xor(siou, d);
    inv += __shfl_xor(inv, d);
    unmatched += __shfl_xor(unmatched, d);
  }
  if (lane == 0) {
    const float valid_ce = ce / (float)V, valid_siou = siou / (float)V;
    const float n = n_dev != nullptr ? *n_dev : (float)N;
    const float invalid_ce = unmatched ? inv / (n * (float)unmatched) : 0.f;
    out[0] = valid_ce + invalid_ce + valid_siou;
    out[1] = valid_ce;
    out[2] = invalid_ce;
    out[3] = valid_siou;
  }
}

// d out / d stats for upstream gradients g[4] (null = 0) of the four outputs
__global__ __launch_bounds__(256) void ins_loss_bwd_kernel(const float* stats, const float* count, const float* colsum,
                                                           const int* class_of, const int* meta, long N,
                                                           const float* n_dev, int K, int Vmax, const float* g_total,
                                                           const float* g_ce, const float* g_inv, const float* g_siou,
                                                           float* g_stats) {
  __shared__ int n_unmatched;
  const int V = meta[0];
  if (meta[1] != 0 || V <= 0 || V > Vmax) {
    for (int e = threadIdx.x; e < 3 * Vmax * K; e += 256) g_stats[e] = 0.f;
    return;
  }
  if (threadIdx.x == 0) {
    int u = 0;
    for (int k = 0; k < K; ++k) u += class_of[k] < 0;
    n_unmatched = u;
  }
  __syncthreads();
  const float gt = g_total ? *g_total : 0.f;
  const float a = gt + (g_ce ? *g_ce : 0.f), b = gt + (g_inv ? *g_inv : 0.f), s = gt + (g_siou ? *g_siou : 0.f);
  const float n = n_dev != nullptr ? *n_dev : (float)N, fv = (float)V;
  const float* tp = stats + 2L * Vmax * K;
  for (int e = threadIdx.x; e < Vmax * K; e += 256) {
    const int v = e / K, k = e - v * K;
    float g1 = 0.f, g0 = 0.f, gtp = 0.f;
    if (v < V) {
      const int vk = class_of[k];
      if (vk >= 0) {
        if (v == vk) g1 = -a / (n * fv); else g0 = -a / (n * fv);
        const float t = tp[vk * K + k];
        const float D = colsum[K + k] - t + count[vk] + 1e-6f;      // t + fp + fn + 1e-6
        gtp = (v == vk ? -1.f / D : t / (D * D)) * s / fv;
      } else {
        gtp = b / (n * (float)n_unmatched);
      }
    }
    g_stats[e] = g1;
    g_stats[(long)Vmax * K + e] = g0;
    g_stats[2L * Vmax * K + e] = gtp;
  }
}

}  // namespace

extern "C" int dmn_ins_label_prep(const long* labels, long n_rays, int max_classes, long* classes, float* count,
                                  int* meta, void* stream) {
  if (n_rays <= 0 || max_classes <= 0 || max_classes > kMaxCols) return DMN_ERR_BAD_ARG;
  if (!labels || !classes || !count || !meta) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_label_prep_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n_rays, max_classes,
                     classes, count, meta, (const float*)nullptr, (float*)nullptr, 0);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_label_hist(const long* labels, long n_rays, int bins, float* hist, int* meta, void* stream) {
  if (n_rays < 0 || bins <= 0 || bins > kMaxLabel) return DMN_ERR_BAD_ARG;
  if (!labels || !hist || !meta) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_label_prep_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n_rays, 0,
                     (long*)nullptr, (float*)nullptr, meta, (const float*)nullptr, hist, bins);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_label_slots(const long* labels, long n_rays, const float* hist, int bins, int max_classes,
                                   long* classes, float* count, int* meta, void* stream) {
  if (n_rays < 0 || bins <= 0 || bins > kMaxLabel || max_classes <= 0 || max_classes > kMaxCols) return DMN_ERR_BAD_ARG;
  if (!labels || !hist || !classes || !count || !meta) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_label_prep_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, n_rays, max_classes,
                     classes, count, meta, hist, (float*)nullptr, bins);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_stats_sum(const double* partial, long n_rays, int n_cols, int max_classes, float* stats,
                                 void* stream) {
  if (n_rays < 0 || n_cols <= 0 || max_classes <= 0) return DMN_ERR_BAD_ARG;
  if (!partial || !stats) return DMN_ERR_BAD_ARG;
  const long elems = 3L * max_classes * n_cols;
  hipLaunchKernelGGL(ins_stats_sum_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     partial, dmn_ins_stats_chunks(n_rays), elems, stats);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_stats_chunks(long n_rays) {      // ray chunks of dmn_ins_stats_scan: ~512 rays per workgroup
  if (n_rays <= 0) return 1;
  const long c = (n_rays + 511) / 512;
  return (int)(c < 1 ? 1 : (c > 16 ? 16 : c));
}

extern "C" int dmn_ins_stats_scan(const float* pred, const long* classes, long n_rays, int n_cols, int n_classes,
                                  double* partial, void* stream) {
  if (n_rays < 0 || n_cols <= 0 || n_classes <= 0) return DMN_ERR_BAD_ARG;
  if (n_cols > kMaxCols) return DMN_ERR_UNSUPPORTED;
  if (!pred || !classes || !partial) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_stats_scan_kernel, dim3(n_classes, dmn_ins_stats_chunks(n_rays)), dim3(256), 0,
                     (hipStream_t)stream, pred, classes, n_rays, n_cols, partial, n_classes);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_cost(const double* partial, float* stats, const float* count, const int* meta, long n_rays,
                            const float* n_rays_dev, int n_cols, int max_classes, float* cost, float* colsum,
                            float* host_blob, void* stream) {
  if (n_rays < 0 || (n_rays == 0 && !n_rays_dev) || n_cols <= 0 || n_cols > kMaxCols || max_classes <= 0) return DMN_ERR_BAD_ARG;
  if (!stats || !count || !meta || !cost || !colsum || !host_blob) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_cost_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, dmn_ins_stats_chunks(n_rays),
                     stats, count, meta, n_rays, n_rays_dev, n_cols, max_classes, cost, colsum, host_blob);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_assign(const float* host_blob, int* meta, int n_cols, int max_classes, int* match, void* stream) {
  if (n_cols <= 0 || n_cols > kMaxCols || max_classes <= 0 || max_classes > kMaxCols) return DMN_ERR_BAD_ARG;
  if (!host_blob || !meta || !match) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_assign_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, host_blob, meta, n_cols, max_classes,
                     match);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_loss(const float* cost, const float* colsum, const int* col_of, const int* class_of,
                            const int* meta, long n_rays, const float* n_rays_dev, int n_cols, int max_classes,
                            float* out4, void* stream) {
  if (n_rays < 0 || (n_rays == 0 && !n_rays_dev) || n_cols <= 0 || n_cols > kMaxCols || max_classes <= 0) return DMN_ERR_BAD_ARG;
  if (!cost || !colsum || !col_of || !class_of || !meta || !out4) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cost, colsum, col_of, class_of, meta,
                     n_rays, n_rays_dev, n_cols, max_classes, out4);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_ins_loss_bwd(const float* stats, const float* count, const float* colsum, const int* class_of,
                                const int* meta, long n_rays, const float* n_rays_dev, int n_cols, int max_classes,
                                const float* g_total, const float* g_ce, const float* g_inv, const float* g_siou,
                                float* g_stats, void* stream) {
  if (n_rays < 0 || (n_rays == 0 && !n_rays_dev) || n_cols <= 0 || n_cols > kMaxCols || max_classes <= 0) return DMN_ERR_BAD_ARG;
  if (!stats || !count || !colsum || !class_of || !meta || !g_stats) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(ins_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, count, colsum, class_of,
                     meta, n_rays, n_rays_dev, n_cols, max_classes, g_total, g_ce, g_inv, g_siou, g_stats);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

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
