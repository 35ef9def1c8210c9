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
// summed histogram instead of this shard's own.  Both buffers hold bins + 1 floats: entry [bins] is the out-of-range
// flag, so that after the sum EVERY shard knows that some shard saw a label outside [0, bins) and takes the same branch.
__global__ __launch_bounds__(1024) void ins_label_prep_kernel(const long* labels, long N, int Vmax, long* classes,
                                                              float* count, int* meta, const float* hist_in,
                                                              float* hist_out, int bins) {
  __shared__ int hist[kMaxLabel];
  __shared__ int slot[kMaxLabel];
  __shared__ int bad;
  const int tid = threadIdx.x;
  const int limit = bins > 0 ? bins : kMaxLabel;
  hist[tid] = hist_in != nullptr && tid < bins ? (int)hist_in[tid] : 0;
  if (tid == 0) bad = hist_in != nullptr && hist_in[bins] != 0.f ? 1 : 0;
  __syncthreads();
  for (long i = tid; i < N; i += 1024) {
    const long l = labels[i];
    if (l < 0 || l >= limit) bad = 1; else if (hist_in == nullptr) atomicAdd(&hist[l], 1);
  }
  __syncthreads();
  if (hist_out != nullptr) {
    if (tid < bins) hist_out[tid] = (float)hist[tid];
    if (tid == 0) {
      hist_out[bins] = (float)bad;
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
// wave-wide min / max, result in every lane: DPP inside the 16-lane rows, then the four row results by v_readlane
// (ds_bpermute-based shuffles cost ~100 cycles each in a dependent chain: 129 us per call of the kernel below with them)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)b), hi = (unsigned)dpp_i<CTRL>((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double lane_d(double v, int l) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_min_d(double v) {
  double t;
  t = dpp_d<0xB1>(v); v = t < v ? t : v;
  t = dpp_d<0x4E>(v); v = t < v ? t : v;
  t = dpp_d<0x124>(v); v = t < v ? t : v;
  t = dpp_d<0x128>(v); v = t < v ? t : v;
  const double a = lane_d(v, 0), b = lane_d(v, 16), c = lane_d(v, 32), d = lane_d(v, 48);
  const double ab = b < a ? b : a, cd = d < c ? d : c;
  return cd < ab ? cd : ab;
}
template <bool MAX>
__device__ __forceinline__ int wave_ext_i(int v) {
  int t;
  t = dpp_i<0xB1>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
  t = dpp_i<0x4E>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
  t = dpp_i<0x124>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
  t = dpp_i<0x128>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16),
            c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const int ab = MAX ? (a > b ? a : b) : (a < b ? a : b), cd = MAX ? (c > d ? c : d) : (c < d ? c : d);
  return MAX ? (ab > cd ? ab : cd) : (ab < cd ? ab : cd);
}

__device__ __forceinline__ int sel_i(int a0, int a1, int j) {      // element j of a [128] array held as two registers per lane
  const int l = __builtin_amdgcn_readfirstlane(j & 63);
  return j < 64 ? __builtin_amdgcn_readlane(a0, l) : __builtin_amdgcn_readlane(a1, l);
}
__device__ __forceinline__ double sel_d(double a0, double a1, int j) {
  const int l = __builtin_amdgcn_readfirstlane(j & 63);
  return j < 64 ? lane_d(a0, l) : lane_d(a1, l);
}
__device__ __forceinline__ int first_col(bool hit0, bool hit1) {    // lowest column index with a hit (-1: none)
  const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
  return m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : -1);
}

// State in registers: lane l owns columns l, l + 64 (v, shortest path costs, path, row4col, position in scipy's `remaining`
// order, visited flag) and rows l, l + 64 (u, col4row, visited flag); the chosen column / row is read from its owner by
// v_readlane.  LDS holds the cost matrix and, once per row, the path costs for the dual update.
// A failed assignment (labels the kernels do not take, a cost matrix without a finite matching) is reported per call and
// per problem: every entry of that problem's matching is kNoAssignment, which the loss kernels test; the label meta is
// shared by the calls of a step and cached across steps, so it stays read-only here.
constexpr int kNoAssignment = -2;
__global__ __launch_bounds__(64) void ins_assign_kernel(const float* host_blob, const int* meta, int K, int Vmax, int* match,
                                                        int cost_in_lds) {
  extern __shared__ float cmat[];                        // the V x K cost matrix (cost_in_lds; else read from global memory)
  __shared__ double spc_lds[kMaxCols];
  const int lane = threadIdx.x;
  const int V = meta[0];
  host_blob += (long)blockIdx.x * (2 + (long)Vmax * K);  // one workgroup per problem (the maps of a step share labels and meta)
  match += (long)blockIdx.x * (Vmax + K);
  const float* const gcost = host_blob + 2;
  if (cost_in_lds) {
    const int n = (V < Vmax ? (V > 0 ? V : 0) : Vmax) * K;
    for (int e = lane; e < n; e += 64) cmat[e] = gcost[e];
    __syncthreads();
  }
  auto cost_at = [&](int e) { return cost_in_lds ? cmat[e] : gcost[e]; };     // an LDS read, not a flat one
  const double inf = __builtin_huge_val();
  const int c0 = lane, c1 = lane + 64;                   // this lane's columns (and rows)
  const bool has0 = c0 < K, has1 = c1 < K;
  double u0 = 0.0, u1 = 0.0, v0 = 0.0, v1 = 0.0;
  int col4row0 = -1, col4row1 = -1, row4col0 = -1, row4col1 = -1;
  bool failed = meta[1] != 0 || V <= 0 || V > K || V > Vmax;
  for (int cur = 0; cur < V && !failed; ++cur) {
    double spc0 = inf, spc1 = inf;
    int path0 = -1, path1 = -1;
    int pos0 = K - 1 - c0, pos1 = K - 1 - c1;            // scipy fills `remaining` in reverse column order
    bool sc0 = false, sc1 = false, sr0 = false, sr1 = false;
    int num_remaining = K, sink = -1, i = cur;
    double min_val = 0.0;
    while (sink < 0) {
      if (i == c0) sr0 = true;
      if (i == c1) sr1 = true;
      const double ui = sel_d(u0, u1, i);
      // relax the paths through row i; then the shortest over the remaining columns with scipy's choice among equals
      if (has0 && !sc0) {
        const double r = min_val + (double)cost_at(i * K + c0) - ui - v0;
        if (r < spc0) { path0 = i; spc0 = r; }
      }
      if (has1 && !sc1) {
        const double r = min_val + (double)cost_at(i * K + c1) - ui - v1;
        if (r < spc1) { path1 = i; spc1 = r; }
      }
      const bool live0 = has0 && !sc0, live1 = has1 && !sc1;
      double best = live0 ? spc0 : inf;
      best = live1 && spc1 < best ? spc1 : best;
      best = wave_min_d(best);
      if (!(best < inf)) { failed = true; break; }
      const bool tie0 = live0 && spc0 == best, tie1 = live1 && spc1 == best;
      int last_free = -1, first_any = 1 << 30;           // positions in `remaining` among the columns at `best`
      if (tie0) { if (row4col0 < 0) last_free = pos0; first_any = pos0; }
      if (tie1) {
        if (row4col1 < 0) last_free = pos1 > last_free ? pos1 : last_free;
        first_any = pos1 < first_any ? pos1 : first_any;
      }
      last_free = wave_ext_i<true>(last_free);
      first_any = wave_ext_i<false>(first_any);
      const int index = last_free >= 0 ? last_free : first_any;
      const int j = first_col(live0 && pos0 == index, live1 && pos1 == index);      // remaining[index]
      min_val = best;
      const int rj = sel_i(row4col0, row4col1, j);
      if (rj < 0) sink = j; else i = rj;
      // remaining[index] = remaining[--num_remaining]; column j leaves
      --num_remaining;
      const int moved = first_col(live0 && pos0 == num_remaining, live1 && pos1 == num_remaining);
      if (moved == c0) pos0 = index;
      if (moved == c1) pos1 = index;
      if (j == c0) sc0 = true;
      if (j == c1) sc1 = true;
    }
    if (failed) break;
    // dual variables (scipy's order of operations); the path costs go through LDS for the per-row gather
    __syncthreads();
    if (has0) spc_lds[c0] = spc0;
    if (has1) spc_lds[c1] = spc1;
    __syncthreads();
    if (c0 == cur) u0 += min_val; else if (sr0 && c0 < V) u0 += min_val - spc_lds[col4row0];
    if (c1 == cur) u1 += min_val; else if (sr1 && c1 < V) u1 += min_val - spc_lds[col4row1];
    if (sc0) v0 -= min_val - spc0;
    if (sc1) v1 -= min_val - spc1;
    // augmentation along the path
    int j = sink;
    while (true) {
      const int r = sel_i(path0, path1, j);
      if (j == c0) row4col0 = r;
      if (j == c1) row4col1 = r;
      const int t = sel_i(col4row0, col4row1, r);
      if (r == c0) col4row0 = j;
      if (r == c1) col4row1 = j;
      j = t;
      if (r == cur) break;
    }
  }
  if (c0 < Vmax) match[c0] = failed ? kNoAssignment : (c0 < V ? col4row0 : -1);
  if (c1 < Vmax) match[c1] = failed ? kNoAssignment : (c1 < V ? col4row1 : -1);
  if (has0) match[Vmax + c0] = failed ? kNoAssignment : row4col0;
  if (has1) match[Vmax + c1] = failed ? kNoAssignment : row4col1;
}

// loss terms for a matching: col_of[v] = column matched to class v (v < V), class_of[k] = class matched to column k or -1
// out = {valid_ce + invalid_ce + valid_siou, valid_ce, invalid_ce, valid_siou} (evaluator.py:27-37)
__global__ __launch_bounds__(64) void ins_loss_kernel(const float* cost, const float* colsum, const int* col_of,
                                                      const int* class_of, const int* meta, long N, const float* n_dev,
                                                      int K, int Vmax, float* out) {
  const int V = meta[0];
  const int lane = threadIdx.x;
  if (meta[1] != 0 || V <= 0 || V > Vmax || col_of[0] == kNoAssignment) {   // no assignment (device-side matching failed)
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
  if (meta[1] != 0 || V <= 0 || V > Vmax || class_of[0] == kNoAssignment) {
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

extern "C" int dmn_ins_assign(const float* host_blob, const int* meta, int n_cols, int max_classes, int* match, int problems,
                              void* stream) {
  if (n_cols <= 0 || n_cols > kMaxCols || max_classes <= 0 || max_classes > kMaxCols || problems <= 0) return DMN_ERR_BAD_ARG;
  if (!host_blob || !meta || !match) return DMN_ERR_BAD_ARG;
  const size_t bytes = (size_t)max_classes * n_cols * sizeof(float);
  const int in_lds = bytes <= 48 * 1024;               // 110 x 110; beyond that the matrix stays in global memory
  hipLaunchKernelGGL(ins_assign_kernel, dim3(problems), dim3(64), in_lds ? bytes : 0, (hipStream_t)stream, host_blob, meta,
                     n_cols, max_classes, match, in_lds);
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
