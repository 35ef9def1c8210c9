// Per-ray compositing and resampling arithmetic shared by the stand-alone kernels (render_ops.hip) and the fused
// field + compositing kernels (field_fused_x3.hip): both must produce the same bits, so both call these functions.
// One wavefront owns one ray; a slab is up to 64 consecutive samples of it.
//   render_train        networks/render.py:6-28      manipulator_render  networks/manipulator.py:86-105
//   sample_pdf          networks/helpers.py:123-155   resample + merge    networks/render.py:66-70
#pragma once
#include <hip/hip_runtime.h>

namespace dmn {

constexpr int kWideChannels = 32;   // wider raw rows are composited lane-per-channel (no lane-per-sample reductions)

// Sum over the 64 lanes, result in every lane.  Within a 16-lane row by DPP (quad permutes, then row
// rotates by 4 and 8: register-to-register, no LDS crossbar round trips as ds_bpermute-based shuffles
// have), then the four row totals through v_readlane.  Latency of the dependent chain matters here: the
// compositing kernel reduces one channel after another.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);   // row_ror:4
  v += dpp_mov<0x128>(v);   // row_ror:8  -> every lane holds its row's total
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

// inclusive scan (sum or product) across the 64 lanes, in fp64
template <bool PRODUCT>
__device__ __forceinline__ double wave_scan(double v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double t = __shfl_up(v, d);
    if (lane >= d) v = PRODUCT ? v * t : v + t;
  }
  return v;
}


// ||rays_d||, rounded as torch.norm does it for three components (render.py:12)
__device__ __forceinline__ float ray_norm(const float* d) {
  const float dx = d[0], dy = d[1], dz = d[2];
  return __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
}

// running state of one ray across its slabs
struct RayComposite {
  float tot[3] = {0.f, 0.f, 0.f};   // lane l: running sums of channels l, 64 + l, 128 + l
  double carry = 1.0;               // transmittance entering the slab
};

// One slab: samples s0 .. s0+cnt-1 of the ray, `rows` = their raw rows [cnt][C] (LDS or global memory), `zr` = the
// ray's depths.  Returns this lane's compositing weight w (lane = sample s0 + lane; 0 beyond cnt).
//   dists = cat([z[1:] - z[:-1], 1e10]) * ||d||; alpha = 1 - exp(-relu(sigma) * dists)          (render.py:7,9-12,16)
//   T = exclusive cumprod(1 - alpha + 1e-10), in fp64 as ATen's CPU cumprod (acc_type<float>)    (render.py:18)
//   channel sums: rgb through sigmoid (:14,19), channel 3 carries depth (:20), instance logits raw (:24)
// `maps` = false skips the channel sums (st.tot stays 0).
// WIDE = false: lane = sample, one wave reduction per channel.  WIDE = true: the lanes turn to channels (lane l owns
// channels l, 64 + l, 128 + l) and walk the slab's rows, one contiguous wave access per row.
template <bool WIDE>
__device__ __forceinline__ float composite_slab(RayComposite& st, const float* rows, float zc, float znext, int s0,
                                                int cnt, int S, int C, float norm, int lane, bool maps = true,
                                                int c_lo = 0, int c_hi = 1 << 30) {
  // zc = z[s0 + lane], znext = z[s0 + lane + 1] (unused for the ray's last sample): loaded by the caller, which may
  // have them in flight long before (the fused kernel asks for them ahead of the network evaluation)
  const int s = s0 + lane;
  const bool ok = lane < cnt;
  if (!ok) zc = 0.f;
  float delta = 1e10f;
  if (ok && s + 1 < S) delta = __fsub_rn(znext, zc);
  delta = __fmul_rn(delta, norm);
  const float sigma = ok ? rows[(long)lane * C + 3] : 0.f;
  const float alpha = ok ? __fsub_rn(1.f, expf(-__fmul_rn(fmaxf(sigma, 0.f), delta))) : 0.f;
  const double f = ok ? (double)__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f) : 1.0;
  const double incl = wave_scan<true>(f, lane);
  double excl = __shfl_up(incl, 1);
  if (lane == 0) excl = 1.0;
  const float T = (float)(st.carry * excl);
  st.carry = st.carry * __shfl(incl, 63);
  const float w = ok ? __fmul_rn(alpha, T) : 0.f;
  if (!maps) return w;              // only the weights are wanted (density-only coarse pass of an image loop)

  if (!WIDE) {
    // channels c_lo .. c_hi-1 only (the fused kernel splits a ray's channels over two waves; every channel's sum is
    // its own chain, so the split changes no bit)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
      const int cend = min(min(C, c_hi) - cb * 64, 64);
      for (int cc = max(c_lo - cb * 64, 0); cc < cend; ++cc) {
        const int c = cb * 64 + cc;
        float v = ok ? rows[(long)lane * C + c] : 0.f;
        if (c < 3) v = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v)));
        if (c == 3) v = zc;
        v = wave_sum(__fmul_rn(w, v));
        if (lane == cc) st.tot[cb] += v;
      }
    }
  } else {
    const float* row = rows;
#pragma unroll 8
    for (int sidx = 0; sidx < cnt; ++sidx, row += C) {
      const float ws = __shfl(w, sidx), zs = __shfl(zc, sidx);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        const int c = cb * 64 + lane;
        if (c < C) {
          float v = row[c];
          if (cb == 0) {
            if (lane < 3) v = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v)));
            if (lane == 3) v = zs;
          }
          st.tot[cb] = fmaf(ws, v, st.tot[cb]);
        }
      }
    }
  }
  return w;
}

// the same with the two depths read here (stand-alone kernels: many waves per CU hide the latency)
template <bool WIDE>
__device__ __forceinline__ float composite_slab(RayComposite& st, const float* rows, const float* zr, int s0, int cnt,
                                                int S, int C, float norm, int lane, bool maps = true) {
  const int s = s0 + lane;
  const float zc = lane < cnt ? zr[s] : 0.f;
  const float znext = (lane < cnt && s + 1 < S) ? zr[s + 1] : 0.f;
  return composite_slab<WIDE>(st, rows, zc, znext, s0, cnt, S, C, norm, lane, maps);
}

// the maps of ray n: rgb[3], depth, ins_map = sigmoid(sum w * logits)[: K]   (render.py:24-26; K keeps or drops the
// last, "air", channel)
__device__ __forceinline__ void composite_outputs(const RayComposite& st, long n, int C, int keep_last, int lane,
                                                  float* rgb, float* depth, float* ins, int c_lo = 0,
                                                  int c_hi = 1 << 30) {
  const int K = keep_last ? C - 4 : C - 5;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) {
    const int c = cb * 64 + lane;
    if (c >= C || c < c_lo || c >= c_hi) continue;
    if (c < 3) {
      if (rgb) rgb[3 * n + c] = st.tot[cb];
    } else if (c == 3) {
      if (depth) depth[n] = st.tot[cb];
    } else if (c - 4 < K) {
      if (ins) ins[n * K + (c - 4)] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-st.tot[cb])));
    }
  }
}

// cdf[0..B-1] from B-1 weights: w += 1e-5; pdf = w / sum(w); cdf = [0, cumsum(pdf)]
// (helpers.py:125-128).  The running sum is fp64, rounded per element (ATen CPU cumsum).
__device__ __forceinline__ void build_cdf(const float* __restrict__ w, int nw, float* cdf, int lane) {
  float part = 0.f;
  for (int j = lane; j < nw; j += 64) part += __fadd_rn(w[j], 1e-5f);
  const float total = wave_sum(part);
  double carry = 0.0;
  if (lane == 0) cdf[0] = 0.f;
  for (int j0 = 0; j0 < nw; j0 += 64) {
    const int j = j0 + lane;
    const double p = j < nw ? (double)__fdiv_rn(__fadd_rn(w[j], 1e-5f), total) : 0.0;
    const double incl = wave_scan<false>(p, lane) + carry;
    if (j < nw) cdf[j + 1] = (float)incl;
    carry = __shfl(incl, 63);
  }
}

// one draw: searchsorted(right) -> gather -> lerp (helpers.py:139-153)
__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int B, float u) {
  int lo = 0, hi = B;                      // first index with cdf[idx] > u
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  const int below = max(lo - 1, 0), above = min(lo, B - 1);
  const float cb = cdf[below], ca = cdf[above];
  float den = __fsub_rn(ca, cb);
  if (den < 1e-5f) den = 1.f;
  const float t = __fdiv_rn(__fsub_rn(u, cb), den);
  const float bb = bins[below], ba = bins[above];
  return __fadd_rn(bb, __fmul_rn(t, __fsub_rn(ba, bb)));
}

// torch.linspace(0, 1, n) as ATen's CPU kernel evaluates it in fp32: step = 1/(n-1); lower half
// start + step*k, upper half end - step*(n-1-k) with the multiply-subtract fused (bit-identical to
// torch.linspace for the sizes tested: 40, 64, 128).
__device__ __forceinline__ float linspace01(int k, int n) {
  if (n == 1) return 0.f;
  const float step = __fdiv_rn(1.f, (float)(n - 1));
  return k < n / 2 ? __fmul_rn(step, (float)k) : fmaf(-step, (float)(n - 1 - k), 1.f);
}

// ascending rank sort of L values held in LDS `vals` into global `out` (ties by index).  NaNs order after every
// number (as torch.sort does), again by index, so the ranks stay a permutation and every output slot is written.
__device__ __forceinline__ void rank_sort_to(const float* vals, int L, float* __restrict__ out, int lane) {
  for (int e0 = 0; e0 < L; e0 += 256) {   // four candidates per lane per pass
    float v[4];
    int id[4], rank[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      id[q] = e0 + q * 64 + lane;
      v[q] = id[q] < L ? vals[id[q]] : 0.f;
      rank[q] = 0;
    }
#pragma unroll 8                      // the LDS reads of eight candidates in flight (one wave may be alone on its CU)
    for (int f = 0; f < L; ++f) {
      const float x = vals[f];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool xn = x != x, vn = v[q] != v[q];
        const bool before = x < v[q] || (vn && !xn), same = x == v[q] || (xn && vn);
        rank[q] += (before || (same && f < id[q])) ? 1 : 0;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (id[q] < L) out[rank[q]] = v[q];
  }
}


// Ascending sort of all[0 .. L) = [run A (n_a values) | run B] into `out`.  When both runs are already ascending -- the
// coarse depths always are, the resampled ones are whenever the draws are (det = True: a linspace through a monotone
// inverse CDF) -- every value's rank is its index in its own run plus a binary search in the other run (a stable
// merge: equal values keep A before B): ~8 steps per value instead of the L comparisons of the rank sort.  The sorted
// output is the same whichever way it is found.  Returns through `out` only.
__device__ __forceinline__ void merge_or_rank_sort(const float* all, int n_a, int L, float* __restrict__ out, int lane) {
  bool ascending = true;
  for (int j = lane; j + 1 < L; j += 64)
    if (j + 1 != n_a) ascending = ascending && (all[j] <= all[j + 1]);         // false with a NaN on either side
  if (__all(ascending)) {
    const float* const b = all + n_a;
    const int n_b = L - n_a;
    for (int id = lane; id < L; id += 64) {
      const float v = all[id];
      int lo = 0, hi, r;
      if (id < n_a) {                      // #{B < v}
        hi = n_b;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (b[mid] < v) lo = mid + 1; else hi = mid;
        }
        r = id + lo;
      } else {                             // #{A <= v}
        hi = n_a;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (all[mid] <= v) lo = mid + 1; else hi = mid;
        }
        r = (id - n_a) + lo;
      }
      out[r] = v;
    }
    return;
  }
  rank_sort_to(all, L, out, lane);
}

// z_fine = sort(cat([z, sample_pdf(mid(z), weights[1:-1], n_new)]))  for one ray (render.py:66-70).  `zr` = the ray's S
// depths, `weights` its S compositing weights (LDS or global), `u` its n_new draws or null (linspace: det = True),
// `all` [S + n_new], `cdf` [S], `bins` [S]: LDS scratch of this wave.
__device__ __forceinline__ void resample_merge_ray(const float* zr, const float* weights, const float* u, int S, int n_new,
                                                   float* all, float* cdf, float* bins, float* z_fine_row, int lane) {
  const int B = S - 1, L = S + n_new;
  for (int j = lane; j < S; j += 64) all[j] = zr[j];
  if (n_new > 0) {
    // z_vals_mid = .5 * (z[1:] + z[:-1])   (render.py:66)
    for (int j = lane; j < B; j += 64) bins[j] = __fmul_rn(0.5f, __fadd_rn(zr[j + 1], zr[j]));
    // weights[..., 1:-1]   (render.py:67)
    build_cdf(weights + 1, S - 2, cdf, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < n_new; k += 64) {
      const float uk = u ? u[k] : linspace01(k, n_new);
      all[S + k] = invert_cdf(cdf, bins, B, uk);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // z_vals_fine = sort(cat([z_coarse, z_samples]))   (render.py:70)
  merge_or_rank_sort(all, S, L, z_fine_row, lane);
}

}  // namespace dmn
