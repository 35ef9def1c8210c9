// The non-MLP stages of the DM-NeRF render path for gfx950: ray generation, stratified jitter,
// standalone positional encoder, alpha compositing (LDS-staged sample blocks + wave-level
// product scan), inverse-CDF resampling, depth merge, weight-blob gather.
// Wavefront = 64 lanes throughout; one wavefront owns one ray in the per-ray kernels.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_layout.h"
#include "trig.h"

using namespace dmn;

namespace {

constexpr int kWavesPerBlock = 4;

inline int launch_status() { return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH; }

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
// (networks/manipulator.py:86-105)
//
// One wavefront per ray.  The ray's raw block is walked in 64-sample slabs: a slab (64 x C
// contiguous floats) is copied global -> LDS with wave-wide contiguous loads (every HBM byte is
// fetched once, fully coalesced), lane i then owns sample i of the slab.  Transmittance is an
// exclusive product scan across the wavefront in fp64 (PyTorch's CPU cumprod accumulates in
// double, ATen acc_type<float>), carried from slab to slab.  Channel sums are wave reductions;
// lane (c mod 64) keeps the running total of channel c.
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

constexpr int kWideChannels = 32;   // wider raw rows are composited by composite_wide_kernel (no LDS slab)

__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_kernel(CompositeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C, S = a.S;
  float* slab = smem + (long)wave * 64 * C;
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  const bool ray_ok = n < a.n_rays;
  const long nc = ray_ok ? n : a.n_rays - 1;

  const float dx = a.rays_d[3 * nc + 0], dy = a.rays_d[3 * nc + 1], dz = a.rays_d[3 * nc + 2];
  const float norm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zr = a.z + nc * a.zs;
  const float* rawr = a.raw + nc * (long)S * C;

  float tot[3] = {0.f, 0.f, 0.f};   // lane l: running sums of channels l, 64+l, 128+l
  double carry = 1.0;               // transmittance entering the slab
  const int n_slabs = (S + 63) / 64;
  for (int b = 0; b < n_slabs; ++b) {
    const int s0 = b * 64;
    const int cnt = min(64, S - s0);
    const int nfl = cnt * C;
    const float* src = rawr + (long)s0 * C;
    __syncthreads();   // previous slab fully consumed (uniform trip count across the block)
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      const int nv = nfl >> 2;
      for (int i = lane; i < nv; i += 64)
        reinterpret_cast<float4*>(slab)[i] = reinterpret_cast<const float4*>(src)[i];
      for (int i = (nv << 2) + lane; i < nfl; i += 64) slab[i] = src[i];
    } else {
      for (int i = lane; i < nfl; i += 64) slab[i] = src[i];
    }
    __syncthreads();

    const int s = s0 + lane;
    const bool ok = lane < cnt;
    const float zc = ok ? zr[s] : 0.f;
    // dists = cat([z[1:] - z[:-1], 1e10]) * ||d||   (render.py:9-12)
    float delta = 1e10f;
    if (ok && s + 1 < S) delta = __fsub_rn(zr[s + 1], zc);
    delta = __fmul_rn(delta, norm);
    const float sigma = ok ? slab[lane * C + 3] : 0.f;
    // alpha = 1 - exp(-relu(sigma) * dists)   (render.py:7,16)
    const float alpha = ok ? __fsub_rn(1.f, expf(-__fmul_rn(fmaxf(sigma, 0.f), delta))) : 0.f;
    // T = exclusive cumprod(1 - alpha + 1e-10)   (render.py:18)
    const double f = ok ? (double)__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f) : 1.0;
    const double incl = wave_scan<true>(f, lane);
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0;
    const float T = (float)(carry * excl);
    carry = carry * __shfl(incl, 63);
    const float w = ok ? __fmul_rn(alpha, T) : 0.f;
    if (ok && ray_ok && a.weights) a.weights[n * S + s] = w;

    // channel sums: rgb through sigmoid (render.py:14,19), channel 3 carries depth (render.py:20),
    // instance logits raw (render.py:24)
    // lane = sample, one wave reduction per channel (rows of up to kWideChannels channels; wider rows go to
    // composite_wide_kernel)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
      const int cend = min(C - cb * 64, 64);
      for (int cc = 0; cc < cend; ++cc) {
        const int c = cb * 64 + cc;
        float v = ok ? slab[lane * C + c] : 0.f;
        if (c < 3) v = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v)));
        if (c == 3) v = zc;
        v = wave_sum(__fmul_rn(w, v));
        if (lane == cc) tot[cb] += v;
      }
    }
  }

  if (!ray_ok) return;
  const int K = a.keep_last ? C - 4 : C - 5;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) {
    const int c = cb * 64 + lane;
    if (c >= C) continue;
    if (c < 3) {
      if (a.rgb) a.rgb[3 * n + c] = tot[cb];
    } else if (c == 3) {
      if (a.depth) a.depth[n] = tot[cb];
    } else if (c - 4 < K) {
      // ins_map = sigmoid(sum w * logits)   (render.py:24-26)
      if (a.ins) a.ins[n * K + (c - 4)] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-tot[cb])));
    }
  }
}

// Compositing for wide rows (C > kWideChannels; Replica / ScanNet carry ~100 channels): no LDS slab.  Lane = sample
// reads only the density column for the transmittance scan; then the lanes turn to channels (lane l owns channels
// l, 64 + l, 128 + l) and walk the slab's samples with rows read straight from global memory -- every row is one
// contiguous wave access, each raw byte comes from HBM once (the density column's lines are met again in L2), and
// without a 25 KB slab per ray the occupancy is no longer set by LDS.  Same arithmetic as composite_kernel's walk.
__global__ __launch_bounds__(64 * kWavesPerBlock) void composite_wide_kernel(CompositeArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int C = a.C, S = a.S;
  const long n = (long)blockIdx.x * kWavesPerBlock + wave;
  if (n >= a.n_rays) return;                        // no block-level synchronisation in this kernel

  const float dx = a.rays_d[3 * n + 0], dy = a.rays_d[3 * n + 1], dz = a.rays_d[3 * n + 2];
  const float norm = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
  const float* zr = a.z + n * a.zs;
  const float* rawr = a.raw + n * (long)S * C;

  float tot[3] = {0.f, 0.f, 0.f};
  double carry = 1.0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int cnt = min(64, S - s0);
    const int s = s0 + lane;
    const bool ok = lane < cnt;
    const float zc = ok ? zr[s] : 0.f;
    float delta = 1e10f;
    if (ok && s + 1 < S) delta = __fsub_rn(zr[s + 1], zc);
    delta = __fmul_rn(delta, norm);
    const float sigma = ok ? rawr[(long)s * C + 3] : 0.f;
    const float alpha = ok ? __fsub_rn(1.f, expf(-__fmul_rn(fmaxf(sigma, 0.f), delta))) : 0.f;
    const double f = ok ? (double)__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f) : 1.0;
    const double incl = wave_scan<true>(f, lane);
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 1.0;
    const float T = (float)(carry * excl);
    carry = carry * __shfl(incl, 63);
    const float w = ok ? __fmul_rn(alpha, T) : 0.f;
    if (ok && a.weights) a.weights[n * S + s] = w;

    const float* row = rawr + (long)s0 * C;
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
          tot[cb] = fmaf(ws, v, tot[cb]);
        }
      }
    }
  }

  const int K = a.keep_last ? C - 4 : C - 5;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) {
    const int c = cb * 64 + lane;
    if (c >= C) continue;
    if (c < 3) {
      if (a.rgb) a.rgb[3 * n + c] = tot[cb];
    } else if (c == 3) {
      if (a.depth) a.depth[n] = tot[cb];
    } else if (c - 4 < K) {
      if (a.ins) a.ins[n * K + (c - 4)] = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-tot[cb])));
    }
  }
}

// ------------------------------------------------------------------------------------------
// sample_pdf (networks/helpers.py:123-155) and the resample + merge step of dm_nerf
// (networks/render.py:66-70).  One wavefront per ray, pdf/cdf and candidates in LDS.
// ------------------------------------------------------------------------------------------

// cdf[0..B-1] from B-1 weights: w += 1e-5; pdf = w / sum(w); cdf = [0, cumsum(pdf)]
// (helpers.py:125-128).  The running sum is fp64, rounded per element (ATen CPU cumsum).
__device__ void build_cdf(const float* __restrict__ w, int nw, float* cdf, int lane) {
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
__device__ void rank_sort_to(const float* vals, int L, float* __restrict__ out, int lane) {
  for (int e0 = 0; e0 < L; e0 += 256) {   // four candidates per lane per pass
    float v[4];
    int id[4], rank[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      id[q] = e0 + q * 64 + lane;
      v[q] = id[q] < L ? vals[id[q]] : 0.f;
      rank[q] = 0;
    }
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
  const int S = a.S, B = S - 1, L = S + a.n_new;
  // per-wave LDS: all[L] (coarse then new depths), cdf[B], bins[B]
  float* all = smem + (long)wave * (L + 2 * S);
  float* cdf = all + L;
  float* bins = cdf + S;
  const float* zr = a.z + n * a.zs;
  for (int j = lane; j < S; j += 64) all[j] = zr[j];
  if (a.n_new > 0) {
    // z_vals_mid = .5 * (z[1:] + z[:-1])   (render.py:66)
    for (int j = lane; j < B; j += 64) bins[j] = __fmul_rn(0.5f, __fadd_rn(zr[j + 1], zr[j]));
    // weights[..., 1:-1]   (render.py:67)
    build_cdf(a.weights + n * S + 1, S - 2, cdf, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < a.n_new; k += 64) {
      const float u = a.u ? a.u[n * a.u_stride + k] : linspace01(k, a.n_new);
      all[S + k] = invert_cdf(cdf, bins, B, u);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // z_vals_fine = sort(cat([z_coarse, z_samples]))   (render.py:70)
  rank_sort_to(all, L, a.z_fine + n * L, lane);
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
  rank_sort_to(all, L, m.out + n * L, lane);
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
