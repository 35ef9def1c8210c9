// Field kernel of the full-precision bf16 arithmetic ("bf16x6", field_x6.h): encodings -> the whole network -> raw rows,
// as field_fwd.hip does in exact fp32 (networks/render.py:37-38,49-61,71-83, networks/dm_nerf.py:80-106); the saving
// variant for training, the blob gather with the three-way split, and the C entry points.
#include <hip/hip_runtime.h>

#include "../../include/dmnerf_hip.h"
#include "field_x6.h"

using namespace dmn;

namespace {

// The workgroup's 128 raw rows are one contiguous block of `raw` (128 x C floats): they are gathered in LDS -- in the stage
// buffers, which nobody reads any more -- and leave as full 16-byte-per-lane non-temporal stores.  Written row by row
// from the lanes that hold the values (dword stores with a 72-byte stride), non-temporal stores reach memory as partial
// writes (WRITE_SIZE 3.8 GB for 0.59 GB of rows per 8.2 M-point launch), and plain ones allocate in the XCD's L2 beside
// the weight stream, which then misses 2.3x as often (experiments/README.md, Round 4).
__device__ __forceinline__ void store_rows_coalesced(const float* stage, float* dst, int n_floats, int tid) {
  const int nvec = n_floats >> 2;
  for (int i = tid; i < nvec; i += 256)
    __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(stage)[i], reinterpret_cast<f32x4*>(dst) + i);
  const int rest = n_floats & 3;
  if (tid < rest) __builtin_nontemporal_store(stage[4 * nvec + tid], dst + 4 * nvec + tid);
}

// SAVE: also write the save record the backward kernels read (training; field_layout.h "Training")
template <int T, bool SAVE>
__global__ __launch_bounds__(256, 1) void field_fwd_x6_kernel(const FieldArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[kX6LdsBytes];
  float* const cst = reinterpret_cast<float*>(lds + kX6Slots * kX6StageBytes);
  const int tid = threadIdx.x;
  const int lane = tid & 63, p = lane & 31, h = lane >> 5;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* const blob = static_cast<const char*>(a.blob);
  Ctx6 c{blob + kConstFloats * 4, lds, nullptr, 0, x6_total_stages(T), wave_s, lane};

  // const section + stage 0 start streaming while the encodings are computed
  for (int k = wave_s; k < kConstChunks; k += 4)
    glds16(blob + k * 1024 + lane * 16, lds + kX6Slots * kX6StageBytes + k * 1024);
  x6_load_stage(c, 0);
  if (kX6Slots == 3) x6_load_stage(c, 1);

  const long m = ((long)blockIdx.x * 4 + wave_s) * 32 + p;
  const long mc = m < a.M ? m : a.M - 1;
  float pe[32], qe[16];
  point_encodings(a, mc, h, pe, qe);
  // training: this wave's save record, lane offset included
  float* const rec = SAVE ? a.save + ((long)blockIdx.x * 4 + wave_s) * kSaveTileFloats + lane : nullptr;
  if (SAVE) {
#pragma unroll
    for (int j = 0; j < 32; ++j) record_store(rec + (kSavePe + j) * 64, pe[j]);
#pragma unroll
    for (int j = 0; j < 16; ++j) record_store(rec + (kSaveQe + j) * 64, qe[j]);
  }

  float sigma, rgb[3], insv[16 * T];
  const int C = 4 + a.ins_ch;
  float* const stage = reinterpret_cast<float*>(lds);                 // the stage buffers, once the network is through
  float* const row = stage + (wave_s * 32 + p) * C;
  const long first = (long)blockIdx.x * 128;
  const int n_out = (int)(a.M - first < 128 ? a.M - first : 128) * C;
  if (x6_network<T, SAVE>(c, cst, pe, qe, h, a.sigma_only != 0, rec, sigma, rgb, insv)) {
    // density only: raw = [0, 0, 0, sigma, 0 ...].  (A branch of its own that ends the kernel: see x6_network.  The
    // network waited for this wave's LDS-DMA of the next stage before it returned; behind the barrier all of it landed.)
    __syncthreads();
    if (h == 0) {
      row[0] = 0.f; row[1] = 0.f; row[2] = 0.f;
      row[3] = sigma;
    }
#pragma unroll
    for (int i = 0; i < 16 * T; ++i) {
      const int ch = feat_of(i, 0) + 4 * h;
      if (ch < a.ins_ch) row[4 + ch] = 0.f;
    }
    __syncthreads();
    store_rows_coalesced(stage, a.raw + first * C, n_out, tid);
    return;
  }
  // ---- raw = cat([rgb, density, ins]) (dm_nerf.py:105)
  __syncthreads();                                   // every wave is done with the last weight stage (no DMA in flight)
  if (h == 0) {
    row[0] = rgb[0]; row[1] = rgb[1]; row[2] = rgb[2];
    row[3] = sigma;
  }
#pragma unroll
  for (int i = 0; i < 16 * T; ++i) {
    const int ch = feat_of(i, 0) + 4 * h;
    if (ch < a.ins_ch) row[4 + ch] = insv[i];
  }
  __syncthreads();
  store_rows_coalesced(stage, a.raw + first * C, n_out, tid);
  // the clamped tail copies of the three-slot ring must land before the LDS is released
  if (kX6Slots == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool SAVE>
int launch_x6_t(const FieldArgs& a, int ins_num, hipStream_t st) {
  const long blocks = (a.M + 127) / 128;
  if (blocks > 0x7fffffffL) return DMN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), block(256);
  switch (ins_tiles(ins_num)) {
    case 1: hipLaunchKernelGGL((field_fwd_x6_kernel<1, SAVE>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((field_fwd_x6_kernel<2, SAVE>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((field_fwd_x6_kernel<3, SAVE>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((field_fwd_x6_kernel<4, SAVE>), grid, block, 0, st, a); break;
    default: return DMN_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

int launch_x6(const FieldArgs& a, int ins_num, hipStream_t st) {
  if (ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_UNSUPPORTED;
  if (a.M <= 0) return DMN_OK;
  return a.save != nullptr ? launch_x6_t<true>(a, ins_num, st) : launch_x6_t<false>(a, ins_num, st);
}

// blob gather with the exact three-way split; entry i < kConstFloats is an fp32 constant, the rest are bf16 values
// whose plane (hi / mid / lo) is the position of their chunk modulo 3
__global__ void pack_x6_kernel(const float* __restrict__ params, const int32_t* __restrict__ index, long index_len,
                               char* __restrict__ blob) {
  float* cst = reinterpret_cast<float*>(blob);
  __bf16* stream = reinterpret_cast<__bf16*>(blob + kConstFloats * 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < index_len; i += (long)gridDim.x * blockDim.x) {
    const int32_t k = index[i];
    const float v = k < 0 ? 0.f : params[k];
    if (i < kConstFloats) {
      cst[i] = v;
    } else {
      const long e = i - kConstFloats;
      const int plane = (int)((e / kX3ChunkHalfs) % 3);
      const __bf16 hi = (__bf16)v;
      const float r1 = v - (float)hi;
      const __bf16 mid = (__bf16)r1;
      const __bf16 lo = (__bf16)(r1 - (float)mid);
      stream[e] = plane == 0 ? hi : (plane == 1 ? mid : lo);
    }
  }
}

}  // namespace

extern "C" int dmn_field_pack_x6(const float* params, const int32_t* index, int ins_num, void* blob, void* stream) {
  if (!params || !index || !blob || ins_num < 0 || ins_tiles(ins_num) > kMaxInsTiles) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_x6_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, index,
                     x6_index_len(ins_num), (char*)blob);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

// the same gather for any x6-format index table (the backward-data blob: dmn_field_bwd_pack_index_x6)
extern "C" int dmn_field_pack_x6_n(const float* params, const int32_t* index, long index_len, void* blob, void* stream) {
  if (!params || !index || !blob || index_len < kConstFloats) return DMN_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_x6_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, params, index, index_len, (char*)blob);
  return hipGetLastError() == hipSuccess ? DMN_OK : DMN_ERR_LAUNCH;
}

extern "C" int dmn_field_fwd_rays_x6(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                     long n_rays, int n_samples, const void* blob, int ins_num, int sigma_only,
                                     float* raw, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, nullptr, sigma_only ? 1 : 0};
  return launch_x6(a, ins_num, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_embedded_x6(const float* embedded, long n_points, const void* blob, int ins_num,
                                         float* raw, void* stream) {
  if (n_points < 0 || ins_num < 0) return DMN_ERR_BAD_ARG;
  if (n_points == 0) return DMN_OK;
  if (!embedded || !blob || !raw) return DMN_ERR_BAD_ARG;
  FieldArgs a{nullptr, nullptr, nullptr, 0, embedded, blob, raw, nullptr, n_points, 1, ins_num + 1, nullptr, 0};
  return launch_x6(a, ins_num, (hipStream_t)stream);
}

extern "C" int dmn_field_fwd_rays_x6_save(const float* rays_o, const float* rays_d, const float* z, long z_ray_stride,
                                          long n_rays, int n_samples, const void* blob, int ins_num, float* raw,
                                          float* save, void* stream) {
  if (n_rays < 0 || n_samples <= 0 || ins_num < 0 || z_ray_stride < 0) return DMN_ERR_BAD_ARG;
  if (n_rays == 0) return DMN_OK;
  if (!rays_o || !rays_d || !z || !blob || !raw || !save) return DMN_ERR_BAD_ARG;
  FieldArgs a{rays_o, rays_d, z, z_ray_stride, nullptr, blob, raw, nullptr, n_rays * (long)n_samples, n_samples,
              ins_num + 1, save, 0};
  return launch_x6(a, ins_num, (hipStream_t)stream);
}
