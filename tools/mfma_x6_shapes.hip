// The bf16x6 layer loop (field_x6.h: k-steps outer, three weight planes per (k-step, tile) group read from LDS, six MFMA
// products per group, the next k-step's operands read out of the finished accumulator set, activated and split three ways
// in the MFMA shadow, one LDS-DMA piece per group from a 4 MiB stream) on the two bf16 MFMA shapes:
//   SHAPE 0   v_mfma_f32_32x32x16_bf16   32 points = one tile,  8 feature tiles of 32, k-step 16:  6 MFMAs per group
//   SHAPE 1   v_mfma_f32_16x16x32_bf16   32 points = two tiles, 16 feature tiles of 16, k-step 32: 12 MFMAs per group
// Same FLOP, same LDS bytes, same vector work per group; only the shape differs.  A chain of 256 -> 256 ReLU layers on
// random He-scaled weights (about half of the activations are zero, as in the real network), all 256 CUs, one wave per
// SIMD.  Prints ms, TFLOP/s of bf16 MFMA issued and the engine clock (s_memtime / s_memrealtime) per shape, interleaved.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=400000 tools/mfma_x6_shapes.hip -o gpurun_out/mfma_x6_shapes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef SCALAR_SUB
#define SCALAR_SUB 0
#endif
#ifndef CHAINS
#define CHAINS 2
#endif
#ifndef INTERLEAVE
#define INTERLEAVE 0
#endif
typedef __bf16 frag __attribute__((ext_vector_type(8)));
typedef __bf16 pair __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kStageBytes = 48 * 1024;
constexpr int kStreamStages = 86;

__device__ __forceinline__ float2v widen(pair p) {
  const unsigned u = __builtin_bit_cast(unsigned, p);
  return (float2v){__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ void split3_pair(float x0, float x1, frag& fh, frag& fm, frag& fl, int j) {
  const float2v x = {x0, x1};
  const pair hp = __builtin_convertvector(x, pair);
#if SCALAR_SUB      // two v_sub_f32 instead of one v_pk_add_f32 (a packed-f32 instruction beside MFMAs costs ~13 cycles more)
  const float2v wh = widen(hp);
  float a0, a1, b0, b1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(a0) : "v"(x0), "v"(wh[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(a1) : "v"(x1), "v"(wh[1]));
  const float2v r1 = {a0, a1};
  const pair mp = __builtin_convertvector(r1, pair);
  const float2v wm = widen(mp);
  asm("v_sub_f32 %0, %1, %2" : "=v"(b0) : "v"(a0), "v"(wm[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(b1) : "v"(a1), "v"(wm[1]));
  const float2v r2 = {b0, b1};
  const pair lp = __builtin_convertvector(r2, pair);
#else
  const float2v r1 = x - widen(hp);
  const pair mp = __builtin_convertvector(r1, pair);
  const float2v r2 = r1 - widen(mp);
  const pair lp = __builtin_convertvector(r2, pair);
#endif
  fh[j] = hp[0]; fh[j + 1] = hp[1];
  fm[j] = mp[0]; fm[j + 1] = mp[1];
  fl[j] = lp[0]; fl[j + 1] = lp[1];
}
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ float acc_read(float v) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v));
  return x;
}
template <int OFF>
__device__ __forceinline__ void glds16_at(const char* src, unsigned lane_bytes, unsigned lds_base) {
  const unsigned long long a = (unsigned long long)src;
  asm volatile("s_add_i32 m0, %2, %4\n\ts_add_u32 s98, %0, %4\n\ts_addc_u32 s99, %1, 0\n\tglobal_load_lds_dwordx4 %3, s[98:99]"
               : : "s"((unsigned)a), "s"((unsigned)(a >> 32)), "s"(lds_base), "v"(lane_bytes), "n"(OFF)
               : "m0", "s98", "s99", "scc");
}

struct Ctx {
  const char* stream;
  char* lds;
  const char* buf;
  int g, wave_s, lane;
};
__device__ __forceinline__ void stage_begin(Ctx& c) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  c.buf = c.lds + (c.g & 1) * kStageBytes + c.lane * 16;
  ++c.g;
}
template <int Q>
__device__ __forceinline__ void load_piece(const Ctx& c) {
  const int g = c.g % kStreamStages;
  const char* src = c.stream + ((unsigned)g * (unsigned)kStageBytes + (unsigned)(c.wave_s * 12 * 1024));
  const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)c.lds +
                       (unsigned)((c.g & 1) * kStageBytes + c.wave_s * 12 * 1024);
  glds16_at<Q * 1024>(src, (unsigned)c.lane * 16u, dst);
}
__device__ __forceinline__ void piece(const Ctx& c, int q) {
  switch (q) {
    case 0: load_piece<0>(c); break;   case 1: load_piece<1>(c); break;   case 2: load_piece<2>(c); break;
    case 3: load_piece<3>(c); break;   case 4: load_piece<4>(c); break;   case 5: load_piece<5>(c); break;
    case 6: load_piece<6>(c); break;   case 7: load_piece<7>(c); break;   case 8: load_piece<8>(c); break;
    case 9: load_piece<9>(c); break;   case 10: load_piece<10>(c); break; default: load_piece<11>(c); break;
  }
}

// ---- SHAPE 0: one 256 -> 256 layer on 32x32x16, 16 k-steps x 8 tiles, 16 groups per 48 KiB stage -----------------------
__device__ __forceinline__ void layer32(Ctx& c, f32x16* out, const f32x16* in) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t][r] = 0.01f;
  frag bh, bm, bl, nh, nm, nl;
  auto build_pair = [&](int ks, int p, frag& fh, frag& fm, frag& fl) {
    const int i = 8 * ks + 2 * p;
    const float x0 = relu_bits(acc_read(in[i >> 4][i & 15])), x1 = relu_bits(acc_read(in[(i + 1) >> 4][(i + 1) & 15]));
    split3_pair(x0, x1, fh, fm, fl, 2 * p);
  };
#pragma unroll
  for (int p = 0; p < 4; ++p) build_pair(0, p, bh, bm, bl);
  frag ah[2], am[2], al[2];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int gi = (ks % 2) * 8 + t, ar = (ks * 8 + t) & 1;
      if (gi == 0) {
        stage_begin(c);
        ah[ar] = *reinterpret_cast<const frag*>(c.buf);
        am[ar] = *reinterpret_cast<const frag*>(c.buf + 1024);
        al[ar] = *reinterpret_cast<const frag*>(c.buf + 2048);
      }
      if (gi + 1 < 16) {
        ah[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1)) * 1024);
        am[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 1) * 1024);
        al[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 2) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ar], bh, out[t], 0, 0, 0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ar], bl, out[t], 0, 0, 0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ar], bm, out[t], 0, 0, 0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[ar], bh, out[t], 0, 0, 0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ar], bm, out[t], 0, 0, 0);
      out[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ar], bh, out[t], 0, 0, 0);
      if (gi < 12) piece(c, gi);
      if (ks + 1 < 16) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if ((p * 8) / 4 == t) build_pair(ks + 1, p, nh, nm, nl);
      }
    }
    bh = nh; bm = nm; bl = nl;
  }
}

// ---- SHAPE 1: the same layer on 16x16x32: 8 k-steps of 32 x 16 feature tiles x 2 point tiles ---------------------------
// accumulators acc[p][t] (point tile p, feature tile t), 4 registers each; the B operand of k-step ks for point tile p is
// built from tiles 2 ks and 2 ks + 1 of the finished set (4 + 4 values per lane: the k order is a permutation the weight
// packing absorbs, as in field_layout.h).  CHAINS: dependent accumulation chains in flight: 2 = the two point tiles of
// one feature tile (12 MFMAs per group), 4 = two feature tiles at a time (24 MFMAs per double group, issued round-robin)
__device__ __forceinline__ void layer16(Ctx& c, f32x4 (*out)[16], const f32x4 (*in)[16]) {
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[p][t][r] = 0.01f;
  frag bh[2], bm[2], bl[2], nh[2], nm[2], nl[2];
  auto build_pair = [&](int ks, int q, frag* fh, frag* fm, frag* fl) {     // q = 0..7: point tile q >> 2, value pair q & 3
    const int p = q >> 2, j = q & 3, t = 2 * ks + (j >> 1), r = 2 * (j & 1);
    const float x0 = relu_bits(acc_read(in[p][t][r])), x1 = relu_bits(acc_read(in[p][t][r + 1]));
    split3_pair(x0, x1, fh[p], fm[p], fl[p], 2 * j);
  };
#pragma unroll
  for (int q = 0; q < 8; ++q) build_pair(0, q, bh, bm, bl);
  constexpr int TG = CHAINS == 4 ? 2 : 1;      // feature tiles per group
  frag ah[2][TG], am[2][TG], al[2][TG];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
    for (int t = 0; t < 16; t += TG) {
      const int gi = t, ar = (t / TG) & 1;           // 16 tiles = one stage per k-step
      if (gi == 0) {
        stage_begin(c);
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          ah[ar][u] = *reinterpret_cast<const frag*>(c.buf + (3 * u) * 1024);
          am[ar][u] = *reinterpret_cast<const frag*>(c.buf + (3 * u + 1) * 1024);
          al[ar][u] = *reinterpret_cast<const frag*>(c.buf + (3 * u + 2) * 1024);
        }
      }
      if (gi + TG < 16) {
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          ah[ar ^ 1][u] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + TG + u)) * 1024);
          am[ar ^ 1][u] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + TG + u) + 1) * 1024);
          al[ar ^ 1][u] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + TG + u) + 2) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#define PROD(A, B)                                                                                     \
  _Pragma("unroll") for (int u = 0; u < TG; ++u) _Pragma("unroll") for (int p = 0; p < 2; ++p)          \
      out[p][t + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ar][u], B[p], out[p][t + u], 0, 0, 0);
#if CHAINS == 4      // round-robin over the four chains, product by product
      PROD(al, bh) PROD(ah, bl) PROD(am, bm) PROD(am, bh) PROD(ah, bm) PROD(ah, bh)
#else
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ar][0], bh[p], out[p][t], 0, 0, 0);
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ar][0], bl[p], out[p][t], 0, 0, 0);
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[ar][0], bm[p], out[p][t], 0, 0, 0);
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[ar][0], bh[p], out[p][t], 0, 0, 0);
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ar][0], bm[p], out[p][t], 0, 0, 0);
        out[p][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ar][0], bh[p], out[p][t], 0, 0, 0);
#if CHAINS == 1      // one dependent chain at a time: the six products of a point tile back to back
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#endif
#undef PROD
#pragma unroll
      for (int u = 0; u < TG; ++u)
        if (gi + u < 12) piece(c, gi + u);
      if (ks + 1 < 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q * 2 >= t && q * 2 < t + TG) build_pair(ks + 1, q, nh, nm, nl);      // one value pair per two tiles
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) { bh[p] = nh[p]; bm[p] = nm[p]; bl[p] = nl[p]; }
  }
}

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void chain(const char* stream, const float* seed, float* out, int layer_pairs,
                                                unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[2 * kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
  Ctx c{stream, lds, nullptr, 0, wave_s, lane};
  // stage 0
  for (int q = 0; q < 12; ++q)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(stream + (wave_s * 12 + q) * 1024 + lane * 16),
                                     (__attribute__((address_space(3))) void*)(lds + (wave_s * 12 + q) * 1024), 16, 0, 0);
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 0) {
    f32x16 a[8], b[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) a[t][r] = seed[(blockIdx.x * 256 + tid + 131 * (16 * t + r)) & 65535];
#pragma unroll 1
    for (int it = 0; it < layer_pairs; ++it) {
      layer32(c, b, a);
      layer32(c, a, b);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += a[t][r];
  } else {
    f32x4 a[2][16], b[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[p][t][r] = seed[(blockIdx.x * 256 + tid + 131 * (64 * p + 4 * t + r)) & 65535];
#pragma unroll 1
    for (int it = 0; it < layer_pairs; ++it) {
      layer16(c, b, a);
      layer16(c, a, b);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += a[p][t][r];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

static unsigned short bf16_rne(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; }

int main(int argc, char** argv) {
  const int layer_pairs = argc > 1 ? atoi(argv[1]) : 6, reps = argc > 2 ? atoi(argv[2]) : 5, grid = argc > 3 ? atoi(argv[3]) : 256 * 24;
  // stream: [hi, mid, lo] 1 KiB chunks of He-scaled random weights (uniform, variance 2/256)
  const size_t stream_bytes = (size_t)kStreamStages * kStageBytes;
  std::vector<unsigned short> h(stream_bytes / 2);
  srand(3);
  for (size_t chunk = 0; chunk < stream_bytes / 1024; chunk += 3)
    for (int e = 0; e < 512; ++e) {
      const float w = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.153f;
      const unsigned short hi = bf16_rne(w);
      const float r1 = w - bf16_f(hi);
      const unsigned short mid = bf16_rne(r1);
      const unsigned short lo = bf16_rne(r1 - bf16_f(mid));
      h[chunk * 512 + e] = hi; h[(chunk + 1) * 512 + e] = mid; h[(chunk + 2) * 512 + e] = lo;
    }
  std::vector<float> seed(65536);
  for (auto& x : seed) x = (rand() / (float)RAND_MAX * 2.f - 1.f);
  char* stream; float *dseed, *out; unsigned long long* clk;
  hipMalloc(&stream, stream_bytes); hipMalloc(&dseed, 65536 * 4); hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&clk, 16);
  hipMemcpy(stream, h.data(), stream_bytes, hipMemcpyHostToDevice);
  hipMemcpy(dseed, seed.data(), 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)grid * 4 * layer_pairs * 2 * 6.0 * 2.0 * 256 * 256 * 32;   // bf16 MFMA FLOP issued
  std::vector<float> first(8);
  for (int rep = 0; rep < reps; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      float ms = 0.f;
      hipEventRecord(e0);
      if (shape == 0) chain<0><<<grid, 256>>>(stream, dseed, out, layer_pairs, clk);
      else chain<1><<<grid, 256>>>(stream, dseed, out, layer_pairs, clk);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      unsigned long long ck[2]; hipMemcpy(ck, clk, 16, hipMemcpyDeviceToHost);
      hipMemcpy(first.data(), out, 32, hipMemcpyDeviceToHost);
      printf("rep %d  %-24s %8.3f ms  %7.1f TFLOP/s bf16 issued  (%6.1f algorithmic)  clock %4.0f MHz  cycles/workgroup %llu  out[0] %g\n", rep,
             shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", ms, flop / ms * 1e-9, flop / 6 / ms * 1e-9,
             (double)ck[0] / ck[1] * 100.0, ck[0], first[0]);
    }
  return 0;
}
