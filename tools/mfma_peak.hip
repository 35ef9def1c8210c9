// Sustained (power-limited) MFMA rate of the device: back-to-back v_mfma on register operands holding random data,
// nothing else in the loop.  Gives the ceiling a real kernel can be priced against next to the nominal 2.4 GHz peak.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void spin(const float* seed, float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x;
  f16v acc[4];
  f4v acq[4];      // 16x16 accumulators (KIND 3, 4)
  for (int t = 0; t < 4; ++t) {
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int i = 0; i < 4; ++i) acq[t][i] = 0.f;
  }
  h8 ah[2], bh[2];
  b8 ab[2], bb[2];
  float a32[2], b32[2];
  for (int j = 0; j < 2; ++j) {
    for (int i = 0; i < 8; ++i) {
      const float x = seed[(lane * 37 + j * 8 + i) & 4095], y = seed[(lane * 53 + 17 + j * 8 + i) & 4095];
      ah[j][i] = (_Float16)x; bh[j][i] = (_Float16)y;
      ab[j][i] = (__bf16)x; bb[j][i] = (__bf16)y;
    }
    a32[j] = seed[(lane + j) & 4095]; b32[j] = seed[(lane * 3 + j) & 4095];
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[u & 1], bh[(u >> 1) & 1], acc[t], 0, 0, 0);
        if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[u & 1], bb[(u >> 1) & 1], acc[t], 0, 0, 0);
        if (KIND == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[u & 1], b32[(u >> 1) & 1], acc[t], 0, 0, 0);
        if (KIND == 3) acq[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[u & 1], bb[(u >> 1) & 1], acq[t], 0, 0, 0);
        if (KIND == 4) acq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u & 1], bh[(u >> 1) & 1], acq[t], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 16; ++i) s += acc[t][i] + acq[t][i & 3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main() {
  std::vector<float> h(4096);
  srand(1);
  for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
  float *seed, *out; unsigned long long* clk;
  hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[5] = {"f32_32x32x16_f16", "f32_32x32x16_bf16", "f32_32x32x2_f32", "f32_16x16x32_bf16", "f32_16x16x32_f16"};
  const double flop_per[5] = {2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 16, 2.0 * 32 * 32 * 2, 2.0 * 16 * 16 * 32, 2.0 * 16 * 16 * 32};
  for (int kind = 0; kind < 5; ++kind)
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
      const int grid = 256 * wgs_per_cu, iters = kind == 2 ? 20000 : (kind >= 3 ? 80000 : 40000);
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {      // last repetition is reported (clock has settled)
        hipEventRecord(e0);
        if (kind == 0) spin<0><<<grid, 256>>>(seed, out, iters, clk);
        if (kind == 1) spin<1><<<grid, 256>>>(seed, out, iters, clk);
        if (kind == 2) spin<2><<<grid, 256>>>(seed, out, iters, clk);
        if (kind == 3) spin<3><<<grid, 256>>>(seed, out, iters, clk);
        if (kind == 4) spin<4><<<grid, 256>>>(seed, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flops = (double)grid * 4 * iters * 32 * flop_per[kind];
      printf("%-18s %d waves/SIMD: %.1f ms  %.1f TFLOP/s  engine clock %.0f MHz\n", names[kind], wgs_per_cu, ms,
             flops / ms * 1e-9, (double)c[0] / c[1] * 100.0);
    }
  return 0;
}
