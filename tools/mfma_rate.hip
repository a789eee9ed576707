// mfma_rate.hip -- issue cost of v_mfma_f32_4x4x1_16b_f32 on gfx950, alone, in dependent chains of 4
// (the K = w, x, y, z chain of k_density_quad) and next to VALU work.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate && tools/mfma_rate
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

// MODE 0: 8 independent MFMAs per trip (8 accumulators)
// MODE 1: 2 chains of 4 dependent MFMAs per trip
// MODE 2: MODE 1 + 24 independent VALU (v_fma) per trip
// MODE 3: 24 VALU per trip only
// MODE 4: 1 chain: 16x16x4 (one instruction does K = 4) x2 per trip
// MODE 5: MODE 4 + 24 VALU
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  f32x4_t acc[8];
  float x[8];
  for (int i = 0; i < 8; ++i) {
    acc[i] = f32x4_t{a, b, a, b};
    x[i] = threadIdx.x * 1e-3f + i;
  }
  const f32x4_t cin = {a, a, a, a};
  for (int it = 0; it < ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
      f32x4_t d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[0], b, cin, 0, 0, 0);
      f32x4_t d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[1], b, cin, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[2], a, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[3], a, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[4], b, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[5], b, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[6], a, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x[7], a, d1, 0, 0, 0);
      acc[0] += d0;
      acc[1] += d1;
    }
    if (MODE == 4 || MODE == 5) {
      f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0], b, cin, 0, 0, 0);
      f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1], b, cin, 0, 0, 0);
      acc[0] += d0;
      acc[1] += d1;
    }
    if (MODE == 2 || MODE == 3 || MODE == 5) {
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        const int i = u & 7;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]), "v"(x[(i + 3) & 7]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256 * waves_per_simd;
  hipMalloc(&d, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 1e-7f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, 1.0001f, 1e-7f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double trips_per_simd = (double)ITER * waves_per_simd;
  printf("%-34s waves/SIMD %d : %.3f ms -> %.1f clk per trip per SIMD @2.4GHz\n", name, waves_per_simd, ms,
         ms * 1e6 / trips_per_simd * 2.4);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0>("8 independent 4x4x1", w);
    run<1>("2 chains of 4 4x4x1 (+8 v_add)", w);
    run<2>("2 chains of 4 4x4x1 + 24 v_fma", w);
    run<3>("24 v_fma", w);
    run<4>("2 x 16x16x4 (+8 v_add)", w);
    run<5>("2 x 16x16x4 + 24 v_fma", w);
  }
  return 0;
}
