// mfma_layout.hip -- which lane's A and B operand end up in D[v] of lane L for
// v_mfma_f32_4x4x1_16b_f32 on gfx950?  k_density_quad (kernels_tiled.hpp) relies on
//   D[v] (lane L) = A (lane 4*(L/4) + v) * B (lane L) + C[v].
// Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_layout tools/mfma_layout.hip && tools/mfma_layout
// Exit code 0 and "layout as assumed" when it holds.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void probe(float* out_a, float* out_b, float* out_c) {
  const int lane = threadIdx.x;
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  // A = lane + 1, B = 1: D[v] names the lane whose A was used
  f32x4_t da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 1.0f, zero, 0, 0, 0);
  // A = 1, B = lane + 1: D[v] names the lane whose B was used
  f32x4_t db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(lane + 1), zero, 0, 0, 0);
  // C pass-through: C[v] = 10 v + 1 with A = 0
  const f32x4_t cc = {1.f, 11.f, 21.f, 31.f};
  f32x4_t dc = __builtin_amdgcn_mfma_f32_4x4x1f32(0.0f, 0.0f, cc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) {
    out_a[lane * 4 + v] = da[v];
    out_b[lane * 4 + v] = db[v];
    out_c[lane * 4 + v] = dc[v];
  }
}

int main() {
  float *da, *db, *dc;
  hipMalloc(&da, 256 * sizeof(float));
  hipMalloc(&db, 256 * sizeof(float));
  hipMalloc(&dc, 256 * sizeof(float));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dc);
  float ha[256], hb[256], hc[256];
  hipMemcpy(ha, da, sizeof(ha), hipMemcpyDeviceToHost);
  hipMemcpy(hb, db, sizeof(hb), hipMemcpyDeviceToHost);
  hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int a_lane = (int)ha[l * 4 + v] - 1, b_lane = (int)hb[l * 4 + v] - 1;
      if (a_lane != 4 * (l / 4) + v || b_lane != l || hc[l * 4 + v] != 10.f * v + 1.f) ++bad;
    }
  for (int l = 0; l < 8; ++l)
    std::printf("lane %d: A from lanes %g %g %g %g | B from lanes %g %g %g %g | C %g %g %g %g\n", l, ha[l * 4] - 1,
                ha[l * 4 + 1] - 1, ha[l * 4 + 2] - 1, ha[l * 4 + 3] - 1, hb[l * 4] - 1, hb[l * 4 + 1] - 1, hb[l * 4 + 2] - 1,
                hb[l * 4 + 3] - 1, hc[l * 4], hc[l * 4 + 1], hc[l * 4 + 2], hc[l * 4 + 3]);
  std::printf(bad ? "LAYOUT DIFFERS from the assumption (%d mismatches)\n" : "layout as assumed (%d mismatches)\n", bad);
  return bad ? 1 : 0;
}
