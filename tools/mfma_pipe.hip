// mfma_pipe.hip -- VERDICT r02 item 2: can the density sweep's distance test move to the f32 matrix pipe if the
// loop is SOFTWARE-PIPELINED?  The r02 probe (tools/mfma_rate.hip) fed the MFMAs from VALU results of the same
// trip and consumed their outputs at once: one serial chain per wave.  Here the MFMA operand comes straight
// from a ds_read, B and C are loop constants (the 16 targets' coefficients and their a0), and the VALU
// post-processing (v_max, v_fmac, v_alignbit per output: 12 per MFMA) works on the outputs of the PREVIOUS trip.
//
// One v_mfma_f32_16x16x4_f32 = 16 candidates x 16 targets = 256 pair tests (q = a0 + w_j + (2/h^2) x_i.x_j as
// D = A(16x4 candidates) * B(4x16 targets) + C).  A trip here does two of them: 512 pair tests.  The VALU sweep
// of k_density_tiled does 512 pair tests with 8 ds_read_b128 + 56 VALU (MODE 4, the product's own loop body).
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_pipe.hip -o /tmp/mfma_pipe && /tmp/mfma_pipe
//   rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES \
//             --kernel-trace -d out -- /tmp/mfma_pipe
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int ITER = 4096;
constexpr int kRec = 2048;  // float4 records in LDS (32 KB)

__device__ __forceinline__ void post(const f32x4_t& d, float (&acc)[4], unsigned int (&mask)[4]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {  // exactly three VALU instructions per output
    float q;
    const float dv = d[v];
    asm("v_max_f32_e32 %0, 0, %1" : "=v"(q) : "v"(dv));
    asm("v_fmac_f32_e32 %0, %1, %1" : "+v"(acc[v]) : "v"(q));
    asm("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mask[v]) : "v"(dv));  // mask = 2 mask + sign(d)
  }
}

// MODE 0: two MFMAs per trip, operands from LDS, outputs only xor-folded (the matrix pipe alone)
// MODE 1: the 24 VALU of two MFMAs' post-processing on values read from LDS (the VALU side alone)
// MODE 2: both, the VALU on the outputs of the PREVIOUS trip (software-pipelined)
// MODE 3: both, the VALU on this trip's outputs (the r02 shape)
// MODE 4: the product's VALU sweep for the same 512 pair tests: 8 x (ds_read_b128 + 7 VALU)
// MODE 5: two targets per lane: 4 x (ds_read_b128 + 14 VALU) -- the same 512 pair tests with half the LDS reads
// MODE 6: MODE 4's 56 VALU on one record read once per trip (the VALU side of the product loop alone)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float bcoef, float a0) {
  __shared__ float4 lds[kRec];
  for (int i = threadIdx.x; i < kRec; i += 256)
    lds[i] = make_float4(0.01f * (i & 63), 0.02f * (i & 31), -0.01f * (i & 15), -0.3f - 1e-3f * (i & 7));
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float* ldsf = reinterpret_cast<const float*>(lds);
  // A operand: lane l reads component l/16 of candidate l%16 of the block: 64 consecutive dwords
  const int aoff = (lane & 15) * 4 + (lane >> 4);
  const f32x4_t cin = {a0, a0, a0, a0};
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, acc2[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned int m0[4] = {0u, 0u, 0u, 0u}, m1[4] = {0u, 0u, 0u, 0u};
  f32x4_t p0 = cin, p1 = cin;
  unsigned int fold = 0u;
  int base = (threadIdx.x >> 6) * 64;
  for (int it = 0; it < ITER; ++it) {
    base = (base + 64) & (kRec - 64);
    if constexpr (MODE == 0 || MODE == 2 || MODE == 3) {
      const float a_0 = ldsf[base * 4 + aoff], a_1 = ldsf[(base + 16) * 4 + aoff];
      const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_0, bcoef, cin, 0, 0, 0);
      const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_1, bcoef, cin, 0, 0, 0);
      if constexpr (MODE == 0) {
        fold ^= __float_as_uint(d0[0]) ^ __float_as_uint(d1[3]);
      } else if constexpr (MODE == 2) {
        post(p0, acc, m0);
        post(p1, acc2, m1);
        p0 = d0;
        p1 = d1;
      } else {
        post(d0, acc, m0);
        post(d1, acc2, m1);
      }
    }
    if constexpr (MODE == 1) {
      const float4 r0 = lds[base + lane], r1 = lds[base + 64 - lane];
      post(f32x4_t{r0.x, r0.y, r0.z, r0.w}, acc, m0);
      post(f32x4_t{r1.x, r1.y, r1.z, r1.w}, acc2, m1);
    }
    if constexpr (MODE == 4 || MODE == 5 || MODE == 6) {
      const float sx = bcoef, sy = bcoef * 1.1f, sz = bcoef * 0.9f;
      const float tx = bcoef * 0.7f, ty = bcoef * 1.3f, tz = bcoef * 0.8f, b0 = a0 * 0.5f;
      auto test = [&](const float4& cnd, float ux, float uy, float uz, float u0, unsigned int& mk, float& ac) {
        float q;
        const float t = __builtin_fmaf(cnd.y, uy, __builtin_fmaf(cnd.x, ux, cnd.w + u0));
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(q) : "v"(cnd.z), "v"(uz), "v"(t));
        asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mk) : "v"(q) : "vcc");
        ac = __builtin_fmaf(q, q, ac);
      };
      float4 once = lds[base + (lane >> 3)];
#pragma unroll
      for (int u = 0; u < (MODE == 5 ? 4 : 8); ++u) {
        float4 cnd = once;
        if constexpr (MODE != 6) cnd = lds[base + u + (lane >> 3)];
        else asm volatile("" : "+v"(cnd.x));  // (keeps the eight tests apart)
        test(cnd, sx, sy, sz, a0, m0[0], acc[u & 1]);
        if constexpr (MODE == 5) test(cnd, tx, ty, tz, b0, m1[0], acc2[u & 1]);
      }
    }
  }
  float s = p0[0] + p1[1];
  for (int v = 0; v < 4; ++v) s += acc[v] + acc2[v] + (float)(m0[v] ^ m1[v]);
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)fold;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256 * waves_per_simd;  // one 4-wave block per CU and wave-per-SIMD step
  hipMalloc(&d, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 0.37f, 0.25f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, 0.37f, 0.25f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double trips_per_simd = (double)ITER * waves_per_simd;
  printf("%-58s waves/SIMD %d : %.3f ms -> %6.1f clk per 512 pair tests per SIMD @2.4GHz\n", name, waves_per_simd, ms,
         ms * 1e6 / trips_per_simd * 2.4);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("2 x mfma_16x16x4 from ds_read (matrix pipe alone)", w);
    run<1>("24 VALU post-processing alone", w);
    run<2>("2 x mfma + 24 VALU, software-pipelined", w);
    run<3>("2 x mfma + 24 VALU on the same trip's outputs", w);
    run<4>("product loop: 8 x (ds_read_b128 + 7 VALU)", w);
    run<5>("two targets per lane: 4 x (ds_read_b128 + 14 VALU)", w);
    run<6>("product loop's 56 VALU, one ds_read_b128 per trip", w);
  }
  return 0;
}
