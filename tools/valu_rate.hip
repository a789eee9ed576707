// valu_rate.hip -- microbenchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 (and v_max, v_rsq)
// on gfx950.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, UNROLL = 16;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x[8];
  v2f y[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = v2f{x[i], x[i] + 0.5f}; }
  v2f a2{a, a}, b2{b, b};
  unsigned int msk = 0u;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int i = u & 7;
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
      if (MODE == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      if (MODE == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]));
      if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
      if (MODE == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
      // distinct source registers (as in a real loop): d = s0*s1 + s2 with s0,s1,s2 all different
      if (MODE == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(y[i]) : "v"(y[(i + 1) & 7]), "v"(y[(i + 2) & 7]), "v"(y[(i + 3) & 7]));
      if (MODE == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]), "v"(x[(i + 3) & 7]));
      if (MODE == 8) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y[i]) : "v"(y[(i + 1) & 7]), "v"(y[(i + 2) & 7]));
      if (MODE == 9) asm volatile("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(y[i]) : "v"(y[(i + 1) & 7]), "v"(y[(i + 3) & 7]));
      if (MODE == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(y[i]) : "v"(y[(i + 1) & 7]), "v"(y[(i + 2) & 7]), "v"(y[(i + 3) & 7]));
      // the density sweep's candidate test as in kernels_tiled.hpp (7 instructions: add, 3 fma the last one clamped,
      // compare + add-with-carry into the mask, fma into one of two accumulators), operands in registers, no LDS
      if (MODE == 11) {
        float t;
        asm volatile("v_add_f32 %0, %3, %4\n\tv_fmac_f32 %0, %5, %6\n\tv_fmac_f32 %0, %7, %8\n\tv_fma_f32 %0, %9, %10, %0 clamp\n\t"
                     "v_cmp_lt_f32 vcc, 0, %0\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\tv_fmac_f32 %2, %0, %0"
                     : "=&v"(t), "+v"(msk), "+v"(x[u & 1])
                     : "v"(x[2]), "v"(x[3 + (u & 3)]), "v"(x[7]), "v"(a), "v"(x[6]), "v"(b), "v"(x[5]), "v"(y[u & 7].x)
                     : "vcc");
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
  s += (float)msk;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd blocks of 4 waves)
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
  const double instr_per_simd = (double)ITER * UNROLL * waves_per_simd * (MODE == 11 ? 7 : 1);  // wave-instructions per SIMD
  printf("%-14s waves/SIMD %d : %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f clk @2.4GHz)\n", name,
         waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(d);
}

int main() {
  for (int w : {2, 4, 8}) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32", w);
    run<2>("v_max_f32", w);
    run<3>("v_rsq_f32", w);
    run<4>("v_pk_add_f32", w);
    run<5>("v_pk_mul_f32", w);
    run<6>("pk_fma 3src", w);
    run<7>("fma 3src", w);
    run<8>("pk_add 2src", w);
    run<9>("pk_fma a*a+c", w);
    run<10>("pk_fma neg", w);
    run<11>("density test", w);
  }
  return 0;
}
