// sph_device.hpp -- device-side constants, kernel functions and the neighbour sweep of
// the gfx950 SPH engine.  Written for CDNA4 only (64-wide wavefronts); no portability
// layer.  Reference citations are file:line inside the dieselfluid repository.
//
// Two arithmetic modes, selected at launch by template parameter:
//   EXACT (FAST=false): one IEEE float32 rounding per reference operation, correctly
//     rounded sqrt/divide, double pow -- the translation unit is compiled with
//     -ffp-contract=off so nothing fuses implicitly.
//   FAST  (FAST=true):  explicit fma, v_rcp_f32 / v_rsq_f32, float32 log1p/expm1 EOS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace dsl {

constexpr int kWave = 64;

struct DevConsts {
  int n;
  // kernel/std_kernel.go:20-31
  float h, hh, inv_h, inv_hh, A, B, C, W0;
  float r2_thr;  // smallest float whose correctly rounded square root is >= h: (r2 < r2_thr) == (sqrt(r2) < h)
  float mass, inv_mass, ref_density, mu, dt, delta;
  // model/model.go:92-101
  float eos_wg, eos_gamma, eos_d0_grad;
  float pressure_sign;
  int visc_running_mass;
  float reset[3], ext[3];
  int wcsph_pressure_force, wcsph_viscosity;
  float pci_max_error;
  float xsph_eps, st_kappa;  // build-defined XSPH / cohesion terms, 0 = off
  int walls;
  float bmin[3], bmax[3], rest;
  // uniform grid: cell edge = h, x-fastest linearisation
  float gmin[3];
  float inv_cell;
  float cell;  // cell edge: h, or h (1 + skin) while the skin step owns the grid (kernels_skin.hpp)
  int dims[3];
  int ncell;
  // slab ownership (multi-GPU): axis < 0 = everything owned
  int slab_axis;
  float slab_lo, slab_hi;
  float chk_lo, chk_hi;  // split slab step: integrated coordinates must stay in [chk_lo, chk_hi)
  int split_cl, split_ch;  // split slab step: band cell layers are < split_cl or >= split_ch
  int split_part;          // 0 = integrate everything, 1 = band cell layers only, 2 = the others only
  int own_c0, own_c1;      // cell layers [own_c0, own_c1) along the slab axis can hold owned particles
  // slab mode keeps the live particle count on the device (no host sync per step);
  // nullptr = use n
  const int* n_ptr;
};

// three SoA component arrays
struct Soa3 {
  float *x, *y, *z;
};
struct CSoa3 {
  const float *x, *y, *z;
};

__device__ __forceinline__ int live_n(const DevConsts& c) { return c.n_ptr ? *c.n_ptr : c.n; }

// Boundary particles (particle_array.go:94-128, sph_field.go:75-85): position-only particles behind the
// fluid in the reference's positions slice (index >= N()).  On the device they live in the same sorted
// arrays -- a cell's slots are ascending in particle id, so fluid comes before boundary exactly as in
// the reference's sample lists -- and are told apart by their id.  They are candidates of every
// neighbour sum with Get()'s values (density 0, press 0, velocity 0: rho = 0, P/rho^2 = 0/0 = NaN,
// v = 0 are what the arrays hold for them), never targets: no pass writes them, the integrators
// carry them over unchanged.
struct Bnd {
  const int* ids;  // slot -> particle id; nullptr when the system has no boundary particles
  int n_fluid;
  __device__ __forceinline__ bool is(int slot) const { return ids != nullptr && ids[slot] >= n_fluid; }
};

// Host-visible counters living in device memory (fluid.go:25-26, pcisph_darwin.go:46-98).
struct DevStats {
  unsigned int max_vel_bits;  // max |v|^2 (non-negative float bit patterns order like unsigned ints)
  unsigned int max_f_bits;    // max |F|^2
  unsigned int pci_cur_err_bits;   // running max of the current PCISPH iteration
  unsigned int pci_last_err_bits;  // max_error_ratio of the last completed iteration
  int pci_iters;
  int pci_done;
  int max_cell_count;
  int band_missed;  // split slab step: an interior particle moved further than the split margin
  int reserved0_;
  // slab mode, PCISPH: a DensityF query point (the predictor's position, which the reference never re-synchronises)
  // lies more than h beyond a slab plane, i.e. outside what the 2h ghost band covers: its sum is missing neighbours
  // that live on another rank, and the run no longer equals the single-domain run.  Read with
  // dsl_pcisph_get_binning(h, .., .., &escaped) (blocking); the slab step drivers do not look at it themselves
  int pci_escaped;
};

// The skin step's device-resident state (kernels_skin.hpp; DSL_OPT_SKIN): neighbour LISTS are built against the cut-off
// h (1 + s) and walked for as many steps as no particle can have moved further than s h / 2 since they were built.  That
// decision is the device's -- k_skin_decide, first kernel of every step -- so that the host never waits for it: every
// kernel of the rebuild chain takes a gate and returns at once while `rebuild` is 0.
struct SkinState {
  int rebuild;      // this step rebuilds: counting sort, candidate sweep, lists
  int ids_sel;      // which of the two slot -> particle maps is current (a rebuild flips it on the device)
  float disp;       // the largest displacement of any particle since the lists were built (as of the last decision)
  unsigned int disp2_bits;  // max |x - x_build|^2 after the step just integrated (bits of a non-negative float)
  int force;        // host request: rebuild at the next step whatever the bound says
  int n_steps, n_rebuilds;
  int list_overflow;  // targets whose list did not fit (they take the global-memory sweep)
  float budget;     // s h / 2, less a rounding margin
  float dt;
  unsigned int history;  // bit k: the step k steps ago rebuilt
  int give_up;      // five of the last 16 steps rebuilt: the flow outruns the skin, the lists no longer pay (host: suspend)
  int unlisted;     // targets of the last rebuild that got no list (tiles beyond the LDS budget): they take the global-memory sweep
  int n_live;       // particles
  // of the last rebuild: list fields the targets need, and fields they hold once padded to their wave's longest list
  unsigned int fields_own, fields_padded;
  // Lists are built at REFERENCE positions x + tau v (where the particle will be about half way through the lists' life)
  // rather than where it is at the build: the same budget then covers the way from -tau v to +tau v around the
  // reference -- up to twice the steps (profiles/r04_ballistic_skin.jsonl).  Any reference is sound: displacement is
  // measured against it, and |tau v| <= predict * budget holds at the build because vmax2 is the true maximum.
  float tau;                // time by which this build's reference positions run ahead (0: none)
  float predict;            // fraction of the budget the build itself may use up (DSL_OPT_SKIN_PREDICT; 0: references = positions)
  unsigned int vmax2_bits;  // max |v|^2 of the velocities the last step stored (bits; +inf: unknown)
};
// the gate every kernel of the rebuild chain takes (st == nullptr: no gate, the kernel always runs)
struct SkinGate {
  const SkinState* st;
  __device__ __forceinline__ bool closed() const { return st != nullptr && st->rebuild == 0; }
};

// ---------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ float dsl_sqrt(float x) {
  if constexpr (FAST) return __builtin_amdgcn_sqrtf(x);
  else return __builtin_sqrtf(x);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
}
template <bool FAST>
__device__ __forceinline__ float dsl_div(float a, float b) {
  if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b);
  else return a / b;
}

// IEEE float32 division with the divisor's part of the work shared (DSL_MATH_EXACT walks: three quotients by the same
// dist per pair, and quotients by the constants h and h^2).  `n / d` compiles to v_div_scale x2, v_rcp, two fma that
// refine the reciprocal, q = n r, three fma that correct q, v_div_fmas, v_div_fixup (LLVM's AMDGPU lowering of a
// correctly rounded fdiv).  exact_divisor() is that sequence's divisor half (rcp + its refinement), exact_div() its
// numerator half with v_div_scale left out and v_div_fmas as the plain fma it is when nothing was scaled -- the SAME
// operations on the same values, so the same bits, whenever v_div_scale would have returned its operands unchanged:
// d normal, 1/d normal, n = 0 or |n| >= 2^-103, n/d normal and below 2^96.  exact_div_ok() is the condition the tiled
// EXACT kernels check per staged tile (coordinates and h between 2^-20 and 2^20 in magnitude, or exactly 0: every
// nonzero coordinate difference is then >= 2^-43, every distance <= 2^22); a tile that fails it takes the
// global-memory sweep, which divides with `/`.
struct ExactDivisor {
  float d, r1;
};
__device__ __forceinline__ ExactDivisor exact_divisor(float d) {
  const float r0 = __builtin_amdgcn_rcpf(d);
  const float e0 = __builtin_fmaf(-d, r0, 1.0f);
  return ExactDivisor{d, __builtin_fmaf(e0, r0, r0)};
}
// the same for a wave-uniform divisor (h, h^2): both halves live in scalar registers
__device__ __forceinline__ ExactDivisor exact_divisor_uniform(float d) {
  const ExactDivisor D = exact_divisor(d);
  return ExactDivisor{__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(D.d))),
                      __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(D.r1)))};
}
__device__ __forceinline__ float exact_div(float n, const ExactDivisor& D) {
  const float q0 = n * D.r1;
  const float e1 = __builtin_fmaf(-D.d, q0, n);
  const float q1 = __builtin_fmaf(e1, D.r1, q0);
  const float e2 = __builtin_fmaf(-D.d, q1, n);
  const float q2 = __builtin_fmaf(e2, D.r1, q1);
  return __builtin_amdgcn_div_fixupf(q2, D.d, n);
}
__device__ __forceinline__ bool exact_div_ok(float v) {
  const float a = fabsf(v);
  return v == 0.0f || (a >= 0x1p-20f && a <= 0x1p20f);  // (false for NaN / Inf)
}

// kernel/std_kernel.go:33-39  F(x) = x>=h ? 0 : A*q*q, q = 1 - x*x/(h*h)
template <bool FAST>
__device__ __forceinline__ float kern_F(const DevConsts& c, float dist) {
  if (dist >= c.h) return 0.0f;
  float xx = dist * dist;
  float q;
  if constexpr (FAST) q = __builtin_fmaf(-xx, c.inv_hh, 1.0f);
  else q = 1.0f - xx / c.hh;
  float aq = c.A * q;
  return aq * q;
}
// kernel/std_kernel.go:54-60  O1D(x) = x>=h ? 0 : B*q*q, q = 1 - x/h
template <bool FAST>
__device__ __forceinline__ float kern_O1D(const DevConsts& c, float dist) {
  if (dist >= c.h) return 0.0f;
  float q;
  if constexpr (FAST) q = __builtin_fmaf(-dist, c.inv_h, 1.0f);
  else q = 1.0f - dist / c.h;
  float bq = c.B * q;
  return bq * q;
}
// kernel/std_kernel.go:63-71  O2D(x) = x>h ? 0 : C*q
template <bool FAST>
__device__ __forceinline__ float kern_O2D(const DevConsts& c, float dist) {
  if (dist > c.h) return 0.0f;
  float q;
  if constexpr (FAST) q = __builtin_fmaf(-dist, c.inv_h, 1.0f);
  else q = 1.0f - dist / c.h;
  return c.C * q;
}

// model/model.go:92-101 TaitEos(x, d0, 0)
template <bool FAST>
__device__ __forceinline__ float tait_eos(const DevConsts& c, float x, float d0) {
  if (x <= d0) x = d0;
  float ratio = dsl_div<FAST>(x, d0);
  float pw;
  if constexpr (FAST) {
    // (1+e)^g - 1 without the cancellation of powf(...)-1: expm1(g log1p(e)).  Weakly compressible
    // flow keeps e below a few per cent, so the common case is a short series (relative error 1e-6
    // for e <= 0.5, checked against float64) instead of two library calls (~150 instructions).
    const float e = ratio - 1.0f;
    if (e <= 0.5f) {
      const float t = e * __builtin_amdgcn_rcpf(2.0f + e);  // log1p(e) = 2 atanh(e / (2 + e))
      const float t2 = t * t;
      float pl = 1.0f / 9.0f;
      pl = __builtin_fmaf(pl, t2, 1.0f / 7.0f);
      pl = __builtin_fmaf(pl, t2, 1.0f / 5.0f);
      pl = __builtin_fmaf(pl, t2, 1.0f / 3.0f);
      pl = __builtin_fmaf(pl, t2, 1.0f);
      const float u = c.eos_gamma * ((2.0f * t) * pl);
      float pe = 1.0f / 720.0f;  // expm1(u), u <= 1/8: series
      pe = __builtin_fmaf(pe, u, 1.0f / 120.0f);
      pe = __builtin_fmaf(pe, u, 1.0f / 24.0f);
      pe = __builtin_fmaf(pe, u, 1.0f / 6.0f);
      pe = __builtin_fmaf(pe, u, 0.5f);
      pe = __builtin_fmaf(pe, u, 1.0f);
      pw = u <= 0.125f ? u * pe : __builtin_amdgcn_exp2f(u * 1.4426950408889634f) - 1.0f;
    } else {
      pw = expm1f(c.eos_gamma * log1pf(e));
    }
  } else {
    pw = (float)(pow((double)ratio, (double)c.eos_gamma) - 1.0);
  }
  float y = c.eos_wg * pw;
  return y + 0.0f;
}

// Cell coordinate rule (DESIGN.md "grid"): floor((p-gmin)*inv_cell) clamped; NaN -> 0.
__device__ __forceinline__ int cell_coord(float p, float gmin, float inv_cell, int dim) {
  float f = floorf((p - gmin) * inv_cell);
  if (!(f >= 0.0f)) return 0;
  if (f >= (float)dim) return dim - 1;
  return (int)f;
}
__device__ __forceinline__ int cell_of(const DevConsts& c, float x, float y, float z) {
  int cx = cell_coord(x, c.gmin[0], c.inv_cell, c.dims[0]);
  int cy = cell_coord(y, c.gmin[1], c.inv_cell, c.dims[1]);
  int cz = cell_coord(z, c.gmin[2], c.inv_cell, c.dims[2]);
  return (cz * c.dims[1] + cy) * c.dims[0] + cx;
}

// slab ownership: ghosts (outside [lo,hi) along the slab axis, or NaN) are never integrated
__device__ __forceinline__ bool slab_owned(const DevConsts& c, float x, float y, float z) {
  if (c.slab_axis < 0) return true;
  const float p = c.slab_axis == 0 ? x : (c.slab_axis == 1 ? y : z);
  return p >= c.slab_lo && p < c.slab_hi;  // false for NaN
}

// slab mode: has a PCISPH query point left the region this rank's particles + 2h ghost band cover (see DevStats)?
__device__ __forceinline__ bool pci_query_escaped(const DevConsts& c, float qx, float qy, float qz) {
  if (c.slab_axis < 0) return false;
  const float q = c.slab_axis == 0 ? qx : (c.slab_axis == 1 ? qy : qz);
  return q < c.slab_lo - c.h || q >= c.slab_hi + c.h;
}

// Split slab step: the force pass first integrates the particles of the BAND cell layers --
// cell index < split_cl or >= split_ch along the slab axis: everything within width + margin of
// a slab plane -- then, while that band is packed and sent, the particles of the other layers.
// The same rule, on a particle's cell before the step, tells the band pack which slots the first
// launch has integrated.
constexpr int kTB = 4;  // tile edge in cells (kernels_tiled.hpp)
__device__ __forceinline__ bool slab_band_cell(const DevConsts& c, int ca) { return ca < c.split_cl || ca >= c.split_ch; }
__device__ __forceinline__ int slab_axis_cell(const DevConsts& c, float x, float y, float z) {
  const int a = c.slab_axis;
  return cell_coord(a == 0 ? x : (a == 1 ? y : z), c.gmin[a], c.inv_cell, c.dims[a]);
}

// Where neighbour candidates come from: the uniform grid (cell_start) or the reference's
// LSH sampler (sampler/lsh/lsh.go): one list of 100 samples per bucket, the bucket chosen by
// the random-projection hash of the query position.
constexpr int kLshSamples = 100;  // lsh.go:17
struct Neigh {
  const int* cell_start;
  const int* samples;  // [buckets][kLshSamples]; non-null selects LSH mode
  const float* hashv;  // [hash_bits][3]
  int hash_bits, buckets;
};
__device__ __forceinline__ Neigh grid_neigh(const int* cell_start) { return Neigh{cell_start, nullptr, nullptr, 0, 0}; }

// lsh.go:51-56 sgn, :102-111 Hash (dot product left to right, no fma: vector.go:268-276)
__device__ __forceinline__ int lsh_hash(const Neigh& nb, float x, float y, float z) {
  long long hash = 0;
  for (int i = 0; i < nb.hash_bits; ++i) {
    const float t0 = x * nb.hashv[3 * i], t1 = y * nb.hashv[3 * i + 1], t2 = z * nb.hashv[3 * i + 2];
    const float d = (t0 + t1) + t2;
    hash = (hash << 1) + (d <= 0.0f ? 0 : 1);
  }
  return (int)(hash % nb.buckets);
}

// In lsh_ref mode the reference relies on the kernel cut-offs alone (duplicates and
// non-neighbours are visited); the geometric mode defines the neighbour set as dist < h.
__device__ __forceinline__ bool in_support(const DevConsts& c, const Neigh& nb, float dist) {
  return nb.samples != nullptr || dist < c.h;
}

template <class Body>
__device__ __forceinline__ void for_each_grid_candidate(const DevConsts& c, const int* __restrict__ cell_start,
                                                        float x, float y, float z, Body&& body);

// Candidate sweep.  LSH: the 100 samples of the query's bucket in list order (lsh.go:136-181).
// Grid: the 27 cells around (x,y,z) as 9 x-runs; thanks to the x-fastest linearisation the
// three cells of a run are one contiguous slot range.  Order: z, y, then slots ascending --
// the same order the oracle's DSLO_ORDER_CELL uses.
template <class Body>
__device__ __forceinline__ void for_each_candidate(const DevConsts& c, const Neigh& nb, float x, float y, float z,
                                                   Body&& body) {
  if (nb.samples != nullptr) {
    const int* l = nb.samples + (size_t)lsh_hash(nb, x, y, z) * kLshSamples;
    for (int k = 0; k < kLshSamples; ++k) body(l[k]);
  } else {
    for_each_grid_candidate(c, nb.cell_start, x, y, z, body);
  }
}

template <class Body>
__device__ __forceinline__ void for_each_grid_candidate(const DevConsts& c, const int* __restrict__ cell_start,
                                                        float x, float y, float z, Body&& body) {
  const int nx = c.dims[0], ny = c.dims[1], nz = c.dims[2];
  const int cx = cell_coord(x, c.gmin[0], c.inv_cell, nx);
  const int cy = cell_coord(y, c.gmin[1], c.inv_cell, ny);
  const int cz = cell_coord(z, c.gmin[2], c.inv_cell, nz);
  const int x0 = cx > 0 ? cx - 1 : 0;
  const int x1 = cx < nx - 1 ? cx + 1 : nx - 1;
  for (int dz = -1; dz <= 1; ++dz) {
    const int zz = cz + dz;
    if (zz < 0 || zz >= nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = cy + dy;
      if (yy < 0 || yy >= ny) continue;
      const int row = (zz * ny + yy) * nx;
      const int jb = cell_start[row + x0];
      const int je = cell_start[row + x1 + 1];
      for (int j = jb; j < je; ++j) body(j);
    }
  }
}

// squared distance in the reference's rounding order (vector.go:301-308 Mag)
template <bool FAST>
__device__ __forceinline__ float dist2(float dx, float dy, float dz) {
  if constexpr (FAST) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  else {
    float s = dx * dx;
    float t = dy * dy;
    s = s + t;
    t = dz * dz;
    return s + t;
  }
}

// wave-level max of non-negative float bit patterns.  The global is only touched when it
// would grow: same-address atomics serialise in L2 (500k of them per 16M-particle step cost
// milliseconds, profiles/r01_v2_pmc.md), while the running maximum settles after a few waves.
__device__ __forceinline__ void wave_atomic_max(unsigned int* addr, unsigned int v) {
  for (int off = kWave / 2; off > 0; off >>= 1) {
    unsigned int o = __shfl_xor(v, off, kWave);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && v != 0u) {
    const unsigned int cur = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v > cur) atomicMax(addr, v);
  }
}
__device__ __forceinline__ unsigned int nonneg_bits(float v) {
  return v > 0.0f ? __float_as_uint(v) : 0u;  // NaN and <=0 contribute nothing
}

}  // namespace dsl
