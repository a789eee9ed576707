// kernels_sph.hpp -- the SPH passes of model/sph/fluid.go as gfx950 kernels over the
// cell-sorted SoA arrays.  One lane per particle slot; the neighbour sweep walks the
// nine x-runs of the 27 surrounding cells (sph_device.hpp).  All passes are Jacobi:
// they read state the pass itself never writes (the fused force+integrate kernel writes
// the other half of the ping-pong pair).
#pragma once

#include "kernels_grid.hpp"

namespace dsl {

// ---------------------------------------------------------------------------------
// D: SPHField.Density (model/field/sph_field.go:155-172) for every slot, plus the
// per-particle pressure term P(rho)/rho^2 of SPHField.Gradient (sph_field.go:192-196,
// field_types.go:39-42) so the force sweep does not call pow per neighbour.
// ---------------------------------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_density(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p,
                                                    float* __restrict__ rho, float* __restrict__ pterm) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  if (bnd.is(i)) {  // Get(): density 0, PressureField.Value(0)/0^2 = 0/0 (field_types.go:39-42)
    rho[i] = 0.0f;
    pterm[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  const float xi = p.x[i], yi = p.y[i], zi = p.z[i];
  float density = 0.0f;
  for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
    if (j == i) return;
    const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
    const float r2 = dist2<FAST>(dx, dy, dz);
    if constexpr (FAST) {
      if (r2 < c.hh) {
        const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
        density = __builtin_fmaf(c.mass * c.A, q * q, density);
      }
    } else {
      const float dist = dsl_sqrt<false>(r2);
      if (in_support(c, nb, dist)) {
        const float w = kern_F<false>(c, dist);
        density += c.mass * w;
      }
    }
  });
  rho[i] = density;
  const float pr = tait_eos<FAST>(c, density, c.eos_d0_grad);
  pterm[i] = dsl_div<FAST>(pr, density * density);
}

// PR: SPH.PressureAll (fluid.go:134-142): Press = TaitEos(rho, D0, 0)
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_pressure(DevConsts c, Bnd bnd, const float* __restrict__ rho,
                                                     float* __restrict__ press) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  if (bnd.is(i)) return;  // PressureAll loops over N() particles (fluid.go:134-142)
  press[i] = tait_eos<FAST>(c, rho[i], c.ref_density);
}

// P(rho)/rho^2 for densities that arrive by upload instead of from k_density
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_pterm(DevConsts c, Bnd bnd, const float* __restrict__ rho,
                                                  float* __restrict__ pterm) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  if (bnd.is(i)) return;
  const float d = rho[i];
  pterm[i] = dsl_div<FAST>(tait_eos<FAST>(c, d, c.eos_d0_grad), d * d);
}

// ---------------------------------------------------------------------------------
// Shared neighbour sweep for G (sph_field.go:175-200) and V (sph_field.go:251-269).
// accG accumulates grad*(Pi/rho_i^2 + Pj/rho_j^2); accV is LaplacianForce's running sum.
// ---------------------------------------------------------------------------------
// accX / accS (optional, both or none): the build-defined XSPH sum_j (m/rho_j)(v_j - v_i) F(r)
// and cohesion sum_j m (x_j - x_i) F(r) terms; they need v even when WANT_V is false.
template <bool FAST, bool WANT_G, bool WANT_V>
__device__ __forceinline__ void force_sweep(const DevConsts& c, Neigh nb, int i,
                                            const CSoa3& p, const CSoa3& v, const float* __restrict__ rho,
                                            const float* __restrict__ pterm, float accG[3], float accV[3],
                                            float* accX = nullptr, float* accS = nullptr) {
  const float xi = p.x[i], yi = p.y[i], zi = p.z[i];
  float vxi = 0.f, vyi = 0.f, vzi = 0.f, pti = 0.f;
  const bool xs = accX != nullptr;
  if (WANT_V || xs) {
    vxi = v.x[i];
    vyi = v.y[i];
    vzi = v.z[i];
  }
  if constexpr (WANT_G) pti = pterm[i];
  for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
    if (j == i) return;
    // dir = x_j - x_i (sph_field.go:189); |x_i - x_j| has the same squares
    const float dx = p.x[j] - xi, dy = p.y[j] - yi, dz = p.z[j] - zi;
    const float r2 = dist2<FAST>(dx, dy, dz);
    if constexpr (FAST) {
      if (!(r2 < c.hh)) return;
      if (xs) {
        const float q2 = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
        const float fw = c.A * q2 * q2;
        const float ws = c.mass * fw, wx = ws * __builtin_amdgcn_rcpf(rho[j]);
        accS[0] = __builtin_fmaf(dx, ws, accS[0]);
        accS[1] = __builtin_fmaf(dy, ws, accS[1]);
        accS[2] = __builtin_fmaf(dz, ws, accS[2]);
        accX[0] = __builtin_fmaf(v.x[j] - vxi, wx, accX[0]);
        accX[1] = __builtin_fmaf(v.y[j] - vyi, wx, accX[1]);
        accX[2] = __builtin_fmaf(v.z[j] - vzi, wx, accX[2]);
      }
      const float rinv = __builtin_amdgcn_rsqf(r2);  // r2 == 0 -> inf, handled below
      const float dist = r2 * rinv;
      const float q = __builtin_fmaf(-(r2 > 0.f ? dist : 0.f), c.inv_h, 1.0f);
      if constexpr (WANT_G) {
        const float s = -(c.B * q) * q;  // -O1D
        const float F = pti + pterm[j];
        const float k = (r2 > 0.f) ? s * F * rinv : 0.0f;  // Norm() of a zero vector is zero
        accG[0] = __builtin_fmaf(dx, k, accG[0]);
        accG[1] = __builtin_fmaf(dy, k, accG[1]);
        accG[2] = __builtin_fmaf(dz, k, accG[2]);
      }
      if constexpr (WANT_V) {
        const float w = (c.C * q) * __builtin_amdgcn_rcpf(rho[j]);
        if (c.visc_running_mass) {
          accV[0] = __builtin_fmaf(v.x[j] - vxi, w, accV[0]) * c.mass;
          accV[1] = __builtin_fmaf(v.y[j] - vyi, w, accV[1]) * c.mass;
          accV[2] = __builtin_fmaf(v.z[j] - vzi, w, accV[2]) * c.mass;
        } else {
          const float wm = w * c.mass;
          accV[0] = __builtin_fmaf(v.x[j] - vxi, wm, accV[0]);
          accV[1] = __builtin_fmaf(v.y[j] - vyi, wm, accV[1]);
          accV[2] = __builtin_fmaf(v.z[j] - vzi, wm, accV[2]);
        }
      }
    } else {
      const float dist = dsl_sqrt<false>(r2);
      if (!in_support(c, nb, dist)) return;
      if (xs) {
        const float fw = kern_F<false>(c, dist);
        const float ws = c.mass * fw;
        const float tsx = dx * ws, tsy = dy * ws, tsz = dz * ws;
        accS[0] = accS[0] + tsx;
        accS[1] = accS[1] + tsy;
        accS[2] = accS[2] + tsz;
        const float wx = (c.mass / rho[j]) * fw;
        const float txx = (v.x[j] - vxi) * wx, txy = (v.y[j] - vyi) * wx, txz = (v.z[j] - vzi) * wx;
        accX[0] = accX[0] + txx;
        accX[1] = accX[1] + txy;
        accX[2] = accX[2] + txz;
      }
      if constexpr (WANT_G) {
        float nx = 0.f, ny = 0.f, nz = 0.f;  // vector.go:322-331 Norm
        if (dist != 0.0f) {
          nx = dx / dist;
          ny = dy / dist;
          nz = dz / dist;
        }
        const float s = -kern_O1D<false>(c, dist);  // std_kernel.go:74-76 Grad
        const float gx = nx * s, gy = ny * s, gz = nz * s;
        const float F = pti + pterm[j];
        const float tx = gx * F, ty = gy * F, tz = gz * F;
        accG[0] = accG[0] + tx;
        accG[1] = accG[1] + ty;
        accG[2] = accG[2] + tz;
      }
      if constexpr (WANT_V) {
        const float inv = 1.0f / rho[j];
        const float ux = (v.x[j] - vxi) * inv, uy = (v.y[j] - vyi) * inv, uz = (v.z[j] - vzi) * inv;
        const float o2 = kern_O2D<false>(c, dist);
        const float tx = ux * o2, ty = uy * o2, tz = uz * o2;
        if (c.visc_running_mass) {  // sph_field.go:265: (force + t) * m
          const float sx = accV[0] + tx, sy = accV[1] + ty, sz = accV[2] + tz;
          accV[0] = sx * c.mass;
          accV[1] = sy * c.mass;
          accV[2] = sz * c.mass;
        } else {
          const float mx = tx * c.mass, my = ty * c.mass, mz = tz * c.mass;
          accV[0] = accV[0] + mx;
          accV[1] = accV[1] + my;
          accV[2] = accV[2] + mz;
        }
      }
    }
  });
}

// G: SPH.GradientPressureForce (fluid.go:164-172): F_i += sign * rho_i*m * accG
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_gradient(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p,
                                                     const float* __restrict__ rho, const float* __restrict__ pterm,
                                                     Soa3 f, const DevStats* stats, int honour_done) {
  if (honour_done && stats->pci_done) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c) || bnd.is(i)) return;  // GradientPressureForce loops over N() particles (fluid.go:164-172)
  float accG[3] = {0.f, 0.f, 0.f}, accV[3] = {0.f, 0.f, 0.f};
  CSoa3 nov{nullptr, nullptr, nullptr};
  force_sweep<FAST, true, false>(c, nb, i, p, nov, rho, pterm, accG, accV);
  const float dm = rho[i] * c.mass;
  const float gx = accG[0] * dm, gy = accG[1] * dm, gz = accG[2] * dm;
  const float sx = gx * c.pressure_sign, sy = gy * c.pressure_sign, sz = gz * c.pressure_sign;
  f.x[i] += sx;
  f.y[i] += sy;
  f.z[i] += sz;
}

// V: SPH.ViscousAll (fluid.go:146-152): F_i += mu * LaplacianForce(i)
// with_xs (PCISPH with the build-defined terms): the same sweep also adds the cohesion force and
// stores the XSPH correction for Update
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_viscous(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p, CSoa3 v,
                                                    const float* __restrict__ rho, Soa3 f, int with_xs, Soa3 xsph) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c) || bnd.is(i)) return;  // ViscousAll loops over N() particles (fluid.go:146-152)
  float accG[3] = {0.f, 0.f, 0.f}, accV[3] = {0.f, 0.f, 0.f}, accX[3] = {0.f, 0.f, 0.f}, accS[3] = {0.f, 0.f, 0.f};
  force_sweep<FAST, false, true>(c, nb, i, p, v, rho, nullptr, accG, accV, with_xs ? accX : nullptr,
                                 with_xs ? accS : nullptr);
  const float tx = accV[0] * c.mu, ty = accV[1] * c.mu, tz = accV[2] * c.mu;
  f.x[i] += tx;
  f.y[i] += ty;
  f.z[i] += tz;
  if (with_xs) {
    const float sx = accS[0] * c.st_kappa, sy = accS[1] * c.st_kappa, sz = accS[2] * c.st_kappa;
    f.x[i] += sx;
    f.y[i] += sy;
    f.z[i] += sz;
    xsph.x[i] = accX[0] * c.xsph_eps;
    xsph.y[i] = accX[1] * c.xsph_eps;
    xsph.z[i] = accX[2] * c.xsph_eps;
  }
}

// X: SPH.ExternalAll (fluid.go:155-161)
__global__ __launch_bounds__(kBlock) void k_external(int n, Soa3 f, float ex, float ey, float ez) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  f.x[i] += ex;
  f.y[i] += ey;
  f.z[i] += ez;
}

// U: SPH.Update (fluid.go:175-197) for one particle, then the build-defined wall box.
__device__ __forceinline__ void integrate_core(const DevConsts& c, float fx, float fy, float fz, float& px, float& py,
                                               float& pz, float& vx, float& vy, float& vz, unsigned int& vbits,
                                               unsigned int& fbits, float xsx = 0.f, float xsy = 0.f,
                                               float xsz = 0.f) {
  const float ax = fx * c.inv_mass, ay = fy * c.inv_mass, az = fz * c.inv_mass;
  const float dvx = ax * c.dt, dvy = ay * c.dt, dvz = az * c.dt;
  vx += dvx;
  vy += dvy;
  vz += dvz;
  float ux = vx, uy = vy, uz = vz;
  if (c.xsph_eps != 0.0f) {  // build-defined XSPH: positions advect with the smoothed velocity
    ux = vx + xsx;
    uy = vy + xsy;
    uz = vz + xsz;
  }
  const float dpx = ux * c.dt, dpy = uy * c.dt, dpz = uz * c.dt;
  px += dpx;
  py += dpy;
  pz += dpz;
  // maxVel / maxF (fluid.go:186-191) are tracked as SQUARED magnitudes: the square root is monotone
  // and correctly rounded, so taking it once, of the maximum, on the host gives the same float --
  // and saves two IEEE square roots (~40 instructions) per particle and step here
  const float vm = dist2<false>(vx, vy, vz);
  const float fm = dist2<false>(fx, fy, fz);
  const unsigned int vb = nonneg_bits(vm), fb = nonneg_bits(fm);
  vbits = vb > vbits ? vb : vbits;
  fbits = fb > fbits ? fb : fbits;
  if (c.walls) {
    float* P[3] = {&px, &py, &pz};
    float* V[3] = {&vx, &vy, &vz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (*P[a] < c.bmin[a]) {
        *P[a] = c.bmin[a];
        if (*V[a] < 0.0f) *V[a] = -*V[a] * c.rest;
      }
      if (*P[a] > c.bmax[a]) {
        *P[a] = c.bmax[a];
        if (*V[a] > 0.0f) *V[a] = -*V[a] * c.rest;
      }
    }
  }
}

__device__ __forceinline__ void integrate_one(const DevConsts& c, float fx, float fy, float fz, float& px, float& py,
                                              float& pz, float& vx, float& vy, float& vz, DevStats* stats,
                                              float xsx = 0.f, float xsy = 0.f, float xsz = 0.f) {
  unsigned int vb = 0u, fb = 0u;
  integrate_core(c, fx, fy, fz, px, py, pz, vx, vy, vz, vb, fb, xsx, xsy, xsz);
  wave_atomic_max(&stats->max_vel_bits, vb);
  wave_atomic_max(&stats->max_f_bits, fb);
}

// stand-alone Update: in place (per-particle, no neighbour reads)
__global__ __launch_bounds__(kBlock) void k_update(DevConsts c, Bnd bnd, Soa3 p, Soa3 v, CSoa3 f, int forces_uniform,
                                                   DevStats* stats, CSoa3 xsph) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  float fx = c.reset[0], fy = c.reset[1], fz = c.reset[2];
  float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  const bool live = i < live_n(c) && !bnd.is(i);  // Update loops over N() particles (fluid.go:175-197)
  if (live) {
    if (!forces_uniform) {
      fx = f.x[i];
      fy = f.y[i];
      fz = f.z[i];
    }
    px = p.x[i];
    py = p.y[i];
    pz = p.z[i];
    vx = v.x[i];
    vy = v.y[i];
    vz = v.z[i];
  } else {
    fx = fy = fz = 0.f;
  }
  const bool owned = live && slab_owned(c, px, py, pz);
  if (!owned) fx = fy = fz = vx = vy = vz = 0.f;  // keep ghosts out of the max|v|, max|F| counters
  float xsx = 0.f, xsy = 0.f, xsz = 0.f;
  if (owned && c.xsph_eps != 0.0f && xsph.x != nullptr) {
    xsx = xsph.x[i];
    xsy = xsph.y[i];
    xsz = xsph.z[i];
  }
  integrate_one(c, fx, fy, fz, px, py, pz, vx, vy, vz, stats, xsx, xsy, xsz);
  if (owned) {
    p.x[i] = px;
    p.y[i] = py;
    p.z[i] = pz;
    v.x[i] = vx;
    v.y[i] = vy;
    v.z[i] = vz;
  } else if (live) {
    const float qnan = __uint_as_float(0x7fc00000u);  // ghost: dropped at the next neighbour build
    p.x[i] = qnan;
    p.y[i] = qnan;
    p.z[i] = qnan;
  }
}

// ---------------------------------------------------------------------------------
// Fused WCSPH force + integrate: [G] [V] X (PR) U of one step in one neighbour sweep.
// Reads the sorted state (pin, vin, rho, pterm, optional forces), writes new positions
// and velocities to the other half of the ping-pong pair.  PressureAll's result is
// overwritten by Update (Press = 0, fluid.go:192) so it is not materialised.
// ---------------------------------------------------------------------------------
template <bool FAST, bool WANT_G, bool WANT_V>
__global__ __launch_bounds__(kBlock) void k_force_integrate(DevConsts c, Neigh nb, Bnd bnd, CSoa3 pin,
                                                            CSoa3 vin, const float* __restrict__ rho,
                                                            const float* __restrict__ pterm, CSoa3 fin,
                                                            int forces_uniform, Soa3 pout, Soa3 vout,
                                                            DevStats* stats) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool live = i < live_n(c);
  const bool boundary = live && bnd.is(i);  // carried over unchanged into the other half of the ping-pong pair
  const bool owned = live && !boundary && slab_owned(c, pin.x[i], pin.y[i], pin.z[i]);
  float fx = 0.f, fy = 0.f, fz = 0.f, px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
  float accX[3] = {0.f, 0.f, 0.f}, accS[3] = {0.f, 0.f, 0.f};
  if (owned) {
    fx = c.reset[0];
    fy = c.reset[1];
    fz = c.reset[2];
    if (!forces_uniform) {
      fx = fin.x[i];
      fy = fin.y[i];
      fz = fin.z[i];
    }
    float accG[3] = {0.f, 0.f, 0.f}, accV[3] = {0.f, 0.f, 0.f};
    const bool xs = c.xsph_eps != 0.0f || c.st_kappa != 0.0f;
    if (WANT_G || WANT_V || xs)
      force_sweep<FAST, WANT_G, WANT_V>(c, nb, i, pin, vin, rho, pterm, accG, accV, xs ? accX : nullptr,
                                        xs ? accS : nullptr);
    if constexpr (WANT_G) {
      const float dm = rho[i] * c.mass;
      const float gx = accG[0] * dm, gy = accG[1] * dm, gz = accG[2] * dm;
      const float sx = gx * c.pressure_sign, sy = gy * c.pressure_sign, sz = gz * c.pressure_sign;
      fx += sx;
      fy += sy;
      fz += sz;
    }
    if constexpr (WANT_V) {
      const float tx = accV[0] * c.mu, ty = accV[1] * c.mu, tz = accV[2] * c.mu;
      fx += tx;
      fy += ty;
      fz += tz;
    }
    if (xs) {
      const float sx = accS[0] * c.st_kappa, sy = accS[1] * c.st_kappa, sz = accS[2] * c.st_kappa;
      fx += sx;
      fy += sy;
      fz += sz;
    }
    fx += c.ext[0];
    fy += c.ext[1];
    fz += c.ext[2];
    px = pin.x[i];
    py = pin.y[i];
    pz = pin.z[i];
    vx = vin.x[i];
    vy = vin.y[i];
    vz = vin.z[i];
  }
  integrate_one(c, fx, fy, fz, px, py, pz, vx, vy, vz, stats, accX[0] * c.xsph_eps, accX[1] * c.xsph_eps,
                accX[2] * c.xsph_eps);
  if (owned) {
    pout.x[i] = px;
    pout.y[i] = py;
    pout.z[i] = pz;
    vout.x[i] = vx;
    vout.y[i] = vy;
    vout.z[i] = vz;
  } else if (live) {
    const float qnan = __uint_as_float(0x7fc00000u);  // ghost: dropped at the next neighbour build
    pout.x[i] = boundary ? pin.x[i] : qnan;
    pout.y[i] = boundary ? pin.y[i] : qnan;
    pout.z[i] = boundary ? pin.z[i] : qnan;
    vout.x[i] = vin.x[i];
    vout.y[i] = vin.y[i];
    vout.z[i] = vin.z[i];
  }
}

// ---------------------------------------------------------------------------------
// Field operators that no solver calls (model/field/sph_field.go:124-135,203-294):
// Div, Curl, Laplacian, Interpolate.  Scalar fields: density, or PressureField.Value =
// TaitEos(rho, 87.0, 0) (field_types.go:39-42); tensor fields: velocity or force.
// ---------------------------------------------------------------------------------
enum { kOpDiv = 0, kOpCurl = 1 };

// Div (sph_field.go:203-227) / Curl (:272-294) of a tensor field t
template <bool FAST, int OP>
__global__ __launch_bounds__(kBlock) void k_field_div_curl(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p,
                                                           CSoa3 t, const float* __restrict__ rho, Soa3 out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c) || bnd.is(i)) return;
  const float xi = p.x[i], yi = p.y[i], zi = p.z[i];
  float div = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
    if (j == i) return;
    const float dx = p.x[j] - xi, dy = p.y[j] - yi, dz = p.z[j] - zi;
    const float dist = dsl_sqrt<FAST>(dist2<FAST>(dx, dy, dz));
    if (!in_support(c, nb, dist)) return;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (dist != 0.0f) {
      nx = dsl_div<FAST>(dx, dist);
      ny = dsl_div<FAST>(dy, dist);
      nz = dsl_div<FAST>(dz, dist);
    }
    const float s = -kern_O1D<FAST>(c, dist);
    const float gx = nx * s, gy = ny * s, gz = nz * s;
    const float w = dsl_div<FAST>(c.mass, rho[j]);
    const float sx = t.x[j] * w, sy = t.y[j] * w, sz = t.z[j] * w;
    if constexpr (OP == kOpDiv) {
      const float t0 = sx * gx, t1 = sy * gy, t2 = sz * gz;  // vector.go:268-276 Dot
      const float d = (t0 + t1) + t2;
      div += d;
    } else {
      const float a0 = sy * gz, b0 = sz * gy;  // vector.go:283-297 Cross
      const float a1 = sz * gx, b1 = gz * sx;
      const float a2 = sx * gy, b2 = gx * sy;
      const float c0 = a0 - b0, c1 = a1 - b1, c2 = a2 - b2;
      cx = cx + c0;
      cy = cy + c1;
      cz = cz + c2;
    }
  });
  if constexpr (OP == kOpDiv) out.x[i] = div;
  else {
    out.x[i] = cx;
    out.y[i] = cy;
    out.z[i] = cz;
  }
}

template <bool FAST>
__device__ __forceinline__ float scalar_field(const DevConsts& c, int field, const float* __restrict__ rho, int j) {
  const float d = rho[j];
  return field == 1 ? tait_eos<FAST>(c, d, c.eos_d0_grad) : d;
}

// Laplacian (sph_field.go:230-248): sum_j m ((f_j - f_i)/rho_j) O2D(r)
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_field_laplacian(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p,
                                                            const float* __restrict__ rho, int field,
                                                            float* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c) || bnd.is(i)) return;
  const float xi = p.x[i], yi = p.y[i], zi = p.z[i];
  const float fi = scalar_field<FAST>(c, field, rho, i);
  float sum = 0.f;
  for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
    if (j == i) return;
    const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
    const float dist = dsl_sqrt<FAST>(dist2<FAST>(dx, dy, dz));
    if (!in_support(c, nb, dist)) return;
    const float df = scalar_field<FAST>(c, field, rho, j) - fi;
    const float tt = c.mass * dsl_div<FAST>(df, rho[j]);
    const float u = tt * kern_O2D<FAST>(c, dist);
    sum += u;
  });
  out[i] = sum;
}

// Interpolate (sph_field.go:124-135) at arbitrary positions q: sum_j (m/rho_j) F(r) f_j
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_field_interpolate(DevConsts c, Neigh nb, CSoa3 p,
                                                              const float* __restrict__ rho, int field, int nq,
                                                              const float* __restrict__ q, float* __restrict__ out) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= nq) return;
  const float xi = q[3 * k], yi = q[3 * k + 1], zi = q[3 * k + 2];
  float sum = 0.f;
  for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
    const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
    const float dist = dsl_sqrt<FAST>(dist2<FAST>(dx, dy, dz));
    if (!in_support(c, nb, dist)) return;
    const float weight = dsl_div<FAST>(c.mass, rho[j]) * kern_F<FAST>(c, dist);
    const float u = weight * scalar_field<FAST>(c, field, rho, j);
    sum += u;
  });
  out[k] = sum;
}

// ---------------------------------------------------------------------------------
// PCISPH (solver/pcisph/pcisph_darwin.go:52-99)
// ---------------------------------------------------------------------------------

// predict :57-73 -- _vel += (F/m) dt ; _pos += _vel dt (state persists across steps)
__device__ __forceinline__ bool pci_left_tile(const DevConsts& c, float x, float y, float z, float qx, float qy, float qz);
__device__ __forceinline__ void pci_drift_add(unsigned int* __restrict__ drift, unsigned int n_out, unsigned int n_all);
// (`drift`: how many predicted positions have left their particle's tile, see k_pci_predict_bin; nullptr in LSH mode)
__global__ __launch_bounds__(kBlock) void k_pci_predict(DevConsts c, Bnd bnd, CSoa3 p, CSoa3 f, Soa3 pp, Soa3 pv,
                                                        unsigned int* __restrict__ drift, DevStats* stats) {
  if (stats->pci_done) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  bool mine = false, left = false;
  if (i < live_n(c) && !bnd.is(i)) {  // the predictor loops over N() particles (pcisph_darwin.go:57-73)
    const float ax = f.x[i] * c.inv_mass, ay = f.y[i] * c.inv_mass, az = f.z[i] * c.inv_mass;
    const float dvx = ax * c.dt, dvy = ay * c.dt, dvz = az * c.dt;
    const float tvx = pv.x[i] + dvx, tvy = pv.y[i] + dvy, tvz = pv.z[i] + dvz;
    const float dpx = tvx * c.dt, dpy = tvy * c.dt, dpz = tvz * c.dt;
    const float qx = pp.x[i] + dpx, qy = pp.y[i] + dpy, qz = pp.z[i] + dpz;
    pp.x[i] = qx;
    pp.y[i] = qy;
    pp.z[i] = qz;
    pv.x[i] = tvx;
    pv.y[i] = tvy;
    pv.z[i] = tvz;
    mine = true;
    if (drift != nullptr) left = pci_left_tile(c, p.x[i], p.y[i], p.z[i], qx, qy, qz);
    if (c.slab_axis >= 0 && slab_owned(c, p.x[i], p.y[i], p.z[i]) && pci_query_escaped(c, qx, qy, qz)) stats->pci_escaped = 1;
  }
  if (drift != nullptr) pci_drift_add(drift, left ? 1u : 0u, mine ? 1u : 0u);
}

// DF + pressure accumulate :76-92 -- SPHField.DensityF (sph_field.go:137-152): starts at
// W0, includes self, neighbours' CURRENT positions around the PREDICTED position.
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_pci_density(DevConsts c, Neigh nb, Bnd bnd, CSoa3 p,
                                                        CSoa3 pp, float* __restrict__ press, DevStats* stats) {
  if (stats->pci_done) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned int err_bits = 0u;
  if (i < live_n(c) && !bnd.is(i)) {  // (pcisph_darwin.go:76-92 loops over N() particles)
    const float xi = pp.x[i], yi = pp.y[i], zi = pp.z[i];
    float density = c.W0;
    for_each_candidate(c, nb, xi, yi, zi, [&](int j) {
      const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
      const float r2 = dist2<FAST>(dx, dy, dz);
      if constexpr (FAST) {
        if (r2 < c.hh) {
          const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
          density = __builtin_fmaf(c.mass * c.A, q * q, density);
        }
      } else {
        const float dist = dsl_sqrt<false>(r2);
        if (in_support(c, nb, dist)) {
          const float w = kern_F<false>(c, dist);
          density += c.mass * w;
        }
      }
    });
    const float density_error = density - c.ref_density;
    const float abs_err = dsl_div<FAST>(density_error, c.ref_density);
    const float dp = density_error * c.delta;
    press[i] += dp;
    // slab mode: a ghost's predicted density is meaningless (it sees half a neighbourhood) and must
    // not decide the iteration's error
    if (slab_owned(c, p.x[i], p.y[i], p.z[i])) err_bits = nonneg_bits(abs_err);
  }
  wave_atomic_max(&stats->pci_cur_err_bits, err_bits);
}

// ---------------------------------------------------------------------------------
// DensityF with the QUERY points binned (round 3).  The reference never brings its predictor state back to the
// particles (pcisph_darwin.go:28-41: `_pos`, `_vel` are seeded once and advanced four times per step), so DensityF's
// query points -- the predicted positions -- leave their particles behind: by step 50 of the 4M scene the median
// distance is 0.27 h, by step 200 2.8 h, by step 600 15 h.  A sweep launched in the PARTICLES' slot order then has 64
// lanes looking at 64 unrelated places (12.8 ms per iteration at step 600 where the first steps take 0.22).  Here the
// queries get a counting sort of their own, by the cell of the query point in the particles' grid, every correction
// iteration: predict + cell + rank (k_pci_predict_bin), the prefix scan of the build (k_scan_sums / k_scan_apply on
// the query histogram), one record per query in cell order (k_pci_query_scatter), and the sweep in THAT order
// (k_pci_density_binned): neighbouring lanes read the same candidates again.  Per query the candidates, their order and
// the arithmetic are those of k_pci_density, so DSL_MATH_EXACT stays bit for bit the oracle's; the order of the queries
// inside a cell comes from atomics and does not matter (a query's sum does not depend on the others).
// ---------------------------------------------------------------------------------
// `drift` is 512 counters: [w & 255] += queries of wave w whose tile is not their particle's, [256 + (w & 255)] += queries
// of wave w; the host looks at them every few steps and switches the solver to the binned form for good.
__device__ __forceinline__ bool pci_left_tile(const DevConsts& c, float x, float y, float z, float qx, float qy, float qz) {
  bool out = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float pa = a == 0 ? x : (a == 1 ? y : z), qa = a == 0 ? qx : (a == 1 ? qy : qz);
    out |= (cell_coord(pa, c.gmin[a], c.inv_cell, c.dims[a]) / kTB) != (cell_coord(qa, c.gmin[a], c.inv_cell, c.dims[a]) / kTB);
  }
  return out;
}
__device__ __forceinline__ void pci_drift_add(unsigned int* __restrict__ drift, unsigned int n_out, unsigned int n_all) {
  for (int off = kWave / 2; off > 0; off >>= 1) {
    n_out += __shfl_xor(n_out, off, kWave);
    n_all += __shfl_xor(n_all, off, kWave);
  }
  if ((threadIdx.x & (kWave - 1)) == 0 && n_all != 0u) {
    const int slot = (blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave)) & 255;
    if (n_out != 0u) atomicAdd(&drift[slot], n_out);
    atomicAdd(&drift[256 + slot], n_all);
  }
}

// predict (:57-73) for every particle the un-binned path would predict -- GHOSTS: every live fluid slot (the passes of
// the per-pass path), otherwise owned slots only (k_pci_density_tiled) -- then, ADD_G: F += the cached gradient term
// (:93, k_pci_density_tiled's tail), and the query's cell and its rank inside that cell.  Equal-cell runs of consecutive
// lanes share one atomic (k_cell_rank's ballot trick: early in a run every query still sits in its particle's cell).
// A query further than h outside the grid's bounds has no neighbour -- provided every particle lies inside them, which the
// build checks (k_cell_rank: `off_grid`; the walls keep particles in the box, but nothing obliges a host to put the
// grid around the box).  Such a query is finished here (density = W0, :76-92 on that value) instead of being clamped
// into the grid's outermost cells with thousands of others: the predictor knows no walls, and by step 1500 of the 4M
// scene four queries out of five are below the floor.
// ROWS (kernels_tiled.hpp: tile_setup_load_counts): the query's record goes straight into slot `rank` of its cell's row
// of `row_slots` records (qrows), or, when the row is full, behind the spill list (spill, n_spill), which a
// global-memory sweep finishes; qrank is then unused and n_qtiles is cleared by the host (the spill counter sits
// next to it and is written by this very launch).
// ROWS, INCR (the second and later iterations of a step): the rows are KEPT from the previous iteration.  A query point
// moves a few thousandths of h per iteration, so nearly every query is still in the cell it was in: it overwrites its own
// record (qslot[i], the record's index, written by the step's first iteration) -- no atomic, where the full form pays ~4M
// of them with a return value per iteration, 0.16 ms at 4M whatever their scope (tools/atomic_scope.hip).  A query
// that has changed cells leaves a tombstone (particle -1: the sweep computes it and drops the result) and takes the next
// free slot of its new cell's row; the counts only ever grow inside a step, and the step's first iteration starts from
// empty rows again (the sort has re-numbered the particles by then anyway).
template <bool GHOSTS, bool ADD_G, bool FAST, bool ROWS = false, bool INCR = false>
__global__ __launch_bounds__(kBlock) void k_pci_predict_bin(DevConsts c, Bnd bnd, CSoa3 p, Soa3 pp, Soa3 pv, CSoa3 gterm,
                                                            Soa3 frc, int* __restrict__ qcount, int* __restrict__ qrank,
                                                            int* __restrict__ n_qtiles, unsigned int* __restrict__ drift,
                                                            const int* __restrict__ off_grid, int build_seq,
                                                            float* __restrict__ press, DevStats* stats,
                                                            float4* __restrict__ qrows = nullptr,
                                                            int row_slots = 0, float4* __restrict__ spill = nullptr,
                                                            int* __restrict__ n_spill = nullptr,
                                                            int* __restrict__ qslot = nullptr) {
  static_assert(!INCR || ROWS, "only the query rows persist between iterations");
  if (stats->pci_done) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (!ROWS && i == 0 && n_qtiles != nullptr) *n_qtiles = 0;  // the length of this iteration's query-tile list (k_qtile_list)
  const int lane = threadIdx.x & (kWave - 1);
  const bool all_inside = *off_grid != build_seq;  // (the current build's number: k_cell_rank)
  int cell = -1;
  bool left = false, mine = false;
  unsigned int err_bits = 0u;
  float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the particle's own position is only read where something asks for it: slab ownership, the drift statistic)
  // Everything the kernel reads per particle is requested BEFORE its first store: pp, pv and frc are written below, and
  // a load behind those stores (the position for the drift statistic, the query's slot of the previous iteration) has
  // to wait for them -- one more round trip each, in a kernel that is nothing but round trips.
  const bool want_pos = drift != nullptr || c.slab_axis >= 0;
  float opx = 0.f, opy = 0.f, opz = 0.f;
  int old_slot = -1;
  if (i < live_n(c)) {
    if (want_pos) {
      opx = p.x[i];
      opy = p.y[i];
      opz = p.z[i];
    }
    if constexpr (INCR) old_slot = qslot[i];
  }
  if (i < live_n(c) && !bnd.is(i) && (GHOSTS || c.slab_axis < 0 || slab_owned(c, opx, opy, opz))) {
    const float fx = frc.x[i], fy = frc.y[i], fz = frc.z[i];
    const float ax = fx * c.inv_mass, ay = fy * c.inv_mass, az = fz * c.inv_mass;
    const float dvx = ax * c.dt, dvy = ay * c.dt, dvz = az * c.dt;
    const float tvx = pv.x[i] + dvx, tvy = pv.y[i] + dvy, tvz = pv.z[i] + dvz;
    const float dpx = tvx * c.dt, dpy = tvy * c.dt, dpz = tvz * c.dt;
    const float qx = pp.x[i] + dpx, qy = pp.y[i] + dpy, qz = pp.z[i] + dpz;
    pp.x[i] = qx;
    pp.y[i] = qy;
    pp.z[i] = qz;
    pv.x[i] = tvx;
    pv.y[i] = tvy;
    pv.z[i] = tvz;
    if constexpr (ADD_G) {
      frc.x[i] = fx + gterm.x[i];
      frc.y[i] = fy + gterm.y[i];
      frc.z[i] = fz + gterm.z[i];
    }
    mine = true;
    if (drift != nullptr) left = pci_left_tile(c, opx, opy, opz, qx, qy, qz);
    if (c.slab_axis >= 0 && (!GHOSTS || slab_owned(c, opx, opy, opz)) && pci_query_escaped(c, qx, qy, qz))
      stats->pci_escaped = 1;
    bool far = false;
    if (all_inside) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float f = ((a == 0 ? qx : (a == 1 ? qy : qz)) - c.gmin[a]) * c.inv_cell;  // in cells (a cell is h wide)
        far |= f < -1.01f || f > (float)c.dims[a] + 1.01f;
      }
    }
    if (far) {
      const float density_error = c.W0 - c.ref_density;
      const float abs_err = dsl_div<FAST>(density_error, c.ref_density);
      press[i] += density_error * c.delta;
      if (!GHOSTS || c.slab_axis < 0 || slab_owned(c, opx, opy, opz)) err_bits = nonneg_bits(abs_err);
    } else {
      cell = cell_of(c, qx, qy, qz);
      rec = make_float4(qx, qy, qz, __int_as_float(i));
    }
  }
  // Equal-cell runs of consecutive lanes share one atomic (early on the queries of a cell's particles are still
  // together).  Later they are not: 64 consecutive particles' queries sit in 62 different cells by step 400 of the 4M
  // scene (tools/pci_query_spread.py) -- the drift is noisy, not a smooth displacement -- and grouping equal cells across
  // the whole wave (tried) then finds nothing to group.  Those ~4M device-scope atomics with a return value on random
  // words of a 16 MB histogram are ~120 us of this kernel, about what that many random 64-byte accesses cost.
  if constexpr (INCR) {
    if (i < live_n(c)) {
      const int old = old_slot;
      if (old >= 0 && old / row_slots == cell) {
        qrows[old] = rec;
      } else {
        if (old >= 0) reinterpret_cast<int*>(qrows)[(size_t)old * 4 + 3] = -1;
        int now = -1;
        if (cell >= 0) {
          const int r = atomicAdd(&qcount[cell], 1);
          if (r < row_slots) {
            now = cell * row_slots + r;
            qrows[now] = rec;
          } else {
            spill[atomicAdd(n_spill, 1)] = rec;
          }
        }
        if (now != old) qslot[i] = now;
      }
    }
    if (drift != nullptr) pci_drift_add(drift, left ? 1u : 0u, mine ? 1u : 0u);
    wave_atomic_max(&stats->pci_cur_err_bits, err_bits);
    return;
  }
  const int prev = __shfl_up(cell, 1, kWave);
  const bool head = (lane == 0) || (cell != prev);
  const unsigned long long heads = __ballot(head);
  const unsigned long long le = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int head_lane = 63 - __builtin_clzll(le);
  const unsigned long long above = (head_lane == 63) ? 0ull : (heads & ~((2ull << head_lane) - 1ull));
  const int next = above ? __builtin_ctzll(above) : kWave;
  int base = 0;
  if (lane == head_lane && cell >= 0) base = atomicAdd(&qcount[cell], next - head_lane);
  base = __shfl(base, head_lane, kWave);
  if constexpr (ROWS) {
    if (cell >= 0) {
      const int r = base + (lane - head_lane);
      if (r < row_slots) qrows[(size_t)cell * row_slots + r] = rec;
      else spill[atomicAdd(n_spill, 1)] = rec;  // (rare: more queries in one cell than a row holds)
      if (qslot != nullptr) qslot[i] = r < row_slots ? cell * row_slots + r : -1;
    } else if (qslot != nullptr && i < live_n(c)) {
      qslot[i] = -1;
    }
  } else {
    if (i < live_n(c)) qrank[i] = cell >= 0 ? base + (lane - head_lane) : -1;
  }
  if (drift != nullptr) pci_drift_add(drift, left ? 1u : 0u, mine ? 1u : 0u);
  wave_atomic_max(&stats->pci_cur_err_bits, err_bits);
}

// one record per query, in the order of the queries' cells: the predicted position and the particle's slot
__global__ __launch_bounds__(kBlock) void k_pci_query_scatter(DevConsts c, CSoa3 pp, const int* __restrict__ qrank,
                                                              const int* __restrict__ qstart, float4* __restrict__ qrec,
                                                              const DevStats* stats) {
  if (stats->pci_done) return;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  const int r = qrank[i];
  if (r < 0) return;
  const float qx = pp.x[i], qy = pp.y[i], qz = pp.z[i];
  qrec[qstart[cell_of(c, qx, qy, qz)] + r] = make_float4(qx, qy, qz, __int_as_float(i));
}

// DF + pressure accumulate :76-92 for query record k (k_pci_density's body; the query comes from its record)
// (`count`: the number of records -- the query prefix's last entry, or the spill counter of the query rows; grid-stride)
template <bool FAST>
__global__ __launch_bounds__(kBlock) void k_pci_density_binned(DevConsts c, Neigh nb, CSoa3 p, const float4* __restrict__ qrec,
                                                               const int* __restrict__ count, float* __restrict__ press,
                                                               DevStats* stats) {
  if (stats->pci_done) return;
  unsigned int err_bits = 0u;
  const int n = *count;
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
    const float4 rec = qrec[k];
    const int i = __float_as_int(rec.w);
    const float xi = rec.x, yi = rec.y, zi = rec.z;
    float density = c.W0;
    for_each_grid_candidate(c, nb.cell_start, xi, yi, zi, [&](int j) {
      const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
      const float r2 = dist2<FAST>(dx, dy, dz);
      if constexpr (FAST) {
        if (r2 < c.hh) {
          const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
          density = __builtin_fmaf(c.mass * c.A, q * q, density);
        }
      } else {
        const float dist = dsl_sqrt<false>(r2);
        if (in_support(c, nb, dist)) {
          const float w = kern_F<false>(c, dist);
          density += c.mass * w;
        }
      }
    });
    const float density_error = density - c.ref_density;
    const float abs_err = dsl_div<FAST>(density_error, c.ref_density);
    const float dp = density_error * c.delta;
    press[i] += dp;
    if (c.slab_axis < 0 || slab_owned(c, p.x[i], p.y[i], p.z[i])) {
      const unsigned int eb = nonneg_bits(abs_err);
      err_bits = eb > err_bits ? eb : err_bits;
    }
  }
  wave_atomic_max(&stats->pci_cur_err_bits, err_bits);
}

// end of one correction iteration :95-98
// (`iter_counters`: the query-tile list length and the spill count of the binned iteration -- zero for the next one)
__global__ void k_pci_check(DevConsts c, DevStats* stats, int* __restrict__ iter_counters) {
  if (iter_counters != nullptr) iter_counters[0] = iter_counters[1] = 0;
  if (stats->pci_done) return;
  const unsigned int e = stats->pci_cur_err_bits;
  stats->pci_last_err_bits = e;
  stats->pci_iters += 1;
  stats->pci_cur_err_bits = 0u;
  if (__uint_as_float(e) <= c.pci_max_error) stats->pci_done = 1;
}
__global__ void k_pci_reset(DevStats* stats, int* __restrict__ iter_counters) {
  if (iter_counters != nullptr) iter_counters[0] = iter_counters[1] = 0;
  stats->pci_done = 0;
  stats->pci_iters = 0;
  stats->pci_cur_err_bits = 0u;
  stats->pci_last_err_bits = 0u;
}

}  // namespace dsl
