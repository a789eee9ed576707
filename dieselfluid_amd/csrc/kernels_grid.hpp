// kernels_grid.hpp -- uniform-grid neighbour table built by a GPU counting sort:
//   k_cell_rank  : cell hash + wave-aggregated histogram (segmented by __ballot)
//   k_scan_*     : exclusive prefix sum of the cell histogram, wavefront scan
//                  (__shfl_up) staged through LDS, three phases
//   k_scatter    : counting-sort scatter of the SoA particle arrays
// Replaces sampler/lsh (sampler/lsh/lsh.go:102-133) as the neighbour structure, as
// BASELINE.json's north_star asks; gfx950 only.
#pragma once

#include "sph_device.hpp"

namespace dsl {

constexpr int kBlock = 256;
constexpr int kScanTile = 4096;  // cells per scan block: 4 sub-tiles of 256 lanes x int4

// ---------------------------------------------------------------------------------
// cell hash + rank inside the cell.  Consecutive lanes usually hold particles of the
// same cell (the input is the previous step's sorted order), so equal-cell runs are
// found with one ballot and only the run's first lane issues the atomic.
// ---------------------------------------------------------------------------------
// Slab mode: stale ghosts carry NaN positions (written by the integrate kernels); they go
// to the extra bucket `ncell` and so sort behind every live particle.  A particle that has
// just crossed the slab plane stays finite: it serves as a ghost for one more step, because
// the neighbour packed its own band before receiving it.
__global__ __launch_bounds__(kBlock) void k_cell_rank(DevConsts c, const float* __restrict__ px,
                                                      const float* __restrict__ py,
                                                      const float* __restrict__ pz, int* __restrict__ cellid,
                                                      int* __restrict__ rank, int* __restrict__ cell_count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  int cell = -1;
  const int n = live_n(c);
  if (i < n) {
    const float x = px[i], y = py[i], z = pz[i];
    cell = cell_of(c, x, y, z);
    if (c.slab_axis >= 0) {
      const bool finite = (x == x) && (y == y) && (z == z);
      if (!finite) cell = c.ncell;
    }
  }
  const int prev = __shfl_up(cell, 1, kWave);
  const bool head = (lane == 0) || (cell != prev);
  const unsigned long long heads = __ballot(head);
  const unsigned long long le = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int head_lane = 63 - __builtin_clzll(le);  // le always has bit 0 set
  const unsigned long long above = (head_lane == 63) ? 0ull : (heads & ~((2ull << head_lane) - 1ull));
  const int next = above ? __builtin_ctzll(above) : kWave;
  const int run = next - head_lane;
  int base = 0;
  if (lane == head_lane && cell >= 0) base = atomicAdd(&cell_count[cell], run);
  base = __shfl(base, head_lane, kWave);
  if (i < n) {
    cellid[i] = cell;
    rank[i] = base + (lane - head_lane);
  }
}

// ---------------------------------------------------------------------------------
// prefix sum
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    int t = __shfl_up(v, o, kWave);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; lds holds 4 wave sums
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int& total) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x >> 6;
  const int inc = wave_inclusive_scan(v);
  if (lane == kWave - 1) lds[wid] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    const int s = lds[w];
    if (w < wid) off += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return off + inc - v;
}

// phase 1: one sum per 4096-cell tile
__global__ __launch_bounds__(kBlock) void k_scan_sums(const int* __restrict__ count, int* __restrict__ block_sums) {
  __shared__ int lds[kBlock / kWave];
  const int4* src = reinterpret_cast<const int4*>(count + (size_t)blockIdx.x * kScanTile);
  int s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int4 v = src[k * kBlock + threadIdx.x];
    s += (v.x + v.y) + (v.z + v.w);
  }
  int total;
  block_exclusive_scan(s, lds, total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// phase 2: exclusive scan of the tile sums in place (single block)
__global__ __launch_bounds__(kBlock) void k_scan_top(int* __restrict__ block_sums, int nb) {
  __shared__ int lds[kBlock / kWave];
  int carry = 0;
  for (int base = 0; base < nb; base += kBlock) {
    const int idx = base + threadIdx.x;
    const int v = idx < nb ? block_sums[idx] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, total);
    if (idx < nb) block_sums[idx] = carry + ex;
    carry += total;
  }
}

// phase 3: exclusive scan inside each tile + tile offset; also tracks the fullest cell
__global__ __launch_bounds__(kBlock) void k_scan_apply(const int* __restrict__ count,
                                                       const int* __restrict__ block_sums,
                                                       int* __restrict__ cell_start, DevStats* stats) {
  __shared__ int lds[kBlock / kWave];
  const int4* src = reinterpret_cast<const int4*>(count + (size_t)blockIdx.x * kScanTile);
  int4* dst = reinterpret_cast<int4*>(cell_start + (size_t)blockIdx.x * kScanTile);
  int carry = block_sums[blockIdx.x];
  int mx = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int4 v = src[k * kBlock + threadIdx.x];
    mx = max(max(mx, max(v.x, v.y)), max(v.z, v.w));
    const int s = (v.x + v.y) + (v.z + v.w);
    int total;
    const int ex = carry + block_exclusive_scan(s, lds, total);
    int4 o;
    o.x = ex;
    o.y = ex + v.x;
    o.z = o.y + v.y;
    o.w = o.z + v.z;
    dst[k * kBlock + threadIdx.x] = o;
    carry += total;
  }
  for (int off = kWave / 2; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0 && mx > 0) atomicMax(&stats->max_cell_count, mx);
}

// ---------------------------------------------------------------------------------
// counting-sort scatter of every live SoA array (positions, velocities, and when they
// are materialised the forces and the PCISPH predictor state) plus the slot->particle map
// ---------------------------------------------------------------------------------
constexpr int kMaxScatter = 16;
struct ScatterArrays {
  const float* src[kMaxScatter];
  float* dst[kMaxScatter];
  int nf;
  const int* ids_src;
  int* ids_dst;
};

__global__ __launch_bounds__(kBlock) void k_scatter(DevConsts c, ScatterArrays a, const int* __restrict__ cellid,
                                                    const int* __restrict__ rank,
                                                    const int* __restrict__ cell_start) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  const int d = cell_start[cellid[i]] + rank[i];
  a.ids_dst[d] = a.ids_src[i];
  for (int f = 0; f < a.nf; ++f) a.dst[f][d] = a.src[f][i];
}

// ---------------------------------------------------------------------------------
// host <-> device layout conversion: the reference keeps xyz interleaved on the host
// (model/particle_array.go:5-15); the device keeps SoA in cell-sorted order with
// ids[slot] = original particle index.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_unpack3(int n, const float* __restrict__ stage, const int* __restrict__ ids,
                                                    float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ z) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const int o = ids[s];
  x[s] = stage[3 * o];
  y[s] = stage[3 * o + 1];
  z[s] = stage[3 * o + 2];
}
__global__ __launch_bounds__(kBlock) void k_unpack1(int n, const float* __restrict__ stage, const int* __restrict__ ids,
                                                    float* __restrict__ x) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  x[s] = stage[ids[s]];
}
__global__ __launch_bounds__(kBlock) void k_pack3(int n, float* __restrict__ stage, const int* __restrict__ ids,
                                                  const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ z, int sorted_order) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const int o = sorted_order ? s : ids[s];
  stage[3 * o] = x[s];
  stage[3 * o + 1] = y[s];
  stage[3 * o + 2] = z[s];
}
__global__ __launch_bounds__(kBlock) void k_pack1(int n, float* __restrict__ stage, const int* __restrict__ ids,
                                                  const float* __restrict__ x, int sorted_order) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  stage[sorted_order ? s : ids[s]] = x[s];
}
// ---------------------------------------------------------------------------------
// slab halo: band selection (wave-aggregated append) and record append
// ---------------------------------------------------------------------------------
constexpr int kRecord = 7;  // x,y,z,vx,vy,vz,id-bits

// Message layout: record 0 is a header whose first word holds the record count (int bits);
// records 1..count follow.  Both bands are selected in one pass; counters[side] counts.
__global__ __launch_bounds__(kBlock) void k_slab_pack(DevConsts c, float bound_lo, float bound_hi, int want_lo,
                                                      int want_hi, const float* __restrict__ px,
                                                      const float* __restrict__ py, const float* __restrict__ pz,
                                                      const float* __restrict__ vx, const float* __restrict__ vy,
                                                      const float* __restrict__ vz, const int* __restrict__ ids,
                                                      float* __restrict__ out_lo, float* __restrict__ out_hi,
                                                      int capacity, int* __restrict__ counters) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  bool take[2] = {false, false};
  float x = 0.f, y = 0.f, z = 0.f;
  if (i < live_n(c)) {
    x = px[i];
    y = py[i];
    z = pz[i];
    const float p = c.slab_axis == 0 ? x : (c.slab_axis == 1 ? y : z);
    const bool finite = (x == x) && (y == y) && (z == z);  // ghosts carry NaN after the step
    take[0] = want_lo && finite && (p < bound_lo);
    take[1] = want_hi && finite && (p >= bound_hi);
  }
#pragma unroll
  for (int side = 0; side < 2; ++side) {
    const unsigned long long m = __ballot(take[side]);
    if (m == 0ull) continue;
    const int leader = __builtin_ctzll(m);
    int base = 0;
    if (lane == leader) base = atomicAdd(&counters[side], __builtin_popcountll(m));
    base = __shfl(base, leader, kWave);
    if (take[side]) {
      const int d = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (d < capacity) {
        float* r = (side == 0 ? out_lo : out_hi) + (size_t)(d + 1) * kRecord;
        r[0] = x;
        r[1] = y;
        r[2] = z;
        r[3] = vx[i];
        r[4] = vy[i];
        r[5] = vz[i];
        r[6] = __int_as_float(ids[i]);
      }
    }
  }
}
// header = min(count, capacity); an overflow is recorded for the host to find later
__global__ void k_slab_header(const int* __restrict__ counters, float* out_lo, float* out_hi, int capacity,
                              int* __restrict__ overflow) {
  for (int side = 0; side < 2; ++side) {
    float* o = side == 0 ? out_lo : out_hi;
    if (!o) continue;
    int n = counters[side];
    if (n > capacity) {
      atomicMax(overflow, n);
      n = capacity;
    }
    o[0] = __int_as_float(n);
  }
}

// appends the records of a message (count in its header) behind the current particles
__global__ __launch_bounds__(kBlock) void k_slab_append(const float* __restrict__ msg, int capacity,
                                                        const int* __restrict__ n_cur, int room,
                                                        float* __restrict__ px, float* __restrict__ py,
                                                        float* __restrict__ pz, float* __restrict__ vx,
                                                        float* __restrict__ vy, float* __restrict__ vz,
                                                        int* __restrict__ ids, int* __restrict__ overflow) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  int count = __float_as_int(msg[0]);
  count = count < 0 ? 0 : (count > capacity ? capacity : count);
  const int at = *n_cur;
  if (at + count > room) {  // does not fit: record it, append nothing
    if (k == 0) atomicMax(overflow, at + count);
    return;
  }
  if (k >= count) return;
  const float* r = msg + (size_t)(k + 1) * kRecord;
  const int d = at + k;
  px[d] = r[0];
  py[d] = r[1];
  pz[d] = r[2];
  vx[d] = r[3];
  vy[d] = r[4];
  vz[d] = r[5];
  ids[d] = __float_as_int(r[6]);
}
__global__ void k_slab_bump(const float* __restrict__ msg, int capacity, int* __restrict__ n_cur, int room) {
  int count = __float_as_int(msg[0]);
  count = count < 0 ? 0 : (count > capacity ? capacity : count);
  if (*n_cur + count <= room) *n_cur += count;
}
__global__ void k_set_count(int* __restrict__ n_cur, const int* __restrict__ src) { *n_cur = *src; }

__global__ __launch_bounds__(kBlock) void k_count_owned(DevConsts c, const float* __restrict__ px,
                                                        const float* __restrict__ py, const float* __restrict__ pz,
                                                        int* __restrict__ counter) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool own = i < live_n(c) && slab_owned(c, px[i], py[i], pz[i]);
  const unsigned long long m = __ballot(own);
  if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicAdd(counter, __builtin_popcountll(m));
}

// every stride-th particle (by host index) of a 3-component buffer: out[id/stride] = value(id)
__global__ __launch_bounds__(kBlock) void k_pack3_decimated(int n, int stride, float* __restrict__ stage,
                                                            const int* __restrict__ ids, const float* __restrict__ x,
                                                            const float* __restrict__ y, const float* __restrict__ z) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const int o = ids[s];
  if (o % stride != 0) return;
  const int k = o / stride;
  stage[3 * k] = x[s];
  stage[3 * k + 1] = y[s];
  stage[3 * k + 2] = z[s];
}

__global__ __launch_bounds__(kBlock) void k_iota(int n, int* __restrict__ ids) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s < n) ids[s] = s;
}
__global__ __launch_bounds__(kBlock) void k_fill3(int n, float* __restrict__ x, float* __restrict__ y,
                                                  float* __restrict__ z, float vx, float vy, float vz) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  x[s] = vx;
  y[s] = vy;
  z[s] = vz;
}

}  // namespace dsl
