// kernels_grid.hpp -- uniform-grid neighbour table built by a GPU counting sort:
//   k_cell_rank  : cell hash + wave-aggregated histogram (segmented by __ballot); rank inside the cell
//   k_scan_*     : exclusive prefix sum of the cell histogram, wavefront scan
//                  (__shfl_up) staged through LDS, three phases
//   k_scatter    : counting-sort scatter of the SoA particle arrays
// Replaces sampler/lsh (sampler/lsh/lsh.go:102-133) as the neighbour structure, as
// BASELINE.json's north_star asks; gfx950 only.
#pragma once

#include <climits>

#include "sph_device.hpp"

namespace dsl {

constexpr int kBlock = 256;
// in-cell ordering in ONE scatter pass (r03): k_cell_rank drops every particle's id into slot `rank` of its cell's own
// kCellKeys-entry key row (cell_keys[cell][rank]; ranks of a cell are 0 .. count-1, so the first `count` entries of a
// row are exactly this build's members), and k_scatter, for a cell that needs ordering, reads the row and counts the
// smaller ids.  The row array costs kCellKeys x 4 bytes per GRID CELL (2.1 GB for the 16M scene's 16.4M cells -- of
// 288 GB), touched only where particles are.  A cell with more members than a row holds falls back to the two-pass
// form (k_scatter_ordered), launched as a small flag-gated grid.
constexpr int kCellKeys = 32;
#ifndef DSL_RANK_UNROLL
#define DSL_RANK_UNROLL 2
#endif
constexpr int kRankUnroll = DSL_RANK_UNROLL;  // particles per lane and trip of k_cell_rank
constexpr int kScanTile = 4096;  // cells per scan block: 4 sub-tiles of 256 lanes x int4

// ---------------------------------------------------------------------------------
// cell hash + rank inside the cell.  Consecutive lanes usually hold particles of the
// same cell (the input is the previous step's sorted order), so equal-cell runs are
// found with one ballot and only the run's first lane issues the atomic.
// ---------------------------------------------------------------------------------
// Slab mode: stale ghosts carry NaN positions (written by the integrate kernels); they get the
// pseudo cell `ncell`, are not counted and are not scattered, so the sorted arrays hold the live
// particles only (cell_start[ncell] = their number).  A particle that has just crossed the
// slab plane stays finite: it serves as a ghost for one more step, because the neighbour
// packed its own band before receiving it.
// the cell of a particle for the sort: cell_of, or the pseudo cell `ncell` for a stale ghost
__device__ __forceinline__ int sort_cell(const DevConsts& c, float x, float y, float z) {
  if (c.slab_axis >= 0 && !((x == x) && (y == y) && (z == z))) return c.ncell;
  return cell_of(c, x, y, z);
}

// a skin build's reference coordinate (one fma, the same in every kernel that needs the particle's cell)
__device__ __forceinline__ float skin_ref(float x, float v, float tau) { return __builtin_fmaf(tau, v, x); }

// `unordered` is a bitmap over the cells: a set bit marks a cell whose slot order the scatter has to
// establish (see k_scatter): it received more than one run, or a run that was not ascending in
// particle id.  A lattice at rest has neither (a cell's particles are one run, in the order the
// previous step left them), so the ordering costs next to nothing there.
// (OFF_GRID is a template parameter: as a run-time test of the pointer the check cost the WCSPH build 0.022 of this
// kernel's 0.122 ms at 16M although it never ran)
template <bool OFF_GRID, bool REF = false>
__global__ __launch_bounds__(kBlock) void k_cell_rank(DevConsts c, const float* __restrict__ px,
                                                      const float* __restrict__ py,
                                                      const float* __restrict__ pz, const int* __restrict__ ids,
                                                      int* __restrict__ rank, int* __restrict__ cell_count,
                                                      unsigned int* __restrict__ unordered, DevStats* stats,
                                                      int* __restrict__ n_tiles, int* __restrict__ cell_keys,
                                                      int* __restrict__ overfull, int* __restrict__ off_grid,
                                                      int build_seq, SkinGate gate = SkinGate{nullptr},
                                                      const int* __restrict__ ids_alt = nullptr, CSoa3 vel = CSoa3{nullptr, nullptr, nullptr}) {
  if (gate.closed()) return;
  // (skin step: the sort is on the REFERENCE positions x + tau v -- sph_device.hpp, SkinState::tau)
  const float tau = REF ? gate.st->tau : 0.0f;
  // (skin step: `ids` is map 0, `ids_alt` map 1; the device state names the map that is current AFTER this rebuild,
  // so the one to read -- the order before the sort -- is the other)
  if (gate.st != nullptr && ids != nullptr && gate.st->ids_sel == 0) ids = ids_alt;
  // the small counters of the later kernels of this build (the fullest-cell statistic of the scan, the tile-list
  // lengths): cleared here, at the head of the build, when the one-launch scan is in use (stats != nullptr)
  if (stats != nullptr && blockIdx.x == 0) {
    if (threadIdx.x == 0) stats->max_cell_count = 0;
    if (n_tiles != nullptr && threadIdx.x < 8) n_tiles[threadIdx.x] = 0;
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int n = live_n(c);
  // (grid-stride: the skin step launches a capped grid -- on the steps that do not rebuild, all a launch does is find
  // its gate closed, and 62,500 workgroups doing that took 14 us; every other caller's grid covers n in one trip)
  // A workgroup takes kRankUnroll x kBlock consecutive particles per trip, a lane one of each sub-block: the loads of
  // all sub-blocks are in flight together, and so are their atomics -- the kernel is two dependent round trips
  // (position -> cell -> returning atomic -> rank) with next to no arithmetic, and one particle per lane left the
  // memory pipe idle between them (0.144 ms for 24 B x 16M particles: 2.7 TB/s).
  for (int first = blockIdx.x * (kRankUnroll * kBlock); first < n; first += gridDim.x * (kRankUnroll * kBlock)) {
  int cell[kRankUnroll], id[kRankUnroll], head_lane[kRankUnroll], base[kRankUnroll];
  bool mark[kRankUnroll];
  // (loads without a branch around them -- a lane past the end reads the last particle and forgets it -- so that the
  // compiler issues all of them before the first wait)
  float x[kRankUnroll], y[kRankUnroll], z[kRankUnroll];
#pragma unroll
  for (int u = 0; u < kRankUnroll; ++u) {
    const int i = min(first + u * kBlock + (int)threadIdx.x, n - 1);
    x[u] = px[i];
    y[u] = py[i];
    z[u] = pz[i];
    if constexpr (REF) {
      x[u] = skin_ref(x[u], vel.x[i], tau);
      y[u] = skin_ref(y[u], vel.y[i], tau);
      z[u] = skin_ref(z[u], vel.z[i], tau);
    }
    id[u] = ids ? ids[i] : 0;
  }
#pragma unroll
  for (int u = 0; u < kRankUnroll; ++u) {
    const bool in = first + u * kBlock + (int)threadIdx.x < n;
    cell[u] = in ? sort_cell(c, x[u], y[u], z[u]) : -1;
    if (!in) id[u] = 0;
    // PCISPH (OFF_GRID; the flag is never cleared: it holds the number of the last build that saw such a particle): does any particle lie outside the
    // grid's bounds, clamped into an outermost cell by the cell rule?  (false for NaN: that particle is nobody's neighbour)
    if constexpr (OFF_GRID) {
      bool off = false;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float f = floorf(((a == 0 ? x[u] : (a == 1 ? y[u] : z[u])) - c.gmin[a]) * c.inv_cell);
        off |= f < 0.0f || f >= (float)c.dims[a];
      }
      if (in && off) *off_grid = build_seq;  // (benign race: every writer stores the same number)
    }
  }
  unsigned long long run_mask[kRankUnroll];
#pragma unroll
  for (int u = 0; u < kRankUnroll; ++u) {
    const int prev = __shfl_up(cell[u], 1, kWave), prev_id = __shfl_up(id[u], 1, kWave);
    const bool head = (lane == 0) || (cell[u] != prev);
    const unsigned long long heads = __ballot(head);
    // lanes that break the ascending-id order of their run (ids == nullptr: nobody asks for an order)
    const unsigned long long breaks = __ballot(ids != nullptr && !head && id[u] <= prev_id);
    const unsigned long long le = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    head_lane[u] = 63 - __builtin_clzll(le);  // le always has bit 0 set
    const unsigned long long above = (head_lane[u] == 63) ? 0ull : (heads & ~((2ull << head_lane[u]) - 1ull));
    const int next = above ? __builtin_ctzll(above) : kWave;
    const int run = next - head_lane[u];
    base[u] = 0;
    const unsigned long long mine = (next == kWave ? ~0ull : ((1ull << next) - 1ull)) & ~((1ull << head_lane[u]) - 1ull);
    run_mask[u] = breaks & mine;
    // (a whole band of stale ghosts would otherwise hammer one counter: same-address atomics serialise)
    mark[u] = lane == head_lane[u] && cell[u] >= 0 && cell[u] != c.ncell;
    if (mark[u]) base[u] = atomicAdd(&cell_count[cell[u]], run);
  }
  // (the atomics of all sub-blocks are in flight; their results are looked at only now)
#pragma unroll
  for (int u = 0; u < kRankUnroll; ++u)
    if (mark[u] && ids != nullptr && (base[u] != 0 || run_mask[u] != 0ull)) atomicOr(&unordered[cell[u] >> 5], 1u << (cell[u] & 31));
#pragma unroll
  for (int u = 0; u < kRankUnroll; ++u) {
    const int i = first + u * kBlock + threadIdx.x;
    const int r = __shfl(base[u], head_lane[u], kWave) + (lane - head_lane[u]);
    if (i < n) rank[i] = r;  // (the scatter recomputes the cell: cheaper than 8 B of traffic)
    if (cell_keys != nullptr && i < n && cell[u] >= 0 && cell[u] != c.ncell) {
      if (r < kCellKeys) cell_keys[(size_t)cell[u] * kCellKeys + r] = id[u];
      else *overfull = 1;  // (rare: benign race, every writer stores 1)
    }
  }
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
// (block 0 also clears the small counters of the later kernels -- the fullest-cell statistic of phase 3, the tile-list
// lengths -- which used to be the job of a one-block middle phase; phase 3 now adds up its predecessors' sums itself)
__global__ __launch_bounds__(kBlock) void k_scan_sums(const int* __restrict__ count, int* __restrict__ block_sums,
                                                       DevStats* stats, int* __restrict__ n_tiles,
                                                       SkinGate gate = SkinGate{nullptr}) {
  __shared__ int lds[kBlock / kWave];
  if (gate.closed()) return;
  if (blockIdx.x == 0 && stats != nullptr) {  // (stats == nullptr: a scan that is not the particles' build -- PCISPH query bins)
    if (threadIdx.x == 0) stats->max_cell_count = 0;
    if (n_tiles && threadIdx.x < 8) n_tiles[threadIdx.x] = 0;
  }
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

// phase 2: exclusive scan inside each tile + tile offset; also tracks the fullest cell
// (the histogram is left zeroed for the next build: saves that build a memset launch)
__global__ __launch_bounds__(kBlock) void k_scan_apply(int* __restrict__ count,
                                                       const int* __restrict__ block_sums,
                                                       int* __restrict__ cell_start, DevStats* stats,
                                                       SkinGate gate = SkinGate{nullptr}) {
  __shared__ int carry_sums[kBlock / kWave];
  __shared__ int wave_sums[4][kBlock / kWave];
  if (gate.closed()) return;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
  int4* src = reinterpret_cast<int4*>(count + (size_t)blockIdx.x * kScanTile);
  int4* dst = reinterpret_cast<int4*>(cell_start + (size_t)blockIdx.x * kScanTile);
  // The tile's four counts per lane first, all four loads in flight and ahead of the carry's own round trip: behind a
  // barrier the compiler may not move a load up, and a slab rank's grid is ONE round of 722 blocks whose time is one
  // block's chain of round trips and barriers (18 us for 24 MB in the first form: load, scan with two barriers, store,
  // four times over behind the carry's scan -- ten barriers; r04 slab rank profile).  ONE barrier now: the carry is a sum,
  // not a scan, and the four sub-tiles' wave sums meet in LDS together with it.
  int4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = src[k * kBlock + threadIdx.x];
  // the tile's offset: the sum of the tile sums in front of it (at most a few thousand ints, L2-resident: cheaper than
  // the one-block launch that used to turn them into a prefix)
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) part += block_sums[i];
  for (int off = kWave / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, kWave);
  if (lane == 0) carry_sums[wid] = part;
  int mx = 0, s4[4], inc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (most of the box is empty: only counts that are there are cleared)
    if ((v[k].x | v[k].y | v[k].z | v[k].w) != 0) src[k * kBlock + threadIdx.x] = make_int4(0, 0, 0, 0);
    mx = max(max(mx, max(v[k].x, v[k].y)), max(v[k].z, v[k].w));
    s4[k] = (v[k].x + v[k].y) + (v[k].z + v[k].w);
    inc[k] = wave_inclusive_scan(s4[k]);
    if (lane == kWave - 1) wave_sums[k][wid] = inc[k];
  }
  __syncthreads();
  int carry = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) carry += carry_sums[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
      const int t = wave_sums[k][w];
      if (w < wid) off += t;
      tot += t;
    }
    const int ex = carry + off + inc[k] - s4[k];
    int4 o;
    o.x = ex;
    o.y = ex + v[k].x;
    o.z = o.y + v[k].y;
    o.w = o.z + v[k].z;
    dst[k * kBlock + threadIdx.x] = o;
    carry += tot;
  }
  for (int off = kWave / 2; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, kWave));
  // read first: almost every wave finds its maximum already recorded, and thousands of atomics on
  // one address would serialise
  if (stats != nullptr && (threadIdx.x & (kWave - 1)) == 0 && mx > 0 &&
      mx > __hip_atomic_load(&stats->max_cell_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(&stats->max_cell_count, mx);
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

// Deterministic in-cell order.  k_cell_rank's rank inside the cell is the order in which the waves'
// atomics happened to land, different from run to run -- and with it the order of every neighbour
// sum, i.e. the last bits of every result.  The particles of the cells k_cell_rank has marked (more
// than one run, or a run not ascending in id) are therefore scattered in two steps: k_scatter only puts
// their ids at the slots the atomic ranks name and marks them; k_scatter_ordered then
// lets every one of them count the ids of its cell that are smaller than its own and take that slot.
// Slots inside a cell are then ascending in particle id whatever the atomics did: the order of the
// oracle's DSLO_ORDER_CELL and, cell by cell, of the reference's bucket lists (lsh.go:113-118 appends
// in particle order).  Equal ids (there should be none) keep the atomic order among themselves.
struct ScatterOrder {
  const unsigned int* unordered;  // bitmap from k_cell_rank; nullptr = keep the order the atomics left
  int* keys;                      // ids at the atomic slots (marked cells only)
  unsigned char* later;           // per particle: 1 = belongs to a marked cell, k_scatter_ordered places it
  int* dest;                      // optional: final slot of every particle (to permute derived arrays)
  const int* cell_keys;           // the cells' key rows written by k_cell_rank (nullptr: two-pass ordering only)
  const int* overfull;            // set by k_cell_rank when some cell holds more than kCellKeys particles
};
// (No compacted work list: appending to one costs an atomic per wave on a single counter -- 250k
// same-address atomics per 16M-particle build serialise in L2 and took 2.8 ms once the flow had
// developed.  A byte per particle costs 32 MB of traffic and no atomics.)

// the cell a particle is sorted into, and (skin step) the reference position that decides it
struct SortPos {
  float x, y, z;
};
template <bool REF>
__device__ __forceinline__ SortPos sort_pos(const CSoa3& pos, const CSoa3& vel, float tau, int i) {
  if constexpr (REF) return SortPos{skin_ref(pos.x[i], vel.x[i], tau), skin_ref(pos.y[i], vel.y[i], tau), skin_ref(pos.z[i], vel.z[i], tau)};
  return SortPos{pos.x[i], pos.y[i], pos.z[i]};
}
template <bool REF>
__device__ __forceinline__ void scatter_move(const ScatterArrays& a, const ScatterOrder& o, int i, int d, int id,
                                             int* __restrict__ ids_dst, const Soa3& ref, const SortPos& sp) {
  ids_dst[d] = id;
  for (int f = 0; f < a.nf; ++f) a.dst[f][d] = a.src[f][i];
  if constexpr (REF) {  // (skin step: the reference positions in sorted order -- what the lists are built at)
    ref.x[d] = sp.x;
    ref.y[d] = sp.y;
    ref.z[d] = sp.z;
  }
  if (o.dest) o.dest[i] = d;
}

// `pos`: the unsorted positions the ranks were computed from
// (skin step: the host passes map 0 as a.ids_src and map 1 as a.ids_dst; the device state names the map that is current
// AFTER this rebuild -- the sort's destination.  The kernel argument itself stays untouched: written to, the whole
// struct moves from the kernarg segment into scratch memory and the scatter takes 5 x as long.)
struct ScatterIds {
  const int* src;
  int* dst;
};
__device__ __forceinline__ ScatterIds skin_ids(const ScatterArrays& a, const SkinGate& gate) {
  // (two selects on one flag, not "if (..) return {dst, src}; return {src, dst};": in the batched k_scatter hipcc 7.2
  // hoisted that form's swap of the two pointers above its test of ids_sel and swapped them on BOTH paths -- s_mov pairs
  // in front of the s_cmp in the ISA -- so that every second rebuild read the wrong map; tests/test_gpu_skin.py fails
  // on it at once.  This form compiles to s_cselect_b32.)
  const bool swap_ids = gate.st != nullptr && gate.st->ids_sel == 0;  // 1 -> 0
  return ScatterIds{swap_ids ? a.ids_dst : a.ids_src, swap_ids ? const_cast<int*>(a.ids_src) : a.ids_dst};
}
// The scatter is a chain of dependent memory round trips -- position -> cell -> cell_start -> key row -> stores --
// and with one particle per lane nothing but other waves covers them.  Everything that does not depend on the cell
// (the payload of every array, the id, the rank) is therefore requested up front, together with the position, and
// everything that depends on the cell alone (both ends of its slot range, its word of the `unordered` bitmap) in one
// second batch; the stores go out last, back to back.  (The first form moved the arrays in a loop of load -> wait ->
// store, one array at a time over pointers fetched from the kernel arguments inside the loop: 10 serial round
// trips per particle, 0.39 ms for the developed 16M flow at 30 % of the HBM roof; now 0.23.)
// NFMAX: 6 = positions and velocities only (every WCSPH build; no per-array test), kMaxScatter = any a.nf.
template <int NFMAX>
struct ScatterPayload {
  float v[NFMAX];
};
template <int NFMAX>
__device__ __forceinline__ void scatter_fetch(const ScatterArrays& a, int i, ScatterPayload<NFMAX>& p) {
#pragma unroll
  for (int f = 0; f < NFMAX; ++f) {
    p.v[f] = 0.0f;
    if (NFMAX == 6 || f < a.nf) p.v[f] = a.src[f][i];
  }
}
template <bool REF, int NFMAX>
__device__ __forceinline__ void scatter_store(const ScatterArrays& a, const ScatterOrder& o, int i, int d, int id,
                                              int* __restrict__ ids_dst, const Soa3& ref, const SortPos& sp,
                                              const ScatterPayload<NFMAX>& p) {
  ids_dst[d] = id;
#pragma unroll
  for (int f = 0; f < NFMAX; ++f)
    if (NFMAX == 6 || f < a.nf) a.dst[f][d] = p.v[f];
  if constexpr (REF) {  // (skin step: the reference positions in sorted order -- what the lists are built at)
    ref.x[d] = sp.x;
    ref.y[d] = sp.y;
    ref.z[d] = sp.z;
  }
  if (o.dest) o.dest[i] = d;
}

// kScatterUnroll particles per lane (i, i + kBlock, ...): twice the loads in flight per wave for the same chain.
#ifndef DSL_SCATTER_UNROLL
#define DSL_SCATTER_UNROLL 2
#endif
constexpr int kScatterUnroll = DSL_SCATTER_UNROLL;
template <bool REF = false, int NFMAX = kMaxScatter>
__global__ __launch_bounds__(kBlock) void k_scatter(DevConsts c, ScatterArrays a, ScatterOrder o, CSoa3 pos,
                                                    const int* __restrict__ rank,
                                                    const int* __restrict__ cell_start, SkinGate gate = SkinGate{nullptr},
                                                    CSoa3 vel = CSoa3{nullptr, nullptr, nullptr},
                                                    Soa3 ref = Soa3{nullptr, nullptr, nullptr}) {
  constexpr int U = kScatterUnroll;
  if (gate.closed()) return;
  const ScatterIds ids = skin_ids(a, gate);
  const int* ids_src = ids.src;
  int* ids_dst = ids.dst;
  const float tau = REF ? gate.st->tau : 0.0f;
  const int n = live_n(c);
  // the sort's positions (skin step: and velocities) are the first arrays of the payload in every caller: read once
  const bool pos_in_payload = a.nf >= (REF ? 6 : 3) && pos.x == a.src[0] && pos.y == a.src[1] && pos.z == a.src[2] &&
                              (!REF || (vel.x == a.src[3] && vel.y == a.src[4] && vel.z == a.src[5]));
  // (grid-stride: the skin step launches a capped grid, see k_cell_rank; every other caller's grid covers n / U in one trip)
  for (int first = blockIdx.x * (U * kBlock) + threadIdx.x; first < n; first += gridDim.x * (U * kBlock)) {
  ScatterPayload<NFMAX> pay[U];
  int id[U], r[U], cell[U], s[U], e[U], d[U];
  unsigned int uword[U];
  SortPos sp[U];
  bool in[U], later[U];
  // batch 1: everything that does not depend on the cell (a lane past the end reads the last particle and drops it)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    in[u] = first + u * kBlock < n;
    const int i = min(first + u * kBlock, n - 1);
    scatter_fetch<NFMAX>(a, i, pay[u]);
    id[u] = ids_src[i];
    r[u] = rank[i];
    if (pos_in_payload) {
      if constexpr (REF) sp[u] = SortPos{skin_ref(pay[u].v[0], pay[u].v[3], tau), skin_ref(pay[u].v[1], pay[u].v[4], tau), skin_ref(pay[u].v[2], pay[u].v[5], tau)};
      else sp[u] = SortPos{pay[u].v[0], pay[u].v[1], pay[u].v[2]};
    } else {
      sp[u] = sort_pos<REF>(pos, vel, tau, i);
    }
  }
  // batch 2: what depends on the cell alone
#pragma unroll
  for (int u = 0; u < U; ++u) {
    cell[u] = sort_cell(c, sp[u].x, sp[u].y, sp[u].z);
    const int cc = min(cell[u], c.ncell - 1);  // (a stale ghost -- pseudo cell ncell -- reads the last cell and is dropped)
    s[u] = cell_start[cc];
    e[u] = cell_start[cc + 1];
    uword[u] = o.unordered != nullptr ? o.unordered[cc >> 5] : 0u;
  }
  // the slot: the atomic rank, or -- a cell whose order the sort has to establish -- the number of smaller ids in the
  // cell's key row
#pragma unroll
  for (int u = 0; u < U; ++u) {
    d[u] = s[u] + r[u];
    later[u] = in[u] && cell[u] != c.ncell && ((uword[u] >> (cell[u] & 31)) & 1u) != 0u;
    if (later[u] && o.cell_keys != nullptr) {
      const int cnt = e[u] - s[u];
      if (cnt <= kCellKeys) {  // the common case: the cell's ids are in its key row, count the smaller ones
        const int4* row = reinterpret_cast<const int4*>(o.cell_keys + (size_t)cell[u] * kCellKeys);
        int below = 0;
        for (int k0 = 0; k0 < cnt; k0 += 8) {  // eight ids per trip (a cell holds ~8)
          const int4 v = row[k0 / 4], w = row[k0 / 4 + 1];
          const int key[8] = {v.x, v.y, v.z, v.w, w.x, w.y, w.z, w.w};
#pragma unroll
          for (int q = 0; q < 8; ++q)
            below += (k0 + q < cnt && (key[q] < id[u] || (key[q] == id[u] && k0 + q < r[u]))) ? 1 : 0;
        }
        d[u] = s[u] + below;
        later[u] = false;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = first + u * kBlock;
    if (!in[u]) continue;
    if (cell[u] != c.ncell) {  // (a stale ghost is dropped)
      if (later[u]) o.keys[d[u]] = id[u];
      else scatter_store<REF, NFMAX>(a, o, i, d[u], id[u], ids_dst, ref, sp[u], pay[u]);
    }
    if (o.unordered != nullptr && o.cell_keys == nullptr) o.later[i] = later[u] ? 1 : 0;
  }
  }
}

template <bool REF = false>
__global__ __launch_bounds__(kBlock) void k_scatter_ordered(DevConsts c, ScatterArrays a, ScatterOrder o, CSoa3 pos,
                                                            const int* __restrict__ rank,
                                                            const int* __restrict__ cell_start,
                                                            SkinGate gate = SkinGate{nullptr},
                                                            CSoa3 vel = CSoa3{nullptr, nullptr, nullptr},
                                                            Soa3 ref = Soa3{nullptr, nullptr, nullptr}) {
  if (gate.closed()) return;
  const ScatterIds ids = skin_ids(a, gate);
  const float tau = REF ? gate.st->tau : 0.0f;
  if (o.cell_keys != nullptr) {
    // fallback of the one-pass ordering: only the marked cells with more than kCellKeys members are left, and only
    // if k_cell_rank has seen such a cell at all.  A small grid strides over the particles.
    if (*o.overfull == 0) return;
    const int n = live_n(c);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
      const SortPos sp = sort_pos<REF>(pos, vel, tau, i);
      const int cell = sort_cell(c, sp.x, sp.y, sp.z);
      if (cell == c.ncell) continue;
      const int s = cell_start[cell], e = cell_start[cell + 1];
      if (e - s <= kCellKeys || !((o.unordered[cell >> 5] >> (cell & 31)) & 1u)) continue;
      const int id = ids.src[i], mine = s + rank[i];
      int below = 0;
      for (int k = s; k < e; ++k) {
        const int key = o.keys[k];
        below += (key < id || (key == id && k < mine)) ? 1 : 0;
      }
      scatter_move<REF>(a, o, i, s + below, id, ids.dst, ref, sp);
    }
    return;
  }
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c) || !o.later[i]) return;
  const SortPos sp = sort_pos<REF>(pos, vel, tau, i);
  const int cell = sort_cell(c, sp.x, sp.y, sp.z);
  const int id = ids.src[i];
  const int s = cell_start[cell], e = cell_start[cell + 1], mine = s + rank[i];
  int below = 0;
  // eight ids per trip, all eight loads in flight together (a cell holds ~8)
  for (int k0 = s; k0 < e; k0 += 8) {
    int key[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) key[u] = k0 + u < e ? o.keys[k0 + u] : INT_MAX;
#pragma unroll
    for (int u = 0; u < 8; ++u) below += (key[u] < id || (key[u] == id && k0 + u < mine)) ? 1 : 0;
  }
  scatter_move<REF>(a, o, i, s + below, id, ids.dst, ref, sp);
}

// dst[dest[i]] = src[i]: a derived per-particle array follows the sort
__global__ __launch_bounds__(kBlock) void k_permute1(DevConsts c, const int* __restrict__ dest, CSoa3 pos,
                                                     const float* __restrict__ src, float* __restrict__ dst) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= live_n(c)) return;
  if (sort_cell(c, pos.x[i], pos.y[i], pos.z[i]) == c.ncell) return;  // stale ghost: dropped, dest[i] not written
  dst[dest[i]] = src[i];
}

// ---------------------------------------------------------------------------------
// host <-> device layout conversion: the reference keeps xyz interleaved on the host
// (model/particle_array.go:5-15); the device keeps SoA in cell-sorted order with
// ids[slot] = original particle index.
// ---------------------------------------------------------------------------------
// `limit`: particles in the host buffer (N() for everything but positions, which hold Total());
// `zero_id` >= 0: that particle is stored at the origin (the reference's Get(N()), dslsph.hip)
__global__ __launch_bounds__(kBlock) void k_unpack3(int n, const float* __restrict__ stage, const int* __restrict__ ids,
                                                    float* __restrict__ x, float* __restrict__ y,
                                                    float* __restrict__ z, int limit, int zero_id) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const int o = ids[s];
  if (o >= limit) return;
  const bool zero = o == zero_id;
  x[s] = zero ? 0.0f : stage[3 * o];
  y[s] = zero ? 0.0f : stage[3 * o + 1];
  z[s] = zero ? 0.0f : stage[3 * o + 2];
}
__global__ __launch_bounds__(kBlock) void k_unpack1(int n, const float* __restrict__ stage, const int* __restrict__ ids,
                                                    float* __restrict__ x, int limit) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const int o = ids[s];
  if (o < limit) x[s] = stage[o];
}
// ParticleArray.AddBoundaryParticles (particle_array.go:123-128): nb position-only particles behind the
// current ones, ids first_id .., velocity 0
__global__ __launch_bounds__(kBlock) void k_append_boundary(int nb, int at, int first_id, int zero_id,
                                                            const float* __restrict__ stage, float* __restrict__ x,
                                                            float* __restrict__ y, float* __restrict__ z,
                                                            float* __restrict__ vx, float* __restrict__ vy,
                                                            float* __restrict__ vz, int* __restrict__ ids, Soa3 pcip,
                                                            Soa3 pciv, float* __restrict__ rho,
                                                            float* __restrict__ pterm) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= nb) return;
  const int d = at + k;
  rho[d] = 0.0f;  // as ParticleArray.Get reads a boundary particle (particle_array.go:94-117): density 0, P/rho^2 = 0/0
  pterm[d] = __uint_as_float(0x7fc00000u);
  const bool zero = first_id + k == zero_id;
  const float px = zero ? 0.0f : stage[3 * k], py = zero ? 0.0f : stage[3 * k + 1], pz = zero ? 0.0f : stage[3 * k + 2];
  x[d] = px;
  y[d] = py;
  z[d] = pz;
  vx[d] = 0.0f;
  vy[d] = 0.0f;
  vz[d] = 0.0f;
  ids[d] = first_id + k;
  if (pcip.x != nullptr) {
    pcip.x[d] = px;
    pcip.y[d] = py;
    pcip.z[d] = pz;
    pciv.x[d] = 0.0f;
    pciv.y[d] = 0.0f;
    pciv.z[d] = 0.0f;
  }
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
constexpr int kRecord = 7;      // full record: x,y,z,vx,vy,vz,id-bits
constexpr int kRecordPci = 13;  // ... + the PCISPH predictor state (_pos, _vel) once dsl_pcisph_begin has run
// position-only record: x,y,z,id-bits.  (The id used to stay behind, 12 bytes per record: the receiver numbered these
// ghosts itself, so inside a cell they sat in message order instead of id order -- the densities of the ghosts next
// to them were then summed in another order than a single-domain run sums them, and EXACT slabs were one ulp away
// from it in a few particles per step.  With the id every cell of every rank is in the single-domain order.)
constexpr int kRecordX = 4;

// Message layout (floats): header of `rec` words ([0] = full-record count, [1] =
// position-only count, int bits), cap_full full records of `rec` words (rec = kRecord or
// kRecordPci), cap_x position-only records.
// Particles within width_full of the plane (and migrants beyond it) travel as full records;
// the rest of the band only feeds the receiver's ghost densities and travels as positions.
struct SlabBands {
  float full_lo, band_lo;  // lo side: p < full_lo -> full record, else p < band_lo -> position only
  float full_hi, band_hi;  // hi side: p >= full_hi -> full record, else p >= band_hi -> position only
};
__device__ __forceinline__ float* slab_full_record(float* msg, int rec, int k) { return msg + (size_t)(k + 1) * rec; }
__device__ __forceinline__ float* slab_x_record(float* msg, int rec, int cap_full, int k) {
  return msg + (size_t)(cap_full + 1) * rec + (size_t)k * kRecordX;
}

// Band selection without atomics (a band is a contiguous slot range in cell order, so per-wave
// atomics on four counters would all collide: same-address atomics serialise): every block
// of kPackChunk slots counts its records per category, one block turns the counts into offsets
// and writes the headers, then the same blocks write their records at those offsets.  The
// message order is the slot order.
// Categories: [0] lo full, [1] hi full, [2] lo position-only, [3] hi position-only.
// `old_axis` (may be null) is the slab-axis coordinate array BEFORE the integration that wrote
// px..vz: with it only the particles of the band cell layers are looked at, which is what lets the
// pack run while the interior tiles are still being integrated (their new entries are not read).
constexpr int kPackIters = 8;
constexpr int kPackChunk = kBlock * kPackIters;  // slots per block; wave w owns kPackIters x 64 consecutive ones

// bit (4*it + k) of the result: slot (base + it*64 + lane) belongs to category k
__device__ __forceinline__ unsigned int slab_classify(const DevConsts& c, const SlabBands& sb,
                                                      const float* __restrict__ old_axis,
                                                      const float* __restrict__ pa, bool want_lo, bool want_hi,
                                                      int base, int lane, int n) {
  unsigned int flags = 0u;
#pragma unroll
  for (int it = 0; it < kPackIters; ++it) {
    const int i = base + it * kWave + lane;
    bool look = i < n;
    if (look && old_axis) {
      const int a = c.slab_axis;
      look = slab_band_cell(c, cell_coord(old_axis[i], c.gmin[a], c.inv_cell, c.dims[a]));
    }
    if (!look) continue;
    const float p = pa[i];  // NaN (a ghost of the step that has just been integrated) fails every test
    unsigned int f = 0u;
    if (want_lo) f |= p < sb.full_lo ? 1u : (p < sb.band_lo ? 4u : 0u);
    if (want_hi) f |= p >= sb.full_hi ? 2u : (p >= sb.band_hi ? 8u : 0u);
    flags |= f << (4 * it);
  }
  return flags;
}

__global__ __launch_bounds__(kBlock) void k_slab_count(DevConsts c, SlabBands sb, const float* __restrict__ old_axis,
                                                       const float* __restrict__ pa, int want_lo, int want_hi,
                                                       int* __restrict__ block_counts) {
  __shared__ int wsum[kBlock / kWave][4];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
  const int base = blockIdx.x * kPackChunk + wid * (kPackIters * kWave);
  const unsigned int flags = slab_classify(c, sb, old_axis, pa, want_lo, want_hi, base, lane, live_n(c));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < kPackIters; ++it) cnt += __builtin_popcountll(__ballot((flags >> (4 * it + k)) & 1u));
    if (lane == 0) wsum[wid][k] = cnt;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    int t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += wsum[w][threadIdx.x];
    block_counts[blockIdx.x * 4 + threadIdx.x] = t;
  }
}

// one block: exclusive scan of block_counts (in place) per category, then the two headers;
// overflows and high-water marks are kept for the host to find later.
// slab_state: [0] overflow, [1] high water full, [2] high water position-only
constexpr int kOffsBlock = 256;  // one wave per SIMD: fits next to a resident tile workgroup (split step)
__global__ __launch_bounds__(kOffsBlock) void k_slab_offsets(int* __restrict__ block_counts, int nblk, float* out_lo,
                                                             float* out_hi, int cap_full, int cap_x,
                                                             int* __restrict__ slab_state) {
  __shared__ int part[kOffsBlock / kWave][4];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid >> 6;
  const int per = (nblk + kOffsBlock - 1) / kOffsBlock;
  const int b0 = tid * per, b1 = min(b0 + per, nblk);
  int mine[4] = {0, 0, 0, 0};
  for (int b = b0; b < b1; ++b)
#pragma unroll
    for (int k = 0; k < 4; ++k) mine[k] += block_counts[b * 4 + k];
  int excl[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int inc = mine[k];
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(inc, o, kWave);
      if (lane >= o) inc += t;
    }
    excl[k] = inc - mine[k];
    if (lane == kWave - 1) part[wid][k] = inc;
  }
  __syncthreads();
  int total[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int before = 0, all = 0;
    for (int w = 0; w < kOffsBlock / kWave; ++w) {
      const int v = part[w][k];
      if (w < wid) before += v;
      all += v;
    }
    excl[k] += before;
    total[k] = all;
  }
  for (int b = b0; b < b1; ++b)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = block_counts[b * 4 + k];
      block_counts[b * 4 + k] = excl[k];
      excl[k] += v;
    }
  if (tid == 0) {
    for (int side = 0; side < 2; ++side) {
      float* o = side == 0 ? out_lo : out_hi;
      if (!o) continue;
      int nf = total[side], nx = total[2 + side];
      atomicMax(&slab_state[1], nf);
      atomicMax(&slab_state[2], nx);
      if (nf > cap_full) {
        atomicMax(&slab_state[0], nf);
        nf = cap_full;
      }
      if (nx > cap_x) {
        atomicMax(&slab_state[0], nx);
        nx = cap_x;
      }
      o[0] = __int_as_float(nf);
      o[1] = __int_as_float(nx);
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_slab_write(DevConsts c, SlabBands sb, const float* __restrict__ old_axis,
                                                       const float* __restrict__ pa, const float* __restrict__ px,
                                                       const float* __restrict__ py, const float* __restrict__ pz,
                                                       const float* __restrict__ vx, const float* __restrict__ vy,
                                                       const float* __restrict__ vz, const int* __restrict__ ids,
                                                       float* __restrict__ out_lo, float* __restrict__ out_hi,
                                                       int cap_full, int cap_x, const int* __restrict__ block_offsets,
                                                       int rec, CSoa3 pcip, CSoa3 pciv) {
  __shared__ int wsum[kBlock / kWave][4];
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
  const int base = blockIdx.x * kPackChunk + wid * (kPackIters * kWave);
  const unsigned int flags = slab_classify(c, sb, old_axis, pa, out_lo != nullptr, out_hi != nullptr, base, lane, live_n(c));
  unsigned long long bal[4][kPackIters];
  int cnt[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cnt[k] = 0;
#pragma unroll
    for (int it = 0; it < kPackIters; ++it) {
      bal[k][it] = __ballot((flags >> (4 * it + k)) & 1u);
      cnt[k] += __builtin_popcountll(bal[k][it]);
    }
    if (lane == 0) wsum[wid][k] = cnt[k];
  }
  __syncthreads();
  if ((cnt[0] | cnt[1] | cnt[2] | cnt[3]) == 0) return;  // wave-uniform
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (cnt[k] == 0) continue;
    int at = block_offsets[blockIdx.x * 4 + k];
    for (int w = 0; w < wid; ++w) at += wsum[w][k];
    float* msg = (k & 1) ? out_hi : out_lo;
#pragma unroll
    for (int it = 0; it < kPackIters; ++it) {
      if ((flags >> (4 * it + k)) & 1u) {
        const int i = base + it * kWave + lane;
        const int d = at + __builtin_popcountll(bal[k][it] & below);
        if (k < 2) {
          if (d < cap_full) {
            float* r = slab_full_record(msg, rec, d);
            r[0] = px[i];
            r[1] = py[i];
            r[2] = pz[i];
            r[3] = vx[i];
            r[4] = vy[i];
            r[5] = vz[i];
            r[6] = __int_as_float(ids[i]);
            if (rec == kRecordPci) {
              r[7] = pcip.x[i];
              r[8] = pcip.y[i];
              r[9] = pcip.z[i];
              r[10] = pciv.x[i];
              r[11] = pciv.y[i];
              r[12] = pciv.z[i];
            }
          }
        } else if (d < cap_x) {
          float* r = slab_x_record(msg, rec, cap_full, d);
          r[0] = px[i];
          r[1] = py[i];
          r[2] = pz[i];
          r[3] = __int_as_float(ids[i]);
        }
      }
      at += __builtin_popcountll(bal[k][it]);
    }
  }
}

// (r03, measured and removed: the same pack as ONE launch -- a block takes its place in the message with one atomic per
// category, the last block out, found by a done-counter, writes the headers -- took 204 us instead of the 30 us of
// count / offsets / write on a rank of 3M slots: ~1500 blocks' atomics on ONE done-counter serialise at ~130 ns
// each.  The same lesson as the band counters of round 1; profiles/README.md.)
__device__ __forceinline__ void slab_counts(const float* msg, int cap_full, int cap_x, int& nf, int& nx) {
  nf = __float_as_int(msg[0]);
  nx = __float_as_int(msg[1]);
  nf = nf < 0 ? 0 : (nf > cap_full ? cap_full : nf);
  nx = nx < 0 ? 0 : (nx > cap_x ? cap_x : nx);
}

// appends the records of up to two messages behind the current particles: per message the full
// records first, then the position-only ones (velocity 0; ghosts by construction: beyond width_full of the plane).
// blockIdx.y selects the message; the second one lands behind the first.
__global__ __launch_bounds__(kBlock) void k_slab_append(const float* __restrict__ msg0,
                                                        const float* __restrict__ msg1, int cap_full, int cap_x,
                                                        const int* __restrict__ n_cur, int room,
                                                        float* __restrict__ px, float* __restrict__ py,
                                                        float* __restrict__ pz, float* __restrict__ vx,
                                                        float* __restrict__ vy, float* __restrict__ vz,
                                                        int* __restrict__ ids, int* __restrict__ slab_state, int rec,
                                                        Soa3 pcip, Soa3 pciv, int axis, float shift0, float shift1) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  // periodic images: the records of message 0 / 1 are moved by shift0 / shift1 along the slab axis
  const float sh = blockIdx.y == 0 ? shift0 : shift1;
  const float shx = axis == 0 ? sh : 0.f, shy = axis == 1 ? sh : 0.f, shz = axis == 2 ? sh : 0.f;
  int nf0 = 0, nx0 = 0, nf1 = 0, nx1 = 0;
  if (msg0) slab_counts(msg0, cap_full, cap_x, nf0, nx0);
  if (msg1) slab_counts(msg1, cap_full, cap_x, nf1, nx1);
  const int at0 = *n_cur;
  if (at0 + nf0 + nx0 + nf1 + nx1 > room) {  // does not fit: record it, append nothing
    if (k == 0 && blockIdx.y == 0) atomicMax(&slab_state[0], at0 + nf0 + nx0 + nf1 + nx1);
    return;
  }
  const float* msg = blockIdx.y == 0 ? msg0 : msg1;
  if (!msg) return;
  const int nf = blockIdx.y == 0 ? nf0 : nf1, nx = blockIdx.y == 0 ? nx0 : nx1;
  const int at = blockIdx.y == 0 ? at0 : at0 + nf0 + nx0;
  if (k < nf) {
    const float* r = msg + (size_t)(k + 1) * rec;
    const int d = at + k;
    px[d] = r[0] + shx;
    py[d] = r[1] + shy;
    pz[d] = r[2] + shz;
    vx[d] = r[3];
    vy[d] = r[4];
    vz[d] = r[5];
    ids[d] = __float_as_int(r[6]);
    if (rec == kRecordPci) {  // a migrant keeps its predictor state (the reference never re-synchronises it)
      pcip.x[d] = r[7] + shx;
      pcip.y[d] = r[8] + shy;
      pcip.z[d] = r[9] + shz;
      pciv.x[d] = r[10];
      pciv.y[d] = r[11];
      pciv.z[d] = r[12];
    }
  } else if (k >= cap_full && k - cap_full < nx) {
    const int j = k - cap_full;
    const float* r = msg + (size_t)(cap_full + 1) * rec + (size_t)j * kRecordX;
    const int d = at + nf + j;
    px[d] = r[0] + shx;
    py[d] = r[1] + shy;
    pz[d] = r[2] + shz;
    vx[d] = 0.f;
    vy[d] = 0.f;
    vz[d] = 0.f;
    ids[d] = __float_as_int(r[3]);  // the global id: cells are ordered by id on every rank as in a single-domain run
    if (rec == kRecordPci) {  // ghosts: never predicted, any finite value will do
      pcip.x[d] = r[0] + shx;
      pcip.y[d] = r[1] + shy;
      pcip.z[d] = r[2] + shz;
      pciv.x[d] = 0.f;
      pciv.y[d] = 0.f;
      pciv.z[d] = 0.f;
    }
  }
}
__global__ void k_slab_bump(const float* __restrict__ msg0, const float* __restrict__ msg1, int cap_full, int cap_x,
                            int* __restrict__ n_cur, int room) {
  int nf0 = 0, nx0 = 0, nf1 = 0, nx1 = 0;
  if (msg0) slab_counts(msg0, cap_full, cap_x, nf0, nx0);
  if (msg1) slab_counts(msg1, cap_full, cap_x, nf1, nx1);
  const int add = nf0 + nx0 + nf1 + nx1;
  if (*n_cur + add <= room) *n_cur += add;
}
__global__ void k_set_count(int* __restrict__ n_cur, const int* __restrict__ src) { *n_cur = *src; }

__global__ __launch_bounds__(kBlock) void k_count_owned(DevConsts c, const float* __restrict__ px,
                                                        const float* __restrict__ py, const float* __restrict__ pz,
                                                        int* __restrict__ counter) {
  __shared__ int wsum[kBlock / kWave];
  int cnt = 0;
  const int n = live_n(c);
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)  // few blocks: few same-address atomics
    cnt += slab_owned(c, px[i], py[i], pz[i]) ? 1 : 0;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += wsum[w];
    if (t) atomicAdd(counter, t);
  }
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
