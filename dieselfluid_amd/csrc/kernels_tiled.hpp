// kernels_tiled.hpp -- LDS-tiled density and force+integrate kernels (FAST math mode).
//
// The lane-per-particle kernels of kernels_sph.hpp issue three to eight scattered global
// loads per candidate and are bound by vector-memory instruction issue, not by HBM or by
// the VALU (profiles/r01_v1_*).  Here a 256-thread workgroup owns a tile of 4x4x4 grid
// cells.  It stages the tile plus a one-cell halo (6x6x6 cells, 36 x-rows, each row one
// contiguous slot range thanks to the x-fastest cell order) from the SoA arrays in HBM
// into LDS as float4 records with coalesced loads, then every lane sweeps the 9 x-runs of
// its own particle with one ds_read_b128 (two for the force pass) per candidate.
// Out-of-range candidates contribute exactly zero through q = max(1 - r/h, 0), so the
// candidate loop needs no branch and may over-read into the 4 far-away pad records that
// end every staged row; that lets it be unrolled by 4 with a single bounds test.
//
// Neighbour masks: only ~29 of the 216 swept candidates are within h.  The density pass
// knows which (q > 0) and emits one 32-bit mask per x-run (9 words per particle, bit order =
// visit order); the force sweeps then walk only the set bits with the full per-pair
// arithmetic.  Masks stay valid while positions and slot order do (same step); runs longer
// than 32 candidates get a second word, runs longer than 64 clear their bit in the particle's
// "valid" word and are swept in full.
//
// Work distribution: k_tile_list compacts the non-empty tiles (the dam-break box is ~8x
// larger than the fluid); a persistent grid walks that list, contiguous chunks per XCD so
// that neighbouring tiles (which share halo rows) hit the same L2.
// Tiles whose halo does not fit the LDS budget fall back to the global-memory sweep for
// that tile only.
#pragma once

#include "kernels_sph.hpp"

namespace dsl {

constexpr int kTH = kTB + 2;           // with halo
constexpr int kTRows = kTH * kTH;      // 36 staged x-rows
constexpr int kTBlock = 512;           // threads per tile workgroup (8 waves)
constexpr int kTPad = 4;               // pad records per staged row (over-read guard)
constexpr int kTCap = 2304;            // staged records per tile (1.33 x the 1728 of 8 per cell)
constexpr float kFar = 1.0e15f;        // pad coordinate: finite, far outside any domain
constexpr int kTileLists = 5;
// per particle (SoA, stride = capacity): words 0-8 the masks of the 9 runs (first 32 candidates),
// words 10-18 the masks of candidates 32-63 (written and read only for runs that long).  Word 9 used to hold
// "which runs have masks": that is a rule on the run's length (at most 64 candidates), which every reader
// has from the tile table -- no longer written or read.
constexpr int kMaskWords = 19;
constexpr int kMaskValid = 9, kMaskHigh = 10;

struct TileMeta {
  // per interior row ir = 0..15 (ONE ds_read_b128 per target): .x = global slot of the row's target 0 minus the row's
  // first target index, .y = the same for its LDS record, .z = the target indices at which tile-local x cells 2 and 3
  // begin (16 bits each), .w = where cell 4 begins
  int4 trow[kTB * kTB];
  int row_gs[kTRows];        // first global slot of the staged row
  int row_len[kTRows];       // particles in the row
  int row_lds[kTRows + 1];   // first LDS record of the row (rows are kTPad apart)
  // the x-run of a target in tile-local x cell lx = 1..4 on staged row r: LDS records [first, end) of the row's
  // cells lx-1 .. lx+1, packed first | end << 16 (one ds_read_b32 where the sweeps used to read three table
  // entries and add them up, 9 times per target and kernel)
  int run[kTRows * kTB];
  int tprefix[kTB * kTB + 1];     // prefix of target counts over the 16 interior rows
  int overflow;
  int tile;                       // the tile's id in the tile grid
  int centre[3];                  // float bits: the tile centre in world coordinates (origin of the tile-relative records)
  int pad_;
  // k_density_pair: prefix, over the 16 interior rows, of the rows' PAIR SLOTS (a slot = two targets of one cell, or a
  // cell's odd one out: sum over the row's four cells of (count + 1) / 2)
  int pprefix[kTB * kTB + 1];
  int pad2_[3];                   // 360 ints: a whole number of 16-byte pieces
};
// A tile's table is built ONCE per neighbour build, by k_tile_desc, into a global array of these
// (one per non-empty tile, in the order of tile list 0); the sweeping kernels copy it into LDS, one
// dword per lane, one tile ahead of its use.  (Each kernel used to derive it again for every tile it
// visited -- seven dependent loads per row, a scan and two more barriers in front of every staging.)
constexpr int kMetaInts = 360;
static_assert(sizeof(TileMeta) == kMetaInts * sizeof(int), "TileMeta is copied as kMetaInts dwords");

// __syncthreads() with the wave's own LDS traffic drained first, stated explicitly.  hipcc leaves the
// `s_waitcnt lgkmcnt(0)` in front of an s_barrier to its waitcnt pass, and at the head of the double-buffered loops
// below (LDS writes at the END of the loop body, the barrier at its HEAD, reached over the back edge) that pass
// emitted none: the ISA had `ds_write_b128 ... s_branch ... s_barrier`.  A wave could then pass the barrier while a
// sibling's last records or table words were still on their way, and one 16M run in three met a stale record
// within a few thousand steps (densities off, then a blow-up): profiles/README.md, r03.
__device__ __forceinline__ void sync_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

struct TileGrid {
  int tnx, tny, tnz, ntiles;
  // Enumeration order of the tile list (k_tile_list): boxes of bx x by x bz tiles, the boxes x-fastest and the
  // tiles of a box x-fastest -- NOT the tile grid's own linear order.  The persistent kernels deal the list to
  // the 8 XCDs in groups of 128 consecutive entries (TileWalk), so with 128-tile boxes every XCD works on one
  // compact 3-D box of tiles at a time and the halo rows its tiles share are served by that XCD's L2: the
  // tiles of a box fetch 1.3 x its particles from HBM, where 128 tiles of the linear order (a 32 x 4 x 1 sheet)
  // fetch 1.7 x and share nothing with the sheets above and below.  bx = 0: linear order.
  int bx, by, bz, nbx, nby, nlist;  // nlist = list threads = boxes * box size
  // cell layers a tile owns along z: kTB, or 3 for the skin step (kernels_skin.hpp) -- its cells are h (1 + s) wide, and
  // 4 x 4 x 3 of them hold the ~512 targets and ~1900 staged records that 4 x 4 x 4 cells of edge h hold: the tile
  // tables keep their shape (interior rows 12..15 and staged rows 30..35 are simply empty)
  int tbz;
};
__device__ __forceinline__ int tile_of_list_thread(const TileGrid& tg, int t) {
  if (tg.bx == 0) return t < tg.ntiles ? t : -1;
  const int per = tg.bx * tg.by * tg.bz;
  const int box = t / per, w = t - box * per;
  const int ox = box % tg.nbx, oy = (box / tg.nbx) % tg.nby, oz = box / (tg.nbx * tg.nby);
  const int tx = ox * tg.bx + w % tg.bx, ty = oy * tg.by + (w / tg.bx) % tg.by, tz = oz * tg.bz + w / (tg.bx * tg.by);
  return (tx < tg.tnx && ty < tg.tny && tz < tg.tnz) ? tx + tg.tnx * (ty + tg.tny * tz) : -1;
}

#ifdef DSL_DIAG_STAMPS  // diagnostic build only: per-phase clocks of wave 0 of every block (s_memtime)
__device__ unsigned long long g_diag[32];
#define DSL_STAMP(var) const unsigned long long var = clock64()
#define DSL_STAMP_ADD(slot, a, b) \
  if (threadIdx.x == 0) atomicAdd(&g_diag[slot], (unsigned long long)((b) - (a)))
#else
#define DSL_STAMP(var)
#define DSL_STAMP_ADD(slot, a, b)
#endif

// ---------------------------------------------------------------------------------
// non-empty tile list
// ---------------------------------------------------------------------------------
// tiles holds kTileLists lists of tg.ntiles entries each: [0] every non-empty tile; slab mode
// also [4] the tiles with cell layers this slab owns and [3] the rest (ghosts only: the force
// pass just marks those for removal), and for the split force pass [1] the owning tiles that
// contain band cell layers and [2] those that contain other layers (a tile straddling the
// limit is in both).
// desc_of: for every entry of the lists 1.. the position of its tile in list 0, i.e. the index of
// its descriptor (k_tile_desc); list 0's own entries are their own positions.
__global__ __launch_bounds__(kBlock) void k_tile_list(DevConsts c, TileGrid tg, const int* __restrict__ cell_start,
                                                      int* __restrict__ tiles, int* __restrict__ n_tiles,
                                                      int* __restrict__ short_pass_tiles, int* __restrict__ n_live,
                                                      int* __restrict__ desc_of, unsigned int* __restrict__ unordered,
                                                      int unordered_words, int* __restrict__ overfull,
                                                      SkinGate gate = SkinGate{nullptr}) {
  if (gate.closed()) return;
  const int lt = blockIdx.x * kBlock + threadIdx.x;  // list thread: tiles are enumerated box by box (TileGrid)
  if (lt == 0) *overfull = 0;  // k_cell_rank's "a cell outgrew its key row" flag has been consumed by the scatter
  // the sort's "cells to order" bitmap (kernels_grid.hpp) has been consumed: clean for the next build
  // (64 cells per tile = two words per tile thread; saves the build a memset launch)
  if (unordered != nullptr) {
    if (2 * lt < unordered_words) unordered[2 * lt] = 0u;
    if (2 * lt + 1 < unordered_words) unordered[2 * lt + 1] = 0u;
  }
  // slab mode: the sort has dropped the stale ghosts; the live count (kept on the device) is the
  // start of the pseudo cell behind the last one.  Nothing in this launch reads the count.
  if (lt == 0 && n_live) *n_live = cell_start[c.ncell];
  const int lane = threadIdx.x & (kWave - 1);
  const int t = tile_of_list_thread(tg, lt);
  int cnt = 0;
  bool in_band = false, in_inner = false, owning = false, ghosts = false;
  if (t >= 0) {
    const int tx = t % tg.tnx, ty = (t / tg.tnx) % tg.tny, tz = t / (tg.tnx * tg.tny);
    const int nx = c.dims[0], ny = c.dims[1], nz = c.dims[2];
    const int xa = tx * kTB, xb = min(xa + kTB, nx);
    // (fixed trip counts: all 32 loads of the tile's 16 rows are in flight together)
#pragma unroll
    for (int dz = 0; dz < kTB; ++dz)
#pragma unroll
      for (int dy = 0; dy < kTB; ++dy) {
        const int z = tz * tg.tbz + dz, y = ty * kTB + dy;
        if (dz < tg.tbz && z < nz && y < ny) {
          const int row = (z * ny + y) * nx;
          cnt += cell_start[row + xb] - cell_start[row + xa];
        }
      }
    if (cnt > 0 && c.slab_axis >= 0) {
      const int a = c.slab_axis, ta = a == 0 ? tx : (a == 1 ? ty : tz);
      const int a0 = ta * kTB, a1 = min(a0 + kTB, c.dims[a]);
      owning = max(a0, c.own_c0) < min(a1, c.own_c1);
      ghosts = !owning;
      in_band = owning && (a0 < c.split_cl || a1 > c.split_ch);
      in_inner = owning && max(a0, c.split_cl) < min(a1, c.split_ch);
    }
  }
  // One atomic per BLOCK and list: same-address atomics serialise (one per wave was ~1500 of them on one
  // counter at 16M particles, 22 of the kernel's 40 us).  The waves' counts meet in LDS, thread l asks for
  // list l (l = kTileLists: the count of tiles whose last pass is short -- more than kTBlock targets, or at most
  // half of it -- from which the kernels pick the instantiation that shares such passes out).
  __shared__ int wave_count[kBlock / kWave][kTileLists + 1];
  __shared__ int block_base[kTileLists + 1];
  const int wid = threadIdx.x >> 6;
  auto member = [&](int l) {
    return l == 0 ? cnt > 0
                  : (l == 1 ? in_band
                            : (l == 2 ? in_inner
                                      : (l == 3 ? ghosts : (l == 4 ? owning : (cnt > kTBlock || (cnt > 0 && cnt <= kTBlock / 2))))));
  };
  unsigned long long mask_of[kTileLists + 1];
#pragma unroll
  for (int l = 0; l <= kTileLists; ++l) {
    mask_of[l] = __ballot(member(l));
    if (lane == 0) wave_count[wid][l] = __builtin_popcountll(mask_of[l]);
  }
  sync_lds();
  if (threadIdx.x <= kTileLists) {
    int total = 0;
    for (int w = 0; w < kBlock / kWave; ++w) total += wave_count[w][threadIdx.x];
    int* counter = threadIdx.x == kTileLists ? short_pass_tiles : n_tiles + threadIdx.x;
    block_base[threadIdx.x] = total > 0 ? atomicAdd(counter, total) : 0;
  }
  sync_lds();
  int pos0 = 0;
#pragma unroll
  for (int l = 0; l < kTileLists; ++l) {
    if (!member(l)) continue;
    int base = block_base[l];
    for (int w = 0; w < wid; ++w) base += wave_count[w][l];
    const int pos = base + __builtin_popcountll(mask_of[l] & ((1ull << lane) - 1ull));
    if (l == 0) pos0 = pos;
    tiles[(size_t)l * tg.ntiles + pos] = t;
    desc_of[(size_t)l * tg.ntiles + pos] = pos0;
  }
}

// ---------------------------------------------------------------------------------
// tile tables: row slot ranges, LDS offsets, per-cell offsets, target prefix (k_tile_desc)
// ---------------------------------------------------------------------------------
// tile_setup_load: lane r < kTRows loads the cell table entries of staged row r of the tile
struct TileSetupRegs {
  int s[kTH + 1];  // cell_start of the row's 6 cells and the end; all 0 for a row outside the grid
};
__device__ __forceinline__ void tile_setup_load(const DevConsts& c, const TileGrid& tg, int tile,
                                                const int* __restrict__ cell_start, TileSetupRegs& r) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k <= kTH; ++k) r.s[k] = 0;
  if (tid >= kTRows) return;
  const int tx = tile % tg.tnx, ty = (tile / tg.tnx) % tg.tny, tz = tile / (tg.tnx * tg.tny);
  const int nx = c.dims[0], ny = c.dims[1], nz = c.dims[2];
  const int ry = tid % kTH, rz = tid / kTH;
  const int y = ty * kTB - 1 + ry, z = tz * tg.tbz - 1 + rz;
  if (y >= 0 && y < ny && z >= 0 && z < nz && rz <= tg.tbz + 1) {
    const int row = (z * ny + y) * nx;
    // cells tx*4-1 .. tx*4+4, clamped to the grid: out-of-grid cells are empty
#pragma unroll
    for (int k = 0; k <= kTH; ++k) {
      int x = tx * kTB - 1 + k;
      x = x < 0 ? 0 : (x > nx ? nx : x);
      r.s[k] = cell_start[row + x];
    }
  }
}
// Query ROWS (PCISPH, binned queries, FAST): every grid cell owns kQueryRow record slots, a query takes slot
// atomicAdd(count[cell]) of its cell's row -- no prefix scan and no scatter pass between the predictor and the sweep.
// A tile's target table is then made of its cells' COUNTS: lane r builds the pseudo prefix 0, c1, c1+c2, ... of staged
// row r's four interior cells (the same shape tile_setup_load gives the particles' table) and leaves the counts zero
// for the next iteration -- every cell with a count belongs to exactly one listed tile.
constexpr int kQueryRow = 32;
// (keep: the counts stay -- the rows are carried into the next iteration, k_pci_predict_bin<.., INCR>)
__device__ __forceinline__ void tile_setup_load_counts(const DevConsts& c, const TileGrid& tg, int tile,
                                                       int* __restrict__ qcount, TileSetupRegs& r, bool keep) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k <= kTH; ++k) r.s[k] = 0;
  if (tid >= kTRows) return;
  const int tx = tile % tg.tnx, ty = (tile / tg.tnx) % tg.tny, tz = tile / (tg.tnx * tg.tny);
  const int nx = c.dims[0], ny = c.dims[1], nz = c.dims[2];
  const int ry = tid % kTH, rz = tid / kTH;
  const int y = ty * kTB - 1 + ry, z = tz * kTB - 1 + rz;
  if (ry >= 1 && ry <= kTB && rz >= 1 && rz <= kTB && y < ny && z < nz) {
    const int row = (z * ny + y) * nx;
    int cnt[kTB];
#pragma unroll
    for (int k = 0; k < kTB; ++k) {
      const int x = tx * kTB + k;
      cnt[k] = x < nx ? qcount[row + x] : 0;
    }
#pragma unroll
    for (int k = 0; k < kTB; ++k) {
      const int x = tx * kTB + k;
      if (cnt[k] != 0 && !keep) qcount[row + x] = 0;
      r.s[k + 2] = r.s[k + 1] + min(cnt[k], kQueryRow);
    }
    r.s[kTH] = r.s[kTB + 1];
  }
}

// One wave per non-empty tile (tile list 0): the tile's table, written where the sweeping kernels
// pick it up.  ~31k tiles x 1.5 KB at 16M particles.  Lane r holds staged row r; the two prefix sums
// run over the lanes (no LDS, no barrier), every lane writes its own row's entries.
// QUERY tiles (PCISPH with binned queries, k_pci_density_qtiled): `target_start` is then the prefix of the QUERY
// histogram over the same grid cells -- the tile's targets are the query records of its 64 interior cells, its staged
// rows the particles' as ever; target_start == cell_start is the particles' own build.
__global__ __launch_bounds__(kWave) void k_tile_desc(DevConsts c, TileGrid tg, const int* __restrict__ cell_start,
                                                     const int* __restrict__ target_start,
                                                     const int* __restrict__ tiles, const int* __restrict__ n_tiles,
                                                     int* __restrict__ desc, int* __restrict__ query_counts = nullptr,
                                                     int lds_cap = kTCap, SkinGate gate = SkinGate{nullptr},
                                                     bool keep_counts = false) {
  // (query_counts != nullptr: query rows, see tile_setup_load_counts; target_start is then unused)
  // (lds_cap: staged records the sweeping kernel's LDS image holds -- kTCap, or the skin step's wider image)
  if (gate.closed()) return;
  const int n = *n_tiles;
  const int lane = threadIdx.x;
  const int ry = lane % kTH, rz = lane / kTH;
  const bool row = lane < kTRows;
  const bool interior = row && ry >= 1 && ry <= kTB && rz >= 1 && rz <= tg.tbz;
  const bool queries = target_start != cell_start;
  // (two tiles per trip: the second one's loads travel under the first one's sums)
  auto table = [&](int item, int tile, const TileSetupRegs& r, const TileSetupRegs& q) {
    TileMeta* out = reinterpret_cast<TileMeta*>(desc + (size_t)item * kMetaInts);
    const int len = r.s[kTH] - r.s[0];
    const int v = row ? len + kTPad : 0;
    const int inc = wave_inclusive_scan(v);  // LDS offsets: rows are kTPad apart
    const int tv = interior ? q.s[kTB + 1] - q.s[1] : 0;
    const int tinc = wave_inclusive_scan(tv);  // targets: interior rows, interior cells
    int pv = 0;  // pair slots of the row's four interior cells
    if (interior) {
#pragma unroll
      for (int k = 1; k <= kTB; ++k) pv += (q.s[k + 1] - q.s[k] + 1) >> 1;
    }
    const int pinc = wave_inclusive_scan(pv);
    if (row) {
      out->row_gs[lane] = r.s[0];
      out->row_len[lane] = len;
      const int lds0 = inc - v;
      out->row_lds[lane] = lds0;
#pragma unroll
      for (int lx = 1; lx <= kTB; ++lx)
        out->run[lane * kTB + lx - 1] = (lds0 + (r.s[lx - 1] - r.s[0])) | ((lds0 + (r.s[lx + 2] - r.s[0])) << 16);
      // (a tile of fewer than kTB cell layers -- TileGrid::tbz -- still fills all 16 entries: the layers it does not own
      // are rows without targets, and the target lookup's binary search runs over all of them)
      if (ry >= 1 && ry <= kTB && rz >= 1 && rz <= kTB) {
        const int ir = (rz - 1) * kTB + (ry - 1), t0 = tinc - tv;  // the row's first target
        out->tprefix[ir] = t0;
        out->pprefix[ir] = pinc - pv;
        // (query rows: .x = the grid cell of the row's first interior cell -- a record is named by its cell and its
        // place in the cell's row)
        int first = q.s[1] - t0;
        if (query_counts != nullptr)
          first = (((tile / (tg.tnx * tg.tny)) * kTB + rz - 1) * c.dims[1] + ((tile / tg.tnx) % tg.tny) * kTB + ry - 1) * c.dims[0] +
                  (tile % tg.tnx) * kTB;
        out->trow[ir] = make_int4(first, lds0 + (r.s[1] - r.s[0]) - t0,
                                  (t0 + (q.s[2] - q.s[1])) | ((t0 + (q.s[3] - q.s[1])) << 16), t0 + (q.s[4] - q.s[1]));
      }
      if (lane == kTRows - 1) {  // (the last interior row is lane 28: this lane's inclusive sums are the totals)
        out->row_lds[kTRows] = inc;
        // (a tile that overflows the LDS budget is never staged: its packed run bounds may be garbage.  Target
        // indices fit 16 bits as long as the staged records do -- the targets are among them; QUERY targets are not,
        // and a tile with more of them than 16 bits count takes the global-memory sweep as well)
        out->overflow = (inc > lds_cap || tinc > 0xffff) ? 1 : 0;
        out->tprefix[kTB * kTB] = tinc;
        out->pprefix[kTB * kTB] = pinc;
        out->tile = tile;
        out->centre[0] = __float_as_int(c.gmin[0] + ((tile % tg.tnx) * kTB + 0.5f * kTB) * c.cell);
        out->centre[1] = __float_as_int(c.gmin[1] + (((tile / tg.tnx) % tg.tny) * kTB + 0.5f * kTB) * c.cell);
        out->centre[2] = __float_as_int(c.gmin[2] + ((tile / (tg.tnx * tg.tny)) * tg.tbz + 0.5f * tg.tbz) * c.cell);
      }
    }
  };
  for (int item = blockIdx.x; item < n; item += 2 * gridDim.x) {
    const int item2 = item + gridDim.x;
    const bool two = item2 < n;
    const int tile = tiles[item], tile2 = two ? tiles[item2] : 0;
    TileSetupRegs r, r2;
    tile_setup_load(c, tg, tile, cell_start, r);
    if (two) tile_setup_load(c, tg, tile2, cell_start, r2);
    if (query_counts != nullptr) {
      TileSetupRegs q, q2;
      tile_setup_load_counts(c, tg, tile, query_counts, q, keep_counts);
      if (two) tile_setup_load_counts(c, tg, tile2, query_counts, q2, keep_counts);
      table(item, tile, r, q);
      if (two) table(item2, tile2, r2, q2);
    } else if (queries) {
      TileSetupRegs q, q2;
      tile_setup_load(c, tg, tile, target_start, q);
      if (two) tile_setup_load(c, tg, tile2, target_start, q2);
      table(item, tile, r, q);
      if (two) table(item2, tile2, r2, q2);
    } else {
      table(item, tile, r, r);
      if (two) table(item2, tile2, r2, r2);
    }
  }
}

// the tiles that hold at least one QUERY (PCISPH with binned queries): one list, in the boxed order of k_tile_list;
// the list's length is cleared by k_pci_predict_bin at the head of the iteration
// (ROWS: `qstart` is the query COUNT per cell, see tile_setup_load_counts)
template <bool ROWS>
__global__ __launch_bounds__(kBlock) void k_qtile_list(DevConsts c, TileGrid tg, const int* __restrict__ qstart,
                                                       int* __restrict__ qtiles, int* __restrict__ n_qtiles,
                                                       const DevStats* stats) {
  if (stats->pci_done) return;
  const int lt = blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
  const int t = tile_of_list_thread(tg, lt);
  int cnt = 0;
  if (t >= 0) {
    const int tx = t % tg.tnx, ty = (t / tg.tnx) % tg.tny, tz = t / (tg.tnx * tg.tny);
    const int nx = c.dims[0], ny = c.dims[1], nz = c.dims[2];
    const int xa = tx * kTB, xb = min(xa + kTB, nx);
#pragma unroll
    for (int dz = 0; dz < kTB; ++dz)
#pragma unroll
      for (int dy = 0; dy < kTB; ++dy) {
        const int z = tz * kTB + dz, y = ty * kTB + dy;
        if (z < nz && y < ny) {
          const int row = (z * ny + y) * nx;
          if constexpr (ROWS) {
#pragma unroll
            for (int k = 0; k < kTB; ++k) cnt += xa + k < nx ? qstart[row + xa + k] : 0;
          } else {
            cnt += qstart[row + xb] - qstart[row + xa];
          }
        }
      }
  }
  __shared__ int wave_count[kBlock / kWave];
  __shared__ int block_base;
  const unsigned long long mine = __ballot(cnt > 0);
  if (lane == 0) wave_count[wid] = __builtin_popcountll(mine);
  sync_lds();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < kBlock / kWave; ++w) total += wave_count[w];
    block_base = total > 0 ? atomicAdd(n_qtiles, total) : 0;  // (one atomic per block: same-address atomics serialise)
  }
  sync_lds();
  if (cnt > 0) {
    int base = block_base;
    for (int w = 0; w < wid; ++w) base += wave_count[w];
    qtiles[base + __builtin_popcountll(mine & ((1ull << lane) - 1ull))] = t;
  }
}

// the copy of a tile's table into LDS, in two halves: the request (one dword per lane of the first six
// waves, nothing waits) and, once it has landed, the LDS write; a barrier makes it visible
__device__ __forceinline__ int tile_meta_request(const int* __restrict__ desc, int desc_index) {
  const int tid = threadIdx.x;
  return tid < kMetaInts ? desc[(size_t)desc_index * kMetaInts + tid] : 0;
}
__device__ __forceinline__ void tile_meta_store(TileMeta& m, int word) {
  const int tid = threadIdx.x;
  if (tid < kMetaInts) reinterpret_cast<int*>(&m)[tid] = word;
}

// target index inside the tile -> its staged row, global slot, LDS record and tile-local x cell (1..4): the cell is
// read off the row's cell table, i.e. exactly the cell the sort put the particle in (no second evaluation of the
// cell rule, no load of the position)
struct TileTarget {
  int srow, g, own, lx;
};
__device__ __forceinline__ TileTarget tile_target(const TileMeta& m, int t) {
  int ir = 0;  // largest interior row with tprefix[ir] <= t: binary search over the 16 rows
  ir += (t >= m.tprefix[ir + 8]) ? 8 : 0;
  ir += (t >= m.tprefix[ir + 4]) ? 4 : 0;
  ir += (t >= m.tprefix[ir + 2]) ? 2 : 0;
  ir += (t >= m.tprefix[ir + 1]) ? 1 : 0;
  const int4 w = m.trow[ir];
  TileTarget r;
  r.srow = (ir / kTB + 1) * kTH + (ir % kTB + 1);
  r.g = w.x + t;
  r.own = w.y + t;
  r.lx = 1 + ((t >= (w.z & 0xffff)) ? 1 : 0) + ((t >= (int)((unsigned)w.z >> 16)) ? 1 : 0) + ((t >= w.w) ? 1 : 0);
  return r;
}
// the LDS record range [j, je) of the x-run around tile-local x cell lx on staged row rr
__device__ __forceinline__ void tile_run(const TileMeta& m, int rr, int lx, int& j, int& je) {
  const unsigned int w = (unsigned int)m.run[rr * kTB + lx - 1];
  j = (int)(w & 0xffffu);
  je = (int)(w >> 16);
}

// ds_read_b128 serves a wave in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and
// the same +32 (MI355X_MICROARCH.md, LDS).  Lanes of one cell read the same record
// (broadcast); cells two apart are 16 records = one full 256-byte bank row apart and would
// collide.  Giving every 16-lane group the 16 targets of two x-adjacent cells leaves two
// distinct records 128 bytes apart per group: conflict-free for 8 particles per cell.
__device__ __forceinline__ int b128_group_slot(int lane) {
  const int l = lane & 31;
  int g, idx;
  if (l < 4) { g = 0; idx = l; }
  else if (l < 12) { g = 1; idx = l - 4; }
  else if (l < 16) { g = 0; idx = l - 8; }
  else if (l < 20) { g = 1; idx = l - 8; }
  else if (l < 28) { g = 0; idx = l - 12; }
  else { g = 1; idx = l - 16; }
  return (lane & 32) + 16 * g + idx;
}

// XCD-aware walk of the tile list: blocks b and b+8 share an XCD (and its L2).  The list is dealt
// to the XCDs in groups of up to kWalkGroupMax consecutive tiles (neighbouring tiles share halo rows, so a
// group mostly hits its XCD's L2), round-robin, so that runs of cheap tiles -- ghost layers,
// half-empty layers -- do not all land on one XCD.
constexpr int kWalkGroupMax = 128;
// groups in the whole list below which a group is not doubled again.  64 (groups of 128 tiles at 16M: 243 of them, 30 or 31
// per XCD -- 3 % more work for three of the eight XCDs) until round 4; 512 (groups of 32: 121 or 122 per XCD) measured
// +0.8 % on the bench line and +0.5 % on the developed flow in one call, 1024 / 2048 no better (profiles/r04_tile_groups.jsonl)
#ifndef DSL_WALK_MIN_GROUPS
#define DSL_WALK_MIN_GROUPS 512
#endif
constexpr int kWalkMinGroups = DSL_WALK_MIN_GROUPS;
struct TileWalk {
  int li, lstep, xcd, n;  // (kept to four scalars: the kernels that use it are short of SGPRs)
  __device__ __forceinline__ TileWalk(int n_tiles) {
    n = n_tiles;
    xcd = blockIdx.x & 7;
    li = blockIdx.x >> 3;
    lstep = (gridDim.x + 7) >> 3;
  }
  // group size: a power of two, at most kWalkGroupMax, small enough for >= 8 groups per XCD
  // (a short list dealt in big groups would leave some XCDs a whole group behind)
  __device__ __forceinline__ int group_shift() const {
    int s = 3;
    while ((2 << s) <= kWalkGroupMax && (n >> (s + 1)) >= kWalkMinGroups) ++s;
    return s;
  }
  __device__ __forceinline__ bool next(int& item) {
    const int gs = group_shift(), g = 1 << gs;
    const int lend = ((n + 8 * g - 1) / (8 * g)) * g;  // items an XCD may own
    while (li < lend) {
      item = ((((li >> gs) << 3) + xcd) << gs) + (li & (g - 1));
      li += lstep;
      if (item < n) return true;
    }
    return false;
  }
};

// The walk with the tile index fetched one tile ahead: the list entry of the tile after next is
// requested while the next tile is being set up, so that no wave ever sits out a memory latency for a
// list entry between a barrier and the loads that depend on it (it did: ~11k clocks per tile).
// It hands out descriptor indices (k_tile_desc): `desc_of` is the list's companion written by k_tile_list.
struct TileFeed {
  TileWalk walk;
  const int* __restrict__ desc_of;
  bool have_next;
  int next;
  __device__ __forceinline__ TileFeed(const int* __restrict__ d, int n_tiles) : walk(n_tiles), desc_of(d) {
    int item;
    have_next = walk.next(item);
    next = have_next ? (desc_of ? desc_of[item] : item) : 0;  // (desc_of == nullptr: the list's own positions)
  }
  __device__ __forceinline__ bool pop(int& desc_index) {
    if (!have_next) return false;
    desc_index = next;
    int item;
    have_next = walk.next(item);
    if (have_next) next = desc_of ? desc_of[item] : item;
    return true;
  }
  __device__ __forceinline__ void finish() {}
};

// The tile QUEUE (round 4; list 0 only, whose entries are their own descriptor indices): the tiles are not dealt in
// advance -- every XCD has a counter, a workgroup takes the next position of its XCD's sequence when it starts a tile (the
// same sequence the static walk deals out, so an XCD still works through one box of tiles at a time), and a workgroup
// that drew cheap tiles simply takes more of them.  A workgroup's FIRST tile is the static walk's (position blockIdx / 8
// of its XCD's sequence: no draw, nothing to wait for at the start -- 512 workgroups drawing from eight addresses at once
// cost a 1M-particle launch a third of its time); the counters hand out the positions behind those.  Thread 0 draws (one
// returning atomic, issued a whole tile before its result is needed) and hands the position on through two LDS words,
// written before the barrier that precedes the pop that reads them.  A launch site owns TWO blocks of eight counters and
// uses them alternately (the host flips): a launch zeroes the block it does NOT use, which is the one the next launch
// will -- no memset, and no count of finished workgroups, between launches.  Same interface as TileFeed.
struct TileQueue {
  int* ctr;
  int* slot;
  int n, xcd, gs, k, pending, first;
  __device__ __forceinline__ int item_of(int pos) const {
    const int g = 1 << gs;
    const int item = ((((pos >> gs) << 3) + xcd) << gs) + (pos & (g - 1));
    return item < n ? item : -1;  // (positions map to ascending items: the first one past the list ends the sequence)
  }
  // `counters`: this launch's block of 8; `other`: the block to leave at zero for the next launch
  __device__ __forceinline__ TileQueue(int n_tiles, int* counters, int* other, int* lds_slot)
      : ctr(counters), slot(lds_slot), n(n_tiles), k(0), pending(0) {
    const TileWalk w(n_tiles);
    xcd = w.xcd;
    gs = w.group_shift();
    first = (int)(gridDim.x + 7 - xcd) >> 3;  // workgroups on this XCD: their first tiles are positions 0 .. first - 1
    if (blockIdx.x == 0 && threadIdx.x < 8) other[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
      slot[0] = item_of((int)(blockIdx.x >> 3));
      pending = first + atomicAdd(&ctr[xcd], 1);
    }
    sync_lds();
  }
  __device__ __forceinline__ bool pop(int& desc_index) {
    const int item = __builtin_amdgcn_readfirstlane(slot[k & 1]);  // (the same word for every lane: a scalar from here on)
    if (threadIdx.x == 0) {
      slot[(k + 1) & 1] = item_of(pending);  // (read after the next barrier; its last readers passed the previous one)
      pending = first + atomicAdd(&ctr[xcd], 1);
    }
    k += 1;
    if (item < 0) return false;
    desc_index = item;
    return true;
  }
  __device__ __forceinline__ void finish() {}
};
// what a kernel instantiated with QUEUE walks its tile list with
template <bool QUEUE>
struct TileSource;
template <>
struct TileSource<false> {
  using type = TileFeed;
  static __device__ __forceinline__ TileFeed make(const int* __restrict__ desc_of, int n_tiles, int*, int*) { return TileFeed(desc_of, n_tiles); }
};
template <>
struct TileSource<true> {
  using type = TileQueue;
  // (`ctr`: this launch's block of 8 counters.  A launch site owns two blocks, 32 bytes each, the pair 64-byte aligned --
  // dslsph.hip: walk_ctr_of -- so the other block is found from the address)
  static __device__ __forceinline__ TileQueue make(const int* __restrict__, int n_tiles, int* ctr, int* slot) {
    const bool upper = ((reinterpret_cast<unsigned long long>(ctr) >> 5) & 1ull) != 0ull;  // blocks are 32 bytes, pairs 64-byte aligned
    return TileQueue(n_tiles, ctr, upper ? ctr - 8 : ctr + 8, slot);
  }
};

// q = clamp(a*b + c, 0, 1) in ONE instruction (VOP3 clamp output modifier).  The kernel
// weights are q = 1 - r^2/h^2 or 1 - r/h, never above 1, so this is exactly max(q, 0).
// the same with a wave-uniform factor (kept in an SGPR) and the constant 1: with three "v" operands
// the compiler re-materialised the uniform factor into a VGPR in every iteration of the pair loop
__device__ __forceinline__ float fma1_clamp01_uniform(float a, float b_uniform) {
  float d;
  asm("v_fma_f32 %0, %1, %2, 1.0 clamp" : "=v"(d) : "v"(a), "s"(b_uniform));
  return d;
}
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// mask = 2*mask + (q > 0): compare into VCC, then add-with-carry shifts the bit in
__device__ __forceinline__ void mask_push(unsigned int& mask, float q) {
  asm("v_cmp_lt_f32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(q) : "vcc");
}

// mask = 2*mask + (a < b)
__device__ __forceinline__ void mask_push_lt(unsigned int& mask, float a, float b) {
  asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(b) : "vcc");
}

// Staging.  The vector-memory path takes a wave instruction every 16 clocks whatever its width, and a
// tile used to cost it 120 (density) or 320 (force) one-dword loads.  Now lane t owns the four consecutive
// records 4k .. 4k+3 of staged row r (r = t / 14, k = t % 14: 36 rows x 14 quads on 504 lanes, 56 records
// per row in one go, longer rows finish in a loop): ONE unaligned dwordx4 load per source array brings its
// four values, 3 or 8 loads per lane and tile, and the lane writes its four records as whole float4s.
// A quad may read up to three elements past its row (the next row's, or the padding behind the arrays);
// they are never stored.
constexpr int kQuadsPerRow = 14;
static_assert(kTRows * kQuadsPerRow <= kTBlock, "one quad per lane");
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 load4u(const float* __restrict__ p) {
  const f4u v = *reinterpret_cast<const f4u*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float f4_at(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

template <int NF>
struct StageRegs {
  float4 v[NF];  // v[a] = elements g .. g+3 of source array a
};
// issue: nothing waits here.  Needs row_gs / row_len only.  load4(g, out[NF]): the quad loads at slot g.
// (vt: the quad slot this lane stages, threadIdx.x in the 512-thread kernels; the 256-thread pair kernel calls twice)
template <int NF, class Load4>
__device__ __forceinline__ void stage_issue(const TileMeta& m, Load4&& load4, StageRegs<NF>& sr, int vt = -1) {
  const int t = vt < 0 ? (int)threadIdx.x : vt, r = t / kQuadsPerRow, k = t - r * kQuadsPerRow;
  if (r < kTRows && 4 * k < m.row_len[r]) load4(m.row_gs[r] + 4 * k, sr.v);
}
// commit: registers -> LDS records, the rest of rows longer than 56, the pad records.
// load1(g, out[NF]): one record; store(slot, rec[NF], real)
template <int NF, class Load1, class Store>
__device__ __forceinline__ void stage_commit(const TileMeta& m, const StageRegs<NF>& sr, Load1&& load1, Store&& store, int vt = -1) {
  const int t = vt < 0 ? (int)threadIdx.x : vt, r = t / kQuadsPerRow, k = t - r * kQuadsPerRow;
  if (r >= kTRows) return;
  const int len = m.row_len[r], ls = m.row_lds[r];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (4 * k + i < len) {
      float rec[NF];
#pragma unroll
      for (int a = 0; a < NF; ++a) rec[a] = f4_at(sr.v[a], i);
      store(ls + 4 * k + i, rec, true);
    }
  }
  for (int i = 4 * kQuadsPerRow + k; i < len; i += kQuadsPerRow) {  // rare: more than 56 particles in the row
    float rec[NF];
    load1(m.row_gs[r] + i, rec);
    store(ls + i, rec, true);
  }
  const int npad = m.row_lds[r + 1] - ls - len;
  if (k < npad) {
    float rec[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) rec[a] = 0.0f;
    store(ls + len + k, rec, false);
  }
}
// EXACT: are the first NC source arrays' staged values of this lane all acceptable to exact_div (sph_device.hpp)?
// A row longer than the 56 records the quads cover fails (its tail never passes through these registers).
template <int NF, int NC>
__device__ __forceinline__ bool stage_values_ok(const TileMeta& m, const StageRegs<NF>& sr) {
  const int t = threadIdx.x, r = t / kQuadsPerRow, k = t - r * kQuadsPerRow;
  if (r >= kTRows) return true;
  const int len = m.row_len[r];
  bool ok = len <= 4 * kQuadsPerRow;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (4 * k + i < len) {
#pragma unroll
      for (int a = 0; a < NC; ++a) ok = ok && exact_div_ok(f4_at(sr.v[a], i));
    }
  }
  return ok;
}
// (CHECK: returns stage_values_ok of the first three source arrays, the coordinates)
template <int NF, bool CHECK = false, class Load4, class Load1, class Store>
__device__ __forceinline__ bool stage_rows(const TileMeta& m, Load4&& load4, Load1&& load1, Store&& store) {
  StageRegs<NF> sr;
  DSL_STAMP(s0);
  stage_issue<NF>(m, load4, sr);
  DSL_STAMP(s1);
  DSL_STAMP_ADD(NF == 8 ? 8 : 12, s0, s1);  // staging: load issue
  stage_commit<NF>(m, sr, load1, store);
  DSL_STAMP(s2);
  DSL_STAMP_ADD(NF == 8 ? 9 : 13, s1, s2);  // staging: wait for the data + LDS writes
  if constexpr (CHECK) return stage_values_ok<NF, 3>(m, sr);
  return true;
}

// Which instantiation of the tiled kernels sweeps this neighbour build: the one that shares short
// passes out (SHARE) when more than one tile in twelve has a short last pass, else the plain one.  The
// host launches BOTH; each reads the tile statistics k_tile_list left on the device and the one that is
// not wanted returns at once (~4 us).  (The host used to choose from statistics it polled in mapped
// memory without waiting: which launch saw which build's statistics depended on timing, and the two
// instantiations round differently -- results varied from run to run.)
// stats = the tile-list counters: [0] non-empty tiles, [5] tiles with a short last pass; nullptr = always run
__device__ __forceinline__ bool share_wanted(const int* __restrict__ stats) {
  return (long long)stats[5] * 12 > (long long)stats[0];
}

// Passes over a tile's targets (see k_density_tiled).  Full passes: one lane per target,
// body(false_type, t, 0, 1), lanes permuted for conflict-free ds_read_b128.  When at most half a
// block of targets is left, k = 2, 4, 8 or 16 adjacent lanes share each target:
// body(true_type, t, sub, k) with sub = 0..k-1; the lanes of a group run the body together, so it
// may combine their sums with __shfl_xor over offsets 1..k/2.
// SHARE = false compiles the full passes only (every pass one lane per target): the instantiation
// for a scene whose tiles hold exactly kTBlock targets (a lattice at rest) is a few per cent faster
// without the second copy of the body; the host picks per step from the tile statistics.
// NINE: a remainder of more than BLOCK / 16 targets -- too many for sixteen lanes each -- is shared out NINE lanes to a
// target, seven groups per wave (lane 63 idles), one run per lane: body(true_type, t, sub, 9), whose sums the body
// combines itself over the group's lanes (nine_lane_sum).  With powers of two the developed flow's pair sweep (281 slots
// per tile on 256 lanes: 25 left) gave each of them eight lanes, i.e. TWO runs for lane 0 -- the pass takes as long as its
// busiest lane -- where nine lanes take one.
__device__ __forceinline__ float nine_lane_sum(float v) {
  const int g0 = ((threadIdx.x & (kWave - 1)) / 9) * 9;
  float s = __shfl(v, g0, kWave);
#pragma unroll
  for (int j = 1; j < 9; ++j) s += __shfl(v, g0 + j, kWave);
  return s;
}
template <bool SHARE, int BLOCK = kTBlock, bool NINE = false, class Body>
__device__ __forceinline__ void for_each_target(int ntarg, int tid, int tperm, Body&& body) {
  int tbase = 0;
  for (; SHARE ? ntarg - tbase > BLOCK / 2 : tbase < ntarg; tbase += BLOCK) {
    const int t = tbase + tperm;
    if (t < ntarg) body(std::false_type{}, t, 0, 1);
  }
  if constexpr (SHARE)
  while (tbase < ntarg) {
    const int rem = ntarg - tbase;
    if constexpr (NINE) {
      constexpr int kGroups = (BLOCK / kWave) * 7;  // nine-lane groups per block
      if (rem > BLOCK / 16 && rem <= 2 * kGroups) {
        const int lane = tid & (kWave - 1);
        const int t = (tid / kWave) * 7 + lane / 9;
        if (lane < 63 && t < rem) body(std::true_type{}, tbase + t, lane % 9, 9);
        tbase += min(rem, kGroups);
        continue;
      }
    }
    const int shift = rem > BLOCK / 4 ? 1 : (rem > BLOCK / 8 ? 2 : (rem > BLOCK / 16 ? 3 : 4));  // 16: one run per lane
    const int t = tid >> shift;
    if (t < rem) body(std::true_type{}, tbase + t, tid & ((1 << shift) - 1), 1 << shift);
    tbase += min(rem, BLOCK >> shift);
  }
}

// ---------------------------------------------------------------------------------
// D (tiled): densities + P/rho^2
//
// Candidate coordinates are staged RELATIVE TO THE TILE CENTRE together with w = -|x|^2/h^2, so
// that q = 1 - r^2/h^2 = (1 + w_i) + w_j + (2/h^2) xi.xj costs one add and three fma, the last
// with the clamp modifier (7 VALU instructions per candidate with the mask push and the
// accumulation; the kernel is bound by FP32 issue, profiles/r01_v3_pmc.md).
// Tile-relative values stay below 3.5 cells, which bounds the cancellation error of the
// expanded form at ~2e-6 -- inside the FAST-mode tolerance, and invisible to the cut-off
// (a candidate that far out contributes ~1e-12).
// ---------------------------------------------------------------------------------
// EXACT = true (DSL_MATH_EXACT): the reference's own float32 operations, one rounding each
// (sph_field.go:155-172, std_kernel.go:33-39, vector.go:301-308), summed in the reference's order --
// candidates cell by cell, ascending particle id inside a cell -- so the result is bit for bit the
// oracle's.  Two phases per run: the sweep forms r^2 = (dx*dx + dy*dy) + dz*dz for every candidate and
// pushes (r^2 < r2_thr) into the mask, r2_thr being the smallest float whose correctly rounded square root
// is >= h, i.e. exactly the reference's `dist < h`; then the set bits are walked in ascending candidate
// order with IEEE sqrt and divide, a handful of candidates instead of all of them.  One lane per target
// always (sharing a target out would re-associate the sum).
template <bool SHARE, bool EXACT = false>
// (4 waves per SIMD = two workgroups per CU.  Held to 80 VGPRs for three workgroups the FAST sweep still
// compiles without spills and runs 7 % SLOWER, 0.845 against 0.794 ms at 16M: the kernel is bound by vector
// instruction issue, not by latency, and more resident waves only add contention.)
__global__ __launch_bounds__(kTBlock, 4) void k_density_tiled(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                          const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                          const int* __restrict__ cell_start, Bnd bnd, CSoa3 p,
                                                          float* __restrict__ rho, float* __restrict__ pterm,
                                                          unsigned int* __restrict__ nmask, int mstride) {
  // FAST: the LDS image is DOUBLE-BUFFERED (r03).  A tile's life used to be barrier, staging (load issue, one exposed
  // memory round trip, LDS writes), barrier, sweep: per-phase clocks put the sweep at 57 % of it, and with two
  // workgroups per CU a quarter of the time NEITHER was sweeping (profiles/r03_*).  Now the loads of tile k+1 are
  // issued right behind the barrier that starts the sweep of tile k (12 registers per lane carry them through it),
  // every wave commits them to the OTHER image as soon as ITS OWN sweep is over -- no barrier in between: that image
  // was last read by the sweep of tile k-1, which every wave had left before this tile's barrier -- and the table of
  // tile k+2 rotates through a third copy the same way.  One barrier per tile, no exposed round trip.
  // 2 x 36.9 KB + 3 tables = 78 KB per workgroup: two workgroups still fit a CU's 160 KB.
  // EXACT keeps one image (its second array holds the pre-filter's tile-relative records) and the old order.
#ifdef DSL_DENSITY_SINGLE_BUFFER  // (A/B and diagnostic builds)
  constexpr bool DB = false;
#else
  constexpr bool DB = !EXACT;
#endif
  __shared__ TileMeta metas[DB ? 3 : 2];
  __shared__ float4 Abuf[DB ? 2 : 1][kTCap];
  // EXACT: A holds the raw coordinates the exact walk needs, R the tile-relative records of the FAST test, which
  // the candidate sweep uses as a conservative pre-filter (see below)
  __shared__ float4 R[EXACT ? kTCap : 1];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  static_assert(!(EXACT && SHARE), "the exact sum is sequential: one lane per target");
  // (n_tiles is the base of the tile-list counters here; a slab always has half-empty ghost tiles: the host
  // launches the pass-sharing instantiation alone)
  if (!EXACT && c.slab_axis < 0 && share_wanted(n_tiles) != SHARE) return;
  auto load4 = [&](int g, float4* o) {
    o[0] = load4u(p.x + g);
    o[1] = load4u(p.y + g);
    o[2] = load4u(p.z + g);
  };
  auto load1 = [&](int g, float* o) {
    o[0] = p.x[g];
    o[1] = p.y[g];
    o[2] = p.z[g];
  };
  // EXACT: the walk divides with exact_div (sph_device.hpp), which equals `/` bit for bit while every staged
  // coordinate is 0 or between 2^-20 and 2^20 in magnitude; a tile with any other value takes the global sweep
  // (checked on the staging registers, stage_values_ok; rows longer than 56 records are not exempt: their tail is
  // loaded record by record, so such a tile is simply taken as unsafe)
  // registers -> records of image `img`, relative to the centre of the tile `mt` describes
  auto commit = [&](const TileMeta& mt, const StageRegs<3>& sr, float4* img) {
    const float ox = __int_as_float(mt.centre[0]), oy = __int_as_float(mt.centre[1]), oz = __int_as_float(mt.centre[2]);
    stage_commit<3>(mt, sr, load1, [&](int slot, const float* o, bool real) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
      if (real) {
        const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
        v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
      }
      if constexpr (EXACT) {  // raw coordinates; a pad is far away from everything
        img[slot] = real ? make_float4(o[0], o[1], o[2], 0.0f) : make_float4(kFar, kFar, kFar, 0.0f);
        R[slot] = v;
      } else {
        img[slot] = v;
      }
    });
  };
  // the sweep of one staged tile: table m, image A
  auto sweep = [&](const TileMeta& m, const float4* __restrict__ A, bool force_global) {
    DSL_STAMP(d2);
    const bool ovf = m.overflow != 0 || force_global;
    const int ntarg = m.tprefix[kTB * kTB];
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(lane);
    // Targets are taken kTBlock at a time, one lane each.  A tile's LDS slot is held for as long as
    // its slowest pass, so a short pass -- the few targets beyond kTBlock once the lattice has
    // melted, or a half-empty tile at a slab end -- is shared out: 2, 4 or 8 adjacent lanes per
    // target, each sweeping every k-th of the 9 runs, sums combined by shuffles (for_each_target).
    for_each_target<SHARE>(ntarg, tid, tperm, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      const TileTarget tt = tile_target(m, t);
      const int srow = tt.srow, g = tt.g, lx = tt.lx;
      float acc = 0.0f, acc1 = 0.0f, self_term = 1.0f;
      if constexpr (EXACT) {
        if (!ovf) {
          const int own = tt.own;
          const float4 me = A[own];
          const int pad_rec = m.row_lds[1] - kTPad;
          float density = 0.0f;
          // The candidate sweep only has to find a SUPERSET of {dist < h}: the walk below forms the reference's
          // own dist and applies the reference's own cut-off to every candidate it visits (one outside adds +0), and so
          // does the force walk over the same masks.  So the sweep uses the FAST test on tile-relative records,
          // q = 1 - r^2/h^2 in 4 operations, and keeps everything above -1e-3: its error is ~3e-5 for a target inside
          // its tile (|x| <= 2.5 cells, candidates within 3.5).  A target far outside (a particle beyond the grid box,
          // clamped into an edge cell) would lose that bound to cancellation: a wave with such a lane takes the
          // exact r^2 test (one wave-uniform branch per chunk).
          const float4 mer = R[own];
          const float two_hh = 2.0f * c.inv_hh;
          const float qsx = two_hh * mer.x, qsy = two_hh * mer.y, qsz = two_hh * mer.z, qa0 = 1.0f + mer.w;
          const float reach = 4.0f * c.h;
          const bool far_lane = !(fabsf(mer.x) <= reach && fabsf(mer.y) <= reach && fabsf(mer.z) <= reach);
          const bool prefilter = __builtin_amdgcn_ballot_w64(far_lane) == 0ull;
          // SPHField.Density's loop body for one candidate record (sph_field.go:164-170): exactly the
          // operations of k_density<false>; a pad record or the particle itself adds +0
          // (x^2 / h^2 by exact_div: the constant's reciprocal refinement is hoisted; bit for bit `/` on this tile)
          const ExactDivisor Dhh = exact_divisor_uniform(c.hh);
          auto add = [&](const float4& cnd, bool counts) {
            const float dx = me.x - cnd.x, dy = me.y - cnd.y, dz = me.z - cnd.z;
            const float dist = dsl_sqrt<false>(dist2<false>(dx, dy, dz));
            float w = 0.0f;  // kern_F<false> (std_kernel.go:33-39)
            if (!(dist >= c.h)) {
              const float xx = dist * dist;
              const float q = 1.0f - exact_div(xx, Dhh);
              const float aq = c.A * q;
              w = aq * q;
            }
            w = counts ? w : 0.0f;
            const float mw = c.mass * w;
            density = density + mw;
          };
          auto run_by_run = [&](int ri) {
            const int rr = srow + (ri / 3 - 1) * kTH + (ri % 3 - 1);
            int j, je;
            tile_run(m, rr, lx, j, je);
            if (!(j < je)) nmask[(size_t)ri * mstride + g] = 0u;  // an empty run still has a (read) mask word
            // chunks of up to 32 candidates: sweep -> mask word -> walk of its set bits, first candidate first
            for (int chunk = 0; j < je; ++chunk, j += 32) {
              const int jend = min(je, j + 32);
              unsigned int mask = 0u;
              if (prefilter) {
                for (int jj = j; jj < jend; jj += 4) {
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                    const float4 cnd = R[jj + u];
                    const float t = __builtin_fmaf(cnd.z, qsz, __builtin_fmaf(cnd.y, qsy, __builtin_fmaf(cnd.x, qsx, cnd.w + qa0)));
                    mask_push_lt(mask, -1.0e-3f, t);
                  }
                }
              } else {
                for (int jj = j; jj < jend; jj += 4) {
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                    const float4 cnd = A[jj + u];
                    const float dx = me.x - cnd.x, dy = me.y - cnd.y, dz = me.z - cnd.z;
                    mask_push_lt(mask, dist2<false>(dx, dy, dz), c.r2_thr);
                  }
                }
              }
              // the sweep tests whole groups of 4: the up to 3 records behind the run belong to a cell the
              // reference's stencil does not visit (or are pads); their bits, the lowest, are dropped
              mask &= ~0u << ((4 - ((jend - j) & 3)) & 3);
              if (chunk == 0) nmask[(size_t)ri * mstride + g] = mask;
              else if (chunk == 1) nmask[(size_t)(kMaskHigh + ri) * mstride + g] = mask;
              const int top = j + ((jend - j + 3) & ~3) - 1;  // record of bit 0
              unsigned int mm = mask;
              auto take = [&](bool& counts) {
                const bool has = mm != 0u;
                const int b = 31 - __builtin_clz(mm | 1u);  // highest set bit = earliest candidate
                const int idx = has ? top - b : pad_rec;
                mm &= ~(1u << b);
                mm = has ? mm : 0u;
                counts = has && idx != own;  // `if i != pIndex` (sph_field.go:164)
                return idx;
              };
              if (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
                bool cp, cq;
                float4 pr = A[take(cp)];
                bool more;
                do {  // two candidates per trip, the next trip's record requested before the arithmetic
                  const float4 qr = A[take(cq)];
                  add(pr, cp);
                  more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
                  const bool cq2 = cq;
                  pr = A[take(cp)];
                  add(qr, cq2);
                } while (more);
              }
            }
          };
          // Three runs per walk (the z-plane's, as in the force kernel): the candidate sweeps of the three runs first -- their
          // mask words are what the force walk reads later anyway -- then ONE wave-synchronised loop over the set bits of
          // all three, a lane taking its runs in order and each run's bits earliest candidate first: the reference's
          // order, hence the same sum bit for bit, and the wave waits for the busiest lane of the plane once instead of
          // three times.  A wave with a run of more than 32 candidates in the plane (its second chunk has to follow its
          // first directly) keeps the loop per run.
#pragma unroll 1
          for (int g3 = 0; g3 < 3; ++g3) {
            int jj[3], jje[3];
            bool longrun = false;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              tile_run(m, srow + (g3 - 1) * kTH + (u - 1), lx, jj[u], jje[u]);
              longrun |= jje[u] - jj[u] > 32;
            }
            if (__builtin_amdgcn_ballot_w64(longrun) != 0ull) {
#pragma unroll 1
              for (int u = 0; u < 3; ++u) run_by_run(3 * g3 + u);
              continue;
            }
            unsigned int mk[3];
            int tp[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int ri = 3 * g3 + u, j = jj[u], jend = jje[u];
              unsigned int mask = 0u;
              if (prefilter) {
                for (int q4 = j; q4 < jend; q4 += 4) {
#pragma unroll
                  for (int v = 0; v < 4; ++v) {
                    const float4 cnd = R[q4 + v];
                    const float t = __builtin_fmaf(cnd.z, qsz, __builtin_fmaf(cnd.y, qsy, __builtin_fmaf(cnd.x, qsx, cnd.w + qa0)));
                    mask_push_lt(mask, -1.0e-3f, t);
                  }
                }
              } else {
                for (int q4 = j; q4 < jend; q4 += 4) {
#pragma unroll
                  for (int v = 0; v < 4; ++v) {
                    const float4 cnd = A[q4 + v];
                    const float dx = me.x - cnd.x, dy = me.y - cnd.y, dz = me.z - cnd.z;
                    mask_push_lt(mask, dist2<false>(dx, dy, dz), c.r2_thr);
                  }
                }
              }
              mask &= ~0u << ((4 - ((jend - j) & 3)) & 3);
              nmask[(size_t)ri * mstride + g] = mask;
              mk[u] = mask;
              tp[u] = j + ((jend - j + 3) & ~3) - 1;  // record of bit 0
            }
            // the queue: empty words to the back, one shift per exhausted word
            unsigned int mm = mk[0], n1 = mk[1], n2 = mk[2];
            int top = tp[0], tp1 = tp[1], tp2 = tp[2];
            if (n1 == 0u) {
              n1 = n2;
              tp1 = tp2;
              n2 = 0u;
            }
            if (mm == 0u) {
              mm = n1;
              top = tp1;
              n1 = n2;
              tp1 = tp2;
              n2 = 0u;
            }
            auto take = [&](bool& counts) {
              const bool has = mm != 0u;
              const int b = 31 - __builtin_clz(mm | 1u);  // highest set bit = earliest candidate
              const int idx = has ? top - b : pad_rec;
              mm = has ? (mm & ~(1u << b)) : 0u;
              counts = has && idx != own;  // `if i != pIndex` (sph_field.go:164)
              if (mm == 0u) {
                mm = n1;
                top = tp1;
                n1 = n2;
                tp1 = tp2;
                n2 = 0u;
              }
              return idx;
            };
            if (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
              bool cp, cq;
              float4 pr = A[take(cp)];
              bool more;
              do {
                const float4 qr = A[take(cq)];
                add(pr, cp);
                more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
                const bool cq2 = cq;
                pr = A[take(cp)];
                add(qr, cq2);
              } while (more);
            }
          }
          acc = density;
        } else if (sub == 0) {
          const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
          for_each_grid_candidate(c, cell_start, xi, yi, zi, [&](int j) {
            if (j == g) return;
            const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
            const float dist = dsl_sqrt<false>(dist2<false>(dx, dy, dz));
            if (dist < c.h) {
              const float w = kern_F<false>(c, dist);
              acc += c.mass * w;
            }
          });
        }
        if (bnd.is(g)) {
          rho[g] = 0.0f;
          pterm[g] = __uint_as_float(0x7fc00000u);
          return;
        }
        rho[g] = acc;
        const float pr = tait_eos<false>(c, acc, c.eos_d0_grad);
        pterm[g] = dsl_div<false>(pr, acc * acc);
        return;
      }
      if (!ovf) {
        const float4 me = A[tt.own];
        // a particle whose position has gone NaN (the reference produces such next to boundary particles)
        // meets nobody, itself included: q is NaN, clamped to 0, for every candidate
        self_term = (me.x == me.x && me.y == me.y && me.z == me.z) ? 1.0f : 0.0f;
        const float two_hh = 2.0f * c.inv_hh;
        const float sx = two_hh * me.x, sy = two_hh * me.y, sz = two_hh * me.z, a0 = 1.0f + me.w;
        // one x-run of candidates (row rr of the staged tile, the 3 cells around the target's)
        auto sweep_run = [&](int ri, int rr) {
          int j, je;
          tile_run(m, rr, lx, j, je);
          unsigned int mask = 0u;
#ifdef DSL_DIAG_NO_SWEEP  // timing-only build: the per-tile fixed cost (set-up + staging + epilogue)
          je = j;
#endif
          // q = clamp(1 - r^2/h^2), r^2 = |xi|^2 + |xj|^2 - 2 xi.xj, with everything but the three
          // products folded into the staged w_j = -|xj|^2/h^2 and the per-target constants
          auto test4 = [&](int jj) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float4 cnd = A[jj + u];
              const float q = fma_clamp01(cnd.z, sz, __builtin_fmaf(cnd.y, sy, __builtin_fmaf(cnd.x, sx, cnd.w + a0)));
              mask_push(mask, q);
              if (u & 1) acc1 = __builtin_fmaf(q, q, acc1);
              else acc = __builtin_fmaf(q, q, acc);
            }
          };
          auto sweep_to = [&](int jend) {  // 8 candidates per trip while they last, then at most one block of 4
            for (; j + 4 < jend; j += 8) {
              test4(j);
              test4(j + 4);
            }
            if (j < jend) {
              test4(j);
              j += 4;
            }
          };
          const int j32 = j + 32;
          sweep_to(min(je, j32));
          nmask[(size_t)ri * mstride + g] = mask;
          if (je > j32) {  // a second word for candidates 32-63 (rare: occupancy 8.3 +- 2 per cell once melted);
            mask = 0u;     // a run longer than 64 leaves garbage in it, its valid bit is clear
            sweep_to(je);
            nmask[(size_t)(kMaskHigh + ri) * mstride + g] = mask;
          }
        };
        if constexpr (!SHARED) {
          int ri = 0;
#pragma unroll 1
          for (int dz = -kTH; dz <= kTH; dz += kTH) {
#pragma unroll 1
            for (int dy = -1; dy <= 1; ++dy, ++ri) sweep_run(ri, srow + dz + dy);
          }
        } else {
#pragma unroll 1
          for (int ri = sub; ri < 9; ri += k) sweep_run(ri, srow + (ri / 3 - 1) * kTH + (ri % 3 - 1));
        }
        acc += acc1;
      } else if (sub == 0) {
        const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
        for_each_grid_candidate(c, cell_start, xi, yi, zi, [&](int j) {
          if (j == g) return;
          const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
          const float r2 = dist2<true>(dx, dy, dz);
          if (r2 < c.hh) {
            const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
            acc = __builtin_fmaf(c.mass * c.A, q * q, acc);
          }
        });
      }
      if constexpr (SHARED) {
        for (int o = 1; o < k; o <<= 1) {  // the lanes of a group are active together
          acc += __shfl_xor(acc, o, kWave);
        }
        if (sub != 0) return;
      }
      if (!ovf) acc = (acc - self_term) * (c.mass * c.A);  // the particle met itself once (q = 1)
      if (bnd.is(g)) {  // a boundary particle reads as density 0, P/rho^2 = 0/0 (Bnd, sph_device.hpp)
        rho[g] = 0.0f;
        pterm[g] = __uint_as_float(0x7fc00000u);
        return;
      }
      rho[g] = acc;
      // An isolated particle (rho = 0) has no neighbour for which the reference would ever form
      // P/rho^2 (sph_field.go:183-199 only does so inside the j != i loop); the masked sweeps do
      // visit the particle itself, so its own term has to be a harmless 0 rather than 0/0.
      const float pr = tait_eos<true>(c, acc, c.eos_d0_grad);
      pterm[g] = acc > 0.0f ? dsl_div<true>(pr, acc * acc) : 0.0f;
    });
    DSL_STAMP(d3);
    DSL_STAMP_ADD(6, d2, d3);
  };
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (!have) return;
  tile_meta_store(metas[0], tile_meta_request(desc, di));
  if constexpr (DB) {
    bool have_next = feed.pop(di);
    int table_word = have_next ? tile_meta_request(desc, di) : 0;
    sync_lds();  // the first tile's table is visible
    {                 // the first tile is staged in the open: nothing to hide it under yet
      StageRegs<3> sr;
      if (!metas[0].overflow) {
        stage_issue<3>(metas[0], load4, sr);
        commit(metas[0], sr, Abuf[0]);
      }
    }
    if (have_next) tile_meta_store(metas[1], table_word);
    int mc = 0, mn = 1, mnn = 2;  // tables of this tile, the next one, the one after
    for (int cur = 0; have; cur ^= 1) {
      DSL_STAMP(d0);
      sync_lds();  // image `cur` is complete, the next tile's table visible, image `cur ^ 1` and table `mnn` free
      DSL_STAMP(d1);
      DSL_STAMP_ADD(4, d0, d1);
      StageRegs<3> sr;
      bool have_nn = false, stage_next = false;
      table_word = 0;
      if (have_next) {
        stage_next = metas[mn].overflow == 0;
        if (stage_next) stage_issue<3>(metas[mn], load4, sr);
        have_nn = feed.pop(di);
        if (have_nn) table_word = tile_meta_request(desc, di);
      }
      DSL_STAMP(d1b);
      DSL_STAMP_ADD(12, d1, d1b);  // next tile: load issue
      sweep(metas[mc], Abuf[cur], false);
      DSL_STAMP(d4);
      if (stage_next) commit(metas[mn], sr, Abuf[cur ^ 1]);
      if (have_nn) tile_meta_store(metas[mnn], table_word);
      DSL_STAMP(d5);
      DSL_STAMP_ADD(13, d4, d5);  // next tile: wait for its data (if it has not landed under the sweep) + LDS writes
      have = have_next;
      have_next = have_nn;
      const int t = mc;
      mc = mn;
      mn = mnn;
      mnn = t;
    }
  } else {
    for (int cur = 0; have; cur ^= 1) {
      TileMeta& m = metas[cur];
      sync_lds();  // the previous tile's sweep is over: its LDS records are free, this tile's table is visible
      have = feed.pop(di);
      int table_word = 0;
      if (have) table_word = tile_meta_request(desc, di);
      bool unsafe = EXACT && !exact_div_ok(c.h);
      if (!m.overflow) {
        StageRegs<3> sr;
        stage_issue<3>(m, load4, sr);
        commit(m, sr, Abuf[0]);
        if constexpr (EXACT) unsafe |= !stage_values_ok<3, 3>(m, sr);
      }
      if (have) tile_meta_store(metas[cur ^ 1], table_word);  // (requested before the staging loads: it has landed with them)
      bool force_global = false;
      if constexpr (EXACT) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (see sync_lds)
        force_global = __syncthreads_or(unsafe ? 1 : 0) != 0;
      }
      else sync_lds();
      sweep(m, Abuf[0], force_global);
    }
  }
}

// ---------------------------------------------------------------------------------
// D (tiled), TWO TARGETS PER LANE (round 3; FAST arithmetic, single domain and slabs)
//
// The density sweep is bound by VALU issue, and a third of what it issues is not candidate tests: per-run set-up, loop
// control, the wait for every record (profiles/r03_mfma_pipelined_experiment.md; a probe build that ran a SECOND test on
// every record it read took 1.19 ms for twice the tests where the kernel takes 0.79 for once).  Two targets of the SAME
// cell sweep the same nine runs, so a lane takes a pair of them: every record is read once and tested twice, every run is
// set up once for two.  256-thread workgroups (a tile's 512 targets are 256 pairs), four per CU (39.6 KB of LDS each),
// single-buffered.  A slot is two targets of one cell, or a cell's odd one out (its second test then runs on a target
// that nothing can meet); TileMeta::pprefix is the slots' prefix over the tile's 16 interior rows.  Same records,
// same q, same masks, same sums in the same order per target as k_density_tiled<.., false>: the results are the same bits.
// ---------------------------------------------------------------------------------
constexpr int kPBlock = 256;
#ifndef DSL_PAIR_NINE
#define DSL_PAIR_NINE 1
#endif
constexpr bool kPairNine = DSL_PAIR_NINE != 0;  // nine-lane groups for the pair sweep's remainders (for_each_target)
struct PairSlot {
  int srow, lx, g, own;  // staged row and tile-local x cell of the slot's cell; global slot and LDS record of its first target
  int rowbase, j;        // trow[ir].x and the first target's place inside its cell (query rows: k_pci_density_qpair)
  bool two;              // the slot holds two targets (g + 1, own + 1 is the second)
};
__device__ __forceinline__ PairSlot pair_slot(const TileMeta& m, int u) {
  int ir = 0;  // largest interior row with pprefix[ir] <= u
  ir += (u >= m.pprefix[ir + 8]) ? 8 : 0;
  ir += (u >= m.pprefix[ir + 4]) ? 4 : 0;
  ir += (u >= m.pprefix[ir + 2]) ? 2 : 0;
  ir += (u >= m.pprefix[ir + 1]) ? 1 : 0;
  const int4 w = m.trow[ir];
  const int b1 = m.tprefix[ir], b2 = w.z & 0xffff, b3 = (int)((unsigned)w.z >> 16), b4 = w.w, b5 = m.tprefix[ir + 1];
  const int n1 = (b2 - b1 + 1) >> 1, n2 = (b3 - b2 + 1) >> 1, n3 = (b4 - b3 + 1) >> 1;
  int q = u - m.pprefix[ir];
  int lx = 1, begin = b1, end = b2;
  if (q >= n1) {
    q -= n1, lx = 2, begin = b2, end = b3;
    if (q >= n2) {
      q -= n2, lx = 3, begin = b3, end = b4;
      if (q >= n3) q -= n3, lx = 4, begin = b4, end = b5;
    }
  }
  const int t0 = begin + 2 * q;
  PairSlot r;
  r.srow = (ir / kTB + 1) * kTH + (ir % kTB + 1);
  r.lx = lx;
  r.g = w.x + t0;
  r.own = w.y + t0;
  r.rowbase = w.x;
  r.j = 2 * q;
  r.two = t0 + 1 < end;
  return r;
}

// WIDE (the skin step's candidate sweep, kernels_skin.hpp): the grid's cells are h (1 + s) wide, the masks take every
// candidate within that distance (`wide_thr` = 1 - (1 + s)^2, less a rounding margin: the test is 1 - r^2/h^2 > wide_thr),
// densities are not formed here (k_density_list forms them, every step, from the lists these masks become); the tiles
// are 4 x 4 x 3 of those cells (TileGrid::tbz), whose image fits the same kTCap records.
template <bool SHARE, bool WIDE = false>
__global__ __launch_bounds__(kPBlock, 4) void k_density_pair(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                             const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                             const int* __restrict__ cell_start, Bnd bnd, CSoa3 p,
                                                             float* __restrict__ rho, float* __restrict__ pterm,
                                                             unsigned int* __restrict__ nmask, int mstride,
                                                             float wide_thr = 0.0f, SkinGate gate = SkinGate{nullptr}) {
  __shared__ TileMeta metas[2];
  __shared__ float4 A[kTCap];
  const int tid = threadIdx.x;
  if (gate.closed()) return;
  // (WIDE: the skin step launches the pass-sharing instantiation alone -- its tiles of 4 x 4 x 3 wide cells hold a few
  // targets more or less than a block -- and saves a launch that would only find out it is not wanted)
  if (!WIDE && c.slab_axis < 0 && share_wanted(n_tiles) != SHARE) return;
  auto load4 = [&](int g, float4* o) {
    o[0] = load4u(p.x + g);
    o[1] = load4u(p.y + g);
    o[2] = load4u(p.z + g);
  };
  auto load1 = [&](int g, float* o) {
    o[0] = p.x[g];
    o[1] = p.y[g];
    o[2] = p.z[g];
  };
  // a tile's table: kMetaInts dwords, two per lane
  auto meta_request = [&](int desc_index, int (&w)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * kPBlock;
      w[k] = i < kMetaInts ? desc[(size_t)desc_index * kMetaInts + i] : 0;
    }
  };
  auto meta_store = [&](TileMeta& m, const int (&w)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * kPBlock;
      if (i < kMetaInts) reinterpret_cast<int*>(&m)[i] = w[k];
    }
  };
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (!have) return;
  {
    int w[2];
    meta_request(di, w);
    meta_store(metas[0], w);
  }
  for (int cur = 0; have; cur ^= 1) {
    TileMeta& m = metas[cur];
    sync_lds();  // the previous tile's sweep is over: its LDS records are free, this tile's table is visible
    have = feed.pop(di);
    int tw[2] = {0, 0};
    if (have) meta_request(di, tw);
    const bool ovf = m.overflow != 0;
    if (!ovf) {
      const float ox = __int_as_float(m.centre[0]), oy = __int_as_float(m.centre[1]), oz = __int_as_float(m.centre[2]);
      auto store = [&](int slot, const float* o, bool real) {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
        if (real) {
          const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
          v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
        }
        A[slot] = v;
      };
      // 36 rows x 14 quads = 504 quad slots on 256 lanes: two per lane, all six loads in flight together
      StageRegs<3> sr0, sr1;
      stage_issue<3>(m, load4, sr0, tid);
      stage_issue<3>(m, load4, sr1, tid + kPBlock);
      stage_commit<3>(m, sr0, load1, store, tid);
      stage_commit<3>(m, sr1, load1, store, tid + kPBlock);
    }
    if (have) meta_store(metas[cur ^ 1], tw);
    sync_lds();
    if (ovf) {
      // A tile beyond the LDS budget: its targets one per lane, by the target prefix alone, the grid's 27 cells from
      // global memory.  (Not by pair slots: a tile table packs its in-row cell boundaries into 16 bits -- TileMeta::trow
      // -- and a tile of more than 65535 particles, which only ever exists as such a tile, would get wrong slots.)
      if constexpr (!WIDE) {
        const int ntarg = m.tprefix[kTB * kTB];
        for (int t = tid; t < ntarg; t += kPBlock) {
          const int g = tile_target(m, t).g;
          if (bnd.is(g)) {
            rho[g] = 0.0f;
            pterm[g] = __uint_as_float(0x7fc00000u);
            continue;
          }
          const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
          float a = 0.0f;
          for_each_grid_candidate(c, cell_start, xi, yi, zi, [&](int j) {
            if (j == g) return;
            const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
            const float r2 = dist2<true>(dx, dy, dz);
            if (r2 < c.hh) {
              const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
              a = __builtin_fmaf(c.mass * c.A, q * q, a);
            }
          });
          rho[g] = a;
          const float pr = tait_eos<true>(c, a, c.eos_d0_grad);
          pterm[g] = a > 0.0f ? dsl_div<true>(pr, a * a) : 0.0f;
        }
      }
      continue;
    }
    const int nslots = m.pprefix[kTB * kTB];
    for_each_target<SHARE, kPBlock, kPairNine>(nslots, tid, tid, [&](auto shared_c, int u, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      const PairSlot ps = pair_slot(m, u);
      const int srow = ps.srow, lx = ps.lx, g0 = ps.g;
      const bool two = ps.two;
      float acc[2] = {0.0f, 0.0f}, accb[2] = {0.0f, 0.0f}, self_term[2] = {1.0f, 1.0f};
      if (!ovf) {
        const float4 me0 = A[ps.own];
        float4 me1 = A[ps.own + (two ? 1 : 0)];
        // a particle whose position has gone NaN (the reference produces such next to boundary particles)
        // meets nobody, itself included: q is NaN, clamped to 0, for every candidate
        self_term[0] = (me0.x == me0.x && me0.y == me0.y && me0.z == me0.z) ? 1.0f : 0.0f;
        self_term[1] = (me1.x == me1.x && me1.y == me1.y && me1.z == me1.z) ? 1.0f : 0.0f;
        const float two_hh = 2.0f * c.inv_hh;
        const float sx0 = two_hh * me0.x, sy0 = two_hh * me0.y, sz0 = two_hh * me0.z, a00 = 1.0f + me0.w;
        const float sx1 = two_hh * me1.x, sy1 = two_hh * me1.y, sz1 = two_hh * me1.z;
        const float a01 = two ? 1.0f + me1.w : -1.0e30f;  // (an odd one out: its second test meets nobody)
        // one x-run of candidates (row rr of the staged tile, the 3 cells around the slot's cell), both targets
        auto sweep_run = [&](int ri, int rr) {
          int j, je;
          tile_run(m, rr, lx, j, je);
          unsigned int mask0 = 0u, mask1 = 0u;
          auto test4 = [&](int jj) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const float4 cnd = A[jj + v];
              if constexpr (WIDE) {  // 1 - r^2/h^2 unclamped against the skin's threshold; nothing is summed
                const float t0 = __builtin_fmaf(cnd.z, sz0, __builtin_fmaf(cnd.y, sy0, __builtin_fmaf(cnd.x, sx0, cnd.w + a00)));
                const float t1 = __builtin_fmaf(cnd.z, sz1, __builtin_fmaf(cnd.y, sy1, __builtin_fmaf(cnd.x, sx1, cnd.w + a01)));
                mask_push_lt(mask0, wide_thr, t0);
                mask_push_lt(mask1, wide_thr, t1);
                continue;
              }
              const float q0 = fma_clamp01(cnd.z, sz0, __builtin_fmaf(cnd.y, sy0, __builtin_fmaf(cnd.x, sx0, cnd.w + a00)));
              const float q1 = fma_clamp01(cnd.z, sz1, __builtin_fmaf(cnd.y, sy1, __builtin_fmaf(cnd.x, sx1, cnd.w + a01)));
              mask_push(mask0, q0);
              mask_push(mask1, q1);
              if (v & 1) {
                accb[0] = __builtin_fmaf(q0, q0, accb[0]);
                accb[1] = __builtin_fmaf(q1, q1, accb[1]);
              } else {
                acc[0] = __builtin_fmaf(q0, q0, acc[0]);
                acc[1] = __builtin_fmaf(q1, q1, acc[1]);
              }
            }
          };
          auto sweep_to = [&](int jend) {  // 8 candidates per trip while they last, then at most one block of 4
            for (; j + 4 < jend; j += 8) {
              test4(j);
              test4(j + 4);
            }
            if (j < jend) {
              test4(j);
              j += 4;
            }
          };
          const int j32 = j + 32;
          sweep_to(min(je, j32));
          nmask[(size_t)ri * mstride + g0] = mask0;
          if (two) nmask[(size_t)ri * mstride + g0 + 1] = mask1;
          if (je > j32) {  // a second word for candidates 32-63; a run longer than 64 leaves garbage in it, its valid bit is clear
            mask0 = mask1 = 0u;
            sweep_to(je);
            nmask[(size_t)(kMaskHigh + ri) * mstride + g0] = mask0;
            if (two) nmask[(size_t)(kMaskHigh + ri) * mstride + g0 + 1] = mask1;
          }
        };
        if constexpr (!SHARED) {
          int ri = 0;
#pragma unroll 1
          for (int dz = -kTH; dz <= kTH; dz += kTH) {
#pragma unroll 1
            for (int dy = -1; dy <= 1; ++dy, ++ri) sweep_run(ri, srow + dz + dy);
          }
        } else {
#pragma unroll 1
          for (int ri = sub; ri < 9; ri += k) sweep_run(ri, srow + (ri / 3 - 1) * kTH + (ri % 3 - 1));
        }
        acc[0] += accb[0];
        acc[1] += accb[1];
      } else if (!WIDE && sub == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (e == 1 && !two) break;
          const int g = g0 + e;
          const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
          float a = 0.0f;
          for_each_grid_candidate(c, cell_start, xi, yi, zi, [&](int j) {
            if (j == g) return;
            const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
            const float r2 = dist2<true>(dx, dy, dz);
            if (r2 < c.hh) {
              const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
              a = __builtin_fmaf(c.mass * c.A, q * q, a);
            }
          });
          acc[e] = a;
        }
      }
      if constexpr (WIDE) return;  // (masks only: a tile that overflows the image has none, k_list_build marks its targets)
      if constexpr (SHARED) {
        if (k == 9) {  // (nine adjacent lanes, one run each: for_each_target<.., NINE>)
          acc[0] = nine_lane_sum(acc[0]);
          acc[1] = nine_lane_sum(acc[1]);
        } else {
          for (int o = 1; o < k; o <<= 1) {  // the lanes of a group are active together
            acc[0] += __shfl_xor(acc[0], o, kWave);
            acc[1] += __shfl_xor(acc[1], o, kWave);
          }
        }
        if (sub != 0) return;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && !two) break;
        const int g = g0 + e;
        float a = acc[e];
        if (!ovf) a = (a - self_term[e]) * (c.mass * c.A);  // the particle met itself once (q = 1)
        if (bnd.is(g)) {  // a boundary particle reads as density 0, P/rho^2 = 0/0 (Bnd, sph_device.hpp)
          rho[g] = 0.0f;
          pterm[g] = __uint_as_float(0x7fc00000u);
          continue;
        }
        rho[g] = a;
        // (an isolated particle's own term must be a harmless 0 rather than 0/0: see k_density_tiled)
        const float pr = tait_eos<true>(c, a, c.eos_d0_grad);
        pterm[g] = a > 0.0f ? dsl_div<true>(pr, a * a) : 0.0f;
      }
    });
  }
}

// ---------------------------------------------------------------------------------
// [G] [V] X U (tiled): fused WCSPH force + integrate
// ---------------------------------------------------------------------------------
// OUT selects what happens to the swept sums:
//   kOutIntegrate  fused WCSPH step: F = reset/forces [+G] [+V] + ext, then Update (pout/vout)
//   kOutAddForce   the reference's stand-alone passes: forces[pout] += G-term (+ V-term)
//   kOutStore      pout = G-term (PCISPH: the gradient term is the same in every iteration)
//   kOutPci        both of those in ONE sweep (PCISPH step set-up): forces[pout] += V-term (+ cohesion), gout =
//                  G-term; each sum is formed exactly as in its own pass (the two passes cost 0.245 + 0.204 ms at
//                  4M particles, mostly per-tile and per-run work they share)
constexpr int kOutIntegrate = 0, kOutAddForce = 1, kOutStore = 2, kOutPci = 3;

// WANT_XS adds the build-defined XSPH and cohesion sums (BASELINE configs[4]) to the same sweep;
// with kOutAddForce the XSPH correction is stored through `vout` for the later Update.
// SLAB compiles in the multi-GPU logic (ownership, ghost tiles, split step); the single-domain
// instantiations carry none of it.
// EXACT = true (DSL_MATH_EXACT): the reference's own float32 operations in the reference's order (runs in
// z, y order, candidates ascending, one lane per target), IEEE sqrt and divide, bit for bit the oracle's
// Gradient / LaplacianForce / Update; same staging, same masks (written by the EXACT density sweep).
template <bool WANT_G, bool WANT_V, int OUT = kOutIntegrate, bool WANT_XS = false, bool SLAB = false, bool SHARE = true,
          bool EXACT = false, bool QUEUE = false>
// (Two 8-wave workgroups per CU need <= 128 VGPRs.  The headline instantiation gets there on its
// own and schedules best unconstrained; the others are held to 4 waves/SIMD.)
__global__ __launch_bounds__(kTBlock, (WANT_G && WANT_V && !WANT_XS && !SLAB && !SHARE && !EXACT && OUT == kOutIntegrate) ? 1 : 4) void k_force_integrate_tiled(
    DevConsts c, TileGrid tg, const int* __restrict__ desc_of, const int* __restrict__ n_tiles,
    const int* __restrict__ ghost_desc_of, const int* __restrict__ n_ghost_tiles, const int* __restrict__ desc,
    const int* __restrict__ cell_start, CSoa3 pin, CSoa3 vin, const float* __restrict__ rho,
    const float* __restrict__ pterm, CSoa3 fin, int forces_uniform, Soa3 pout, Soa3 vout, DevStats* stats,
    const unsigned int* __restrict__ nmask, int mstride, const int* __restrict__ share_stats, Bnd bnd, Soa3 gout,
    int* __restrict__ walk_ctr = nullptr) {
  static_assert(!(EXACT && SHARE), "the exact sums are sequential: one lane per target");
  if (share_stats != nullptr && share_wanted(share_stats) != SHARE) return;
  __shared__ TileMeta metas[2];
  __shared__ float4 A[kTCap];  // x,y,z,P/rho^2
  __shared__ float4 B[kTCap];  // vx,vy,vz,1/rho
  __shared__ int feed_slot[2];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  unsigned int vbits = 0u, fbits = 0u;  // max|v|, max|F| of this lane over all its tiles
  // SLAB: after the tiles with owned cell layers, the ghost-only tiles (second list, walked the same
  // XCD-chunked way so that their small cost is spread evenly): nothing is staged for those, their
  // targets are just marked for removal (a particle that float rounding puts on the owned side of
  // the plane after all takes the global-memory sweep).
  const int nphase = (SLAB && ghost_desc_of != nullptr) ? 2 : 1;
  for (int phase = 0; phase < nphase; ++phase) {
  const bool ghost_tile = phase == 1;
  static_assert(!(QUEUE && SLAB), "the tile queue walks list 0 of a single domain");
  typename TileSource<QUEUE>::type feed = TileSource<QUEUE>::make(ghost_tile ? ghost_desc_of : desc_of, ghost_tile ? *n_ghost_tiles : *n_tiles, walk_ctr, feed_slot);
  // The NEXT tile's table (k_tile_desc) travels, one dword per lane, under the current tile's staging and
  // is put into the other LDS copy behind it: a tile starts with ONE barrier and its table in place.
  int di = 0;
  bool have = feed.pop(di);
  if (phase == 1) sync_lds();  // (the first phase's last sweep may still be reading its table)
  if (have) tile_meta_store(metas[0], tile_meta_request(desc, di));
  for (int cur = 0; have; cur ^= 1) {
    TileMeta& m = metas[cur];
    DSL_STAMP(t0);
    sync_lds();  // the previous tile's sweep is over, this tile's table is visible
    have = feed.pop(di);
    int table_word = 0;
    if (have) table_word = tile_meta_request(desc, di);
    DSL_STAMP(t1);
    DSL_STAMP_ADD(0, t0, t1);
    const bool ovf = m.overflow != 0;
    const bool nolds = ovf || ghost_tile;  // owned targets of such a tile take the global-memory sweep
    // The first pass's per-target mask words (valid word, first run word) are requested BEFORE the
    // staging loads, so that their latency passes under the staging wait instead of in front of the sweep.
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(lane);
    unsigned int pre_word = 0u;
    if (!nolds && nmask != nullptr && tperm < m.tprefix[kTB * kTB]) {
      const int g0 = tile_target(m, tperm).g;
      pre_word = nmask[(size_t)(EXACT ? 0 : 4) * mstride + g0];  // the run visited first (FAST: the centre run, whatever the lane's mirroring)
    }
    DSL_STAMP(t1b);
    DSL_STAMP_ADD(10, t1, t1b);  // first-pass mask word requests
    // EXACT: the walk divides with exact_div (sph_device.hpp), which equals `/` bit for bit while every staged
    // coordinate is 0 or between 2^-20 and 2^20 in magnitude; a tile with any other value takes the global sweep
    bool unsafe = EXACT && !exact_div_ok(c.h);
    if (!nolds) {
      const bool values_ok = stage_rows<8, EXACT>(
          m,
          [&](int g, float4* o) {
            o[0] = load4u(pin.x + g);
            o[1] = load4u(pin.y + g);
            o[2] = load4u(pin.z + g);
            o[3] = (WANT_G || bnd.ids != nullptr) ? load4u(pterm + g) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (WANT_V || WANT_XS) {
              o[4] = load4u(vin.x + g);
              o[5] = load4u(vin.y + g);
              o[6] = load4u(vin.z + g);
              o[7] = load4u(rho + g);
            } else {
              o[4] = o[5] = o[6] = o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          },
          [&](int g, float* o) {
            o[0] = pin.x[g];
            o[1] = pin.y[g];
            o[2] = pin.z[g];
            o[3] = (WANT_G || bnd.ids != nullptr) ? pterm[g] : 0.f;  // (NaN marks a boundary particle)
            if constexpr (WANT_V || WANT_XS) {
              o[4] = vin.x[g];
              o[5] = vin.y[g];
              o[6] = vin.z[g];
              o[7] = rho[g];
            }
          },
          [&](int slot, const float* o, bool real) {
            float4 a = make_float4(kFar, kFar, kFar, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
            // (a particle whose position has gone NaN -- the reference produces such next to boundary
            // particles -- is no one's neighbour there (dist < h is false); as a pad record it is none here
            // either, also in the runs that are walked without a mask)
            if constexpr (EXACT) {
              // raw values.  The reference forms 1/rho_j pair by pair (sph_field.go:262: Scale(.., 1/jDensity)); the
              // quotient depends on j alone, so the very same IEEE division is done once per staged record instead
              // (the XSPH sum divides the mass by rho_j: that instantiation keeps rho itself)
              if (real) {
                a = make_float4(o[0], o[1], o[2], o[3]);
                if constexpr (WANT_V || WANT_XS) b = make_float4(o[4], o[5], o[6], WANT_XS ? o[7] : 1.0f / o[7]);
              }
            } else if (real && o[0] == o[0] && o[1] == o[1] && o[2] == o[2]) {
              a = make_float4(o[0], o[1], o[2], o[3]);
              // 1/rho = 0 for an isolated particle (rho = 0): it only ever meets itself.  A boundary
              // particle (rho = 0 as well, P/rho^2 = NaN) is divided by as the reference does: 1/0 = +inf
              if constexpr (WANT_V || WANT_XS)
                b = make_float4(o[4], o[5], o[6],
                                o[7] > 0.0f ? __builtin_amdgcn_rcpf(o[7]) : (o[3] != o[3] ? __builtin_inff() : 0.0f));
            }
            A[slot] = a;
            if constexpr (WANT_V || WANT_XS) B[slot] = b;
          });
      if constexpr (EXACT) unsafe |= !values_ok;
    }
    DSL_STAMP(t1c);
    if (have) tile_meta_store(metas[cur ^ 1], table_word);  // (requested before the staging loads: it has landed with them)
    bool slow = nolds;  // this tile's targets take the global-memory sweep
    if constexpr (EXACT) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (see sync_lds)
      slow = (__syncthreads_or(unsafe ? 1 : 0) != 0) || nolds;
    }
    else sync_lds();
    DSL_STAMP(t1d);
    DSL_STAMP_ADD(11, t1c, t1d);  // barrier behind the staging
    DSL_STAMP(t2);
    DSL_STAMP_ADD(1, t1, t2);
    const int ntarg = m.tprefix[kTB * kTB];
    // short passes are shared out as in k_density_tiled: k lanes per target, each walking every
    // k-th run; after the butterfly the group's first lane finishes the target
    // (nine-lane groups for remainders of 33 .. 112 targets, as in the pair sweep: measured 3 % SLOWER on the developed
    // flow -- 6240-6255 against 6420-6475 in one call, profiles/r04_force_nine_lane_groups_ab.jsonl: seven more sums to
    // combine over nine lanes in a kernel at its register limit -- and removed)
    for_each_target<SHARE>(ntarg, tid, tperm, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      const bool live = true;
      DSL_STAMP(t3);
      const TileTarget tt = tile_target(m, t);
      const int srow = tt.srow, g = tt.g;
      float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f, fx = 0.f, fy = 0.f, fz = 0.f;
      float xsx = 0.f, xsy = 0.f, xsz = 0.f;  // XSPH sum / correction
      float gfx = 0.f, gfy = 0.f, gfz = 0.f;  // kOutPci: the gradient term, kept apart from the force
      bool owned = false;
      float pti_staged = 0.f;
      if (live) {
        // the target's own record is in the staged tile: no second trip to global memory (the reads
        // are unconditional and the rare unstaged tile overrides them, so that the compiler keeps
        // LDS and global loads apart instead of merging them into flat loads)
        {
          const int own = slow ? 0 : tt.own;
          const float4 a = A[own];
          px = a.x;
          py = a.y;
          pz = a.z;
          pti_staged = a.w;
          if (!EXACT && px == kFar) px = py = pz = __uint_as_float(0x7fc00000u);  // staged as a pad record: its position is NaN
          if constexpr (WANT_V || WANT_XS) {
            const float4 b = B[own];
            vx = b.x;
            vy = b.y;
            vz = b.z;
          }
        }
        if (slow) {
          px = pin.x[g];
          py = pin.y[g];
          pz = pin.z[g];
        }
        if (slow || !(WANT_V || WANT_XS)) {
          vx = vin.x[g];
          vy = vin.y[g];
          vz = vin.z[g];
        }
        owned = !SLAB || OUT != kOutIntegrate || slab_owned(c, px, py, pz);
        if (bnd.is(g)) {  // never a target: carried over unchanged (fused step) or left alone
          if constexpr (OUT == kOutIntegrate) {
            if (!SHARED || sub == 0) {
              pout.x[g] = px;
              pout.y[g] = py;
              pout.z[g] = pz;
              vout.x[g] = vin.x[g];
              vout.y[g] = vin.y[g];
              vout.z[g] = vin.z[g];
            }
          }
          return;
        }
        if constexpr (SLAB && OUT == kOutIntegrate) {
          // split slab step: this launch integrates the band cell layers or the others, not both
          if (owned && c.split_part != 0 && slab_band_cell(c, slab_axis_cell(c, px, py, pz)) != (c.split_part == 1)) return;
        }
      }
      if (owned) {
        float gx = 0.f, gy = 0.f, gz = 0.f, lx_ = 0.f, ly_ = 0.f, lz_ = 0.f;
        float cohx = 0.f, cohy = 0.f, cohz = 0.f;  // cohesion sum
        if (!slow) {
          if constexpr (WANT_G || WANT_V || WANT_XS) {
            const float pti = WANT_G ? pti_staged : 0.f;
            const float ninvh = -c.inv_h;
            float lw_ = 0.f, xw_ = 0.f;
            const float ninvhh = -c.inv_hh;
            const int lx = tt.lx;
            const int pad_rec = m.row_lds[1] - kTPad;  // first pad record of staged row 0: (kFar, kFar, kFar, 0), (0, 0, 0, 0)
            // full per-pair arithmetic for candidate record j: `fetch` issues the LDS reads, `accum`
            // does the arithmetic, so that the walk below can have the NEXT records in flight while
            // it works on the current ones
            struct PairRec {
              float4 a, b;
              int idx;
            };
            auto fetch = [&](int j) {
              PairRec r;
              r.a = A[j];
              r.b = r.a;
              if constexpr (WANT_V || WANT_XS) r.b = B[j];
              r.idx = j;
              return r;
            };
            const int own_rec = tt.own;
            // EXACT: the loop bodies of SPHField.Gradient (sph_field.go:183-199), LaplacianForce (:259-266) and
            // the build-defined sums for one candidate, operation by operation as force_sweep<false> has them;
            // the particle itself, a pad record and a candidate at dist >= h add +0
            // (divisions: exact_div -- the divisor's reciprocal refinement shared by the three quotients dx, dy, dz / dist,
            // dist / h formed once for O1D and O2D, x^2 / h^2 with the constant's refinement hoisted; bit for bit `/` on
            // a tile that passed exact_div_ok.  188 -> ~150 VALU instructions per pair.)
            const ExactDivisor Dh = exact_divisor_uniform(c.h), Dhh = exact_divisor_uniform(c.hh);
            auto accum_exact = [&](const PairRec& rec) {
              const float4 a = rec.a, b = rec.b;
              const float dx = a.x - px, dy = a.y - py, dz = a.z - pz;  // dir = x_j - x_i (sph_field.go:189)
              const float dist = dsl_sqrt<false>(dist2<false>(dx, dy, dz));
              const bool in = rec.idx != own_rec && dist < c.h;
              const float q1 = 1.0f - exact_div(dist, Dh);  // 1 - x/h of O1D and O2D (std_kernel.go:54-71)
              if constexpr (WANT_XS) {
                float fw = 0.0f;  // kern_F<false> (std_kernel.go:33-39)
                if (!(dist >= c.h)) {
                  const float xx = dist * dist;
                  const float q = 1.0f - exact_div(xx, Dhh);
                  const float aq = c.A * q;
                  fw = aq * q;
                }
                const float ws = c.mass * fw;
                const float tsx = dx * ws, tsy = dy * ws, tsz = dz * ws;
                cohx = cohx + (in ? tsx : 0.0f);
                cohy = cohy + (in ? tsy : 0.0f);
                cohz = cohz + (in ? tsz : 0.0f);
                const float wx = (c.mass / b.w) * fw;
                const float txx = (b.x - vx) * wx, txy = (b.y - vy) * wx, txz = (b.z - vz) * wx;
                xsx = xsx + (in ? txx : 0.0f);
                xsy = xsy + (in ? txy : 0.0f);
                xsz = xsz + (in ? txz : 0.0f);
              }
              if constexpr (WANT_G) {
                float nx = 0.f, ny = 0.f, nz = 0.f;  // vector.go:322-331 Norm
                const ExactDivisor Dd = exact_divisor(dist);
                const float ex = exact_div(dx, Dd), ey = exact_div(dy, Dd), ez = exact_div(dz, Dd);
                if (dist != 0.0f) {
                  nx = ex;
                  ny = ey;
                  nz = ez;
                }
                const float bq = c.B * q1;  // kern_O1D<false>
                const float sgrad = -((dist >= c.h) ? 0.0f : bq * q1);  // std_kernel.go:74-76 Grad
                const float ggx = nx * sgrad, ggy = ny * sgrad, ggz = nz * sgrad;
                const float F = pti_staged + a.w;
                const float tx = ggx * F, ty = ggy * F, tz = ggz * F;
                gx = gx + (in ? tx : 0.0f);
                gy = gy + (in ? ty : 0.0f);
                gz = gz + (in ? tz : 0.0f);
              }
              if constexpr (WANT_V) {
                const float inv = WANT_XS ? 1.0f / b.w : b.w;
                const float ux = (b.x - vx) * inv, uy = (b.y - vy) * inv, uz = (b.z - vz) * inv;
                const float o2 = (dist > c.h) ? 0.0f : c.C * q1;  // kern_O2D<false>
                const float tx = ux * o2, ty = uy * o2, tz = uz * o2;
                if (c.visc_running_mass) {  // sph_field.go:265: (force + t) * m
                  const float sx = lx_ + tx, sy = ly_ + ty, sz = lz_ + tz;
                  lx_ = in ? sx * c.mass : lx_;
                  ly_ = in ? sy * c.mass : ly_;
                  lz_ = in ? sz * c.mass : lz_;
                } else {
                  const float mx = tx * c.mass, my = ty * c.mass, mz = tz * c.mass;
                  lx_ = lx_ + (in ? mx : 0.0f);
                  ly_ = ly_ + (in ? my : 0.0f);
                  lz_ = lz_ + (in ? mz : 0.0f);
                }
              }
            };
            auto accum_fast = [&](const PairRec& rec) {
              const float4 a = rec.a;
              const float dx = a.x - px, dyy = a.y - py, dzz = a.z - pz;
              float r2 = __builtin_fmaf(dzz, dzz, __builtin_fmaf(dyy, dyy, dx * dx));
              r2 = fmaxf(r2, 1.0e-30f);  // the particle itself: keeps rsq finite, all terms stay 0
              const float rinv = __builtin_amdgcn_rsqf(r2);
              const float dist = r2 * rinv;
              const float q = fma1_clamp01_uniform(dist, ninvh);
              if constexpr (WANT_G) {
                const float k = (q * q) * (pti + a.w) * rinv;
                gx = __builtin_fmaf(dx, k, gx);
                gy = __builtin_fmaf(dyy, k, gy);
                gz = __builtin_fmaf(dzz, k, gz);
              }
              const float4 b = rec.b;
              if constexpr (WANT_V) {
                // sum_j (v_j - v_i) w_j = sum_j v_j w_j - v_i sum_j w_j
                const float w = q * b.w;
                lx_ = __builtin_fmaf(b.x, w, lx_);
                ly_ = __builtin_fmaf(b.y, w, ly_);
                lz_ = __builtin_fmaf(b.z, w, lz_);
                lw_ += w;
              }
              if constexpr (WANT_XS) {
                // F(r)/A = (1 - r^2/h^2)^2 for both build-defined terms; the particle itself has d = 0
                // and v_j = v_i, so it contributes nothing
                const float q2 = fma1_clamp01_uniform(r2, ninvhh);
                const float fw = q2 * q2;
                cohx = __builtin_fmaf(dx, fw, cohx);
                cohy = __builtin_fmaf(dyy, fw, cohy);
                cohz = __builtin_fmaf(dzz, fw, cohz);
                const float wx = fw * b.w;
                xsx = __builtin_fmaf(b.x, wx, xsx);
                xsy = __builtin_fmaf(b.y, wx, xsy);
                xsz = __builtin_fmaf(b.z, wx, xsz);
                xw_ += wx;
              }
            };
            auto accum = [&](const PairRec& rec) {
              if constexpr (EXACT) accum_exact(rec);
              else accum_fast(rec);
            };
            const bool first_pass = !SHARED && t == tperm;  // (its mask words were requested before the staging)
            // Which runs have mask words?  Those of at most 64 candidates -- a rule on the run's length, which this kernel
            // reads off the same tile table the density sweep used, so the "valid" word the density kernels used to
            // write per particle (and this kernel to fetch ahead of everything else) is gone: 8 B per particle of
            // traffic and one load in the prologue; adding 8 B of mask traffic instead had cost the force kernel 3.7 %
            // and the density kernel 1.2 % (profiles/README.md, r03).
            const bool have_masks = nmask != nullptr;
            auto run_masked = [&](int j, int je) { return have_masks && je - j <= 64; };
            DSL_STAMP(t4);
            DSL_STAMP_ADD(2, t3, t4);
            // Walk the in-range bits of one run (row rr of the staged tile, the 3 cells around the
            // target's); bit (K-1-k) <-> candidate k of the run, K = run length rounded up to the
            // density sweep's unroll of 4.  A run without a mask (more than 32 candidates, or no masks
            // at all) is walked in chunks of 32 with every bit set.
            // `first_word`: the run's first mask word, loaded by the caller one run ahead so that its
            // global-memory latency passes under the previous run's walk
            // (the run's LDS record range [j, je) likewise comes from the caller)
            auto run_bounds_of_row = [&](int rr, int& j, int& je) { tile_run(m, rr, lx, j, je); };
            auto run_bounds = [&](int ri, int& j, int& je) {
              run_bounds_of_row(srow + (ri / 3 - 1) * kTH + (ri % 3 - 1), j, je);
            };
            // `second_word`: the mask of candidates 32-63, requested by the caller together with the first
            // word.  (Once the lattice has melted ~10 % of the runs are that long, so in nearly every run
            // of a wave SOME lane needs its second word: fetched here, inside the walk, that was one exposed
            // HBM latency per run and wave.)
            // the in-range bits of one mask word (or of a chunk of up to 32 unmasked candidates): bit b <-> record top - b
            // the in-range bits of one mask word (or of a chunk of up to 32 unmasked candidates): bit b <-> record top - b
            auto walk_bits = [&](unsigned int mm, const int top) {
#ifdef DSL_DIAG_NO_SWEEP  // timing-only build: measures the per-tile fixed cost (set-up + staging + epilogue)
              mm = 0u;
#endif
              // Every lane stays in the walk until the last bit of the wave is done: a lane that has
              // run out of bits works on a far-away pad record, whose terms are all exactly zero.  That
              // makes the loop wave-uniform (no exec bookkeeping) and lets it be software-pipelined:
              // kPairsPerTrip pairs per trip, the next trip's LDS reads issued before the current
              // trip's arithmetic.  (The old loop waited out one LDS latency plus one dependent chain
              // of ~20 VALU instructions per pair: 5.9 clocks per instruction at 4 waves per SIMD.)
              auto take = [&]() {
                const bool has = mm != 0u;
                if constexpr (EXACT) {  // highest bit = earliest candidate first: the reference's order
                  const int b = 31 - __builtin_clz(mm | 1u);
                  const int idx = has ? top - b : pad_rec;
                  mm = has ? (mm & ~(1u << b)) : 0u;
                  return idx;
                } else {
                  const int idx = has ? top - __builtin_ctz(mm) : pad_rec;
                  mm &= mm - 1u;
                  return idx;
                }
              };
              if (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
                // (one exit, both pairs of a trip unconditional: with an exit between a fetch and its
                // use the compiler sinks the LDS reads below the branch and the prefetch is gone; the
                // price is one all-pad pair when the wave's longest lane has an odd number of bits)
                PairRec p = fetch(take());
                bool more;
                // (two pairs per trip cost one all-pad pair whenever the wave's longest lane has an odd number
                // of bits, ~0.5 per run; one pair per trip avoids that and pays it back in register copies --
                // 35 instead of 31 VALU instructions per pair: measured equal, 0.940 vs 0.937 ms)
                do {
                  const PairRec q = fetch(take());
                  accum(p);
                  more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
                  p = fetch(take());
                  accum(q);
                } while (more);
              }
            };
            // A run: its first 32 candidates in straight-line code -- on a lattice, and wherever no lane of the
            // wave has a run longer than that, this is all there is -- then, lane by lane, the chunks behind them
            // (the second mask word, or chunks of 32 with every bit set for a run without a mask).
            auto walk_run = [&](int ri, const int j, const int je, unsigned int first_word, unsigned int second_word) {
              const bool has_mask = run_masked(j, je);
              // (a run without a mask -- longer than 64 candidates, or no masks at all -- is rare: the selects
              // between a mask word and "every bit set" are only compiled into the path some lane needs them on)
              // (not in the XSPH / cohesion instantiations: they are at the register limit, a second copy of the walk spills)
              const bool all_masked = !WANT_XS && __builtin_amdgcn_ballot_w64(!has_mask) == 0ull;
              auto chunk = [&](int from, unsigned int word) {
                const int clen = min(je - from, 32);
                // (a mask's bit order counts from the run length rounded up to the density sweep's unroll of 4)
                if (all_masked) {
                  walk_bits(word, from + ((clen + 3) & ~3) - 1);
                } else {
                  const unsigned int all = clen >= 32 ? ~0u : ((1u << clen) - 1u);
                  walk_bits(has_mask ? word : all, from + (has_mask ? ((clen + 3) & ~3) : clen) - 1);
                }
              };
              // (r03, measured and removed: walking a long run's second word in the SAME loop as the first -- a lane moving
              // on to it when its first word is used up -- instead of in a loop of its own behind it: developed-flow force
              // 1.42 ms with, 1.40 without.  The second loops are not where the developed flow's +54 % comes from.)
              chunk(j, first_word);
              if (__builtin_amdgcn_ballot_w64(je - j > 32) != 0ull)
                for (int from = j + 32; from < je; from += 32) chunk(from, second_word);
            };
            if constexpr (!SHARED) {
              // The wave spends, on every run, as many iterations as its busiest lane has neighbours
              // there -- and which runs are busy depends on where in its cell a particle sits (near the
              // +y face: the dy = +1 rows).  Each lane therefore visits the 9 runs in an order mirrored
              // by its own half-cell in y and z, so that at every step all lanes are on a run of similar
              // weight ("towards my corner" first).  Sums are order-independent up to rounding.
              const float fy = (py - c.gmin[1]) * c.inv_cell, fz = (pz - c.gmin[2]) * c.inv_cell;
              const int sy = (fy - floorf(fy)) >= 0.5f ? 1 : -1, sz = (fz - floorf(fz)) >= 0.5f ? 1 : -1;
              // (the centre run, the same for every mirroring, comes first: its mask word can be
              // requested before the position is known, and its walk, the longest, covers the next word's latency)
              // (the loop over the 9 steps is unrolled: a step's (dz, dy) before mirroring are constants, the run
              // index and its staged row one or two operations each instead of a division by 3 per lane)
              auto run_of = [&](int s) {
                if constexpr (EXACT) return s;  // z, y order: the reference's
                const int ms = s == 0 ? 4 : (s <= 4 ? s - 1 : s);
                return ((ms / 3 - 1) * sz + 1) * 3 + ((ms % 3 - 1) * sy + 1);
              };
              auto row_of = [&](int s) {
                if constexpr (EXACT) return srow + (s / 3 - 1) * kTH + (s % 3 - 1);
                const int ms = s == 0 ? 4 : (s <= 4 ? s - 1 : s);
                return srow + (ms / 3 - 1) * sz * kTH + (ms % 3 - 1) * sy;
              };
              // the step's mask word: plane run_of(s) of the particle's column.  The plane offset is the centre
              // run's plus or minus the (64-bit) strides of one run in z and in y, mirrored like the run itself:
              // one or two additions per step instead of a 64-bit multiply per lane
              const unsigned int* mcol = nmask != nullptr ? nmask + g : nullptr;
              const long long my = (long long)(EXACT ? 1 : sy) * mstride, mz = (long long)(EXACT ? 3 : 3 * sz) * mstride;
              auto word_of = [&](int s, int plane0) -> const unsigned int* {
                const int ms = EXACT ? s : (s == 0 ? 4 : (s <= 4 ? s - 1 : s));
                return mcol + (long long)(plane0 + 4) * mstride + (ms / 3 - 1) * mz + (ms % 3 - 1) * my;
              };
              int rn = run_of(0);
              // The runs' first mask words are requested kMaskAhead steps ahead of their walk (the words of steps
              // 0 .. kMaskAhead at once, then one per step): a word comes from HBM -- the masks of 16M particles are
              // 640 MB -- and a single run's walk is over long before that round trip is.
#ifndef DSL_FORCE_PF
#define DSL_FORCE_PF 8
#endif
              constexpr int kMaskAhead = DSL_FORCE_PF;
              unsigned int wq[9];
#pragma unroll
              for (int s = 0; s < 9; ++s) wq[s] = 0u;
              wq[0] = first_pass ? pre_word : (nmask != nullptr ? *word_of(0, 0) : 0u);
#ifndef DSL_FORCE_TRIPLE
#define DSL_FORCE_TRIPLE 1
#endif
              // (not the slab instantiations: they have no lattice twin to fall back on, and on a lattice the queue costs
              // a middle rank of 16M / 8 one per cent -- 0.455 against 0.450 ms per step)
              // (EXACT: a lane still meets its candidates run by run, earliest first -- the reference's order, so the same
              // bits -- and at ~150 VALU instructions per pair the queue is cheap: the lattice gains there too)
#ifndef DSL_FORCE_TRIPLE_EXACT
#define DSL_FORCE_TRIPLE_EXACT 1
#endif
              // (the FAST lattice instantiation, measured with it: force 0.86 -> 1.31 ms -- that one is compiled without a
              // register cap and the queue pushes it past 128 VGPRs, one workgroup per CU)
              constexpr bool kTripleRuns = DSL_FORCE_TRIPLE && (SHARE || (EXACT && DSL_FORCE_TRIPLE_EXACT)) && !SLAB && !WANT_XS && kMaskAhead >= 8;
              // (three runs per loop, below: the last three words are requested while the first three runs are walked)
              constexpr int kPreload = kTripleRuns ? 5 : kMaskAhead;
#pragma unroll
              for (int s = 1; s <= kPreload && s < 9; ++s)
                if (have_masks) wq[s] = *word_of(s, 0);
              int jn, jen;
              run_bounds_of_row(row_of(0), jn, jen);
              auto second_of = [&](int s, int ri, int j, int je) {
                return (je - j > 32 && run_masked(j, je)) ? *word_of(s, kMaskHigh) : 0u;
              };
              unsigned int ahead2 = second_of(0, rn, jn, jen);
              // (r03, measured and removed: a two-run WINDOW for the developed flow -- the wave walks run s until every lane
              // is through with it, but a lane without bits left in run s takes bits of its run s+1 meanwhile, second words
              // as steps 9..17 of the same window -- because there the busiest lane of EACH run sets the pace: 7.7e8 VALU
              // instructions per launch against 4.4e8 on the lattice, ~80 pair iterations per target for ~36 neighbours
              // (profiles/r03_developed_pmc.txt).  Two cursors cost five more VALU per pair and the window moves through
              // 18 steps: force 1.56 ms with it, 1.40 without.)
              // Developed flow (the pass-sharing instantiation): THREE runs per wave-synchronised loop.  The wave spends on
              // every loop as many trips as its busiest lane has bits -- ~9 per run for a mean of 4 once the lattice has
              // melted, 80 pair trips per target for 36 neighbours -- and the busiest lane of a z-plane's three runs together
              // is much closer to the mean than three busiest lanes one after the other; six loop set-ups (and their odd
              // all-pad pairs) per target go as well.  A lane moves to its next non-empty word inside the loop (a three-
              // deep queue of (word, top record): six more VALU per pair).  A lattice has nothing to gain (41 trips for 33
              // neighbours) and keeps the loop per run.
              if constexpr (kTripleRuns) {
                auto walk3 = [&](unsigned int mm, int top, unsigned int n1, int tp1, unsigned int n2, int tp2) {
                  // empty words to the back, so that one shift per exhausted word is enough
                  if (n1 == 0u) {
                    n1 = n2;
                    tp1 = tp2;
                    n2 = 0u;
                  }
                  if (mm == 0u) {
                    mm = n1;
                    top = tp1;
                    n1 = n2;
                    tp1 = tp2;
                    n2 = 0u;
                  }
                  auto take = [&]() {
                    const bool has = mm != 0u;
                    int idx;
                    if constexpr (EXACT) {  // highest bit = earliest candidate first: the reference's order
                      const int b = 31 - __builtin_clz(mm | 1u);
                      idx = has ? top - b : pad_rec;
                      mm = has ? (mm & ~(1u << b)) : 0u;
                    } else {
                      idx = has ? top - __builtin_ctz(mm) : pad_rec;
                      mm &= mm - 1u;
                    }
                    if (mm == 0u) {
                      mm = n1;
                      top = tp1;
                      n1 = n2;
                      tp1 = tp2;
                      n2 = 0u;
                    }
                    return idx;
                  };
                  if (__builtin_amdgcn_ballot_w64(mm != 0u) != 0ull) {
                    PairRec pr = fetch(take());
                    bool more;
                    do {
                      const PairRec q = fetch(take());
                      accum(pr);
                      more = __builtin_amdgcn_ballot_w64(mm != 0u) != 0ull;
                      pr = fetch(take());
                      accum(q);
                    } while (more);
                  }
                };
#pragma unroll
                for (int g3 = 0; g3 < 3; ++g3) {
                  int jj[3], jje[3], rr[3];
                  unsigned int ww[3], w2[3];
                  bool unmasked = false, longrun = false;
#pragma unroll
                  for (int u = 0; u < 3; ++u) {
                    const int s = 3 * g3 + u;
                    rr[u] = run_of(s);
                    run_bounds_of_row(row_of(s), jj[u], jje[u]);
                    ww[u] = wq[s];
                    if (g3 == 0 && have_masks) wq[s + 6] = *word_of(s + 6, 0);
                    w2[u] = second_of(s, rr[u], jj[u], jje[u]);  // (requested here, used behind the first words' walk)
                    unmasked |= !run_masked(jj[u], jje[u]);
                    longrun |= jje[u] - jj[u] > 32;
                  }
                  // (EXACT: a run's candidates 32-63 have to follow its first 32 directly -- the reference's order -- so a
                  // wave with such a run somewhere in this z-plane takes the loop per run)
                  if (__builtin_amdgcn_ballot_w64(unmasked || (EXACT && longrun)) == 0ull) {
                    {
                      int tp[3];
#pragma unroll
                      for (int u = 0; u < 3; ++u) tp[u] = jj[u] + ((min(jje[u] - jj[u], 32) + 3) & ~3) - 1;
                      walk3(ww[0], tp[0], ww[1], tp[1], ww[2], tp[2]);
                    }
                    if (__builtin_amdgcn_ballot_w64(longrun) != 0ull) {  // candidates 32-63 of the three runs
                      // (the runs' bounds are read again rather than kept alive through the walk above: registers)
                      int tp[3];
#pragma unroll
                      for (int u = 0; u < 3; ++u) {
                        int j2, je2;
                        run_bounds_of_row(row_of(3 * g3 + u), j2, je2);
                        tp[u] = j2 + 32 + ((min(je2 - j2 - 32, 32) + 3) & ~3) - 1;
                      }
                      walk3(w2[0], tp[0], w2[1], tp[1], w2[2], tp[2]);
                    }
                  } else {  // (a run without a mask -- longer than 64 candidates -- somewhere in the wave: run by run)
#pragma unroll 1
                    for (int u = 0; u < 3; ++u) {
                      const int uu = u;
                      walk_run(uu == 0 ? rr[0] : (uu == 1 ? rr[1] : rr[2]), uu == 0 ? jj[0] : (uu == 1 ? jj[1] : jj[2]),
                               uu == 0 ? jje[0] : (uu == 1 ? jje[1] : jje[2]), uu == 0 ? ww[0] : (uu == 1 ? ww[1] : ww[2]),
                               uu == 0 ? w2[0] : (uu == 1 ? w2[1] : w2[2]));
                    }
                  }
                }
              } else {
#pragma unroll
              for (int s = 0; s < 9; ++s) {
                const unsigned int word = wq[s], word2 = ahead2;
                const int j = jn, je = jen, ri = rn;
                if (s + kMaskAhead + 1 < 9)
                  if (have_masks) wq[s + kMaskAhead + 1] = *word_of(s + kMaskAhead + 1, 0);
                if (s < 8) {
                  rn = run_of(s + 1);
                  run_bounds_of_row(row_of(s + 1), jn, jen);
                  ahead2 = second_of(s + 1, rn, jn, jen);
                }
                walk_run(ri, j, je, word, word2);
              }
              }
            } else {
#pragma unroll 1
              for (int ri = sub; ri < 9; ri += k) {
                int j, je;
                run_bounds(ri, j, je);
                const bool masked = run_masked(j, je);
                walk_run(ri, j, je, masked ? nmask[(size_t)ri * mstride + g] : 0u,
                         (masked && je - j > 32) ? nmask[(size_t)(kMaskHigh + ri) * mstride + g] : 0u);
              }
            }
            DSL_STAMP(t5);
            DSL_STAMP_ADD(3, t4, t5);
            if constexpr (SHARED)
            for (int o = 1; o < k; o <<= 1) {  // the lanes of a group are active together
              if constexpr (WANT_G) {
                gx += __shfl_xor(gx, o, kWave);
                gy += __shfl_xor(gy, o, kWave);
                gz += __shfl_xor(gz, o, kWave);
              }
              if constexpr (WANT_V) {
                lx_ += __shfl_xor(lx_, o, kWave);
                ly_ += __shfl_xor(ly_, o, kWave);
                lz_ += __shfl_xor(lz_, o, kWave);
                lw_ += __shfl_xor(lw_, o, kWave);
              }
              if constexpr (WANT_XS) {
                cohx += __shfl_xor(cohx, o, kWave);
                cohy += __shfl_xor(cohy, o, kWave);
                cohz += __shfl_xor(cohz, o, kWave);
                xsx += __shfl_xor(xsx, o, kWave);
                xsy += __shfl_xor(xsy, o, kWave);
                xsz += __shfl_xor(xsz, o, kWave);
                xw_ += __shfl_xor(xw_, o, kWave);
              }
            }
            if constexpr (!EXACT) {
            if constexpr (WANT_V) {
              lx_ = __builtin_fmaf(-vx, lw_, lx_);
              ly_ = __builtin_fmaf(-vy, lw_, ly_);
              lz_ = __builtin_fmaf(-vz, lw_, lz_);
            }
            if constexpr (WANT_XS) {
              const float ma = c.mass * c.A;
              cohx *= ma;
              cohy *= ma;
              cohz *= ma;
              xsx = __builtin_fmaf(-vx, xw_, xsx) * ma;
              xsy = __builtin_fmaf(-vy, xw_, xsy) * ma;
              xsz = __builtin_fmaf(-vz, xw_, xsz) * ma;
            }
            // constant factors taken out of the sums: -O1D = -B q^2, O2D = C q, times m
            const float sg = -c.B;
            gx *= sg;
            gy *= sg;
            gz *= sg;
            const float sv = c.C * c.mass;
            lx_ *= sv;
            ly_ *= sv;
            lz_ *= sv;
            }  // !EXACT
          }
        } else {
          float accG[3] = {0.f, 0.f, 0.f}, accV[3] = {0.f, 0.f, 0.f}, accX[3] = {0.f, 0.f, 0.f},
                accS[3] = {0.f, 0.f, 0.f};
          if constexpr (WANT_G || WANT_V || WANT_XS)
            force_sweep<!EXACT, WANT_G, WANT_V>(c, grid_neigh(cell_start), g, pin, vin, rho, pterm, accG, accV,
                                              WANT_XS ? accX : nullptr, WANT_XS ? accS : nullptr);
          cohx = accS[0];
          cohy = accS[1];
          cohz = accS[2];
          xsx = accX[0];
          xsy = accX[1];
          xsz = accX[2];
          gx = accG[0];
          gy = accG[1];
          gz = accG[2];
          lx_ = accV[0];
          ly_ = accV[1];
          lz_ = accV[2];
        }
        if constexpr (OUT == kOutIntegrate) {
          fx = c.reset[0];
          fy = c.reset[1];
          fz = c.reset[2];
          if (!forces_uniform) {
            fx = fin.x[g];
            fy = fin.y[g];
            fz = fin.z[g];
          }
        }
        if constexpr (EXACT) {  // the roundings of k_force_integrate<false> (fluid.go:164-172,146-152)
          if constexpr (WANT_G) {
            const float dm = rho[g] * c.mass;
            const float tx = gx * dm, ty = gy * dm, tz = gz * dm;
            const float sx = tx * c.pressure_sign, sy = ty * c.pressure_sign, sz = tz * c.pressure_sign;
            if constexpr (OUT == kOutPci) {
              gfx = 0.0f + sx;
              gfy = 0.0f + sy;
              gfz = 0.0f + sz;
            } else {
              fx += sx;
              fy += sy;
              fz += sz;
            }
          }
          if constexpr (WANT_V) {
            const float tx = lx_ * c.mu, ty = ly_ * c.mu, tz = lz_ * c.mu;
            fx += tx;
            fy += ty;
            fz += tz;
          }
          if constexpr (WANT_XS) {
            const float sx = cohx * c.st_kappa, sy = cohy * c.st_kappa, sz = cohz * c.st_kappa;
            fx += sx;
            fy += sy;
            fz += sz;
            xsx *= c.xsph_eps;
            xsy *= c.xsph_eps;
            xsz *= c.xsph_eps;
          }
        } else {
        if constexpr (WANT_G) {
          const float dm = rho[g] * c.mass * c.pressure_sign;
          if constexpr (OUT == kOutPci) {
            gfx = __builtin_fmaf(gx, dm, 0.0f);
            gfy = __builtin_fmaf(gy, dm, 0.0f);
            gfz = __builtin_fmaf(gz, dm, 0.0f);
          } else {
            fx = __builtin_fmaf(gx, dm, fx);
            fy = __builtin_fmaf(gy, dm, fy);
            fz = __builtin_fmaf(gz, dm, fz);
          }
        }
        if constexpr (WANT_V) {
          fx = __builtin_fmaf(lx_, c.mu, fx);
          fy = __builtin_fmaf(ly_, c.mu, fy);
          fz = __builtin_fmaf(lz_, c.mu, fz);
        }
        if constexpr (WANT_XS) {
          fx = __builtin_fmaf(cohx, c.st_kappa, fx);
          fy = __builtin_fmaf(cohy, c.st_kappa, fy);
          fz = __builtin_fmaf(cohz, c.st_kappa, fz);
          xsx *= c.xsph_eps;
          xsy *= c.xsph_eps;
          xsz *= c.xsph_eps;
        }
        }
        if constexpr (OUT == kOutIntegrate) {
          fx += c.ext[0];
          fy += c.ext[1];
          fz += c.ext[2];
        }
      } else {
        vx = vy = vz = 0.f;  // not integrated: keep ghosts and idle lanes out of the counters
      }
      if constexpr (SHARED) {
        if (sub != 0) return;  // the group's first lane finishes the target
      }
      if constexpr (OUT == kOutAddForce || OUT == kOutPci) {
        if (OUT == kOutPci && forces_uniform) {
          // (PCISPH step set-up with the forces still at their reset value, nowhere materialised: this IS the fill --
          // reset + f is the one addition `+=` would have done on the filled array)
          pout.x[g] = c.reset[0] + fx;
          pout.y[g] = c.reset[1] + fy;
          pout.z[g] = c.reset[2] + fz;
        } else {
          pout.x[g] += fx;
          pout.y[g] += fy;
          pout.z[g] += fz;
        }
        if constexpr (WANT_XS) {  // XSPH correction for the later Update
          vout.x[g] = xsx;
          vout.y[g] = xsy;
          vout.z[g] = xsz;
        }
        if constexpr (OUT == kOutPci) {
          gout.x[g] = gfx;
          gout.y[g] = gfy;
          gout.z[g] = gfz;
        }
        return;
      }
      if constexpr (OUT == kOutStore) {
        pout.x[g] = fx;
        pout.y[g] = fy;
        pout.z[g] = fz;
        return;
      }
      float npx = px, npy = py, npz = pz, nvx = vx, nvy = vy, nvz = vz;
      integrate_core(c, fx, fy, fz, npx, npy, npz, nvx, nvy, nvz, vbits, fbits, xsx, xsy, xsz);
      if (owned) {
        // split slab step: an interior-tile particle must not end up inside a band that has
        // already been packed (chk_* are -inf/+inf unless this is the interior launch)
        if constexpr (SLAB) {
          const float pa = c.slab_axis == 0 ? npx : (c.slab_axis == 1 ? npy : npz);
          if (pa < c.chk_lo || pa >= c.chk_hi) stats->band_missed = 1;
        }
        pout.x[g] = npx;
        pout.y[g] = npy;
        pout.z[g] = npz;
        vout.x[g] = nvx;
        vout.y[g] = nvy;
        vout.z[g] = nvz;
      } else if (live) {
        const float qnan = __uint_as_float(0x7fc00000u);  // ghost: dropped at the next neighbour build
        pout.x[g] = qnan;
        pout.y[g] = qnan;
        pout.z[g] = qnan;
        vout.x[g] = vin.x[g];
        vout.y[g] = vin.y[g];
        vout.z[g] = vin.z[g];
      }
    });
  }
  feed.finish();
  }  // phase
  if constexpr (OUT == kOutIntegrate) {
    wave_atomic_max(&stats->max_vel_bits, vbits);
    wave_atomic_max(&stats->max_f_bits, fbits);
  }
}

// ---------------------------------------------------------------------------------
// One PCISPH correction iteration up to its convergence check (pcisph_darwin.go:52-94), per target:
// predict (_vel += F/m dt, _pos += _vel dt, :57-73; the operations of k_pci_predict), DF (tiled):
// SPHField.DensityF at the PREDICTED position (sph_field.go:137-152) + pressure accumulate + max error
// (:76-92), then GradientPressureForce's cached term F += G (:93).  Every piece reads only the target's own
// state and its neighbours' CURRENT positions, so the three passes of an iteration are one launch (they used to
// be three: 0.042 + 0.192 + 0.026 ms at 4M particles).  Candidates are the current positions of the staged
// tile; a target whose predicted position has left the tile interior (the reference never re-synchronises
// the predictor, so it may) falls back to the global-memory sweep.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kTBlock) void k_pci_density_tiled(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                              const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                              const int* __restrict__ cell_start, Bnd bnd, CSoa3 p, Soa3 pp,
                                                              Soa3 pv, CSoa3 gterm, Soa3 frc, float* __restrict__ press,
                                                              unsigned int* __restrict__ drift, DevStats* stats) {
  if (stats->pci_done) return;
  // the LDS image is double-buffered exactly as in k_density_tiled: the next tile's loads are issued behind the
  // barrier that starts this tile's sweep, committed to the other image by every wave as soon as its own sweep is
  // over; one barrier per tile (there were three, and the table was fetched in the open)
  __shared__ TileMeta metas[3];
  __shared__ float4 Abuf[2][kTCap];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  unsigned int ebits = 0u;
  unsigned int n_left = 0u, n_all = 0u;  // queries that have left their particle's tile / all (k_pci_predict_bin)
  auto load4 = [&](int g, float4* o) {
    o[0] = load4u(p.x + g);
    o[1] = load4u(p.y + g);
    o[2] = load4u(p.z + g);
  };
  auto load1 = [&](int g, float* o) {
    o[0] = p.x[g];
    o[1] = p.y[g];
    o[2] = p.z[g];
  };
  auto commit = [&](const TileMeta& mt, const StageRegs<3>& sr, float4* img) {
    const float ox = __int_as_float(mt.centre[0]), oy = __int_as_float(mt.centre[1]), oz = __int_as_float(mt.centre[2]);
    stage_commit<3>(mt, sr, load1, [&](int slot, const float* o, bool real) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
      if (real) {  // records as in k_density_tiled: tile-relative x, y, z and w = -|x|^2/h^2
        const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
        v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
      }
      img[slot] = v;
    });
  };
  auto sweep = [&](const TileMeta& m, const float4* __restrict__ A) {
    const int tile = m.tile;
    const bool ovf = m.overflow != 0;
    const int tx = tile % tg.tnx, ty = (tile / tg.tnx) % tg.tny, tz = tile / (tg.tnx * tg.tny);
    const float ox = __int_as_float(m.centre[0]), oy = __int_as_float(m.centre[1]), oz = __int_as_float(m.centre[2]);
    const int ntarg = m.tprefix[kTB * kTB];
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(lane);
    for (int t = tperm; t < ntarg; t += kTBlock) {
      const int g = tile_target(m, t).g;
      // slab mode: a ghost's predicted density is meaningless (half a neighbourhood, no predictor
      // state) and must not decide the iteration's error
      // (a ghost is not predicted either: its predictor state travels with the particle from its owner)
      if (!slab_owned(c, p.x[g], p.y[g], p.z[g]) || bnd.is(g)) continue;
      const float fx = frc.x[g], fy = frc.y[g], fz = frc.z[g];
      const float gx = gterm.x[g], gy = gterm.y[g], gz = gterm.z[g];
      const float ax = fx * c.inv_mass, ay = fy * c.inv_mass, az = fz * c.inv_mass;
      const float dvx = ax * c.dt, dvy = ay * c.dt, dvz = az * c.dt;
      const float tvx = pv.x[g] + dvx, tvy = pv.y[g] + dvy, tvz = pv.z[g] + dvz;
      const float dpx = tvx * c.dt, dpy = tvy * c.dt, dpz = tvz * c.dt;
      const float qx = pp.x[g] + dpx, qy = pp.y[g] + dpy, qz = pp.z[g] + dpz;
      pp.x[g] = qx;
      pp.y[g] = qy;
      pp.z[g] = qz;
      pv.x[g] = tvx;
      pv.y[g] = tvy;
      pv.z[g] = tvz;
      frc.x[g] = fx + gx;
      frc.y[g] = fy + gy;
      frc.z[g] = fz + gz;
      // tile-local cell of the predicted position; the LDS image covers it and its 26
      // neighbours only while it stays inside the tile interior (1..4 per axis)
      const int lx = cell_coord(qx, c.gmin[0], c.inv_cell, c.dims[0]) - (tx * kTB - 1);
      const int ly = cell_coord(qy, c.gmin[1], c.inv_cell, c.dims[1]) - (ty * kTB - 1);
      const int lz = cell_coord(qz, c.gmin[2], c.inv_cell, c.dims[2]) - (tz * kTB - 1);
      const bool inside = lx >= 1 && lx <= kTB && ly >= 1 && ly <= kTB && lz >= 1 && lz <= kTB;
      n_all += 1u;
      n_left += inside ? 0u : 1u;
      if (pci_query_escaped(c, qx, qy, qz)) stats->pci_escaped = 1;
      float density;
      if (!ovf && inside) {
        const float rx = qx - ox, ry = qy - oy, rz = qz - oz;
        const float two_hh = 2.0f * c.inv_hh;
        const float sx = two_hh * rx, sy = two_hh * ry, sz = two_hh * rz;
        const float a0 = 1.0f - c.inv_hh * __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
        float acc = 0.f, acc1 = 0.f;
        const int qrow = lz * kTH + ly;
        // q = clamp(1 - r^2/h^2) = clamp(a0 + w_j + (2/h^2) q.x_j), as in k_density_tiled
        auto test4 = [&](int jj) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 cnd = A[jj + u];
            const float q = fma_clamp01(cnd.z, sz, __builtin_fmaf(cnd.y, sy, __builtin_fmaf(cnd.x, sx, cnd.w + a0)));
            if (u & 1) acc1 = __builtin_fmaf(q, q, acc1);
            else acc = __builtin_fmaf(q, q, acc);
          }
        };
#pragma unroll 1
        for (int dz = -kTH; dz <= kTH; dz += kTH) {
#pragma unroll 1
          for (int dy = -1; dy <= 1; ++dy) {
            int j, je;
            tile_run(m, qrow + dz + dy, lx, j, je);
            for (; j + 4 < je; j += 8) {
              test4(j);
              test4(j + 4);
            }
            if (j < je) test4(j);
          }
        }
        density = __builtin_fmaf(acc + acc1, c.mass * c.A, c.W0);  // starts at W0, self included
      } else {
        density = c.W0;
        for_each_grid_candidate(c, cell_start, qx, qy, qz, [&](int j) {
          const float dx = qx - p.x[j], dy = qy - p.y[j], dz = qz - p.z[j];
          const float r2 = dist2<true>(dx, dy, dz);
          if (r2 < c.hh) {
            const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
            density = __builtin_fmaf(c.mass * c.A, q * q, density);
          }
        });
      }
      const float density_error = density - c.ref_density;
      const float abs_err = density_error * __builtin_amdgcn_rcpf(c.ref_density);
      press[g] += density_error * c.delta;
      const unsigned int eb = nonneg_bits(abs_err);
      ebits = eb > ebits ? eb : ebits;
    }
  };
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) {
    tile_meta_store(metas[0], tile_meta_request(desc, di));
    bool have_next = feed.pop(di);
    int table_word = have_next ? tile_meta_request(desc, di) : 0;
    sync_lds();  // the first tile's table is visible
    {
      StageRegs<3> sr;
      if (!metas[0].overflow) {
        stage_issue<3>(metas[0], load4, sr);
        commit(metas[0], sr, Abuf[0]);
      }
    }
    if (have_next) tile_meta_store(metas[1], table_word);
    int mc = 0, mn = 1, mnn = 2;  // tables of this tile, the next one, the one after
    for (int cur = 0; have; cur ^= 1) {
      sync_lds();  // image `cur` is complete, the next tile's table visible, image `cur ^ 1` and table `mnn` free
      StageRegs<3> sr;
      bool have_nn = false, stage_next = false;
      table_word = 0;
      if (have_next) {
        stage_next = metas[mn].overflow == 0;
        if (stage_next) stage_issue<3>(metas[mn], load4, sr);
        have_nn = feed.pop(di);
        if (have_nn) table_word = tile_meta_request(desc, di);
      }
      sweep(metas[mc], Abuf[cur]);
      if (stage_next) commit(metas[mn], sr, Abuf[cur ^ 1]);
      if (have_nn) tile_meta_store(metas[mnn], table_word);
      have = have_next;
      have_next = have_nn;
      const int t = mc;
      mc = mn;
      mn = mnn;
      mnn = t;
    }
  }
  wave_atomic_max(&stats->pci_cur_err_bits, ebits);
  pci_drift_add(drift, n_left, n_all);
}

// ---------------------------------------------------------------------------------
// DensityF over BINNED queries, LDS-tiled (kernels_sph.hpp: k_pci_predict_bin has the why).  The tiles are those that
// hold queries (k_qtile_list), a tile's table comes from k_tile_desc with the query prefix as its target table: targets
// are the query records of the tile's 64 interior cells (predicted position + the particle's slot), candidates the
// particles of the 6 x 6 x 6 cells around them, staged exactly as in k_pci_density_tiled, and the sweep is that
// kernel's.  A query's cell comes from its place in the table -- the cell the sort put it in -- so a query that the
// cell rule clamped into the grid's outermost cells (the predictor knows no walls: by step 1500 of the 4M scene four
// out of five are below the floor) sweeps those cells' neighbourhood and finds nothing within h, as in the reference.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kTBlock) void k_pci_density_qtiled(DevConsts c, TileGrid tg, const int* __restrict__ n_qtiles,
                                                               const int* __restrict__ desc, const int* __restrict__ cell_start,
                                                               CSoa3 p, const float4* __restrict__ qrec,
                                                               float* __restrict__ press, DevStats* stats) {
  if (stats->pci_done) return;
  __shared__ TileMeta metas[3];
  __shared__ float4 Abuf[2][kTCap];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  unsigned int ebits = 0u;
  auto load4 = [&](int g, float4* o) {
    o[0] = load4u(p.x + g);
    o[1] = load4u(p.y + g);
    o[2] = load4u(p.z + g);
  };
  auto load1 = [&](int g, float* o) {
    o[0] = p.x[g];
    o[1] = p.y[g];
    o[2] = p.z[g];
  };
  auto commit = [&](const TileMeta& mt, const StageRegs<3>& sr, float4* img) {
    const float ox = __int_as_float(mt.centre[0]), oy = __int_as_float(mt.centre[1]), oz = __int_as_float(mt.centre[2]);
    stage_commit<3>(mt, sr, load1, [&](int slot, const float* o, bool real) {
      float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
      if (real) {
        const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
        v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
      }
      img[slot] = v;
    });
  };
  auto sweep = [&](const TileMeta& m, const float4* __restrict__ A) {
    const bool ovf = m.overflow != 0;
    const float ox = __int_as_float(m.centre[0]), oy = __int_as_float(m.centre[1]), oz = __int_as_float(m.centre[2]);
    const int ntarg = m.tprefix[kTB * kTB];
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(lane);
    for (int t = tperm; t < ntarg; t += kTBlock) {
      float4 rec;
      int lx = 1, qrow = 0;
      if (!ovf) {
        const TileTarget tt = tile_target(m, t);
        rec = qrec[tt.g];
        lx = tt.lx;
        qrow = tt.srow;
      } else {  // (no 16-bit cell boundaries to go by: the records of the tile's rows, one row after the other)
        int ir = 0;
        ir += (t >= m.tprefix[ir + 8]) ? 8 : 0;
        ir += (t >= m.tprefix[ir + 4]) ? 4 : 0;
        ir += (t >= m.tprefix[ir + 2]) ? 2 : 0;
        ir += (t >= m.tprefix[ir + 1]) ? 1 : 0;
        rec = qrec[m.trow[ir].x + t];
      }
      const float qx = rec.x, qy = rec.y, qz = rec.z;
      const int g = __float_as_int(rec.w);
      float density;
      if (!ovf) {
        const float rx = qx - ox, ry = qy - oy, rz = qz - oz;
        const float two_hh = 2.0f * c.inv_hh;
        const float sx = two_hh * rx, sy = two_hh * ry, sz = two_hh * rz;
        const float a0 = 1.0f - c.inv_hh * __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
        float acc = 0.f, acc1 = 0.f;
        auto test4 = [&](int jj) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 cnd = A[jj + u];
            const float q = fma_clamp01(cnd.z, sz, __builtin_fmaf(cnd.y, sy, __builtin_fmaf(cnd.x, sx, cnd.w + a0)));
            if (u & 1) acc1 = __builtin_fmaf(q, q, acc1);
            else acc = __builtin_fmaf(q, q, acc);
          }
        };
#pragma unroll 1
        for (int dz = -kTH; dz <= kTH; dz += kTH) {
#pragma unroll 1
          for (int dy = -1; dy <= 1; ++dy) {
            int j, je;
            tile_run(m, qrow + dz + dy, lx, j, je);
            for (; j + 4 < je; j += 8) {
              test4(j);
              test4(j + 4);
            }
            if (j < je) test4(j);
          }
        }
        density = __builtin_fmaf(acc + acc1, c.mass * c.A, c.W0);  // starts at W0, self included
      } else {
        density = c.W0;
        for_each_grid_candidate(c, cell_start, qx, qy, qz, [&](int j) {
          const float dx = qx - p.x[j], dy = qy - p.y[j], dz = qz - p.z[j];
          const float r2 = dist2<true>(dx, dy, dz);
          if (r2 < c.hh) {
            const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
            density = __builtin_fmaf(c.mass * c.A, q * q, density);
          }
        });
      }
      const float density_error = density - c.ref_density;
      const float abs_err = density_error * __builtin_amdgcn_rcpf(c.ref_density);
      press[g] += density_error * c.delta;
      const unsigned int eb = nonneg_bits(abs_err);
      ebits = eb > ebits ? eb : ebits;
    }
  };
  TileFeed feed(nullptr, *n_qtiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) {
    tile_meta_store(metas[0], tile_meta_request(desc, di));
    bool have_next = feed.pop(di);
    int table_word = have_next ? tile_meta_request(desc, di) : 0;
    sync_lds();  // the first tile's table is visible
    {
      StageRegs<3> sr;
      if (!metas[0].overflow) {
        stage_issue<3>(metas[0], load4, sr);
        commit(metas[0], sr, Abuf[0]);
      }
    }
    if (have_next) tile_meta_store(metas[1], table_word);
    int mc = 0, mn = 1, mnn = 2;  // tables of this tile, the next one, the one after
    for (int cur = 0; have; cur ^= 1) {
      sync_lds();  // image `cur` is complete, the next tile's table visible, image `cur ^ 1` and table `mnn` free
      StageRegs<3> sr;
      bool have_nn = false, stage_next = false;
      table_word = 0;
      if (have_next) {
        stage_next = metas[mn].overflow == 0;
        if (stage_next) stage_issue<3>(metas[mn], load4, sr);
        have_nn = feed.pop(di);
        if (have_nn) table_word = tile_meta_request(desc, di);
      }
      sweep(metas[mc], Abuf[cur]);
      if (stage_next) commit(metas[mn], sr, Abuf[cur ^ 1]);
      if (have_nn) tile_meta_store(metas[mnn], table_word);
      have = have_next;
      have_next = have_nn;
      const int t = mc;
      mc = mn;
      mn = mnn;
      mnn = t;
    }
  }
  wave_atomic_max(&stats->pci_cur_err_bits, ebits);
}

// The same sweep with TWO queries of one cell per lane (k_density_pair's scheme: every staged record read once and
// tested twice, every run set up once for two; 256-thread workgroups, four per CU, single-buffered).  Binned queries are
// what makes this possible for DensityF: the un-binned kernel's targets are the particles of a cell, whose query points
// the reference never brings back to that cell.  A slot is two queries of one cell or a cell's odd one out
// (TileMeta::pprefix, here over the QUERY counts); short last passes are shared out among 2 .. 16 lanes per slot.
// ROWS: the records live in per-cell rows of kQueryRow slots (tile_setup_load_counts) instead of one sorted array.
template <bool ROWS>
__global__ __launch_bounds__(kPBlock, 4) void k_pci_density_qpair(DevConsts c, TileGrid tg, const int* __restrict__ n_qtiles,
                                                                 const int* __restrict__ desc, const int* __restrict__ cell_start,
                                                                 CSoa3 p, const float4* __restrict__ qrec,
                                                                 float* __restrict__ press, DevStats* stats) {
  if (stats->pci_done) return;
  __shared__ TileMeta metas[2];
  __shared__ float4 A[kTCap];
  const int tid = threadIdx.x;
  unsigned int ebits = 0u;
  auto load4 = [&](int g, float4* o) {
    o[0] = load4u(p.x + g);
    o[1] = load4u(p.y + g);
    o[2] = load4u(p.z + g);
  };
  auto load1 = [&](int g, float* o) {
    o[0] = p.x[g];
    o[1] = p.y[g];
    o[2] = p.z[g];
  };
  auto meta_request = [&](int desc_index, int (&w)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * kPBlock;
      w[k] = i < kMetaInts ? desc[(size_t)desc_index * kMetaInts + i] : 0;
    }
  };
  auto meta_store = [&](TileMeta& m, const int (&w)[2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * kPBlock;
      if (i < kMetaInts) reinterpret_cast<int*>(&m)[i] = w[k];
    }
  };
  auto finish = [&](int g, float density) {  // pressure accumulate + the iteration's error (pcisph_darwin.go:76-92)
    if (ROWS && g < 0) return;  // (the tombstone of a query that has moved to another cell's row: k_pci_predict_bin<.., INCR>)
    const float density_error = density - c.ref_density;
    const float abs_err = density_error * __builtin_amdgcn_rcpf(c.ref_density);
    press[g] += density_error * c.delta;
    const unsigned int eb = nonneg_bits(abs_err);
    ebits = eb > ebits ? eb : ebits;
  };
  TileFeed feed(nullptr, *n_qtiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) {
    int w[2];
    meta_request(di, w);
    meta_store(metas[0], w);
  }
  for (int cur = 0; have; cur ^= 1) {
    TileMeta& m = metas[cur];
    sync_lds();  // the previous tile's sweep is over: its LDS records are free, this tile's table is visible
    have = feed.pop(di);
    int tw[2] = {0, 0};
    if (have) meta_request(di, tw);
    const bool ovf = m.overflow != 0;
    const float ox = __int_as_float(m.centre[0]), oy = __int_as_float(m.centre[1]), oz = __int_as_float(m.centre[2]);
    if (!ovf) {
      auto store = [&](int slot, const float* o, bool real) {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
        if (real) {
          const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
          v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
        }
        A[slot] = v;
      };
      StageRegs<3> sr0, sr1;
      stage_issue<3>(m, load4, sr0, tid);
      stage_issue<3>(m, load4, sr1, tid + kPBlock);
      stage_commit<3>(m, sr0, load1, store, tid);
      stage_commit<3>(m, sr1, load1, store, tid + kPBlock);
    }
    if (have) meta_store(metas[cur ^ 1], tw);
    sync_lds();
    if (ovf) {
      // (more candidates than the LDS image holds, or more queries than 16 bits count: one query per lane, global memory)
      const int ntarg = m.tprefix[kTB * kTB];
      for (int t = tid; t < ntarg; t += kPBlock) {
        int ir = 0;
        ir += (t >= m.tprefix[ir + 8]) ? 8 : 0;
        ir += (t >= m.tprefix[ir + 4]) ? 4 : 0;
        ir += (t >= m.tprefix[ir + 2]) ? 2 : 0;
        ir += (t >= m.tprefix[ir + 1]) ? 1 : 0;
        const int4 w = m.trow[ir];
        size_t at = (size_t)(w.x + t);
        if constexpr (ROWS) {  // (at most 64 x kQueryRow targets: the 16-bit cell boundaries are valid)
          const int b2 = w.z & 0xffff, b3 = (int)((unsigned)w.z >> 16), b4 = w.w;
          const int lx = (t >= b2 ? 1 : 0) + (t >= b3 ? 1 : 0) + (t >= b4 ? 1 : 0);
          const int begin = lx == 0 ? m.tprefix[ir] : (lx == 1 ? b2 : (lx == 2 ? b3 : b4));
          at = (size_t)(w.x + lx) * kQueryRow + (t - begin);
        }
        const float4 rec = qrec[at];
        float density = c.W0;
        for_each_grid_candidate(c, cell_start, rec.x, rec.y, rec.z, [&](int j) {
          const float dx = rec.x - p.x[j], dy = rec.y - p.y[j], dz = rec.z - p.z[j];
          const float r2 = dist2<true>(dx, dy, dz);
          if (r2 < c.hh) {
            const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
            density = __builtin_fmaf(c.mass * c.A, q * q, density);
          }
        });
        finish(__float_as_int(rec.w), density);
      }
      continue;
    }
    const int nslots = m.pprefix[kTB * kTB];
    // (nine-lane groups for the remainders, as in k_density_pair: measured neutral here -- 4M PCISPH after 400 steps 1700-1707
    // without, 1701-1704 with, profiles/r04_qpair_nine_lane_groups_ab.jsonl -- and not taken)
    for_each_target<true, kPBlock>(nslots, tid, tid, [&](auto shared_c, int u, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      const PairSlot ps = pair_slot(m, u);
      const int srow = ps.srow, lx = ps.lx;
      const bool two = ps.two;
      const size_t at = ROWS ? (size_t)(ps.rowbase + lx - 1) * kQueryRow + ps.j : (size_t)ps.g;
      const float4 r0 = qrec[at];
      const float4 r1 = qrec[at + (two ? 1 : 0)];
      const float two_hh = 2.0f * c.inv_hh;
      const float x0 = r0.x - ox, y0 = r0.y - oy, z0 = r0.z - oz, x1 = r1.x - ox, y1 = r1.y - oy, z1 = r1.z - oz;
      const float sx0 = two_hh * x0, sy0 = two_hh * y0, sz0 = two_hh * z0;
      const float sx1 = two_hh * x1, sy1 = two_hh * y1, sz1 = two_hh * z1;
      const float a00 = 1.0f - c.inv_hh * __builtin_fmaf(z0, z0, __builtin_fmaf(y0, y0, x0 * x0));
      const float a01 = two ? 1.0f - c.inv_hh * __builtin_fmaf(z1, z1, __builtin_fmaf(y1, y1, x1 * x1)) : -1.0e30f;
      float acc[2] = {0.0f, 0.0f}, accb[2] = {0.0f, 0.0f};
      auto sweep_run = [&](int rr) {
        int j, je;
        tile_run(m, rr, lx, j, je);
        auto test4 = [&](int jj) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float4 cnd = A[jj + v];
            const float q0 = fma_clamp01(cnd.z, sz0, __builtin_fmaf(cnd.y, sy0, __builtin_fmaf(cnd.x, sx0, cnd.w + a00)));
            const float q1 = fma_clamp01(cnd.z, sz1, __builtin_fmaf(cnd.y, sy1, __builtin_fmaf(cnd.x, sx1, cnd.w + a01)));
            if (v & 1) {
              accb[0] = __builtin_fmaf(q0, q0, accb[0]);
              accb[1] = __builtin_fmaf(q1, q1, accb[1]);
            } else {
              acc[0] = __builtin_fmaf(q0, q0, acc[0]);
              acc[1] = __builtin_fmaf(q1, q1, acc[1]);
            }
          }
        };
        for (; j + 4 < je; j += 8) {
          test4(j);
          test4(j + 4);
        }
        if (j < je) test4(j);
      };
      if constexpr (!SHARED) {
#pragma unroll 1
        for (int dz = -kTH; dz <= kTH; dz += kTH) {
#pragma unroll 1
          for (int dy = -1; dy <= 1; ++dy) sweep_run(srow + dz + dy);
        }
      } else {
#pragma unroll 1
        for (int ri = sub; ri < 9; ri += k) sweep_run(srow + (ri / 3 - 1) * kTH + (ri % 3 - 1));
      }
      acc[0] += accb[0];
      acc[1] += accb[1];
      if constexpr (SHARED) {
        for (int o = 1; o < k; o <<= 1) {  // the lanes of a group are active together
          acc[0] += __shfl_xor(acc[0], o, kWave);
          acc[1] += __shfl_xor(acc[1], o, kWave);
        }
        if (sub != 0) return;
      }
      finish(__float_as_int(r0.w), __builtin_fmaf(acc[0], c.mass * c.A, c.W0));  // starts at W0, self included
      if (two) finish(__float_as_int(r1.w), __builtin_fmaf(acc[1], c.mass * c.A, c.W0));
    });
  }
  wave_atomic_max(&stats->pci_cur_err_bits, ebits);
}

}  // namespace dsl
