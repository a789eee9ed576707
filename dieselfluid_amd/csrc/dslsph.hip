// dslsph.hip -- C ABI (include/dslsph.h) of the gfx950 SPH particle-step engine.
//
// Device data layout (all float32, SoA, resident in HBM for the life of the handle):
//   pos/vel        2 x 6 arrays  ping-pong: the counting sort scatters cur -> other, the
//                                fused force+integrate kernel writes other (Jacobi)
//   ids            2 x 1 int     slot -> original particle index (host order)
//   forces         2 x 3 arrays  only materialised while they differ from force_reset
//   pci pos/vel    2 x 6 arrays  PCISPH predictor state (pcisph_darwin.go:28-41)
//   rho, pterm, press            per-slot density, P/rho^2, "pressures" buffer
//   rank, cell_count, cell_start, block_sums   neighbour table
// Host buffers are the reference's interleaved xyz float32 (model/particle_array.go:5-15).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see
// __graft_entry__.build()).  gfx950 only; no CPU fallback exists in this library.
#include "../../include/dslsph.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "kernels_lsh.hpp"
#include "kernels_tiled.hpp"
#include "kernels_skin.hpp"

struct SlabLink;

using namespace dsl;

namespace {
std::string g_create_error;
}

// DSL_OPT_SKIN of a DSL_MATH_FAST handle ...  (0.08: measured best of 0.06 .. 0.10 on the 16M lattice -- 9234 / 9387 /
// 9288 M particle-steps/s, profiles/r04 -- and a tile of 4 x 4 x 3 such cells stays a fifth below the LDS image's kTCap
// records where 0.10 sits at its edge: at 0.11 the lattice's fullest tiles no longer fit and the step takes 3.4 ms)
constexpr float kSkinDefault = 0.07f;  // (0.08 before the lists were built at predicted positions: profiles/r04_skin_sweep_predict.jsonl)
// the lists are built where the particles will be about half way through the lists' life: the fastest particle may use up
// this fraction of the displacement budget at the build itself (DSL_OPT_SKIN_PREDICT; 0: built where the particles are)
constexpr float kSkinPredict = 0.8f;
constexpr int kSkinDefaultParticles = 200000;  // ... of at least this many particles (measured: +24 % at 262k, +21 % at 1M, +15 % at 2M, +19 % at 16M)

struct dsl_handle {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  dsl_params prm{};
  DevConsts c{};
  int n = 0, cap = 0, ncell = 0, ncell_pad = 0, nscan = 0;
  // boundary particles (particle_array.go:123-128): ids n_fluid .. n-1.  The reference's Get() reads
  // index == N() as the zero particle (particle_array.go:98,107), so the first boundary particle takes
  // part in every sum AT THE ORIGIN; the device copy holds (0,0,0) for it and b0_pos what was uploaded.
  int nb = 0;  // boundary particles; N() = n - nb (n itself changes in slab mode, where nb is always 0)
  float b0_pos[3] = {0.f, 0.f, 0.f};
  bool ids_global = false;  // dsl_set_ids replaced the host-order map
  int* dcounter = nullptr;
  // slab mode: [0] live particle count, [1..4] band counters (lo/hi full, lo/hi position-only),
  // [5] overflow high-water mark, [6],[7] largest full / position-only band counts seen
  int* dn = nullptr;
  int* pack_counts = nullptr;  // band selection: per-block record counts, then offsets (kernels_grid.hpp)
  // split slab step (dsl_force_pass_split): band tiles first, interior tiles after the pack
  float split_margin = 0.0f, split_width = 0.0f;
  bool split_pending = false;
  bool pci_split_guard = false;  // between DSL_PCI_BEGIN_STEP and DSL_PCI_END_STEP
  hipEvent_t ev_band = nullptr;
  // DSL_NEIGH_LSH_REF
  bool lsh = false;
  int lsh_bits = 0, lsh_cap = 0;
  float* hashv = nullptr;
  int *bucket_of = nullptr, *lsh_table = nullptr, *lsh_len = nullptr, *lsh_samples = nullptr;
  TileGrid tg{};
  int *tiles = nullptr, *n_tiles = nullptr;
  int *tile_desc_of = nullptr, *tile_desc = nullptr;  // per list entry: index of the tile's table; the tables (k_tile_desc)
  // SoA state
  float* pv[2][6] = {};
  int* ids[2] = {};
  float* frc[2][3] = {};
  float* pci[2][6] = {};
  float *rho = nullptr, *pterm = nullptr, *press = nullptr, *scratch1 = nullptr;
  float* gterm[3] = {};  // PCISPH: cached pressure-gradient force term
  // PCISPH with the DensityF query points in bins of their own (kernels_sph.hpp: k_pci_predict_bin): the histogram over
  // the particles' grid cells, its prefix, the scan's tile sums, one record per query, and the 512 drift counters the
  // un-binned kernels keep.  pci_bin_mode: 0 = the library switches when 0.2 % of the predicted positions have left their
  // particle's tile (looked at every kPciDriftPeriod steps; a one-way latch until the next dsl_pcisph_begin), 1 = always,
  // -1 = never (dsl_pcisph_set_binning; DSL_PCI_BINNED presets it)
  int pci_bin_mode = 0;
  int build_seq = 0;  // neighbour builds so far (k_cell_rank<true> stamps its off-grid flag with it)
  // the query histogram was handed to an iteration that may not have cleaned it again (the table builder / the scan do,
  // behind the predictor: an error return in between leaves it dirty -- the sort's sort_scratch_dirty, for the queries)
  bool qcount_dirty = false;
  bool pci_counters_clean = false;  // n_qtiles[0..1] are known to be zero (the first binned iteration clears them itself)
  bool pci_iter_pending = false;  // DSL_PCI_ITERATE without its DSL_PCI_CHECK yet (the check clears the iteration's counters)
  bool pci_binned = false;
  bool pci_qpair = true;   // ... two queries per lane (DSL_PCI_QPAIR=0: one)
  bool pci_qtiled = true;  // FAST: sweep the binned queries tile by tile out of LDS (DSL_PCI_QTILED=0: the global-memory sweep)
  int64_t pci_steps = 0;  // completed steps since dsl_pcisph_begin
  int *qcount = nullptr, *qstart = nullptr, *qsums = nullptr;
  int *qtiles = nullptr, *n_qtiles = nullptr, *qtile_desc = nullptr;  // FAST: the tiles that hold queries and their tables
  // FAST, two queries per lane: kQueryRow record slots per grid cell instead of the sorted array (no prefix scan, no
  // scatter pass: kernels_tiled.hpp, tile_setup_load_counts); DSL_PCI_QROWS=0, or an allocation that fails, keeps the array
  float4* qrows = nullptr;
  bool pci_qrows = true;
  // ... and the rows kept from one correction iteration to the next inside a step (k_pci_predict_bin<.., INCR>): qslot =
  // every particle's record index; pci_rows_live = this step's rows hold every query (its first iteration filled them)
  int* qslot = nullptr;
  bool pci_qincr = true;
  bool pci_rows_live = false;
  float4* qrec = nullptr;
  unsigned int* pci_drift = nullptr;
  unsigned int* pci_drift_host = nullptr;  // pinned: the snapshot of the counters the next look reads
  hipEvent_t ev_drift = nullptr;
  bool drift_pending = false;
  float* xsph[3] = {};   // build-defined XSPH correction of the current step (PCISPH path)
  unsigned int* nmask = nullptr;  // FAST: per-particle in-range masks from the density sweep (kMaskWords x cap)
  bool masks_valid = false;
  int *rank = nullptr, *cell_count = nullptr, *cell_start = nullptr, *block_sums = nullptr;
  // in-cell ordering of the counting sort (kernels_grid.hpp, k_scatter): bitmap of the cells to order,
  // their ids at the slots the atomic ranks name, one byte per particle that marks their particles
  unsigned int* unordered = nullptr;
  int *sort_keys = nullptr, *sort_work = nullptr;
  int* cell_keys = nullptr;  // kCellKeys ids per grid cell: the one-pass in-cell ordering (kernels_grid.hpp); DSL_OPT_CELL_KEYS = 0: nullptr
  int* cell_keys_alloc = nullptr;
  float* stage = nullptr;
  DevStats* dstats = nullptr;
  int cur_pv = 0, cur_ids = 0, cur_f = 0, cur_pci = 0;
  bool grid_valid = false, forces_uniform = false, press_zero = true, pci_active = false, dens_fresh = false;
  // dens_held: rho / pterm hold per-particle values in the CURRENT slot order (possibly stale ones: after a mass
  // change or new boundary particles the reference, too, keeps the old density on the same particle until the
  // next DensityAll), so a sort has to carry them.  dens_fresh: they also belong to the current positions / mass.
  bool dens_held = false;
  // the histogram / "cells to order" bitmap were handed to a build that may not have cleaned them again
  bool sort_scratch_dirty = false;
  // (a one-launch decoupled look-back scan was measured in round 3: 6 x SLOWER at 16.4M cells, 0.46 against 0.077 ms --
  // all 4004 tiles are resident at once, so the look-back is one long chain of agent-scope round trips through the
  // eight XCDs' separate L2s; profiles/README.md, r03.  Removed in round 4.)
  bool density_pair = true;  // FAST density with two targets per lane (DSL_DENSITY_PAIR=0: the lane-per-target kernel)
  int max_persistent_blocks = 0;  // DSL_PERSISTENT_BLOCKS (tests)
  int64_t steps = 0;
  size_t dev_bytes = 0;  // device memory this handle has allocated (DSL_OPT_DEVICE_BYTES)
  // the skin step (kernels_skin.hpp; DSL_OPT_SKIN): neighbour lists against h (1 + skin), walked until some particle may
  // have moved skin * h / 2.  skin_live: the device's SkinState is the authority on which slot -> particle map is current
  // and the grid's cells are h (1 + skin) wide -- every other entry point settles that first (skin_settle).
  float skin = 0.0f;
  bool skin_live = false;
  int64_t skin_steps_total = 0, skin_rebuilds_total = 0;  // of the skin episodes already settled
  bool skin_list_overflow = false;
  // the flow outran the skin (SkinState::give_up, looked at every kSkinLook steps): plain steps until step skin_retry_at
  int64_t skin_retry_at = 0, skin_live_steps = 0;
  int skin_suspensions = 0;
  int tile_box[3] = {8, 4, 4};
  int alloc_ntiles = 0;  // tiles the tile lists and tables were allocated for
  int tile_tbz = kTB;  // cell layers per tile along z (kernels_tiled.hpp: TileGrid::tbz); 3 while the skin step owns the grid
  SkinState* skin_state = nullptr;
  uint4* lists = nullptr;
  float* pvz[6] = {};  // the sort's output set of a skin step (Z)
  float* pvr[3] = {};  // ... and the build's reference positions x + tau v in the same order (SkinState::tau)
  float skin_predict = kSkinPredict;  // DSL_OPT_SKIN_PREDICT
  bool list_build_lockstep = true;    // DSL_OPT_LIST_BUILD
  int grid_oversub = 8;               // DSL_OPT_GRID_OVERSUB
  // DSL_OPT_TILE_QUEUE: the single-domain tile kernels draw their tiles from per-XCD counters (kernels_tiled.hpp: TileFeed,
  // dynamic mode) instead of walking a share dealt in advance.  16 launch sites x 16 ints, left at zero by every launch.
  int tile_queue = 1;  // 0 never, 1 from 8M particles on, 2 always (tests)
  int* walk_ctr = nullptr;
  int walk_parity[16] = {};
  std::string err;
  SlabLink* link = nullptr;  // dsl_slab_attach: the slab's RCCL link to its neighbours (slab_link.hpp)
  // timing
  int timing = 0;  // 0 off, 1 every kernel, 2 the step's dominant kernels only
  std::vector<hipEvent_t> pool;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[DSL_K_COUNT];
  double total_ms[DSL_K_COUNT] = {};
  int64_t launches[DSL_K_COUNT] = {};
};

namespace {

int fail(dsl_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                             \
  do {                                                                                               \
    hipError_t e__ = (expr);                                                                         \
    if (e__ != hipSuccess)                                                                           \
      return fail((h), DSL_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));          \
  } while (0)

#define CHECK_HANDLE_ONLY(h)                              \
  do {                                                    \
    if (!(h)) return fail(nullptr, DSL_ERR_INVALID, "null handle"); \
    hipError_t e__ = hipSetDevice((h)->device);           \
    if (e__ != hipSuccess) return fail((h), DSL_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e__)); \
  } while (0)
int skin_settle(dsl_handle* h);
// every entry point but the step driver itself first brings a handle that is in the middle of skin steps back to the
// plain state (one small device read: kernels_skin.hpp)
#define CHECK_HANDLE(h)                                   \
  do {                                                    \
    CHECK_HANDLE_ONLY(h);                                 \
    if ((h)->skin_live)                                   \
      if (int rc__ = skin_settle(h)) return rc__;         \
  } while (0)

inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }
void update_split(dsl_handle* h);
// number of slots a per-particle launch has to cover: exact without slabs; in slab mode the
// live count lives on the device, so cover the whole capacity (idle blocks exit at once)
inline int launch_n(const dsl_handle* h);

// Fills the device constants from the parameter block; host arithmetic mirrors
// kernel.Build_Kernel (kernel/std_kernel.go:20-31) in float32.
int make_consts(dsl_handle* h, const dsl_params& p, DevConsts& c) {
  if (!(p.h > 0.0f)) return fail(h, DSL_ERR_INVALID, "h must be > 0");
  if (!(p.mass > 0.0f)) return fail(h, DSL_ERR_INVALID, "mass must be > 0");
  if (p.n_particles <= 0) return fail(h, DSL_ERR_INVALID, "n_particles must be > 0");
  if (p.n_boundary < 0) return fail(h, DSL_ERR_INVALID, "n_boundary must be >= 0");
  if (p.neigh_mode != DSL_NEIGH_GRID && p.neigh_mode != DSL_NEIGH_LSH_REF) return fail(h, DSL_ERR_INVALID, "bad neigh_mode");
  if (p.neigh_mode == DSL_NEIGH_LSH_REF && p.math_mode != DSL_MATH_EXACT)
    return fail(h, DSL_ERR_UNSUPPORTED, "DSL_NEIGH_LSH_REF is a parity mode: it needs DSL_MATH_EXACT");
  if (p.neigh_mode == DSL_NEIGH_LSH_REF && (p.lsh_buckets < 1 || p.lsh_buckets > 4096))
    return fail(h, DSL_ERR_INVALID, "lsh_buckets out of range");
  if (p.math_mode != DSL_MATH_EXACT && p.math_mode != DSL_MATH_FAST) return fail(h, DSL_ERR_INVALID, "bad math_mode");
  c.n = p.n_particles + p.n_boundary;  // Total(): every slot is a neighbour candidate
  const float hh = p.h;
  c.h = hh;
  c.hh = hh * hh;
  c.inv_h = 1.0f / hh;
  c.inv_hh = 1.0f / c.hh;
  {  // std::sqrt(float) is correctly rounded and monotone: walk to the first float whose root reaches h
    float t = c.hh;
    while (std::sqrt(t) >= hh && t > 0.0f) t = std::nextafter(t, 0.0f);
    while (std::sqrt(t) < hh) t = std::nextafter(t, INFINITY);
    c.r2_thr = t;
  }
  const float H3 = hh * hh * hh, H4 = hh * hh * hh * hh, H5 = hh * hh * hh * hh * hh;
  const double PI = 3.141592653589;  // kernel/std_kernel.go:5
  c.A = 315.0f / ((float)(64.0 * PI) * H3);
  c.B = -45.0f / ((float)PI * H4);
  c.C = 90.0f / ((float)PI * H5);
  c.W0 = (c.A * 1.0f) * 1.0f;  // F(0): q = 1 - 0/hh = 1
  c.mass = p.mass;
  c.inv_mass = 1.0f / p.mass;
  c.ref_density = p.ref_density;
  c.mu = p.mu;
  c.dt = p.dt;
  c.delta = p.delta;
  c.eos_wg = p.eos_w / p.eos_gamma;
  c.eos_gamma = p.eos_gamma;
  c.eos_d0_grad = p.eos_d0_grad;
  c.pressure_sign = p.pressure_sign;
  c.visc_running_mass = p.visc_running_mass;
  for (int a = 0; a < 3; ++a) {
    c.reset[a] = p.force_reset[a];
    c.ext[a] = p.external[a];
    c.bmin[a] = p.box_min[a];
    c.bmax[a] = p.box_max[a];
    c.gmin[a] = p.grid_min[a];
  }
  c.wcsph_pressure_force = p.wcsph_pressure_force;
  c.wcsph_viscosity = p.wcsph_viscosity;
  c.pci_max_error = p.pci_max_error;
  c.xsph_eps = p.xsph_eps;
  c.st_kappa = p.st_kappa;
  c.walls = p.walls;
  c.rest = p.restitution;
  c.inv_cell = 1.0f / hh;
  c.cell = hh;
  long long ncell = 1;
  for (int a = 0; a < 3; ++a) {
    int d = (int)ceilf((p.grid_max[a] - p.grid_min[a]) * c.inv_cell);
    if (!(p.grid_max[a] > p.grid_min[a])) return fail(h, DSL_ERR_INVALID, "grid_max must exceed grid_min");
    if (d < 1) d = 1;
    c.dims[a] = d;
    ncell *= d;
  }
  if (ncell > (1ll << 30)) return fail(h, DSL_ERR_INVALID, "grid has more than 2^30 cells; shrink the grid box or enlarge h");
  c.ncell = (int)ncell;
  if (p.capacity != 0 && p.capacity < p.n_particles + p.n_boundary)
    return fail(h, DSL_ERR_INVALID, "capacity must be >= n_particles + n_boundary");
  c.slab_axis = -1;
  c.slab_lo = -INFINITY;
  c.slab_hi = INFINITY;
  c.chk_lo = -INFINITY;
  c.chk_hi = INFINITY;
  c.split_cl = 0;
  c.split_ch = INT_MAX;
  c.split_part = 0;
  c.own_c0 = 0;
  c.own_c1 = INT_MAX;
  c.n_ptr = nullptr;
  return DSL_OK;
}

inline int launch_n(const dsl_handle* h) { return h->c.n_ptr ? h->cap : h->n; }

// the tile grid over the current cell grid (kernels_tiled.hpp: TileGrid)
int set_tile_grid(dsl_handle* h) {
  TileGrid& tg = h->tg;
  tg.tnx = (h->c.dims[0] + kTB - 1) / kTB;
  tg.tny = (h->c.dims[1] + kTB - 1) / kTB;
  tg.tbz = h->tile_tbz;
  tg.tnz = (h->c.dims[2] + tg.tbz - 1) / tg.tbz;
  tg.ntiles = tg.tnx * tg.tny * tg.tnz;
  const int bx = h->tile_box[0], by = h->tile_box[1], bz = h->tile_box[2];
  tg.bx = bx;
  tg.by = by;
  tg.bz = bz;
  tg.nbx = bx ? (tg.tnx + bx - 1) / bx : 0;
  tg.nby = bx ? (tg.tny + by - 1) / by : 0;
  const long long nlist = bx ? (long long)tg.nbx * tg.nby * ((tg.tnz + bz - 1) / bz) * (bx * by * bz) : tg.ntiles;
  if (nlist > 0x7fffffffLL) return fail(h, DSL_ERR_INVALID, "tile grid too large");
  tg.nlist = (int)nlist;
  return DSL_OK;
}

// Cells `edge` wide over the same grid box: the skin step widens them to h (1 + skin) and back.  Every array was sized
// for the finest grid (edge = h, at creation), a coarser one fits in; the histogram and the "cells to order" bitmap are
// left clean by every complete build whatever the grid.
int set_cell_edge(dsl_handle* h, float edge, int tbz = kTB) {
  DevConsts& c = h->c;
  h->tile_tbz = tbz;
  c.cell = edge;
  c.inv_cell = 1.0f / edge;
  long long ncell = 1;
  for (int a = 0; a < 3; ++a) {
    int d = (int)ceilf((h->prm.grid_max[a] - h->prm.grid_min[a]) * c.inv_cell);
    if (d < 1) d = 1;
    c.dims[a] = d;
    ncell *= d;
  }
  c.ncell = (int)ncell;
  h->ncell = c.ncell;
  h->ncell_pad = ((h->ncell + 1 + kScanTile - 1) / kScanTile) * kScanTile;
  h->nscan = h->ncell_pad / kScanTile;
  h->grid_valid = false;
  h->masks_valid = false;
  return set_tile_grid(h);
}

template <class T>
int dev_alloc(dsl_handle* h, T** p, size_t count) {
  // (64 bytes of padding: the staging quads of the tiled kernels read up to three elements past a row, the ordered
  // scatter up to seven keys past a cell)
  hipError_t e = hipMalloc((void**)p, count * sizeof(T) + 64);
  if (e != hipSuccess) return fail(h, DSL_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  // Every array starts from zeros (NewParticleArray zero-fills its slices too, particle_array.go:18-33): hipMalloc hands
  // out whatever the previous owner left, and nothing a run computes may depend on that -- neither the padding the
  // unaligned staging quads read past a row nor a plane that is only written for some particles.  Once per handle.
  e = hipMemsetAsync(*p, 0, count * sizeof(T) + 64, h->stream);
  if (e != hipSuccess) return fail(h, DSL_ERR_DEVICE, std::string("hipMemsetAsync: ") + hipGetErrorString(e));
  h->dev_bytes += count * sizeof(T) + 64;
  return DSL_OK;
}

// Two side arrays cost memory per GRID CELL, not per particle: the cells' key rows of the one-pass in-cell ordering (128 B
// per cell) and the PCISPH query rows (512 B per cell).  In the dam-break box (8 x the fluid's volume) that is 128 / 512 B
// per particle; a tall or sparse domain multiplies it.  Each is therefore only allocated within a budget -- the larger of
// 4 GiB and 64 B per particle slot, or dsl_params.reserved[0] MiB if that is set -- and the kernels fall back to the forms
// that need none (two-pass ordering; the queries' sorted array), which every parity test also runs.
bool side_array_fits(const dsl_handle* h, size_t bytes) {
  size_t budget = std::max((size_t)4 << 30, (size_t)64 * (size_t)h->cap);
  if (h->prm.reserved[0] > 0) budget = (size_t)h->prm.reserved[0] << 20;
  return bytes <= budget;
}

hipEvent_t get_event(dsl_handle* h) {
  if (!h->pool.empty()) {
    hipEvent_t e = h->pool.back();
    h->pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

template <class F>
int timed(dsl_handle* h, int kid, F&& launch) {
  if (h->timing == 1 || (h->timing == 2 && (kid == DSL_K_DENSITY || kid == DSL_K_FORCE_INTEGRATE || kid == DSL_K_PCI_DENSITY))) {
    hipEvent_t a = get_event(h), b = get_event(h);
    (void)hipEventRecord(a, h->stream);
    launch();
    (void)hipEventRecord(b, h->stream);
    h->pending[kid].emplace_back(a, b);
  } else {
    launch();
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, DSL_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(e));
  return DSL_OK;
}

int drain_timing(dsl_handle* h) {
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int k = 0; k < DSL_K_COUNT; ++k) {
    for (auto& pr : h->pending[k]) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
        h->total_ms[k] += ms;
        h->launches[k] += 1;
      }
      h->pool.push_back(pr.first);
      h->pool.push_back(pr.second);
    }
    h->pending[k].clear();
  }
  return DSL_OK;
}

CSoa3 cpos(dsl_handle* h) { return {h->pv[h->cur_pv][0], h->pv[h->cur_pv][1], h->pv[h->cur_pv][2]}; }
CSoa3 cvel(dsl_handle* h) { return {h->pv[h->cur_pv][3], h->pv[h->cur_pv][4], h->pv[h->cur_pv][5]}; }
Soa3 mpos(dsl_handle* h, int w) { return {h->pv[w][0], h->pv[w][1], h->pv[w][2]}; }
Soa3 mvel(dsl_handle* h, int w) { return {h->pv[w][3], h->pv[w][4], h->pv[w][5]}; }
Soa3 mfrc(dsl_handle* h) { return {h->frc[h->cur_f][0], h->frc[h->cur_f][1], h->frc[h->cur_f][2]}; }
CSoa3 cfrc(dsl_handle* h) { return {h->frc[h->cur_f][0], h->frc[h->cur_f][1], h->frc[h->cur_f][2]}; }
Soa3 mpcip(dsl_handle* h) { return {h->pci[h->cur_pci][0], h->pci[h->cur_pci][1], h->pci[h->cur_pci][2]}; }
Soa3 mpciv(dsl_handle* h) { return {h->pci[h->cur_pci][3], h->pci[h->cur_pci][4], h->pci[h->cur_pci][5]}; }

inline int n_fluid_of(const dsl_handle* h) { return h->n - h->nb; }
Bnd bnd_of(const dsl_handle* h) { return Bnd{h->nb > 0 ? h->ids[h->cur_ids] : nullptr, n_fluid_of(h)}; }

Neigh neigh(const dsl_handle* h) {
  if (h->lsh) return Neigh{h->cell_start, h->lsh_samples, h->hashv, h->lsh_bits, h->prm.lsh_buckets};
  return Neigh{h->cell_start, nullptr, nullptr, 0, 0};
}

// forces are kept implicit (== force_reset) after Update until a pass needs the array
int materialise_forces(dsl_handle* h) {
  if (!h->forces_uniform) return DSL_OK;
  Soa3 f = mfrc(h);
  const DevConsts& c = h->c;
  hipLaunchKernelGGL(k_fill3, dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, launch_n(h), f.x, f.y, f.z, c.reset[0],
                     c.reset[1], c.reset[2]);
  HIP_TRY(h, hipGetLastError());
  h->forces_uniform = false;
  return DSL_OK;
}
int materialise_press(dsl_handle* h) {
  if (!h->press_zero) return DSL_OK;
  HIP_TRY(h, hipMemsetAsync(h->press, 0, sizeof(float) * (size_t)launch_n(h), h->stream));
  h->press_zero = false;
  return DSL_OK;
}

// cell hash -> histogram -> prefix sum -> counting-sort scatter.  carry_derived also
// permutes rho/pterm/press so that an explicit dsl_build_neighbours keeps them usable.
// HashSampler.UpdateSampler on the device: nothing is re-ordered, slots stay in host order
int build_lsh(dsl_handle* h) {
  const int n = h->n;
  CSoa3 p = cpos(h);
  Neigh nb = neigh(h);
  const int B = h->prm.lsh_buckets;
  int rc = timed(h, DSL_K_CELL_RANK, [&] {
    hipLaunchKernelGGL(k_lsh_bucket, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, n, nb, p.x, p.y, p.z, h->bucket_of);
    hipLaunchKernelGGL(k_lsh_table, dim3(B), dim3(kWave), 0, h->stream, n, h->bucket_of, h->lsh_cap, h->lsh_table,
                       h->lsh_len);
    hipLaunchKernelGGL(k_lsh_samples, dim3((B + 63) / 64), dim3(64), 0, h->stream, B, h->lsh_cap, h->lsh_table,
                       h->lsh_len, h->lsh_samples);
  });
  if (rc) return rc;
  h->grid_valid = true;
  return DSL_OK;
}

int build_grid(dsl_handle* h, bool carry_derived) {
  if (h->split_pending) return fail(h, DSL_ERR_INVALID, "a split force pass is in flight: finish it with DSL_SPLIT_INNER");
  if (h->lsh) return build_lsh(h);
  const int n = launch_n(h);
  h->masks_valid = false;  // slot order changes
  const DevConsts& c = h->c;
  CSoa3 p = cpos(h);
  // (the histogram and the "cells to order" bitmap are clean: zeroed at creation, and again by
  // k_scan_apply / k_tile_list of the previous build)
  const bool ordered = !h->prm.sort_unordered;
  if (h->sort_scratch_dirty) {  // an earlier build stopped between k_cell_rank and the kernels that clean up behind it
    HIP_TRY(h, hipMemsetAsync(h->cell_count, 0, sizeof(int) * (size_t)h->ncell_pad, h->stream));
    if (h->unordered) HIP_TRY(h, hipMemsetAsync(h->unordered, 0, sizeof(unsigned int) * (size_t)(h->ncell_pad / 32), h->stream));
  }
  h->sort_scratch_dirty = true;
  // PCISPH: "a particle lies outside the grid's bounds" (dcounter[4] = the number of the last build that saw one;
  // k_pci_predict_bin finishes far-away queries on the spot when this build saw none)
  h->build_seq = h->build_seq == INT_MAX ? 1 : h->build_seq + 1;  // (never 0: the flag word starts zeroed)
  int rc = timed(h, DSL_K_CELL_RANK, [&] {
    if (h->pci_active)
      hipLaunchKernelGGL(k_cell_rank<true>, dim3(grid_for((n + kRankUnroll - 1) / kRankUnroll)), dim3(kBlock), 0, h->stream, c, p.x, p.y, p.z,
                         ordered ? h->ids[h->cur_ids] : nullptr, h->rank, h->cell_count, h->unordered,
                         nullptr, nullptr, ordered ? h->cell_keys : nullptr, h->dcounter + 3, h->dcounter + 4, h->build_seq);
    else
      hipLaunchKernelGGL(k_cell_rank<false>, dim3(grid_for((n + kRankUnroll - 1) / kRankUnroll)), dim3(kBlock), 0, h->stream, c, p.x, p.y, p.z,
                         ordered ? h->ids[h->cur_ids] : nullptr, h->rank, h->cell_count, h->unordered,
                         nullptr, nullptr, ordered ? h->cell_keys : nullptr, h->dcounter + 3, nullptr, 0);
  });
  if (rc) return rc;
  rc = timed(h, DSL_K_SCAN, [&] {
    hipLaunchKernelGGL(k_scan_sums, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->cell_count, h->block_sums, h->dstats,
                       !h->lsh ? h->n_tiles : nullptr);
    hipLaunchKernelGGL(k_scan_apply, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->cell_count, h->block_sums,
                       h->cell_start, h->dstats);
  });
  if (rc) return rc;
  ScatterArrays a{};
  int nf = 0;
  const int s = h->cur_pv, d = s ^ 1;
  for (int k = 0; k < 6; ++k) {
    a.src[nf] = h->pv[s][k];
    a.dst[nf++] = h->pv[d][k];
  }
  if (!h->forces_uniform)
    for (int k = 0; k < 3; ++k) {
      a.src[nf] = h->frc[h->cur_f][k];
      a.dst[nf++] = h->frc[h->cur_f ^ 1][k];
    }
  if (h->pci_active)
    for (int k = 0; k < 6; ++k) {
      a.src[nf] = h->pci[h->cur_pci][k];
      a.dst[nf++] = h->pci[h->cur_pci ^ 1][k];
    }
  a.nf = nf;
  a.ids_src = h->ids[h->cur_ids];
  a.ids_dst = h->ids[h->cur_ids ^ 1];
  // derived arrays (an explicit dsl_build_neighbours behind a density pass: rare) follow through the
  // particles' final slots, which the scatter leaves in place of the ranks
  float* derived[3] = {(carry_derived && h->dens_held) ? h->rho : nullptr, (carry_derived && h->dens_held) ? h->pterm : nullptr,
                       (carry_derived && !h->press_zero) ? h->press : nullptr};
  const bool want_dest = derived[0] || derived[1] || derived[2];
  ScatterOrder so{ordered ? h->unordered : nullptr, h->sort_keys, reinterpret_cast<unsigned char*>(h->sort_work),
                  want_dest ? h->rank : nullptr, ordered ? h->cell_keys : nullptr, h->dcounter + 3};
  rc = timed(h, DSL_K_SCATTER, [&] {
    if (a.nf == 6)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter<false, 6>), dim3(grid_for((n + kScatterUnroll - 1) / kScatterUnroll)), dim3(kBlock), 0, h->stream, c, a, so, p, h->rank, h->cell_start);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter<false, kMaxScatter>), dim3(grid_for((n + kScatterUnroll - 1) / kScatterUnroll)), dim3(kBlock), 0, h->stream, c, a, so, p, h->rank, h->cell_start);
    if (ordered)  // (with key rows: the flag-gated fallback for cells of more than kCellKeys particles, a small grid)
      hipLaunchKernelGGL(k_scatter_ordered, dim3(h->cell_keys ? std::min(grid_for(n), 1024) : grid_for(n)), dim3(kBlock), 0,
                         h->stream, c, a, so, p, h->rank, h->cell_start);
  });
  if (rc) return rc;
  if (carry_derived) {
    for (float* arr : derived) {
      if (!arr) continue;
      hipLaunchKernelGGL(k_permute1, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, c, h->rank, p, arr, h->scratch1);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipMemcpyAsync(arr, h->scratch1, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    }
  } else {
    h->dens_fresh = false;
    h->dens_held = false;
  }
  h->cur_pv ^= 1;
  h->cur_ids ^= 1;
  if (!h->forces_uniform) h->cur_f ^= 1;
  if (h->pci_active) h->cur_pci ^= 1;
  h->grid_valid = true;
  if (!h->lsh) {
    rc = timed(h, DSL_K_TILE_LIST, [&] {
      hipLaunchKernelGGL(k_tile_list, dim3(grid_for(h->tg.nlist)), dim3(kBlock), 0, h->stream, h->c, h->tg,
                         h->cell_start, h->tiles, h->n_tiles, h->n_tiles + 5, h->c.n_ptr ? h->dn : nullptr, h->tile_desc_of,
                         h->unordered, h->ncell_pad / 32, h->dcounter + 3);
      // the tables of the non-empty tiles, once per build for every kernel that sweeps tiles
      hipLaunchKernelGGL(k_tile_desc, dim3(std::min(h->tg.ntiles, 8192)), dim3(kWave), 0, h->stream, h->c, h->tg,
                         h->cell_start, h->cell_start, h->tiles, h->n_tiles, h->tile_desc);
    });
    if (rc) return rc;
    h->sort_scratch_dirty = false;  // k_scan_apply and k_tile_list are queued: they leave both arrays clean
  }
  if (h->c.n_ptr && h->lsh) {
    // the sort has dropped the stale ghosts; the live count stays on the device (otherwise k_tile_list did it)
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, h->stream, h->dn, h->cell_start + h->ncell);
    HIP_TRY(h, hipGetLastError());
  }
  return DSL_OK;
}

int ensure_grid(dsl_handle* h) { return h->grid_valid ? DSL_OK : build_grid(h, true); }

template <class Launch>
int by_math(dsl_handle* h, Launch&& l) {
  if (h->prm.math_mode == DSL_MATH_FAST) l(std::true_type{});
  else l(std::false_type{});
  return DSL_OK;
}

// FAST mode uses the LDS-tiled kernels; the reference's running-mass viscosity recurrence
// (sph_field.go:265) is order- and membership-dependent for m != 1 and stays on the
// branching lane-per-particle kernel.
// EXACT mode on the LDS-tiled kernels (same staging and masks, the reference's own operations and
// order): single-domain grid runs; slab ranks and lsh_ref keep the lane-per-particle kernels.
bool exact_tiled(const dsl_handle* h) {
  return h->prm.math_mode == DSL_MATH_EXACT && !h->lsh && h->c.slab_axis < 0 && h->nmask != nullptr;
}
bool use_tiled(const dsl_handle* h) {
  if (h->prm.math_mode != DSL_MATH_FAST || h->lsh) return false;
  if (h->c.wcsph_viscosity && h->c.visc_running_mass && h->c.mass != 1.0f) return false;
  return true;
}
// `lists`: the skin step's list kernels -- their tiles cost the same and their workgroups stay persistent
int persistent_grid(const dsl_handle* h, int blocks_per_cu, bool lists = false) {
  // DSL_OPT_GRID_OVERSUB (default 8): eight times the workgroups the chip holds, each with an eighth of the share -- the
  // hardware hands them out as workgroups retire, which evens out tiles of unequal cost: developed flow density 0.96 ->
  // 0.89 ms, force 1.29 -> 1.24, lattice 0.64 -> 0.61 / 0.86 -> 0.83 (profiles/r04_grid_oversub.jsonl; 16 and 32: less)
  // Never fewer than ~4 tiles per workgroup (a workgroup fetches its next tile's table and records under the current
  // tile's sweep: with one tile each there is nothing to hide behind -- 4M PCISPH 2520 -> 2355 M particle-steps/s at 8x),
  // and not on slab ranks (a rank of 16M / 8 measured 0.42 -> 0.43 ms per step).  The host knows the particle count, not
  // the number of non-empty tiles: ~480 particles per tile.
  // Measured: wins at 16M (+3.5 % lattice, +4 % developed) and 64M (+4 % PCISPH), loses at 4M (PCISPH 2530 -> 2410 even at
  // 4 tiles per workgroup: every workgroup's first tile is fetched in the open) -- so only from 8M particles on.
  int g = 256 * blocks_per_cu;
  if (!lists && h->c.slab_axis < 0 && h->n >= 8000000) {
    const long long by_tiles = (long long)h->n / (480 * 4);
    const long long want = (long long)g * h->grid_oversub;
    g = (int)std::max<long long>(g, std::min(want, by_tiles));
  }
  // (tests: DSL_PERSISTENT_BLOCKS caps the grid, so that a small scene makes every workgroup walk MANY tiles -- the
  // tile-to-tile hand-over inside a workgroup is where the double-buffered loops can go wrong, and a test scene of a
  // few hundred tiles otherwise gives each workgroup one)
  if (h->max_persistent_blocks > 0 && g > h->max_persistent_blocks) g = h->max_persistent_blocks;
  if (g > h->tg.ntiles) g = h->tg.ntiles;
  g = (g + 7) & ~7;  // the XCD walk deals tiles in eighths
  return g < 8 ? 8 : g;
}

// the tile queue's counters of one launch site (nullptr: the static walk); slab ranks keep the static walk -- their band and
// interior launches overlap on two streams and walk lists other than list 0
// (the density kernels keep the static walk: measured slower with the queue -- density walk 0.43 -> 0.55 ms for a register,
// pair sweep of the developed flow 0.90 -> 0.98 -- where the force kernels gain 3-5 %: profiles/r04_tile_queue.jsonl)
enum { kSiteForceList = 1, kSiteForceTiled = 3 };
// (and only from 8M particles on, like the oversubscribed grids: with a handful of tiles per workgroup the draws are not
// hidden -- 1M particles, 4 tiles per workgroup: force walk 0.063 -> 0.098 ms, profiles/r04_tile_queue_1m_4m.jsonl)
int* walk_ctr_of(dsl_handle* h, int site) {
  const bool on = h->tile_queue == 2 || (h->tile_queue == 1 && h->n >= 8000000);
  if (!(on && h->walk_ctr != nullptr && h->c.slab_axis < 0)) return nullptr;
  // (a host that is capturing this stream into a graph of its own gets the static walk: a replayed launch would find the
  // counter block its captured twin left behind, not the zeroed one -- the two blocks alternate per LAUNCH CALL)
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) return nullptr;
  // a site's two blocks of eight counters are used alternately: a launch leaves the block it does not use at zero
  // (kernels_tiled.hpp: TileQueue).  Called ONCE per step and site.
  h->walk_parity[site] ^= 1;
  return h->walk_ctr + 16 * site + 8 * h->walk_parity[site];
}

int density_pass(dsl_handle* h) {
  const DevConsts& c = h->c;
  CSoa3 p = cpos(h);
  if (h->prm.math_mode == DSL_MATH_FAST) {
    // both instantiations are launched; the one the tile statistics of this build do not ask for
    // returns at once (kernels_tiled.hpp: share_wanted)
    int rc = timed(h, DSL_K_DENSITY, [&] {
#define DSL_LAUNCH_DENSITY(KERNEL)                                                                                   \
  hipLaunchKernelGGL(KERNEL, dim3(persistent_grid(h, 4)), dim3(kTBlock), 0, h->stream, c, h->tg, h->tile_desc_of, h->n_tiles, \
                     h->tile_desc, h->cell_start, bnd_of(h), p, h->rho, h->pterm, h->nmask, h->cap)
      if (h->density_pair) {  // two targets per lane (kernels_tiled.hpp: k_density_pair), 256-thread workgroups, four per CU
#define DSL_LAUNCH_PAIR(KERNEL)                                                                                      \
  hipLaunchKernelGGL(KERNEL, dim3(persistent_grid(h, 8)), dim3(kPBlock), 0, h->stream, c, h->tg, h->tile_desc_of, h->n_tiles, \
                     h->tile_desc, h->cell_start, bnd_of(h), p, h->rho, h->pterm, h->nmask, h->cap)
        if (c.slab_axis < 0) DSL_LAUNCH_PAIR(k_density_pair<false>);
        DSL_LAUNCH_PAIR(k_density_pair<true>);
#undef DSL_LAUNCH_PAIR
        return;
      }
      if (c.slab_axis < 0) DSL_LAUNCH_DENSITY(k_density_tiled<false>);
      DSL_LAUNCH_DENSITY(k_density_tiled<true>);
#undef DSL_LAUNCH_DENSITY
    });
    if (rc) return rc;
    h->dens_fresh = h->dens_held = true;
    h->masks_valid = true;  // until positions or the slot order change
    return DSL_OK;
  }
  const bool xt = exact_tiled(h);
  int rc = timed(h, DSL_K_DENSITY, [&] {
    if (xt)
      hipLaunchKernelGGL((k_density_tiled<false, true>), dim3(persistent_grid(h, 2)), dim3(kTBlock), 0, h->stream, c, h->tg,
                         h->tile_desc_of, h->n_tiles, h->tile_desc, h->cell_start, bnd_of(h), p, h->rho, h->pterm, h->nmask, h->cap);
    else
      by_math(h, [&](auto fast) {
        hipLaunchKernelGGL((k_density<decltype(fast)::value>), dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, c,
                           neigh(h), bnd_of(h), p, h->rho, h->pterm);
      });
  });
  if (rc) return rc;
  h->dens_fresh = h->dens_held = true;
  if (xt) h->masks_valid = true;
  return DSL_OK;
}

// part 0: every tile, then the state flips to the integrated arrays.  Split slab step:
// part 1 = band tiles only (state unchanged, dsl_slab_pack_band reads the output arrays next),
// part 2 = the remaining tiles, then the flip.
int force_integrate(dsl_handle* h, int part = 0) {
  DevConsts c = h->c;
  c.split_part = part;
  if (part == 2) {  // an interior particle that reaches a packed band moved further than the margin
    c.chk_lo = c.slab_lo + h->split_width;
    c.chk_hi = c.slab_hi - h->split_width;
  }
  // tile lists (k_tile_list): single domain 0; slab mode 4 = owning tiles (1 / 2 for the split
  // step's band / interior launch) plus 3 = ghost-only tiles, visited by one launch per step
  const bool slab = c.slab_axis >= 0;
  const int la = !slab ? 0 : (part == 0 ? 4 : part);
  const int* tiles = h->tile_desc_of + (size_t)la * h->tg.ntiles;
  const int* n_tiles = h->n_tiles + la;
  const int* gtiles = (slab && part != 2) ? h->tile_desc_of + (size_t)3 * h->tg.ntiles : nullptr;
  const int* n_gtiles = h->n_tiles + 3;
  CSoa3 p = cpos(h), v = cvel(h), f = cfrc(h);
  const int o = h->cur_pv ^ 1;
  Soa3 po = mpos(h, o), vo = mvel(h, o);
  const int uni = h->forces_uniform ? 1 : 0;
  const bool G = c.wcsph_pressure_force != 0, V = c.wcsph_viscosity != 0;
  int rc = DSL_OK;
  if (use_tiled(h)) {
    rc = timed(h, DSL_K_FORCE_INTEGRATE, [&] {
      int* wq = walk_ctr_of(h, kSiteForceTiled);
      int gsz = persistent_grid(h, 2, wq != nullptr);
      // Two of these workgroups fill a CU's vector registers, and a persistent grid keeps them
      // filled until it ends: the band pack and the transfer kernels that are meant to run UNDER
      // the interior launch would not get a wave in before it is over (measured: a 7 us kernel
      // took 83 us).  The interior launch therefore leaves one workgroup slot free on an eighth
      // of the CUs (+6 % on its own time).
      if (part == 2 && gsz >= 64) gsz = (gsz - gsz / 8) & ~7;
      dim3 g(gsz), b(kTBlock);
#define DSL_LAUNCH_FT4(GG, VV, XX, SS, HH)                                                                       \
  if (!(SS) && wq != nullptr) DSL_LAUNCH_FT5(GG, VV, XX, false, HH, true);                                       \
  else DSL_LAUNCH_FT5(GG, VV, XX, SS, HH, false)
#define DSL_LAUNCH_FT5(GG, VV, XX, SS, HH, QQ)                                                                   \
  hipLaunchKernelGGL((k_force_integrate_tiled<GG, VV, kOutIntegrate, XX, SS, HH, false, QQ>), g, b, 0, h->stream, c, \
                     h->tg, tiles, n_tiles, gtiles, n_gtiles, h->tile_desc, h->cell_start, p, v, h->rho, h->pterm, f, uni,   \
                     po, vo, h->dstats, h->masks_valid ? h->nmask : nullptr, h->cap, ((XX) || (SS)) ? nullptr : h->n_tiles, bnd_of(h), \
                     Soa3{nullptr, nullptr, nullptr}, wq)
  // (the XSPH / cohesion variant and the slab variant -- a slab always has half-empty ghost tiles -- exist
  // as the pass-sharing instantiation only; of the other two the device picks: kernels_tiled.hpp, share_wanted)
#define DSL_LAUNCH_FT3(GG, VV, XX, SS)                                        \
  do {                                                                        \
    if ((XX) || (SS)) DSL_LAUNCH_FT4(GG, VV, XX, SS, true);                   \
    else {                                                                    \
      DSL_LAUNCH_FT4(GG, VV, false, SS, false);                               \
      DSL_LAUNCH_FT4(GG, VV, false, SS, true);                                \
    }                                                                         \
  } while (0)
#define DSL_LAUNCH_FT2(GG, VV, XX)                     \
  do {                                                \
    if (c.slab_axis >= 0) DSL_LAUNCH_FT3(GG, VV, XX, true); \
    else DSL_LAUNCH_FT3(GG, VV, XX, false);           \
  } while (0)
#define DSL_LAUNCH_FT(GG, VV)                 \
  do {                                        \
    if (XS) DSL_LAUNCH_FT2(GG, VV, true);     \
    else DSL_LAUNCH_FT2(GG, VV, false);       \
  } while (0)
      const bool XS = c.xsph_eps != 0.0f || c.st_kappa != 0.0f;
      if (G && V) DSL_LAUNCH_FT(true, true);
      else if (G) DSL_LAUNCH_FT(true, false);
      else if (V) DSL_LAUNCH_FT(false, true);
      else DSL_LAUNCH_FT(false, false);
#undef DSL_LAUNCH_FT5
#undef DSL_LAUNCH_FT4
#undef DSL_LAUNCH_FT3
#undef DSL_LAUNCH_FT2
#undef DSL_LAUNCH_FT
    });
  } else if (exact_tiled(h) && h->masks_valid) {
    rc = timed(h, DSL_K_FORCE_INTEGRATE, [&] {
      dim3 g(persistent_grid(h, 2)), b(kTBlock);
      const bool XS = c.xsph_eps != 0.0f || c.st_kappa != 0.0f;
#define DSL_LAUNCH_FX(GG, VV, XX)                                                                                  \
  hipLaunchKernelGGL((k_force_integrate_tiled<GG, VV, kOutIntegrate, XX, false, false, true>), g, b, 0, h->stream, c, \
                     h->tg, tiles, n_tiles, gtiles, n_gtiles, h->tile_desc, h->cell_start, p, v, h->rho, h->pterm, f, uni, po, vo, \
                     h->dstats, h->nmask, h->cap, nullptr, bnd_of(h), Soa3{nullptr, nullptr, nullptr})
      if (XS) {
        if (G && V) DSL_LAUNCH_FX(true, true, true);
        else if (G) DSL_LAUNCH_FX(true, false, true);
        else if (V) DSL_LAUNCH_FX(false, true, true);
        else DSL_LAUNCH_FX(false, false, true);
      } else {
        if (G && V) DSL_LAUNCH_FX(true, true, false);
        else if (G) DSL_LAUNCH_FX(true, false, false);
        else if (V) DSL_LAUNCH_FX(false, true, false);
        else DSL_LAUNCH_FX(false, false, false);
      }
#undef DSL_LAUNCH_FX
    });
  } else
  rc = timed(h, DSL_K_FORCE_INTEGRATE, [&] {
    by_math(h, [&](auto fast) {
      constexpr bool FAST = decltype(fast)::value;
      dim3 g(grid_for(launch_n(h))), b(kBlock);
#define DSL_LAUNCH_FI(GG, VV)                                                                                    \
  hipLaunchKernelGGL((k_force_integrate<FAST, GG, VV>), g, b, 0, h->stream, c, neigh(h), bnd_of(h), p, v, h->rho,          \
                     h->pterm, f, uni, po, vo, h->dstats)
      if (G && V) DSL_LAUNCH_FI(true, true);
      else if (G) DSL_LAUNCH_FI(true, false);
      else if (V) DSL_LAUNCH_FI(false, true);
      else DSL_LAUNCH_FI(false, false);
#undef DSL_LAUNCH_FI
    });
  });
  if (rc) return rc;
  if (part == 1) return DSL_OK;
  h->cur_pv = o;
  h->masks_valid = false;    // positions moved
  h->forces_uniform = true;  // Update resets every force to force_reset (fluid.go:193)
  h->press_zero = true;      // ... and every pressure to 0 (fluid.go:192)
  h->grid_valid = false;     // positions moved
  return DSL_OK;
}

int gradient_pass(dsl_handle* h, int honour_done) {
  const DevConsts& c = h->c;
  CSoa3 p = cpos(h);
  Soa3 f = mfrc(h);
  return timed(h, DSL_K_GRADIENT, [&] {
    by_math(h, [&](auto fast) {
      hipLaunchKernelGGL((k_gradient<decltype(fast)::value>), dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, c,
                         neigh(h), bnd_of(h), p, h->rho, h->pterm, f, h->dstats, honour_done);
    });
  });
}

// with_xs: also add the cohesion force and store the XSPH correction (PCISPH with the
// build-defined terms)
int viscous_pass(dsl_handle* h, int with_xs = 0) {
  const DevConsts& c = h->c;
  CSoa3 p = cpos(h), v = cvel(h);
  Soa3 f = mfrc(h);
  Soa3 xs{h->xsph[0], h->xsph[1], h->xsph[2]};
  return timed(h, DSL_K_VISCOUS, [&] {
    by_math(h, [&](auto fast) {
      hipLaunchKernelGGL((k_viscous<decltype(fast)::value>), dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, c,
                         neigh(h), bnd_of(h), p, v, h->rho, f, with_xs, xs);
    });
  });
}

// use_xs: Update advects positions with v + the XSPH correction stored by the viscous sweep
int update_pass(dsl_handle* h, bool use_xs = false) {
  const DevConsts& c = h->c;
  CSoa3 xs{use_xs ? h->xsph[0] : nullptr, use_xs ? h->xsph[1] : nullptr, use_xs ? h->xsph[2] : nullptr};
  Soa3 p = mpos(h, h->cur_pv), v = mvel(h, h->cur_pv);
  CSoa3 f = cfrc(h);
  const int uni = h->forces_uniform ? 1 : 0;
  int rc = timed(h, DSL_K_UPDATE, [&] {
    hipLaunchKernelGGL(k_update, dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, c, bnd_of(h), p, v, f, uni, h->dstats,
                       xs);
  });
  if (rc) return rc;
  h->forces_uniform = true;
  h->press_zero = true;
  h->grid_valid = false;
  h->masks_valid = false;
  return DSL_OK;
}

// exact live count on the host (slab mode: one small device read)
int host_count(dsl_handle* h, int* out) {
  if (!h->c.n_ptr) {
    *out = h->n;
    return DSL_OK;
  }
  HIP_TRY(h, hipMemcpyAsync(out, h->dn, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n = *out;
  return DSL_OK;
}

struct BufInfo {
  int comps;
};
bool buf_info(int buffer, BufInfo& bi) {
  switch (buffer) {
    case DSL_BUF_POSITIONS:
    case DSL_BUF_VELOCITIES:
    case DSL_BUF_FORCES:
    case DSL_BUF_PCI_POSITIONS:
    case DSL_BUF_PCI_VELOCITIES:
      bi.comps = 3;
      return true;
    case DSL_BUF_DENSITIES:
    case DSL_BUF_PRESSURES:
      bi.comps = 1;
      return true;
    default:
      return false;
  }
}

void free_all(dsl_handle* h) {
  for (int w = 0; w < 2; ++w) {
    for (int k = 0; k < 6; ++k) {
      (void)hipFree(h->pv[w][k]);
      (void)hipFree(h->pci[w][k]);
    }
    for (int k = 0; k < 3; ++k) (void)hipFree(h->frc[w][k]);
    if (w == 0)
      for (int k = 0; k < 3; ++k) {
        (void)hipFree(h->gterm[k]);
        (void)hipFree(h->xsph[k]);
      }
    (void)hipFree(h->ids[w]);
  }
  (void)hipFree(h->rho);
  (void)hipFree(h->pterm);
  (void)hipFree(h->press);
  (void)hipFree(h->scratch1);
  (void)hipFree(h->rank);
  (void)hipFree(h->sort_keys);
  (void)hipFree(h->cell_keys_alloc);
  (void)hipFree(h->sort_work);
  (void)hipFree(h->unordered);
  (void)hipFree(h->cell_count);
  (void)hipFree(h->cell_start);
  (void)hipFree(h->block_sums);
  (void)hipFree(h->qcount);
  (void)hipFree(h->qstart);
  (void)hipFree(h->qsums);
  (void)hipFree(h->qrec);
  (void)hipFree(h->qtiles);
  (void)hipFree(h->n_qtiles);
  (void)hipFree(h->qtile_desc);
  (void)hipFree(h->qrows);
  (void)hipFree(h->qslot);
  (void)hipFree(h->pci_drift);
  if (h->pci_drift_host) (void)hipHostFree(h->pci_drift_host);
  if (h->ev_drift) (void)hipEventDestroy(h->ev_drift);
  (void)hipFree(h->skin_state);
  (void)hipFree(h->walk_ctr);
  (void)hipFree(h->lists);
  for (int k = 0; k < 6; ++k) (void)hipFree(h->pvz[k]);
  for (int k = 0; k < 3; ++k) (void)hipFree(h->pvr[k]);
  (void)hipFree(h->stage);
  (void)hipFree(h->dstats);
  (void)hipFree(h->dcounter);
  (void)hipFree(h->dn);
  (void)hipFree(h->pack_counts);
  (void)hipFree(h->tiles);
  (void)hipFree(h->tile_desc_of);
  (void)hipFree(h->tile_desc);
  (void)hipFree(h->n_tiles);
  (void)hipFree(h->hashv);
  (void)hipFree(h->bucket_of);
  (void)hipFree(h->lsh_table);
  (void)hipFree(h->lsh_len);
  (void)hipFree(h->lsh_samples);
  (void)hipFree(h->nmask);
  for (hipEvent_t e : h->pool) (void)hipEventDestroy(e);
  for (auto& v : h->pending)
    for (auto& pr : v) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  if (h->ev_band) (void)hipEventDestroy(h->ev_band);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

// (every pointer is checked by itself: a retry after a failed allocation neither leaks nor re-allocates)
int alloc_pci(dsl_handle* h) {
  for (int k = 0; k < 3; ++k)
    if (!h->gterm[k])
      if (int rc = dev_alloc(h, &h->gterm[k], (size_t)h->cap)) return rc;
  for (int k = 0; k < 3; ++k)
    if (!h->xsph[k])
      if (int rc = dev_alloc(h, &h->xsph[k], (size_t)h->cap)) return rc;
  for (int w = 0; w < 2; ++w)
    for (int k = 0; k < 6; ++k)
      if (!h->pci[w][k])
        if (int rc = dev_alloc(h, &h->pci[w][k], (size_t)h->cap)) return rc;
  if (!h->pci_drift)
    if (int rc = dev_alloc(h, &h->pci_drift, (size_t)512)) return rc;
  if (!h->pci_drift_host) HIP_TRY(h, hipHostMalloc((void**)&h->pci_drift_host, 512 * sizeof(unsigned int), hipHostMallocDefault));
  if (!h->ev_drift) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_drift, hipEventDisableTiming));
  return DSL_OK;
}

bool pci_tiled(const dsl_handle* h);
// the query bins are allocated when the binned form is first used (every array zero-filled by dev_alloc; k_scan_apply
// leaves the histogram clean behind every use)
int alloc_query_bins(dsl_handle* h) {
  int rc = DSL_OK;
  if (!h->qcount && (rc = dev_alloc(h, &h->qcount, (size_t)h->ncell_pad))) return rc;
  if (!h->qstart && (rc = dev_alloc(h, &h->qstart, (size_t)h->ncell_pad))) return rc;
  if (!h->qsums && (rc = dev_alloc(h, &h->qsums, (size_t)h->nscan))) return rc;
  if (!h->qrec && (rc = dev_alloc(h, &h->qrec, (size_t)h->cap))) return rc;
  if (pci_tiled(h)) {  // the LDS-tiled sweep over query tiles (kernels_tiled.hpp: k_pci_density_qtiled)
    if (!h->qtiles && (rc = dev_alloc(h, &h->qtiles, (size_t)h->tg.ntiles))) return rc;
    if (!h->n_qtiles && (rc = dev_alloc(h, &h->n_qtiles, (size_t)8))) return rc;
    if (!h->qtile_desc && (rc = dev_alloc(h, &h->qtile_desc, (size_t)std::min(h->tg.ntiles, h->cap) * kMetaInts))) return rc;
    if (h->pci_qrows && h->pci_qpair && h->pci_qtiled && !h->qrows && !side_array_fits(h, (size_t)h->ncell * kQueryRow * sizeof(float4)))
      h->pci_qrows = false;  // (512 bytes per grid cell: 2.1 GB for the 4M scene's box, 33 GB for the 64M scene's)
    if (h->pci_qrows && h->pci_qpair && h->pci_qtiled && !h->qrows) {
      if (dev_alloc(h, &h->qrows, (size_t)h->ncell * kQueryRow) != DSL_OK) {
        h->qrows = nullptr;
        h->pci_qrows = false;
        h->err.clear();
        (void)hipGetLastError();  // (since ROCm 7.0 the last NON-SUCCESS error sticks: the next launch check would report this one)
      }
    }
    // (a record index is an int: 2^31 / kQueryRow grid cells)
    if (h->qrows && h->pci_qincr && !h->qslot && (size_t)h->ncell * kQueryRow < ((size_t)1 << 31))
      if ((rc = dev_alloc(h, &h->qslot, (size_t)h->cap))) return rc;
  }
  return DSL_OK;
}

}  // namespace

#include "slab_link.hpp"

extern "C" {

const char* dsl_version(void) { return "dslsph 0.1 (gfx950)"; }

const char* dsl_last_error(dsl_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dsl_params_reference(dsl_params* p, int n3) {
  if (!p || n3 <= 0) return fail(nullptr, DSL_ERR_INVALID, "dsl_params_reference: bad argument");
  std::memset(p, 0, sizeof(*p));
  p->struct_size = sizeof(dsl_params);
  p->abi_version = DSL_ABI_VERSION;
  const int num = n3 * n3 * n3;                  // fluid.go:51
  p->n_particles = num;
  p->n_boundary = 0;
  p->lsh_buckets = 255;                          // fluid.go:64
  p->lsh_bucket_size = (int)((float)(num / 255) * 1.5f);  // lsh.go:33
  p->dt = 0.01f;                                 // fluid.go:112
  p->mass = 1.0f;                                // fluid.go:56
  p->delta = 0.0f;
  p->max_vel = 0.0f;
  p->h = 1.0f;                                   // fluid.go:48
  p->ref_density = (float)num / 8.0f;            // fluid.go:55, point-grid.go:44-46
  p->mu = 1.3059f;                               // fluid.go:18
  p->eos_w = 2.15f;                              // model.go:94
  p->eos_gamma = 7.16f;                          // model.go:93
  p->eos_d0_grad = 87.0f;                        // model.go:41
  p->pressure_sign = 1.0f;
  p->visc_running_mass = 1;
  p->force_reset[1] = -9.81f * p->mass;          // fluid.go:193
  p->external[1] = -9.81f;                       // wcsph.go:19
  p->pci_max_iters = 5;                          // pcisph_darwin.go:49
  p->pci_max_error = 0.01f;                      // pcisph_darwin.go:50
  for (int a = 0; a < 3; ++a) {
    p->box_min[a] = -1.0f;
    p->box_max[a] = 1.0f;
    p->grid_min[a] = -4.0f;
    p->grid_max[a] = 4.0f;
  }
  p->neigh_mode = DSL_NEIGH_GRID;
  p->math_mode = DSL_MATH_EXACT;
  return DSL_OK;
}

int dsl_create(const dsl_params* params, int device, dsl_handle** out) {
  if (!params || !out) return fail(nullptr, DSL_ERR_INVALID, "dsl_create: null argument");
  *out = nullptr;
  if (params->struct_size != sizeof(dsl_params) || params->abi_version != DSL_ABI_VERSION)
    return fail(nullptr, DSL_ERR_INVALID, "dsl_create: dsl_params size/version mismatch (use dsl_params_reference)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, DSL_ERR_DEVICE, "dsl_create: no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, DSL_ERR_INVALID, "dsl_create: device ordinal out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, DSL_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
  dsl_handle* h = new (std::nothrow) dsl_handle();
  if (!h) return fail(nullptr, DSL_ERR_NOMEM, "dsl_create: out of host memory");
  h->device = device;
  h->prm = *params;
  int rc = make_consts(h, h->prm, h->c);
  if (rc) {
    g_create_error = h->err;
    delete h;
    return rc;
  }
  h->n = h->c.n;
  h->nb = params->n_boundary;
  h->cap = params->capacity > 0 ? params->capacity : h->n;
  h->ncell = h->c.ncell;
  h->ncell_pad = ((h->ncell + 1 + kScanTile - 1) / kScanTile) * kScanTile;
  h->nscan = h->ncell_pad / kScanTile;
  const size_t n = (size_t)h->cap;
  auto bail = [&](int code) {
    g_create_error = h->err;
    free_all(h);
    delete h;
    return code;
  };
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) {
    h->err = "hipStreamCreate failed";
    return bail(DSL_ERR_DEVICE);
  }
  h->stream = h->own_stream;
  for (int w = 0; w < 2; ++w) {
    for (int k = 0; k < 6; ++k)
      if ((rc = dev_alloc(h, &h->pv[w][k], n))) return bail(rc);
    for (int k = 0; k < 3; ++k)
      if ((rc = dev_alloc(h, &h->frc[w][k], n))) return bail(rc);
    if ((rc = dev_alloc(h, &h->ids[w], n))) return bail(rc);
  }
  if ((rc = dev_alloc(h, &h->rho, n)) || (rc = dev_alloc(h, &h->pterm, n)) || (rc = dev_alloc(h, &h->press, n)) ||
      (rc = dev_alloc(h, &h->scratch1, n)) || (rc = dev_alloc(h, &h->rank, n)) || (rc = dev_alloc(h, &h->sort_keys, n)) ||
      (rc = dev_alloc(h, &h->sort_work, n)) || (rc = dev_alloc(h, &h->unordered, (size_t)h->ncell_pad / 32)) ||
      (rc = dev_alloc(h, &h->cell_count, (size_t)h->ncell_pad)) ||
      (rc = dev_alloc(h, &h->cell_start, (size_t)h->ncell_pad)) ||
      (rc = dev_alloc(h, &h->block_sums, (size_t)h->nscan)) || (rc = dev_alloc(h, &h->stage, n * 3)) ||
      (rc = dev_alloc(h, &h->dstats, 1)) || (rc = dev_alloc(h, &h->dcounter, 8)) || (rc = dev_alloc(h, &h->dn, 16)) ||
      (rc = dev_alloc(h, &h->walk_ctr, 256)) ||
      (rc = dev_alloc(h, &h->pack_counts, (size_t)4 * ((n + kPackChunk - 1) / kPackChunk))))
    return bail(rc);
  // the sort keeps these two clean between builds (k_scan_apply, k_tile_list)
  if (hipMemsetAsync(h->cell_count, 0, sizeof(int) * (size_t)h->ncell_pad, h->stream) != hipSuccess ||
      hipMemsetAsync(h->unordered, 0, sizeof(unsigned int) * (size_t)(h->ncell_pad / 32), h->stream) != hipSuccess) {
    h->err = "hipMemsetAsync failed";
    return bail(DSL_ERR_DEVICE);
  }
  h->lsh = params->neigh_mode == DSL_NEIGH_LSH_REF;
  // (the cells' key rows of the one-pass in-cell ordering: DSL_OPT_CELL_KEYS = 0 leaves them unused)
  if (!h->lsh && side_array_fits(h, (size_t)h->ncell_pad * kCellKeys * sizeof(int)) &&
      (rc = dev_alloc(h, &h->cell_keys_alloc, (size_t)h->ncell_pad * kCellKeys)))
    return bail(rc);
  h->cell_keys = h->cell_keys_alloc;
  if (h->lsh) {
    const int B = params->lsh_buckets;
    h->lsh_cap = params->lsh_bucket_size > kLshSamples ? params->lsh_bucket_size : kLshSamples;
    h->lsh_bits = 8;  // lsh.Allocate(num, 255, 8, ..) fluid.go:64
    if ((rc = dev_alloc(h, &h->hashv, 32 * 3)) || (rc = dev_alloc(h, &h->bucket_of, n)) ||
        (rc = dev_alloc(h, &h->lsh_table, (size_t)B * h->lsh_cap)) || (rc = dev_alloc(h, &h->lsh_len, (size_t)B)) ||
        (rc = dev_alloc(h, &h->lsh_samples, (size_t)B * kLshSamples)))
      return bail(rc);
    if (hipMemsetAsync(h->hashv, 0, 32 * 3 * sizeof(float), h->stream) != hipSuccess ||
        hipMemsetAsync(h->lsh_table, 0, (size_t)B * h->lsh_cap * sizeof(int), h->stream) != hipSuccess) {
      h->err = "hipMemsetAsync failed";
      return bail(DSL_ERR_DEVICE);
    }
  }
  if ((rc = set_tile_grid(h))) return bail(rc);
  // (room for the skin step's grid as well: wider cells, but tiles of 3 cell layers in z -- at most 4/3 of these tiles)
  h->alloc_ntiles = h->tg.tnx * h->tg.tny * ((h->c.dims[2] + 2) / 3);
  if (h->alloc_ntiles < h->tg.ntiles) h->alloc_ntiles = h->tg.ntiles;
  if ((rc = dev_alloc(h, &h->tiles, (size_t)kTileLists * h->alloc_ntiles)) || (rc = dev_alloc(h, &h->n_tiles, 8))) return bail(rc);
  // a non-empty tile holds a particle: at most min(tiles, capacity) tables (1.5 KB each)
  if ((rc = dev_alloc(h, &h->tile_desc_of, (size_t)kTileLists * h->alloc_ntiles)) ||
      (rc = dev_alloc(h, &h->tile_desc, (size_t)std::min(h->alloc_ntiles, h->cap) * kMetaInts)))
    return bail(rc);
  if (!h->lsh && (rc = dev_alloc(h, &h->nmask, (size_t)kMaskWords * n))) return bail(rc);
  // NewParticleArray zero-fills every slice (particle_array.go:18-33)
  hipError_t me = hipSuccess;
  for (int k = 0; k < 6 && me == hipSuccess; ++k) me = hipMemsetAsync(h->pv[0][k], 0, n * sizeof(float), h->stream);
  for (int k = 0; k < 3 && me == hipSuccess; ++k) me = hipMemsetAsync(h->frc[0][k], 0, n * sizeof(float), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(h->rho, 0, n * sizeof(float), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(h->pterm, 0, n * sizeof(float), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(h->press, 0, n * sizeof(float), h->stream);
  if (me == hipSuccess) me = hipMemsetAsync(h->dstats, 0, sizeof(DevStats), h->stream);
  if (me != hipSuccess) {
    h->err = std::string("hipMemsetAsync: ") + hipGetErrorString(me);
    return bail(DSL_ERR_DEVICE);
  }
  hipLaunchKernelGGL(k_iota, dim3(grid_for(h->cap)), dim3(kBlock), 0, h->stream, h->cap, h->ids[0]);
  DevStats init{};
  // (params->max_vel, the reference's initial maxVel, is folded in by dsl_get_stats: the device
  // counters hold squared magnitudes)
  if (hipMemcpyAsync(h->dstats, &init, sizeof(init), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) {
    h->err = "device initialisation failed";
    return bail(DSL_ERR_DEVICE);
  }
  h->forces_uniform = false;  // uploaded / zero forces are honoured until the first Update
  h->press_zero = true;
  // the skin step is the default where it pays: DSL_MATH_FAST, enough particles for a step to outweigh the ~40 us of
  // gated launches it adds (dsl_set_option(DSL_OPT_SKIN) overrides)
  if (params->math_mode == DSL_MATH_FAST && h->n >= kSkinDefaultParticles) h->skin = kSkinDefault;
  *out = h;
  return DSL_OK;
}

int dsl_destroy(dsl_handle* h) {
  if (!h) return DSL_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
#ifdef DSL_DIAG_STAMPS
  {
    unsigned long long d[32] = {};
    (void)hipMemcpyFromSymbol(d, HIP_SYMBOL(dsl::g_diag), sizeof(d));
    std::fprintf(stderr,
                 "[dsl diag] wave-0 clocks. force: setup %llu staging %llu (mask requests %llu, load issue %llu, wait+LDS "
                 "%llu, barrier %llu) target-prologue %llu sweep %llu | density: setup %llu staging %llu (load issue %llu, "
                 "wait+LDS %llu) sweep %llu\n",
                 d[0], d[1], d[10], d[8], d[9], d[11], d[2], d[3], d[4], d[5], d[12], d[13], d[6]);
    std::fprintf(stderr, "[dsl diag] density staging: commit %llu, row table + barrier %llu, scan + record requests %llu\n", d[14],
                 d[15], d[16]);
  }
#endif
  (void)dsl_slab_detach(h);
  free_all(h);
  delete h;
  return DSL_OK;
}

int dsl_set_params(dsl_handle* h, const dsl_params* p) {
  CHECK_HANDLE(h);
  if (!p || p->struct_size != sizeof(dsl_params) || p->abi_version != DSL_ABI_VERSION)
    return fail(h, DSL_ERR_INVALID, "dsl_set_params: dsl_params size/version mismatch");
  DevConsts c{};
  if (int rc = make_consts(h, *p, c)) return rc;
  if (p->n_particles != h->prm.n_particles || p->capacity != h->prm.capacity || c.ncell != h->c.ncell ||
      c.dims[0] != h->c.dims[0] || c.dims[1] != h->c.dims[1] || c.dims[2] != h->c.dims[2] || c.h != h->c.h)
    return fail(h, DSL_ERR_INVALID, "dsl_set_params: n_particles, capacity, h and the grid box are fixed at creation");
  if (p->n_boundary != h->nb)
    return fail(h, DSL_ERR_INVALID,
                "dsl_set_params: n_boundary must be the current boundary count (dsl_get_params after dsl_add_boundary_particles)");
  c.n = h->c.n;  // live count (slab mode changes it)
  c.slab_axis = h->c.slab_axis;
  c.slab_lo = h->c.slab_lo;
  c.slab_hi = h->c.slab_hi;
  c.n_ptr = h->c.n_ptr;
  c.split_cl = h->c.split_cl;
  c.split_ch = h->c.split_ch;
  c.own_c0 = h->c.own_c0;
  c.own_c1 = h->c.own_c1;
  if (std::memcmp(c.reset, h->c.reset, sizeof(c.reset)) != 0 && h->forces_uniform) {
    if (int rc = materialise_forces(h)) return rc;  // keep the old implicit value
  }
  // derived per-particle state: rho scales with the mass, P/rho^2 follows the EOS constants
  const bool mass_changed = c.mass != h->c.mass;
  const bool eos_changed = c.eos_wg != h->c.eos_wg || c.eos_gamma != h->c.eos_gamma || c.eos_d0_grad != h->c.eos_d0_grad;
  h->prm = *p;
  h->c = c;
  if (mass_changed) {
    // densities (and P/rho^2) are stale -- like the reference's, which keeps the old value on the same particle until
    // the next DensityAll: they stay held (a sort still carries them), they are just no longer fresh
    h->dens_fresh = false;
  } else if (eos_changed && h->dens_held) {
    dim3 g(grid_for(launch_n(h))), b(kBlock);
    by_math(h, [&](auto fast) {
      hipLaunchKernelGGL((k_pterm<decltype(fast)::value>), g, b, 0, h->stream, h->c, bnd_of(h), h->rho, h->pterm);
    });
    HIP_TRY(h, hipGetLastError());
  }
  return DSL_OK;
}

int dsl_get_params(dsl_handle* h, dsl_params* out) {
  CHECK_HANDLE(h);
  if (!out) return fail(h, DSL_ERR_INVALID, "null out");
  *out = h->prm;
  return DSL_OK;
}

int dsl_set_stream(dsl_handle* h, void* s) {
  CHECK_HANDLE(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)s;  // NULL is HIP's default (null) stream, a perfectly valid choice
  return DSL_OK;
}
int dsl_use_own_stream(dsl_handle* h) {
  CHECK_HANDLE(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return DSL_OK;
}

int dsl_upload(dsl_handle* h, int buffer, const float* host, size_t count) {
  CHECK_HANDLE(h);
  BufInfo bi;
  if (!host || !buf_info(buffer, bi)) return fail(h, DSL_ERR_INVALID, "dsl_upload: bad buffer id or null pointer");
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  // positions hold Total() particles, everything else N() (particle_array.go:18-33)
  const bool with_boundary = h->nb > 0;
  const int limit = (buffer == DSL_BUF_POSITIONS || !with_boundary) ? ncur : n_fluid_of(h);
  const size_t n = (size_t)ncur;
  if (count != (size_t)limit * bi.comps)
    return fail(h, DSL_ERR_INVALID,
                with_boundary ? "dsl_upload: count does not match the buffer size (positions: N + Nboundary particles, others: N)"
                              : "dsl_upload: count does not match the buffer size");
  if (h->ids_global) return fail(h, DSL_ERR_INVALID, "dsl_upload: host order is gone after dsl_set_ids");
  HIP_TRY(h, hipMemcpyAsync(h->stage, host, count * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const int* ids = h->ids[h->cur_ids];
  dim3 g(grid_for(h->n)), b(kBlock);
  switch (buffer) {
    case DSL_BUF_POSITIONS: {
      Soa3 p = mpos(h, h->cur_pv);
      const int zero_id = with_boundary ? n_fluid_of(h) : -1;  // Get(N()) is the zero particle (particle_array.go:98,107)
      if (with_boundary) std::memcpy(h->b0_pos, host + (size_t)3 * n_fluid_of(h), sizeof(h->b0_pos));
      hipLaunchKernelGGL(k_unpack3, g, b, 0, h->stream, h->n, h->stage, ids, p.x, p.y, p.z, limit, zero_id);
      h->grid_valid = false;
      h->masks_valid = false;
      break;
    }
    case DSL_BUF_VELOCITIES: {
      Soa3 v = mvel(h, h->cur_pv);
      hipLaunchKernelGGL(k_unpack3, g, b, 0, h->stream, h->n, h->stage, ids, v.x, v.y, v.z, limit, -1);
      break;
    }
    case DSL_BUF_FORCES: {
      Soa3 f = mfrc(h);
      hipLaunchKernelGGL(k_unpack3, g, b, 0, h->stream, h->n, h->stage, ids, f.x, f.y, f.z, limit, -1);
      h->forces_uniform = false;
      break;
    }
    case DSL_BUF_DENSITIES:
      hipLaunchKernelGGL(k_unpack1, g, b, 0, h->stream, h->n, h->stage, ids, h->rho, limit);
      // pterm must follow an uploaded density
      by_math(h, [&](auto fast) {
        hipLaunchKernelGGL((k_pterm<decltype(fast)::value>), g, b, 0, h->stream, h->c, bnd_of(h), h->rho, h->pterm);
      });
      h->dens_fresh = h->dens_held = true;
      break;
    case DSL_BUF_PRESSURES:
      hipLaunchKernelGGL(k_unpack1, g, b, 0, h->stream, h->n, h->stage, ids, h->press, limit);
      h->press_zero = false;
      break;
    case DSL_BUF_PCI_POSITIONS:
    case DSL_BUF_PCI_VELOCITIES: {
      if (int rc = alloc_pci(h)) return rc;
      if (!h->pci_active) {
        // pcisph_darwin.go:28-41: the state starts as a copy of positions/velocities
        for (int k = 0; k < 6; ++k)
          HIP_TRY(h, hipMemcpyAsync(h->pci[h->cur_pci][k], h->pv[h->cur_pv][k], n * sizeof(float),
                                    hipMemcpyDeviceToDevice, h->stream));
        h->pci_active = true;
      }
      Soa3 d = buffer == DSL_BUF_PCI_POSITIONS ? mpcip(h) : mpciv(h);
      hipLaunchKernelGGL(k_unpack3, g, b, 0, h->stream, h->n, h->stage, ids, d.x, d.y, d.z, limit, -1);
      break;
    }
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // host pointer is not retained (cgo rule)
  return DSL_OK;
}

int dsl_add_boundary_particles(dsl_handle* h, const float* host_positions, size_t count) {
  CHECK_HANDLE(h);
  if (!host_positions || count == 0 || count % 3 != 0)
    return fail(h, DSL_ERR_INVALID, "dsl_add_boundary_particles: count must be a positive multiple of 3");
  if (h->c.n_ptr) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_add_boundary_particles: not available in slab mode");
  if (h->ids_global) return fail(h, DSL_ERR_INVALID, "dsl_add_boundary_particles: particle ids were replaced (dsl_set_ids)");
  const int nb = (int)(count / 3);
  if ((long long)h->n + nb > h->cap)
    return fail(h, DSL_ERR_NOMEM, "dsl_add_boundary_particles: dsl_params.capacity leaves no room (needs n_particles + all boundary particles)");
  HIP_TRY(h, hipMemcpyAsync(h->stage, host_positions, count * sizeof(float), hipMemcpyHostToDevice, h->stream));
  const bool first = h->nb == 0;
  const int n_fluid = n_fluid_of(h);
  if (first) std::memcpy(h->b0_pos, host_positions, sizeof(h->b0_pos));
  Soa3 p = mpos(h, h->cur_pv), v = mvel(h, h->cur_pv);
  Soa3 pcip{nullptr, nullptr, nullptr}, pciv{nullptr, nullptr, nullptr};
  if (h->pci_active) {
    pcip = mpcip(h);
    pciv = mpciv(h);
  }
  hipLaunchKernelGGL(k_append_boundary, dim3(grid_for(nb)), dim3(kBlock), 0, h->stream, nb, h->n, h->n, n_fluid, h->stage,
                     p.x, p.y, p.z, v.x, v.y, v.z, h->ids[h->cur_ids], pcip, pciv, h->rho, h->pterm);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n += nb;
  h->nb += nb;
  h->c.n = h->n;
  h->prm.n_boundary += nb;
  h->grid_valid = false;
  h->masks_valid = false;
  // the fluid's densities are stale now (new neighbours) but still sit on their particles: the next sort carries
  // them, the appended slots read as Get() reads a boundary particle (rho = 0, P/rho^2 = 0/0)
  h->dens_fresh = false;
  return DSL_OK;
}

static int download_impl(dsl_handle* h, int buffer, float* host, size_t count, int sorted_order) {
  CHECK_HANDLE(h);
  BufInfo bi;
  if (!host || !buf_info(buffer, bi)) return fail(h, DSL_ERR_INVALID, "dsl_download: bad buffer id or null pointer");
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  const bool with_boundary = h->nb > 0;
  const int limit = (buffer == DSL_BUF_POSITIONS || !with_boundary) ? ncur : n_fluid_of(h);
  if (with_boundary && sorted_order && buffer != DSL_BUF_POSITIONS)
    return fail(h, DSL_ERR_UNSUPPORTED, "dsl_download_sorted: with boundary particles only positions have a slot-order image");
  if (count != (size_t)limit * bi.comps)
    return fail(h, DSL_ERR_INVALID,
                with_boundary ? "dsl_download: count does not match the buffer size (positions: N + Nboundary particles, others: N)"
                              : "dsl_download: count does not match the buffer size");
  if (h->ids_global && !sorted_order)
    return fail(h, DSL_ERR_INVALID, "dsl_download: host order is gone after dsl_set_ids; use dsl_download_sorted + dsl_download_ids");
  const int* ids = h->ids[h->cur_ids];
  dim3 g(grid_for(h->n)), b(kBlock);
  switch (buffer) {
    case DSL_BUF_POSITIONS: {
      CSoa3 p = cpos(h);
      hipLaunchKernelGGL(k_pack3, g, b, 0, h->stream, h->n, h->stage, ids, p.x, p.y, p.z, sorted_order);
      break;
    }
    case DSL_BUF_VELOCITIES: {
      CSoa3 v = cvel(h);
      hipLaunchKernelGGL(k_pack3, g, b, 0, h->stream, h->n, h->stage, ids, v.x, v.y, v.z, sorted_order);
      break;
    }
    case DSL_BUF_FORCES: {
      if (int rc = materialise_forces(h)) return rc;
      CSoa3 f = cfrc(h);
      hipLaunchKernelGGL(k_pack3, g, b, 0, h->stream, h->n, h->stage, ids, f.x, f.y, f.z, sorted_order);
      break;
    }
    case DSL_BUF_DENSITIES:
      hipLaunchKernelGGL(k_pack1, g, b, 0, h->stream, h->n, h->stage, ids, h->rho, sorted_order);
      break;
    case DSL_BUF_PRESSURES:
      if (int rc = materialise_press(h)) return rc;
      hipLaunchKernelGGL(k_pack1, g, b, 0, h->stream, h->n, h->stage, ids, h->press, sorted_order);
      break;
    case DSL_BUF_PCI_POSITIONS:
    case DSL_BUF_PCI_VELOCITIES: {
      if (!h->pci_active) return fail(h, DSL_ERR_INVALID, "dsl_download: PCISPH state not initialised (dsl_pcisph_begin)");
      Soa3 s = buffer == DSL_BUF_PCI_POSITIONS ? mpcip(h) : mpciv(h);
      hipLaunchKernelGGL(k_pack3, g, b, 0, h->stream, h->n, h->stage, ids, s.x, s.y, s.z, sorted_order);
      break;
    }
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host, h->stage, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // the positions slice itself keeps what was uploaded for particle N(); only Get() reads it as the origin
  if (with_boundary && buffer == DSL_BUF_POSITIONS && !sorted_order) std::memcpy(host + (size_t)3 * n_fluid_of(h), h->b0_pos, sizeof(h->b0_pos));
  return DSL_OK;
}

int dsl_download(dsl_handle* h, int buffer, float* host, size_t count) {
  return download_impl(h, buffer, host, count, 0);
}
int dsl_download_sorted(dsl_handle* h, int buffer, float* host, size_t count) {
  return download_impl(h, buffer, host, count, 1);
}
int dsl_download_ids(dsl_handle* h, int32_t* ids, size_t count) {
  CHECK_HANDLE(h);
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  if (!ids || count != (size_t)ncur) return fail(h, DSL_ERR_INVALID, "dsl_download_ids: bad argument");
  HIP_TRY(h, hipMemcpyAsync(ids, h->ids[h->cur_ids], count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}
int dsl_download_cell_start(dsl_handle* h, int32_t* cs, size_t count) {
  CHECK_HANDLE(h);
  if (!cs || count != (size_t)h->ncell + 1) return fail(h, DSL_ERR_INVALID, "dsl_download_cell_start: count must be cells+1");
  if (!h->grid_valid || h->lsh) return fail(h, DSL_ERR_INVALID, "dsl_download_cell_start: no valid grid table");
  HIP_TRY(h, hipMemcpyAsync(cs, h->cell_start, count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

int dsl_download_decimated(dsl_handle* h, int buffer, int stride, float* host, size_t count) {
  CHECK_HANDLE(h);
  int n = 0;
  if (int rc = host_count(h, &n)) return rc;
  if (h->ids_global) return fail(h, DSL_ERR_INVALID, "dsl_download_decimated: host order is gone after dsl_set_ids");
  if (!host || stride < 1 || (buffer != DSL_BUF_POSITIONS && buffer != DSL_BUF_VELOCITIES))
    return fail(h, DSL_ERR_INVALID, "dsl_download_decimated: positions or velocities, stride >= 1");
  const size_t want = 3 * (((size_t)n + stride - 1) / stride);
  if (count != want) return fail(h, DSL_ERR_INVALID, "dsl_download_decimated: count must be 3*ceil(N/stride)");
  CSoa3 a = buffer == DSL_BUF_POSITIONS ? cpos(h) : cvel(h);
  hipLaunchKernelGGL(k_pack3_decimated, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, n, stride, h->stage,
                     h->ids[h->cur_ids], a.x, a.y, a.z);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host, h->stage, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

int dsl_device_pointers(dsl_handle* h, int buffer, const float** xyz, const int32_t** ids, int* n) {
  CHECK_HANDLE(h);
  if (!xyz || (buffer != DSL_BUF_POSITIONS && buffer != DSL_BUF_VELOCITIES))
    return fail(h, DSL_ERR_INVALID, "dsl_device_pointers: positions or velocities");
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;  // also drains the stream
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  CSoa3 a = buffer == DSL_BUF_POSITIONS ? cpos(h) : cvel(h);
  xyz[0] = a.x;
  xyz[1] = a.y;
  xyz[2] = a.z;
  if (ids) *ids = h->ids[h->cur_ids];
  if (n) *n = ncur;
  return DSL_OK;
}

int dsl_set_hash_vectors(dsl_handle* h, const float* vectors, int hash_bits) {
  CHECK_HANDLE(h);
  if (!h->lsh) return fail(h, DSL_ERR_INVALID, "dsl_set_hash_vectors: the handle is not in DSL_NEIGH_LSH_REF mode");
  if (!vectors || hash_bits < 1 || hash_bits > 32) return fail(h, DSL_ERR_INVALID, "dsl_set_hash_vectors: 1..32 hash bits");
  HIP_TRY(h, hipMemcpyAsync(h->hashv, vectors, (size_t)hash_bits * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->lsh_bits = hash_bits;
  h->grid_valid = false;
  return DSL_OK;
}

int dsl_lsh_download_table(dsl_handle* h, int32_t* out, size_t count) {
  CHECK_HANDLE(h);
  if (!h->lsh) return fail(h, DSL_ERR_INVALID, "dsl_lsh_download_table: the handle is not in DSL_NEIGH_LSH_REF mode");
  const int B = h->prm.lsh_buckets, S = h->prm.lsh_bucket_size;
  if (!out || count != (size_t)B * S) return fail(h, DSL_ERR_INVALID, "dsl_lsh_download_table: count must be buckets*bucket_size");
  if (int rc = ensure_grid(h)) return rc;
  std::vector<int> tab((size_t)B * h->lsh_cap), len(B);
  HIP_TRY(h, hipMemcpyAsync(tab.data(), h->lsh_table, tab.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(len.data(), h->lsh_len, len.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int b = 0; b < B; ++b)  // GetData1D, lsh.go:70-80: zero-filled beyond the bucket's length
    for (int j = 0; j < S; ++j) out[(size_t)b * S + j] = (j < len[b] && j < h->lsh_cap) ? tab[(size_t)b * h->lsh_cap + j] : 0;
  return DSL_OK;
}

int dsl_build_neighbours(dsl_handle* h) {
  CHECK_HANDLE(h);
  return build_grid(h, true);
}

int dsl_density_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (int rc = ensure_grid(h)) return rc;
  return density_pass(h);
}

int dsl_pressure_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  const DevConsts& c = h->c;
  int rc = timed(h, DSL_K_PRESSURE, [&] {
    by_math(h, [&](auto fast) {
      hipLaunchKernelGGL((k_pressure<decltype(fast)::value>), dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, c,
                         bnd_of(h), h->rho, h->press);
    });
  });
  if (rc) return rc;
  h->press_zero = false;
  return DSL_OK;
}

int dsl_viscous_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (int rc = ensure_grid(h)) return rc;
  if (int rc = materialise_forces(h)) return rc;
  return viscous_pass(h);
}

int dsl_external_pass(dsl_handle* h, const float f[3]) {
  CHECK_HANDLE(h);
  if (!f) return fail(h, DSL_ERR_INVALID, "null force");
  if (int rc = materialise_forces(h)) return rc;
  Soa3 F = mfrc(h);
  return timed(h, DSL_K_EXTERNAL, [&] {
    hipLaunchKernelGGL(k_external, dim3(grid_for(launch_n(h))), dim3(kBlock), 0, h->stream, launch_n(h), F, f[0], f[1], f[2]);
  });
}

int dsl_gradient_pressure_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (int rc = ensure_grid(h)) return rc;
  if (int rc = materialise_forces(h)) return rc;
  return gradient_pass(h, 0);
}

int dsl_update_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  return update_pass(h);
}

int dsl_force_pass(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (int rc = ensure_grid(h)) return rc;
  return force_integrate(h);
}

// *ncur = slots to launch over (Total()); the operators are evaluated for the N() fluid particles
static int field_prepare(dsl_handle* h, int* ncur) {
  if (h->c.n_ptr) return fail(h, DSL_ERR_UNSUPPORTED, "field operators are not available in slab mode");
  if (int rc = ensure_grid(h)) return rc;
  return host_count(h, ncur);
}

static int field_div_curl(dsl_handle* h, int tensor_buffer, float* host_out, size_t count, bool curl) {
  CHECK_HANDLE(h);
  int n = 0;
  if (int rc = field_prepare(h, &n)) return rc;
  if (!host_out || count != (size_t)n_fluid_of(h) * (curl ? 3 : 1)) return fail(h, DSL_ERR_INVALID, "field operator: bad output size");
  CSoa3 t;
  if (tensor_buffer == DSL_BUF_VELOCITIES) t = cvel(h);
  else if (tensor_buffer == DSL_BUF_FORCES) {
    if (int rc = materialise_forces(h)) return rc;
    t = cfrc(h);
  } else return fail(h, DSL_ERR_INVALID, "field operator: tensor field must be velocities or forces");
  CSoa3 p = cpos(h);
  // results in slot order: x -> scratch1, y/z -> the idle half of the force ping-pong pair
  Soa3 out{h->scratch1, h->frc[h->cur_f ^ 1][1], h->frc[h->cur_f ^ 1][2]};
  by_math(h, [&](auto fast) {
    constexpr bool F = decltype(fast)::value;
    if (curl)
      hipLaunchKernelGGL((k_field_div_curl<F, kOpCurl>), dim3(grid_for(n)), dim3(kBlock), 0, h->stream, h->c,
                         neigh(h), bnd_of(h), p, t, h->rho, out);
    else
      hipLaunchKernelGGL((k_field_div_curl<F, kOpDiv>), dim3(grid_for(n)), dim3(kBlock), 0, h->stream, h->c,
                         neigh(h), bnd_of(h), p, t, h->rho, out);
  });
  HIP_TRY(h, hipGetLastError());
  const int* ids = h->ids[h->cur_ids];
  if (curl) hipLaunchKernelGGL(k_pack3, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, n, h->stage, ids, out.x, out.y, out.z, 0);
  else hipLaunchKernelGGL(k_pack1, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, n, h->stage, ids, out.x, 0);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host_out, h->stage, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

int dsl_field_divergence(dsl_handle* h, int tensor_buffer, float* host_out, size_t count) {
  return field_div_curl(h, tensor_buffer, host_out, count, false);
}
int dsl_field_curl(dsl_handle* h, int tensor_buffer, float* host_out, size_t count) {
  return field_div_curl(h, tensor_buffer, host_out, count, true);
}

static int scalar_field_id(int buffer) {
  return buffer == DSL_BUF_DENSITIES ? 0 : (buffer == DSL_BUF_PRESSURES ? 1 : -1);
}

int dsl_field_laplacian(dsl_handle* h, int scalar_buffer, float* host_out, size_t count) {
  CHECK_HANDLE(h);
  int n = 0;
  if (int rc = field_prepare(h, &n)) return rc;
  const int field = scalar_field_id(scalar_buffer);
  if (!host_out || count != (size_t)n_fluid_of(h) || field < 0) return fail(h, DSL_ERR_INVALID, "dsl_field_laplacian: bad argument");
  CSoa3 p = cpos(h);
  by_math(h, [&](auto fast) {
    hipLaunchKernelGGL((k_field_laplacian<decltype(fast)::value>), dim3(grid_for(n)), dim3(kBlock), 0, h->stream, h->c,
                       neigh(h), bnd_of(h), p, h->rho, field, h->scratch1);
  });
  hipLaunchKernelGGL(k_pack1, dim3(grid_for(n)), dim3(kBlock), 0, h->stream, n, h->stage, h->ids[h->cur_ids], h->scratch1, 0);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host_out, h->stage, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

int dsl_field_interpolate(dsl_handle* h, int scalar_buffer, const float* host_positions, size_t npos, float* host_out) {
  CHECK_HANDLE(h);
  int n = 0;
  if (int rc = field_prepare(h, &n)) return rc;
  const int field = scalar_field_id(scalar_buffer);
  if (!host_positions || !host_out || field < 0) return fail(h, DSL_ERR_INVALID, "dsl_field_interpolate: bad argument");
  if (npos > (size_t)h->cap) return fail(h, DSL_ERR_INVALID, "dsl_field_interpolate: at most `capacity` query positions per call");
  if (npos == 0) return DSL_OK;
  CSoa3 p = cpos(h);
  HIP_TRY(h, hipMemcpyAsync(h->stage, host_positions, npos * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  by_math(h, [&](auto fast) {
    hipLaunchKernelGGL((k_field_interpolate<decltype(fast)::value>), dim3(grid_for((int)npos)), dim3(kBlock), 0, h->stream,
                       h->c, neigh(h), p, h->rho, field, (int)npos, h->stage, h->scratch1);
  });
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host_out, h->scratch1, npos * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

namespace {
// ---------------------------------------------------------------------------------------
// the skin step (kernels_skin.hpp): neighbour lists that live for several steps
// ---------------------------------------------------------------------------------------
// Every kSkinLook skin steps the host reads one word of the device state: has the flow outrun the skin (five of the last
// 16 steps rebuilt)?  Then the lists cost more than they save and the step goes back to sorting and sweeping every step
// for kSkinRetry steps.  Both happen at step counts fixed in advance: results do not depend on timing or on how the
// steps were grouped into calls.
constexpr int kSkinLook = 32;
constexpr int kSkinRetry = 2048;

bool skin_usable(const dsl_handle* h) {
  return h->skin > 0.0f && h->steps >= h->skin_retry_at && h->prm.math_mode == DSL_MATH_FAST && !h->lsh && h->c.slab_axis < 0 && !h->c.n_ptr && h->nb == 0 &&
         !h->pci_active && h->c.xsph_eps == 0.0f && h->c.st_kappa == 0.0f && use_tiled(h) && h->nmask != nullptr &&
         (h->c.wcsph_pressure_force != 0 || h->c.wcsph_viscosity != 0);
}

int skin_alloc(dsl_handle* h) {
  int rc = DSL_OK;
  if (!h->skin_state && (rc = dev_alloc(h, &h->skin_state, 1))) return rc;
  if (!h->lists && (rc = dev_alloc(h, &h->lists, (size_t)kLMaxChunks * h->cap))) return rc;
  for (int k = 0; k < 6; ++k)
    if (!h->pvz[k] && (rc = dev_alloc(h, &h->pvz[k], (size_t)h->cap))) return rc;
  for (int k = 0; k < 3; ++k)
    if (!h->pvr[k] && (rc = dev_alloc(h, &h->pvr[k], (size_t)h->cap))) return rc;
  return DSL_OK;
}

// from the plain state into skin steps: wide cells, the device state seeded with the host's view, a rebuild forced
int skin_enter(dsl_handle* h) {
  if (int rc = skin_alloc(h)) return rc;
  if (h->sort_scratch_dirty) {
    HIP_TRY(h, hipMemsetAsync(h->cell_count, 0, sizeof(int) * (size_t)h->ncell_pad, h->stream));
    if (h->unordered) HIP_TRY(h, hipMemsetAsync(h->unordered, 0, sizeof(unsigned int) * (size_t)(h->ncell_pad / 32), h->stream));
    h->sort_scratch_dirty = false;
  }
  if (int rc = set_cell_edge(h, h->c.h * (1.0f + h->skin), 3)) return rc;
  if (h->tg.ntiles > h->alloc_ntiles) {  // (cannot happen for skin <= 0.2: wider cells, 4/3 the layers)
    (void)set_cell_edge(h, h->c.h);
    return fail(h, DSL_ERR_INVALID, "skin step: the tile grid outgrew its allocation");
  }
  SkinState init{};
  init.ids_sel = h->cur_ids;
  init.force = 1;
  init.budget = 0.5f * h->skin * h->c.h * (1.0f - 1.0e-3f);
  init.dt = h->c.dt;
  init.n_live = h->n;
  init.predict = h->skin_predict;
  init.vmax2_bits = 0x7f800000u;  // (unknown until a skin step has integrated: the first build is where the particles are)
  HIP_TRY(h, hipMemcpyAsync(h->skin_state, &init, sizeof(init), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // (`init` lives on this stack frame; entering is rare)
  h->skin_live = true;
  h->skin_live_steps = 0;
  return DSL_OK;
}

int skin_step(dsl_handle* h) {
  const DevConsts& c = h->c;
  SkinState* st = h->skin_state;
  const SkinGate gate{st};
  const int n = h->n;
  const int X = h->cur_pv, Y = X ^ 1;
  const CSoa3 pX = cpos(h), vX = cvel(h);
  const CSoa3 pZ{h->pvz[0], h->pvz[1], h->pvz[2]}, vZ{h->pvz[3], h->pvz[4], h->pvz[5]};
  // the build's REFERENCE positions x + tau v in sorted order (SkinState::tau): what the sort's cells, the candidate
  // sweep and the lists are built at, and what displacement is measured against
  const CSoa3 pR{h->pvr[0], h->pvr[1], h->pvr[2]};
  const Soa3 pRw{h->pvr[0], h->pvr[1], h->pvr[2]};
  const bool ordered = !h->prm.sort_unordered;
  hipLaunchKernelGGL(k_skin_decide, dim3(1), dim3(1), 0, h->stream, st);
  HIP_TRY(h, hipGetLastError());
  // the rebuild chain: launched every step, every kernel returns at once unless this step rebuilds
  int rc = timed(h, DSL_K_CELL_RANK, [&] {
    hipLaunchKernelGGL((k_cell_rank<false, true>), dim3(std::min(grid_for((n + kRankUnroll - 1) / kRankUnroll), 4096)), dim3(kBlock), 0, h->stream, c, pX.x, pX.y, pX.z,
                       ordered ? h->ids[0] : nullptr, h->rank, h->cell_count, h->unordered, nullptr, nullptr,
                       ordered ? h->cell_keys : nullptr, h->dcounter + 3, nullptr, 0, gate, h->ids[1], vX);
  });
  if (rc) return rc;
  rc = timed(h, DSL_K_SCAN, [&] {
    hipLaunchKernelGGL(k_scan_sums, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->cell_count, h->block_sums, h->dstats,
                       h->n_tiles, gate);
    hipLaunchKernelGGL(k_scan_apply, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->cell_count, h->block_sums,
                       h->cell_start, h->dstats, gate);
  });
  if (rc) return rc;
  ScatterArrays a{};
  for (int k = 0; k < 6; ++k) {
    a.src[k] = h->pv[X][k];
    a.dst[k] = h->pvz[k];
  }
  a.nf = 6;
  a.ids_src = h->ids[0];  // (the kernels swap the two by the device's ids_sel)
  a.ids_dst = h->ids[1];
  ScatterOrder so{ordered ? h->unordered : nullptr, h->sort_keys, reinterpret_cast<unsigned char*>(h->sort_work), nullptr,
                  ordered ? h->cell_keys : nullptr, h->dcounter + 3};
  rc = timed(h, DSL_K_SCATTER, [&] {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter<true, 6>), dim3(std::min(grid_for((n + kScatterUnroll - 1) / kScatterUnroll), 4096)), dim3(kBlock), 0, h->stream, c, a, so, pX, h->rank, h->cell_start, gate,
                       vX, pRw);
    if (ordered)
      hipLaunchKernelGGL(k_scatter_ordered<true>, dim3(h->cell_keys ? std::min(grid_for(n), 1024) : grid_for(n)), dim3(kBlock), 0,
                         h->stream, c, a, so, pX, h->rank, h->cell_start, gate, vX, pRw);
  });
  if (rc) return rc;
  rc = timed(h, DSL_K_TILE_LIST, [&] {
    hipLaunchKernelGGL(k_tile_list, dim3(grid_for(h->tg.nlist)), dim3(kBlock), 0, h->stream, c, h->tg, h->cell_start, h->tiles,
                       h->n_tiles, h->n_tiles + 5, nullptr, h->tile_desc_of, h->unordered, h->ncell_pad / 32, h->dcounter + 3,
                       gate);
    hipLaunchKernelGGL(k_tile_desc, dim3(std::min(h->tg.ntiles, 8192)), dim3(kWave), 0, h->stream, c, h->tg, h->cell_start,
                       h->cell_start, h->tiles, h->n_tiles, h->tile_desc, nullptr, kTCap, gate);
  });
  if (rc) return rc;
  // candidates within h (1 + skin) of the sorted positions -> masks -> lists
  const double reach = 1.0 + (double)h->skin;
  const float wide_thr = (float)(1.0 - reach * reach * 1.0004 - 1.0e-4);
  rc = timed(h, DSL_K_NEIGH_LISTS, [&] {
    hipLaunchKernelGGL((k_density_pair<true, true>), dim3(persistent_grid(h, 8, true)), dim3(kPBlock), 0, h->stream, c, h->tg,
                       h->tile_desc_of, h->n_tiles, h->tile_desc, h->cell_start, bnd_of(h), pR, h->rho, h->pterm, h->nmask,
                       h->cap, wide_thr, gate);
    if (h->list_build_lockstep)
      hipLaunchKernelGGL(k_list_build, dim3(persistent_grid(h, 2, true)), dim3(kLBlock), 0, h->stream, c, h->tg, h->tile_desc_of,
                         h->n_tiles, h->tile_desc, h->nmask, h->cap, h->lists, h->cap, st, gate, pR, wide_thr);
    else
      hipLaunchKernelGGL(k_list_build_v1, dim3(persistent_grid(h, 2, true)), dim3(kLBlock), 0, h->stream, c, h->tg, h->tile_desc_of,
                         h->n_tiles, h->tile_desc, h->nmask, h->cap, h->lists, h->cap, st, gate, pR, wide_thr);
  });
  if (rc) return rc;
  // the step itself: densities and the fused force + integrate over the lists
  rc = timed(h, DSL_K_DENSITY, [&] {
    hipLaunchKernelGGL(k_density_list, dim3(persistent_grid(h, 3, true)), dim3(kLBlock), 0, h->stream, c, h->tg, h->tile_desc_of,
                       h->n_tiles, h->tile_desc, h->cell_start, st, pX, pZ, h->lists, h->cap, h->rho, h->pterm);
  });
  if (rc) return rc;
  const Soa3 po = mpos(h, Y), vo = mvel(h, Y);
  const bool G = c.wcsph_pressure_force != 0, V = c.wcsph_viscosity != 0;
  rc = timed(h, DSL_K_FORCE_INTEGRATE, [&] {
    dim3 g(persistent_grid(h, 2, true)), b(kLBlock);
    int* wql = walk_ctr_of(h, kSiteForceList);
#define DSL_LAUNCH_FL2(GG, VV, QQ)                                                                                           \
  hipLaunchKernelGGL((k_force_list<GG, VV, QQ>), g, b, 0, h->stream, c, h->tg, h->tile_desc_of, h->n_tiles, h->tile_desc,   \
                     h->cell_start, st, pX, vX, pZ, vZ, pR, h->rho, h->pterm, h->lists, h->cap, po, vo, h->dstats, wql)
#define DSL_LAUNCH_FL(GG, VV)                                             \
  do {                                                                    \
    if (wql != nullptr) DSL_LAUNCH_FL2(GG, VV, true);                     \
    else DSL_LAUNCH_FL2(GG, VV, false);                                   \
  } while (0)
    if (G && V) DSL_LAUNCH_FL(true, true);
    else if (G) DSL_LAUNCH_FL(true, false);
    else DSL_LAUNCH_FL(false, true);
#undef DSL_LAUNCH_FL
#undef DSL_LAUNCH_FL2
  });
  if (rc) return rc;
  h->cur_pv = Y;
  h->masks_valid = false;
  h->grid_valid = false;
  h->dens_fresh = h->dens_held = true;
  h->forces_uniform = true;
  h->press_zero = true;
  h->skin_live_steps += 1;
  if (h->skin_live_steps % kSkinLook == 0 || h->skin_live_steps == 2) {  // (2: the first rebuild's tile statistics are in)
    int give_up = 0;
    HIP_TRY(h, hipMemcpyAsync(&give_up, &st->give_up, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (give_up) {
      // (every suspension in a row doubles the wait, up to 16 x: a flow that has developed rarely calms down again)
      h->skin_retry_at = h->steps + 1 + ((int64_t)kSkinRetry << std::min(h->skin_suspensions, 4));
      h->skin_suspensions += 1;
      return skin_settle(h);
    }
  }
  return DSL_OK;
}

// back to the plain state: which slot -> particle map is current is the device's knowledge (one small read)
int skin_settle(dsl_handle* h) {
  SkinState s{};
  HIP_TRY(h, hipMemcpyAsync(&s, h->skin_state, sizeof(s), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->cur_ids = s.ids_sel & 1;
  h->skin_steps_total += s.n_steps;
  h->skin_rebuilds_total += s.n_rebuilds;
  h->skin_list_overflow |= s.list_overflow != 0;
  h->skin_live = false;
  h->dens_held = false;  // (rho / pterm are in slot order, but nothing downstream may rely on lists any more)
  h->dens_fresh = false;
  return set_cell_edge(h, h->c.h);
}

}  // namespace

int dsl_wcsph_step(dsl_handle* h, int nsteps) {
  CHECK_HANDLE_ONLY(h);
  for (int s = 0; s < nsteps; ++s) {
    // (forces that differ from force_reset -- uploaded ones, before the first Update -- take one plain step first)
    if (skin_usable(h) && (h->skin_live || h->forces_uniform)) {
      if (!h->skin_live)
        if (int rc = skin_enter(h)) return rc;
      if (int rc = skin_step(h)) return rc;
      h->steps++;
      continue;
    }
    if (h->skin_live)
      if (int rc = skin_settle(h)) return rc;
    if (int rc = build_grid(h, false)) return rc;  // NN(): geometric neighbour rule -> every step
    if (int rc = density_pass(h)) return rc;        // DensityAll   wcsph.go:18
    if (int rc = force_integrate(h)) return rc;     // ExternalAll, PressureAll, Update wcsph.go:19-21
    h->steps++;
  }
  return DSL_OK;
}

int dsl_set_option(dsl_handle* h, int option, double value) {
  CHECK_HANDLE(h);
  switch (option) {
    case DSL_OPT_SKIN:
      // (0.2: the wide sweep's LDS image holds (6 cells x 2 (1 + s) particles)^3 <= kTCapWide records)
      if (!(value >= 0.0 && value <= 0.2)) return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_SKIN is a fraction of h in [0, 0.2]");
      h->skin = (float)value;
      h->skin_retry_at = 0;
      return DSL_OK;
    case DSL_OPT_SKIN_PREDICT:
      if (!(value >= 0.0 && value <= 0.95)) return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_SKIN_PREDICT is a fraction of the displacement budget in [0, 0.95]");
      h->skin_predict = (float)value;
      return DSL_OK;
    // the fall-back forms of the kernels (A/B runs, tests: each is product code some configuration or failure path
    // reaches, tests/test_gpu_variants.py holds every one of them to the default's parity bar)
    case DSL_OPT_DENSITY_PAIR: h->density_pair = value != 0.0; return DSL_OK;
    case DSL_OPT_CELL_KEYS:
      h->cell_keys = value != 0.0 ? h->cell_keys_alloc : nullptr;
      h->grid_valid = false;
      return DSL_OK;
    case DSL_OPT_PERSISTENT_BLOCKS:
      if (!(value >= 0.0 && value <= 65536.0)) return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_PERSISTENT_BLOCKS is a workgroup count (0 = no cap)");
      h->max_persistent_blocks = (int)value;
      return DSL_OK;
    case DSL_OPT_TILE_BOX: {  // bx | by << 8 | bz << 16; 0 = the tile grid's linear order
      const int v = (int)value, bx = v & 255, by = (v >> 8) & 255, bz = (v >> 16) & 255;
      if (value < 0.0 || value >= 16777216.0 || (v != 0 && (bx == 0 || by == 0 || bz == 0)))
        return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_TILE_BOX is bx | by << 8 | bz << 16 (all three positive), or 0");
      h->tile_box[0] = bx;
      h->tile_box[1] = by;
      h->tile_box[2] = bz;
      h->grid_valid = false;
      h->masks_valid = false;
      return set_tile_grid(h);
    }
    case DSL_OPT_PCI_QTILED: h->pci_qtiled = value != 0.0; return DSL_OK;
    case DSL_OPT_PCI_QPAIR: h->pci_qpair = value != 0.0; return DSL_OK;
    case DSL_OPT_PCI_QROWS: h->pci_qrows = value != 0.0; return DSL_OK;
    case DSL_OPT_LIST_BUILD: h->list_build_lockstep = value != 0.0; return DSL_OK;
    case DSL_OPT_TILE_QUEUE:
      if (!(value == 0.0 || value == 1.0 || value == 2.0)) return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_TILE_QUEUE is 0, 1 or 2");
      h->tile_queue = (int)value;
      return DSL_OK;
    case DSL_OPT_GRID_OVERSUB:
      if (!(value >= 1.0 && value <= 64.0)) return fail(h, DSL_ERR_INVALID, "dsl_set_option: DSL_OPT_GRID_OVERSUB is a factor in [1, 64]");
      h->grid_oversub = (int)value;
      return DSL_OK;
    case DSL_OPT_PCI_QINCR:
      h->pci_qincr = value != 0.0;
      h->pci_rows_live = false;
      return DSL_OK;
    default:
      return fail(h, DSL_ERR_INVALID, "dsl_set_option: unknown or read-only option");
  }
}

int dsl_get_option(dsl_handle* h, int option, double* value) {
  CHECK_HANDLE_ONLY(h);
  if (!value) return fail(h, DSL_ERR_INVALID, "dsl_get_option: null output");
  SkinState s{};
  if (h->skin_live && (option == DSL_OPT_SKIN_STEPS || option == DSL_OPT_SKIN_REBUILDS || option == DSL_OPT_SKIN_LIST_OVERFLOW ||
                       option == DSL_OPT_SKIN_FIELDS_OWN || option == DSL_OPT_SKIN_FIELDS_PADDED || option == DSL_OPT_SKIN_TAU_STEPS)) {
    HIP_TRY(h, hipMemcpyAsync(&s, h->skin_state, sizeof(s), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  switch (option) {
    case DSL_OPT_SKIN: *value = h->skin; return DSL_OK;
    case DSL_OPT_SKIN_STEPS: *value = (double)(h->skin_steps_total + s.n_steps); return DSL_OK;
    case DSL_OPT_SKIN_REBUILDS: *value = (double)(h->skin_rebuilds_total + s.n_rebuilds); return DSL_OK;
    case DSL_OPT_SKIN_LIST_OVERFLOW: *value = (h->skin_list_overflow || s.list_overflow != 0) ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_SKIN_SUSPENSIONS: *value = (double)h->skin_suspensions; return DSL_OK;
    case DSL_OPT_DEVICE_BYTES: *value = (double)h->dev_bytes; return DSL_OK;
    case DSL_OPT_SKIN_FIELDS_OWN: *value = (double)s.fields_own; return DSL_OK;
    case DSL_OPT_SKIN_FIELDS_PADDED: *value = (double)s.fields_padded; return DSL_OK;
    case DSL_OPT_SKIN_PREDICT: *value = h->skin_predict; return DSL_OK;
    case DSL_OPT_SKIN_TAU_STEPS: *value = s.dt > 0.0f ? (double)s.tau / (double)s.dt : 0.0; return DSL_OK;
    case DSL_OPT_DENSITY_PAIR: *value = h->density_pair ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_CELL_KEYS: *value = h->cell_keys != nullptr ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_PERSISTENT_BLOCKS: *value = (double)h->max_persistent_blocks; return DSL_OK;
    case DSL_OPT_TILE_BOX: *value = (double)(h->tile_box[0] | (h->tile_box[1] << 8) | (h->tile_box[2] << 16)); return DSL_OK;
    case DSL_OPT_PCI_QTILED: *value = h->pci_qtiled ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_PCI_QPAIR: *value = h->pci_qpair ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_PCI_QROWS: *value = h->pci_qrows ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_LIST_BUILD: *value = h->list_build_lockstep ? 1.0 : 0.0; return DSL_OK;
    case DSL_OPT_GRID_OVERSUB: *value = (double)h->grid_oversub; return DSL_OK;
    case DSL_OPT_TILE_QUEUE: *value = (double)h->tile_queue; return DSL_OK;
    case DSL_OPT_PCI_QINCR: *value = h->pci_qincr ? 1.0 : 0.0; return DSL_OK;
    default: return fail(h, DSL_ERR_INVALID, "dsl_get_option: unknown option");
  }
}

int dsl_pcisph_begin(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (int rc = alloc_pci(h)) return rc;
  const size_t n = (size_t)h->cap;  // (slab mode keeps the live count on the device: copy every slot)
  for (int k = 0; k < 6; ++k)
    HIP_TRY(h, hipMemcpyAsync(h->pci[h->cur_pci][k], h->pv[h->cur_pv][k], n * sizeof(float), hipMemcpyDeviceToDevice,
                              h->stream));
  h->pci_active = true;
  // the predicted positions are the particles' again: the un-binned sweeps are the fast ones until they drift apart
  if (h->pci_steps != 0) HIP_TRY(h, hipMemsetAsync(h->pci_drift, 0, 512 * sizeof(unsigned int), h->stream));
  h->pci_steps = 0;
  h->pci_binned = false;
  if (h->drift_pending) HIP_TRY(h, hipEventSynchronize(h->ev_drift));  // (the snapshot copy may still be in flight)
  h->drift_pending = false;
  HIP_TRY(h, hipMemsetAsync(&h->dstats->pci_escaped, 0, sizeof(int), h->stream));
  return DSL_OK;
}

namespace {
// Measured on the 4M scene (tools/pci_switch_point.py, profiles/r03_pci_switch_point.jsonl): one correction iteration
// of the un-binned form costs 0.24 ms while every query sits in its particle's tile, 0.35 with 0.2 % of them outside
// (a wave with one such lane sweeps global memory for it), 0.83 with 1.9 %; the binned form 0.37 .. 0.48 whatever the
// drift.  Hence the low threshold and the short period.
constexpr int kPciDriftPeriod = 4;           // steps between two looks at the drift counters
constexpr double kPciDriftFraction = 0.002;  // of the predicted positions outside their particle's tile: bin the queries

// Every kPciDriftPeriod steps: how many DensityF query points have left their particle's tile (counted by the un-binned
// kernels)?  No stall: the counters are copied to pinned memory asynchronously and cleared, and it is the NEXT look, a
// period of queued steps later, that reads the copy (waiting for its event only throttles a host that has run more than
// a period ahead of the device).  Both happen at step counts fixed in advance, so a run's arithmetic does not depend on
// how its steps were grouped into calls or on timing.
int pci_drift_check(dsl_handle* h) {
  if (h->pci_bin_mode != 0 || h->pci_binned || h->lsh || h->pci_steps == 0 || h->pci_steps % kPciDriftPeriod != 0) return DSL_OK;
  // (a host that is capturing this stream into a graph of its own gets no look: waiting for an event is not capturable,
  // and a replayed graph could not change its kernels anyway -- such a host sets the mode with dsl_pcisph_set_binning)
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) return DSL_OK;
  if (h->drift_pending) {
    HIP_TRY(h, hipEventSynchronize(h->ev_drift));
    h->drift_pending = false;
    unsigned long long left = 0, all = 0;
    for (int k = 0; k < 256; ++k) {
      left += h->pci_drift_host[k];
      all += h->pci_drift_host[256 + k];
    }
    if (all != 0 && (double)left > kPciDriftFraction * (double)all) {
      h->pci_binned = true;
      return DSL_OK;
    }
  }
  HIP_TRY(h, hipMemcpyAsync(h->pci_drift_host, h->pci_drift, 512 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipEventRecord(h->ev_drift, h->stream));
  HIP_TRY(h, hipMemsetAsync(h->pci_drift, 0, 512 * sizeof(unsigned int), h->stream));
  h->drift_pending = true;
  return DSL_OK;
}

// FAST mode: LDS-tiled sweeps; the gradient term (a function of x and rho only,
// field_types.go:39-42) is identical in every correction iteration, so it is swept once
// and re-added.  The running-mass viscosity recurrence for m != 1 needs the branching kernel.
bool pci_tiled(const dsl_handle* h) {
  return h->prm.math_mode == DSL_MATH_FAST && !(h->c.visc_running_mass && h->c.mass != 1.0f);
}
bool pci_extra_terms(const dsl_handle* h) { return h->c.xsph_eps != 0.0f || h->c.st_kappa != 0.0f; }

// NN, DensityAll, ViscousAll (pcisph_darwin.go:43-45) and the loop set-up
int pci_begin_step(dsl_handle* h) {
  const DevConsts& c = h->c;
  h->pci_rows_live = false;  // (the sort re-numbers the particles: the query rows start afresh)
  if (int rc = pci_drift_check(h)) return rc;
  if (int rc = build_grid(h, false)) return rc;
  if (int rc = density_pass(h)) return rc;        // DensityAll  pcisph_darwin.go:44
  // Forces still at their reset value (Update leaves them so, nothing materialised): the tiled set-up sweep below writes
  // reset + its terms itself instead of adding to a filled array -- when every live slot is one of its targets (no
  // boundary particles, no ghosts), so that no slot is left unwritten
  const bool fill_in_sweep = h->forces_uniform && pci_tiled(h) && h->nb == 0 && !h->c.n_ptr && !h->lsh;
  if (!fill_in_sweep)
    if (int rc = materialise_forces(h)) return rc;
  if (int rc = materialise_press(h)) return rc;
  CSoa3 p = cpos(h), v = cvel(h);
  Soa3 F = mfrc(h);
  CSoa3 cF{F.x, F.y, F.z};
  const bool XS = pci_extra_terms(h);
  if (pci_tiled(h)) {
    Soa3 G{h->gterm[0], h->gterm[1], h->gterm[2]};
    Soa3 none{nullptr, nullptr, nullptr};
    Soa3 xs{h->xsph[0], h->xsph[1], h->xsph[2]};
    // ViscousAll :45 (+ cohesion, XSPH sums) and GradientPressureForce's term, which is the same in every
    // correction iteration: one sweep over the masks for both
    int rc = timed(h, DSL_K_VISCOUS, [&] {
      if (XS)
        hipLaunchKernelGGL((k_force_integrate_tiled<true, true, kOutPci, true>), dim3(persistent_grid(h, 2)),
                           dim3(kTBlock), 0, h->stream, c, h->tg, h->tile_desc_of, h->n_tiles, nullptr, nullptr, h->tile_desc, h->cell_start, p, v, h->rho,
                           h->pterm, cF, fill_in_sweep ? 1 : 0, F, xs, h->dstats, h->masks_valid ? h->nmask : nullptr, h->cap, nullptr, bnd_of(h), G);
      else
        hipLaunchKernelGGL((k_force_integrate_tiled<true, true, kOutPci>), dim3(persistent_grid(h, 2)),
                           dim3(kTBlock), 0, h->stream, c, h->tg, h->tile_desc_of, h->n_tiles, nullptr, nullptr, h->tile_desc, h->cell_start, p, v, h->rho,
                           h->pterm, cF, fill_in_sweep ? 1 : 0, F, none, h->dstats, h->masks_valid ? h->nmask : nullptr, h->cap, nullptr, bnd_of(h), G);
    });
    if (rc) return rc;
    if (fill_in_sweep) h->forces_uniform = false;
  } else {
    if (int rc = viscous_pass(h, XS ? 1 : 0)) return rc;
  }
  hipLaunchKernelGGL(k_pci_reset, dim3(1), dim3(1), 0, h->stream, h->dstats, h->n_qtiles);
  h->pci_iter_pending = false;
  HIP_TRY(h, hipGetLastError());
  return DSL_OK;
}

// one correction iteration up to (not including) the convergence check :52-94
int pci_iterate(dsl_handle* h) {
  const DevConsts& c = h->c;
  dim3 g(grid_for(launch_n(h))), b(kBlock);
  CSoa3 p = cpos(h);
  Soa3 F = mfrc(h);
  CSoa3 cF{F.x, F.y, F.z};
  Soa3 pp = mpcip(h), pvv = mpciv(h);
  const bool tiled = pci_tiled(h);
  CSoa3 cpp{pp.x, pp.y, pp.z};
  if (!h->lsh && (h->pci_bin_mode > 0 || (h->pci_bin_mode == 0 && h->pci_binned))) {
    // the query points have left their particles (or the host asked for it): counting sort of the queries by their own
    // cells, then the sweep in that order (kernels_sph.hpp: k_pci_predict_bin)
    if (int rc = alloc_query_bins(h)) return rc;
    CSoa3 cG{h->gterm[0], h->gterm[1], h->gterm[2]};
    const bool qtiled = tiled && h->pci_qtiled;
    const bool rows = qtiled && h->pci_qpair && h->pci_qrows && h->qrows != nullptr;
    // rows kept inside a step: the first iteration fills them (and leaves the counts standing), the later ones move only
    // the queries that have changed cells
    const bool keep = rows && h->pci_qincr && h->qslot != nullptr;
    const bool incr = keep && h->pci_rows_live;
    if (h->qcount_dirty && !incr) HIP_TRY(h, hipMemsetAsync(h->qcount, 0, sizeof(int) * (size_t)h->ncell_pad, h->stream));
    h->qcount_dirty = true;
    h->pci_rows_live = false;
    // n_qtiles[0] tile list length, [1] spilled queries: left at zero by k_pci_reset / k_pci_check (a memset node costs
    // 7 us on the device, four per step); only a host that iterates twice without the check in between gets one here
    if (rows && (h->pci_iter_pending || !h->pci_counters_clean)) HIP_TRY(h, hipMemsetAsync(h->n_qtiles, 0, 2 * sizeof(int), h->stream));
    h->pci_counters_clean = true;
    h->pci_iter_pending = true;
    int rc = timed(h, DSL_K_PCI_PREDICT, [&] {
      if (rows) {
        if (incr)
          hipLaunchKernelGGL((k_pci_predict_bin<false, true, true, true, true>), g, b, 0, h->stream, c, bnd_of(h), p, pp, pvv, cG, F,
                             h->qcount, nullptr, nullptr, nullptr, h->dcounter + 4, h->build_seq, h->press, h->dstats, h->qrows,
                             kQueryRow, h->qrec, h->n_qtiles + 1, h->qslot);
        else
          hipLaunchKernelGGL((k_pci_predict_bin<false, true, true, true>), g, b, 0, h->stream, c, bnd_of(h), p, pp, pvv, cG, F,
                             h->qcount, nullptr, nullptr, nullptr, h->dcounter + 4, h->build_seq, h->press, h->dstats, h->qrows,
                             kQueryRow, h->qrec, h->n_qtiles + 1, keep ? h->qslot : nullptr);
        hipLaunchKernelGGL(k_qtile_list<true>, dim3(grid_for(h->tg.nlist)), dim3(kBlock), 0, h->stream, c, h->tg, h->qcount,
                           h->qtiles, h->n_qtiles, h->dstats);
        hipLaunchKernelGGL(k_tile_desc, dim3(std::min(h->tg.ntiles, 8192)), dim3(kWave), 0, h->stream, c, h->tg,
                           h->cell_start, nullptr, h->qtiles, h->n_qtiles, h->qtile_desc, h->qcount, kTCap, SkinGate{nullptr}, keep);
        return;
      }
      if (tiled)
        hipLaunchKernelGGL((k_pci_predict_bin<false, true, true>), g, b, 0, h->stream, c, bnd_of(h), p, pp, pvv, cG, F,
                           h->qcount, h->rank, h->n_qtiles, nullptr, h->dcounter + 4, h->build_seq, h->press, h->dstats);
      else
        by_math(h, [&](auto fast) {
          hipLaunchKernelGGL((k_pci_predict_bin<true, false, decltype(fast)::value>), g, b, 0, h->stream, c, bnd_of(h), p, pp,
                             pvv, cG, F, h->qcount, h->rank, nullptr, nullptr, h->dcounter + 4, h->build_seq, h->press, h->dstats);
        });
      hipLaunchKernelGGL(k_scan_sums, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->qcount, h->qsums, nullptr, nullptr);
      hipLaunchKernelGGL(k_scan_apply, dim3(h->nscan), dim3(kBlock), 0, h->stream, h->qcount, h->qsums, h->qstart, nullptr);
      hipLaunchKernelGGL(k_pci_query_scatter, g, b, 0, h->stream, c, cpp, h->rank, h->qstart, h->qrec, h->dstats);
      if (qtiled) {  // the tiles that hold queries, and their tables
        hipLaunchKernelGGL(k_qtile_list<false>, dim3(grid_for(h->tg.nlist)), dim3(kBlock), 0, h->stream, c, h->tg, h->qstart,
                           h->qtiles, h->n_qtiles, h->dstats);
        hipLaunchKernelGGL(k_tile_desc, dim3(std::min(h->tg.ntiles, 8192)), dim3(kWave), 0, h->stream, c, h->tg,
                           h->cell_start, h->qstart, h->qtiles, h->n_qtiles, h->qtile_desc);
      }
    });
    if (rc) return rc;
    HIP_TRY(h, hipGetLastError());
    h->qcount_dirty = keep;  // (otherwise the kernels that leave the histogram zeroed are queued)
    h->pci_rows_live = keep;
    rc = timed(h, DSL_K_PCI_DENSITY, [&] {
      if (rows) {
        hipLaunchKernelGGL(k_pci_density_qpair<true>, dim3(persistent_grid(h, 8)), dim3(kPBlock), 0, h->stream, c, h->tg,
                           h->n_qtiles, h->qtile_desc, h->cell_start, p, h->qrows, h->press, h->dstats);
        // the queries that found their cell's row full (a global-memory sweep over a short list; usually empty)
        hipLaunchKernelGGL((k_pci_density_binned<true>), dim3(std::min(grid_for(launch_n(h)), 1024)), b, 0, h->stream, c,
                           neigh(h), p, h->qrec, h->n_qtiles + 1, h->press, h->dstats);
        return;
      }
      if (qtiled) {
        if (h->pci_qpair)  // two queries of one cell per lane (256-thread workgroups, four per CU)
          hipLaunchKernelGGL(k_pci_density_qpair<false>, dim3(persistent_grid(h, 8)), dim3(kPBlock), 0, h->stream, c, h->tg,
                             h->n_qtiles, h->qtile_desc, h->cell_start, p, h->qrec, h->press, h->dstats);
        else
          hipLaunchKernelGGL(k_pci_density_qtiled, dim3(persistent_grid(h, 2)), dim3(kTBlock), 0, h->stream, c, h->tg,
                             h->n_qtiles, h->qtile_desc, h->cell_start, p, h->qrec, h->press, h->dstats);
        return;
      }
      by_math(h, [&](auto fast) {
        hipLaunchKernelGGL((k_pci_density_binned<decltype(fast)::value>), g, b, 0, h->stream, c, neigh(h), p, h->qrec,
                           h->qstart + h->c.ncell, h->press, h->dstats);
      });
    });
    if (rc) return rc;
    return tiled ? DSL_OK : gradient_pass(h, 1);  // (tiled: F += the cached term went with the predictor)
  }
  if (tiled) {  // predict, DensityF + pressure, F += cached gradient term: one launch (kernels_tiled.hpp)
    CSoa3 cG{h->gterm[0], h->gterm[1], h->gterm[2]};
    return timed(h, DSL_K_PCI_DENSITY, [&] {
      hipLaunchKernelGGL(k_pci_density_tiled, dim3(persistent_grid(h, 4)), dim3(kTBlock), 0, h->stream, c, h->tg,
                         h->tile_desc_of, h->n_tiles, h->tile_desc, h->cell_start, bnd_of(h), p, pp, pvv, cG, F, h->press,
                         h->pci_drift, h->dstats);
    });
  }
  int rc = timed(h, DSL_K_PCI_PREDICT, [&] {
    hipLaunchKernelGGL(k_pci_predict, g, b, 0, h->stream, c, bnd_of(h), p, cF, pp, pvv, h->lsh ? nullptr : h->pci_drift,
                       h->dstats);
  });
  if (rc) return rc;
  rc = timed(h, DSL_K_PCI_DENSITY, [&] {
    by_math(h, [&](auto fast) {
      hipLaunchKernelGGL((k_pci_density<decltype(fast)::value>), g, b, 0, h->stream, c, neigh(h), bnd_of(h), p, cpp,
                         h->press, h->dstats);
    });
  });
  if (rc) return rc;
  return gradient_pass(h, 1);  // GradientPressureForce :93
}

int pci_check(dsl_handle* h) {                   // :95-98
  hipLaunchKernelGGL(k_pci_check, dim3(1), dim3(1), 0, h->stream, h->c, h->dstats, h->n_qtiles);
  h->pci_iter_pending = false;
  HIP_TRY(h, hipGetLastError());
  return DSL_OK;
}

int pci_end_step(dsl_handle* h) {                // Update :101 (positions advect with v + XSPH)
  if (int rc = update_pass(h, pci_extra_terms(h))) return rc;
  h->steps++;
  h->pci_steps++;
  return DSL_OK;
}
}  // namespace

int dsl_pcisph_step(dsl_handle* h, int nsteps) {
  CHECK_HANDLE(h);
  if (h->c.n_ptr)
    return fail(h, DSL_ERR_UNSUPPORTED,
                "dsl_pcisph_step: in slab mode drive the step with dsl_pcisph_phase (the iteration error is global)");
  if (!h->pci_active) {
    if (int rc = dsl_pcisph_begin(h)) return rc;
  }
  for (int s = 0; s < nsteps; ++s) {
    if (int rc = pci_begin_step(h)) return rc;
    for (int it = 0; it < h->prm.pci_max_iters; ++it) {
      if (int rc = pci_iterate(h)) return rc;
      if (int rc = pci_check(h)) return rc;
    }
    if (int rc = pci_end_step(h)) return rc;
  }
  return DSL_OK;
}

int dsl_pcisph_phase(dsl_handle* h, int phase) {
  CHECK_HANDLE(h);
  if (!h->pci_active) return fail(h, DSL_ERR_INVALID, "dsl_pcisph_phase: call dsl_pcisph_begin first");
  switch (phase) {
    case DSL_PCI_BEGIN_STEP: h->pci_split_guard = true; return pci_begin_step(h);
    case DSL_PCI_ITERATE: return pci_iterate(h);
    case DSL_PCI_CHECK: return pci_check(h);
    case DSL_PCI_END_STEP: h->pci_split_guard = false; return pci_end_step(h);
    default: return fail(h, DSL_ERR_INVALID, "dsl_pcisph_phase: bad phase");
  }
}

int dsl_pcisph_error_word(dsl_handle* h, uint32_t* dev_word, int store) {
  CHECK_HANDLE(h);
  if (!dev_word) return fail(h, DSL_ERR_INVALID, "dsl_pcisph_error_word: null pointer");
  uint32_t* mine = &h->dstats->pci_cur_err_bits;
  HIP_TRY(h, hipMemcpyAsync(store ? mine : dev_word, store ? dev_word : mine, sizeof(uint32_t), hipMemcpyDeviceToDevice,
                            h->stream));
  return DSL_OK;
}

int dsl_pcisph_set_binning(dsl_handle* h, int mode) {
  CHECK_HANDLE(h);
  if (mode < -1 || mode > 1) return fail(h, DSL_ERR_INVALID, "dsl_pcisph_set_binning: mode is -1 (never), 0 (automatic) or 1 (always)");
  if (mode > 0 && h->lsh) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_pcisph_set_binning: lsh_ref candidates do not come from grid cells");
  h->pci_bin_mode = mode;
  return DSL_OK;
}

int dsl_pcisph_get_binning(dsl_handle* h, int* mode, int* active, int* escaped) {
  CHECK_HANDLE(h);
  if (mode) *mode = h->pci_bin_mode;
  if (active) *active = (!h->lsh && (h->pci_bin_mode > 0 || (h->pci_bin_mode == 0 && h->pci_binned))) ? 1 : 0;
  if (escaped) {  // (the one blocking part: the flag lives on the device)
    int e = 0;
    HIP_TRY(h, hipMemcpyAsync(&e, &h->dstats->pci_escaped, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *escaped = e ? 1 : 0;
  }
  return DSL_OK;
}

int dsl_slab_config(dsl_handle* h, int axis, float lo, float hi) {
  CHECK_HANDLE(h);
  if (axis < -1 || axis > 2 || !(lo < hi)) return fail(h, DSL_ERR_INVALID, "dsl_slab_config: bad axis or empty range");
  if (h->lsh) return fail(h, DSL_ERR_UNSUPPORTED, "lsh_ref buckets are angular cones through the whole domain: no slabs");
  if (axis >= 0 && h->nb > 0) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_slab_config: boundary particles are not supported in slab mode");
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  h->c.slab_axis = axis;
  h->c.slab_lo = axis < 0 ? -INFINITY : lo;
  h->c.slab_hi = axis < 0 ? INFINITY : hi;
  if (axis >= 0) {
    const int init[16] = {ncur};
    HIP_TRY(h, hipMemcpyAsync(h->dn, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->c.n_ptr = h->dn;
  } else {
    h->c.n_ptr = nullptr;
    h->c.n = ncur;
  }
  update_split(h);
  h->grid_valid = false;
  return DSL_OK;
}

size_t dsl_slab_message_floats(int cap_full, int cap_xonly) {
  if (cap_full < 0 || cap_xonly < 0) return 0;
  return (size_t)(cap_full + 1) * kRecord + (size_t)cap_xonly * kRecordX;
}

int dsl_slab_record_floats(dsl_handle* h) { return h && h->pci_active ? kRecordPci : kRecord; }

size_t dsl_slab_message_floats_for(dsl_handle* h, int cap_full, int cap_xonly) {
  if (cap_full < 0 || cap_xonly < 0) return 0;
  return (size_t)(cap_full + 1) * dsl_slab_record_floats(h) + (size_t)cap_xonly * kRecordX;
}

namespace {
// band cell layers: every cell that reaches within width + margin of a slab plane
void update_split(dsl_handle* h) {
  DevConsts& c = h->c;
  c.split_cl = 0;
  c.split_ch = INT_MAX;
  c.own_c0 = 0;
  c.own_c1 = INT_MAX;
  if (c.slab_axis < 0) return;
  const int a = c.slab_axis;
  // Cell layers that can hold owned particles.  A plane that coincides with a cell plane counts as
  // exact (1e-3 cell of slack): an owned particle that float rounding leaves in a ghost-only tile is
  // still integrated correctly, through the global-memory sweep (k_force_integrate_tiled).
  if (std::isfinite(c.slab_lo)) {
    const float f = std::floor((c.slab_lo - c.gmin[a]) * c.inv_cell + 1.0e-3f);
    c.own_c0 = f <= 0.0f ? 0 : (f >= (float)c.dims[a] ? c.dims[a] : (int)f);
  }
  if (std::isfinite(c.slab_hi)) {
    const float f = std::ceil((c.slab_hi - c.gmin[a]) * c.inv_cell - 1.0e-3f);
    c.own_c1 = f <= 0.0f ? 0 : (f >= (float)c.dims[a] ? c.dims[a] : (int)f);
  }
  if (!(h->split_width > 0.0f)) return;
  const float reach = h->split_width + h->split_margin;
  if (std::isfinite(c.slab_lo)) {
    // (1e-3 cell of slack: a reach that ends on a cell plane must not spill into the next layer)
    const float f = std::ceil((c.slab_lo + reach - c.gmin[a]) * c.inv_cell - 1.0e-3f);
    c.split_cl = f <= 0.0f ? 0 : (f >= (float)c.dims[a] ? c.dims[a] : (int)f);
  }
  if (std::isfinite(c.slab_hi)) {
    const float f = std::floor((c.slab_hi - reach - c.gmin[a]) * c.inv_cell + 1.0e-3f);
    c.split_ch = f <= 0.0f ? 0 : (f >= (float)c.dims[a] ? c.dims[a] : (int)f);
  } else {
    c.split_ch = c.dims[a];
  }
  if (c.split_ch < c.split_cl) c.split_ch = c.split_cl;  // thin slab: every layer is a band layer
}
}  // namespace

int dsl_slab_split(dsl_handle* h, float width, float margin) {
  CHECK_HANDLE(h);
  if (!(width >= 0.0f) || !(margin >= 0.0f)) return fail(h, DSL_ERR_INVALID, "dsl_slab_split: bad width or margin");
  if (h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_slab_split: a split force pass is in flight");
  h->split_width = width;
  h->split_margin = margin;
  update_split(h);
  h->grid_valid = false;  // the band / interior tile lists are made by the neighbour build
  return DSL_OK;
}

namespace {
int slab_pack_on(dsl_handle* h, hipStream_t st, float width_full, float width, bool from_output, float* dev_lo,
                 float* dev_hi, int cap_full, int cap_x) {
  SlabBands sb;
  sb.full_lo = h->c.slab_lo + width_full;
  sb.band_lo = h->c.slab_lo + width;
  sb.full_hi = h->c.slab_hi - width_full;
  sb.band_hi = h->c.slab_hi - width;
  CSoa3 p = cpos(h), v = cvel(h);
  const float* old_axis = nullptr;
  if (from_output) {  // band phase of the split step: integrated values are in the other half
    old_axis = h->c.slab_axis == 0 ? p.x : (h->c.slab_axis == 1 ? p.y : p.z);
    Soa3 po = mpos(h, h->cur_pv ^ 1), vo = mvel(h, h->cur_pv ^ 1);
    p = CSoa3{po.x, po.y, po.z};
    v = CSoa3{vo.x, vo.y, vo.z};
  }
  const float* pa = h->c.slab_axis == 0 ? p.x : (h->c.slab_axis == 1 ? p.y : p.z);
  const int nblk = (h->cap + kPackChunk - 1) / kPackChunk;
  CSoa3 pcip{nullptr, nullptr, nullptr}, pciv{nullptr, nullptr, nullptr};
  if (h->pci_active) {  // migrants take their predictor state along
    Soa3 a = mpcip(h), b = mpciv(h);
    pcip = CSoa3{a.x, a.y, a.z};
    pciv = CSoa3{b.x, b.y, b.z};
  }
  hipLaunchKernelGGL(k_slab_count, dim3(nblk), dim3(kBlock), 0, st, h->c, sb, old_axis, pa, dev_lo ? 1 : 0,
                     dev_hi ? 1 : 0, h->pack_counts);
  hipLaunchKernelGGL(k_slab_offsets, dim3(1), dim3(kOffsBlock), 0, st, h->pack_counts, nblk, dev_lo, dev_hi, cap_full,
                     cap_x, h->dn + 5);
  hipLaunchKernelGGL(k_slab_write, dim3(nblk), dim3(kBlock), 0, st, h->c, sb, old_axis, pa, p.x, p.y, p.z, v.x, v.y, v.z,
                     h->ids[h->cur_ids], dev_lo, dev_hi, cap_full, cap_x, h->pack_counts, dsl_slab_record_floats(h), pcip,
                     pciv);
  HIP_TRY(h, hipGetLastError());
  return DSL_OK;
}
int slab_pack_check(dsl_handle* h, const char* who, float width_full, float width, float* dev_lo, float* dev_hi,
                    int cap_full, int cap_x) {
  if (!h->c.n_ptr) return fail(h, DSL_ERR_INVALID, std::string(who) + ": no slab configured");
  if ((!dev_lo && !dev_hi) || cap_full < 0 || cap_x < 0 || !(width_full >= 0.0f) || !(width >= width_full))
    return fail(h, DSL_ERR_INVALID, std::string(who) + ": bad argument");
  return DSL_OK;
}
}  // namespace

int dsl_slab_pack(dsl_handle* h, float width_full, float width, float* dev_lo, float* dev_hi, int cap_full,
                  int cap_xonly) {
  CHECK_HANDLE(h);
  if (int rc = slab_pack_check(h, "dsl_slab_pack", width_full, width, dev_lo, dev_hi, cap_full, cap_xonly)) return rc;
  if (h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_slab_pack: a split force pass is in flight");
  return slab_pack_on(h, h->stream, width_full, width, false, dev_lo, dev_hi, cap_full, cap_xonly);
}

int dsl_slab_pack_band(dsl_handle* h, float width_full, float* dev_lo, float* dev_hi, int cap_full, int cap_xonly,
                       void* stream) {
  CHECK_HANDLE(h);
  if (int rc = slab_pack_check(h, "dsl_slab_pack_band", width_full, h->split_width, dev_lo, dev_hi, cap_full, cap_xonly))
    return rc;
  if (!h->split_pending)
    return fail(h, DSL_ERR_INVALID, "dsl_slab_pack_band: call dsl_force_pass_split(DSL_SPLIT_BAND) first");
  hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->stream;
  if (st != h->stream) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_band, 0));
  return slab_pack_on(h, st, width_full, h->split_width, true, dev_lo, dev_hi, cap_full, cap_xonly);
}

int dsl_force_pass_split(dsl_handle* h, int phase) {
  CHECK_HANDLE(h);
  if (phase == DSL_SPLIT_BAND) {
    if (h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_force_pass_split: band phase already done");
    if (!h->c.n_ptr || !(h->split_width > 0.0f))
      return fail(h, DSL_ERR_INVALID, "dsl_force_pass_split: needs dsl_slab_config and dsl_slab_split");
    if (!use_tiled(h)) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_force_pass_split: needs the tiled kernels (DSL_MATH_FAST)");
    if (int rc = ensure_grid(h)) return rc;
    if (!h->ev_band) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_band, hipEventDisableTiming));
    if (int rc = force_integrate(h, 1)) return rc;
    HIP_TRY(h, hipEventRecord(h->ev_band, h->stream));
    h->split_pending = true;
    return DSL_OK;
  }
  if (phase == DSL_SPLIT_INNER) {
    if (!h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_force_pass_split: band phase missing");
    h->split_pending = false;
    return force_integrate(h, 2);
  }
  return fail(h, DSL_ERR_INVALID, "dsl_force_pass_split: bad phase");
}

static int slab_append_shifted(dsl_handle* h, const float* dev_message_a, const float* dev_message_b, int cap_full,
                               int cap_xonly, float shift_a, float shift_b) {
  CHECK_HANDLE(h);
  if (!h->c.n_ptr) return fail(h, DSL_ERR_INVALID, "dsl_slab_append: no slab configured");
  if ((!dev_message_a && !dev_message_b) || cap_full < 0 || cap_xonly < 0)
    return fail(h, DSL_ERR_INVALID, "dsl_slab_append: bad argument");
  if (h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_slab_append: a split force pass is in flight");
  if (h->pci_active && h->pci_split_guard) return fail(h, DSL_ERR_INVALID, "dsl_slab_append: PCISPH step in flight");
  if (!h->forces_uniform) return fail(h, DSL_ERR_INVALID, "dsl_slab_append: forces must be uniform (dsl_reset_forces)");
  if (cap_full + cap_xonly == 0) return DSL_OK;
  Soa3 p = mpos(h, h->cur_pv), v = mvel(h, h->cur_pv);
  Soa3 pcip{nullptr, nullptr, nullptr}, pciv{nullptr, nullptr, nullptr};
  if (h->pci_active) {
    pcip = mpcip(h);
    pciv = mpciv(h);
  }
  hipLaunchKernelGGL(k_slab_append, dim3(grid_for(cap_full + cap_xonly), 2), dim3(kBlock), 0, h->stream, dev_message_a,
                     dev_message_b, cap_full, cap_xonly, h->dn, h->cap, p.x, p.y, p.z, v.x, v.y, v.z,
                     h->ids[h->cur_ids], h->dn + 5, dsl_slab_record_floats(h), pcip, pciv, h->c.slab_axis, shift_a, shift_b);
  hipLaunchKernelGGL(k_slab_bump, dim3(1), dim3(1), 0, h->stream, dev_message_a, dev_message_b, cap_full, cap_xonly,
                     h->dn, h->cap);
  HIP_TRY(h, hipGetLastError());
  h->grid_valid = false;
  h->dens_fresh = false;
  h->dens_held = false;  // (the appended records have no densities yet; the slab step's sort does not carry them)
  return DSL_OK;
}

int dsl_slab_append2(dsl_handle* h, const float* dev_message_a, const float* dev_message_b, int cap_full,
                     int cap_xonly) {
  return slab_append_shifted(h, dev_message_a, dev_message_b, cap_full, cap_xonly, 0.0f, 0.0f);
}

int dsl_slab_append(dsl_handle* h, const float* dev_message, int cap_full, int cap_xonly) {
  return dsl_slab_append2(h, dev_message, nullptr, cap_full, cap_xonly);
}

int dsl_get_count(dsl_handle* h, int* n_live, int* n_owned) {
  CHECK_HANDLE(h);
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  if (n_live) *n_live = ncur;
  if (n_owned) {
    *n_owned = ncur;
    if (h->c.slab_axis >= 0 && ncur > 0) {
      HIP_TRY(h, hipMemsetAsync(h->dcounter, 0, sizeof(int), h->stream));
      CSoa3 p = cpos(h);
      hipLaunchKernelGGL(k_count_owned, dim3(1024), dim3(kBlock), 0, h->stream, h->c, p.x, p.y, p.z, h->dcounter);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipMemcpyAsync(n_owned, h->dcounter, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
  }
  return DSL_OK;
}

int dsl_slab_status(dsl_handle* h, int32_t status[4], int reset_high_water) {
  CHECK_HANDLE(h);
  if (!status) return fail(h, DSL_ERR_INVALID, "dsl_slab_status: null output");
  int st[3] = {0, 0, 0};
  int missed = 0;
  if (h->c.n_ptr) {
    HIP_TRY(h, hipMemcpyAsync(st, h->dn + 5, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&missed, &h->dstats->band_missed, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (reset_high_water) HIP_TRY(h, hipMemsetAsync(h->dn + 6, 0, 2 * sizeof(int), h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  status[0] = st[0];
  status[1] = missed;
  status[2] = st[1];
  status[3] = st[2];
  return DSL_OK;
}

int dsl_slab_overflow(dsl_handle* h, int* high_water) {
  int32_t st[4] = {0, 0, 0, 0};
  if (int rc = dsl_slab_status(h, st, 0)) return rc;
  if (high_water) *high_water = st[0];
  return DSL_OK;
}

// ---------------------------------------------------------------------------------------
// multi-GPU behind the C ABI: communicator, slab link, step drivers (slab_link.hpp)
// ---------------------------------------------------------------------------------------
const char* dsl_comm_last_error(void) { return g_comm_error.c_str(); }

int dsl_comm_unique_id(uint8_t id[DSL_COMM_ID_BYTES]) {
  if (!id) return DSL_ERR_INVALID;
  if (!rccl_load()) {
    g_comm_error = rccl().err;
    return DSL_ERR_UNSUPPORTED;
  }
  static_assert(DSL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  ncclUniqueId u;
  const ncclResult_t r = rccl().GetUniqueId(&u);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r);
    return DSL_ERR_DEVICE;
  }
  std::memcpy(id, u.internal, DSL_COMM_ID_BYTES);
  return DSL_OK;
}

int dsl_comm_create(int nranks, int rank, const uint8_t id[DSL_COMM_ID_BYTES], int device, dsl_comm** out) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) {
    g_comm_error = "dsl_comm_create: bad argument";
    return DSL_ERR_INVALID;
  }
  *out = nullptr;
  if (!rccl_load()) {
    g_comm_error = rccl().err;
    return DSL_ERR_UNSUPPORTED;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_comm_error = "dsl_comm_create: hipSetDevice failed";
    return DSL_ERR_DEVICE;
  }
  ncclUniqueId u;
  std::memcpy(u.internal, id, DSL_COMM_ID_BYTES);
  dsl_comm* c = new (std::nothrow) dsl_comm();
  if (!c) return DSL_ERR_NOMEM;
  const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
    delete c;
    return DSL_ERR_DEVICE;
  }
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  *out = c;
  return DSL_OK;
}

int dsl_comm_create_all(int ndev, const int* devices, dsl_comm** out) {
  if (!out || !devices || ndev < 1) {
    g_comm_error = "dsl_comm_create_all: bad argument";
    return DSL_ERR_INVALID;
  }
  if (!rccl_load()) {
    g_comm_error = rccl().err;
    return DSL_ERR_UNSUPPORTED;
  }
  std::vector<ncclComm_t> comms((size_t)ndev);
  const ncclResult_t r = rccl().CommInitAll(comms.data(), ndev, devices);
  if (r != ncclSuccess) {
    g_comm_error = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r);
    return DSL_ERR_DEVICE;
  }
  for (int k = 0; k < ndev; ++k) {
    dsl_comm* c = new dsl_comm();
    c->comm = comms[(size_t)k];
    c->nranks = ndev;
    c->rank = k;
    c->device = devices[k];
    out[k] = c;
  }
  return DSL_OK;
}

int dsl_comm_create_custom(int nranks, int rank, int device, const dsl_transport* t, dsl_comm** out) {
  if (!out || !t || nranks < 1 || rank < 0 || rank >= nranks || !t->group_start || !t->group_end || !t->send || !t->recv ||
      !t->all_reduce_max_u32) {
    g_comm_error = "dsl_comm_create_custom: bad argument (every callback of dsl_transport is required)";
    return DSL_ERR_INVALID;
  }
  dsl_comm* c = new (std::nothrow) dsl_comm();
  if (!c) return DSL_ERR_NOMEM;
  c->custom = true;
  c->tr = *t;
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  *out = c;
  return DSL_OK;
}

int dsl_comm_destroy(dsl_comm* c) {
  if (!c) return DSL_OK;
  if (c->comm && !c->custom && rccl().lib) (void)rccl().CommDestroy(c->comm);
  delete c;
  return DSL_OK;
}

int dsl_comm_count(dsl_comm* c, int* nranks) {
  if (!c || !nranks) {
    g_comm_error = "dsl_comm_count: bad argument";
    return DSL_ERR_INVALID;
  }
  *nranks = c->nranks;
  if (!c->custom && c->comm) {
    int n = 0;
    const ncclResult_t r = rccl().CommCount(c->comm, &n);
    if (r != ncclSuccess) {
      g_comm_error = std::string("ncclCommCount: ") + rccl().GetErrorString(r);
      return DSL_ERR_DEVICE;
    }
    *nranks = n;
  }
  return DSL_OK;
}

int dsl_create_multi(const dsl_params* params, int ndev, const int* devices, dsl_handle** handles, dsl_comm** comms) {
  if (!params || !devices || !handles || !comms || ndev < 1) return fail(nullptr, DSL_ERR_INVALID, "dsl_create_multi: bad argument");
  for (int k = 0; k < ndev; ++k) handles[k] = nullptr, comms[k] = nullptr;
  for (int k = 0; k < ndev; ++k) {
    if (int rc = dsl_create(&params[k], devices[k], &handles[k])) {
      for (int j = 0; j < k; ++j) (void)dsl_destroy(handles[j]), handles[j] = nullptr;
      return rc;
    }
  }
  if (int rc = dsl_comm_create_all(ndev, devices, comms)) {
    for (int j = 0; j < ndev; ++j) (void)dsl_destroy(handles[j]), handles[j] = nullptr;
    return fail(nullptr, rc, g_comm_error);
  }
  return DSL_OK;
}

int dsl_slab_detach(dsl_handle* h) {
  if (!h || !h->link) return DSL_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  SlabLink* L = h->link;
  if (L->comm_stream) (void)hipStreamSynchronize(L->comm_stream);
  for (int k = 0; k < 2; ++k) {
    (void)hipFree(L->send[k]);
    (void)hipFree(L->recv[k]);
  }
  (void)hipFree(L->dev_words);
  if (L->ev_pack) (void)hipEventDestroy(L->ev_pack);
  if (L->ev_xfer) (void)hipEventDestroy(L->ev_xfer);
  if (L->comm_stream) (void)hipStreamDestroy(L->comm_stream);
  delete L;
  h->link = nullptr;
  return DSL_OK;
}

int dsl_slab_attach(dsl_handle* h, dsl_comm* comm, int lo_rank, int hi_rank, float width_full, float width, int cap_full,
                    int cap_xonly, int overlap) {
  CHECK_HANDLE(h);
  if (!h->c.n_ptr) return fail(h, DSL_ERR_INVALID, "dsl_slab_attach: call dsl_slab_config first");
  if ((lo_rank >= 0 || hi_rank >= 0) && !comm) return fail(h, DSL_ERR_INVALID, "dsl_slab_attach: neighbours need a communicator");
  if (comm && (lo_rank >= comm->nranks || hi_rank >= comm->nranks))
    return fail(h, DSL_ERR_INVALID, "dsl_slab_attach: neighbour rank outside the communicator");
  if (cap_full < 0 || cap_xonly < 0 || !(width_full >= 0.0f) || !(width >= width_full))
    return fail(h, DSL_ERR_INVALID, "dsl_slab_attach: bad widths or capacities");
  if (overlap && !use_tiled(h)) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_slab_attach: the split step needs DSL_MATH_FAST");
  (void)dsl_slab_detach(h);
  SlabLink* L = new (std::nothrow) SlabLink();
  if (!L) return fail(h, DSL_ERR_NOMEM, "dsl_slab_attach: out of host memory");
  h->link = L;
  L->comm = comm;
  L->lo = lo_rank;
  L->hi = hi_rank;
  L->width_full = width_full;
  L->width = width;
  L->cap_full = cap_full;
  L->cap_x = cap_xonly;
  L->max_full = 2 * cap_full;  // room for the re-plan to grow the messages (the same on every rank)
  L->max_x = 2 * cap_xonly;
  L->overlap = overlap != 0 && (lo_rank >= 0 || hi_rank >= 0);
  L->buf_floats = (size_t)(L->max_full + 1) * kRecordPci + (size_t)L->max_x * kRecordX;  // (13-float records once PCISPH runs)
  for (int k = 0; k < 2; ++k) {
    if (int rc = dev_alloc(h, &L->send[k], L->buf_floats)) return rc;
    if (int rc = dev_alloc(h, &L->recv[k], L->buf_floats)) return rc;
    HIP_TRY(h, hipMemsetAsync(L->send[k], 0, L->buf_floats * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(L->recv[k], 0, L->buf_floats * sizeof(float), h->stream));
  }
  if (int rc = dev_alloc(h, &L->dev_words, 4)) return rc;
  {  // the transfer stream gets the highest priority: a hardware queue of its own (two plain streams may share
     // one, and then the transfer kernels wait behind the interior force launch they are meant to run under)
    int least = 0, greatest = 0;
    HIP_TRY(h, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(h, hipStreamCreateWithPriority(&L->comm_stream, hipStreamNonBlocking, greatest));
  }
  HIP_TRY(h, hipEventCreateWithFlags(&L->ev_pack, hipEventDisableTiming));
  HIP_TRY(h, hipEventCreateWithFlags(&L->ev_xfer, hipEventDisableTiming));
  if (L->overlap) {
    if (int rc = dsl_slab_split(h, width, h->split_margin > 0.0f ? h->split_margin : width)) return rc;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

namespace {
// one fixed-size message per neighbour and direction, all four transfers in one RCCL group
int link_post(dsl_handle* h, hipStream_t st) {
  SlabLink& L = *h->link;
  if (L.lo < 0 && L.hi < 0) return DSL_OK;
  const size_t n = dsl_slab_message_floats_for(h, L.cap_full, L.cap_x);
  dsl_comm* comm = L.comm;
  if (int rc = xfer_group_start(h, comm)) return rc;
  // (an error between group start and group end must not leave the group open: every later RCCL call of the
  // process would be queued into it and nothing would ever be sent -- close it, keep the first error)
  auto body = [&]() -> int {
    if (L.lo >= 0)
      if (int rc = xfer_send(h, comm, L.send[0], n, L.lo, st)) return rc;
    if (L.hi >= 0)
      if (int rc = xfer_send(h, comm, L.send[1], n, L.hi, st)) return rc;
    // two messages between the same pair of ranks (two ranks with periodic images, or a rank that is
    // its own neighbour in a test) are matched in issue order: the peer's LOW band arrives from above
    const bool same_peer = L.lo >= 0 && L.lo == L.hi;
    if (same_peer) {
      if (int rc = xfer_recv(h, comm, L.recv[1], n, L.hi, st)) return rc;
      if (int rc = xfer_recv(h, comm, L.recv[0], n, L.lo, st)) return rc;
    } else {
      if (L.lo >= 0)
        if (int rc = xfer_recv(h, comm, L.recv[0], n, L.lo, st)) return rc;
      if (L.hi >= 0)
        if (int rc = xfer_recv(h, comm, L.recv[1], n, L.hi, st)) return rc;
    }
    return DSL_OK;
  };
  const int rc = body();
  if (rc != DSL_OK) {
    const std::string first = h->err;
    (void)xfer_group_end(h, comm);
    h->err = first;
    return rc;
  }
  return xfer_group_end(h, comm);
}

int link_append(dsl_handle* h) {
  SlabLink& L = *h->link;
  if (L.lo < 0 && L.hi < 0) return DSL_OK;
  if (int rc = slab_append_shifted(h, L.lo >= 0 ? L.recv[0] : nullptr, L.hi >= 0 ? L.recv[1] : nullptr, L.cap_full, L.cap_x,
                                   L.shift_from_lo, L.shift_from_hi))
    return rc;
  L.ghosts_in = true;
  return DSL_OK;
}

// unsplit exchange on the handle's stream: pack, transfer, append
int link_exchange(dsl_handle* h) {
  SlabLink& L = *h->link;
  if (L.lo < 0 && L.hi < 0) return DSL_OK;
  if (int rc = slab_pack_on(h, h->stream, L.width_full, L.width, false, L.lo >= 0 ? L.send[0] : nullptr,
                            L.hi >= 0 ? L.send[1] : nullptr, L.cap_full, L.cap_x))
    return rc;
  if (int rc = link_post(h, h->stream)) return rc;
  return link_append(h);
}

// every rank learns the largest band counts (and any overflow) seen anywhere since the last
// re-plan; all ranks switch to the same new message capacities.  Blocking.
int link_replan(dsl_handle* h) {
  SlabLink& L = *h->link;
  int32_t st[4] = {0, 0, 0, 0};
  if (int rc = dsl_slab_status(h, st, 1)) return rc;
  if (L.comm && L.comm->nranks > 1) {
    HIP_TRY(h, hipMemcpyAsync(L.dev_words, st, sizeof(st), hipMemcpyHostToDevice, h->stream));
    if (int rc = xfer_all_reduce_max(h, L.comm, L.dev_words, 4, h->stream)) return rc;  // (counts: never negative)
    HIP_TRY(h, hipMemcpyAsync(st, L.dev_words, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (st[0] != 0)
    return fail(h, DSL_ERR_OVERFLOW,
                "slab exchange overflow: " + std::to_string(st[0]) +
                    " records did not fit a band message or the particle capacity on some rank; particles were lost -- "
                    "enlarge cap_full / cap_xonly / dsl_params.capacity");
  if (st[1] != 0)
    return fail(h, DSL_ERR_OVERFLOW, "split slab step: a particle outran the margin on some rank (ghosts were missed); enlarge the margin");
  const long long want_full = (long long)(1.15 * st[2]) + 1024, want_x = (long long)(1.15 * st[3]) + 1024;
  L.cap_full = (int)(want_full < L.max_full ? want_full : L.max_full);
  L.cap_x = (int)(want_x < L.max_x ? want_x : L.max_x);
  return DSL_OK;
}
}  // namespace

int dsl_slab_image_shift(dsl_handle* h, float from_lo, float from_hi) {
  CHECK_HANDLE(h);
  if (!h->link) return fail(h, DSL_ERR_INVALID, "dsl_slab_image_shift: call dsl_slab_attach first");
  h->link->shift_from_lo = from_lo;
  h->link->shift_from_hi = from_hi;
  return DSL_OK;
}

int dsl_slab_exchange(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (!h->link) return fail(h, DSL_ERR_INVALID, "dsl_slab_exchange: call dsl_slab_attach first");
  if (h->split_pending) return fail(h, DSL_ERR_INVALID, "dsl_slab_exchange: a split force pass is in flight");
  return link_exchange(h);
}

int dsl_slab_replan(dsl_handle* h) {
  CHECK_HANDLE(h);
  if (!h->link) return fail(h, DSL_ERR_INVALID, "dsl_slab_replan: call dsl_slab_attach first");
  return link_replan(h);
}

int dsl_slab_wcsph_step(dsl_handle* h, int nsteps) {
  CHECK_HANDLE(h);
  if (!h->link) return fail(h, DSL_ERR_INVALID, "dsl_slab_wcsph_step: call dsl_slab_attach first");
  SlabLink& L = *h->link;
  const bool alone = L.lo < 0 && L.hi < 0;
  // (Round 2 could replay the step's kernel segments as hipGraphs: measured 5 % SLOWER than launching its ~20 kernels one
  // by one -- the host needs 45 us per step for them either way, the GPU 500 us; profiles/r02_slab_graphs_on_off.txt.
  // Removed in round 4.)
  for (int s = 0; s < nsteps; ++s) {
    if (!L.ghosts_in)
      if (int rc = link_exchange(h)) return rc;  // first step: migrants + 2h ghosts from both neighbours
    L.ghosts_in = false;
    // counting sort (drops the previous step's ghosts), densities of owned + ghosts
    if (int rc = build_grid(h, false)) return rc;
    if (int rc = density_pass(h)) return rc;
    if (alone || !L.overlap) {
      // forces and integration of the owned particles (ghosts are marked for removal), band pack, transfer, append
      if (int rc = force_integrate(h)) return rc;
      if (!alone) {
        if (int rc = slab_pack_on(h, h->stream, L.width_full, L.width, false, L.lo >= 0 ? L.send[0] : nullptr,
                                  L.hi >= 0 ? L.send[1] : nullptr, L.cap_full, L.cap_x))
          return rc;
        if (int rc = link_post(h, h->stream)) return rc;
        if (int rc = link_append(h)) return rc;
        L.ghosts_in = true;
      }
    } else {
      // band layers first; their pack and the RCCL transfer (side stream) run under the interior launch
      if (int rc = force_integrate(h, 1)) return rc;
      h->split_pending = true;
      if (int rc = slab_pack_on(h, h->stream, L.width_full, h->split_width, true, L.lo >= 0 ? L.send[0] : nullptr,
                                L.hi >= 0 ? L.send[1] : nullptr, L.cap_full, L.cap_x))
        return rc;
      // the interior launch is queued BEHIND the pack and BEFORE the transfer is posted: the GPU goes straight from
      // the pack into it, the transfer kernels (side stream, behind the pack's event) join it
      HIP_TRY(h, hipEventRecord(L.ev_pack, h->stream));
      h->split_pending = false;
      if (int rc = force_integrate(h, 2)) return rc;
      HIP_TRY(h, hipStreamWaitEvent(L.comm_stream, L.ev_pack, 0));
      if (int rc = link_post(h, L.comm_stream)) return rc;
      HIP_TRY(h, hipEventRecord(L.ev_xfer, L.comm_stream));
      HIP_TRY(h, hipStreamWaitEvent(h->stream, L.ev_xfer, 0));
      if (int rc = link_append(h)) return rc;
      L.ghosts_in = true;
    }
    h->steps++;
    L.steps++;
    if (!alone && L.steps % L.replan_every == 0)
      if (int rc = link_replan(h)) return rc;
  }
  return DSL_OK;
}

int dsl_slab_pcisph_step(dsl_handle* h, int nsteps) {
  CHECK_HANDLE(h);
  if (!h->link) return fail(h, DSL_ERR_INVALID, "dsl_slab_pcisph_step: call dsl_slab_attach first");
  if (!h->pci_active) return fail(h, DSL_ERR_INVALID, "dsl_slab_pcisph_step: call dsl_pcisph_begin first");
  SlabLink& L = *h->link;
  const bool alone = L.lo < 0 && L.hi < 0;
  const bool reduce = L.comm && L.comm->nranks > 1;
  for (int s = 0; s < nsteps; ++s) {
    if (!L.ghosts_in)
      if (int rc = link_exchange(h)) return rc;
    L.ghosts_in = false;
    h->pci_split_guard = true;
    if (int rc = pci_begin_step(h)) return rc;
    for (int it = 0; it < h->prm.pci_max_iters; ++it) {
      if (int rc = pci_iterate(h)) return rc;
      // the early-out of pcisph_darwin.go:95-98 is decided by the maximum over all ranks: the error
      // word holds a non-negative float, whose bits order like an unsigned integer
      if (reduce)
        if (int rc = xfer_all_reduce_max(h, L.comm, &h->dstats->pci_cur_err_bits, 1, h->stream)) return rc;
      if (int rc = pci_check(h)) return rc;
    }
    h->pci_split_guard = false;
    if (int rc = pci_end_step(h)) return rc;  // Update; ghosts are marked for removal
    if (int rc = link_exchange(h)) return rc;
    L.steps++;
    if (!alone && L.steps % L.replan_every == 0)
      if (int rc = link_replan(h)) return rc;
  }
  return DSL_OK;
}

int dsl_set_ids(dsl_handle* h, const int32_t* ids, size_t count) {
  CHECK_HANDLE(h);
  int ncur = 0;
  if (int rc = host_count(h, &ncur)) return rc;
  if (!ids || count != (size_t)ncur) return fail(h, DSL_ERR_INVALID, "dsl_set_ids: count must equal the particle count");
  if (h->nb > 0) return fail(h, DSL_ERR_UNSUPPORTED, "dsl_set_ids: boundary particles are told apart by their ids");
  // ids follow the host order of dsl_upload: slot s currently holds host index cur_ids[s]
  std::vector<int> cur(count), out(count);
  HIP_TRY(h, hipMemcpyAsync(cur.data(), h->ids[h->cur_ids], count * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (size_t s = 0; s < count; ++s) {
    if (cur[s] < 0 || (size_t)cur[s] >= count) return fail(h, DSL_ERR_INVALID, "dsl_set_ids: ids were already replaced");
    out[s] = ids[cur[s]];
  }
  HIP_TRY(h, hipMemcpyAsync(h->ids[h->cur_ids], out.data(), count * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->ids_global = true;
  return DSL_OK;
}

int dsl_reset_forces(dsl_handle* h) {
  CHECK_HANDLE(h);
  h->forces_uniform = true;
  return DSL_OK;
}

int dsl_get_stats(dsl_handle* h, dsl_stats* out) {
  CHECK_HANDLE_ONLY(h);
  if (!out) return fail(h, DSL_ERR_INVALID, "null out");
  DevStats d{};
  HIP_TRY(h, hipMemcpyAsync(&d, h->dstats, sizeof(d), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float v2 = 0.f, f2 = 0.f;
  std::memcpy(&v2, &d.max_vel_bits, 4);
  std::memcpy(&f2, &d.max_f_bits, 4);
  out->max_vel = std::sqrt(v2);  // float overload: correctly rounded, as the per-particle square root was
  out->max_f = std::sqrt(f2);
  if (h->prm.max_vel > out->max_vel) out->max_vel = h->prm.max_vel;  // maxVel starts at the parameter (fluid.go:25)
  std::memcpy(&out->pci_max_error, &d.pci_last_err_bits, 4);
  out->pci_iters = d.pci_iters;
  out->steps = h->steps;
  for (int a = 0; a < 3; ++a) out->grid_dims[a] = h->c.dims[a];
  out->grid_cells = h->c.ncell;
  out->max_cell_count = d.max_cell_count;
  return DSL_OK;
}

int dsl_sync(dsl_handle* h) {
  CHECK_HANDLE_ONLY(h);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return DSL_OK;
}

int dsl_timing_enable(dsl_handle* h, int on) {
  CHECK_HANDLE_ONLY(h);
  h->timing = on < 0 ? 0 : (on > 2 ? 1 : on);
  return DSL_OK;
}
int dsl_timing_reset(dsl_handle* h) {
  CHECK_HANDLE_ONLY(h);
  if (int rc = drain_timing(h)) return rc;
  for (int k = 0; k < DSL_K_COUNT; ++k) {
    h->total_ms[k] = 0.0;
    h->launches[k] = 0;
  }
  return DSL_OK;
}
int dsl_timing_get(dsl_handle* h, int kid, double* avg_ms, int64_t* launches) {
  CHECK_HANDLE_ONLY(h);
  if (kid < 0 || kid >= DSL_K_COUNT) return fail(h, DSL_ERR_INVALID, "bad kernel id");
  if (int rc = drain_timing(h)) return rc;
  if (avg_ms) *avg_ms = h->launches[kid] ? h->total_ms[kid] / (double)h->launches[kid] : 0.0;
  if (launches) *launches = h->launches[kid];
  return DSL_OK;
}

}  // extern "C"
