// kernels_skin.hpp -- the WCSPH step with neighbour LISTS that live for several steps (FAST math, single domain).
//
// Round 3 left the step at grid build 0.50 + candidate sweep 0.65 + force walk 0.87 ms for 16M particles, every step,
// on a neighbourhood that in a dam-break changes by a fraction of a per cent per step (sph_field.go:155-172 runs over
// the sampler's lists the same way; the reference itself rebuilds its sampler only every 4th call, fluid.go:208-215).
// Here the expensive part -- counting sort, 216-candidate sweep -- is done once per k steps against the cut-off
// h (1 + s) on a grid whose cells are h (1 + s) wide, and its result is kept as one LIST per particle:
//
//   * an entry is the 16-bit BYTE OFFSET of the neighbour's record in the LDS image of the particle's tile (the tile
//     tables, the slot order and therefore these offsets stay valid until the next rebuild); eight 16-bit fields make
//     a 16-byte chunk -- the first field of a particle's first chunk is its entry count -- and chunk c of slot g lives
//     at lists[c * capacity + g]: one global_load_dwordx4 per lane brings eight neighbours.  While the skin step owns
//     the grid a tile is 4 x 4 x 3 cells (TileGrid::tbz): with cells h (1 + s) wide that is the ~512 targets and ~1900
//     staged records of the plain step's 4 x 4 x 4 tile, so one 8-wave workgroup walks a tile with every wave busy
//     and two of them share a CU's LDS, as in the plain step's kernels;
//   * a list is valid while no particle is further than s h / 2 from the REFERENCE position its list was built at:
//     every pair closer than h now had references closer than h (1 + s).  The reference is x + tau v -- where the particle
//     will be about half way through the lists' life (SkinState::tau, chosen by k_skin_decide at every rebuild; any
//     reference is sound) -- written out in sorted order by the build's scatter and never overwritten between rebuilds.
//     The displacement is measured ON THE DEVICE (k_force_list: |x - reference|) and k_skin_decide, the first kernel of
//     every step, turns its maximum into the step's `rebuild` flag; the kernels of the rebuild chain are launched every
//     step and return at once while the flag is down.  No host round trip, and the decision is a function of the
//     simulation state alone;
//   * between rebuilds a step is two kernels: k_density_list (sph_field.go:155-172 over the list: 7 VALU
//     instructions per LISTED pair where the sweep spends 7 per CANDIDATE, 216 of them) and k_force_list (the fused
//     gradient + viscosity + Update walk: one v_and / v_lshr per pair where the mask walk spends six on bit
//     arithmetic, no per-run set-up, every lane busy until ITS list ends).  A pair beyond h contributes exactly 0
//     (q = max(1 - r/h, 0)), so the sums are the reference's sums over { |x_i - x_j| < h }.
//
// The sort's output buffer is a THIRD position / velocity set (Z): a step reads X (or Z when this step rebuilt -- the
// kernels pick by the device flag) and always writes Y, so the host knows every pointer without knowing the flag;
// which of the two slot -> particle maps is current is device state too and is read back when the host next needs it.
#pragma once

#include "kernels_tiled.hpp"

namespace dsl {

constexpr int kLEntries = 8;                // 16-bit fields per 16-byte chunk
constexpr int kLMaxChunks = 12;             // at most 95 listed neighbours per particle; more: the global-memory sweep
constexpr int kLMaxEntries = kLMaxChunks * kLEntries - 1;
constexpr int kLBlock = 512;                // 8 waves, two workgroups per CU (a tile's targets beyond 512 share lanes: for_each_target)
constexpr unsigned int kLGlobal = 0xffffu;  // count sentinel: this particle takes the global-memory sweep
constexpr float kSkinTauSteps = 16.0f;      // cap on SkinState::tau, in steps
constexpr int kLQuads = 14;                 // staging: quads of 4 records per row and pass (36 rows x 14 = 504 lanes)
static_assert(kTRows * kLQuads <= kLBlock, "one quad per lane");

// first kernel of every skin step: this step's rebuild flag.  The lists are good for the positions the previous step
// left if no particle is further than s h / 2 from the reference position its list was built at; k_force_list measures
// exactly that (the references sit in sorted order in a set of their own, which no step overwrites).
__global__ void k_skin_decide(SkinState* st) {
  const float d = __builtin_sqrtf(__uint_as_float(st->disp2_bits)) * (1.0f + 1.0e-6f);
  const bool rb = st->force != 0 || !(d <= st->budget);
  st->rebuild = rb ? 1 : 0;
  // (more than one particle in 64 without a list -- wide cells crowd the tiles' LDS images -- and the global-memory
  // sweep they fall back to costs more than the lists save: the plain step, whose cells are h wide, is the faster one)
  if (st->unlisted * 64 > st->n_live) st->give_up = 1;
  if (rb) {
    st->ids_sel ^= 1;
    st->n_rebuilds += 1;
    st->unlisted = 0;
    st->fields_own = st->fields_padded = 0u;
    // how far ahead of the particles this build's reference positions run: the fastest particle uses up `predict` of
    // the budget at the build itself (vmax2: the previous step's velocities, i.e. the ones this build sorts; unknown
    // after an upload -- +inf -- gives 0).  At most kSkinTauSteps steps: beyond that nothing was gained in the runs measured.
    const float vmax = __builtin_sqrtf(__uint_as_float(st->vmax2_bits)) * (1.0f + 1.0e-6f);
    float tau = 0.0f;
    if (st->predict > 0.0f && vmax > 0.0f && vmax < 3.0e38f) tau = fminf(st->predict * st->budget / vmax, kSkinTauSteps * st->dt);
    st->tau = tau;
  }
  st->disp = d;
  st->force = 0;
  st->disp2_bits = 0u;
  st->vmax2_bits = 0u;
  st->n_steps += 1;
  // a rebuild adds 3.1 ms to a step that costs 1.4 where the plain step costs 1.9 (16M, end of round 4): lists pay while
  // they live 6-7 steps or more.  The library gives up later than that, at five rebuilds in 16 steps: a suspension lasts
  // thousands of steps, and the first steps of a run -- the lattice's jitter relaxing -- rebuild two or three times as often
  // as the hundreds that follow (with three in 16 a skin of 0.065 was suspended at step 32 of the bench scene and the run
  // lost 17 %: profiles/r04_skin_sweep_final.jsonl)
  st->history = (st->history << 1) | (rb ? 1u : 0u);
  if (st->n_steps >= 16 && __builtin_popcount(st->history & 0xffffu) >= 5) st->give_up = 1;
}

// a tile's far-away record (the first pad record of staged row 0), as a list entry
__device__ __forceinline__ unsigned int pad_entry(const TileMeta& m) { return (unsigned int)(m.row_lds[1] - kTPad) << 4; }

// ---------------------------------------------------------------------------------
// masks of the wide candidate sweep (k_density_pair<.., WIDE>) -> lists.  One lane per target; the tile's table says
// which LDS records a run's mask bits stand for.
//
// The passes over a tile's targets are the walks' own (for_each_target<true, kLBlock>: one lane per target while more
// than half a block is left, then k lanes per target), because a list is laid out for the wave that will walk it:
//   * in a full pass all 64 lists of a wave are padded with the far-away record to the length of the longest of them,
//     and field 0 of every first chunk holds that common length (in fields, itself included): the walk's loop is
//     then a scalar one -- no per-lane counters, no ballots, no selects between "my next chunk" and padding;
//   * a target of a short pass keeps its own length in field 0 (its k lanes split the chunks among them).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kLBlock) void k_list_build_v1(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                        const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                        const unsigned int* __restrict__ nmask, int mstride,
                                                        uint4* __restrict__ lists, int lstride, SkinState* st,
                                                        SkinGate gate, CSoa3 p, float wide_thr) {
  if (gate.closed()) return;
  __shared__ TileMeta metas[2];
  // A lane finishes its chunks at its own pace (the bit loops below diverge), and a chunk stored the moment it was full
  // went out as a lone 16-byte write: 1.6 GB of them per 16M rebuild, 1.7 ms.  The first kLStaged chunks of a list
  // therefore wait in LDS (a lane reads back only what it wrote itself: no barrier) and leave together at the end of the
  // pass, 1 KB per wave and store.  Longer lists (more than 63 entries: rare) store the rest directly.
  constexpr int kLStaged = 8;
  __shared__ uint4 held[kLStaged][kLBlock];
  __shared__ unsigned int tile_fields[2];
  const int tid = threadIdx.x;
  if (tid < 2) tile_fields[tid] = 0u;
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) tile_meta_store(metas[0], tile_meta_request(desc, di));
  for (int cur = 0; have; cur ^= 1) {
    const TileMeta& m = metas[cur];
    sync_lds();  // this tile's table is visible, the other copy is free
    have = feed.pop(di);
    int table_word = 0;
    if (have) table_word = tile_meta_request(desc, di);
    const int ntarg = m.tprefix[kTB * kTB];
    const unsigned int pad = pad_entry(m);
    if (m.overflow != 0 && tid == 0) atomicAdd(&st->unlisted, ntarg);
    for_each_target<true, kLBlock>(ntarg, tid, tid, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      if (SHARED && sub != 0) return;  // (a short pass's list is built by the first lane of its group)
      const TileTarget tt = tile_target(m, t);
      const int g = tt.g;
      // a 128-bit shift register: the k-th field pushed into a chunk ends up in bits 16k .. 16k+15.  Field 0 of chunk 0
      // is the length, patched in at the end.
      unsigned int b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u;
      int n = 0;  // fields pushed, the length's included
      auto push = [&](unsigned int entry) {
        b0 = __builtin_amdgcn_alignbit(b1, b0, 16);
        b1 = __builtin_amdgcn_alignbit(b2, b1, 16);
        b2 = __builtin_amdgcn_alignbit(b3, b2, 16);
        b3 = (b3 >> 16) | (entry << 16);
        n += 1;
        if ((n & (kLEntries - 1)) == 0) {
          const int ch = (n >> 3) - 1;
          if (ch < kLStaged) held[ch][tid] = make_uint4(b0, b1, b2, b3);
          else if (ch < kLMaxChunks) lists[(size_t)ch * lstride + g] = make_uint4(b0, b1, b2, b3);
        }
      };
      auto emit = [&](int rec) {
        if (rec != tt.own) push((unsigned int)rec << 4);  // (the particle itself is no entry: sph_field.go:164)
      };
      const bool masks = m.overflow == 0;  // (a tile beyond the LDS budget was not swept: the global-memory sweep)
      if (masks) {
        // the nine runs' bounds and mask words first, all loads in flight together (one after the other they were nine
        // to eighteen exposed round trips per target)
        int rj[9], rje[9];
        unsigned int w1[9], w2[9];
#pragma unroll
        for (int ri = 0; ri < 9; ++ri) {
          tile_run(m, tt.srow + (ri / 3 - 1) * kTH + (ri % 3 - 1), tt.lx, rj[ri], rje[ri]);
          const int len = rje[ri] - rj[ri];
          w1[ri] = (len > 0 && len <= 64) ? nmask[(size_t)ri * mstride + g] : 0u;
          w2[ri] = (len > 32 && len <= 64) ? nmask[(size_t)(kMaskHigh + ri) * mstride + g] : 0u;
        }
        push(0u);  // the length's field
#pragma unroll
        for (int ri = 0; ri < 9; ++ri) {
          const int j = rj[ri], je = rje[ri], len = je - j;
          if (len > 64) {  // a run without masks (rare: three cells of more than 21 particles each): the sweep's test, from global memory
            const int rr = tt.srow + (ri / 3 - 1) * kTH + (ri % 3 - 1);
            const int g0 = m.row_gs[rr] - m.row_lds[rr];  // global slot of LDS record r of this row: g0 + r
            const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
            for (int r = j; r < je; ++r) {
              const float dx = xi - p.x[g0 + r], dy = yi - p.y[g0 + r], dz = zi - p.z[g0 + r];
              const float tq = __builtin_fmaf(-dist2<true>(dx, dy, dz), c.inv_hh, 1.0f);
              if (tq > wide_thr) emit(r);
            }
            continue;
          }
          // bit b of a mask word <-> record top - b, top counted from the chunk's length rounded up to the sweep's unroll of 4
          unsigned int w = w1[ri];
          int top = j + ((min(len, 32) + 3) & ~3) - 1;
          while (w != 0u) {
            const int b = __builtin_ctz(w);
            w &= w - 1u;
            emit(top - b);
          }
          w = w2[ri];
          top = j + 32 + ((len - 32 + 3) & ~3) - 1;
          while (w != 0u) {
            const int b = __builtin_ctz(w);
            w &= w - 1u;
            emit(top - b);
          }
        }
      }
      const bool fits = masks && n <= kLMaxChunks * kLEntries;
      if (masks && !fits) st->list_overflow = 1;  // (benign race: every writer stores 1)
      int fields = fits ? n : 0;
      const int own_fields = fields;
      if constexpr (!SHARED) {  // the wave's longest list (ballots count the lanes that are here: a pass's last wave may be short)
        int wmax = 0;
#pragma unroll
        for (int bit = 6; bit >= 0; --bit) {
          const int cand = wmax | (1 << bit);
          if (__builtin_amdgcn_ballot_w64(fields >= cand) != 0ull) wmax = cand;
        }
        fields = wmax;
      }
      // statistics of the build (DSL_OPT_SKIN_FIELDS_*): summed in LDS, one pair of global atomics per tile
      atomicAdd(&tile_fields[0], (unsigned int)own_fields);
      atomicAdd(&tile_fields[1], (unsigned int)(fits ? fields : 0));
      const int nch = (fields + kLEntries - 1) / kLEntries;  // (full pass: the same for the whole wave)
      if (fits)
        while (n < nch * kLEntries) push(pad);  // the far-away record up to the end of the wave's last chunk
      // the held chunks leave together; field 0 of the first one is the length (or the mark of an unlisted target)
      for (int ch = 0; ch < min(nch, kLStaged); ++ch) {
        uint4 v = held[ch][tid];
        if (ch == 0) v.x = (v.x & 0xffff0000u) | (unsigned int)fields;
        if (fits) lists[(size_t)ch * lstride + g] = v;
      }
      if (!fits) lists[g] = make_uint4(kLGlobal, 0u, 0u, 0u);
    });
    if (have) tile_meta_store(metas[cur ^ 1], table_word);
    sync_lds();
    if (tid == 0) {
      atomicAdd(&st->fields_own, tile_fields[0]);
      atomicAdd(&st->fields_padded, tile_fields[1]);
      tile_fields[0] = tile_fields[1] = 0u;
    }
  }
}

// ---------------------------------------------------------------------------------
// The same lists, built in lock step (round 4; k_list_build_v1 above is the first form, kept as DSL_OPT_LIST_BUILD = 0).
// v1 walks a target's 18 mask words one `while (w)` loop after the other: every loop lasts as long as the wave's busiest
// lane (~97 trips for ~38 entries), and a lane finishes its chunks at its own pace (hence `held`).  Here a lane first
// puts its NON-EMPTY words, each with the record its bit 0 stands for, into a queue of its own in LDS ([item][lane]:
// conflict-free whatever the item index); the wave's longest list is known from the popcounts before a single entry
// exists, and ONE scalar loop over that many fields follows in which every lane produces exactly one field per trip --
// the next bit of its current word (the next word when that is exhausted), or the far-away record once its list has
// ended.  All lanes complete chunk c in the same trip: a chunk leaves as one 1 KB store per wave, straight from
// registers.  Entry order, padding and field 0 are v1's: the two kernels write the same bytes (but for the entries of
// runs without masks, which v1 has in their run's place and ascending, this one last and descending).
// ---------------------------------------------------------------------------------
constexpr int kLQueue = 24;  // queue items per target: 18 mask words + the words of runs without masks (more: unlisted)
__global__ __launch_bounds__(kLBlock, 4) void k_list_build(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                        const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                        const unsigned int* __restrict__ nmask, int mstride,
                                                        uint4* __restrict__ lists, int lstride, SkinState* st,
                                                        SkinGate gate, CSoa3 p, float wide_thr) {
  if (gate.closed()) return;
  __shared__ TileMeta metas[2];
  __shared__ unsigned int qword[kLQueue][kLBlock];
  __shared__ unsigned short qtop[kLQueue][kLBlock];
  __shared__ unsigned int tile_fields[2];
  const int tid = threadIdx.x;
  if (tid < 2) tile_fields[tid] = 0u;
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) tile_meta_store(metas[0], tile_meta_request(desc, di));
  for (int cur = 0; have; cur ^= 1) {
    const TileMeta& m = metas[cur];
    sync_lds();  // this tile's table is visible, the other copy is free
    have = feed.pop(di);
    int table_word = 0;
    if (have) table_word = tile_meta_request(desc, di);
    const int ntarg = m.tprefix[kTB * kTB];
    const unsigned int pad = pad_entry(m);
    if (m.overflow != 0 && tid == 0) atomicAdd(&st->unlisted, ntarg);
    for_each_target<true, kLBlock>(ntarg, tid, tid, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      if (SHARED && sub != 0) return;  // (a short pass's list is built by the first lane of its group)
      const TileTarget tt = tile_target(m, t);
      const int g = tt.g;
      const bool masks = m.overflow == 0;  // (a tile beyond the LDS budget was not swept: the global-memory sweep)
      // ---- the queue: non-empty mask words in v1's order, own record struck out, and their popcount
      int nq = 0, total = 0;
      bool too_many = false;
      auto enqueue = [&](unsigned int w, int top) {
        if (w != 0u) {
          if (nq < kLQueue) {
            qword[nq][tid] = w;
            qtop[nq][tid] = (unsigned short)top;
          } else {
            too_many = true;
          }
          nq += 1;
          total += __builtin_popcount(w);
        }
      };
      if (masks) {
        unsigned int run[9], w1[9], w2[9];  // (a run's LDS records: first | end << 16, as the tile table packs them)
        unsigned int unmasked = 0u;
#pragma unroll
        for (int ri = 0; ri < 9; ++ri) {  // (all loads in flight together)
          run[ri] = (unsigned int)m.run[(tt.srow + (ri / 3 - 1) * kTH + (ri % 3 - 1)) * kTB + tt.lx - 1];
          const int len = (int)(run[ri] >> 16) - (int)(run[ri] & 0xffffu);
          w1[ri] = (len > 0 && len <= 64) ? nmask[(size_t)ri * mstride + g] : 0u;
          w2[ri] = (len > 32 && len <= 64) ? nmask[(size_t)(kMaskHigh + ri) * mstride + g] : 0u;
          if (len > 64) unmasked |= 1u << ri;
        }
#pragma unroll
        for (int ri = 0; ri < 9; ++ri) {
          const int j = (int)(run[ri] & 0xffffu), len = (int)(run[ri] >> 16) - j;
          // bit b of a mask word <-> record top - b, top counted from the chunk's length rounded up to the sweep's unroll of 4
          unsigned int a = w1[ri], b = w2[ri];
          const int top1 = j + ((min(len, 32) + 3) & ~3) - 1;
          const int top2 = j + 32 + ((len - 32 + 3) & ~3) - 1;
          if (ri == 4) {  // (the particle itself is no entry: sph_field.go:164; its record lies in its own row's run)
            const int b1 = top1 - tt.own, b2 = top2 - tt.own;
            if (b1 >= 0 && b1 < 32) a &= ~(1u << b1);
            if (b2 >= 0 && b2 < 32) b &= ~(1u << b2);
          }
          enqueue(a, top1);
          enqueue(b, top2);
        }
        // runs without masks (rare: three cells of more than 21 particles each) come last: the sweep's test, from global
        // memory, 32 records per word (bit b <-> record top - b, as in the masks)
        while (unmasked != 0u) {
          const int ri = __builtin_ctz(unmasked);
          unmasked &= unmasked - 1u;
          const int rr = tt.srow + (ri / 3 - 1) * kTH + (ri % 3 - 1);
          int j, je;
          tile_run(m, rr, tt.lx, j, je);
          const int g0 = m.row_gs[rr] - m.row_lds[rr];  // global slot of LDS record r of this row: g0 + r
          const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
          for (int r0 = j; r0 < je; r0 += 32) {
            unsigned int w = 0u;
            const int top = r0 + 31;
            for (int r = r0; r < min(r0 + 32, je); ++r) {
              const float dx = xi - p.x[g0 + r], dy = yi - p.y[g0 + r], dz = zi - p.z[g0 + r];
              const float tq = __builtin_fmaf(-dist2<true>(dx, dy, dz), c.inv_hh, 1.0f);
              if (tq > wide_thr && r != tt.own) w |= 1u << (top - r);
            }
            enqueue(w, top);
          }
        }
      }
      const int n_own = 1 + total;  // fields, the length's included
      const bool fits = masks && !too_many && n_own <= kLMaxChunks * kLEntries;
      if (masks && !fits) st->list_overflow = 1;  // (benign race: every writer stores 1)
      int fields = fits ? n_own : 0;
      const int own_fields = fields;
      // the longest list of the lanes that are here (a full pass pads every list to it; a short pass only runs that long)
      int wmax = 0;
#pragma unroll
      for (int bit = 6; bit >= 0; --bit) {
        const int cand = wmax | (1 << bit);
        if (__builtin_amdgcn_ballot_w64(fields >= cand) != 0ull) wmax = cand;
      }
      if constexpr (!SHARED) fields = fits ? wmax : 0;
      // statistics of the build (DSL_OPT_SKIN_FIELDS_*): summed in LDS, one pair of global atomics per tile
      atomicAdd(&tile_fields[0], (unsigned int)own_fields);
      atomicAdd(&tile_fields[1], (unsigned int)fields);
      const int nch = (fields + kLEntries - 1) / kLEntries;       // this lane's chunks (full pass: the wave's)
      const int nch_wave = (wmax + kLEntries - 1) / kLEntries;    // the loop's
      // ---- one field per trip and lane
      unsigned int b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u;            // a 128-bit shift register: field k of a chunk ends up in bits 16k .. 16k+15
      unsigned int k0 = 0u, k1 = 0u, k2 = 0u, k3 = 0u;            // chunk 0, kept for its length field
      int qi = 0;
      unsigned int w = 0u, top = 0u;
      // (the next item is fetched one trip before it is needed)
      unsigned int wn = (fits && nq > 0) ? qword[0][tid] : 0u, tn = (fits && nq > 0) ? qtop[0][tid] : 0u;
      const int nq_live = fits ? nq : 0;
      for (int it = 0; it < nch_wave * kLEntries; ++it) {
        unsigned int entry = pad;
        if (it == 0) {
          entry = 0u;  // the length's field
        } else {
          if (w == 0u && qi < nq_live) {
            w = wn;
            top = tn;
            qi += 1;
            if (qi < nq_live) {
              wn = qword[qi][tid];
              tn = qtop[qi][tid];
            }
          }
          if (w != 0u) {
            const int b = __builtin_ctz(w);
            w &= w - 1u;
            entry = (top - (unsigned int)b) << 4;
          }
        }
        b0 = __builtin_amdgcn_alignbit(b1, b0, 16);
        b1 = __builtin_amdgcn_alignbit(b2, b1, 16);
        b2 = __builtin_amdgcn_alignbit(b3, b2, 16);
        b3 = (b3 >> 16) | (entry << 16);
        if ((it & (kLEntries - 1)) == kLEntries - 1) {
          const int ch = it >> 3;
          if (ch == 0) {
            k0 = b0, k1 = b1, k2 = b2, k3 = b3;
          } else if (ch < nch) {
            lists[(size_t)ch * lstride + g] = make_uint4(b0, b1, b2, b3);
          }
        }
      }
      // field 0 of the first chunk is the length (or the mark of an unlisted target)
      if (fits) lists[g] = make_uint4((k0 & 0xffff0000u) | (unsigned int)fields, k1, k2, k3);
      else lists[g] = make_uint4(kLGlobal, 0u, 0u, 0u);
    });
    if (have) tile_meta_store(metas[cur ^ 1], table_word);
    sync_lds();
    if (tid == 0) {
      atomicAdd(&st->fields_own, tile_fields[0]);
      atomicAdd(&st->fields_padded, tile_fields[1]);
      tile_fields[0] = tile_fields[1] = 0u;
    }
  }
}

// ---------------------------------------------------------------------------------
// staging of a tile for the list walks: lane t owns quad t % kLQuads of staged row t / kLQuads (and every kLQuads-th
// quad behind it: wide cells make long rows).  load4(g, float4 out[NF]) / store(record, float rec[NF], real)
// ---------------------------------------------------------------------------------
template <int NF, class Load4, class Store>
__device__ __forceinline__ void stage_tile(const TileMeta& m, Load4&& load4, Store&& store) {
  const int t = threadIdx.x, r = t / kLQuads, k = t - r * kLQuads;
  if (r >= kTRows) return;
  const int len = m.row_len[r], gs = m.row_gs[r], ls = m.row_lds[r];
  for (int q = k; 4 * q < len; q += kLQuads) {
    float4 v[NF];
    load4(gs + 4 * q, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (4 * q + i < len) {
        float rec[NF];
#pragma unroll
        for (int a = 0; a < NF; ++a) rec[a] = f4_at(v[a], i);
        store(ls + 4 * q + i, rec, true);
      }
    }
  }
  const int npad = m.row_lds[r + 1] - ls - len;
  if (k < npad) {
    float rec[NF];
#pragma unroll
    for (int a = 0; a < NF; ++a) rec[a] = 0.0f;
    store(ls + len + k, rec, false);
  }
}

__device__ __forceinline__ const float4& lds_at(const float4* base, unsigned int byte_offset) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_offset);
}

// a chunk in registers (plain words: HIP's uint4 goes through memory when a member is rewritten)
struct Chunk {
  unsigned int x, y, z, w;
};
__device__ __forceinline__ Chunk load_chunk(const uint4* __restrict__ lists, size_t index) {
  const uint4 v = lists[index];
  return Chunk{v.x, v.y, v.z, v.w};
}
// a chunk whose eight fields all name the tile's far-away record
__device__ __forceinline__ Chunk pad_chunk(unsigned int pad) {
  const unsigned int w = pad | (pad << 16);
  return Chunk{w, w, w, w};
}
// a particle's first chunk: the count out of field 0, the far-away record in its place
__device__ __forceinline__ unsigned int take_count(Chunk& e, unsigned int pad) {
  const unsigned int cnt = e.x & 0xffffu;
  e.x = (e.x & 0xffff0000u) | pad;
  return cnt;
}
__device__ __forceinline__ int chunks_of(unsigned int cnt) { return cnt == kLGlobal ? 0 : (int)((cnt + kLEntries) >> 3); }

// ---------------------------------------------------------------------------------
// Walking a list.  Full pass (one lane per target): the wave's lists all have the length field 0 names, so the loop over
// its fields is scalar; `fetch(word)` requests the records of a 32-bit word's two fields, the walk bodies below
// pipeline them.  Short pass (k lanes per target, lane `sub` takes chunks sub, sub + k, ...): plain and unpipelined --
// the handful of targets beyond a block, or a sparse tile.
// ---------------------------------------------------------------------------------
// shared pass: calls body(chunk) for every chunk of this lane's share; lanes whose share has ended get the far-away chunk
// until the wave's longest share is through (`fields` = field 0 of the target's first chunk, from the group's lane 0)
template <class Body>
__device__ __forceinline__ void walk_shared(const uint4* __restrict__ lists, int lstride, int g, unsigned int pad, int sub,
                                            int k, unsigned int fields, Body&& body) {
  const int nch = fields == kLGlobal ? 0 : (int)((fields + kLEntries - 1) >> 3);
  for (int ch = sub; __builtin_amdgcn_ballot_w64(ch < nch) != 0ull; ch += k) {
    Chunk e = pad_chunk(pad);
    if (ch < nch) e = load_chunk(lists, (size_t)ch * lstride + g);
    if (ch == 0) e.x = (e.x & 0xffff0000u) | pad;  // (the length's field)
    body(e);
  }
}
// field 0 of slot g's first chunk, as the group's lane 0 reads it
__device__ __forceinline__ unsigned int shared_fields(const uint4* __restrict__ lists, int g, int sub) {
  unsigned int f = 0u;
  if (sub == 0) f = lists[g].x & 0xffffu;
  return (unsigned int)__shfl((int)f, (int)((threadIdx.x & (kWave - 1)) - sub), kWave);
}

// ---------------------------------------------------------------------------------
// D over the lists: SPHField.Density (sph_field.go:155-172) + P/rho^2, FAST arithmetic -- the records, the test and
// the sum are k_density_pair's, taken over the listed pairs only.
// ---------------------------------------------------------------------------------
// (walks its share of the tile list the static way: the tile queue's one register more -- thread 0's pending draw -- costs
// this HBM-bound kernel its sixth wave per SIMD, 0.43 -> 0.55 ms; the force walk gains 5 % from the queue)
__global__ __launch_bounds__(kLBlock, 4) void k_density_list(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                             const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                             const int* __restrict__ cell_start, const SkinState* st,
                                                             CSoa3 pX, CSoa3 pZ, const uint4* __restrict__ lists,
                                                             int lstride, float* __restrict__ rho,
                                                             float* __restrict__ pterm) {
  __shared__ TileMeta metas[2];
  __shared__ float4 A[kTCap];
  const int tid = threadIdx.x;
  const CSoa3 p = st->rebuild ? pZ : pX;  // (this step rebuilt: the sort's output; else the previous step's)
  TileFeed feed(desc_of, *n_tiles);
  int di = 0;
  bool have = feed.pop(di);
  if (have) tile_meta_store(metas[0], tile_meta_request(desc, di));
  for (int cur = 0; have; cur ^= 1) {
    const TileMeta& m = metas[cur];
    sync_lds();  // the previous tile's walk is over, this tile's table is visible
    have = feed.pop(di);
    int table_word = 0;
    if (have) table_word = tile_meta_request(desc, di);
    const float ox = __int_as_float(m.centre[0]), oy = __int_as_float(m.centre[1]), oz = __int_as_float(m.centre[2]);
    const int ntarg = m.tprefix[kTB * kTB];
    const bool staged = m.overflow == 0;
    const unsigned int pad = pad_entry(m);
    // the first pass's targets (one lane each when the tile holds more than half a block): their first two chunks are
    // requested before the staging loads
    Chunk q0 = pad_chunk(pad), q1 = q0;
    const bool first_full = ntarg > kLBlock / 2;
    TileTarget tt0{0, 0, 0, 0};
    // (lanes are permuted inside their wave so that each of ds_read_b128's 16-lane groups holds 16 CONSECUTIVE targets --
    // a cell and a half, whose k-th list entries are records close to each other, i.e. on different banks: a third of
    // the bank conflicts of the unpermuted walk, profiles/r04.  The wave's set of targets is the same: k_list_build's
    // padding holds.)
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(tid & (kWave - 1));
    if (first_full && tperm < ntarg) {
      tt0 = tile_target(m, tperm);
      q0 = load_chunk(lists, tt0.g);
      q1 = load_chunk(lists, (size_t)lstride + tt0.g);
    }
    if (staged)
      stage_tile<3>(
          m,
          [&](int gg, float4* o) {
            o[0] = load4u(p.x + gg);
            o[1] = load4u(p.y + gg);
            o[2] = load4u(p.z + gg);
          },
          [&](int slot, const float* o, bool real) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, -1.0e30f);  // pad: q = clamp(-1e30 + ...) = 0
            if (real) {
              const float x = o[0] - ox, y = o[1] - oy, z = o[2] - oz;
              v = make_float4(x, y, z, -c.inv_hh * __builtin_fmaf(z, z, __builtin_fmaf(y, y, x * x)));
            }
            A[slot] = v;
          });
    if (have) tile_meta_store(metas[cur ^ 1], table_word);
    sync_lds();
    for_each_target<true, kLBlock>(ntarg, tid, tperm, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      TileTarget tt;
      unsigned int fields;
      if constexpr (SHARED) {
        tt = tile_target(m, t);
        fields = shared_fields(lists, tt.g, sub);
      } else {
        if (t == tperm) tt = tt0;
        else {
          tt = tile_target(m, t);
          q0 = load_chunk(lists, tt.g);
          q1 = load_chunk(lists, (size_t)lstride + tt.g);
        }
        fields = take_count(q0, pad);
      }
      const int g = tt.g;
      float acc = 0.0f;
      if (staged && fields != kLGlobal) {
        const float4 me = A[tt.own];
        const float two_hh = 2.0f * c.inv_hh;
        const float sx = two_hh * me.x, sy = two_hh * me.y, sz = two_hh * me.z, a0 = 1.0f + me.w;
        float acc0 = 0.0f, acc1 = 0.0f;
        float4 cq[4];
        auto fetch4 = [&](unsigned int w0, unsigned int w1) {
          cq[0] = lds_at(A, w0 & 0xffffu);
          cq[1] = lds_at(A, w0 >> 16);
          cq[2] = lds_at(A, w1 & 0xffffu);
          cq[3] = lds_at(A, w1 >> 16);
        };
        auto sum4 = [&](const float4 (&r)[4]) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float q = fma_clamp01(r[u].z, sz, __builtin_fmaf(r[u].y, sy, __builtin_fmaf(r[u].x, sx, r[u].w + a0)));
            if (u & 1) acc1 = __builtin_fmaf(q, q, acc1);
            else acc0 = __builtin_fmaf(q, q, acc0);
          }
        };
        if constexpr (SHARED) {
          walk_shared(lists, lstride, g, pad, sub, k, fields, [&](const Chunk& e) {
            fetch4(e.x, e.y);
            float4 c1[4] = {cq[0], cq[1], cq[2], cq[3]};
            fetch4(e.z, e.w);
            sum4(c1);
            sum4(cq);
          });
        } else {
          // The wave's lists have one length (k_list_build): a scalar loop over its fields, software-pipelined in groups
          // of four -- the next group's records are requested before the current group's arithmetic, across chunk
          // boundaries too, so that the LDS pipe and the VALU work at the same time.
          const int nf = __builtin_amdgcn_readfirstlane((int)fields), nch = (nf + kLEntries - 1) >> 3;
          const uint4* lp = lists + (size_t)2 * lstride + g;  // chunk ch + 2 of this slot
          Chunk e = q0, e1 = q1;
          fetch4(e.x, e.y);
          for (int ch = 0;;) {
            float4 c1[4] = {cq[0], cq[1], cq[2], cq[3]};
            fetch4(e.z, e.w);  // the chunk's second half
            sum4(c1);
            if (8 * ch + 4 >= nf) break;  // (the half just requested is never read: four reads more)
            const Chunk en = e1;
            if (ch + 2 < nch) e1 = load_chunk(lp, 0);
            lp += lstride;
            float4 c2[4] = {cq[0], cq[1], cq[2], cq[3]};
            fetch4(en.x, en.y);
            sum4(c2);
            e = en;
            ch += 1;
            if (8 * ch >= nf) break;
          }
        }
        acc = acc0 + acc1;
      } else if (sub == 0) {  // unlisted (a tile beyond the LDS budget, a list beyond kLMaxEntries): the grid's 27 cells
        const float xi = p.x[g], yi = p.y[g], zi = p.z[g];
        float a = 0.0f;
        for_each_grid_candidate(c, cell_start, xi, yi, zi, [&](int j) {
          if (j == g) return;
          const float dx = xi - p.x[j], dy = yi - p.y[j], dz = zi - p.z[j];
          const float r2 = dist2<true>(dx, dy, dz);
          if (r2 < c.hh) {
            const float q = __builtin_fmaf(-r2, c.inv_hh, 1.0f);
            a = __builtin_fmaf(q, q, a);
          }
        });
        acc = a;
      }
      if constexpr (SHARED) {
        for (int o = 1; o < k; o <<= 1) acc += __shfl_xor(acc, o, kWave);  // the lanes of a group are active together
        if (sub != 0) return;
      }
      acc *= c.mass * c.A;
      rho[g] = acc;
      // (an isolated particle's own term must be a harmless 0 rather than 0/0: see k_density_tiled)
      const float pr = tait_eos<true>(c, acc, c.eos_d0_grad);
      pterm[g] = acc > 0.0f ? dsl_div<true>(pr, acc * acc) : 0.0f;
    });
  }
}

// ---------------------------------------------------------------------------------
// [G] [V] X U over the lists: the fused WCSPH force + integrate of k_force_integrate_tiled (FAST arithmetic: the same
// per-pair operations, sph_field.go:175-200,251-269, fluid.go:175-197), walking list entries instead of mask bits.
// Also tracks the step's max |v|^2 for the skin's displacement bound.
// ---------------------------------------------------------------------------------
template <bool WANT_G, bool WANT_V, bool QUEUE>
__global__ __launch_bounds__(kLBlock, 4) void k_force_list(DevConsts c, TileGrid tg, const int* __restrict__ desc_of,
                                                           const int* __restrict__ n_tiles, const int* __restrict__ desc,
                                                           const int* __restrict__ cell_start, SkinState* st, CSoa3 pX,
                                                           CSoa3 vX, CSoa3 pZ, CSoa3 vZ, CSoa3 pR, const float* __restrict__ rho,
                                                           const float* __restrict__ pterm,
                                                           const uint4* __restrict__ lists, int lstride, Soa3 pout,
                                                           Soa3 vout, DevStats* stats, int* __restrict__ walk_ctr) {
  __shared__ TileMeta metas[2];
  __shared__ float4 A[kTCap];  // x, y, z, P/rho^2
  __shared__ float4 B[kTCap];  // vx, vy, vz, 1/rho
  __shared__ int feed_slot[2];
  const int tid = threadIdx.x;
  const bool rb = st->rebuild != 0;
  const CSoa3 pin = rb ? pZ : pX, vin = rb ? vZ : vX;
  unsigned int vbits = 0u, fbits = 0u, dbits = 0u;
  typename TileSource<QUEUE>::type feed = TileSource<QUEUE>::make(desc_of, *n_tiles, walk_ctr, feed_slot);
  int di = 0;
  bool have = feed.pop(di);
  if (have) tile_meta_store(metas[0], tile_meta_request(desc, di));
  for (int cur = 0; have; cur ^= 1) {
    const TileMeta& m = metas[cur];
    sync_lds();
    have = feed.pop(di);
    int table_word = 0;
    if (have) table_word = tile_meta_request(desc, di);
    const int ntarg = m.tprefix[kTB * kTB];
    const bool staged = m.overflow == 0;
    const unsigned int pad = pad_entry(m);
    Chunk q0 = pad_chunk(pad), q1 = q0;
    const bool first_full = ntarg > kLBlock / 2;
    TileTarget tt0{0, 0, 0, 0};
    // (lanes are permuted inside their wave so that each of ds_read_b128's 16-lane groups holds 16 CONSECUTIVE targets --
    // a cell and a half, whose k-th list entries are records close to each other, i.e. on different banks: a third of
    // the bank conflicts of the unpermuted walk, profiles/r04.  The wave's set of targets is the same: k_list_build's
    // padding holds.)
    const int tperm = (tid & ~(kWave - 1)) + b128_group_slot(tid & (kWave - 1));
    if (first_full && tperm < ntarg) {
      tt0 = tile_target(m, tperm);
      q0 = load_chunk(lists, tt0.g);
      q1 = load_chunk(lists, (size_t)lstride + tt0.g);
    }
    if (staged)
      stage_tile<8>(
          m,
          [&](int gg, float4* o) {
            o[0] = load4u(pin.x + gg);
            o[1] = load4u(pin.y + gg);
            o[2] = load4u(pin.z + gg);
            o[3] = WANT_G ? load4u(pterm + gg) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (WANT_V) {
              o[4] = load4u(vin.x + gg);
              o[5] = load4u(vin.y + gg);
              o[6] = load4u(vin.z + gg);
              o[7] = load4u(rho + gg);
            } else {
              o[4] = o[5] = o[6] = o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          },
          [&](int slot, const float* o, bool real) {
            float4 a = make_float4(kFar, kFar, kFar, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (real && o[0] == o[0] && o[1] == o[1] && o[2] == o[2]) {  // (a NaN position is nobody's neighbour)
              a = make_float4(o[0], o[1], o[2], o[3]);
              // 1/rho = 0 for an isolated particle (rho = 0): it only ever meets itself
              if constexpr (WANT_V) b = make_float4(o[4], o[5], o[6], o[7] > 0.0f ? __builtin_amdgcn_rcpf(o[7]) : 0.0f);
            }
            A[slot] = a;
            if constexpr (WANT_V) B[slot] = b;
          });
    if (have) tile_meta_store(metas[cur ^ 1], table_word);
    sync_lds();
    for_each_target<true, kLBlock>(ntarg, tid, tperm, [&](auto shared_c, int t, int sub, int k) {
      constexpr bool SHARED = decltype(shared_c)::value;
      TileTarget tt;
      unsigned int fields;
      if constexpr (SHARED) {
        tt = tile_target(m, t);
        fields = shared_fields(lists, tt.g, sub);
      } else {
        if (t == tperm) tt = tt0;
        else {
          tt = tile_target(m, t);
          q0 = load_chunk(lists, tt.g);
          q1 = load_chunk(lists, (size_t)lstride + tt.g);
        }
        fields = take_count(q0, pad);
      }
      const int g = tt.g;
      float px, py, pz, vx, vy, vz;
      float gx = 0.f, gy = 0.f, gz = 0.f, lx_ = 0.f, ly_ = 0.f, lz_ = 0.f, lw_ = 0.f;
      const bool listed = staged && fields != kLGlobal;
      if (listed) {
        const float4 a = A[tt.own];
        px = a.x, py = a.y, pz = a.z;
        const float pti = WANT_G ? a.w : 0.f;
        if (px == kFar) px = py = pz = __uint_as_float(0x7fc00000u);  // staged as a pad record: its position is NaN
        if constexpr (WANT_V) {
          const float4 b = B[tt.own];
          vx = b.x, vy = b.y, vz = b.z;
        } else {
          vx = vin.x[g], vy = vin.y[g], vz = vin.z[g];
        }
        if constexpr (WANT_G || WANT_V) {
          const float ninvh = -c.inv_h;
          auto accum = [&](const float4& ra, const float4& rb2) {
            const float dx = ra.x - px, dyy = ra.y - py, dzz = ra.z - pz;
            float r2 = __builtin_fmaf(dzz, dzz, __builtin_fmaf(dyy, dyy, dx * dx));
            r2 = fmaxf(r2, 1.0e-30f);  // (a coincident pair: keeps rsq finite, all terms stay 0)
            const float rinv = __builtin_amdgcn_rsqf(r2);
            const float dist = r2 * rinv;
            const float q = fma1_clamp01_uniform(dist, ninvh);
            if constexpr (WANT_G) {
              const float kk = (q * q) * (pti + ra.w) * rinv;
              gx = __builtin_fmaf(dx, kk, gx);
              gy = __builtin_fmaf(dyy, kk, gy);
              gz = __builtin_fmaf(dzz, kk, gz);
            }
            if constexpr (WANT_V) {
              // sum_j (v_j - v_i) w_j = sum_j v_j w_j - v_i sum_j w_j
              const float w = q * rb2.w;
              lx_ = __builtin_fmaf(rb2.x, w, lx_);
              ly_ = __builtin_fmaf(rb2.y, w, ly_);
              lz_ = __builtin_fmaf(rb2.z, w, lz_);
              lw_ += w;
            }
          };
          struct Pair1 {
            float4 a, b;
          };
          auto fetch1 = [&](unsigned int off) {
            Pair1 r;
            r.a = lds_at(A, off);
            r.b = r.a;
            if constexpr (WANT_V) r.b = lds_at(B, off);
            return r;
          };
          if constexpr (SHARED) {
            walk_shared(lists, lstride, g, pad, sub, k, fields, [&](const Chunk& e) {
              const unsigned int w[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const Pair1 r0 = fetch1(w[u] & 0xffffu), r1 = fetch1(w[u] >> 16);
                accum(r0.a, r0.b);
                accum(r1.a, r1.b);
              }
            });
          } else {
            // the wave's lists have one length (k_list_build): a scalar loop over the 32-bit words of the chunks;
            // p0, p1 hold the records of the word being walked, every step consumes one and requests the field two
            // ahead, so four reads are in flight while a pair's ~25 VALU instructions issue
            const int nf = __builtin_amdgcn_readfirstlane((int)fields), nch = (nf + kLEntries - 1) >> 3;
            const uint4* lp = lists + (size_t)2 * lstride + g;  // chunk ch + 2 of this slot
            Chunk e = q0, e1 = q1;
            Pair1 p0 = fetch1(e.x & 0xffffu), p1 = fetch1(e.x >> 16);
            auto word = [&](unsigned int next_word) {
              Pair1 cu = p0;
              p0 = fetch1(next_word & 0xffffu);
              accum(cu.a, cu.b);
              cu = p1;
              p1 = fetch1(next_word >> 16);
              accum(cu.a, cu.b);
            };
            for (int ch = 0;;) {
              word(e.y);
              if (8 * ch + 2 >= nf) break;  // (the word just requested is never read: four reads more)
              word(e.z);
              if (8 * ch + 4 >= nf) break;
              word(e.w);
              if (8 * ch + 6 >= nf) break;
              const Chunk en = e1;
              if (ch + 2 < nch) e1 = load_chunk(lp, 0);
              lp += lstride;
              word(en.x);
              e = en;
              ch += 1;
              if (8 * ch >= nf) break;
            }
          }
        }
      } else {
        px = pin.x[g], py = pin.y[g], pz = pin.z[g];
        vx = vin.x[g], vy = vin.y[g], vz = vin.z[g];
        if (sub == 0) {
          float accG[3] = {0.f, 0.f, 0.f}, accV[3] = {0.f, 0.f, 0.f};
          if constexpr (WANT_G || WANT_V)
            force_sweep<true, WANT_G, WANT_V>(c, grid_neigh(cell_start), g, pin, vin, rho, pterm, accG, accV, nullptr, nullptr);
          gx = accG[0], gy = accG[1], gz = accG[2];
          lx_ = accV[0], ly_ = accV[1], lz_ = accV[2];
        }
      }
      if constexpr (SHARED) {
        for (int o = 1; o < k; o <<= 1) {  // the lanes of a group are active together
          gx += __shfl_xor(gx, o, kWave);
          gy += __shfl_xor(gy, o, kWave);
          gz += __shfl_xor(gz, o, kWave);
          lx_ += __shfl_xor(lx_, o, kWave);
          ly_ += __shfl_xor(ly_, o, kWave);
          lz_ += __shfl_xor(lz_, o, kWave);
          lw_ += __shfl_xor(lw_, o, kWave);
        }
        if (sub != 0) return;
      }
      if (listed) {
        if constexpr (WANT_V) {
          lx_ = __builtin_fmaf(-vx, lw_, lx_);
          ly_ = __builtin_fmaf(-vy, lw_, ly_);
          lz_ = __builtin_fmaf(-vz, lw_, lz_);
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
      }
      float fx = c.reset[0], fy = c.reset[1], fz = c.reset[2];  // (forces are uniform in a skin step: Update left them so)
      if constexpr (WANT_G) {
        const float dm = rho[g] * c.mass * c.pressure_sign;
        fx = __builtin_fmaf(gx, dm, fx);
        fy = __builtin_fmaf(gy, dm, fy);
        fz = __builtin_fmaf(gz, dm, fz);
      }
      if constexpr (WANT_V) {
        fx = __builtin_fmaf(lx_, c.mu, fx);
        fy = __builtin_fmaf(ly_, c.mu, fy);
        fz = __builtin_fmaf(lz_, c.mu, fz);
      }
      fx += c.ext[0];
      fy += c.ext[1];
      fz += c.ext[2];
      integrate_core(c, fx, fy, fz, px, py, pz, vx, vy, vz, vbits, fbits);
      {  // how far from the reference position its list was built at (the sort's output: slot order has not changed since)
        const float ex = px - pR.x[g], ey = py - pR.y[g], ez = pz - pR.z[g];
        const unsigned int db = nonneg_bits(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex)));
        dbits = db > dbits ? db : dbits;
      }
      pout.x[g] = px;
      pout.y[g] = py;
      pout.z[g] = pz;
      vout.x[g] = vx;
      vout.y[g] = vy;
      vout.z[g] = vz;
    });
  }
  feed.finish();
  wave_atomic_max(&stats->max_vel_bits, vbits);
  wave_atomic_max(&stats->max_f_bits, fbits);
  wave_atomic_max(&st->disp2_bits, dbits);
  wave_atomic_max(&st->vmax2_bits, vbits);  // (what the next rebuild's tau is chosen by: |v| before the walls, >= the stored one)
}

}  // namespace dsl
