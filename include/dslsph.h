/*
 * dslsph.h -- C ABI of libdslsph.so, the MI355X (gfx950) SPH particle-step engine that
 * sits where dieselfluid's OpenCL path sits (compute/gpu + solver/pcisph GPU driver).
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes.  No torch types,
 * no C++ types.  Bound from Go through cgo (bindings/go/dslsph), from C++ through
 * dieselfluid_amd/host/dieselfluid.hpp and from Python through ctypes
 * (dieselfluid_amd/_lib.py).  See INTEGRATION.md for the reference-side stubs.
 *
 * Every entry point cites the reference interface it replaces
 * (file:line relative to the dieselfluid repository root).
 *
 * Threading: a handle is single-owner (not thread-safe); every call re-selects the
 * handle's device first because goroutines migrate between OS threads.  Host pointers
 * are never retained after a call returns (cgo pointer rule).
 */
#ifndef DSLSPH_H
#define DSLSPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSL_ABI_VERSION 1

/* status codes: 0 = ok; negative = error, text from dsl_last_error() */
enum {
  DSL_OK = 0,
  DSL_ERR_INVALID = -1,  /* bad argument / bad state            */
  DSL_ERR_DEVICE = -2,   /* HIP runtime error                   */
  DSL_ERR_NOMEM = -3,    /* host or device allocation failure   */
  DSL_ERR_UNSUPPORTED = -4,
  DSL_ERR_OVERFLOW = -5  /* slab exchange: a band message, the particle capacity or the split margin was outgrown */
};

/* Named device buffers; the names are the ones the reference registers in
 * solver/pcisph/pcisph_gpu_darwin.go:67-76 ("positions", "velocities", "forces",
 * "densities", "pressures", "temps").  Host layout is the reference's: xyz interleaved
 * float32 (model/particle_array.go:5-15); the library converts to SoA on device. */
enum {
  DSL_BUF_POSITIONS = 0,  /* N*3 floats */
  DSL_BUF_VELOCITIES = 1, /* N*3 floats */
  DSL_BUF_FORCES = 2,     /* N*3 floats */
  DSL_BUF_DENSITIES = 3,  /* N floats   */
  DSL_BUF_PRESSURES = 4,  /* N floats   */
  DSL_BUF_PCI_POSITIONS = 5,  /* "temps".pos, N*3: predictor state of pcisph_darwin.go:28-41 */
  DSL_BUF_PCI_VELOCITIES = 6, /* "temps".vel, N*3 */
  DSL_BUF_COUNT = 7
};

enum {
  DSL_NEIGH_LSH_REF = 0, /* the reference's sampler: 8-bit random-projection LSH, 255 buckets, 100
                            samples with duplicates (sampler/lsh/lsh.go); parity mode, EXACT math only */
  DSL_NEIGH_GRID = 1     /* all j with |xi-xj| < h via uniform-grid counting sort */
};
enum {
  DSL_MATH_EXACT = 0, /* IEEE float32, one rounding per reference operation, f64 pow */
  DSL_MATH_FAST = 1   /* fused multiply-add, hardware rcp/rsq, float32 pow          */
};

/* Parameter block.  The first nine fields are the reference's "sizes" and "floats"
 * blocks (pcisph_gpu_darwin.go:60-61); the rest are the reference's compile-time
 * constants made explicit.  dsl_params_reference() fills in the reference's values. */
typedef struct dsl_params {
  uint32_t struct_size; /* = sizeof(dsl_params); ABI check */
  uint32_t abi_version; /* = DSL_ABI_VERSION               */
  /* "sizes" */
  int32_t n_particles;     /* field.Particles.N()                   */
  int32_t n_boundary;      /* Total()-N() (0 in sph.Init: fluid.go:70) */
  int32_t lsh_buckets;     /* carried for the Go side; unused here  */
  int32_t lsh_bucket_size; /* carried for the Go side; unused here  */
  /* "floats" */
  float dt;      /* SPH.CFL()   model/sph/fluid.go:111-114 (0.01) */
  float mass;    /* field.Mass()                         (1.0)    */
  float delta;   /* SPH.Delta() fluid.go:204,221-273              */
  float max_vel; /* SPH.MaxV()  initial value                     */
  float h;       /* kernel length fluid.go:48            (1.0)    */
  /* constants of the reference */
  float ref_density; /* ParticleArray.D0() particle_array.go:26 (N/8)  */
  float mu;          /* VISCOSITY_WATER fluid.go:18 (1.3059)           */
  float eos_w;       /* model/model.go:94 (2.15)                       */
  float eos_gamma;   /* model/model.go:93 (7.16)                       */
  float eos_d0_grad; /* FLUID_DENSITY, field_types.go:41 (87.0)        */
  float pressure_sign;       /* +1: fluid.go:168-169 adds the gradient  */
  int32_t visc_running_mass; /* 1: sph_field.go:265 running-sum * m     */
  float force_reset[3];      /* fluid.go:193 (0,-9.81*m,0)              */
  float external[3];         /* wcsph.go:19  (0,-9.81,0)                */
  int32_t wcsph_pressure_force; /* 0: absent from wcsph.go:14-26        */
  int32_t wcsph_viscosity;      /* 0: absent from wcsph.go:14-26        */
  int32_t pci_max_iters;        /* pcisph_darwin.go:49 (5)              */
  float pci_max_error;          /* pcisph_darwin.go:50 (0.01)           */
  /* build-defined: axis-aligned wall box applied at the end of Update */
  int32_t walls;
  float box_min[3], box_max[3];
  float restitution;
  /* uniform grid for the neighbour table; cell edge = h */
  float grid_min[3], grid_max[3];
  int32_t neigh_mode; /* DSL_NEIGH_GRID */
  int32_t math_mode;  /* DSL_MATH_*     */
  int32_t capacity;   /* particle slots to allocate (>= n_particles); 0 = n_particles.
                         Slab ranks need room for migrants and ghosts. */
  /* build-defined terms of BASELINE configs[4] (no reference counterpart), 0 = off:
   *   xsph_eps: positions advect with v + eps * sum_j (m/rho_j) (v_j - v_i) F(r_ij)
   *   st_kappa: cohesion force F_i += kappa * sum_j m (x_j - x_i) F(r_ij)          */
  float xsph_eps, st_kappa;
  /* 0 (default): slots inside a grid cell are ascending in particle id after every neighbour build, so
   * every neighbour sum runs in a fixed order and results are reproducible bit for bit from run to run
   * (and, in DSL_MATH_EXACT, equal to the reference's sums taken cell by cell).  1: keep the order the
   * counting sort's atomics produce (saves the ordering pass of the scatter; last-bit differences between runs). */
  int32_t sort_unordered;
  /* reserved[0]: budget, in MiB, for each of the two side arrays that cost memory per GRID CELL rather than per particle
   * (the cells' key rows of the one-pass in-cell ordering: 128 B per cell; the PCISPH query rows: 512 B per cell); 0 = the
   * larger of 4 GiB and 64 B per particle slot.  An array beyond its budget is not allocated and the kernels take the
   * forms that need none (two-pass ordering, the queries' sorted array): same results, a few per cent slower.
   * reserved[1..3]: 0. */
  int32_t reserved[4];
} dsl_params;

/* Counters the reference keeps on the host (fluid.go:25-26,186-191) plus the PCISPH loop
 * outcome (pcisph_darwin.go:46-98). */
typedef struct dsl_stats {
  float max_vel, max_f;
  float pci_max_error;
  int32_t pci_iters;
  int64_t steps;
  int32_t grid_dims[3];
  int32_t grid_cells;
  int32_t max_cell_count; /* most crowded cell at the last neighbour build */
} dsl_stats;

/* kernel ids for dsl_timing_get */
enum {
  DSL_K_CELL_RANK = 0,
  DSL_K_SCAN = 1,
  DSL_K_SCATTER = 2,
  DSL_K_DENSITY = 3,
  DSL_K_FORCE_INTEGRATE = 4, /* fused WCSPH pressure+viscosity+external+Update */
  DSL_K_PRESSURE = 5,
  DSL_K_VISCOUS = 6,
  DSL_K_GRADIENT = 7,
  DSL_K_EXTERNAL = 8,
  DSL_K_UPDATE = 9,
  DSL_K_PCI_PREDICT = 10,
  DSL_K_PCI_DENSITY = 11,
  DSL_K_TILE_LIST = 12, /* non-empty 4x4x4-cell tiles for the LDS-tiled kernels */
  DSL_K_NEIGH_LISTS = 13, /* skin step: wide candidate sweep + neighbour lists (rebuild steps only) */
  DSL_K_COUNT = 14
};

typedef struct dsl_handle dsl_handle;

/* sph.Init's constants for an n3^3 system (model/sph/fluid.go:41-88). */
int dsl_params_reference(dsl_params *out, int n3);

/* gpu.InitOpenCL + gpu.New_ComputeGPU + New_GPUPredictorCorrector's ten RegisterBuffer
 * calls (compute/gpu/gpu.go:45-119,314; pcisph_gpu_darwin.go:36-76).  `device` is the
 * HIP device ordinal. */
int dsl_create(const dsl_params *params, int device, dsl_handle **out);
int dsl_destroy(dsl_handle *h);

/* Scalars may change between steps (dt, delta, mu, ...); n_particles and the grid box
 * may not. */
int dsl_set_params(dsl_handle *h, const dsl_params *params);
int dsl_get_params(dsl_handle *h, dsl_params *out);

/* ComputeGPU.PassFloatBuffer / ReadFloatBuffer (compute/gpu/gpu.go:343-352,332-341) and
 * the per-step EnqueueReadBufferFloat32 of pcisph_gpu_darwin.go:276-277.  Blocking;
 * `count` is the number of floats and must match the buffer. */
int dsl_upload(dsl_handle *h, int buffer, const float *host, size_t count);
int dsl_download(dsl_handle *h, int buffer, float *host, size_t count);

/* ParticleArray.AddBoundaryParticles (model/particle_array.go:123-128; fed by SPHField.BoundaryParticles,
 * model/field/sph_field.go:75-85, from Mesh.GenerateBoundaryParticles, geom/mesh/mesh.go:60-76):
 * appends count/3 position-only particles behind the current ones; needs room in
 * dsl_params.capacity.  dsl_params.n_boundary > 0 at creation is the same as NewParticleArray(n, nb, ..):
 * nb zero-filled boundary slots.  With boundary particles DSL_BUF_POSITIONS holds Total() = N + Nb
 * particles (fluid first), every other buffer N.
 * Semantics are the reference's, quirks included (oracle/dsl_oracle.c restates them first):
 *   - boundary particles are candidates of every neighbour sum with Get()'s values: position, density
 *     0, press 0, velocity 0 (particle_array.go:94-117); Density/DensityF simply count them
 *     (sph_field.go:143,163), Gradient and LaplacianForce divide by their density 0 without a guard
 *     (sph_field.go:183,259): a fluid particle with a boundary neighbour gets NaN / Inf there;
 *   - Get(N()) -- the FIRST boundary particle -- is the zero particle: it takes part in every sum at the
 *     origin, while the positions slice keeps (and dsl_download returns) what was uploaded;
 *   - no pass writes a boundary particle; Update does not move it.
 * Not available in slab mode or after dsl_set_ids. */
int dsl_add_boundary_particles(dsl_handle *h, const float *host_positions, size_t count);

/* Render hand-off without the per-step full read-back of pcisph_gpu_darwin.go:276-277
 * (SURVEY 8f rank 1):
 * dsl_download_decimated: every stride-th particle (host index 0, stride, 2*stride, ...) of a
 *   3-component buffer, count = 3*ceil(N/stride) floats; blocking.
 * dsl_device_pointers: the live device arrays themselves (SoA, cell-sorted slot order) for a
 *   consumer that can read device memory (another HIP kernel, graphics interop):
 *   xyz[3] component pointers of `buffer`, ids = slot -> particle index, n = slot count.
 *   Valid until the next call that steps, uploads or rebuilds the neighbour table. */
int dsl_download_decimated(dsl_handle *h, int buffer, int stride, float *host, size_t count);
int dsl_device_pointers(dsl_handle *h, int buffer, const float **xyz, const int32_t **ids, int *n);

/* lsh.Allocate's random projection vectors (sampler/lsh/lsh.go:32-40): hash_bits x 3 floats.
 * The reference draws them from math/rand seeded with the wall-clock second; here they are an
 * explicit input.  DSL_NEIGH_LSH_REF only. */
int dsl_set_hash_vectors(dsl_handle *h, const float *vectors, int hash_bits);
/* HashSampler.GetData1D (lsh.go:70-80): the flattened buckets x lsh_bucket_size table */
int dsl_lsh_download_table(dsl_handle *h, int32_t *out, size_t count);

/* SPH.NN() / HashSampler.UpdateSampler (fluid.go:100-102, sampler/lsh/lsh.go:126-133):
 * rebuilds the neighbour table (cell hash, histogram, prefix sum, counting-sort scatter). */
int dsl_build_neighbours(dsl_handle *h);

/* The whole-array passes of model/sph/fluid.go, one to one. */
int dsl_density_pass(dsl_handle *h);                  /* DensityAll            :127-131 */
int dsl_pressure_pass(dsl_handle *h);                 /* PressureAll           :134-142 */
int dsl_viscous_pass(dsl_handle *h);                  /* ViscousAll            :146-152 */
int dsl_external_pass(dsl_handle *h, const float f[3]); /* ExternalAll         :155-161 */
int dsl_gradient_pressure_pass(dsl_handle *h);        /* GradientPressureForce :164-172 */
int dsl_update_pass(dsl_handle *h);                   /* Update                :175-197 */
/* Fused form of [GradientPressureForce][ViscousAll] ExternalAll PressureAll Update used by
 * dsl_wcsph_step; needs densities from dsl_density_pass. */
int dsl_force_pass(dsl_handle *h);

/* The SPHField operators no solver calls (model/field/sph_field.go), evaluated on the current
 * state (densities as left by the last dsl_density_pass).  tensor_buffer is
 * DSL_BUF_VELOCITIES or DSL_BUF_FORCES; scalar_buffer is DSL_BUF_DENSITIES (DensityField) or
 * DSL_BUF_PRESSURES (PressureField.Value = TaitEos(rho, 87.0, 0), field_types.go:39-42).
 * Results come back in host order; blocking. */
int dsl_field_divergence(dsl_handle *h, int tensor_buffer, float *host_out, size_t count);   /* Div         :203-227, N   */
int dsl_field_curl(dsl_handle *h, int tensor_buffer, float *host_out, size_t count);         /* Curl        :272-294, 3N  */
int dsl_field_laplacian(dsl_handle *h, int scalar_buffer, float *host_out, size_t count);    /* Laplacian   :230-248, N   */
int dsl_field_interpolate(dsl_handle *h, int scalar_buffer, const float *host_positions,     /* Interpolate :124-135      */
                          size_t n_positions, float *host_out);

/* Step drivers: one iteration of WCSPH.Run (solver/wcsph/wcsph.go:14-26) and of
 * PciMethod.Run (solver/pcisph/pcisph_darwin.go:43-101) per step.  Asynchronous: they
 * return once the work is queued; dsl_sync/dsl_download/dsl_get_stats wait. */
int dsl_wcsph_step(dsl_handle *h, int nsteps);
int dsl_pcisph_begin(dsl_handle *h); /* pcisph_darwin.go:28-41 predictor copies */
int dsl_pcisph_step(dsl_handle *h, int nsteps);
/* The same step in pieces, for hosts that have to look at the iteration error between the
 * correction sweep and the convergence check (pcisph_darwin.go:95-98) -- slab mode, where the
 * error is the maximum over all ranks:
 *   DSL_PCI_BEGIN_STEP  NN, DensityAll, ViscousAll, loop set-up        (:43-51)
 *   DSL_PCI_ITERATE     predict, DensityF + pressure accumulate, gradient force (:52-94); leaves
 *                       this handle's max density error in a device word
 *   DSL_PCI_CHECK       the early-out test of that word (:95-98); later ITERATEs are no-ops
 *                       once it has passed
 *   DSL_PCI_END_STEP    Update (:101)
 * dsl_pcisph_error_word copies the device word out (store = 0) or in (store = 1), device to
 * device and asynchronously; its bits order like the non-negative float they hold, so a MAX
 * all-reduce over uint32 is the global error. */
enum { DSL_PCI_BEGIN_STEP = 0, DSL_PCI_ITERATE = 1, DSL_PCI_CHECK = 2, DSL_PCI_END_STEP = 3 };
int dsl_pcisph_phase(dsl_handle *h, int phase);
int dsl_pcisph_error_word(dsl_handle *h, uint32_t *dev_word, int store);
/* DensityF's query points are the predictor's positions, and the reference never brings the predictor back to the
 * particles (pcisph_darwin.go:28-41: `_pos`, `_vel` are seeded once, advanced in every correction iteration): within
 * tens of steps they are cells, then whole tiles away from the particle they belong to.  The library then sorts the
 * QUERIES into the particles' grid cells before every DensityF sweep (same candidates, same order, same arithmetic per
 * query: DSL_MATH_EXACT results do not change by a bit).  mode 0 (default): switch when 0.2 % of the queries have left
 * their particle's 4x4x4-cell tile -- looked at every 4 steps through an asynchronous copy of the device's
 * counters that the next look reads (no stall; the decision trails the drift by 4 to 8 steps), a one-way switch until
 * the next dsl_pcisph_begin; 1: always; -1: never.  (While the handle's stream is being captured into a graph of the
 * host's the automatic look is skipped: set the mode.)  In slab mode every rank latches from ITS OWN counters: in
 * DSL_MATH_FAST the binned and un-binned sweeps sum in different orders, so a multi-rank FAST run's last bits depend on
 * the decomposition -- set the mode explicitly (the same on every rank) where that matters; DSL_MATH_EXACT does not care.
 * The slab step drivers do not look at `escaped` themselves: a host that runs PCISPH across slabs polls it.
 * dsl_pcisph_get_binning: the mode; whether the next correction iteration sorts its queries; and, slab mode (blocking
 * if asked for), whether a query point of an owned particle has drifted more than h beyond a slab plane since
 * dsl_pcisph_begin, i.e. out of what the 2h ghost band covers: from then on this rank's predicted densities -- the
 * pressure accumulator and the iteration's convergence error, nothing else reads them (pcisph_darwin.go:76-98) -- miss
 * neighbours that live on another rank.  Any of the three pointers may be NULL. */
int dsl_pcisph_set_binning(dsl_handle *h, int mode);
int dsl_pcisph_get_binning(dsl_handle *h, int *mode, int *active, int *escaped);

int dsl_get_stats(dsl_handle *h, dsl_stats *out);
int dsl_sync(dsl_handle *h); /* Queue.Finish() pcisph_gpu_darwin.go:261,271 */

/* Run on an existing hipStream_t (e.g. the caller framework's current stream) so that the
 * caller's copies/collectives and the engine's kernels are ordered.  NULL is HIP's default
 * stream.  dsl_use_own_stream goes back to the handle's private non-blocking stream. */
int dsl_set_stream(dsl_handle *h, void *hip_stream);
int dsl_use_own_stream(dsl_handle *h);

/* Per-kernel device timing with HIP events recorded on the launch stream.  on = 1: every kernel
 * (1.8 % of a 2.5 ms step at 16M particles, ~10 % of a 0.5 ms slab step); on = 2: only the
 * dominant kernels of a step (DSL_K_DENSITY, DSL_K_FORCE_INTEGRATE, DSL_K_PCI_DENSITY); 0: off. */
int dsl_timing_enable(dsl_handle *h, int on);
int dsl_timing_reset(dsl_handle *h);
int dsl_timing_get(dsl_handle *h, int kernel_id, double *avg_ms, int64_t *launches);

/* Debug/parity aid: state in the device's current (cell-sorted) order.
 * ids[s] is the original particle index held by slot s. */
int dsl_download_sorted(dsl_handle *h, int buffer, float *host, size_t count);
int dsl_download_ids(dsl_handle *h, int32_t *ids, size_t count);
int dsl_download_cell_start(dsl_handle *h, int32_t *cell_start, size_t count);

/* ---- multi-GPU: spatial slabs, one process per GPU -----------------------------------
 * The reference has no multi-process path; these entry points are
 * what a host needs to run one slab per GPU and exchange a halo with its two neighbours
 * (over RCCL, by whatever transport the host owns).
 *
 * A message is a DEVICE buffer of dsl_slab_message_floats(cap_full, cap_xonly) floats:
 *   header  : 7 words; [0] = number of full records, [1] = number of position-only records
 *             (int32 bits)
 *   full    : cap_full records of 7 floats: x,y,z,vx,vy,vz and the global particle id as
 *             raw int32 bits -- every owned particle within width_full of the plane, and the
 *             migrants beyond it
 *   x-only  : cap_xonly records of 4 floats: x,y,z and the global id (int32 bits) -- the rest of the
 *             band (out to `width`); they only feed the receiver's ghost densities.  The id keeps
 *             every cell of every rank in the order of a single-domain run (ascending id), which is
 *             what makes DSL_MATH_EXACT slabs bit-identical to it
 * Messages have a fixed size, so a step needs no host-side counts and no host
 * synchronisation: the live particle count stays on the device.  With width_full = h and
 * width = 2h the receiver recomputes the ghosts' densities itself and the force pass needs
 * no second exchange.
 *
 * dsl_slab_config : this handle owns [lo,hi) along `axis` (use -INFINITY / INFINITY at the
 *                   domain ends); particles outside are ghosts: they take part in the
 *                   neighbour sums but are not integrated and are dropped at the next
 *                   neighbour build.
 * dsl_slab_pack   : packs the bands of both sides (either pointer may be NULL) from the
 *                   current state.  Asynchronous.
 * dsl_slab_append : appends the records of a received message behind the current
 *                   particles; full records are owned if inside [lo,hi), ghosts otherwise.
 *                   Asynchronous.
 * dsl_slab_status : [0] largest count that did not fit a message or the particle capacity
 *                   since creation (0 = none), [1] 1 if the split step's margin was ever
 *                   exceeded, [2],[3] largest full / position-only band counts since the last
 *                   reset; blocking.  dsl_slab_overflow returns [0] only.
 * After appending, dsl_build_neighbours drops the previous step's ghosts (the integrate
 * kernels mark them with NaN positions).  A particle that has just crossed the plane stays
 * one more step as a ghost of its old owner.
 *
 * Split step (hides the exchange behind the interior force pass; DSL_MATH_FAST only):
 *   dsl_slab_split(h, width, margin) once; then per step, after the density pass,
 *   dsl_force_pass_split(h, DSL_SPLIT_BAND)   force+integrate for the particles of the grid-cell
 *                                             layers that reach within width+margin of a plane,
 *   dsl_slab_pack_band(h, ..., stream)        packs the integrated band.  stream = NULL: on the
 *                                             handle's stream, between the two launches (three
 *                                             small kernels, 25 us; the host then lets its
 *                                             transfer stream wait for that point).  A side
 *                                             stream is made to wait for the band phase only,
 *                                             but small kernels issued next to the interior
 *                                             launch are starved by it (two of its workgroups
 *                                             fill a CU's vector registers).
 *   dsl_force_pass_split(h, DSL_SPLIT_INNER)  the remaining layers, concurrently with the
 *                                             transfer the host started on `stream`.
 *   `margin` must exceed the distance a particle can move in one step (dsl_slab_status[1]
 *   reports a violation). */
enum { DSL_SPLIT_BAND = 1, DSL_SPLIT_INNER = 2 };
size_t dsl_slab_message_floats(int cap_full, int cap_xonly);
/* Once dsl_pcisph_begin has run, a full record also carries the predictor state (_pos, _vel of
 * pcisph_darwin.go:28-41, which the reference never re-synchronises): 13 floats instead of 7. */
int dsl_slab_record_floats(dsl_handle *h);
size_t dsl_slab_message_floats_for(dsl_handle *h, int cap_full, int cap_xonly);
int dsl_slab_config(dsl_handle *h, int axis, float lo, float hi);
int dsl_slab_split(dsl_handle *h, float width, float margin);
int dsl_slab_pack(dsl_handle *h, float width_full, float width, float *dev_lo, float *dev_hi, int cap_full,
                  int cap_xonly);
int dsl_slab_pack_band(dsl_handle *h, float width_full, float *dev_lo, float *dev_hi, int cap_full, int cap_xonly,
                       void *stream);
int dsl_force_pass_split(dsl_handle *h, int phase);
int dsl_slab_append(dsl_handle *h, const float *dev_message, int cap_full, int cap_xonly);
/* both neighbours' messages in one launch (either may be NULL); b lands behind a */
int dsl_slab_append2(dsl_handle *h, const float *dev_message_a, const float *dev_message_b, int cap_full,
                     int cap_xonly);
int dsl_slab_status(dsl_handle *h, int32_t status[4], int reset_high_water);
int dsl_slab_overflow(dsl_handle *h, int *high_water);
int dsl_get_count(dsl_handle *h, int *n_live, int *n_owned); /* blocking */
/* ---- multi-GPU: the exchange itself, behind the C ABI --------------------------------
 * The reference's only device host is Go (pcisph_gpu_darwin.go:249-286 drives one OpenCL queue); for
 * it to run N > 1 GPUs the halo exchange cannot live in some other language's runtime.  These entry
 * points own an RCCL communicator (bound at run time with dlopen: a single-GPU host never loads
 * RCCL) and drive the whole slab step, transfers included.
 *
 * dsl_comm_unique_id / dsl_comm_create : ncclGetUniqueId on one rank, its 128 bytes handed to every
 *                      rank by whatever channel the host owns (a Go channel, a socket, an MPI or
 *                      torch.distributed broadcast), then ncclCommInitRank; one process per GPU.
 * dsl_comm_create_all / dsl_create_multi : one process, `ndev` devices: ncclCommInitAll, resp. ndev
 *                      handles plus their communicators.  RCCL then wants ONE HOST THREAD PER DEVICE for
 *                      the step drivers below (a goroutine under runtime.LockOSThread each).
 * dsl_slab_attach    : after dsl_slab_config: neighbour ranks (-1 at a domain end), band widths,
 *                      message capacities (they may grow to twice these; identical on every rank),
 *                      overlap = 1 for the split step.  Allocates the message buffers and the
 *                      transfer stream.
 * dsl_slab_exchange  : pack both bands, ncclGroupStart / ncclSend + ncclRecv per neighbour /
 *                      ncclGroupEnd, append.  Asynchronous.
 * dsl_slab_wcsph_step / dsl_slab_pcisph_step : the slab step, nsteps times: exchange for the next
 *                      step at the end of each step (with overlap: band layers first, their pack and
 *                      the transfer on a side stream under the interior force launch); PCISPH adds
 *                      one 4-byte MAX all-reduce of the iteration error per correction iteration.
 *                      Every 8th step the band high-water marks of all ranks are MAX-reduced (one
 *                      host synchronisation) and every rank switches to the same new message
 *                      sizes; an overflow or an outrun split margin ANYWHERE makes the call fail with
 *                      DSL_ERR_OVERFLOW on every rank instead of silently losing particles.
 * dsl_slab_replan    : that re-plan on demand (blocking, collective). */
#define DSL_COMM_ID_BYTES 128
typedef struct dsl_comm dsl_comm;
int dsl_comm_unique_id(uint8_t id[DSL_COMM_ID_BYTES]);
int dsl_comm_create(int nranks, int rank, const uint8_t id[DSL_COMM_ID_BYTES], int device, dsl_comm **out);
int dsl_comm_create_all(int ndev, const int *devices, dsl_comm **out /* ndev */);
int dsl_comm_destroy(dsl_comm *c);
/* ncclCommCount: how many ranks the communicator REALLY spans (a custom transport: the count it was created
 * with).  A bench or a host that asked for N ranks checks this instead of trusting its own arithmetic. */
int dsl_comm_count(dsl_comm *c, int *nranks);
const char *dsl_comm_last_error(void);
/* A communicator over the HOST'S OWN transport instead of RCCL (MPI, a socket layer, a test shim): the slab
 * drivers make exactly the same sequence of calls through this table as they make to RCCL (one group per
 * exchange: group_start, send per neighbour, recv per neighbour, group_end; all_reduce_max for the re-plan
 * words and the PCISPH iteration error).  Buffers are DEVICE pointers; `stream` is the hipStream_t the
 * operation is ordered on: it may start only once the work queued on that stream so far has finished, and
 * work queued on the stream after the call returns must see its result (RCCL's stream semantics; a blocking
 * implementation satisfies them by synchronising the stream first).  Inside a group nothing may block on
 * its peer before group_end.  Every callback returns 0 on success; the table is copied. */
typedef struct dsl_transport {
  void *ctx;
  int (*group_start)(void *ctx);
  int (*group_end)(void *ctx);
  int (*send)(void *ctx, const void *dev_buf, size_t bytes, int peer, void *stream);
  int (*recv)(void *ctx, void *dev_buf, size_t bytes, int peer, void *stream);
  int (*all_reduce_max_u32)(void *ctx, void *dev_buf /* in place */, size_t count, void *stream);
} dsl_transport;
int dsl_comm_create_custom(int nranks, int rank, int device, const dsl_transport *transport, dsl_comm **out);
int dsl_create_multi(const dsl_params *params /* ndev */, int ndev, const int *devices, dsl_handle **handles /* ndev */,
                     dsl_comm **comms /* ndev */);
int dsl_slab_attach(dsl_handle *h, dsl_comm *comm, int lo_rank, int hi_rank, float width_full, float width, int cap_full,
                    int cap_xonly, int overlap);
int dsl_slab_detach(dsl_handle *h);
/* periodic images along the slab axis: records arriving from the lower / upper neighbour are moved by
 * from_lo / from_hi (a ring of ranks closes with -L / +L at its two ends, 0 elsewhere) */
int dsl_slab_image_shift(dsl_handle *h, float from_lo, float from_hi);
int dsl_slab_exchange(dsl_handle *h);
int dsl_slab_replan(dsl_handle *h);
int dsl_slab_wcsph_step(dsl_handle *h, int nsteps);
int dsl_slab_pcisph_step(dsl_handle *h, int nsteps);

/* Library options: behaviour that is the product's own and has no counterpart among the reference's parameters.
 *   DSL_OPT_SKIN (set/get): 0 = off, else a fraction s of h in (0, 0.2]; default 0.07 for DSL_MATH_FAST handles of at
 *       least 200,000 particles, 0 otherwise.  dsl_wcsph_step (DSL_MATH_FAST, grid
 *       neighbours, single domain, no boundary particles) then keeps one neighbour LIST per particle, built against the
 *       cut-off h (1 + s) on cells h (1 + s) wide, and walks it step after step until some particle may have moved
 *       s h / 2 since the build -- measured on the device, by the integrating kernel, against the build's reference
 *       positions (DSL_OPT_SKIN_PREDICT); only then does it sort and sweep again.  The sums are the reference's sums over { |x_i - x_j| < h }
 *       (sph_field.go:155-200,251-269): a listed pair beyond h contributes exactly 0.  The reference itself rebuilds
 *       its sampler only every 4th CacheIncr (fluid.go:208-215).  DSL_MATH_EXACT rebuilds every step.
 *       Once the flow outruns the skin (five of the last 16 steps rebuilt -- a rebuild costs about two steps), or more
 *       than one particle in 64 sits in a tile whose wider cells no longer fit the LDS image, the library suspends it by
 *       itself for the next 2048 steps (twice as long after every suspension in a row, up to 32768), then tries again; it looks every 32 steps, at step counts
 *       fixed in advance, so results never depend on timing.
 *   DSL_OPT_SKIN_STEPS / _REBUILDS / _LIST_OVERFLOW / _SUSPENSIONS (get): skin steps taken, how many of them rebuilt,
 *       whether some particle's list outgrew 95 entries (it then takes the global-memory sweep: correct, slow), how
 *       often the library suspended the skin. */
enum {
  DSL_OPT_SKIN = 1,
  DSL_OPT_SKIN_STEPS = 2,
  DSL_OPT_SKIN_REBUILDS = 3,
  DSL_OPT_SKIN_LIST_OVERFLOW = 4,
  DSL_OPT_SKIN_SUSPENSIONS = 5,
  DSL_OPT_DEVICE_BYTES = 6,       /* (get) device memory this handle has allocated so far */
  DSL_OPT_SKIN_FIELDS_OWN = 7,    /* (get, while skin steps are live) list fields the particles needed at the last rebuild ... */
  DSL_OPT_SKIN_FIELDS_PADDED = 8, /* ... and the fields their lists hold, padded to the longest list of each wave */
  DSL_OPT_SKIN_PREDICT = 9,       /* (set/get) in [0, 0.95], default 0.8: the lists are built at REFERENCE positions x + tau v --
                                     where a particle will be about half way through the lists' life -- and displacement is
                                     measured against those: the same budget s h / 2 then covers the way from -tau v to
                                     +tau v.  tau is chosen on the device at every rebuild so that the fastest particle
                                     uses this fraction of the budget at the build itself (at most 16 steps ahead); any
                                     reference is sound, the displacement test is the same.  0: built where the particles are */
  DSL_OPT_SKIN_TAU_STEPS = 10,    /* (get, while skin steps are live) tau of the last rebuild, in steps */
  /* the kernels' fall-back forms (A/B runs and tests; the defaults are the product).  Each is product code that some
   * configuration or failure path reaches, and tests/test_gpu_variants.py holds each to the default's parity bar. */
  DSL_OPT_DENSITY_PAIR = 16,      /* 1: FAST density sweep with two targets per lane (default); 0: one lane per target */
  DSL_OPT_CELL_KEYS = 17,         /* 1: in-cell ordering from per-cell key rows in the scatter pass itself; 0: two passes */
  DSL_OPT_TILE_BOX = 18,          /* tile-list enumeration boxes bx | by << 8 | bz << 16 (default 8,4,4); 0: linear order */
  DSL_OPT_PERSISTENT_BLOCKS = 19, /* cap on the persistent grids' workgroups (tests: few workgroups walk many tiles); 0: none */
  DSL_OPT_PCI_QTILED = 20,        /* binned DensityF: 1 LDS sweep over query tiles (default), 0 global-memory sweep */
  DSL_OPT_PCI_QPAIR = 21,         /* ... two queries of one cell per lane (default 1) */
  DSL_OPT_PCI_QROWS = 22,         /* ... per-cell query rows instead of a sorted array (default 1; 512 B per GRID CELL,
                                     allocated when the binned form is first used: 33 GB for the 64M scene's box) */
  DSL_OPT_LIST_BUILD = 24,        /* skin step: 1 the lists are built in lock step -- every lane of a wave produces one field per trip
                                     from a queue of its non-empty mask words (default); 0: one bit loop per mask word */
  DSL_OPT_GRID_OVERSUB = 25,      /* the tile kernels' grids are this many times the workgroups a chip holds at once (default 1:
                                     persistent workgroups, each walks its share of the tile list; k > 1: the hardware hands
                                     out k times as many, shorter shares as workgroups retire -- evens out tiles of unequal cost) */
  DSL_OPT_TILE_QUEUE = 26,        /* the two force kernels of a single domain draw their tiles from per-XCD counters as they go (a
                                     workgroup that drew cheap tiles takes more of them) instead of walking a share dealt in
                                     advance: 1 (default) from 8M particles on, 2 always, 0 never */
  DSL_OPT_PCI_QINCR = 23          /* ... the rows kept from one correction iteration of a step to the next: only a query that
                                     has changed cells is moved (default 1; 0: every iteration fills the rows afresh) */
};
int dsl_set_option(dsl_handle *h, int option, double value);
int dsl_get_option(dsl_handle *h, int option, double *value);

/* global particle ids of the current slots (host order of dsl_upload); default 0..n-1 */
int dsl_set_ids(dsl_handle *h, const int32_t *ids, size_t count);
/* Marks every force as equal to force_reset (the state Update leaves, fluid.go:193) */
int dsl_reset_forces(dsl_handle *h);

/* Error text of the last failing call on this handle (NULL handle: creation errors).
 * Replaces log.Fatalf in compute/gpu/gpu.go:52,66,234,339: the process is never aborted. */
const char *dsl_last_error(dsl_handle *h);
const char *dsl_version(void);

#ifdef __cplusplus
}
#endif
#endif
