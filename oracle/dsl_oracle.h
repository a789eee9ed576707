/*
 * dsl_oracle.h -- CPU restatement of dieselfluid's SPH particle-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline.  The product (libdslsph.so) never
 * links, loads or falls back to this code.
 *
 * PARITY STATUS: UNPINNED BY THE REFERENCE.  The reference (Go) cannot be built or
 * imported in this image (no Go toolchain; solver/pcisph is darwin-only), and its own
 * tests hold no SPH golden vectors.  This restatement follows the
 * Go source operation by operation (file:line cited at every function) and is pinned
 * only by (a) the exact vector checks of math/math_test.go:13-110 and (b) the
 * hand-derived known-answer values listed in SURVEY.md section 8c
 * (tests/test_oracle_kat.py).
 *
 * Arithmetic rules (match Go on amd64, GOAMD64=v1): every operation is a separately
 * rounded float32 operation (compile with -ffp-contract=off, no -ffast-math); sqrt and
 * pow are evaluated in double and rounded to float32 (math/vector/vector.go:301-308,
 * model/model.go:92-101).
 *
 * All citations are relative to /root/reference/.
 */
#ifndef DSL_ORACLE_H
#define DSL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- neighbour candidate rule --------------------------------------------------- */
enum {
  DSLO_NEIGH_LSH_REF = 0, /* bit-faithful sampler/lsh/lsh.go (100 samples, duplicates) */
  DSLO_NEIGH_GRID = 1,    /* all j with |xi-xj| < h, found with a uniform cell grid      */
  DSLO_NEIGH_ALL = 2      /* all j with |xi-xj| < h, brute force (validates GRID)        */
};
enum {
  DSLO_ORDER_CELL = 0,     /* candidates visited cell by cell (z,y,x), ascending in cell */
  DSLO_ORDER_ASCENDING = 1 /* candidates visited in ascending particle index             */
};
#define DSLO_SAMPLES 100   /* sampler/lsh/lsh.go:17 */
#define DSLO_MAX_HASH_BITS 32

/* kernel/std_kernel.go:7-17 */
typedef struct {
  float A, B, C, H1, H_, H2, H3, H4, H5;
} dslo_kernel;

/* model/particle.go:4-10 */
typedef struct {
  float position[3], velocity[3], force[3];
  float density, press;
} dslo_particle;

/* model/particle_array.go:5-15 */
typedef struct {
  float *positions;  /* (n_particles+n_boundary)*3, xyz interleaved */
  float *velocities; /* n*3 */
  float *densities;  /* n   */
  float *forces;     /* n*3 */
  float *pressures;  /* n   */
  int n_particles, n_boundary;
  float mass, reference_density;
} dslo_particles;

/* sampler/lsh/lsh.go:20-27 plus the build's grid candidate table */
typedef struct {
  int mode, order;
  /* lsh_ref */
  int buckets, size, hash_bits;
  float hash_vectors[DSLO_MAX_HASH_BITS][3];
  int **table; /* NULL == Go nil bucket */
  int *len, *cap;
  /* grid */
  float cell, gmin[3];
  int dims[3];
  int *cell_start; /* ncell+1 */
  int *cell_items; /* ascending particle index inside each cell */
  int ncell;
} dslo_sampler;

/* Everything that is a compile-time constant or Init argument in the reference, made
 * explicit so that the build's own dam-break scene is one point in the same parameter
 * space.  dslo_params_reference() returns the reference's values. */
typedef struct {
  int n3;            /* lattice edge; N = n3^3 (fluid.go:51) */
  int neigh_mode, neigh_order;
  float h;           /* fluid.go:48 (1.0)            */
  float mass;        /* fluid.go:56 (1.0)            */
  float ref_density; /* fluid.go:55 (N / 8)          */
  float mu;          /* fluid.go:18 (1.3059)         */
  float dt;          /* fluid.go:112 (0.01)          */
  float eos_w;       /* model.go:94 (2.15)           */
  float eos_gamma;   /* model.go:93 (7.16)           */
  float eos_d0_grad; /* field_types.go:41 (87.0)     */
  float pressure_sign;  /* +1: sph_field.go:192-198 / fluid.go:168-169 add the gradient */
  int visc_running_mass;/* 1: sph_field.go:265 multiplies the running sum by m          */
  float force_reset[3]; /* fluid.go:193 (0,-9.81*m,0) */
  float external[3];    /* wcsph.go:19 (0,-9.81,0)    */
  int wcsph_pressure_force; /* 0 in the reference loop (wcsph.go:14-26) */
  int wcsph_viscosity;      /* 0 in the reference loop                  */
  int pci_max_iters;        /* pcisph_darwin.go:49 (5)                  */
  float pci_max_error;      /* pcisph_darwin.go:50 (0.01)               */
  /* build-defined (no reference counterpart): axis-aligned wall box */
  int walls;
  float box_min[3], box_max[3];
  float restitution;
  /* grid geometry for DSLO_NEIGH_GRID */
  float grid_min[3], grid_max[3];
  /* > 0: ParticleArray.SetReferenceDensity(d0) after construction (particle_array.go:35-37);
   * 0: keep NewParticleArray's ReferenceDensity = ref_density * mass (particle_array.go:26) */
  float d0_override;
  /* build-defined terms of BASELINE configs[4] (no reference counterpart), 0 = off:
   *   xsph_eps : positions advect with v + eps * sum_j (m/rho_j) (v_j - v_i) F(r_ij)
   *   st_kappa : cohesion force F_i += kappa * sum_j m (x_j - x_i) F(r_ij)            */
  float xsph_eps, st_kappa;
} dslo_params;

/* model/sph/fluid.go:23-33 (+ solver state of pcisph_darwin.go:28-41) */
typedef struct {
  dslo_params prm;
  dslo_kernel kern;
  dslo_particles parts;
  dslo_sampler smp;
  float time, max_vel, max_f, cache_life, mu, delta;
  int particles;
  float *xsph; /* n*3: XSPH velocity correction of the current step (build-defined) */
  /* PCISPH predictor state; allocated by dslo_pcisph_begin */
  float *pci_pos, *pci_vel;
  float pci_last_error;
  int pci_last_iters;
  /* scratch */
  int *scratch;
  int scratch_cap;
} dslo_sph;

/* ---- K: kernel/std_kernel.go --------------------------------------------------- */
dslo_kernel dslo_build_kernel(float h);
float dslo_kernel_F(const dslo_kernel *k, float x);
float dslo_kernel_O1D(const dslo_kernel *k, float x);
float dslo_kernel_O2D(const dslo_kernel *k, float x);
void dslo_kernel_grad(const dslo_kernel *k, float x, const float dir[3], float out[3]);

/* ---- E: model/model.go:92-101 --------------------------------------------------- */
float dslo_tait_eos_ex(float x, float d0, float p0, float w, float g);
float dslo_tait_eos(float x, float d0, float p0);

/* ---- math/vector --------------------------------------------------------------- */
float dslo_vec_mag(const float *v, int n);
float dslo_vec_dist3(const float a[3], const float b[3]);
float dslo_vec_dot3(const float a[3], const float b[3]);
void dslo_vec_norm3(const float a[3], float out[3]);
void dslo_vec_cross3(const float a[3], const float b[3], float out[3]);
int dslo_vec_add(const float *a, int na, const float *b, int nb, float *out);   /* returns the result's length */
int dslo_vec_scale(const float *a, int na, float k, float *out);
int dslo_vec_sub(const float *b, int nb, const float *a, int na, float *out);
void dslo_vec_proj3(const float a[3], const float n[3], float out[3]);
void dslo_vec_refl3(const float v[3], const float n[3], float out[3]);

/* ---- P: model/particle_array.go -------------------------------------------------- */
int dslo_particles_init(dslo_particles *p, int n, int nb, float density, float mass);
void dslo_particles_free(dslo_particles *p);
dslo_particle dslo_particles_get(const dslo_particles *p, int index);
void dslo_particles_set(dslo_particles *p, int index, const dslo_particle *q);

/* ---- N: sampler/lsh/lsh.go -------------------------------------------------------- */
int dslo_lsh_size(int num_particles, int buckets);
int dslo_lsh_hash(const dslo_sampler *s, const float pos[3]);
void dslo_sampler_update(dslo_sph *s);
int dslo_get_samples(dslo_sph *s, int i, const int **out);
int dslo_get_samples_from_position(dslo_sph *s, const float pos[3], const int **out);
void dslo_lsh_get_data_1d(const dslo_sph *s, int *out /* buckets*size */);

/* ---- I / S0: geom/grid/point-grid.go, model/sph/fluid.go:41-88 ------------------- */
void dslo_lattice_positions(int n3, const float origin[3], int origin_len, float *pos);
dslo_params dslo_params_reference(int n3);
/* hash_vectors: hash_bits*3 floats (an explicit input: Go's math/rand stream is not
 * reproducible here, sampler/lsh/lsh.go:32-40).  Runs the whole of sph.Init. */
dslo_sph *dslo_sph_init(const dslo_params *prm, const float origin[3], int origin_len,
                        const float *hash_vectors, int hash_bits, int pci);
/* Same system object, but state supplied by the caller instead of the lattice + the
 * Init-time passes (used for the seeded/jittered parity inputs and the dam-break). */
dslo_sph *dslo_sph_from_state(const dslo_params *prm, int n, const float *pos,
                              const float *vel, const float *force,
                              const float *hash_vectors, int hash_bits);
void dslo_sph_free(dslo_sph *s);
/* boundary particles: particle_array.go:123-128, sph_field.go:75-85, geom/mesh/mesh.go:60-76 */
int dslo_sph_add_boundary(dslo_sph *s, const float *positions, int nb);
void dslo_mesh_boundary_particles(const float *vertices, int nverts, float *out);

/* ---- passes: model/sph/fluid.go:111-277 ---------------------------------------- */
float dslo_cfl(dslo_sph *s);
void dslo_density_all(dslo_sph *s);
void dslo_pressure_all(dslo_sph *s);
void dslo_viscous_all(dslo_sph *s);
void dslo_external_all(dslo_sph *s, const float f[3]);
void dslo_gradient_pressure_force(dslo_sph *s);
void dslo_update(dslo_sph *s);
float dslo_cache_incr(dslo_sph *s, int *rebuilt);
float dslo_pcidelta(dslo_sph *s);
float dslo_density_f(dslo_sph *s, const float pos[3]);
void dslo_surface_tension_all(dslo_sph *s); /* build-defined */
void dslo_xsph_all(dslo_sph *s);            /* build-defined */

/* ---- unused-by-solvers field operators: model/field/sph_field.go:124-135,203-294 ---- */
float dslo_field_div(dslo_sph *s, int i, int tensor_field);
void dslo_field_curl(dslo_sph *s, int i, int tensor_field, float out[3]);
float dslo_field_laplacian(dslo_sph *s, int i, int scalar_field);
float dslo_field_interpolate(dslo_sph *s, const float pos[3], int scalar_field);

/* ---- drivers: solver/wcsph/wcsph.go:14-26, solver/pcisph/pcisph_darwin.go:24-118 -- */
void dslo_wcsph_step(dslo_sph *s);
void dslo_pcisph_begin(dslo_sph *s);
void dslo_pcisph_step(dslo_sph *s);

/* ---- build-defined synthetic dam-break (no reference counterpart) ----------------- */
uint64_t dslo_xorshift64s(uint64_t *state);
uint64_t dslo_splitmix64(uint64_t seed, uint64_t counter);
void dslo_dambreak_positions(int n3, float dx, float jitter, uint64_t seed, float *pos);

#ifdef __cplusplus
}
#endif
#endif
