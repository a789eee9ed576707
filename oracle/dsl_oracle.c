/*
 * dsl_oracle.c -- CPU restatement of dieselfluid's SPH particle-step hot path.
 * TEST INFRASTRUCTURE ONLY; see dsl_oracle.h for the rules and the parity status
 * ("parity unpinned by the reference").  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fno-fast-math).
 *
 * Every function cites the reference file:line (relative to /root/reference/) whose
 * arithmetic it restates, in the same operation order, one float32 rounding per Go
 * operation.
 */
#include "dsl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* =====================================================================================
 * math/vector/vector.go
 * ===================================================================================== */

/* vector.go:301-308  Mag: float32 running sum of squares, float64 sqrt, round. */
float dslo_vec_mag(const float *v, int n) {
  float size = 0.0f;
  for (int i = 0; i < n; i++) size += v[i] * v[i];
  return (float)sqrt((double)size);
}

/* vector.go:199-207 Sub(b,a) = Add(b, Scale(a,-1)); vector.go:439-441 Dist = Mag(Sub(a,b)) */
static void vsub3(const float b[3], const float a[3], float out[3]) {
  for (int i = 0; i < 3; i++) {
    float na = a[i] * -1.0f;
    out[i] = b[i] + na;
  }
}
float dslo_vec_dist3(const float a[3], const float b[3]) {
  float d[3];
  vsub3(a, b, d);
  return dslo_vec_mag(d, 3);
}

/* vector.go:268-276 Dot: a0*b0 + a1*b1 + a2*b2, left to right. */
float dslo_vec_dot3(const float a[3], const float b[3]) {
  float t0 = a[0] * b[0];
  float t1 = a[1] * b[1];
  float t2 = a[2] * b[2];
  float s = t0 + t1;
  return s + t2;
}

/* vector.go:322-331 Norm: divide by Mag unless it is zero (then the zero vector). */
void dslo_vec_norm3(const float a[3], float out[3]) {
  float l = dslo_vec_mag(a, 3);
  out[0] = out[1] = out[2] = 0.0f;
  if (l != 0.0f)
    for (int i = 0; i < 3; i++) out[i] = a[i] / l;
}

/* vector.go:282-296 Cross (pinned by math_test.go:75). */
void dslo_vec_cross3(const float a[3], const float b[3], float out[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - b[2] * a[0];
  out[2] = a[0] * b[1] - b[0] * a[1];
}

/* vector.go:167-179 Add: the zero 3-vector when either operand is empty or the lengths differ (the
 * rule that collapses sph.Init's lattice for a zero-length origin); n_out = length of
 * the result. */
int dslo_vec_add(const float *a, int na, const float *b, int nb, float *out) {
  if (na == 0 || na != nb) {
    out[0] = out[1] = out[2] = 0.0f;
    return 3;
  }
  for (int i = 0; i < nb; i++) out[i] = a[i] + b[i];
  return nb;
}
/* vector.go:232-241 Scale */
int dslo_vec_scale(const float *a, int na, float k, float *out) {
  if (na == 0) {
    out[0] = out[1] = out[2] = 0.0f;
    return 3;
  }
  for (int i = 0; i < na; i++) out[i] = a[i] * k;
  return na;
}
/* vector.go:199-207 Sub(b, a) = Add(b, Scale(a, -1)) */
int dslo_vec_sub(const float *b, int nb, const float *a, int na, float *out) {
  if (na == 0 || na != nb) {
    out[0] = out[1] = out[2] = 0.0f;
    return 3;
  }
  float t[16];
  int n = dslo_vec_scale(a, na > 16 ? 16 : na, -1.0f, t);
  return dslo_vec_add(b, nb > 16 ? 16 : nb, t, n, out);
}
/* vector.go:349-353 Proj(a, n) = Scale(Norm(n), Dot(a, n) / Mag(n)) (pinned by math_test.go:90-98) */
void dslo_vec_proj3(const float a[3], const float n[3], float out[3]) {
  float vn[3];
  dslo_vec_norm3(n, vn);
  float k = dslo_vec_dot3(a, n) / dslo_vec_mag(n, 3);
  dslo_vec_scale(vn, 3, k, out);
}
/* vector.go:382-385 Refl(v, n) = Sub(v, Scale(n, Dot(v, n) * 2)) (pinned by math_test.go:100-106) */
void dslo_vec_refl3(const float v[3], const float n[3], float out[3]) {
  float b[3];
  dslo_vec_scale(n, 3, dslo_vec_dot3(v, n) * 2.0f, b);
  dslo_vec_sub(v, 3, b, 3, out);
}

/* =====================================================================================
 * K: kernel/std_kernel.go
 * ===================================================================================== */

/* std_kernel.go:5,20-31.  PI is the untyped constant 3.141592653589; constant
 * sub-expressions (64*PI) fold exactly and are then converted to float32. */
dslo_kernel dslo_build_kernel(float h) {
  static const double PI = 3.141592653589;
  dslo_kernel k;
  k.H1 = h;
  k.H_ = h;
  k.H2 = h * h;
  k.H3 = h * h * h;
  k.H4 = h * h * h * h;
  k.H5 = h * h * h * h * h;
  float c64pi = (float)(64.0 * PI);
  float cpi = (float)PI;
  k.A = 315.0f / (c64pi * k.H3);
  k.B = -45.0f / (cpi * k.H4);
  k.C = 90.0f / (cpi * k.H5);
  return k;
}

/* std_kernel.go:33-39 */
float dslo_kernel_F(const dslo_kernel *k, float x) {
  if (x >= k->H_) return 0.0f;
  float xx = x * x;
  float hh = k->H_ * k->H_;
  float q = 1.0f - xx / hh;
  float aq = k->A * q;
  return aq * q;
}

/* std_kernel.go:54-60 */
float dslo_kernel_O1D(const dslo_kernel *k, float x) {
  if (x >= k->H_) return 0.0f;
  float q = 1.0f - x / k->H_;
  float bq = k->B * q;
  return bq * q;
}

/* std_kernel.go:63-71 (note '>' not '>=') */
float dslo_kernel_O2D(const dslo_kernel *k, float x) {
  if (x > k->H_) return 0.0f;
  float q = 1.0f - x / k->H_;
  return k->C * q;
}

/* std_kernel.go:74-76  Grad = Scale(dir, -O1D(x)) */
void dslo_kernel_grad(const dslo_kernel *k, float x, const float dir[3], float out[3]) {
  float s = -dslo_kernel_O1D(k, x);
  for (int i = 0; i < 3; i++) out[i] = dir[i] * s;
}

/* =====================================================================================
 * E: model/model.go:92-101
 * ===================================================================================== */
float dslo_tait_eos_ex(float x, float d0, float p0, float w, float g) {
  if (x <= d0) x = d0;
  float ratio = x / d0;
  float wg = w / g;
  float pw = (float)(pow((double)ratio, (double)g) - 1.0);
  float y = wg * pw;
  return y + p0;
}
float dslo_tait_eos(float x, float d0, float p0) {
  return dslo_tait_eos_ex(x, d0, p0, 2.15f, 7.16f);
}

/* =====================================================================================
 * P: model/particle_array.go
 * ===================================================================================== */

/* particle_array.go:18-33 */
int dslo_particles_init(dslo_particles *p, int n, int nb, float density, float mass) {
  memset(p, 0, sizeof(*p));
  p->positions = (float *)calloc((size_t)(n + nb) * 3 + 1, sizeof(float));
  p->velocities = (float *)calloc((size_t)n * 3 + 1, sizeof(float));
  p->densities = (float *)calloc((size_t)n + 1, sizeof(float));
  p->forces = (float *)calloc((size_t)n * 3 + 1, sizeof(float));
  p->pressures = (float *)calloc((size_t)n + 1, sizeof(float));
  if (!p->positions || !p->velocities || !p->densities || !p->forces || !p->pressures) return -1;
  p->mass = mass;
  p->reference_density = density * mass;
  p->n_particles = n;
  p->n_boundary = nb;
  return 0;
}
void dslo_particles_free(dslo_particles *p) {
  free(p->positions);
  free(p->velocities);
  free(p->densities);
  free(p->forces);
  free(p->pressures);
  memset(p, 0, sizeof(*p));
}
static int parts_total(const dslo_particles *p) { return p->n_particles + p->n_boundary; }

/* model.go:117-130 Float3_set: silently does nothing when x+2 > len(buffer). */
static void float3_set(int x, float a[3], const float *buffer, long len) {
  if ((long)x + 2 > len) return;
  if (len < 3) return;
  a[0] = buffer[x];
  a[1] = buffer[x + 1];
  a[2] = buffer[x + 2];
}
/* model.go:103-115 */
static void float3_buffer_set(int x, float *buffer, long len, const float b[3]) {
  if ((long)x + 2 > len) return;
  if (len < 3) return;
  buffer[x] = b[0];
  buffer[x + 1] = b[1];
  buffer[x + 2] = b[2];
}

/* particle_array.go:94-117.  index == n_particles (and anything >= Total) yields the
 * zero particle; n_particles < index < Total is a boundary particle (position only). */
dslo_particle dslo_particles_get(const dslo_particles *p, int index) {
  dslo_particle q;
  memset(&q, 0, sizeof(q));
  int x = index * 3;
  long lp = (long)parts_total(p) * 3, ln = (long)p->n_particles * 3;
  if (index > p->n_particles && index < parts_total(p)) {
    float3_set(x, q.position, p->positions, lp);
    return q;
  }
  if (index >= 0 && index < p->n_particles) {
    float3_set(x, q.position, p->positions, lp);
    float3_set(x, q.velocity, p->velocities, ln);
    float3_set(x, q.force, p->forces, ln);
    q.density = p->densities[index];
    q.press = p->pressures[index];
    return q;
  }
  return q;
}

/* particle_array.go:86-93 */
void dslo_particles_set(dslo_particles *p, int index, const dslo_particle *q) {
  int x = index * 3;
  long lp = (long)parts_total(p) * 3, ln = (long)p->n_particles * 3;
  float3_buffer_set(x, p->positions, lp, q->position);
  float3_buffer_set(x, p->velocities, ln, q->velocity);
  float3_buffer_set(x, p->forces, ln, q->force);
  p->densities[index] = q->density;
  p->pressures[index] = q->press;
}

/* =====================================================================================
 * N: sampler/lsh/lsh.go  (+ the build's grid / brute-force candidate rules)
 * ===================================================================================== */

/* lsh.go:33  factor := int(float32(num_particles/buckets) * LOAD_FACTOR) */
int dslo_lsh_size(int num_particles, int buckets) {
  float f = (float)(num_particles / buckets);
  return (int)(f * 1.5f);
}

/* lsh.go:51-56 sgn, lsh.go:102-111 Hash */
int dslo_lsh_hash(const dslo_sampler *s, const float pos[3]) {
  long hash = 0;
  for (int i = 0; i < s->hash_bits; i++) {
    hash = hash << 1;
    float d = dslo_vec_dot3(pos, s->hash_vectors[i]);
    hash += (d <= 0.0f) ? 0 : 1;
  }
  return (int)(hash % s->buckets);
}

static void lsh_reset(dslo_sampler *s) { /* lsh.go:120-124 */
  for (int i = 0; i < s->buckets; i++) {
    free(s->table[i]);
    s->table[i] = NULL;
    s->len[i] = s->cap[i] = 0;
  }
}
static void lsh_insert(dslo_sampler *s, int hash, int particle) { /* lsh.go:113-118 */
  if (s->len[hash] == s->cap[hash]) {
    s->cap[hash] = s->cap[hash] ? s->cap[hash] * 2 : 16;
    s->table[hash] = (int *)realloc(s->table[hash], sizeof(int) * (size_t)s->cap[hash]);
  }
  s->table[hash][s->len[hash]++] = particle;
}

/* Cell coordinate rule shared by the grid candidate table (build-defined). */
static int cell_coord(float p, float gmin, float inv_cell, int dim) {
  float f = floorf((p - gmin) * inv_cell);
  if (!(f >= 0.0f)) return 0; /* also NaN */
  if (f >= (float)dim) return dim - 1;
  return (int)f;
}
static void grid_setup(dslo_sph *s) {
  dslo_sampler *g = &s->smp;
  g->cell = s->prm.h;
  float inv = 1.0f / g->cell;
  g->ncell = 1;
  for (int a = 0; a < 3; a++) {
    g->gmin[a] = s->prm.grid_min[a];
    int d = (int)ceilf((s->prm.grid_max[a] - s->prm.grid_min[a]) * inv);
    if (d < 1) d = 1;
    g->dims[a] = d;
    g->ncell *= d;
  }
  g->cell_start = (int *)calloc((size_t)g->ncell + 1, sizeof(int));
  g->cell_items = (int *)calloc((size_t)parts_total(&s->parts) + 1, sizeof(int));
}
static int grid_cell_of(const dslo_sampler *g, const float p[3]) {
  float inv = 1.0f / g->cell;
  int cx = cell_coord(p[0], g->gmin[0], inv, g->dims[0]);
  int cy = cell_coord(p[1], g->gmin[1], inv, g->dims[1]);
  int cz = cell_coord(p[2], g->gmin[2], inv, g->dims[2]);
  return (cz * g->dims[1] + cy) * g->dims[0] + cx;
}
static void grid_update(dslo_sph *s) {
  dslo_sampler *g = &s->smp;
  int total = parts_total(&s->parts);
  memset(g->cell_start, 0, sizeof(int) * ((size_t)g->ncell + 1));
  /* binned where Get(i) says the particle is, as HashSampler.UpdateSampler hashes Get(i).Position
   * (lsh.go:126-133): index == n_particles, the first boundary particle, reads as the origin
   * (particle_array.go:94-117) */
  for (int i = 0; i < total; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    g->cell_start[grid_cell_of(g, q.position) + 1]++;
  }
  for (int c = 0; c < g->ncell; c++) g->cell_start[c + 1] += g->cell_start[c];
  int *fill = (int *)malloc(sizeof(int) * ((size_t)g->ncell + 1));
  memcpy(fill, g->cell_start, sizeof(int) * ((size_t)g->ncell + 1));
  for (int i = 0; i < total; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    g->cell_items[fill[grid_cell_of(g, q.position)]++] = i;
  }
  free(fill);
}

/* lsh.go:126-133 UpdateSampler */
void dslo_sampler_update(dslo_sph *s) {
  if (s->smp.mode == DSLO_NEIGH_LSH_REF) {
    lsh_reset(&s->smp);
    int total = parts_total(&s->parts);
    for (int i = 0; i < total; i++) {
      dslo_particle q = dslo_particles_get(&s->parts, i);
      lsh_insert(&s->smp, dslo_lsh_hash(&s->smp, q.position), i);
    }
  } else if (s->smp.mode == DSLO_NEIGH_GRID) {
    grid_update(s);
  }
}

static void scratch_reserve(dslo_sph *s, int n) {
  if (n > s->scratch_cap) {
    s->scratch_cap = n * 2 + 128;
    s->scratch = (int *)realloc(s->scratch, sizeof(int) * (size_t)s->scratch_cap);
  }
}
static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* lsh.go:136-158 / 160-181: start at the query's bucket, skip nil buckets cyclically,
 * copy entries until 100 are collected, re-reading the same bucket from its start when
 * it holds fewer than 100 (duplicates).  GRID/ALL: candidate list (distance test is
 * applied by the passes). */
static int samples_for(dslo_sph *s, const float pos[3], const int **out) {
  dslo_sampler *g = &s->smp;
  if (g->mode == DSLO_NEIGH_LSH_REF) {
    scratch_reserve(s, DSLO_SAMPLES);
    int num = 0;
    int index = dslo_lsh_hash(g, pos);
    int empty_run = 0;
    while (num < DSLO_SAMPLES) {
      if (index > g->buckets) index = 0; /* lsh.go:143 (never true) */
      if (g->table[index] == NULL) {
        index++;
        index = index % g->buckets;
        if (++empty_run > g->buckets) break; /* all nil: the Go loop would spin forever */
      } else {
        empty_run = 0;
        for (int j = 0; j < g->len[index] && num < DSLO_SAMPLES; j++) s->scratch[num++] = g->table[index][j];
      }
    }
    *out = s->scratch;
    return num;
  }
  if (g->mode == DSLO_NEIGH_ALL) {
    int total = parts_total(&s->parts);
    scratch_reserve(s, total);
    for (int i = 0; i < total; i++) s->scratch[i] = i;
    *out = s->scratch;
    return total;
  }
  /* GRID */
  float inv = 1.0f / g->cell;
  int cx = cell_coord(pos[0], g->gmin[0], inv, g->dims[0]);
  int cy = cell_coord(pos[1], g->gmin[1], inv, g->dims[1]);
  int cz = cell_coord(pos[2], g->gmin[2], inv, g->dims[2]);
  int num = 0;
  for (int dz = -1; dz <= 1; dz++) {
    int z = cz + dz;
    if (z < 0 || z >= g->dims[2]) continue;
    for (int dy = -1; dy <= 1; dy++) {
      int y = cy + dy;
      if (y < 0 || y >= g->dims[1]) continue;
      for (int dx = -1; dx <= 1; dx++) {
        int x = cx + dx;
        if (x < 0 || x >= g->dims[0]) continue;
        int c = (z * g->dims[1] + y) * g->dims[0] + x;
        int b = g->cell_start[c], e = g->cell_start[c + 1];
        scratch_reserve(s, num + (e - b));
        for (int k = b; k < e; k++) s->scratch[num++] = g->cell_items[k];
      }
    }
  }
  if (g->order == DSLO_ORDER_ASCENDING) qsort(s->scratch, (size_t)num, sizeof(int), cmp_int);
  *out = s->scratch;
  return num;
}
int dslo_get_samples(dslo_sph *s, int i, const int **out) { /* lsh.go:136-158 */
  dslo_particle q = dslo_particles_get(&s->parts, i);
  return samples_for(s, q.position, out);
}
int dslo_get_samples_from_position(dslo_sph *s, const float pos[3], const int **out) { /* lsh.go:160-181 */
  return samples_for(s, pos, out);
}

/* lsh.go:70-80 GetData1D */
void dslo_lsh_get_data_1d(const dslo_sph *s, int *out) {
  const dslo_sampler *g = &s->smp;
  memset(out, 0, sizeof(int) * (size_t)g->buckets * (size_t)g->size);
  for (int i = 0; i < g->buckets; i++)
    for (int j = 0; j < g->size; j++)
      if (g->table[i] != NULL && j < g->len[i]) out[i * g->size + j] = g->table[i][j];
}

/* In lsh_ref mode the reference relies on the kernel cut-offs alone; the geometric
 * modes define the neighbour set as { j : dist < h } (build-defined rule). */
static int in_support(const dslo_sph *s, float dist) {
  return s->smp.mode == DSLO_NEIGH_LSH_REF || dist < s->kern.H_;
}

/* =====================================================================================
 * I: geom/grid/point-grid.go:20-63, model/field/sph_field.go:87-108
 * ===================================================================================== */
void dslo_lattice_positions(int n3, const float origin[3], int origin_len, float *pos) {
  /* BuildGrid: min_bounds = (-1,-1,-1)*scl; V.Add(min_bounds, origin) returns (0,0,0)
   * on a length mismatch (vector.go:171-173). */
  float minb[3], step[3], dim = (float)n3;
  for (int a = 0; a < 3; a++) {
    float m = -1.0f * 1.0f;
    minb[a] = (origin_len == 3) ? m + origin[a] : 0.0f;
  }
  /* BuildKernGrid: step = min_bounds.Scale(-2.0).Mul(inv), inv = 1/dim (point-grid.go:39-40) */
  float inv = 1.0f / dim;
  for (int a = 0; a < 3; a++) {
    float t = minb[a] * -2.0f;
    step[a] = inv * t;
  }
  /* AlignWithGrid: id = k + n3*(i*n3 + j); pos = min + step*(i,j,k) */
  for (int i = 0; i < n3; i++)
    for (int j = 0; j < n3; j++)
      for (int k = 0; k < n3; k++) {
        int id = k + n3 * (i * n3 + j);
        float ijk[3] = {(float)i, (float)j, (float)k};
        for (int a = 0; a < 3; a++) {
          float sv = step[a] * ijk[a];
          pos[3 * id + a] = minb[a] + sv;
        }
      }
}

/* =====================================================================================
 * S0: model/sph/fluid.go:41-88
 * ===================================================================================== */
dslo_params dslo_params_reference(int n3) {
  dslo_params p;
  memset(&p, 0, sizeof(p));
  p.n3 = n3;
  p.neigh_mode = DSLO_NEIGH_LSH_REF;
  p.neigh_order = DSLO_ORDER_CELL;
  p.h = 1.0f;                                  /* fluid.go:48 */
  p.mass = 1.0f;                               /* fluid.go:56 */
  float num = (float)(n3 * n3 * n3);
  float vol = 2.0f * 1.0f * 2.0f * 1.0f * 2.0f * 1.0f; /* point-grid.go:44-46 */
  p.ref_density = num / vol;                   /* fluid.go:55 */
  p.mu = 1.3059f;                              /* fluid.go:18,69 */
  p.dt = 0.01f;                                /* fluid.go:112 */
  p.eos_w = 2.15f;                             /* model.go:94 */
  p.eos_gamma = 7.16f;                         /* model.go:93 */
  p.eos_d0_grad = 87.0f;                       /* model.go:41, field_types.go:41 */
  p.pressure_sign = 1.0f;
  p.visc_running_mass = 1;
  p.force_reset[1] = -9.81f * p.mass;          /* fluid.go:193 */
  p.external[1] = -9.81f;                      /* wcsph.go:19 */
  p.pci_max_iters = 5;                         /* pcisph_darwin.go:49 */
  p.pci_max_error = 0.01f;                     /* pcisph_darwin.go:50 */
  for (int a = 0; a < 3; a++) {
    p.grid_min[a] = -4.0f;
    p.grid_max[a] = 4.0f;
    p.box_min[a] = -1.0f;
    p.box_max[a] = 1.0f;
  }
  p.restitution = 0.0f;
  return p;
}

static dslo_sph *sph_alloc(const dslo_params *prm, int n, const float *hash_vectors, int hash_bits) {
  dslo_sph *s = (dslo_sph *)calloc(1, sizeof(dslo_sph));
  s->prm = *prm;
  s->kern = dslo_build_kernel(prm->h);                 /* fluid.go:53 */
  /* fluid.go:63: NewParticleArray(num, 0, h, ref_density, mass) */
  dslo_particles_init(&s->parts, n, 0, prm->ref_density, prm->mass);
  if (prm->d0_override > 0.0f) s->parts.reference_density = prm->d0_override; /* SetReferenceDensity */
  s->particles = n;
  s->cache_life = 0.8f;                                /* fluid.go:19,67 */
  s->mu = prm->mu;                                     /* fluid.go:69 */
  s->smp.mode = prm->neigh_mode;
  s->smp.order = prm->neigh_order;
  if (s->smp.mode == DSLO_NEIGH_LSH_REF) {             /* lsh.Allocate(num, 255, 8, ..) fluid.go:64 */
    s->smp.buckets = 255;
    s->smp.hash_bits = hash_bits > 0 ? hash_bits : 8;
    s->smp.size = dslo_lsh_size(n, 255);
    s->smp.table = (int **)calloc(255, sizeof(int *));
    s->smp.len = (int *)calloc(255, sizeof(int));
    s->smp.cap = (int *)calloc(255, sizeof(int));
    for (int i = 0; i < s->smp.hash_bits && i < DSLO_MAX_HASH_BITS; i++)
      for (int a = 0; a < 3; a++) s->smp.hash_vectors[i][a] = hash_vectors ? hash_vectors[3 * i + a] : 0.0f;
  } else if (s->smp.mode == DSLO_NEIGH_GRID) {
    grid_setup(s);
  }
  return s;
}

dslo_sph *dslo_sph_init(const dslo_params *prm, const float origin[3], int origin_len,
                        const float *hash_vectors, int hash_bits, int pci) {
  int n3 = prm->n3;
  int num = n3 * n3 * n3;                              /* fluid.go:51 */
  dslo_sph *s = sph_alloc(prm, num, hash_vectors, hash_bits);
  dslo_lattice_positions(n3, origin, origin_len, s->parts.positions); /* fluid.go:71 */
  dslo_sampler_update(s);                              /* fluid.go:72 */
  dslo_density_all(s);                                 /* fluid.go:73 */
  float g[3] = {0.0f, -9.81f * s->parts.mass, 0.0f};
  dslo_external_all(s, g);                             /* fluid.go:74 */
  dslo_viscous_all(s);                                 /* fluid.go:75 */
  dslo_cfl(s);                                         /* fluid.go:76 */
  if (pci) {                                           /* fluid.go:78-82 */
    if (dslo_pcidelta(s) == 0.0f) s->delta = prm->h;
  }
  return s;
}

dslo_sph *dslo_sph_from_state(const dslo_params *prm, int n, const float *pos, const float *vel,
                              const float *force, const float *hash_vectors, int hash_bits) {
  dslo_sph *s = sph_alloc(prm, n, hash_vectors, hash_bits);
  memcpy(s->parts.positions, pos, sizeof(float) * 3 * (size_t)n);
  if (vel) memcpy(s->parts.velocities, vel, sizeof(float) * 3 * (size_t)n);
  if (force) memcpy(s->parts.forces, force, sizeof(float) * 3 * (size_t)n);
  s->time = prm->dt;
  dslo_sampler_update(s);
  return s;
}

/* particle_array.go:123-128 AddBoundaryParticles: appends position-only particles behind the fluid
 * (n_boundary grows; velocities, densities, forces, pressures keep n_particles entries), then the
 * sampler is rebuilt over Total() particles (lsh.go:126-133).  Returns the new Total(). */
int dslo_sph_add_boundary(dslo_sph *s, const float *positions, int nb) {
  dslo_particles *p = &s->parts;
  int total = parts_total(p);
  float *np = (float *)realloc(p->positions, sizeof(float) * ((size_t)(total + nb) * 3 + 1));
  if (!np) return -1;
  p->positions = np;
  memcpy(p->positions + (size_t)total * 3, positions, sizeof(float) * (size_t)nb * 3);
  p->n_boundary += nb;
  if (s->smp.cell_items) {
    int *ci = (int *)realloc(s->smp.cell_items, sizeof(int) * ((size_t)parts_total(p) + 1));
    if (!ci) return -1;
    s->smp.cell_items = ci;
  }
  dslo_sampler_update(s);
  return parts_total(p);
}

/* geom/mesh/mesh.go:60-76 Mesh.GenerateBoundaryParticles(density): one particle per vertex (the
 * density argument is unused); `if x < len(particle_list)-3` leaves the LAST vertex's particle at the
 * origin.  out: nverts*3 floats.  model/field/sph_field.go:75-85 BoundaryParticles feeds every
 * collider's list to AddBoundaryParticles. */
void dslo_mesh_boundary_particles(const float *vertices, int nverts, float *out) {
  int len = nverts * 3;
  for (int i = 0; i < len; i++) out[i] = 0.0f;
  for (int index = 0; index < nverts; index++) {
    int x = index * 3;
    if (x < len - 3) {
      out[x] = vertices[x];
      out[x + 1] = vertices[x + 1];
      out[x + 2] = vertices[x + 2];
    }
  }
}

void dslo_sph_free(dslo_sph *s) {
  if (!s) return;
  if (s->smp.table) {
    for (int i = 0; i < s->smp.buckets; i++) free(s->smp.table[i]);
    free(s->smp.table);
    free(s->smp.len);
    free(s->smp.cap);
  }
  free(s->smp.cell_start);
  free(s->smp.cell_items);
  dslo_particles_free(&s->parts);
  free(s->pci_pos);
  free(s->pci_vel);
  free(s->xsph);
  free(s->scratch);
  free(s);
}

/* =====================================================================================
 * field operators: model/field/sph_field.go
 * ===================================================================================== */

/* sph_field.go:155-172 Density(i) */
static void field_density(dslo_sph *s, int i) {
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float density = 0.0f;
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  float mass = s->parts.mass;
  int total = parts_total(&s->parts);
  for (int j = 0; j < len; j++) {
    int pj = samples[j];
    if (i != pj && pj < total) {
      dslo_particle q = dslo_particles_get(&s->parts, pj);
      float dist = dslo_vec_dist3(pi.position, q.position);
      if (!in_support(s, dist)) continue;
      float w = dslo_kernel_F(&s->kern, dist);
      density += mass * w;
    }
  }
  pi.density = density;
  dslo_particles_set(&s->parts, i, &pi);
}

/* sph_field.go:137-152 DensityF(pos, _): starts at W0 (no mass factor), includes self,
 * neighbours' CURRENT positions, second argument ignored. */
float dslo_density_f(dslo_sph *s, const float pos[3]) {
  const int *samples;
  int len = dslo_get_samples_from_position(s, pos, &samples);
  float density = dslo_kernel_F(&s->kern, 0.0f); /* W0, std_kernel.go:41-43 */
  float mass = s->parts.mass;
  int total = parts_total(&s->parts);
  for (int j = 0; j < len; j++) {
    int pj = samples[j];
    if (pj < total) {
      dslo_particle q = dslo_particles_get(&s->parts, pj);
      float dist = dslo_vec_dist3(pos, q.position);
      if (!in_support(s, dist)) continue;
      float w = dslo_kernel_F(&s->kern, dist);
      density += mass * w;
    }
  }
  return density;
}

/* field_types.go:39-42 PressureField.Value = TaitEos(density, FLUID_DENSITY=87.0, 0) */
static float pressure_field_value(const dslo_sph *s, int i) {
  dslo_particle q = dslo_particles_get(&s->parts, i);
  return dslo_tait_eos_ex(q.density, s->prm.eos_d0_grad, 0.0f, s->prm.eos_w, s->prm.eos_gamma);
}

/* sph_field.go:175-200 Gradient(i, pressure field) */
static void field_gradient(dslo_sph *s, int i, float out[3]) {
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float F = 0.0f;
  float mass = s->parts.mass;
  float acc[3] = {0.0f, 0.0f, 0.0f};
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  float dens = pi.density;
  for (int j = 0; j < len; j++) {
    int jIndex = samples[j];
    if (jIndex != i) {
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float jDensity = q.density;
      float dir[3], nd[3], grad[3];
      vsub3(q.position, pi.position, dir);
      float dist = dslo_vec_mag(dir, 3);
      if (!in_support(s, dist)) continue;
      dslo_vec_norm3(dir, nd);
      dslo_kernel_grad(&s->kern, dist, nd, grad);
      float pi_term = pressure_field_value(s, i) / (dens * dens);
      float pj_term = pressure_field_value(s, jIndex) / (jDensity * jDensity);
      F = pi_term + pj_term;
      for (int a = 0; a < 3; a++) {
        float gs = grad[a] * F;
        acc[a] = acc[a] + gs;
      }
    }
  }
  float dm = dens * mass;
  for (int a = 0; a < 3; a++) out[a] = acc[a] * dm;
}

/* sph_field.go:251-269 LaplacianForce(i, velocity field) */
static void field_laplacian_force(dslo_sph *s, int i, float out[3]) {
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float m = s->parts.mass;
  float force[3] = {0.0f, 0.0f, 0.0f};
  for (int j = 0; j < len; j++) {
    int jIndex = samples[j];
    if (jIndex != i) {
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float jDensity = q.density;
      float dv[3], v[3];
      vsub3(q.velocity, pi.velocity, dv);
      float inv = 1.0f / jDensity;
      for (int a = 0; a < 3; a++) v[a] = dv[a] * inv;
      float dist = dslo_vec_dist3(pi.position, q.position);
      if (!in_support(s, dist)) continue;
      float o2 = dslo_kernel_O2D(&s->kern, dist);
      if (s->prm.visc_running_mass) {
        /* force = force.Add(v.Scale(O2D)).Scale(m): m multiplies the running sum */
        for (int a = 0; a < 3; a++) {
          float t = v[a] * o2;
          float u = force[a] + t;
          force[a] = u * m;
        }
      } else {
        /* build-defined standard form: force += (v*O2D)*m */
        for (int a = 0; a < 3; a++) {
          float t = v[a] * o2;
          float u = t * m;
          force[a] = force[a] + u;
        }
      }
    }
  }
  for (int a = 0; a < 3; a++) out[a] = force[a];
}

/* ---- the field operators no solver calls ------------------------------
 * scalar fields: 0 = DensityField.Value (field_types.go:17-19), 1 = PressureField.Value
 * (TaitEos(rho, 87.0, 0), field_types.go:39-42); tensor fields: 0 = velocity, 1 = force. */
static float scalar_field_value(const dslo_sph *s, int field, int i) {
  if (field == 1) return pressure_field_value(s, i);
  return dslo_particles_get(&s->parts, i).density;
}
static void tensor_field_value(const dslo_sph *s, int field, int i, float out[3]) {
  dslo_particle q = dslo_particles_get(&s->parts, i);
  memcpy(out, field == 1 ? q.force : q.velocity, 3 * sizeof(float));
}

/* sph_field.go:203-227 Div */
float dslo_field_div(dslo_sph *s, int i, int tensor_field) {
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float div = 0.0f, mass = s->parts.mass;
  for (int j = 0; j < len; j++) {
    int jIndex = samples[j];
    if (jIndex != i) {
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float dir[3], nd[3], grad[3], fv[3], sv[3];
      vsub3(q.position, pi.position, dir);
      float dist = dslo_vec_mag(dir, 3);
      if (!in_support(s, dist)) continue;
      dslo_vec_norm3(dir, nd);
      dslo_kernel_grad(&s->kern, dist, nd, grad);
      tensor_field_value(s, tensor_field, jIndex, fv);
      float w = mass / q.density;
      for (int a = 0; a < 3; a++) sv[a] = fv[a] * w;
      div += dslo_vec_dot3(sv, grad);
    }
  }
  return div;
}

/* sph_field.go:272-294 Curl */
void dslo_field_curl(dslo_sph *s, int i, int tensor_field, float out[3]) {
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float curl[3] = {0.0f, 0.0f, 0.0f}, mass = s->parts.mass;
  for (int j = 0; j < len; j++) {
    int jIndex = samples[j];
    if (jIndex != i) {
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float dir[3], nd[3], grad[3], fv[3], sv[3], c[3];
      vsub3(q.position, pi.position, dir);
      float dist = dslo_vec_mag(dir, 3);
      if (!in_support(s, dist)) continue;
      dslo_vec_norm3(dir, nd);
      dslo_kernel_grad(&s->kern, dist, nd, grad);
      tensor_field_value(s, tensor_field, jIndex, fv);
      float w = mass / q.density;
      for (int a = 0; a < 3; a++) sv[a] = fv[a] * w;
      dslo_vec_cross3(sv, grad, c);
      for (int a = 0; a < 3; a++) curl[a] = curl[a] + c[a];
    }
  }
  memcpy(out, curl, sizeof(curl));
}

/* sph_field.go:230-248 Laplacian */
float dslo_field_laplacian(dslo_sph *s, int i, int scalar_field) {
  dslo_particle pi = dslo_particles_get(&s->parts, i);
  const int *samples;
  int len = dslo_get_samples(s, i, &samples);
  float m = s->parts.mass, sum = 0.0f;
  for (int j = 0; j < len; j++) {
    int jIndex = samples[j];
    dslo_particle q = dslo_particles_get(&s->parts, jIndex);
    if (jIndex != i) {
      float dist = dslo_vec_dist3(pi.position, q.position);
      if (!in_support(s, dist)) continue;
      float df = scalar_field_value(s, scalar_field, jIndex) - scalar_field_value(s, scalar_field, i);
      float t = m * (df / q.density);
      sum += t * dslo_kernel_O2D(&s->kern, dist);
    }
  }
  return sum;
}

/* sph_field.go:124-135 Interpolate */
float dslo_field_interpolate(dslo_sph *s, const float pos[3], int scalar_field) {
  const int *samples;
  int len = dslo_get_samples_from_position(s, pos, &samples);
  float sum = 0.0f, mass = s->parts.mass;
  for (int k = 0; k < len; k++) {
    dslo_particle q = dslo_particles_get(&s->parts, samples[k]);
    float dist = dslo_vec_dist3(pos, q.position);
    if (!in_support(s, dist)) continue;
    float weight = mass / q.density * dslo_kernel_F(&s->kern, dist);
    sum += weight * scalar_field_value(s, scalar_field, samples[k]);
  }
  return sum;
}

/* =====================================================================================
 * passes: model/sph/fluid.go
 * ===================================================================================== */

float dslo_cfl(dslo_sph *s) { /* fluid.go:111-114 */
  s->time = s->prm.dt;
  return s->time;
}

void dslo_density_all(dslo_sph *s) { /* fluid.go:127-131 */
  for (int i = 0; i < s->parts.n_particles; i++) field_density(s, i);
}

void dslo_pressure_all(dslo_sph *s) { /* fluid.go:134-142, particle.go:32-35 */
  for (int i = 0; i < s->particles; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    q.press = dslo_tait_eos_ex(q.density, s->parts.reference_density, 0.0f, s->prm.eos_w, s->prm.eos_gamma);
    dslo_particles_set(&s->parts, i, &q);
  }
}

void dslo_viscous_all(dslo_sph *s) { /* fluid.go:146-152 */
  for (int i = 0; i < s->particles; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    float lap[3];
    field_laplacian_force(s, i, lap);
    for (int a = 0; a < 3; a++) {
      float t = lap[a] * s->mu;
      q.force[a] += t;
    }
    dslo_particles_set(&s->parts, i, &q);
  }
}

void dslo_external_all(dslo_sph *s, const float f[3]) { /* fluid.go:155-161 */
  for (int i = 0; i < s->particles; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    for (int a = 0; a < 3; a++) q.force[a] += f[a];
    dslo_particles_set(&s->parts, i, &q);
  }
}

void dslo_gradient_pressure_force(dslo_sph *s) { /* fluid.go:164-172 */
  for (int i = 0; i < s->particles; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    float g[3];
    field_gradient(s, i, g);
    for (int a = 0; a < 3; a++) {
      float t = g[a] * s->prm.pressure_sign; /* +1 in the reference: exact */
      q.force[a] += t;
    }
    dslo_particles_set(&s->parts, i, &q);
  }
}

void dslo_update(dslo_sph *s) { /* fluid.go:175-197 */
  float m = 1.0f / s->parts.mass;
  float ts = dslo_cfl(s);
  for (int i = 0; i < s->particles; i++) {
    dslo_particle q = dslo_particles_get(&s->parts, i);
    for (int a = 0; a < 3; a++) {
      float acc = q.force[a] * m;
      float dv = acc * ts;
      q.velocity[a] += dv;
    }
    for (int a = 0; a < 3; a++) {
      float va = q.velocity[a];
      if (s->prm.xsph_eps != 0.0f && s->xsph) va = va + s->xsph[3 * i + a]; /* build-defined XSPH */
      float dx = va * ts;
      q.position[a] += dx;
    }
    float vm = dslo_vec_mag(q.velocity, 3);
    if (vm > s->max_vel) s->max_vel = vm;
    float fm = dslo_vec_mag(q.force, 3);
    if (fm > s->max_f) s->max_f = fm;
    q.press = 0.0f;
    for (int a = 0; a < 3; a++) q.force[a] = s->prm.force_reset[a];
    if (s->prm.walls) { /* build-defined: clamp + reflect, no reference counterpart */
      for (int a = 0; a < 3; a++) {
        if (q.position[a] < s->prm.box_min[a]) {
          q.position[a] = s->prm.box_min[a];
          if (q.velocity[a] < 0.0f) q.velocity[a] = -q.velocity[a] * s->prm.restitution;
        }
        if (q.position[a] > s->prm.box_max[a]) {
          q.position[a] = s->prm.box_max[a];
          if (q.velocity[a] > 0.0f) q.velocity[a] = -q.velocity[a] * s->prm.restitution;
        }
      }
    }
    dslo_particles_set(&s->parts, i, &q);
  }
}

/* build-defined (BASELINE configs[4]): cohesion F_i += kappa * sum_{j != i} m (x_j - x_i) F(r) */
void dslo_surface_tension_all(dslo_sph *s) {
  if (s->prm.st_kappa == 0.0f) return;
  for (int i = 0; i < s->particles; i++) {
    dslo_particle pi = dslo_particles_get(&s->parts, i);
    const int *samples;
    int len = dslo_get_samples(s, i, &samples);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int j = 0; j < len; j++) {
      int jIndex = samples[j];
      if (jIndex == i) continue;
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float d[3];
      vsub3(q.position, pi.position, d);
      float dist = dslo_vec_mag(d, 3);
      if (!in_support(s, dist)) continue;
      float w = s->parts.mass * dslo_kernel_F(&s->kern, dist);
      for (int a = 0; a < 3; a++) {
        float t = d[a] * w;
        acc[a] = acc[a] + t;
      }
    }
    for (int a = 0; a < 3; a++) {
      float t = acc[a] * s->prm.st_kappa;
      pi.force[a] += t;
    }
    dslo_particles_set(&s->parts, i, &pi);
  }
}

/* build-defined (BASELINE configs[4]): XSPH correction eps * sum_{j != i} (m/rho_j) (v_j - v_i) F(r),
 * evaluated before Update (old velocities); Update advects positions with v + correction. */
void dslo_xsph_all(dslo_sph *s) {
  if (s->prm.xsph_eps == 0.0f) return;
  if (!s->xsph) s->xsph = (float *)calloc((size_t)s->particles * 3 + 1, sizeof(float));
  for (int i = 0; i < s->particles; i++) {
    dslo_particle pi = dslo_particles_get(&s->parts, i);
    const int *samples;
    int len = dslo_get_samples(s, i, &samples);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int j = 0; j < len; j++) {
      int jIndex = samples[j];
      if (jIndex == i) continue;
      dslo_particle q = dslo_particles_get(&s->parts, jIndex);
      float dist = dslo_vec_dist3(pi.position, q.position);
      if (!in_support(s, dist)) continue;
      float w = (s->parts.mass / q.density) * dslo_kernel_F(&s->kern, dist);
      float dv[3];
      vsub3(q.velocity, pi.velocity, dv);
      for (int a = 0; a < 3; a++) {
        float t = dv[a] * w;
        acc[a] = acc[a] + t;
      }
    }
    for (int a = 0; a < 3; a++) s->xsph[3 * i + a] = acc[a] * s->prm.xsph_eps;
  }
}

float dslo_cache_incr(dslo_sph *s, int *rebuilt) { /* fluid.go:208-215 */
  if (rebuilt) *rebuilt = 0;
  s->cache_life *= s->cache_life;
  if (s->cache_life < 0.1f) {
    s->cache_life = 0.8f;
    dslo_sampler_update(s); /* NN() fluid.go:100-102 */
    if (rebuilt) *rebuilt = 1;
  }
  return s->cache_life;
}

/* fluid.go:221-277 pcidelta + computeBeta.  The nested Init(1.0, Vec3(), nil, 8, false)
 * contributes only its lattice positions and kernel (h = 1). */
float dslo_pcidelta(dslo_sph *s) {
  const int n3 = 8, particles = 512;
  float *pos = (float *)malloc(sizeof(float) * 3 * particles);
  float origin[3] = {0.0f, 0.0f, 0.0f};
  dslo_lattice_positions(n3, origin, 3, pos);
  dslo_kernel kern = dslo_build_kernel(1.0f);
  float denom = 0.0f, denom1[3] = {0.0f, 0.0f, 0.0f}, denom2 = 0.0f;
  int mid_index = particles / 2;
  int tracking_index = 0;
  for (int i = 0; i < particles; i++) {
    int mod = i % 2;
    int x = mid_index + tracking_index;
    if (mod != 0) {
      x = mid_index - tracking_index;
      tracking_index++;
    }
    if (x < 0 || x > particles) break;
    /* Particles.Get(x): index == n_particles yields the zero particle (never reached) */
    float point[3] = {0.0f, 0.0f, 0.0f};
    if (x < particles) memcpy(point, &pos[3 * x], sizeof(point));
    float mg = dslo_vec_mag(point, 3);
    float dist2 = mg * mg;
    float h0 = kern.H_;
    if (dist2 < h0 * h0) {
      float dist = dslo_vec_mag(point, 3);
      float dir[3] = {0.0f, 0.0f, 0.0f};
      if (dist > 0.0f) {
        float inv = 1.0f / dist;
        for (int a = 0; a < 3; a++) dir[a] = point[a] * inv;
      }
      float g[3];
      dslo_kernel_grad(&kern, dist, dir, g);
      for (int a = 0; a < 3; a++) denom1[a] = denom1[a] + g[a];
      denom2 += dslo_vec_dot3(g, g);
    }
  }
  free(pos);
  denom += -dslo_vec_dot3(denom1, denom1) - denom2;
  if (denom != 0.0f) {
    /* computeBeta fluid.go:275-277 */
    float t2 = s->time * s->time;
    float m2 = s->parts.mass * s->parts.mass;
    float d2 = s->parts.reference_density * s->parts.reference_density;
    float beta = t2 * m2 * (2.0f / d2);
    s->delta = -1.0f / (beta * denom);
    return s->delta;
  }
  return 0.0f;
}

/* =====================================================================================
 * drivers
 * ===================================================================================== */

/* solver/wcsph/wcsph.go:14-26, one iteration of Run.  The two optional force passes are
 * build-defined (disabled in the reference parameter set). */
void dslo_wcsph_step(dslo_sph *s) {
  dslo_sampler_update(s); /* grid modes: neighbour rule is geometric -> rebuild per step;
                             lsh_ref: the sampler goroutine does this on "SAMPLER_UPDATE" */
  dslo_density_all(s);
  if (s->prm.wcsph_pressure_force) dslo_gradient_pressure_force(s);
  if (s->prm.wcsph_viscosity) dslo_viscous_all(s);
  dslo_surface_tension_all(s);
  dslo_external_all(s, s->prm.external);
  dslo_pressure_all(s);
  dslo_xsph_all(s);
  dslo_update(s);
  dslo_cfl(s);
}

/* pcisph_darwin.go:24-41: predictor copies made once, never re-synchronised. */
void dslo_pcisph_begin(dslo_sph *s) {
  int n = s->parts.n_particles;
  free(s->pci_pos);
  free(s->pci_vel);
  s->pci_pos = (float *)malloc(sizeof(float) * 3 * (size_t)n);
  s->pci_vel = (float *)malloc(sizeof(float) * 3 * (size_t)n);
  memcpy(s->pci_pos, s->parts.positions, sizeof(float) * 3 * (size_t)n);
  memcpy(s->pci_vel, s->parts.velocities, sizeof(float) * 3 * (size_t)n);
}

/* pcisph_darwin.go:43-101, one iteration of the outer loop. */
void dslo_pcisph_step(dslo_sph *s) {
  dslo_particles *f = &s->parts;
  float refDensity = f->reference_density;
  int num = f->n_particles;
  dslo_sampler_update(s);
  dslo_density_all(s);
  dslo_viscous_all(s);
  dslo_surface_tension_all(s);
  float max_error_ratio = 0.0f, density_error = 0.0f;
  int iters = 0;
  for (int iter = 0; iter < s->prm.pci_max_iters; iter++) {
    iters = iter + 1;
    max_error_ratio = 0.0f;
    float t = dslo_cfl(s);
    for (int index = 0; index < num; index++) { /* predict :57-73 */
      int x = index * 3;
      float inv_m = 1.0f / f->mass;
      for (int a = 0; a < 3; a++) {
        float accel = f->forces[x + a] * inv_m;
        float dv = accel * t;
        float tv = s->pci_vel[x + a] + dv;
        float dp = tv * t;
        float tp = s->pci_pos[x + a] + dp;
        s->pci_pos[x + a] = tp;
        s->pci_vel[x + a] = tv;
      }
    }
    for (int index = 0; index < num; index++) { /* pressure from density error :76-92 */
      float calc = dslo_density_f(s, &s->pci_pos[3 * index]);
      density_error = calc - refDensity;
      float abs_err = density_error / refDensity;
      float dp = density_error * s->delta;
      f->pressures[index] += dp;
      if (abs_err > max_error_ratio) max_error_ratio = abs_err;
    }
    dslo_gradient_pressure_force(s); /* :93 */
    if (max_error_ratio <= s->prm.pci_max_error) break; /* :95-98 */
  }
  s->pci_last_error = max_error_ratio;
  s->pci_last_iters = iters;
  dslo_xsph_all(s);
  dslo_update(s); /* :101 */
}

/* =====================================================================================
 * build-defined synthetic dam-break input; no reference counterpart
 * ===================================================================================== */
uint64_t dslo_xorshift64s(uint64_t *state) {
  uint64_t x = *state;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *state = x;
  return x * 0x2545F4914F6CDD1DULL;
}

/* counter-based splitmix64: r(c) = mix(seed + (c+1)*0x9E3779B97F4A7C15) */
uint64_t dslo_splitmix64(uint64_t seed, uint64_t counter) {
  uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* Fluid block of n3^3 particles, spacing dx, lower corner at (0.5dx)^3, uniform jitter
 * of +-jitter*dx per axis drawn from splitmix64(seed, 3*id + axis).
 * id = k + n3*(i*n3 + j) as in AlignWithGrid. */
void dslo_dambreak_positions(int n3, float dx, float jitter, uint64_t seed, float *pos) {
  for (int i = 0; i < n3; i++)
    for (int j = 0; j < n3; j++)
      for (int k = 0; k < n3; k++) {
        int id = k + n3 * (i * n3 + j);
        float ijk[3] = {(float)i, (float)j, (float)k};
        for (int a = 0; a < 3; a++) {
          uint64_t r = dslo_splitmix64(seed, (uint64_t)(3 * (long)id + a));
          float u = (float)(r >> 40) * (1.0f / 16777216.0f); /* [0,1) */
          float jit = (u * 2.0f - 1.0f) * (jitter * dx);
          float base = (ijk[a] + 0.5f) * dx;
          pos[3 * id + a] = base + jit;
        }
      }
}

/* =====================================================================================
 * accessors for the ctypes test harness (oracle/pyoracle.py)
 * ===================================================================================== */
int dslo_n(const dslo_sph *s) { return s->parts.n_particles; }
int dslo_total(const dslo_sph *s) { return s->parts.n_particles + s->parts.n_boundary; }
float *dslo_positions(dslo_sph *s) { return s->parts.positions; }
float *dslo_velocities(dslo_sph *s) { return s->parts.velocities; }
float *dslo_forces(dslo_sph *s) { return s->parts.forces; }
float *dslo_densities(dslo_sph *s) { return s->parts.densities; }
float *dslo_pressures(dslo_sph *s) { return s->parts.pressures; }
float *dslo_pci_positions(dslo_sph *s) { return s->pci_pos; }
float *dslo_pci_velocities(dslo_sph *s) { return s->pci_vel; }
float dslo_get_delta(const dslo_sph *s) { return s->delta; }
void dslo_set_delta(dslo_sph *s, float d) { s->delta = d; }
float dslo_get_time(const dslo_sph *s) { return s->time; }
float dslo_get_max_vel(const dslo_sph *s) { return s->max_vel; }
float dslo_get_max_f(const dslo_sph *s) { return s->max_f; }
float dslo_get_pci_error(const dslo_sph *s) { return s->pci_last_error; }
int dslo_get_pci_iters(const dslo_sph *s) { return s->pci_last_iters; }
int dslo_get_lsh_size(const dslo_sph *s) { return s->smp.size; }
float dslo_get_ref_density(const dslo_sph *s) { return s->parts.reference_density; }
void dslo_get_kernel(const dslo_sph *s, dslo_kernel *out) { *out = s->kern; }
size_t dslo_params_sizeof(void) { return sizeof(dslo_params); }
/* per-particle operators exposed for unit checks */
void dslo_field_gradient(dslo_sph *s, int i, float out[3]) { field_gradient(s, i, out); }
void dslo_field_laplacian_force(dslo_sph *s, int i, float out[3]) { field_laplacian_force(s, i, out); }
int dslo_lsh_hash_pos(const dslo_sph *s, const float pos[3]) { return dslo_lsh_hash(&s->smp, pos); }
