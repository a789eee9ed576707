/* asan_driver.c -- runs the oracle's main paths under -fsanitize=address,undefined
 * (`make -C oracle asan-run`; tests/test_oracle_asan.py).  Test infrastructure only. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsl_oracle.h"

static int finite_all(const float *v, int n) {
  for (int i = 0; i < n; i++)
    if (!isfinite(v[i])) return 0;
  return 1;
}

int main(void) {
  const float origin[3] = {0.f, 0.f, 0.f};
  float hv[8 * 3];
  uint64_t st = 42;
  for (int i = 0; i < 24; i++) hv[i] = (float)((double)(dslo_xorshift64s(&st) >> 11) / 9007199254740992.0) - 0.5f;

  /* sph.Init (lsh_ref, with pcidelta), reference WCSPH and PCISPH loops */
  dslo_params p = dslo_params_reference(8);
  dslo_sph *s = dslo_sph_init(&p, origin, 3, hv, 8, 1);
  if (!s) return 2;
  for (int k = 0; k < 3; k++) dslo_wcsph_step(s);
  dslo_pcisph_begin(s);
  for (int k = 0; k < 2; k++) dslo_pcisph_step(s);
  int ok = finite_all(s->parts.positions, 3 * s->parts.n_particles);
  int *tab = (int *)malloc(sizeof(int) * (size_t)s->smp.buckets * (size_t)s->smp.size);
  dslo_lsh_get_data_1d(s, tab);
  free(tab);
  dslo_sph_free(s);

  /* the zero-length origin of sph_test.go:10: every particle at the origin */
  s = dslo_sph_init(&p, origin, 0, hv, 8, 0);
  if (!s) return 2;
  ok &= s->parts.positions[3 * 100] == 0.0f;
  dslo_sph_free(s);

  /* grid neighbours, build-defined dam-break terms, both orders, field operators */
  for (int order = 0; order < 2; order++) {
    p = dslo_params_reference(8);
    p.neigh_mode = DSLO_NEIGH_GRID;
    p.neigh_order = order;
    p.h = 0.5f;
    p.wcsph_pressure_force = 1;
    p.wcsph_viscosity = 1;
    p.walls = 1;
    p.xsph_eps = 0.25f;
    p.st_kappa = 0.1f;
    for (int a = 0; a < 3; a++) {
      p.box_min[a] = -1.0f;
      p.box_max[a] = 1.0f;
      p.grid_min[a] = -1.5f;
      p.grid_max[a] = 1.5f;
    }
    const int n = 8 * 8 * 8;
    float *pos = (float *)malloc(sizeof(float) * 3 * (size_t)n);
    dslo_lattice_positions(8, origin, 3, pos);
    s = dslo_sph_from_state(&p, n, pos, NULL, NULL, NULL, 0);
    free(pos);
    if (!s) return 2;
    for (int k = 0; k < 3; k++) dslo_wcsph_step(s);
    dslo_pcisph_begin(s);
    dslo_pcisph_step(s);
    dslo_density_all(s);
    float c[3];
    (void)dslo_field_div(s, 7, 0);
    dslo_field_curl(s, 7, 0, c);
    (void)dslo_field_laplacian(s, 7, 0);
    (void)dslo_field_interpolate(s, origin, 1);
    ok &= finite_all(s->parts.positions, 3 * s->parts.n_particles);
    dslo_sph_free(s);
  }
  printf(ok ? "asan driver ok\n" : "asan driver: non-finite state\n");
  return ok ? 0 : 1;
}
