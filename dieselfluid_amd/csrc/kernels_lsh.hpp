// kernels_lsh.hpp -- the reference's LSH sampler on the device (sampler/lsh/lsh.go), for
// bit-level comparison with the reference arithmetic.  Not a performance
// path: the engine's neighbour structure is the uniform grid.
//
//   k_lsh_bucket  : Hash(position) for every particle                      lsh.go:102-111
//   k_lsh_table   : bucket lists in ascending particle order               lsh.go:113-133
//                   (one wave per bucket, ballot-ordered append; only the first `cap`
//                   entries of a list are ever read, so the scan stops there)
//   k_lsh_samples : the 100-sample list GetSamples returns for each bucket lsh.go:136-158
// GetSamples(x) depends only on Hash(position of x), so one list per bucket serves every
// particle and every predicted position (GetSamplesFromPosition, lsh.go:160-181).
#pragma once

#include "kernels_sph.hpp"

namespace dsl {

__global__ __launch_bounds__(kBlock) void k_lsh_bucket(int n, Neigh nb, const float* __restrict__ px,
                                                       const float* __restrict__ py, const float* __restrict__ pz,
                                                       int* __restrict__ bucket_of) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) bucket_of[i] = lsh_hash(nb, px[i], py[i], pz[i]);
}

// one wave (64 lanes) per bucket
__global__ __launch_bounds__(kWave) void k_lsh_table(int n, const int* __restrict__ bucket_of, int cap,
                                                     int* __restrict__ table, int* __restrict__ len) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int base = 0;
  for (int i0 = 0; i0 < n && base < cap; i0 += kWave) {
    const int i = i0 + lane;
    const bool mine = i < n && bucket_of[i] == b;
    const unsigned long long m = __ballot(mine);
    if (mine) {
      const int d = base + __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (d < cap) table[(size_t)b * cap + d] = i;
    }
    base += __builtin_popcountll(m);
  }
  if (lane == 0) len[b] = base < cap ? base : cap;  // only min(len, cap) is ever consulted
}

// lsh.go:136-158: start at the bucket, skip nil buckets cyclically, copy entries until 100 are
// collected, re-reading the same bucket from its start when it holds fewer than 100.
__global__ void k_lsh_samples(int buckets, int cap, const int* __restrict__ table, const int* __restrict__ len,
                              int* __restrict__ samples) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= buckets) return;
  int num = 0, index = b, empty_run = 0;
  while (num < kLshSamples) {
    if (len[index] == 0) {
      index = (index + 1) % buckets;
      if (++empty_run > buckets) break;  // no particle at all: the Go loop would spin forever
    } else {
      empty_run = 0;
      for (int j = 0; j < len[index] && num < kLshSamples; ++j) samples[b * kLshSamples + num++] = table[(size_t)index * cap + j];
    }
  }
  for (; num < kLshSamples; ++num) samples[b * kLshSamples + num] = 0;
}

}  // namespace dsl
