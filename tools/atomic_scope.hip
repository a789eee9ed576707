// What a returning atomic on a random word of a 16 MB histogram costs at device scope and at workgroup scope (which the
// hardware executes in the issuing XCD's L2): 4M of them, the query-binning kernel's pattern (tools only).
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomic_scope tools/atomic_scope.hip && tools/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE>
__global__ void k_atomics(int* hist, int ncell, const int* cells, int* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = cells[i];
  int r;
  if (SCOPE == 0) r = __hip_atomic_fetch_add(&hist[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else r = __hip_atomic_fetch_add(&hist[c], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  out[i] = r;
}
__global__ void k_stream(const int* cells, int* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = cells[i] + 1;
}

int run(int ncell);
int main() {
  for (int ncell : {4096000, 512000, 64000, 8000}) run(ncell);
  return 0;
}
int run(int ncell) {
  const int n = 4096000;
  std::printf("-- histogram of %d words\n", ncell);
  std::vector<int> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (int)(s % (unsigned long long)ncell);
  }
  int *cells, *hist, *out;
  hipMalloc(&cells, n * 4); hipMalloc(&hist, ncell * 4); hipMalloc(&out, n * 4);
  hipMemcpy(cells, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](auto launch, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipMemset(hist, 0, ncell * 4);
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    std::printf("%-28s %.4f ms for %d\n", name, best, n);
  };
  dim3 g((n + 255) / 256), t(256);
  time([&] { hipLaunchKernelGGL(k_stream, g, t, 0, 0, cells, out, n); }, "stream (read 4 B, write 4 B)");
  time([&] { hipLaunchKernelGGL(k_atomics<0>, g, t, 0, 0, hist, ncell, cells, out, n); }, "atomic, device scope");
  time([&] { hipLaunchKernelGGL(k_atomics<1>, g, t, 0, 0, hist, ncell, cells, out, n); }, "atomic, workgroup scope");
  return 0;
}
