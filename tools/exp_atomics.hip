// exp_atomics.hip — what do per-record device-scope atomics on random read-name slots cost on MI355X?
// ("measure the sort-free merge": atomicOr(grpU[qid]) + atomicAdd(seen[qid]) per multi-hit record.)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp_atomics tools/exp_atomics.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u64 mix(u64 x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
// mode 0: atomicOr u64 no return; 1: atomicAdd u32 with return (value used); 2: both (the merge's per-record work);
// 3: plain 8-B gather (read only); 4: plain 8-B scatter store; 5: like 2 but keys are "windowed": record i picks a name in
// [i/W*G, i/W*G + G) (hits spread over a window of W tuples, as bench c3's scatter 2^16)
__global__ __launch_bounds__(256) void k(u64 *tab, u32 *cnt, u64 T, size_t n, int mode, u32 *sink) {
  u32 acc = 0;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
    const u64 h = mix(i);
    u64 q = h % T;
    if (mode == 5) { const u64 W = 65536, G = 6242; q = ((i / W) * G + h % G) % T; }
    const u64 m = 1ull << (h >> 58);
    if (mode == 0) atomicOr(&tab[q], m);
    else if (mode == 1) acc += atomicAdd(&cnt[q], 1u);
    else if (mode == 2 || mode == 5) { atomicOr(&tab[q], m); acc += atomicAdd(&cnt[q], 1u); }
    else if (mode == 3) acc += (u32) tab[q];
    else tab[q] = m;
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  const size_t n = 1 << 24;
  u32 *sink; CHK(hipMalloc(&sink, 4));
  const char *names[] = {"atomicOr u64 (no return)", "atomicAdd u32 (returning)", "atomicOr u64 + returning atomicAdd u32", "plain 8-B gather", "plain 8-B scatter store",
                         "Or + Add, names windowed like c3 (6242 names per 65536 tuples)"};
  for (u64 T : {1ull << 21, 1ull << 24, 1ull << 26}) {
    u64 *tab; u32 *cnt;
    CHK(hipMalloc(&tab, T * 8)); CHK(hipMalloc(&cnt, T * 4));
    CHK(hipMemset(tab, 0, T * 8)); CHK(hipMemset(cnt, 0, T * 4));
    for (int mode = 0; mode < 6; ++mode) {
      hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
      hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, tab, cnt, T, n, mode, sink);   // warm
      CHK(hipEventRecord(a, 0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, tab, cnt, T, n, mode, sink);
      CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
      float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      printf("T=2^%d names  %-62s %.3f ms per 2^24 records = %.1f G records/s\n", (int) __builtin_ctzll(T), names[mode], ms, n / ms / 1e6);
    }
    CHK(hipFree(tab)); CHK(hipFree(cnt));
  }
  return 0;
}
