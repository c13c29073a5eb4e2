// exp_gather.hip — price list for the shapes k_probe's index accesses can take on MI355X (round 3: what bounds the probe once every
// slot fetch is an L2 hit?).  Each lane gathers from a table of 64-B slots at a hashed slot index, in one of several shapes, at the
// probe's occupancy (8 waves per SIMD) or at 4; `MLP` independent gathers are in flight per lane.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/exp_gather tools/exp_gather.hip      run on the GPU box: ./tools/exp_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u64 mix(u64 x) {
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}

enum { S_DWORD = 0, S_X2, S_X4, S_2X4, S_4X4, S_X4_DEP_X4, S_X4_THEN_32, S_LDS_X4, S_BITMAP_THEN_X4, S_N };
static const char *shape_name[S_N] = {
    "4 B (dword)", "8 B (dwordx2)", "16 B (dwordx4)", "32 B (2 x dwordx4, one slot)", "64 B (4 x dwordx4, one slot)",
    "16 B, then 16 B of the same slot (dependent)", "16 B, then 32 B more of the slot for half the lanes (dependent)",
    "16 B straight to LDS (global_load_lds_dwordx4) + ds_read_b128", "bitmap word (L2) -> 16 B slot for the lanes whose bit is set (half)"};

template <int SHAPE, int MLP, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k(const uint4 *__restrict__ tab, u64 nslots, const u32 *__restrict__ bits,
                                                                                             size_t n, u32 *sink) {
  __shared__ uint4 land[MLP][256];
  u32 acc = 0;
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i0 = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * MLP) {
    u64 s[MLP];
#pragma unroll
    for (int m = 0; m < MLP; ++m) s[m] = mix(i0 + m * stride) % nslots;
    if (SHAPE == S_DWORD) {
      u32 v[MLP];
#pragma unroll
      for (int m = 0; m < MLP; ++m) v[m] = ((const u32 *) (tab + s[m] * 4))[0];
#pragma unroll
      for (int m = 0; m < MLP; ++m) acc += v[m];
    } else if (SHAPE == S_X2) {
      uint2 v[MLP];
#pragma unroll
      for (int m = 0; m < MLP; ++m) v[m] = ((const uint2 *) (tab + s[m] * 4))[0];
#pragma unroll
      for (int m = 0; m < MLP; ++m) acc += v[m].x ^ v[m].y;
    } else if (SHAPE == S_X4 || SHAPE == S_2X4 || SHAPE == S_4X4) {
      constexpr int P = SHAPE == S_X4 ? 1 : (SHAPE == S_2X4 ? 2 : 4);
      uint4 v[MLP][P];
#pragma unroll
      for (int m = 0; m < MLP; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) v[m][p] = tab[s[m] * 4 + p];
#pragma unroll
      for (int m = 0; m < MLP; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) acc += v[m][p].x ^ v[m][p].y ^ v[m][p].z ^ v[m][p].w;
    } else if (SHAPE == S_X4_DEP_X4 || SHAPE == S_X4_THEN_32) {
      uint4 v[MLP];
#pragma unroll
      for (int m = 0; m < MLP; ++m) v[m] = tab[s[m] * 4];
#pragma unroll
      for (int m = 0; m < MLP; ++m) {
        acc += v[m].y ^ v[m].z;
        if (SHAPE == S_X4_DEP_X4) { const uint4 w = tab[s[m] * 4 + 1 + (v[m].x & 1u)]; acc += w.x ^ w.w; }
        else if (v[m].x & 1u) { const uint4 w0 = tab[s[m] * 4 + 1], w1 = tab[s[m] * 4 + 2]; acc += w0.x ^ w0.w ^ w1.y; }
      }
    } else if (SHAPE == S_LDS_X4) {
#pragma unroll
      for (int m = 0; m < MLP; ++m) {
        const uint4 *src = tab + s[m] * 4;
        // one wave writes 1 KiB: lane l's 16 bytes land at base + 16 l
        const u32 lds_dst = (u32) (size_t) &land[m][threadIdx.x & ~63u];
        u32 keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int m = 0; m < MLP; ++m) { const uint4 v = land[m][threadIdx.x]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    } else if (SHAPE == S_BITMAP_THEN_X4) {
      u32 b[MLP];
#pragma unroll
      for (int m = 0; m < MLP; ++m) b[m] = bits[s[m] >> 5];
      uint4 v[MLP];
#pragma unroll
      for (int m = 0; m < MLP; ++m) { v[m] = make_uint4(0, 0, 0, 0); if ((b[m] >> (s[m] & 31)) & 1u) v[m] = tab[s[m] * 4]; }
#pragma unroll
      for (int m = 0; m < MLP; ++m) acc += v[m].x ^ v[m].w;
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int SHAPE, int MLP, int WAVES>
static float run(const uint4 *tab, u64 nslots, const u32 *bits, size_t n, u32 *sink) {
  hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<SHAPE, MLP, WAVES>), dim3(2048), dim3(256), 0, 0, tab, nslots, bits, n, sink);
  CHK(hipEventRecord(a, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, MLP, WAVES>), dim3(2048), dim3(256), 0, 0, tab, nslots, bits, n, sink);
  CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
  float ms; CHK(hipEventElapsedTime(&ms, a, b));
  CHK(hipEventDestroy(a)); CHK(hipEventDestroy(b));
  return ms / 5;
}

template <int SHAPE>
static void row(const uint4 *tab, u64 nslots, const u32 *bits, size_t n, u32 *sink) {
  const float a = run<SHAPE, 1, 8>(tab, nslots, bits, n, sink), b = run<SHAPE, 2, 8>(tab, nslots, bits, n, sink), c = run<SHAPE, 4, 8>(tab, nslots, bits, n, sink),
              d = run<SHAPE, 4, 4>(tab, nslots, bits, n, sink);
  printf("  %-84s %.3f  %.3f  %.3f  %.3f\n", shape_name[SHAPE], a, b, c, d);
}

int main() {
  const size_t n = 1 << 24;
  u32 *sink; CHK(hipMalloc(&sink, 4));
  for (u64 mb : {2ull, 24ull, 192ull, 768ull}) {
    const u64 nslots = (mb << 20) / 64;
    uint4 *tab; u32 *bits;
    CHK(hipMalloc(&tab, nslots * 64)); CHK(hipMalloc(&bits, nslots / 8 + 4));
    CHK(hipMemset(tab, 0x5A, nslots * 64)); CHK(hipMemset(bits, 0x55, nslots / 8 + 4));   // every other bit set
    printf("table %llu MB of 64-B slots, bitmap %llu KB; ms per 2^24 gathers at (MLP 1, 8 waves/SIMD) (2, 8) (4, 8) (4, 4 waves/SIMD)\n", mb, nslots / 8 / 1024);
    row<S_DWORD>(tab, nslots, bits, n, sink);
    row<S_X2>(tab, nslots, bits, n, sink);
    row<S_X4>(tab, nslots, bits, n, sink);
    row<S_2X4>(tab, nslots, bits, n, sink);
    row<S_4X4>(tab, nslots, bits, n, sink);
    row<S_X4_DEP_X4>(tab, nslots, bits, n, sink);
    row<S_X4_THEN_32>(tab, nslots, bits, n, sink);
    row<S_LDS_X4>(tab, nslots, bits, n, sink);
    row<S_BITMAP_THEN_X4>(tab, nslots, bits, n, sink);
    CHK(hipFree(tab)); CHK(hipFree(bits));
  }
  return 0;
}
