// Probe of the LDS-DMA builtin on gfx950: destination addressing and EXEC masking.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
__global__ void probe(const u32* __restrict__ src, u32* __restrict__ out, int mode) {
  __shared__ __attribute__((aligned(16))) u32 lds[2 * 64 * 4 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += blockDim.x) lds[i] = 0xdeadbeefu;
  __syncthreads();
  // each lane's source: 16 bytes at src + (63 - lane) * 4 dwords (reversed so we can tell lanes apart)
  const u32* g = src + (wave * 64 + (63 - lane)) * 4;
  u32* base = lds + wave * 256;   // wave-uniform
  if (mode == 0 || (lane % 3) != 0) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)base, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int n = 2 * 64 * 4;
  std::vector<u32> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  u32 *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, (n + 64) * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 128>>>(d, o, mode);
    std::vector<u32> r(n + 64);
    hipMemcpy(r.data(), o, (n + 64) * 4, hipMemcpyDeviceToHost);
    int ok = 1, untouched = 0;
    for (int w = 0; w < 2; ++w)
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
          const u32 got = r[w * 256 + l * 4 + k], want = (w * 64 + (63 - l)) * 4 + k;
          const bool active = mode == 0 || (l % 3) != 0;
          if (active && got != want) { ok = 0; printf("mode %d w %d l %d k %d got %u want %u\n", mode, w, l, k, got, want); }
          if (!active) { if (got == 0xdeadbeefu) ++untouched; else { ok = 0; printf("mode %d inactive lane %d written %u\n", mode, l, got); } }
        }
    for (int i = n; i < n + 64; ++i) if (r[i] != 0xdeadbeefu) { ok = 0; printf("tail %d touched\n", i); }
    printf("mode %d: %s, untouched dwords of masked lanes %d\n", mode, ok ? "OK" : "MISMATCH", untouched);
  }
  return 0;
}
