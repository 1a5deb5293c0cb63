// Probe: buffer_load ... lds (LDS-DMA through a buffer resource) on gfx950 -- what out-of-range lanes do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
__global__ void probe(const u32* __restrict__ src, u32* __restrict__ out, int soff) {
  __shared__ __attribute__((aligned(1024))) u32 lds[512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  // lanes with lane%3==0 are out of range (offset 0x80000000); the others read 16 bytes reversed by lane
  const u32 voff = (lane % 3 == 0) ? 0x80000000u : (u32)((wave * 64 + (63 - lane)) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = lds[i];
}
int main() {
  const int n = 2 * 64 * 4 + 64;
  std::vector<u32> h(n);
  for (int i = 0; i < n; ++i) h[i] = i + 1;
  u32 *d, *o;
  if (hipMalloc(&d, n * 4) != hipSuccess || hipMalloc(&o, 512 * 4) != hipSuccess) return 1;
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int soff = 0; soff <= 64; soff += 64) {
    probe<<<1, 128>>>(d, o, soff);
    std::vector<u32> r(512);
    (void)hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    int ok = 1, zero = 0, untouched = 0, other = 0;
    for (int w = 0; w < 2; ++w)
      for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 4; ++k) {
          const u32 got = r[w * 256 + l * 4 + k];
          if (l % 3 == 0) { if (got == 0) ++zero; else if (got == 0xdeadbeefu) ++untouched; else ++other; }
          else { const u32 want = (w * 64 + (63 - l)) * 4 + k + 1 + soff / 4; if (got != want) { ok = 0; if (other < 5) printf("w %d l %d k %d got %u want %u\n", w, l, k, got, want); } }
        }
    printf("soffset %d: in-range lanes %s; out-of-range lanes: %d zero, %d untouched, %d other dwords\n", soff, ok ? "OK" : "MISMATCH", zero, untouched, other);
  }
  return 0;
}
