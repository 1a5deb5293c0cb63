// Transposed conv 2x2 stride 2 (models/unet2d.py:46-48 of the reference) forward and input
// gradient for the levels with many pixels and few channels, bf16.
//
// Both are GEMMs with a tiny reduction (K = Cin forward, 4*Cout backward: 64..256) over very many
// pixels: HBM-bound streaming.  The whole weight matrix (at most 64 KB) is parked in LDS once per
// workgroup; every wave then takes 32 pixels at a time, loads their K values from global memory
// STRAIGHT INTO MFMA fragment registers (a lane's 16 bytes are 8 consecutive k of one pixel, which
// is exactly the B-operand layout), multiplies against the LDS weights 64 output channels at a
// time and stores.  No LDS staging of pixels, no barrier in the pixel loop, 16 waves per CU.
// Layers with fewer pixel blocks than CUs, or with a weight matrix beyond 64 KB, are cut along the
// output channels as well (blockIdx.y): a workgroup then parks only its slice of the weights and its
// pixel fragments are fetched once per slice (from L2 after the first).
//   forward : out[b][2y+a][2x+c][co] = bias[co] + sum_ci x[b][y][x][ci] * W[ci][co][a][c]   (pixel-shuffle store)
//   backward: dx[b][y][x][ci] = [prev > 0] * sum_{a,c,co} dy[b][2y+a][2x+c][co] * W[ci][co][a][c]
// Output channels are permuted inside each block of 64 (weight row of MFMA tile i, row r is
// channel 16*(r>>2) + 4*i + (r&3)) so that a lane ends up with 16 CONSECUTIVE channels of its
// pixel: two 16-byte stores instead of four 8-byte ones.
#include <atomic>

#include "igemm_impl.h"

namespace mi {
namespace {

constexpr int kTconvThreads = 512;
constexpr int kTconvBlockPix = kTconvThreads / 64 * 32;   // 256 pixels per workgroup iteration
constexpr int kTconvMaxLds = 64 * 1024;

template <int KF, int MODE>
__global__ __launch_bounds__(kTconvThreads, 2) void tconv_kernel(const IgemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = a.N / (int)gridDim.y, nbase = (int)blockIdx.y * NS;   // this workgroup's slice of the output channels
  char* sW = smem;                                            // [NS/16 tiles][KF slabs][16 rows][64 B], swizzled
  float* sBias = (float*)(smem + (size_t)NS * KF * 64);       // [NS] (forward only)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r16 = lane & 15, ch = lane >> 4;
  const __bf16* __restrict__ in = (const __bf16*)a.in;
  const __bf16* __restrict__ wp = (const __bf16*)a.w;

  // ---- weights (and bias) into LDS, once
  const int nchunks = NS * KF * 4;
  for (int idx = t; idx < nchunks; idx += kTconvThreads) {
    const int c = idx & 3, ks = (idx >> 2) % KF, tr = (idx >> 2) / KF;     // tr = tile*16 + row
    const int T = tr >> 4, r = tr & 15;
    const int n = (T >> 2) * 64 + 16 * (r >> 2) + 4 * (T & 3) + (r & 3);
    const u32x4_t v = *(const u32x4_t*)(wp + (int64_t)(nbase + n) * a.Kp + ks * 32 + c * 8);
    *(u32x4_t*)(sW + (T * KF + ks) * 1024 + r * 64 + ((c ^ (((r >> 2) & 1) << 1)) << 4)) = v;
  }
  if (MODE == G_1x1) {
    for (int n = t; n < NS; n += kTconvThreads) {
      const int co = (nbase + n) % a.shufC;
      sBias[n] = co < a.n_real ? a.bias[co] : 0.f;
    }
  }
  __syncthreads();

  const int frag_off = r16 * 64 + ((ch ^ (((r16 >> 2) & 1) << 1)) << 4);
  const int HW = a.Ho * a.Wo;
  const int nblk = (a.M + kTconvBlockPix - 1) / kTconvBlockPix;
  __bf16* __restrict__ out = (__bf16*)a.out;
  const __bf16* __restrict__ mask = (const __bf16*)a.mask;

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    // ---- this wave's 2 x 16 pixels, straight into fragment registers
    bf16x8_t pf[2][KF];
    int64_t opix[2];
    bool live[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = blk * kTconvBlockPix + wave * 32 + j * 16 + r16;
      live[j] = p < a.M;
      const uint32_t pc = live[j] ? (uint32_t)p : 0u;          // tail lanes read pixel 0, never stored
      const uint32_t b = fdiv(pc, a.divHW);
      const uint32_t r = pc - b * (uint32_t)HW;
      const uint32_t y = fdiv(r, a.divW);
      const uint32_t x = r - y * (uint32_t)a.Wo;
      const int64_t up = ((int64_t)(b * a.Ho + y) * 2) * (2 * a.Wo) + 2 * x;   // pixel (2y, 2x) of the 2Ho x 2Wo image
      if (MODE == G_1x1) {
        opix[j] = up;
        const __bf16* src = in + (int64_t)pc * a.in_cs + a.in_co + ch * 8;
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) pf[j][ks] = *(const bf16x8_t*)(src + ks * 32);
      } else {
        opix[j] = pc;
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
          const int k = ks * 32, tap = k / a.Cin, c = k - tap * a.Cin + ch * 8;
          pf[j][ks] = *(const bf16x8_t*)(in + (up + (int64_t)(tap >> 1) * (2 * a.Wo) + (tap & 1)) * a.in_cs + a.in_co + c);
        }
      }
    }
    // ---- 64 output channels at a time against the LDS weights
    for (int nb = 0; nb < NS / 64; ++nb) {
      f32x4_t acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KF; ++ks) {
        bf16x8_t wf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8_t*)(sW + ((nb * 4 + i) * KF + ks) * 1024 + frag_off);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mma_slab(wf[i], pf[j][ks], acc[i][j]);
      }
      const int nl = nb * 64 + 16 * ch;                         // this lane's 16 consecutive channels, inside the slice
      const int n = nbase + nl;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!live[j]) continue;
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) v[4 * i + q] = acc[i][j][q];
        int64_t o;
        if (MODE == G_1x1) {
          const int g = n / a.shufC, co = n - g * a.shufC;
#pragma unroll
          for (int q = 0; q < 16; ++q) v[q] += sBias[nl + q];
          o = (opix[j] + (int64_t)(g >> 1) * (2 * a.Wo) + (g & 1)) * a.out_cs + a.out_co + co;
        } else {
          o = opix[j] * a.out_cs + a.out_co + n;
          if (mask != nullptr) {
            const bf16x8_t m0 = *(const bf16x8_t*)(mask + opix[j] * a.mask_cs + a.mask_co + n);
            const bf16x8_t m1 = *(const bf16x8_t*)(mask + opix[j] * a.mask_cs + a.mask_co + n + 8);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              v[q] = (float)m0[q] > 0.f ? v[q] : 0.f;
              v[8 + q] = (float)m1[q] > 0.f ? v[8 + q] : 0.f;
            }
          }
        }
        bf16x8_t o0, o1;
#pragma unroll
        for (int q = 0; q < 8; ++q) { o0[q] = (__bf16)v[q]; o1[q] = (__bf16)v[8 + q]; }
        *(bf16x8_t*)(out + o) = o0;
        *(bf16x8_t*)(out + o + 8) = o1;
      }
    }
  }
}

// weights + bias can exceed the 64 KB a kernel may take as dynamic LDS by default
template <int KF, int MODE>
int launch_one(const IgemmArgs& a, dim3 grid, size_t shm, hipStream_t s) {
  static std::atomic<bool> raised[64];                        // per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(MI_EINVAL, "tconv: bad device");
  if (!raised[dev].load()) {
    int rc = check_hip(hipFuncSetAttribute((const void*)tconv_kernel<KF, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kTconvMaxLds + 4096), "tconv LDS limit");
    if (rc) return rc;
    raised[dev].store(true);
  }
  hipLaunchKernelGGL((tconv_kernel<KF, MODE>), grid, dim3(kTconvThreads), shm, s, a);
  return check_hip(hipGetLastError(), "tconv launch");
}
template <int MODE>
int launch_kf(const IgemmArgs& a, int KF, dim3 grid, size_t shm, hipStream_t s) {
  switch (KF) {
    case 2: return launch_one<2, MODE>(a, grid, shm, s);
    case 4: return launch_one<4, MODE>(a, grid, shm, s);
    case 8: return launch_one<8, MODE>(a, grid, shm, s);
  }
  return fail(MI_EINVAL, "tconv: unsupported reduction length");
}

}  // namespace

// Channel slices per pixel block: 1 when the pixel blocks alone fill the chip and the whole weight matrix
// fits in 64 KB of LDS; otherwise the smallest power-of-two cut that gives 256 workgroups (and a slice
// that fits).  0 = the layer does not suit this kernel.
static int tconv_slices(const IgemmArgs& a) {
  const int K = a.taps * a.Cin;
  const int nblk = (a.M + kTconvBlockPix - 1) / kTconvBlockPix, n64 = a.N / 64;
  int ns = 1;
  while (ns < n64 && (nblk * ns < 256 || (int64_t)(a.N / ns) * K * 2 > kTconvMaxLds)) ns *= 2;
  if (n64 % ns != 0 || (int64_t)(a.N / ns) * K * 2 > kTconvMaxLds) return 0;
  if (nblk * ns < 128) return 0;        // too few workgroups either way: the generic kernel's smaller tiles win
  return ns;
}

// The shapes this kernel takes: bf16 transposed-conv forward / input gradient with K = 64, 128 or 256
// whose weight matrix, cut into channel slices of at most 64 KB, still gives the chip enough workgroups.
bool tconv_eligible(const IgemmArgs& a) {
  const bool fwd = a.mode == G_1x1 && a.epi == E_SHUFFLE_BIAS && a.taps == 1;
  const bool bwd = a.mode == G_2x2S2 && a.epi == E_PLAIN && a.taps == 4;
  if (!fwd && !bwd) return false;
  const int K = a.taps * a.Cin;
  if (a.Cin % 32 != 0 || (K != 64 && K != 128 && K != 256) || a.N % 64 != 0 || a.Kp < K) return false;
  if (fwd && a.shufC % 16 != 0) return false;
  // the input gradient gains nothing from channel slices (its pixel fragments, 4 taps each, are the bulk of the reads)
  return bwd ? tconv_slices(a) == 1 : tconv_slices(a) > 0;
}

int launch_tconv_bf16(const IgemmArgs& a, hipStream_t s, const char** name) {
  const int K = a.taps * a.Cin, KF = K / 32;
  const bool fwd = a.mode == G_1x1;
  const int ns = tconv_slices(a);
  if (ns <= 0) return fail(MI_EINVAL, "tconv: shape not eligible");
  const size_t shm = (size_t)(a.N / ns) * K * 2 + (fwd ? (size_t)(a.N / ns) * 4 : 0);
  const int nblk = (a.M + kTconvBlockPix - 1) / kTconvBlockPix;
  const dim3 grid(nblk < 512 ? nblk : 512, ns);
  if (name) *name = fwd ? "tconv_fwd" : "tconv_dgrad";
  return fwd ? launch_kf<G_1x1>(a, KF, grid, shm, s) : launch_kf<G_2x2S2>(a, KF, grid, shm, s);
}

}  // namespace mi
