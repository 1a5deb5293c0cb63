// Transposed conv 2x2 stride 2 (models/unet2d.py:46-48 of the reference), forward and input gradient,
// bf16 and fp32, for the levels the LDS-resident kernel of tconv_bf16.hip does not take: few pixels against a
// long reduction (K = Cin forward, 4*Cout backward: 128 ... 2048).
//
// Both directions are plain GEMMs whose operands are each used by few waves, so nothing is staged in
// LDS: every wave loads its MFMA fragments of weights and pixels STRAIGHT from global memory into
// registers (a lane's 16 bytes are 8 consecutive k of one weight row / one pixel, which is exactly the
// operand layout of v_mfma_f32_16x16x32_bf16; in fp32 4 consecutive k, the operands of four
// v_mfma_f32_16x16x4_f32), a whole batch of K slabs in flight before the first MFMA
// waits.  A generic double-buffered tile loop pays one memory round trip per 64 k; here a layer is one
// or two round trips.  The KW waves of a workgroup split the K range and are summed through LDS in wave
// order (fixed order: bitwise repeatable).
//   forward : out[b][2y+a][2x+c][co] = bias[co] + sum_ci x[b][y][x][ci] * W[ci][co][a][c]   (pixel-shuffle store)
//   backward: dx[b][y][x][ci] = [prev > 0] * sum_{a,c,co} dy[b][2y+a][2x+c][co] * W[ci][co][a][c]
// Output channels are permuted inside each block of 64 (row r of MFMA tile t is channel
// 16*(r>>2) + 4*t + (r&3)) so that a lane ends up with 16 CONSECUTIVE channels of its pixel: two
// 16-byte stores per pixel instead of four 8-byte ones.
#include "igemm_impl.h"

namespace mi {
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int ru32x4_t;
constexpr unsigned kROob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rg_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// lanes with an out-of-range offset read zeros and touch no memory
template <typename T> __device__ __forceinline__ typename ElemTraits<T>::frag_t rg_frag(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const ru32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return __builtin_bit_cast(typename ElemTraits<T>::frag_t, v);
}

struct RgemmArgs {
  IgemmArgs g;
  int slabs;        // K / (32 in bf16, 16 in fp32): 64-byte pieces of a row
  int spw;          // slabs per wave
};

// One workgroup = one tile of 16*PF pixels x 64 channels; KW waves split K; U slabs per load batch.
template <typename T, int MODE, int PF, int KW, int U>
__global__ __launch_bounds__(64 * KW) void rgemm_kernel(const RgemmArgs ra) {
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr unsigned EPC = ElemTraits<T>::EPC, ESZ = sizeof(T);
  constexpr int UC = 4 * EPC;            // k per slab
  constexpr int TN = 4;
  __shared__ float red[KW > 1 ? KW - 1 : 1][TN * PF][4][64];
  const IgemmArgs& a = ra.g;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, ch = lane >> 4;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * (16 * PF);
  const int HW = a.Ho * a.Wo;

  unsigned wbase[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n0 + 16 * (r16 >> 2) + 4 * t + (r16 & 3);
    wbase[t] = ((unsigned)n * (unsigned)a.Kp + (unsigned)ch * EPC) * ESZ;
  }
  const __amdgpu_buffer_rsrc_t rin = rg_rsrc(a.in), rw = rg_rsrc(a.w);

  const int s0 = wave * ra.spw;
  const int s1 = s0 + ra.spw < ra.slabs ? s0 + ra.spw : ra.slabs;

  // the weights of the first batch are asked for before the pixel geometry is worked out: they are the
  // operand that comes from far away (read once per layer; the few pixels are read by every workgroup)
  frag_t wf[U][TN];
  auto load_weights = [&](int s) {
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[i][t] = rg_frag<T>(rw, s + i < s1 ? wbase[t] + (unsigned)((s + i) * UC) * ESZ : kROob);
  };
  load_weights(s0);
  __builtin_amdgcn_sched_barrier(0);

  // this lane's pixel of each fragment: operand byte offset, and the pixel (2y, 2x) of the 2Ho x 2Wo image
  unsigned pbase[PF];
  int64_t up[PF];
  bool live[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int p = m0 + j * 16 + r16;
    live[j] = p < a.M;
    const uint32_t pc = live[j] ? (uint32_t)p : 0u;
    const uint32_t b = fdiv(pc, a.divHW);
    const uint32_t r = pc - b * (uint32_t)HW;
    const uint32_t y = fdiv(r, a.divW);
    const uint32_t x = r - y * (uint32_t)a.Wo;
    up[j] = ((int64_t)(b * a.Ho + y) * 2) * (2 * a.Wo) + 2 * x;
    const uint32_t src_pix = MODE == G_1x1 ? pc : (uint32_t)up[j];
    pbase[j] = (src_pix * (uint32_t)a.in_cs + (uint32_t)a.in_co + (uint32_t)ch * EPC) * ESZ;
  }

  f32x4_t acc[TN * PF];
#pragma unroll
  for (int j = 0; j < TN * PF; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int s = s0; s < s1; s += U) {
    frag_t pf[U][PF];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool on = s + i < s1;
      const int k = (s + i) * UC;
      unsigned poff;
      if (MODE == G_1x1) {
        poff = (unsigned)k * ESZ;
      } else {
        // a slab never straddles a tap: Cin is a multiple of 32
        const int tap = k / a.Cin, c = k - tap * a.Cin;
        poff = (unsigned)(((tap >> 1) * (2 * a.Wo) + (tap & 1)) * a.in_cs + c) * ESZ;
      }
#pragma unroll
      for (int j = 0; j < PF; ++j) pf[i][j] = rg_frag<T>(rin, (on && live[j]) ? pbase[j] + poff : kROob);
    }
    __builtin_amdgcn_sched_barrier(0);      // every load of the batch is issued before the first MFMA waits
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc[t * PF + j] = mma_slab(wf[i][t], pf[i][j], acc[t * PF + j]);
    if (s + U < s1) {
      __builtin_amdgcn_sched_barrier(0);
      load_weights(s + U);
    }
  }

  if (KW > 1) {     // the K waves combine in wave order
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < TN * PF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][j][q][lane] = acc[j][q];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 2
    for (int k = 0; k < KW - 1; ++k)
#pragma unroll
      for (int j = 0; j < TN * PF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] += red[k][j][q][lane];
  }

  // this lane: channels n .. n+15 (n = n0 + 16*(lane>>4)) of pixel (lane&15) of each pixel fragment
  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
  const int n = n0 + 16 * ch;
  float biasv[16];
  int g = 0, co = n;
  if (MODE == G_1x1) {
    g = n / a.shufC; co = n - g * a.shufC;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float b = a.bias[co + q < a.n_real ? co + q : 0];
      biasv[q] = co + q < a.n_real ? b : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    if (!live[j]) continue;
    const int p = m0 + j * 16 + r16;
    float v[16];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[4 * t + q] = acc[t * PF + j][q];
    int64_t o;
    if (MODE == G_1x1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] += biasv[q];
      o = (up[j] + (int64_t)(g >> 1) * (2 * a.Wo) + (g & 1)) * a.out_cs + a.out_co + co;
    } else {
      o = (int64_t)p * a.out_cs + a.out_co + n;
      if (mask != nullptr) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4_t mv = load4f(mask + (int64_t)p * a.mask_cs + a.mask_co + n + 4 * q4);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[4 * q4 + q] = mv[q] > 0.f ? v[4 * q4 + q] : 0.f;
        }
      }
    }
    if constexpr (sizeof(T) == 2) {
      bf16x8_t o0, o1;
#pragma unroll
      for (int q = 0; q < 8; ++q) { o0[q] = (__bf16)v[q]; o1[q] = (__bf16)v[8 + q]; }
      *(bf16x8_t*)(out + o) = o0;
      *(bf16x8_t*)(out + o + 8) = o1;
    } else {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) *(f32x4_t*)(out + o + 4 * q4) = f32x4_t{v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]};
    }
  }
}

template <typename T, int MODE, int PF>
int launch_pf(const RgemmArgs& ra, int kw, dim3 grid, hipStream_t s) {
  switch (kw) {
    case 1: hipLaunchKernelGGL((rgemm_kernel<T, MODE, PF, 1, 4>), grid, dim3(64), 0, s, ra); break;
    case 2: hipLaunchKernelGGL((rgemm_kernel<T, MODE, PF, 2, 4>), grid, dim3(128), 0, s, ra); break;
    case 4: hipLaunchKernelGGL((rgemm_kernel<T, MODE, PF, 4, 4>), grid, dim3(256), 0, s, ra); break;
    default: hipLaunchKernelGGL((rgemm_kernel<T, MODE, PF, 8, 4>), grid, dim3(512), 0, s, ra); break;
  }
  return check_hip(hipGetLastError(), "rgemm launch");
}
template <typename T>
int launch_t(const RgemmArgs& ra, bool fwd, int pf, int kw, dim3 grid, hipStream_t s) {
  if (fwd) return pf == 1 ? launch_pf<T, G_1x1, 1>(ra, kw, grid, s) : launch_pf<T, G_1x1, 2>(ra, kw, grid, s);
  return pf == 1 ? launch_pf<T, G_2x2S2, 1>(ra, kw, grid, s) : launch_pf<T, G_2x2S2, 2>(ra, kw, grid, s);
}

}  // namespace

// The shapes this kernel takes: bf16 transposed-conv forward / input gradient with 32-channel-aligned
// operands and byte offsets that fit 31 bits.
bool rgemm_eligible(const IgemmArgs& a, int dtype) {
  const int esz = dtype == MI_BF16 ? 2 : 4;
  const bool fwd = a.mode == G_1x1 && a.epi == E_SHUFFLE_BIAS && a.taps == 1;
  const bool bwd = a.mode == G_2x2S2 && a.epi == E_PLAIN && a.taps == 4;
  if (!fwd && !bwd) return false;
  const int K = a.taps * a.Cin;
  if (a.Cin % 32 != 0 || a.N % 64 != 0 || a.Kp < K || K < 64) return false;
  if (fwd && a.shufC % 16 != 0) return false;
  const int64_t in_pix = bwd ? (int64_t)a.B * a.Hi * a.Wi : (int64_t)a.M;
  if (in_pix * a.in_cs * esz >= (1ll << 31) || (int64_t)a.N * a.Kp * esz >= (1ll << 31)) return false;
  return true;
}

int launch_rgemm(const IgemmArgs& g, int dtype, hipStream_t s, const char** name) {
  RgemmArgs ra{};
  ra.g = g;
  ra.slabs = g.taps * g.Cin / (dtype == MI_BF16 ? 32 : 16);
  // K waves: about four slabs each (one load batch), at most eight waves
  int kw = 1;
  while (kw < 8 && ra.slabs / (2 * kw) >= 4) kw *= 2;
  ra.spw = (ra.slabs + kw - 1) / kw;
  const bool fwd = g.mode == G_1x1;
  // pixel tile: 16 while that gives at most 256 workgroups (their number matters most: a CU has only so many
  // loads in flight), else 32
  const int pf = (int64_t)(g.N / 64) * ((g.M + 15) / 16) <= 256 ? 1 : 2;
  const dim3 grid(g.N / 64, (g.M + 16 * pf - 1) / (16 * pf));
  static const char* const names[2][4] = {{"rgemm_fwd_k1", "rgemm_fwd_k2", "rgemm_fwd_k4", "rgemm_fwd_k8"},
                                          {"rgemm_dgrad_k1", "rgemm_dgrad_k2", "rgemm_dgrad_k4", "rgemm_dgrad_k8"}};
  if (name) *name = names[fwd ? 0 : 1][kw == 1 ? 0 : kw == 2 ? 1 : kw == 4 ? 2 : 3];
  return dtype == MI_BF16 ? launch_t<__bf16>(ra, fwd, pf, kw, grid, s) : launch_t<float>(ra, fwd, pf, kw, grid, s);
}

}  // namespace mi
