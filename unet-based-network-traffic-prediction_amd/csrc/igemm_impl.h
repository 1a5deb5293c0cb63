// Implicit-GEMM convolution kernel for gfx950 (CDNA4), shared by the 3x3 conv forward
// and dgrad, the 2x2-stride-2 transposed conv forward and dgrad (models/unet2d.py:28-31,
// :46-48 of the reference and their autograd backward).
//
//   out[p][n] = epi( sum_{tap,c} in[gather(p,tap)][c] * w[n][tap*Cin + c] )
//
// GEMM view: M = pixels, N = output channels, K = taps*Cin.  The MFMA "A" operand is the
// weight tile (rows = n), the "B" operand the pixel tile (cols = p), so each lane's
// accumulator holds 4 consecutive channels of one pixel and the NHWC store is a vector
// store.  Both LDS tiles are [row][64 B of K]: a 64-byte slab is 32 bf16 or 16 fp32
// k-elements, and the same byte geometry serves both dtypes:
//   bf16: one v_mfma_f32_16x16x32_bf16 per (16x16 tile, slab)
//   fp32: four v_mfma_f32_16x16x4_f32 (element j of each lane's 4 floats feeds MFMA j; any
//         k permutation is valid as long as both operands use it)
// Slab rows are XOR-swizzled (chunk ^= 2*((row>>2)&1)) so the ds_read_b128 fragment reads
// of 16 consecutive rows are bank-conflict-free for any starting row.
#pragma once
#include "common.h"

namespace mi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
// staging registers are native vectors: an aggregate (HIP's uint4 struct) copied out of global
// memory becomes a memcpy into a private array and drags the array into scratch
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  typedef f32x4_t frag_t;
};
template <> struct ElemTraits<__bf16> {
  static constexpr int EPC = 8;
  typedef bf16x8_t frag_t;
};

__device__ __forceinline__ f32x4_t mma_slab(const f32x4_t& w, const f32x4_t& p, f32x4_t acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], p[j], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x4_t mma_slab(const bf16x8_t& w, const bf16x8_t& p, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, p, acc, 0, 0, 0);
}

__device__ __forceinline__ f32x4_t load4f(const float* p) { return *(const f32x4_t*)p; }
__device__ __forceinline__ f32x4_t load4f(const __bf16* p) {
  bf16x4_t v = *(const bf16x4_t*)p;
  f32x4_t r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  return r;
}
__device__ __forceinline__ void store4f(float* p, const f32x4_t& v) { *(f32x4_t*)p = v; }
__device__ __forceinline__ void store4f(__bf16* p, const f32x4_t& v) {
  bf16x4_t r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *(bf16x4_t*)p = r;
}

// 4 waves arranged WM (pixels) x WN (channels); each wave owns TM x TN tiles of 16x16.
template <typename T, int WM, int WN, int TM, int TN, int KS, int MODE>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr int EPC = ElemTraits<T>::EPC;
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int CPS = KS * 4;            // 16-byte chunks per row per stage
  constexpr int RPP = 256 / CPS;         // rows covered by one pass of the 256 threads
  constexpr int A_IT = BM / RPP;
  constexpr int B_IT = (BN + RPP - 1) / RPP;
  constexpr int A_BYTES = KS * BM * 64, B_BYTES = KS * BN * 64, STAGE = A_BYTES + B_BYTES;
  static_assert(BM % RPP == 0, "pixel tile must be a whole number of loader passes");
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int sub = t % CPS, row0 = t / CPS;
  const int p0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const T* __restrict__ in = (const T*)a.in;
  const T* __restrict__ wp = (const T*)a.w;
  const int HW = a.Ho * a.Wo;
  const int64_t zoff_in = ((intptr_t)a.zero - (intptr_t)a.in) / (int64_t)sizeof(T);  // zero page, in elements from `in`

  // Per-row gather state of this thread's A_IT pixel rows.
  int64_t base[A_IT];
  int yy[A_IT], xx[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int p = p0 + row0 + i * RPP;
    if (p < a.M) {
      const uint32_t b = fdiv((uint32_t)p, a.divHW);
      const uint32_t r = (uint32_t)p - b * (uint32_t)HW;
      const uint32_t y = fdiv(r, a.divW);
      const uint32_t x = r - y * (uint32_t)a.Wo;
      yy[i] = (int)y;
      xx[i] = (int)x;
      if (MODE == G_2x2S2)
        base[i] = ((int64_t)(b * a.Hi + 2 * y) * a.Wi + 2 * x) * a.in_cs + a.in_co;
      else
        base[i] = (int64_t)p * a.in_cs + a.in_co;
    } else {
      yy[i] = -(1 << 20);
      xx[i] = -(1 << 20);
      base[i] = -1;
    }
  }
  // K position of this thread's chunk: (tap, chunk-in-tap), advanced by CPS chunks per stage.
  const int cpt = a.Cin / EPC;
  int tap = sub / cpt, cc = sub % cpt;

  u32x4_t ra[A_IT], rb[B_IT];
  auto load_stage = [&](int kstage) {
    int64_t toff = 0;
    int dy = 0, dx = 0;
    if (MODE == G_3x3) {
      const int ty = (tap * 11) >> 5;
      dy = ty - 1;
      dx = tap - ty * 3 - 1;
      toff = (int64_t)(dy * a.Wi + dx) * a.in_cs;
    } else if (MODE == G_2x2S2) {
      toff = (int64_t)((tap >> 1) * a.Wi + (tap & 1)) * a.in_cs;
    }
    toff += cc * EPC;
    const bool tap_ok = tap < a.taps;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      bool ok = tap_ok;
      if (MODE == G_3x3)
        ok = ok && ((unsigned)(yy[i] + dy) < (unsigned)a.Hi) && ((unsigned)(xx[i] + dx) < (unsigned)a.Wi);
      else
        ok = ok && (base[i] >= 0);
      // padding lanes read the zero page: address select before the load, no use of the data
      // until the LDS store, so all staging loads stay in flight across the MFMA phase
      // (integer select of the OFFSET from one base pointer: a select between two pointers
      // makes the optimiser split the load into branches again)
      const int64_t off = ok ? base[i] + toff : zoff_in;
      ra[i] = *(const u32x4_t*)(in + opaque_offset(off));
    }
    const int64_t kel = (int64_t)(kstage * CPS + sub) * EPC;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = row0 + i * RPP;
      rb[i] = *(const u32x4_t*)(wp + (int64_t)(n0 + (row < BN ? row : 0)) * a.Kp + kel);
    }
  };
  auto advance = [&]() {
    cc += CPS;
    while (cc >= cpt) {
      cc -= cpt;
      ++tap;
    }
  };
  auto store_stage = [&](int buf) {
    char* A = smem + buf * STAGE;
    char* Bt = A + A_BYTES;
    const int s = sub >> 2, c = sub & 3;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = row0 + i * RPP;
      *(u32x4_t*)(A + s * (BM * 64) + row * 64 + ((c ^ (((row >> 2) & 1) << 1)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = row0 + i * RPP;
      if (row < BN) *(u32x4_t*)(Bt + s * (BN * 64) + row * 64 + ((c ^ (((row >> 2) & 1) << 1)) << 4)) = rb[i];
    }
  };

  f32x4_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, ch = lane >> 4;
  const int frag_off = r16 * 64 + ((ch ^ (((r16 >> 2) & 1) << 1)) << 4);
  auto compute = [&](int buf) {
    const char* A = smem + buf * STAGE;
    const char* Bt = A + A_BYTES;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      frag_t pf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) pf[j] = *(const frag_t*)(A + s * (BM * 64) + (wm * TM + j) * 1024 + frag_off);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(Bt + s * (BN * 64) + (wn * TN + i) * 1024 + frag_off);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma_slab(wf[i], pf[j], acc[i][j]);
    }
  };

  // Bias of this lane's 4 channels per channel tile, fetched HERE, before the K loop, through a buffer resource that
  // ends at the last real channel (loads beyond it return zero; no bias: an empty resource).  Loaded in the epilogue
  // with a clamped index and a select per value, every load was followed by its own wait: eight round trips at the end
  // of a launch of ~10 us.
  f32x4_t biasv[TN];
  {
    const bool has_bias = a.epi != E_PLAIN;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_bias ? a.bias : (const float*)a.zero), 0, has_bias ? a.n_real * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + (wn * TN + i) * 16 + 4 * (lane >> 4);
      const int c0 = a.epi == E_SHUFFLE_BIAS ? n % a.shufC : n;
#pragma unroll
      for (int q = 0; q < 4; ++q) biasv[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (c0 + q) * 4, 0, 0));
    }
  }

  const int nk = a.Kp / (CPS * EPC);
  load_stage(0);
  store_stage(0);
  advance();
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) load_stage(ks + 1);
    compute(ks & 1);
    if (more) {
      store_stage((ks + 1) & 1);
      advance();
    }
    __syncthreads();
  }

  // Epilogue: lane holds channels n..n+3 (n = tile*16 + 4*(lane>>4)) of pixel (lane&15).
  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
  // ReLU masks of all this lane's outputs, fetched in one batch (one round trip, not one per tile)
  f32x4_t maskv[TM][TN];
  if (mask != nullptr) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int p = p0 + (wm * TM + j) * 16 + r16;
      const int pc = p < a.M ? p : a.M - 1;
#pragma unroll
      for (int i = 0; i < TN; ++i)
        maskv[j][i] = load4f(mask + (int64_t)pc * a.mask_cs + a.mask_co + n0 + (wn * TN + i) * 16 + 4 * ch);
    }
  }
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int p = p0 + (wm * TM + j) * 16 + r16;
    if (p >= a.M) continue;
    int64_t opix = p;
    if (a.epi == E_SHUFFLE_BIAS) {
      const uint32_t b = fdiv((uint32_t)p, a.divHW);
      const uint32_t r = (uint32_t)p - b * (uint32_t)HW;
      const uint32_t y = fdiv(r, a.divW);
      const uint32_t x = r - y * (uint32_t)a.Wo;
      opix = ((int64_t)(b * a.Ho + y) * 2) * (2 * a.Wo) + 2 * x;  // pixel (2y, 2x) of the 2Ho x 2Wo image
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + (wn * TN + i) * 16 + 4 * ch;
      f32x4_t v = acc[i][j];
      int64_t o;
      if (a.epi == E_SHUFFLE_BIAS) {
        const int g = n / a.shufC, co = n - g * a.shufC;
        v += biasv[i];
        o = (opix + (int64_t)(g >> 1) * (2 * a.Wo) + (g & 1)) * a.out_cs + a.out_co + co;
      } else {
        if (a.epi != E_PLAIN) {
          v += biasv[i];
          if (a.epi == E_BIAS_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
          }
        }
        o = opix * a.out_cs + a.out_co + n;
      }
      if (mask != nullptr) {
        const f32x4_t mv = maskv[j][i];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = mv[q] > 0.f ? v[q] : 0.f;
      }
      store4f(out + o, v);
    }
  }
}

template <typename T, int WM, int WN, int TM, int TN, int MODE>
static int launch_cfg(const IgemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  dim3 grid((a.M + BM - 1) / BM, a.N / BN);
  // A K loop iteration is one memory round trip: the transposed convs of the deep levels (64x64 tiles,
  // K of 512 and up) take 128 k per iteration instead of 64, half as many round trips.
  constexpr bool deep = MODE != G_3x3 && BM == 64 && BN == 64 && sizeof(T) == 2;
  if (deep && a.Kp % 128 == 0 && a.taps * a.Cin >= 512)
    hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, deep ? 4 : 2, MODE>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((igemm_kernel<T, WM, WN, TM, TN, 2, MODE>), grid, dim3(256), 0, s, a);
  return check_hip(hipGetLastError(), "igemm launch");
}

template <typename T, int MODE>
static int launch_mode(const IgemmArgs& a, hipStream_t s, const char** name) {
  // Channel tile: the widest of 64/32/16 dividing N.  Pixel tile: 256 unless that leaves
  // the 256 CUs short of workgroups, then 64.
  const int bn = (a.N % 64 == 0) ? 64 : (a.N % 32 == 0) ? 32 : 16;
  const int64_t wgs256 = (int64_t)((a.M + 255) / 256) * (a.N / bn);
  const bool big = wgs256 >= 384;
  if (bn == 64) {
    if (big) { if (name) *name = "igemm_256x64"; return launch_cfg<T, 4, 1, 4, 4, MODE>(a, s); }
    if (name) *name = "igemm_64x64";
    return launch_cfg<T, 2, 2, 2, 2, MODE>(a, s);
  }
  if (bn == 32) {
    if (big) { if (name) *name = "igemm_256x32"; return launch_cfg<T, 4, 1, 4, 2, MODE>(a, s); }
    if (name) *name = "igemm_64x32";
    return launch_cfg<T, 4, 1, 1, 2, MODE>(a, s);
  }
  if (big) { if (name) *name = "igemm_256x16"; return launch_cfg<T, 4, 1, 4, 1, MODE>(a, s); }
  if (name) *name = "igemm_64x16";
  return launch_cfg<T, 4, 1, 1, 1, MODE>(a, s);
}

template <typename T>
static int launch_igemm_t(const IgemmArgs& a, hipStream_t s, const char** name) {
  switch (a.mode) {
    case G_3x3: return launch_mode<T, G_3x3>(a, s, name);
    case G_2x2S2: return launch_mode<T, G_2x2S2>(a, s, name);
    case G_1x1: return launch_mode<T, G_1x1>(a, s, name);
  }
  return fail(MI_EINVAL, "igemm: bad gather mode");
}

}  // namespace mi
