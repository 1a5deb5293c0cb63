// Weight-gradient kernel for gfx950: the autograd backward of the reference's Conv2d /
// ConvTranspose2d layers (models/unet2d.py:28-31, :46-48) with respect to their weights.
//
//   part[z][n][k] = sum_{p in pixel range z} a[p][n] * b[gather(p, tap)][c],  k = tap*Cb + c
//
// GEMM view: the reduction runs over PIXELS, the slow index of both NHWC operands, so
// the LDS tiles stay [pixel][64 channels] as loaded and the MFMA fragments are read
// transposed: ds_read_b64_tr_b16 for bf16 (4 pixels x 16 channels per 16-lane group),
// plain ds_read_b32 for fp32 (v_mfma_f32_16x16x4_f32 wants one float per lane).
// One workgroup owns a 64(n) x 64(k) output tile for one pixel range; ranges are
// combined afterwards in a fixed order (bitwise reproducible, no atomics).
// On images of a few pixels most (pixel, tap) pairs of a 3x3 conv fall on the zero padding
// (20 of 36 on a 2x2 image).  Mode G_3x3P therefore reduces over a per-tap list instead: a
// k-tile lies inside one tap (Cb % 64 == 0), and its workgroups walk only the in-bounds
// positions of that tap, image by image (neighbouring pixels stay neighbours in memory, so the
// rows of a stage spread over all memory channels) -- no padding rows are loaded or multiplied.
// Row sums of a (bias gradient of a conv) and column sums of b (bias gradient of a
// transposed conv) come from extra MFMAs against an all-ones fragment.
#pragma once
#include <type_traits>
#include "igemm_impl.h"

namespace mi {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <typename T> struct WgTraits;
template <> struct WgTraits<__bf16> {
  static constexpr int PB = 128;   // pixels per stage
  static constexpr int ROWB = 128; // bytes per LDS row (64 channels)
};
template <> struct WgTraits<float> {
  static constexpr int PB = 64;
  static constexpr int ROWB = 256;
};

// Byte offset of 16-byte chunk q of row r inside a [rows][64 ch] tile.
// bf16 (128-B rows): the 32-byte column group is XORed with bits 1 and 3 of the row, which
// spreads the 8 rows {r..r+3, r+8..r+11} one transposed read touches over all 64 banks.
// fp32 (256-B rows): the 64-byte column group is XORed with bit 0 of the row (two
// consecutive rows per ds_read_b32 half-wave).
template <typename T> __device__ __forceinline__ int tile_off(int r, int q);
template <> __device__ __forceinline__ int tile_off<__bf16>(int r, int q) {
  const int f = ((r >> 1) & 1) | (((r >> 3) & 1) << 1);
  return r * 128 + (((((q >> 1) ^ f) << 1) | (q & 1)) << 4);
}
template <> __device__ __forceinline__ int tile_off<float>(int r, int q) {
  return r * 256 + (((((q >> 2) ^ (r & 1)) << 2) | (q & 3)) << 4);
}
// fp32 tiles of 32 or 128 channels per row (the 32 x 128 output tile below): the same rule, the 64-byte
// column group XORed with bit 0 of the row (rows 128 bytes apart never share a bank; 512 apart always would)
template <int ROWB> __device__ __forceinline__ int tile_off_f32(int r, int q) {
  return r * ROWB + (((((q >> 2) ^ (r & 1)) << 2) | (q & 3)) << 4);
}

// Fragment of tile column block `ct` (16 channels) for the MFMA k-step `ks`.
// bf16: k-step = 32 pixels; lane (g = lane>>4, i = lane&15) ends up with channel ct*16+i of
//       pixels ks*32 + 8g + 0..7.
__device__ __forceinline__ bf16x8_t read_frag_tr(const char* tile, int ks, int ct, int lane, __bf16) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int r0 = ks * 32 + 8 * g + q;
  const int chunk = 2 * ct + (pp >> 1);
  const int sub = (pp & 1) * 8;
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + tile_off<__bf16>(r0, chunk) + sub));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + tile_off<__bf16>(r0 + 4, chunk) + sub));
  union { s16x4_t s[2]; bf16x8_t v; } u;
  u.s[0] = lo;
  u.s[1] = hi;
  return u.v;
}
// fp32: k-step = 4 pixels; lane holds channel ct*16 + (lane&15) of pixel ks*4 + (lane>>4).
__device__ __forceinline__ float read_frag_tr(const char* tile, int ks, int ct, int lane, float) {
  const int r = ks * 4 + (lane >> 4), c = ct * 16 + (lane & 15);
  return *(const float*)(tile + tile_off<float>(r, c >> 2) + (c & 3) * 4);
}
template <int ROWB> __device__ __forceinline__ float read_frag_f32(const char* tile, int ks, int ct, int lane) {
  const int r = ks * 4 + (lane >> 4), c = ct * 16 + (lane & 15);
  return *(const float*)(tile + tile_off_f32<ROWB>(r, c >> 2) + (c & 3) * 4);
}

__device__ __forceinline__ f32x4_t mma_px(const bf16x8_t& a, const bf16x8_t& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mma_px(float a, float b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ bf16x8_t ones_frag(__bf16) {
  const __bf16 o = (__bf16)1.0f;
  return bf16x8_t{o, o, o, o, o, o, o, o};
}
__device__ __forceinline__ float ones_frag(float) { return 1.0f; }

template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int EPC = ElemTraits<T>::EPC;
  constexpr int PB = WgTraits<T>::PB;
  constexpr int CPR = 64 / EPC;           // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;          // rows per loader pass
  constexpr int IT = PB / RPP;            // chunks per thread per operand per stage
  constexpr int TILE = PB * WgTraits<T>::ROWB;
  constexpr int KSTEPS = PB / (sizeof(T) == 2 ? 32 : 4);
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];  // 2 stages x (a tile, b tile)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // Workgroup ids are dealt to the 8 XCDs (id % 8) in units that read the same pixels -- a whole
  // pixel split, or one n-block of it when the split has more than 32 tiles -- so the operand tiles a
  // unit shares come from one L2 instead of being fetched by every XCD.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / a.usize) * 8 + xcd, mem = slot % a.usize;
  if (unit >= a.units) return;                 // a hole of the deal: 8 does not divide the unit count
  int bx, by, z;
  if (a.usize == a.gx * a.gy) { z = unit; by = mem / a.gx; bx = mem - by * a.gx; }
  else { z = unit / a.gx; bx = unit - z * a.gx; by = mem; }
  const int n0 = bx * 64, k0 = by * 64;
  const int K = a.taps * a.Cb;
  const T* __restrict__ pa = (const T*)a.a;
  const T* __restrict__ pb = (const T*)a.b;
  const int HW = a.Ho * a.Wo;
  // zero page as element offsets from each operand's base (integer offset select, one base pointer)
  const int64_t zoff_a = ((intptr_t)a.zero - (intptr_t)a.a) / (int64_t)sizeof(T);
  const int64_t zoff_b = ((intptr_t)a.zero - (intptr_t)a.b) / (int64_t)sizeof(T);
  int pbeg = z * a.pix_per_split;
  int pend = min(a.M, pbeg + a.pix_per_split);
  // G_3x3P: the reduction index e runs over (image, valid position); [pbeg,pend) is a range of e
  int vy0 = 0, vx0 = 0, nvx = 1, nval = 1;
  uint32_t mul_nv = 0, mul_nx = 0;
  if (MODE == G_3x3P) {
    const int tap = k0 / a.Cb;
    const int ty = tap / 3, ddy = ty - 1, ddx = tap - ty * 3 - 1;
    vy0 = ddy < 0 ? 1 : 0;
    vx0 = ddx < 0 ? 1 : 0;
    nvx = a.Wb - (ddx != 0 ? 1 : 0);
    const int nvy = a.Hb - (ddy != 0 ? 1 : 0);
    pend = min(nvy * nvx * a.B, pend);
    // exact reciprocal multipliers (nval <= 16, e < 2^27, position < 16)
    nval = max(1, nvy * nvx);
    nvx = max(1, nvx);
    mul_nv = 0xFFFFFFFFu / (uint32_t)nval + 1u;
    mul_nx = 256u / (uint32_t)nvx + 1u;
  }
  const int cch = t % CPR, row0 = t / CPR;
  // a-operand chunk: channels n0 + cch*EPC ..
  const int an = n0 + cch * EPC;
  const bool a_ok = an < a.Na;
  // b-operand chunk: k = k0 + cch*EPC -> (tap, c), fixed for the whole kernel
  const int bk = k0 + cch * EPC;
  const int btap = bk / a.Cb, bc = bk - btap * a.Cb;
  const bool b_ok = bk < K;
  int dy = 0, dx = 0;
  int64_t toff = 0;
  if (MODE == G_3x3 || MODE == G_3x3P) {
    const int ty = btap / 3;
    dy = ty - 1;
    dx = btap - ty * 3 - 1;
    toff = (int64_t)(dy * a.Wb + dx) * a.b_cs;
  } else if (MODE == G_2x2S2) {
    toff = (int64_t)((btap >> 1) * a.Wb + (btap & 1)) * a.b_cs;
  }
  toff += a.b_co + bc;

  u32x4_t ra[IT], rb[IT];
  auto load_stage = [&](int pstage) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      // Loads are unconditional; padding elements read the zero page (address select before the
      // load), so no instruction touches the data until the LDS store after the MFMA phase.
      int p = pstage + row0 + i * RPP;
      const bool pv = p < pend;
      if (MODE == G_3x3P) {
        // e -> (image, position index) -> pixel; every listed pixel has its tap neighbour in bounds
        const uint32_t e = pv ? (uint32_t)p : 0u;
        const uint32_t n = nval == 1 ? e : __umulhi(e, mul_nv);   // (the multiplier of 1 does not fit 32 bits)
        const uint32_t pi = e - n * (uint32_t)nval;
        const uint32_t iy = (pi * mul_nx) >> 8, ix = pi - iy * (uint32_t)nvx;
        p = (int)(n * (uint32_t)HW + (vy0 + iy) * (uint32_t)a.Wb + vx0 + ix);
      }
      const bool av = pv && a_ok;
      bool bv = pv && b_ok;
      int64_t boff = (int64_t)p * a.b_cs;
      if (MODE == G_3x3 || MODE == G_2x2S2) {
        const uint32_t pc = pv ? (uint32_t)p : 0u;
        const uint32_t b = fdiv(pc, a.divHW);
        const uint32_t r = pc - b * (uint32_t)HW;
        const uint32_t y = fdiv(r, a.divW);
        const uint32_t x = r - y * (uint32_t)a.Wo;
        if (MODE == G_3x3)
          bv = bv && (unsigned)((int)y + dy) < (unsigned)a.Hb && (unsigned)((int)x + dx) < (unsigned)a.Wb;
        else
          boff = ((int64_t)(b * a.Hb + 2 * y) * a.Wb + 2 * x) * a.b_cs;
      }
      const int64_t oa = av ? (int64_t)p * a.a_cs + a.a_co + an : zoff_a;
      const int64_t ob = bv ? boff + toff : zoff_b;
      ra[i] = *(const u32x4_t*)(pa + opaque_offset(oa));
      rb[i] = *(const u32x4_t*)(pb + opaque_offset(ob));
    }
  };
  auto store_stage = [&](int buf) {
    char* At = smem + buf * 2 * TILE;
    char* Bt = At + TILE;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int row = row0 + i * RPP;
      *(u32x4_t*)(At + tile_off<T>(row, cch)) = ra[i];
      *(u32x4_t*)(Bt + tile_off<T>(row, cch)) = rb[i];
    }
  };

  f32x4_t acc[2][2], accr[2], accc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    accr[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    accc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // row sums need every pixel: in G_3x3P only the centre tap's list has them all
  const int rows_ky = MODE == G_3x3P ? 4 * a.Cb / 64 : 0;
  const bool do_rows = a.rowsum != nullptr && by == rows_ky && wc == 0;
  const bool do_cols = a.colsum != nullptr && bx == 0 && wr == 0;
  const auto ones = ones_frag(T());

  // One copy of the k-step loop per (row sums, column sums) combination, chosen by one wave-uniform branch per stage:
  // with the two flags tested inside every k-step the loop fell into basic blocks and each step read its four
  // fragments, waited for them and only then issued its MFMAs.  Now the fragments of step ks+1 are read under the
  // MFMAs of step ks (two register sets; the order is spelled out for the scheduler).
  auto compute_t = [&](int buf, auto rows_c, auto cols_c) {
    constexpr bool ROWS = decltype(rows_c)::value, COLS = decltype(cols_c)::value;
    const char* At = smem + buf * 2 * TILE;
    const char* Bt = At + TILE;
    typedef decltype(read_frag_tr(At, 0, 0, lane, T())) frag_t;
    frag_t fa[2][2], fb[2][2];
    auto read_step = [&](int ks, int set) {
      fa[set][0] = read_frag_tr(At, ks, 2 * wr, lane, T());
      fa[set][1] = read_frag_tr(At, ks, 2 * wr + 1, lane, T());
      fb[set][0] = read_frag_tr(Bt, ks, 2 * wc, lane, T());
      fb[set][1] = read_frag_tr(Bt, ks, 2 * wc + 1, lane, T());
    };
    constexpr int RD = sizeof(T) == 2 ? 8 : 4;       // LDS read instructions per step (a transposed bf16 fragment is two)
    constexpr int NM = 4 + (ROWS ? 2 : 0) + (COLS ? 2 : 0);
    read_step(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = ks & 1;
      if (ks + 1 < KSTEPS) read_step(ks + 1, c ^ 1);
      // the b fragment goes in as the row operand: a lane then holds four CONSECUTIVE k of one n,
      // and the partials leave as 16-byte stores
      acc[0][0] = mma_px(fb[c][0], fa[c][0], acc[0][0]);
      acc[0][1] = mma_px(fb[c][1], fa[c][0], acc[0][1]);
      acc[1][0] = mma_px(fb[c][0], fa[c][1], acc[1][0]);
      acc[1][1] = mma_px(fb[c][1], fa[c][1], acc[1][1]);
      if (ROWS) {
        accr[0] = mma_px(fa[c][0], ones, accr[0]);
        accr[1] = mma_px(fa[c][1], ones, accr[1]);
      }
      if (COLS) {
        accc[0] = mma_px(ones, fb[c][0], accc[0]);
        accc[1] = mma_px(ones, fb[c][1], accc[1]);
      }
      if (ks + 1 < KSTEPS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, RD / 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (NM > 4) __builtin_amdgcn_sched_group_barrier(0x008, NM - 4, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      }
    }
  };
  auto compute = [&](int buf) {
    if (do_rows && do_cols) compute_t(buf, std::true_type{}, std::true_type{});
    else if (do_rows) compute_t(buf, std::true_type{}, std::false_type{});
    else if (do_cols) compute_t(buf, std::false_type{}, std::true_type{});
    else compute_t(buf, std::false_type{}, std::false_type{});
  };

  const int nst = (pend - pbeg + PB - 1) / PB;
  if (nst > 0) {
    load_stage(pbeg);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const bool more = st + 1 < nst;
      if (more) load_stage(pbeg + (st + 1) * PB);
      compute(st & 1);
      if (more) store_stage((st + 1) & 1);
      __syncthreads();
    }
  }

  // D layout of the products: row (= k) = 4*(lane>>4) + q, col (= n) = lane & 15; of the row/column
  // sums: row = 4*(lane>>4) + q, col = lane & 15 with every column (row) equal.
  // (K and Na are multiples of 16: a 16 x 16 fragment is inside the matrix or outside it as a whole)
  const int rq = 4 * (lane >> 4), cq = lane & 15;
  float* part = a.part + (int64_t)z * a.Na * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + (2 * wc + j) * 16 + rq;
      const int n = n0 + (2 * wr + i) * 16 + cq;
      if (n < a.Na && k < K) *(f32x4_t*)(part + (int64_t)n * K + k) = acc[i][j];
    }
  if (do_rows && cq == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + (2 * wr + i) * 16 + rq + q;
        if (n < a.Na) a.rowsum[(int64_t)z * a.Na + n] = accr[i][q];
      }
  }
  if (do_cols && rq == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + (2 * wc + j) * 16 + cq;
      if (k < K) a.colsum[(int64_t)z * K + k] = accc[j][0];
    }
  }
}

// fp32 3x3 layers with at most 32 output channels (level 0 of the U-Net): a 32(n) x 128(k) output tile.  The 64 x 64
// tile of the kernel above is half padding there -- half of its dy tile is read from the zero page and half of its
// MFMAs multiply zeros.  Same staging, same fragment reads, same fixed-order partials; the four waves sit side by
// side along k, and the two operand tiles have rows of 128 and 512 bytes.
template <typename F>      // (F = float; a template so that every translation unit may include this header)
__global__ __launch_bounds__(256, 2) void wgrad_n32_kernel(const WgradArgs a) {
  static_assert(std::is_same<F, float>::value, "fp32 only");
  constexpr int PB = 64, NT = 32, KT = 128;
  constexpr int ROWB_A = NT * 4, ROWB_B = KT * 4;
  constexpr int TILE_A = PB * ROWB_A, TILE_B = PB * ROWB_B, STAGE = TILE_A + TILE_B;
  constexpr int KSTEPS = PB / 4;
  constexpr int IT_A = PB * (NT / 4) / 256, IT_B = PB * (KT / 4) / 256;     // 16-byte chunks per thread per stage: 2 and 8
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wc = t >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int unit = (slot / a.usize) * 8 + xcd, mem = slot % a.usize;
  if (unit >= a.units) return;
  const int z = unit, by = mem;                 // one n-block: a unit is a whole pixel split
  const int k0 = by * KT;
  const int K = a.taps * a.Cb;
  const float* __restrict__ pa = (const float*)a.a;
  const float* __restrict__ pb = (const float*)a.b;
  const int HW = a.Ho * a.Wo;
  const int64_t zoff_a = ((intptr_t)a.zero - (intptr_t)a.a) / 4;
  const int64_t zoff_b = ((intptr_t)a.zero - (intptr_t)a.b) / 4;
  const int pbeg = z * a.pix_per_split;
  const int pend = min(a.M, pbeg + a.pix_per_split);

  const int cch_a = t % (NT / 4), rowa0 = t / (NT / 4);
  const int an = cch_a * 4;
  const bool a_ok = an < a.Na;
  const int cch_b = t % (KT / 4), rowb0 = t / (KT / 4);
  const int bk = k0 + cch_b * 4;
  const int btap = bk / a.Cb, bc = bk - btap * a.Cb;
  const bool b_ok = bk < K;
  const int ty = btap / 3, dy = ty - 1, dx = btap - ty * 3 - 1;
  const int64_t toff = (int64_t)(dy * a.Wb + dx) * a.b_cs + a.b_co + bc;

  u32x4_t ra[IT_A], rb[IT_B];
  auto load_stage = [&](int pstage) {
#pragma unroll
    for (int i = 0; i < IT_A; ++i) {
      const int p = pstage + rowa0 + i * (256 / (NT / 4));
      const int64_t oa = (p < pend && a_ok) ? (int64_t)p * a.a_cs + a.a_co + an : zoff_a;
      ra[i] = *(const u32x4_t*)(pa + opaque_offset(oa));
    }
#pragma unroll
    for (int i = 0; i < IT_B; ++i) {
      const int p = pstage + rowb0 + i * (256 / (KT / 4));
      const bool pv = p < pend;
      const uint32_t pc = pv ? (uint32_t)p : 0u;
      const uint32_t r = pc - fdiv(pc, a.divHW) * (uint32_t)HW;
      const uint32_t y = fdiv(r, a.divW), x = r - y * (uint32_t)a.Wo;
      const bool bv = pv && b_ok && (unsigned)((int)y + dy) < (unsigned)a.Hb && (unsigned)((int)x + dx) < (unsigned)a.Wb;
      const int64_t ob = bv ? (int64_t)p * a.b_cs + toff : zoff_b;
      rb[i] = *(const u32x4_t*)(pb + opaque_offset(ob));
    }
  };
  auto store_stage = [&](int buf) {
    char* At = smem + buf * STAGE;
    char* Bt = At + TILE_A;
#pragma unroll
    for (int i = 0; i < IT_A; ++i) *(u32x4_t*)(At + tile_off_f32<ROWB_A>(rowa0 + i * (256 / (NT / 4)), cch_a)) = ra[i];
#pragma unroll
    for (int i = 0; i < IT_B; ++i) *(u32x4_t*)(Bt + tile_off_f32<ROWB_B>(rowb0 + i * (256 / (KT / 4)), cch_b)) = rb[i];
  };

  f32x4_t acc[2][2], accr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    accr[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_rows = a.rowsum != nullptr && by == 0 && wc == 0;

  // the fragments of step ks+1 are read under the MFMAs of step ks (two register sets, order spelled out)
  auto compute_t = [&](int buf, auto rows_c) {
    constexpr bool ROWS = decltype(rows_c)::value;
    const char* At = smem + buf * STAGE;
    const char* Bt = At + TILE_A;
    float fa[2][2], fb[2][2];
    auto read_step = [&](int ks, int set) {
      fa[set][0] = read_frag_f32<ROWB_A>(At, ks, 0, lane);
      fa[set][1] = read_frag_f32<ROWB_A>(At, ks, 1, lane);
      fb[set][0] = read_frag_f32<ROWB_B>(Bt, ks, 2 * wc, lane);
      fb[set][1] = read_frag_f32<ROWB_B>(Bt, ks, 2 * wc + 1, lane);
    };
    constexpr int NM = 4 + (ROWS ? 2 : 0);
    read_step(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = ks & 1;
      if (ks + 1 < KSTEPS) read_step(ks + 1, c ^ 1);
      acc[0][0] = mma_px(fb[c][0], fa[c][0], acc[0][0]);
      acc[0][1] = mma_px(fb[c][1], fa[c][0], acc[0][1]);
      acc[1][0] = mma_px(fb[c][0], fa[c][1], acc[1][0]);
      acc[1][1] = mma_px(fb[c][1], fa[c][1], acc[1][1]);
      if (ROWS) {
        accr[0] = mma_px(fa[c][0], 1.0f, accr[0]);
        accr[1] = mma_px(fa[c][1], 1.0f, accr[1]);
      }
      if (ks + 1 < KSTEPS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (NM > 4) __builtin_amdgcn_sched_group_barrier(0x008, NM - 4, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      }
    }
  };

  const int nst = (pend - pbeg + PB - 1) / PB;
  if (nst > 0) {
    load_stage(pbeg);
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
      const bool more = st + 1 < nst;
      if (more) load_stage(pbeg + (st + 1) * PB);
      if (do_rows) compute_t(st & 1, std::true_type{}); else compute_t(st & 1, std::false_type{});
      if (more) store_stage((st + 1) & 1);
      __syncthreads();
    }
  }

  // D layout as above: row (= k) = 4*(lane>>4) + q, col (= n) = lane & 15
  const int rq = 4 * (lane >> 4), cq = lane & 15;
  float* part = a.part + (int64_t)z * a.Na * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + (2 * wc + j) * 16 + rq;
      const int n = i * 16 + cq;
      if (n < a.Na && k < K) *(f32x4_t*)(part + (int64_t)n * K + k) = acc[i][j];
    }
  if (do_rows && cq == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = i * 16 + rq + q;
        if (n < a.Na) a.rowsum[(int64_t)z * a.Na + n] = accr[i][q];
      }
  }
}

template <typename T>
static int launch_wgrad_t(const WgradArgs& a_in, hipStream_t s, const char** name) {
  WgradArgs a = a_in;
  const int K = a.taps * a.Cb;
  if constexpr (std::is_same<T, float>::value) {
    if (a.mode == G_3x3 && a.Na <= 32 && a.colsum == nullptr) {
      a.gx = 1; a.gy = (K + 127) / 128;
      a.usize = a.gy; a.units = a.splits;
      if (name) *name = "wgrad_32x128";
      hipLaunchKernelGGL(wgrad_n32_kernel<T>, dim3((a.units + 7) / 8 * 8 * a.usize), dim3(256), 0, s, a);
      return check_hip(hipGetLastError(), "wgrad launch");
    }
  }
  a.gx = (a.Na + 63) / 64; a.gy = (K + 63) / 64;
  if (a.gx * a.gy <= 32) { a.usize = a.gx * a.gy; a.units = a.splits; }
  else { a.usize = a.gy; a.units = a.splits * a.gx; }
  dim3 grid((a.units + 7) / 8 * 8 * a.usize);
  if (name) *name = "wgrad_64x64";
  switch (a.mode) {
    case G_3x3: hipLaunchKernelGGL((wgrad_kernel<T, G_3x3>), grid, dim3(256), 0, s, a); break;
    case G_3x3P: if (name) *name = "wgrad_64x64_pos"; hipLaunchKernelGGL((wgrad_kernel<T, G_3x3P>), grid, dim3(256), 0, s, a); break;
    case G_2x2S2: hipLaunchKernelGGL((wgrad_kernel<T, G_2x2S2>), grid, dim3(256), 0, s, a); break;
    case G_1x1: hipLaunchKernelGGL((wgrad_kernel<T, G_1x1>), grid, dim3(256), 0, s, a); break;
    default: return fail(MI_EINVAL, "wgrad: bad gather mode");
  }
  return check_hip(hipGetLastError(), "wgrad launch");
}

}  // namespace mi
