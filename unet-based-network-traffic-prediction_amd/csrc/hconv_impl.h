// Halo-tile 3x3 convolution kernel for gfx950: forward and dgrad of the reference's
// Conv2d(k=3, padding=1) layers (models/unet2d.py:28-31, :38-41, :51-54) when the image
// tiles evenly into 256-pixel blocks (row bands of one image, or several whole small images).
//
// Difference to the generic implicit-GEMM kernel (igemm_impl.h): the 9 taps do not re-gather
// the input from global memory.  A workgroup stages ONE zero-ringed halo tile per 64-byte
// channel chunk in LDS -- (R+2) rows of `pitch` pixel slots per group of R image rows -- and
// every tap reads its MFMA pixel fragments from that tile at a constant row offset
// (ty*pitch + tx).  Global->LDS traffic for activations drops from 9x to ~1.3x.
//   * pitch is W+2 rounded up to a multiple of 8 slots: a vertical tap shift then leaves the
//     bank-swizzle bit of a row unchanged, so each lane needs only three precomputed fragment
//     addresses per pixel tile (tx = 0,1,2) plus a uniform ty offset -- no per-tap address math.
//   * 512 threads = 8 waves = 2 per SIMD, each wave 2 pixel tiles x TN channel tiles: while
//     one wave of a SIMD waits on LDS or issues loads/stores, its partner issues MFMAs.
//     (4 waves of 4 pixel tiles each -- fewer LDS reads per MFMA -- measured 25% slower: with
//     one wave per SIMD nothing covers that wave's LDS and global-load waits.)
//   * workgroups are persistent over pixel tiles; when the layer has at most two channel
//     chunks the packed weights of both stay resident in the two LDS stage buffers and are
//     loaded once per workgroup.
//   * padding lanes of the loaders read the zero page (see common.h) so every staging load of
//     a stage is issued back to back and waited for only at the LDS store after the MFMA phase.
#pragma once
#include "igemm_impl.h"

namespace mi {

struct HconvArgs {
  const void* in; int in_cs, in_co;
  const void* w; int Kp, Cp;            // packed weights [N][Kp], Cp channels per tap
  int N; const float* bias; int n_real; int relu;
  void* out; int out_cs, out_co;
  const void* mask; int mask_cs, mask_co;
  void* pool; int pool_cs, pool_co;     // optional: 2x2 max-pool of the (post-ReLU) output, written as well
  int W, R, G;                          // image width, rows per group, groups per tile (G*R*W = 256)
  int pitch;                            // halo slots per image row: W+2 rounded up to a multiple of 8
  int HR;                               // halo slots per tile = G*(R+2)*pitch
  int bands;                            // tiles per image (1 when a tile holds whole images)
  int ntiles, nchunks;
  const void* zero;                     // zero page
  FastDiv d_grp, d_pitch, d_rw, d_w, d_bands;   // by (R+2)*pitch, pitch, R*W, W, bands (filled in by the launcher)
  // two-tensor operands (DMA-fed kernel only): channel chunks from chunk2 on come from in2, output channels
  // from n2 on go to out2 (both with the first tensor's stride, starting at their own channel 0)
  const void* in2; int chunk2;
  void* out2; int n2;
  // fused 1x1 head (DMA-fed kernel, 32-channel tile): see IgemmArgs::head_out
  const float* head_w; const float* head_b; float* head_out; int head_cout, head_hw;
  FastDiv d_hw;                         // by the pixels of one image
};

constexpr int kHconvMaxHalo = 640;      // halo slots of the standard LDS buffer (covers 8 <= W <= 128)
constexpr int kHconvBigHalo = 768;      // 4x4 images (16 per tile, pitch 8): needs the narrower channel tiles

// Bank swizzle of the pixel tile: the 16-byte chunk index of halo slot hr is XORed with 2*bit.
// 16 consecutive slots (W >= 8) need bit 2 of hr; on 4x4 images (pitch 8) a 16-pixel MFMA tile is
// slots {0-3, 8-11, 16-19, 24-27}, whose bit 2 is constant -- a 4-way conflict -- and bit 3 (the
// halo row parity) is the one that spreads them.  A vertical tap shift of one row flips bit 3, so
// that variant flips the chunk bit of the precomputed fragment address for odd ty.
// Output-channel order inside a workgroup's BN = 16*TN channels.  MFMA row r of channel tile i is
// channel 4*TN*(r>>2) + 4*i + (r&3), so the 4*TN accumulator values a lane holds for one pixel
// (rows 4*(lane>>4)+q of every tile i) are 4*TN CONSECUTIVE channels: the epilogue writes them
// with 16-byte stores, and the four lanes of a pixel cover one contiguous 2*BN-byte run.
template <int TN> __device__ __forceinline__ int nmap(int i, int r) { return 4 * TN * (r >> 2) + 4 * i + (r & 3); }
// Bank swizzle of the weight image rows (row = tap*BN + n): a 16-lane fragment read touches rows
// {4*TN*g + 4*i + q}, g,q = 0..3; XORing the chunk with 2*(bit log2(4*TN) of the row) keeps every
// 16-lane phase of the ds_read_b128 on 64 distinct banks.
template <int TN> __device__ __forceinline__ int wswz(int row) {
  constexpr int SH = TN == 4 ? 4 : (TN == 2 ? 3 : 2);
  return ((row >> SH) & 1) << 1;
}
__device__ __forceinline__ void load_run(const float* p, float* v, int n) {
  for (int k = 0; k < n; k += 4) { const f32x4_t t = *(const f32x4_t*)(p + k); v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3]; }
}
__device__ __forceinline__ void load_run(const __bf16* p, float* v, int n) {
  if (n % 8 == 0) {
    for (int k = 0; k < n; k += 8) {
      const bf16x8_t t = *(const bf16x8_t*)(p + k);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k + e] = (float)t[e];
    }
  } else {
    for (int k = 0; k < n; k += 4) { const f32x4_t t = load4f(p + k); v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3]; }
  }
}
__device__ __forceinline__ void store_run(float* p, const float* v, int n) {
  for (int k = 0; k < n; k += 4) *(f32x4_t*)(p + k) = f32x4_t{v[k], v[k + 1], v[k + 2], v[k + 3]};
}
__device__ __forceinline__ void store_run(__bf16* p, const float* v, int n) {
  if (n % 8 == 0) {
    for (int k = 0; k < n; k += 8) {
      bf16x8_t t;
#pragma unroll
      for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[k + e];
      *(bf16x8_t*)(p + k) = t;
    }
  } else {
    for (int k = 0; k < n; k += 4) store4f(p + k, f32x4_t{v[k], v[k + 1], v[k + 2], v[k + 3]});
  }
}

template <bool ROWPAR> __device__ __forceinline__ int hswz(int hr) { return ROWPAR ? (hr >> 2) & 2 : (hr >> 1) & 2; }

template <typename T, int TN, int MH, int TM>
__global__ __launch_bounds__(256 / (TM * 16) * 64) void hconv_kernel(const HconvArgs a) {
  constexpr bool ROWPAR = MH > kHconvMaxHalo;    // the wide-halo variant serves 4x4 images only
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr int EPC = ElemTraits<T>::EPC;
  constexpr int CK = 4 * EPC;                    // channels per 64-byte chunk
  constexpr int BM = 256, BN = TN * 16, NT = BM / (TM * 16) * 64;
  constexpr int A_BYTES = MH * 64;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int A_IT = (MH * 4 + NT - 1) / NT;
  constexpr int W_IT = (9 * BN * 4 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ in = (const T*)a.in;
  const T* __restrict__ wp = (const T*)a.w;
  const int grp_rows = (a.R + 2) * a.pitch;
  const int chq = (t & 3) * EPC;   // element offset of this thread's 16-byte chunk within the 64-byte row
  // zero page as element offsets from each operand's base (integer offset select, one base pointer)
  const int64_t zoff_in = ((intptr_t)a.zero - (intptr_t)a.in) / (int64_t)sizeof(T);
  const int64_t zoff_w = ((intptr_t)a.zero - (intptr_t)a.w) / (int64_t)sizeof(T);

  // ---- loader state: this thread's halo slots (fixed for every tile)
  int relcs[A_IT];        // element offset relative to the tile's first pixel (rel_pixel * in_cs)
  int lds_a[A_IT];        // byte offset in the A buffer (multiple of 16) | kind in the low bits, or -1 (slot unused)
                          // kind: 0 interior, 1 always zero, 2 top halo row, 3 bottom halo row
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int id = t + NT * i;
    const int hr = id >> 2, ch = id & 3;
    relcs[i] = 0; lds_a[i] = -1;
    if (hr < a.HR) {
      const int g = (int)fdiv((uint32_t)hr, a.d_grp), rem = hr - g * grp_rows;
      const int hy = (int)fdiv((uint32_t)rem, a.d_pitch), hx = rem - hy * a.pitch;
      if (hx <= a.W + 1) {
        relcs[i] = (g * a.R * a.W + (hy - 1) * a.W + (hx - 1)) * a.in_cs;
        const int kind = (hx == 0 || hx == a.W + 1) ? 1 : (hy == 0 ? 2 : (hy == a.R + 1 ? 3 : 0));
        lds_a[i] = (hr * 64 + ((ch ^ hswz<ROWPAR>(hr)) << 4)) | kind;
      }
    }
  }
  // weight loader: row = tap*BN + n of the [9*BN][64 B] chunk image
  int woff[W_IT];         // element offset of this thread's weight chunk for channel chunk 0, or -1
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int id = t + NT * i, row = id >> 2;
    const int tap = row / BN, n = row - tap * BN;   // BN is a power of two
    woff[i] = row < 9 * BN ? (n0 + n) * a.Kp + tap * a.Cp + chq : -1;
  }

  // Two staging register sets: the loads of stage s+2 are issued at the start of stage s and
  // written to LDS at the end of stage s+1, so a load has two MFMA phases to land.
  typedef u32x4_t a_regs_t[A_IT];
  typedef u32x4_t w_regs_t[W_IT];
  a_regs_t ra0, ra1;
  w_regs_t rw0, rw1;
#pragma unroll
  for (int i = 0; i < W_IT; ++i) { rw0[i] = u32x4_t{0u, 0u, 0u, 0u}; rw1[i] = u32x4_t{0u, 0u, 0u, 0u}; }
  auto load_a = [&](a_regs_t& ra, int tile, int chunk) {
    const int band = tile - (int)fdiv((uint32_t)tile, a.d_bands) * a.bands;
    const bool top0 = band == 0, bot0 = band == a.bands - 1;
    const int64_t base = (int64_t)tile * BM * a.in_cs + a.in_co + chunk * CK + chq;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int kind = lds_a[i] & 3;   // (-1 & 3) == 3 is harmless: the slot is never stored
      const bool zero = lds_a[i] < 0 || kind == 1 || (kind == 2 && top0) || (kind == 3 && bot0);
      const int64_t off = zero ? zoff_in : base + relcs[i];
      ra[i] = *(const u32x4_t*)(in + opaque_offset(off));
    }
  };
  auto load_w = [&](w_regs_t& rw, int chunk) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int64_t off = woff[i] >= 0 ? (int64_t)woff[i] + chunk * CK : zoff_w;
      rw[i] = *(const u32x4_t*)(wp + opaque_offset(off));
    }
  };
  auto store_a = [&](const a_regs_t& ra, int buf) {
    char* A = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (lds_a[i] >= 0) *(u32x4_t*)(A + (lds_a[i] & ~15)) = ra[i];
  };
  auto store_a_piece = [&](const a_regs_t& ra, int buf, int i) {
    char* A = smem + buf * STAGE;
    if (lds_a[i] >= 0) *(u32x4_t*)(A + (lds_a[i] & ~15)) = ra[i];
  };
  auto store_w_piece = [&](const w_regs_t& rw, int buf, int i) {
    char* Wt = smem + buf * STAGE + A_BYTES;
    const int id = t + NT * i, row = id >> 2, ch = id & 3;
    if (row < 9 * BN) *(u32x4_t*)(Wt + row * 64 + ((ch ^ wswz<TN>(row)) << 4)) = rw[i];
  };
  auto store_w = [&](const w_regs_t& rw, int buf) {
    char* Wt = smem + buf * STAGE + A_BYTES;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int id = t + NT * i, row = id >> 2, ch = id & 3;
      if (row < 9 * BN) *(u32x4_t*)(Wt + row * 64 + ((ch ^ wswz<TN>(row)) << 4)) = rw[i];
    }
  };

  // ---- compute state: fragment addresses of this lane's pixel in each of its 2 tiles, per tx
  const int r16 = lane & 15, ch = lane >> 4;
  int paddr[TM][3];
  int out_off[TM], mask_off[TM];
  int pool_off[TM];                // element offset of this lane's pooled pixel inside the pooled tile, or -1 (not its writer)
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    // Fragment (16 consecutive pixels of a row) -> (wave, j).  For W >= 16 a wave's two fragments are
    // the same 16 columns of two vertically adjacent rows (2k, 2k+1), so the fused 2x2 max-pool pairs
    // rows inside a lane; for narrower images a fragment holds 2 or 4 whole rows itself.
    const int cpr = a.W >= 16 ? a.W >> 4 : 1;           // fragments per image row
    const int frag = a.W >= 16 ? (2 * (wave / cpr) + j) * cpr + (wave % cpr) : wave * TM + j;
    const int pl = frag * 16 + r16;                     // pixel index inside the tile
    const int g = (int)fdiv((uint32_t)pl, a.d_rw), rem = pl - g * (a.R * a.W);
    const int y = (int)fdiv((uint32_t)rem, a.d_w), x = rem - y * a.W;
    const int hrow = g * grp_rows + y * a.pitch + x;   // halo slot of tap (0,0)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int hr = hrow + tx;
      paddr[j][tx] = hr * 64 + ((ch ^ hswz<ROWPAR>(hr)) << 4);
    }
    out_off[j] = pl * a.out_cs + a.out_co + n0 + 4 * TN * ch;
    mask_off[j] = pl * a.mask_cs + a.mask_co + n0 + 4 * TN * ch;
    // the lane of the even row and even column of a 2x2 window writes its maximum
    const int pooled = g * (a.R >> 1) * (a.W >> 1) + (y >> 1) * (a.W >> 1) + (x >> 1);
    pool_off[j] = ((y | x) & 1) ? -1 : pooled * a.pool_cs + a.pool_co + n0 + 4 * TN * ch;
  }
  int wfrag[TN];                   // this lane's weight-fragment address inside a tap's BN rows, per channel tile
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = nmap<TN>(i, r16);
    wfrag[i] = A_BYTES + n * 64 + ((ch ^ wswz<TN>(n)) << 4);
  }
  const int ty_bytes = a.pitch * 64;

  f32x4_t acc[TN][TM];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  // MFMA phase of one staged chunk.  The LDS stores of the NEXT stage (registers loaded one stage
  // ago, buffer last read one stage ago) are spread over the nine taps instead of following the
  // MFMAs as a block: the LDS write path then works in the shadow of the matrix pipe and the
  // workgroup does not sit through a store-only phase before every barrier.
  auto compute = [&](int buf, const a_regs_t& held_a, const w_regs_t& held_w, bool st_a, bool st_w) {
    const char* S = smem + buf * STAGE;
    const int nbuf = buf ^ 1;
    // fragments of tap t+1 are read while the MFMAs of tap t issue (two register sets)
    frag_t pf[2][TM], wf[2][TN];
    auto read_tap = [&](int tap, int set) {
      const int ty = tap / 3, tx = tap % 3;
      const char* Sy = S + ty * ty_bytes;
      const int flip = (ROWPAR && (ty & 1)) ? 32 : 0;
#pragma unroll
      for (int j = 0; j < TM; ++j) pf[set][j] = *(const frag_t*)(Sy + (paddr[j][tx] ^ flip));
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[set][i] = *(const frag_t*)(S + tap * BN * 64 + wfrag[i]);
    };
    constexpr int PIECES = A_IT + W_IT;
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma_slab(wf[tap & 1][i], pf[tap & 1][j], acc[i][j]);
      // pieces tap, tap+9, ... of the next stage's image
#pragma unroll
      for (int pc = tap; pc < PIECES; pc += 9) {
        if (pc < A_IT) { if (st_a) store_a_piece(held_a, nbuf, pc); }
        else if (st_w) store_w_piece(held_w, nbuf, pc - A_IT);
      }
    }
  };
  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
  // bias of this lane's 4*TN consecutive channels, read once (clamped unconditional loads)
  constexpr int CPL = 4 * TN;
  float biasv[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int n = n0 + CPL * ch + k;
    float b = 0.f;
    if (a.bias != nullptr) {
      b = a.bias[n < a.n_real ? n : 0];
      if (n >= a.n_real) b = 0.f;
    }
    biasv[k] = b;
  }
  auto epilogue = [&](int tile) {
    T* otile = out + (int64_t)tile * BM * a.out_cs;
    if (mask != nullptr) {
      // all mask loads first, then all stores: interleaved, every store would wait for its load
      const T* mtile = mask + (int64_t)tile * BM * a.mask_cs;
      float mv[TM][CPL];
#pragma unroll
      for (int j = 0; j < TM; ++j) load_run(mtile + mask_off[j], mv[j], CPL);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float v[CPL];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x = acc[i][j][q] + biasv[4 * i + q];
            v[4 * i + q] = mv[j][4 * i + q] > 0.f ? x : 0.f;
          }
        store_run(otile + out_off[j], v, CPL);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      float v[CPL];
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = acc[i][j][q] + biasv[4 * i + q];
          v[4 * i + q] = (a.relu && !(x > 0.f)) ? 0.f : x;
        }
      store_run(otile + out_off[j], v, CPL);
    }
    if (a.pool != nullptr) {
      // MaxPool2d(2,2) of the tile fused here (models/unet2d.py:68): the window partners are other
      // lanes of the same wave (column: lane^1; row: the wave's other fragment when W >= 16, lane^W
      // when a fragment holds two or four rows).  max commutes with the rounding to the storage type.
      T* ptile = (T*)a.pool + (int64_t)tile * (BM / 4) * a.pool_cs;
      auto val = [&](int i, int j, int q) {
        const float x = acc[i][j][q] + biasv[4 * i + q];
        return (a.relu && !(x > 0.f)) ? 0.f : x;
      };
      if (a.W >= 16) {
        static_assert(TM == 2, "fragments 0 and 1 of a wave are vertically adjacent rows when W >= 16");
        float m[CPL];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float t = fmaxf(val(i, 0, q), val(i, 1, q));
            m[4 * i + q] = fmaxf(t, __shfl_xor(t, 1, 64));
          }
        if (pool_off[0] >= 0) store_run(ptile + pool_off[0], m, CPL);
      } else {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          float m[CPL];
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float t = val(i, j, q);
              t = fmaxf(t, __shfl_xor(t, a.W, 64));
              m[4 * i + q] = fmaxf(t, __shfl_xor(t, 1, 64));
            }
          if (pool_off[j] >= 0) store_run(ptile + pool_off[j], m, CPL);
        }
      }
    }
  };

  // ---- persistent loop over (tile, chunk) stages
  const bool w_resident = a.nchunks <= 2;   // stage parity == chunk parity: both chunks stay in LDS
  const int my_tiles = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstages = my_tiles * a.nchunks;
  if (nstages <= 0) return;
  // load cursor: (tile, chunk) of the next stage to fetch
  int ltile = blockIdx.x, lchunk = 0;
  auto advance_load = [&]() {
    if (++lchunk == a.nchunks) { lchunk = 0; ltile += gridDim.x; }
  };
  // resident weights: stage parity selects the LDS buffer, so both buffers are filled by the
  // first two stages (with a single chunk that is the same chunk twice) and never again
  auto stage_needs_w = [&](int s) { return !w_resident || s < 2; };

  load_a(ra0, ltile, lchunk);
  load_w(rw0, lchunk);
  advance_load();
  store_a(ra0, 0);
  store_w(rw0, 0);
  if (nstages > 1) {                       // stage 1 stays in set 0 until the end of stage 0
    load_a(ra0, ltile, lchunk);
    if (stage_needs_w(1)) load_w(rw0, lchunk);
    advance_load();
  }
  __syncthreads();
  zero_acc();
  int tile = blockIdx.x, chunk = 0;
  // `held` carries stage s+1 (already in flight), `fresh` receives stage s+2
  auto stage = [&](int s, a_regs_t& held_a, w_regs_t& held_w, a_regs_t& fresh_a, w_regs_t& fresh_w) {
    if (s + 2 < nstages) {
      load_a(fresh_a, ltile, lchunk);
      if (stage_needs_w(s + 2)) load_w(fresh_w, lchunk);
      advance_load();
    }
    const bool nxt = s + 1 < nstages;
    compute(s & 1, held_a, held_w, nxt, nxt && stage_needs_w(s + 1));
    if (chunk == a.nchunks - 1) {
      epilogue(tile);
      zero_acc();
    }
    __syncthreads();
    if (++chunk == a.nchunks) { chunk = 0; tile += gridDim.x; }
  };
  for (int s = 0; s < nstages; s += 2) {
    stage(s, ra0, rw0, ra1, rw1);
    if (s + 1 < nstages) stage(s + 1, ra1, rw1, ra0, rw0);
  }
}

// Tiling of an (H, W) image into 256-pixel tiles; returns false when the halo kernel does not apply.
inline bool hconv_tiling(int H, int W, int* R, int* G, int* bands) {
  if (W < 4 || W > 128) return false;
  const int HW = H * W;
  if (HW >= 256) {
    if (256 % W != 0) return false;
    const int r = 256 / W;
    if (H % r != 0) return false;
    *R = r; *G = 1; *bands = H / r;
  } else {
    if (256 % HW != 0) return false;
    *R = H; *G = 256 / HW; *bands = 1;
  }
  // the wide-halo variant's bank swizzle assumes a halo pitch of 8 slots (a vertical tap shift
  // flips bit 3 of the slot index): only 4-wide images have it
  const int pitch = (W + 2 + 7) / 8 * 8, slots = (*G) * (*R + 2) * pitch;
  return slots <= kHconvMaxHalo || (slots <= kHconvBigHalo && pitch == 8);
}

template <typename T>
static int launch_hconv_t(const HconvArgs& a_in, hipStream_t s, const char** name) {
  HconvArgs a = a_in;
  a.d_grp = make_fastdiv((a.R + 2) * a.pitch); a.d_pitch = make_fastdiv(a.pitch); a.d_rw = make_fastdiv(a.R * a.W);
  a.d_w = make_fastdiv(a.W); a.d_bands = make_fastdiv(a.bands);
  const bool bighalo = a.HR > kHconvMaxHalo;    // the larger halo buffer leaves LDS room for 32 channels
  const int bn = (a.N % 64 == 0 && !bighalo) ? 64 : (a.N % 32 == 0) ? 32 : 16;
  const int ny = a.N / bn;
  int gx = a.ntiles;
  const int cap = (256 + ny - 1) / ny;          // one workgroup per CU
  if (gx > cap) gx = cap;
  dim3 grid(gx, ny);
  if (bighalo) {
    if (bn == 32) { if (name) *name = "hconv_256x32_h768"; hipLaunchKernelGGL((hconv_kernel<T, 2, kHconvBigHalo, 2>), grid, dim3(512), 0, s, a); }
    else { if (name) *name = "hconv_256x16_h768"; hipLaunchKernelGGL((hconv_kernel<T, 1, kHconvBigHalo, 2>), grid, dim3(512), 0, s, a); }
  }
  else if (bn == 64) { if (name) *name = "hconv_256x64"; hipLaunchKernelGGL((hconv_kernel<T, 4, kHconvMaxHalo, 2>), grid, dim3(512), 0, s, a); }
  else if (bn == 32) { if (name) *name = "hconv_256x32"; hipLaunchKernelGGL((hconv_kernel<T, 2, kHconvMaxHalo, 2>), grid, dim3(512), 0, s, a); }
  else { if (name) *name = "hconv_256x16"; hipLaunchKernelGGL((hconv_kernel<T, 1, kHconvMaxHalo, 2>), grid, dim3(512), 0, s, a); }
  return check_hip(hipGetLastError(), "hconv launch");
}

}  // namespace mi
