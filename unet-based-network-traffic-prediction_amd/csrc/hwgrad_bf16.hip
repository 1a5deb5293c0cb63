// Halo-tile weight-gradient kernel for gfx950 (bf16): dW of the reference's Conv2d(k=3,
// padding=1) layers (models/unet2d.py:28-31, :38-41, :51-54), i.e. their autograd backward
// w.r.t. the weights, for images that tile evenly into 256-pixel blocks.
//
//   part[s][n][tap*Cb + c] = sum_{p in pixel range s} dy[p][n] * x[p + tap][c]
//
// A 512-thread workgroup owns NB (32|64) dy-channels x 32 x-channels x all 9 taps and walks
// its share of the pixel tiles.  Per tile it stages the dy tile [256 px][NB] and ONE
// zero-ringed halo tile of x [halo rows][32 ch] in LDS; the 9 taps read their MFMA fragments
// from that halo tile at constant row offsets, so x is fetched ~1.3x instead of 9x.  The
// reduction runs over pixels (the slow index of NHWC), so fragments are read transposed with
// ds_read_b64_tr_b16.  Halo rows are laid out with a pitch that is a multiple of 16 rows:
// a vertical tap shift then leaves the swizzle bits unchanged and becomes an immediate
// offset; the three horizontal shifts use three precomputed addresses.
// The tiles are brought in by FOUR LOADER WAVES (waves 8-11) through buffer LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass; lanes of padding slots carry
// an out-of-range offset and write zeros); the eight compute waves only read fragments and issue
// MFMAs, and meet the loaders at one barrier per tile.
// Each wave keeps 2 n-tiles x 1 c-tile x 9 taps of accumulators; waves that share an
// accumulator block split the tile's pixels and are summed through LDS at the end, so a
// workgroup writes one partial slab; the slabs of the pixel splits are summed by the
// fixed-order combine pass (bitwise repeatable, no atomics).
#include "wgrad_impl.h"
#include "hconv_impl.h"

namespace mi {

constexpr int kHwMaxHalo = 768;   // halo rows (pitch-padded) the LDS buffer holds: 16 images of 4x4 at pitch 8

// Bank rules.  One ds_read_b64_tr_b16 serves 32 lanes = 8 pixels x 32 bytes per cycle, and a 32-lane
// group of a fragment read holds pixels p..p+3 and p+4..p+7 of the tile (pixel_of() below).
// dy tile: [256][NB] bf16, the 8 rows of a group are consecutive.  NB = 64 -> 128-byte rows (two
// rows span the 64 banks): 32-byte column group ^ row bits 1-2; NB = 32 -> 64-byte rows (four
// rows span the banks): 32-byte column group ^ row bit 2.
template <int NB> __device__ __forceinline__ int dy_swz(int r);
template <> __device__ __forceinline__ int dy_swz<64>(int r) { return (r >> 1) & 3; }
template <> __device__ __forceinline__ int dy_swz<32>(int r) { return (r >> 2) & 1; }
template <int NB> __device__ __forceinline__ int dy_off(int r, int q /*16-byte chunk*/) {
  return r * (NB * 2) + ((((q >> 1) ^ dy_swz<NB>(r)) << 1 | (q & 1)) << 4);
}
// x halo tile: [rows][32] bf16 = 64-byte rows (four rows span the banks).  The two 4-pixel runs of
// a group lie 4 halo rows apart (images 8 and more wide: same image row or, 8 wide, the next one
// at a pitch of 16) or one image row = 8 halo rows apart (4-wide images, pitch 8): the 32-byte half
// is flipped by that row bit.
template <bool RP> __device__ __forceinline__ int xh_swz(int r) { return (r >> (RP ? 3 : 2)) & 1; }
template <bool RP> __device__ __forceinline__ int xh_off(int r, int q) {
  return r * 64 + ((((q >> 1) ^ xh_swz<RP>(r)) << 1 | (q & 1)) << 4);
}
// Pixel (inside a 32-pixel k-step) that MFMA k-slot (g = lane>>4, q = (lane&15)>>2, half h) stands for:
// both operands use the same map, so any order is a valid reduction order.
__device__ __forceinline__ int pixel_of(int g, int q, int h) { return 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q; }

__device__ __forceinline__ bf16x8_t tr_pair(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  union { s16x4_t s[2]; bf16x8_t v; } u;
  u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
  u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
  return u.v;
}

template <typename V> __device__ __forceinline__ void hw_pin1(V v) { asm volatile("" ::"s"(v)); }
template <typename... A> __device__ __forceinline__ void hw_pin(A... v) { (hw_pin1(v), ...); }

// Work of workgroup (bx, by, bz) = (n-block, c-block, pixel split) of one layer.
// RP: 4-wide images, halo pitch 8 -- a vertical tap shift of one row flips bit 3 of the halo row, i.e. the
// swizzle bit: odd ty reads the other 32-byte half (the rule the conv kernels use for these images).
template <int NB, bool RP>
__device__ __forceinline__ void hwgrad_body(const HwgradArgs& a, const int bx, const int by, const int bz, char* smem) {
  typedef __bf16 T;
  constexpr int COMBOS = (NB / 32) * 2;        // (n-pair, c-tile) accumulator blocks
  constexpr int PG = 8 / COMBOS;               // waves sharing a block split the pixels PG ways
  constexpr int KSP = 8 / PG;                  // 32-pixel k-steps per wave per tile
  constexpr int DY_BYTES = 256 * NB * 2;
  constexpr int X_BYTES = kHwMaxHalo * 64;
  constexpr int STAGE = DY_BYTES + X_BYTES;
  constexpr int DY_CPR = NB / 8;               // 16-byte chunks per dy row
  // every argument is fetched here, in one batch of scalar loads, not as dependent round trips at first use
  hw_pin(a.x2, a.x2_c0, a.dy, a.x, a.part, a.rowsum, a.dy_cs, a.dy_co, a.Na, a.x_cs, a.x_co, a.Cb, a.W, a.R, a.pitch, a.HR, a.bands, a.ntiles,
         a.tiles_per_split, a.m_grp, a.m_pitch, a.d_rw.mul, a.d_rw.shift, a.d_w.mul, a.d_w.shift, a.d_bands.mul, a.d_bands.shift);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n0 = bx * NB, c0 = by * 32, z = bz;
  const T* __restrict__ pdy = (const T*)a.dy;
  // a two-tensor input: the channel blocks from x2_c0 on are the first blocks of the second tensor
  const bool second = a.x2 != nullptr && c0 >= a.x2_c0;
  const T* __restrict__ px = second ? (const T*)a.x2 + (c0 - a.x2_c0) : (const T*)a.x + a.x_co + c0;
  const int grp_rows = (a.R + 2) * a.pitch;

  const int t_begin = z * a.tiles_per_split;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
  if (t_begin >= t_end) return;                 // (cannot happen: the launcher sizes the splits to the tiles)

  if (wave >= 8) {
    // ================================================================ loader waves
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - 8;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int DY_PIECES = DY_BYTES / 1024, DY_PW = DY_PIECES / 4;     // 1 KiB pieces of the dy tile, per loader wave
    constexpr int RPP = 1024 / (NB * 2);                                  // dy rows per piece
    constexpr int X_PW = kHwMaxHalo / 16 / 4;                             // x pieces per loader wave
    // dy: lane L of piece p fills chunk position L % DY_CPR of row p*RPP + L / DY_CPR with the channel chunk the swizzle puts there
    // The first tile's pieces are issued as soon as their offsets exist: dy first (its offsets are
    // cheap), then each halo piece as the geometry loop produces it.
    auto dy_rsrc = [&](int tile) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pdy + ((int64_t)tile * 256 * a.dy_cs + a.dy_co + n0)), 0, 0x7fffffff, 0x00020000);
    };
    auto x_rsrc = [&](int tile) {
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(px + ((int64_t)tile * 256 * a.x_cs - (int64_t)a.W * a.x_cs)), 0,
                                               0x7fffffff, 0x00020000);
    };
    auto dma = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, char* dst) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, (int)voff, 0, 0, 0);
    };
    unsigned dy_voff[DY_PW];
    {
      const __amdgpu_buffer_rsrc_t rd0 = dy_rsrc(t_begin);
#pragma unroll
      for (int i = 0; i < DY_PW; ++i) {
        const int row = (lw + 4 * i) * RPP + lane / DY_CPR, pos = lane % DY_CPR;
        const int c = (((pos >> 1) ^ dy_swz<NB>(row)) << 1) | (pos & 1);
        dy_voff[i] = n0 + c * 8 < a.Na ? (unsigned)(row * a.dy_cs + c * 8) * 2u : OOB;
        dma(rd0, dy_voff[i], smem + lw * 1024 + i * 4096);
      }
    }
    // x halo: piece q covers halo rows [16q, 16q+16); offsets are bytes from (tile's first pixel - one image row)
    const int l4 = lane >> 2, pos4 = lane & 3;
    const unsigned m_grp = a.m_grp, m_pitch = a.m_pitch;     // (from the launcher: two integer divisions are ~50 instructions here)
    const unsigned uW = (unsigned)a.W, uR1 = (unsigned)a.R + 1u, RW = (unsigned)(a.R * a.W);
    const bool single = a.bands == 1;
    const int npieces = (a.HR + 15) / 16;
    const int band0 = t_begin - (int)fdiv((uint32_t)t_begin, a.d_bands) * a.bands;
    const bool ztop0 = !single && band0 == 0, zbot0 = !single && band0 == a.bands - 1;
    const __amdgpu_buffer_rsrc_t rx0 = x_rsrc(t_begin);
    unsigned x_voff[X_PW], topmask = 0, botmask = 0, skipmask = 0;
#pragma unroll
    for (int i = 0; i < X_PW; ++i) {
      const int q = lw + 4 * i;
      // rounds beyond the halo's pieces cost nothing (a 32-wide image has 30 of 48): at kernel entry all twelve waves
      // of the CU run set-up arithmetic and this loop is bound by instruction issue
      if (q >= npieces) { x_voff[i] = OOB; skipmask |= 1u << i; continue; }
      const unsigned hr = (unsigned)(q * 16 + l4);
      const unsigned g = __umul24(hr, m_grp) >> 20, rem = hr - __umul24(g, (unsigned)grp_rows);
      const unsigned hy = __umul24(rem, m_pitch) >> 20, hx = rem - __umul24(hy, (unsigned)a.pitch);
      const bool top = hy == 0, bot = hy == uR1;
      const bool on = (q < npieces) & (hr < (unsigned)a.HR) & (hx - 1u < uW) & !((top | bot) & single);
      const int pix = (int)(__umul24(g, RW) + __umul24(hy, uW) + hx) - (int)(uW + 1u) + a.W;   // + one image row: never negative
      const int c = (((pos4 >> 1) ^ xh_swz<RP>((int)hr)) << 1) | (pos4 & 1);
      x_voff[i] = on ? (unsigned)(__mul24(pix, a.x_cs) + c * 8) * 2u : OOB;
      topmask |= (unsigned)(on & top) << i;
      botmask |= (unsigned)(on & bot) << i;
      const bool any = __builtin_amdgcn_ballot_w64(on) != 0;
      skipmask |= (unsigned)(!any) << i;
      if (any) dma(rx0, ((ztop0 & on & top) | (zbot0 & on & bot)) ? OOB : x_voff[i], smem + DY_BYTES + q * 1024);
      else if (q < npieces) {           // an all-padding piece: zeroed once in both stages, no DMA ever writes it
        *(u32x4_t*)(smem + DY_BYTES + q * 1024 + lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
        *(u32x4_t*)(smem + STAGE + DY_BYTES + q * 1024 + lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
      }
    }
    auto issue_tile = [&](int tile, int buf) {
      char* D = smem + buf * STAGE + lw * 1024;
      const __amdgpu_buffer_rsrc_t rd = dy_rsrc(tile);
#pragma unroll
      for (int i = 0; i < DY_PW; ++i) dma(rd, dy_voff[i], D + i * 4096);
      const __amdgpu_buffer_rsrc_t rx = x_rsrc(tile);
      const int band = tile - (int)fdiv((uint32_t)tile, a.d_bands) * a.bands;
      const unsigned zmask = single ? 0u : ((band == 0 ? topmask : 0u) | (band == a.bands - 1 ? botmask : 0u));
      char* X = D + DY_BYTES;
#pragma unroll
      for (int i = 0; i < X_PW; ++i)
        if (!((skipmask >> i) & 1)) dma(rx, ((zmask >> i) & 1) ? OOB : x_voff[i], X + i * 4096);
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the first tile and the one-time zero stores
    __builtin_amdgcn_s_barrier();
    for (int tile = t_begin; tile < t_end; ++tile) {
      if (tile + 1 < t_end) issue_tile(tile + 1, (tile + 1 - t_begin) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---- wave role and fragment addresses (compute waves)
  const int combo = wave % COMBOS, pg = wave / COMBOS;
  const int ct = combo & 1, np = combo >> 1;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  // dy fragment addresses: the swizzle depends on pixel bits 1-2 only (q and g&1), so the address
  // of k-step k, row half h is a compile-time offset from one base per n-tile
  constexpr int ROWB = NB * 2;
  int a_base[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = np * 32 + nt * 16 + 4 * pp;                    // channel inside the dy tile
    a_base[nt] = dy_off<NB>(pg * KSP * 32 + pixel_of(g, q, 0), col >> 3) + (col & 7) * 2;
  }
  // x fragment addresses [k-step][row half][tx]; 4-wide images: a horizontal shift never carries
  // into the swizzle bit, so tx is a compile-time offset from [..][0] there
  int b_addr[KSP][2][RP ? 1 : 3];
#pragma unroll
  for (int k = 0; k < KSP; ++k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pl = (pg * KSP + k) * 32 + pixel_of(g, q, h);      // pixel inside the tile
      const int gi = (int)fdiv((uint32_t)pl, a.d_rw), rem = pl - gi * (a.R * a.W);
      const int y = (int)fdiv((uint32_t)rem, a.d_w), xx = rem - y * a.W;
      const int hrow = gi * grp_rows + y * a.pitch + xx;            // halo row of tap (0,0)
      const int col = ct * 16 + 4 * pp;
#pragma unroll
      for (int tx = 0; tx < (RP ? 1 : 3); ++tx) b_addr[k][h][tx] = DY_BYTES + xh_off<RP>(hrow + tx, col >> 3) + (col & 7) * 2;
    }
  }
  // 4-wide images: a vertical shift of one image row (8 halo rows) flips the swizzle bit, i.e. the
  // 32-byte half every address of this lane points into: +32 or -32, the same for all of them
  const int flipd = RP ? ((b_addr[0][0][0] & 32) ? -32 : 32) : 0;
  const int ty_bytes = a.pitch * 64;

  f32x4_t acc[9][2], accr[2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) { acc[tp][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[tp][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  accr[0] = accr[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_rows = a.rowsum != nullptr && by == 0 && ct == 0;
  const bf16x8_t ones = ones_frag(T());

#ifndef MI_HWGRAD_LEAD
#define MI_HWGRAD_LEAD 2
#endif
  constexpr int kLead = MI_HWGRAD_LEAD;      // steps of lead of the x fragment reads
  auto compute = [&](int buf) {
    const char* S = smem + buf * STAGE;
    // software pipeline over the flattened (k-step, tap) sequence: the fragments of step i+1
    // are read while the MFMAs of step i issue (two register sets, static indices)
    bf16x8_t af[2][2], bfr[kLead + 1];
    auto read_a = [&](int k, int set) {
      af[set][0] = tr_pair(S + a_base[0] + k * 32 * ROWB, S + a_base[0] + k * 32 * ROWB + 8 * ROWB);
      af[set][1] = tr_pair(S + a_base[1] + k * 32 * ROWB, S + a_base[1] + k * 32 * ROWB + 8 * ROWB);
    };
    auto read_b = [&](int idx) {           // x fragment of flattened step idx into set idx % (kLead + 1)
      const int k = idx / 9, tp = idx % 9;
      const char* Sy = S + (tp / 3) * ty_bytes + (((tp / 3) & 1) ? flipd : 0) + (RP ? (tp % 3) * 64 : 0);
      bfr[idx % (kLead + 1)] = tr_pair(Sy + b_addr[k][0][RP ? 0 : tp % 3], Sy + b_addr[k][1][RP ? 0 : tp % 3]);
    };
    // x fragments are read TWO steps ahead of their MFMAs (three register sets), the dy fragments of the next
    // k-step at its last-but-one tap, and the order is spelled out for the scheduler: one step of two MFMAs covers
    // ~64 cycles of the SIMD's pipe, less than an LDS round trip, and left alone the compiler keeps one step of lead
    read_a(0, 0);
#pragma unroll
    for (int i = 0; i < kLead; ++i) read_b(i);
#ifndef MI_HWGRAD_NO_SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + 2 * kLead, 0);
#endif
    constexpr int STEPS = KSP * 9;
#pragma unroll
    for (int idx = 0; idx < STEPS; ++idx) {
      const int k = idx / 9, tp = idx % 9;
      int nreads = 0;
      if (idx + kLead < STEPS) { read_b(idx + kLead); nreads += 2; }
      if (tp == 7 && k + 1 < KSP) { read_a(k + 1, (k + 1) & 1); nreads += 4; }
      acc[tp][0] = mma_px(af[k & 1][0], bfr[idx % (kLead + 1)], acc[tp][0]);
      acc[tp][1] = mma_px(af[k & 1][1], bfr[idx % (kLead + 1)], acc[tp][1]);
#ifndef MI_HWGRAD_NO_SCHED
      if (idx + kLead < STEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      if (tp == 7 && k + 1 < KSP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#endif
      (void)nreads;
      if (tp == 8 && do_rows) {
        accr[0] = mma_px(af[k & 1][0], ones, accr[0]);
        accr[1] = mma_px(af[k & 1][1], ones, accr[1]);
      }
    }
  };

  __builtin_amdgcn_s_barrier();                 // the first tile has landed
  for (int tile = t_begin; tile < t_end; ++tile) {
    compute((tile - t_begin) & 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- combine the PG pixel groups of every accumulator block through LDS, upper half into
  // lower half, in a fixed order; the staging buffers are free after the last barrier
  constexpr int RV = 9 * 2 * 4 + 8;            // accumulator floats per lane (acc + accr)
  static_assert((PG / 2) * COMBOS * RV * 256 <= 2 * STAGE, "combine buffer must fit in the staging area");
  float* red = (float*)smem;
#pragma unroll
  for (int half = PG / 2; half >= 1; half >>= 1) {
    if (pg >= half && pg < 2 * half) {
      float* dst = red + ((pg - half) * COMBOS + combo) * (RV * 64) + lane;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) dst[((tp * 2 + nt) * 4 + j) * 64] = acc[tp][nt][j];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[(72 + nt * 4 + j) * 64] = accr[nt][j];
    }
    __syncthreads();
    if (pg < half) {
      const float* src = red + (pg * COMBOS + combo) * (RV * 64) + lane;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[tp][nt][j] += src[((tp * 2 + nt) * 4 + j) * 64];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) accr[nt][j] += src[(72 + nt * 4 + j) * 64];
    }
    __syncthreads();
  }
  if (pg != 0) return;

  // ---- write the workgroup's partial slab: D row = n (4*(lane>>4)+j), col = c (lane & 15)
  const int K = 9 * a.Cb;
  const int slab = z;
  float* part = a.part + (int64_t)slab * a.Na * K;
  const int c = c0 + ct * 16 + i16;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + np * 32 + nt * 16 + 4 * g + j;
      if (n < a.Na && c < a.Cb) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) part[(int64_t)n * K + tp * a.Cb + c] = acc[tp][nt][j];
      }
    }
  if (do_rows && i16 == 0) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + np * 32 + nt * 16 + 4 * g + j;
        if (n < a.Na) a.rowsum[(int64_t)slab * a.Na + n] = accr[nt][j];
      }
  }
}

// One launch serves up to two layers (the two convs of a U-Net block see the same pixels): the
// workgroups of both share the CUs, so each layer needs only its share of pixel splits to fill
// the chip and writes that many fewer partial slabs.  Workgroup ids run over layer 0 then layer 1,
// n-block fastest.
template <int NB, bool RP>
__global__ __launch_bounds__(768) void hwgrad_kernel(const HwgradBatch p) {
  constexpr int STAGE = 256 * NB * 2 + kHwMaxHalo * 64;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  int id = blockIdx.x;
  const int li = (p.n > 1 && id >= p.wgs0) ? 1 : 0;
  if (li) id -= p.wgs0;
  const HwgradArgs& a = p.layer[li];
  const int gx = (a.Na + NB - 1) / NB, gy = a.Cb / 32;
  int bx, by, bz;
  if (p.xcd_order) {
    // id = slot*8 + xcd; the XCD's slots run over its units (n-block, split), gy workgroups each
    const int xcd = id & 7, slot = id >> 3;
    const int unit = (slot / gy) * 8 + ((xcd + 8 - p.rot[li]) & 7);   // unit u sits on XCD (u + rot) % 8
    if (unit >= gx * a.splits) return;          // a hole of the deal: 8 does not divide the unit count
    by = slot % gy; bz = unit / gx; bx = unit - bz * gx;
  } else {
    bz = id / (gx * gy);
    const int rem = id - bz * (gx * gy);
    by = rem / gx; bx = rem - by * gx;
  }
  hwgrad_body<NB, RP>(a, bx, by, bz, smem);
}

// Tiling with a 16-row-aligned halo pitch; false when the kernel does not apply.
bool hwgrad_tiling(int H, int W, int* R, int* G, int* bands, int* pitch, int* HR) {
  if (!hconv_tiling(H, W, R, G, bands)) return false;
  *pitch = W == 4 ? 8 : (W + 2 + 15) / 16 * 16;
  *HR = (*G) * (*R + 2) * (*pitch);
  return *HR <= kHwMaxHalo;
}
int hwgrad_slabs_per_split(int) { return 1; }

static int hwgrad_wgs(const HwgradArgs& a, int NB) { return (a.Na + NB - 1) / NB * (a.Cb / 32) * a.splits; }
int launch_hwgrad_bf16(const HwgradArgs* layers, int n, hipStream_t s, const char** name) {
  MI_REQUIRE(n == 1 || n == 2, "hwgrad: one or two layers per launch");
  HwgradBatch p{};
  p.n = n;
  const int NB = layers[0].Na % 64 == 0 ? 64 : 32;
  for (int i = 0; i < n; ++i) {
    HwgradArgs a = layers[i];
    MI_REQUIRE((a.Na % 64 == 0 ? 64 : 32) == NB, "hwgrad: layers of one launch must share the channel-block width");
    a.d_grp = make_fastdiv((a.R + 2) * a.pitch); a.d_pitch = make_fastdiv(a.pitch); a.d_rw = make_fastdiv(a.R * a.W);
    a.d_w = make_fastdiv(a.W); a.d_bands = make_fastdiv(a.bands);
    a.m_grp = ((1u << 20) + (a.R + 2) * a.pitch - 1) / ((a.R + 2) * a.pitch);
    a.m_pitch = ((1u << 20) + a.pitch - 1) / a.pitch;
    p.layer[i] = a;
  }
  // XCD-aware order when no XCD (32 CUs) would be dealt more than 32 workgroups, else the plain one
  auto units = [&](const HwgradArgs& a) { return (a.Na + NB - 1) / NB * a.splits; };
  auto dealt = [&](const HwgradArgs& a) { return (units(a) + 7) / 8 * 8 * (a.Cb / 32); };
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  p.rot[0] = 0; p.rot[1] = n > 1 ? units(p.layer[0]) % 8 : 0;
  for (int i = 0; i < n; ++i)
    for (int u = 0; u < units(p.layer[i]); ++u) load[(u + p.rot[i]) % 8] += p.layer[i].Cb / 32;
  int worst = 0;
  for (int x = 0; x < 8; ++x) worst = load[x] > worst ? load[x] : worst;
  p.xcd_order = worst <= 32;
  p.wgs0 = p.xcd_order ? dealt(p.layer[0]) : hwgrad_wgs(p.layer[0], NB);
  const int total = p.wgs0 + (n > 1 ? (p.xcd_order ? dealt(p.layer[1]) : hwgrad_wgs(p.layer[1], NB)) : 0);
  const bool rp = p.layer[0].pitch == 8;
  for (int i = 0; i < n; ++i) MI_REQUIRE((p.layer[i].pitch == 8) == rp, "hwgrad: layers of one launch must share the halo pitch rule");
  if (NB == 64) {
    if (name) *name = "hwgrad_64x32x9";
    if (rp) hipLaunchKernelGGL((hwgrad_kernel<64, true>), dim3(total), dim3(768), 0, s, p);
    else hipLaunchKernelGGL((hwgrad_kernel<64, false>), dim3(total), dim3(768), 0, s, p);
  } else {
    if (name) *name = "hwgrad_32x32x9";
    if (rp) hipLaunchKernelGGL((hwgrad_kernel<32, true>), dim3(total), dim3(768), 0, s, p);
    else hipLaunchKernelGGL((hwgrad_kernel<32, false>), dim3(total), dim3(768), 0, s, p);
  }
  return check_hip(hipGetLastError(), "hwgrad launch");
}

}  // namespace mi
