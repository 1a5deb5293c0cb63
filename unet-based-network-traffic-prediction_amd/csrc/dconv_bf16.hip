// DMA-fed halo-tile 3x3 convolution for gfx950, bf16: forward and dgrad of the reference's
// Conv2d(k=3, padding=1) layers (models/unet2d.py:28-31, :38-41, :51-54) on images that tile evenly
// into 256-pixel blocks.  Same tile geometry, fragment maps, bank swizzles and epilogue as the
// register-staged kernel in hconv_impl.h; what differs is how the tiles get into LDS:
//   * global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4, 1 KiB per wave
//     instruction): no staging registers, no ds_write pass.  The DMA writes LDS linearly
//     (wave base + lane*16), so the bank swizzle is applied to the per-lane SOURCE address.
//   * the DMAs are issued by FOUR LOADER WAVES (waves 8-11, one per SIMD) that do nothing else.
//     Issuing a 1 KiB piece holds the issuing wave for 100-200 cycles when eight waves load at
//     once; with the loads in the compute waves' own instruction streams that was a third of a
//     stage (measured with in-kernel stamps: 2,200 of 6,100 cycles).  The eight compute waves now
//     run fragment reads and MFMAs only and meet the loaders at one barrier per stage.
//   * the DMAs go through a buffer resource (buffer_load_dwordx4 ... lds): a precomputed 32-bit
//     offset per lane and piece, the (tile, chunk) position in the resource's base -- two
//     instructions per piece.  Lanes of padding slots carry an out-of-range offset: they fetch
//     nothing and write zeros, so the zero ring of the halo tile costs no memory traffic (a 4x4
//     image fetches 16 slots, not 48) and no zero page is needed.
//   * the pixel tiles live in a ring of NA slots sized to the layer's real halo (not the worst
//     case), so layers whose weights stay resident keep up to NA-1 tiles in flight -- what an
//     HBM-bound level-0 layer needs -- while weight-streaming layers double-buffer their
//     weight chunks.
//   * one barrier per (tile, chunk) stage; the loaders wait for their DMAs with counted vmcnt so
//     younger stages stay in flight across the barrier (raw s_barrier: __syncthreads would
//     drain them).
#include "hconv_impl.h"
#include <type_traits>

namespace mi {

constexpr int kDconvLds = 160 * 1024;
constexpr int kDconvAIt = 12;           // pixel-tile pieces per loader wave: ceil(768 / 16 / 4)
constexpr int kDconvThreads = 768;      // 8 compute waves + 4 loader waves

struct DconvArgs {
  HconvArgs h;
  int na;           // pixel-tile ring slots
  int a_slot;       // bytes per slot (halo slots * 64, rounded up to 1 KiB)
  int w_base;       // byte offset of the weight region in LDS
  int w_resident;   // every channel chunk of the weights stays in LDS (loaded once)
  int pa;           // DMA pieces per pixel tile = ceil(HR / 16)
  int remap;        // XCD-aware block order (gx % 8 == 0)
  int gx, ny;
  FastDiv d_ny, d_gx;
  unsigned m_grp, m_pitch;   // ceil(2^20 / ((R+2)*pitch)), ceil(2^20 / pitch): exact 24-bit division of halo slot indices
};

#define MI_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
// Waits until at most n of the wave's vector-memory operations are outstanding (n beyond the
// table waits for a few more than necessary).
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    MI_VMCNT_CASE(0) MI_VMCNT_CASE(1) MI_VMCNT_CASE(2) MI_VMCNT_CASE(3) MI_VMCNT_CASE(4) MI_VMCNT_CASE(5)
    MI_VMCNT_CASE(6) MI_VMCNT_CASE(7) MI_VMCNT_CASE(8) MI_VMCNT_CASE(9) MI_VMCNT_CASE(10) MI_VMCNT_CASE(11)
    MI_VMCNT_CASE(12) MI_VMCNT_CASE(13) MI_VMCNT_CASE(14) MI_VMCNT_CASE(15) MI_VMCNT_CASE(16) MI_VMCNT_CASE(17)
    MI_VMCNT_CASE(18) MI_VMCNT_CASE(19) MI_VMCNT_CASE(20) MI_VMCNT_CASE(21) MI_VMCNT_CASE(22) MI_VMCNT_CASE(23)
    MI_VMCNT_CASE(24) MI_VMCNT_CASE(25) MI_VMCNT_CASE(26) MI_VMCNT_CASE(27) MI_VMCNT_CASE(28) MI_VMCNT_CASE(29)
    MI_VMCNT_CASE(30) MI_VMCNT_CASE(31) MI_VMCNT_CASE(32) MI_VMCNT_CASE(33) MI_VMCNT_CASE(34) MI_VMCNT_CASE(35)
    MI_VMCNT_CASE(36) MI_VMCNT_CASE(37) MI_VMCNT_CASE(38) MI_VMCNT_CASE(39) MI_VMCNT_CASE(40) MI_VMCNT_CASE(41)
    MI_VMCNT_CASE(42) MI_VMCNT_CASE(43) MI_VMCNT_CASE(44) MI_VMCNT_CASE(45) MI_VMCNT_CASE(46) MI_VMCNT_CASE(47)
    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
  }
}

// One LDS-DMA piece through a buffer resource: lane L's 16 bytes at rsrc.base + voff go to lds + L*16.
// Lanes whose offset is out of the resource's range fetch nothing and write ZEROS (probed on gfx950):
// that is how padding slots and the halo rows beyond an image's edge are filled.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
constexpr unsigned kDmaOob = 0x80000000u;
// Value of lane+N inside the lane's 16-lane row (DPP row_shl:N); lanes whose source falls outside the row keep their own.
template <int N> __device__ __forceinline__ float row_shl(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, false));
}
// max(a, b) as ONE instruction (the median of a, b and +inf): fmaxf first canonicalises each operand that the
// compiler cannot prove quiet -- an MFMA result, a DPP move -- with a v_max of its own.  NaN operands give the other one.
// (The +inf passes through an empty asm: a constant one is folded back into the canonicalising form.)
__device__ __forceinline__ float max_raw(float a, float b) {
  float inf = __builtin_inff();
  asm("" : "+s"(inf));
  return __builtin_amdgcn_fmed3f(a, b, inf);
}
// max(v, v of lane+N inside the 16-lane row) in one DPP instruction; lanes whose source falls outside the row keep v.
// (Inline assembly is opaque to the compiler's hazard pass: the two wait states a DPP read of a freshly written
// register needs are written out.)
template <int N> __device__ __forceinline__ float max_row_shl(float v) {
  float r = v;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shl:%2 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v), "n"(N));
  return r;
}
__device__ __forceinline__ unsigned bf16x2_pack(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const bf16x2_t r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
template <typename V> __device__ __forceinline__ void pin_scalar(V v) { asm volatile("" ::"s"(v)); }
template <typename... A> __device__ __forceinline__ void pin_scalars(A... v) { (pin_scalar(v), ...); }

template <int TN, bool ROWPAR>
__global__ __launch_bounds__(kDconvThreads) void dconv_kernel(const DconvArgs d) {
  typedef __bf16 T;
  typedef bf16x8_t frag_t;
  constexpr int TM = 2, BM = 256, BN = TN * 16, CK = 32, EPC = 8;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int WP = 9 * BN / 16;                 // DMA pieces per weight chunk
  constexpr int W_IT = (WP + 3) / 4;              // per loader wave
  constexpr int A_IT = kDconvAIt;
  __shared__ __attribute__((aligned(1024))) char smem[kDconvLds];
  const HconvArgs& a = d.h;

  // every kernel argument is fetched HERE, in one batch: left to first use, the scalar loads came as
  // dependent round trips spread over the prologue (about a microsecond before the first DMA)
  pin_scalars(a.in, a.w, a.out, a.bias, a.mask, a.pool, a.zero, a.in_cs, a.in_co, a.out_cs, a.out_co, a.mask_cs, a.mask_co,
              a.pool_cs, a.pool_co, a.Kp, a.Cp, a.n_real, a.relu, a.W, a.R, a.pitch, a.HR, a.bands, a.ntiles, a.nchunks,
              a.d_bands.mul, a.d_bands.shift, a.d_rw.mul, a.d_rw.shift, a.d_w.mul, a.d_w.shift,
              d.na, d.a_slot, d.w_base, d.w_resident, d.pa, d.remap, d.gx, d.ny, d.d_ny.mul, d.d_ny.shift, d.d_gx.mul, d.d_gx.shift,
              d.m_grp, d.m_pitch, a.in2, a.chunk2, a.out2, a.n2, a.head_w, a.head_b, a.head_out, a.head_cout, a.head_hw,
              a.d_hw.mul, a.d_hw.shift);
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef MI_DCONV_DIAG
  const long long r_entry = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  int bx, by;
  if (d.remap) {      // the ny channel tiles of one pixel tile share an XCD (blocks b and b+8 do)
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
    const int kq = (int)fdiv((uint32_t)k, d.d_ny);
    by = k - kq * d.ny; bx = kq * 8 + xcd;
  } else { by = (int)fdiv(blockIdx.x, d.d_gx); bx = (int)blockIdx.x - by * d.gx; }
  const int n0 = by * BN;
  const int grp_rows = (a.R + 2) * a.pitch;
  const int my_tiles = (int)fdiv((uint32_t)(a.ntiles - bx + d.gx - 1), d.d_gx);
  const int nstages = my_tiles * a.nchunks;       // (tile, chunk) pairs of this workgroup, in order
  if (nstages <= 0) return;
  const int NA = d.na;
  const bool resident = d.w_resident != 0;

  if (wave >= 8) {
    // ================================================================ loader waves
#ifdef MI_DCONV_DIAG
    const long long r_zeroed = r_entry;
#endif
    __builtin_amdgcn_s_setprio(3);    // the loaders' few instructions go first: everything else waits for their data
    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ wp = (const T*)a.w;
    const int lw = wave - 8;
    const int l4 = lane >> 2, cpos = lane & 3;
    // Piece q of a pixel tile covers halo slots [16q, 16q+16): lane L fills chunk position L&3 of
    // slot 16q + (L>>2) with the channel chunk that the swizzle puts there.  Offsets are bytes from
    // (tile's first pixel - one image row): the top halo row of a band lies in the band above.
    const int margin = a.W * a.in_cs;
    unsigned a_voff[A_IT];            // byte offset of this lane's source, or out of range (padding slot: zeros)
    unsigned topmask = 0, botmask = 0;   // bit i: piece i of this lane is a top / bottom halo row of a band
    unsigned skipmask = 0;            // bit i (wave-uniform): piece i does not exist or is all padding (zeroed once, below)
    int cnt_a = 0;                    // pixel-tile pieces this wave issues per tile
    const unsigned uW = (unsigned)a.W, uR1 = (unsigned)a.R + 1u, RW = (unsigned)(a.R * a.W);
    const bool single = a.bands == 1;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {      // branch-free inside a round: 24-bit multiplies and selects only
      const int q = lw + 4 * i;
      // rounds beyond the tile's pieces (half of them for a 32-wide image) cost nothing: at kernel entry all
      // twelve waves of the CU run set-up arithmetic and a round takes ~370 cycles instead of ~150
      if (q >= d.pa) { a_voff[i] = kDmaOob; skipmask |= 1u << i; continue; }
      const unsigned hr = (unsigned)(q * 16 + l4);
      const unsigned g = __umul24(hr, d.m_grp) >> 20, rem = hr - __umul24(g, (unsigned)grp_rows);
      const unsigned hy = __umul24(rem, d.m_pitch) >> 20, hx = rem - __umul24(hy, (unsigned)a.pitch);
      const bool top = hy == 0, bot = hy == uR1;
      const bool on = (q < d.pa) & (hr < (unsigned)a.HR) & (hx - 1u < uW) & !((top | bot) & single);
      const int pix = (int)(__umul24(g, RW) + __umul24(hy, uW) + hx) - (int)(uW + 1u);     // (hy-1)*W + (hx-1) inside group g
      const int rel = __mul24(pix, a.in_cs) + ((cpos ^ hswz<ROWPAR>((int)hr)) * EPC);
      a_voff[i] = on ? (unsigned)(rel + margin) * 2u : kDmaOob;
      topmask |= (unsigned)(on & top) << i;
      botmask |= (unsigned)(on & bot) << i;
      const bool any = __builtin_amdgcn_ballot_w64(on) != 0;
      skipmask |= (unsigned)(!any) << i;
      cnt_a += any;
    }
#ifdef MI_DCONV_DIAG
    const long long r_offs = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // (the FIRST tile is fetched by the compute waves, which have nothing else to do at kernel entry: each works out
    // a few pieces' offsets and issues them at once, a microsecond before this wave's offset table is complete)
#ifdef MI_DCONV_DIAG
    const long long r_first = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // all-padding pieces are zeroed in every ring slot, once: no DMA ever writes these bytes
    for (int i = 0; i < A_IT; ++i)
      if (((skipmask >> i) & 1) && lw + 4 * i < d.pa) {
#pragma unroll 1
        for (int k = 0; k < NA; ++k) *(u32x4_t*)(smem + k * d.a_slot + (lw + 4 * i) * 1024 + lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
      }
    // weights of the later stages (the compute waves, idle until the first data lands, issue chunk 0 / the resident chunks)
    unsigned w_voff[W_IT];            // byte offset of this lane's weight source for channel chunk 0 (streamed weights only)
#pragma unroll
    for (int i = 0; i < W_IT; ++i) w_voff[i] = kDmaOob;
    if (!resident) {
#pragma unroll
      for (int i = 0; i < W_IT; ++i) {
        const int q = lw + 4 * i, row = q * 16 + l4;
        const int tap = row / BN, n = row - tap * BN;
        w_voff[i] = (unsigned)((n0 + n) * a.Kp + tap * a.Cp + ((cpos ^ wswz<TN>(row)) * EPC)) * 2u;
      }
    }
    auto issue_w = [&](int chunk, int wslot) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(wp + chunk * CK);
      char* dst = smem + d.w_base + wslot * W_BYTES + lw * 1024;
#pragma unroll
      for (int i = 0; i < W_IT; ++i)
        if (lw + 4 * i < WP) dma16(rs, w_voff[i], dst + i * 4096);
    };
    auto issue_a = [&](int tile, int chunk, int slot) {
      // a two-tensor input: the chunks from chunk2 on are the first chunks of the second tensor
      const T* src = (a.in2 != nullptr && chunk >= a.chunk2) ? (const T*)a.in2 + (chunk - a.chunk2) * CK : in + a.in_co + chunk * CK;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + ((int64_t)tile * BM * a.in_cs - margin));
      char* dst = smem + slot * d.a_slot + lw * 1024;
      if (a.bands == 1) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
          if (!((skipmask >> i) & 1)) dma16(rs, a_voff[i], dst + i * 4096);
      } else {
        // bands of a tall image: the halo row above the first band / below the last one is padding
        const int band = tile - (int)fdiv((uint32_t)tile, a.d_bands) * a.bands;
        const unsigned zmask = (band == 0 ? topmask : 0u) | (band == a.bands - 1 ? botmask : 0u);
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
          if (!((skipmask >> i) & 1)) dma16(rs, ((zmask >> i) & 1) ? kDmaOob : a_voff[i], dst + i * 4096);
      }
    };
    // load cursor: (tile, chunk, slot) of the next pixel tile to fetch; lstage = tiles issued so far
    int ltile = bx, lchunk = 0, lslot = 0, lstage = 0;
    auto advance_load = [&]() {
      if (++lchunk == a.nchunks) { lchunk = 0; ltile += d.gx; }
      if (++lslot == NA) lslot = 0;
      ++lstage;
    };
    // prologue, continued: the tiles that may stay in flight beyond stage 0
#ifdef MI_DCONV_DIAG
    const long long r_setup = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    advance_load();                    // tile 0 and the first weights are the compute waves' to fetch
#ifdef MI_DCONV_DIAG
    const long long r_a0 = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // Stage 0 needs nothing from this wave: it puts the tile of stage 1 on its way, joins the first barrier, and
    // fetches the other tiles that may be in flight while the compute waves run stage 0.
    if (lstage < NA - 1 && lstage < nstages) { issue_a(ltile, lchunk, lslot); advance_load(); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the one-time zero stores
    __builtin_amdgcn_s_barrier();
    while (lstage < NA - 1 && lstage < nstages) { issue_a(ltile, lchunk, lslot); advance_load(); }
#ifdef MI_DCONV_DIAG
    const long long r_issued = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#ifdef MI_DCONV_DIAG
    const long long r_landed = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    int chunk = 0;
    for (int s = 0; s < nstages; ++s) {
      // weight pieces first, pixel pieces after them: the counted wait below relies on that order
      const bool doW = !resident && s + 1 < nstages;
      const bool doA = lstage < nstages;             // the tile of stage s+NA-1 goes into the slot stage s-1 left
#if defined(MI_DCONV_X_NODMA)
      if (doA) advance_load();
#else
      if (doW) issue_w(chunk + 1 == a.nchunks ? 0 : chunk + 1, (s + 1) & 1);
      if (doA) { issue_a(ltile, lchunk, lslot); advance_load(); }
#endif
      // Everything stage s+1 reads must have landed: its pixel tile and (streaming) its weights.
      // Only the pixel tiles of later stages may stay in flight: with resident weights those of
      // stages s+2 .. s+NA-1, with streamed weights (issued after them) only this stage's own issue.
      int later = lstage - (s + 2);
      if (later < 0) later = 0;
      if (!resident && later > 1) later = 1;
      if (!resident && !doA) later = 0;
#if defined(MI_DCONV_X_NODMA)
      later = 0;
#endif
      wait_vmcnt(later * cnt_a);
      __builtin_amdgcn_s_barrier();
      if (++chunk == a.nchunks) chunk = 0;
    }
#ifdef MI_DCONV_DIAG
    if (blockIdx.x == 17 && lane == 0 && wave == 8)
      printf("dconv loader 8 (x10 ns): offsets %lld first-issue %lld | entry->zeroed %lld  setup %lld  issue A0 %lld (cnt %d)  issue W+ahead %lld  landed %lld\n", r_offs - r_entry, r_first - r_offs, r_zeroed - r_entry,
             r_setup - r_zeroed, r_a0 - r_setup, cnt_a, r_issued - r_a0, r_landed - r_issued);
#endif
    return;
  }

  // ================================================================ compute waves
  // (identical maps to hconv_kernel): fragment addresses of this lane's pixel in each of its 2 tiles, per tx
  const int r16 = lane & 15, ch = lane >> 4;
  // What stage 0 needs is fetched HERE, by the compute waves, in the order it is needed: the bias values, weight
  // chunk 0, the first pixel tile (each wave works out the offsets of a few pieces and issues them at once; the
  // loaders' complete offset tables, which serve every later tile, take 1.4 us to build), then the other resident
  // weight chunks, which may land during stage 0.
  constexpr int CPL = 4 * TN;
  float biasv[CPL];
  {
    // through a buffer resource that ends at the layer's last real channel: the loads beyond it return zero (a
    // dgrad launch, which has no bias, gets an empty resource), so there is no index to clamp and no value to
    // select -- 16 address computations, compares and selects were a fifth of this prologue's instructions
    const int nreal = a.bias != nullptr ? a.n_real : 0;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias != nullptr ? a.bias : (const float*)a.zero), 0, nreal * 4, 0x00020000);
    const int boff = (n0 + CPL * ch) * 4;
#pragma unroll
    for (int k = 0; k < CPL; ++k) biasv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, boff + 4 * k, 0, 0));
  }
  int w_later = 0;          // this wave's weight DMAs that stage 0 does not need
  {
    constexpr int W_IT8 = (WP + 7) / 8;
    const int l4 = lane >> 2, cpos = lane & 3;
    const T* __restrict__ wp = (const T*)a.w;
    unsigned wv[W_IT8];        // this lane's source offsets inside a chunk: the same for every chunk
#pragma unroll
    for (int i = 0; i < W_IT8; ++i) {
      const int q = wave + 8 * i, row = q * 16 + l4;
      const int tap = row / BN, n = row - tap * BN;
      wv[i] = q < WP ? (unsigned)((n0 + n) * a.Kp + tap * a.Cp + ((cpos ^ wswz<TN>(row)) * EPC)) * 2u : kDmaOob;
    }
    auto issue_chunk = [&](int c) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(wp + c * CK);
#pragma unroll
      for (int i = 0; i < W_IT8; ++i)
        if (wave + 8 * i < WP) dma16(rs, wv[i], smem + d.w_base + c * W_BYTES + (wave + 8 * i) * 1024);
    };
    issue_chunk(0);
    {
      // pieces wave, wave+8, ... of the first pixel tile (the loaders' arithmetic, without the bookkeeping)
      const int margin = a.W * a.in_cs;
      const unsigned uW = (unsigned)a.W, uR1 = (unsigned)a.R + 1u, RW = (unsigned)(a.R * a.W);
      const bool single = a.bands == 1;
      const int band0 = bx - (int)fdiv((uint32_t)bx, a.d_bands) * a.bands;
      const bool ztop0 = !single && band0 == 0, zbot0 = !single && band0 == a.bands - 1;
      const __amdgpu_buffer_rsrc_t rs0 = make_rsrc((const T*)a.in + ((int64_t)bx * BM * a.in_cs + a.in_co - margin));
#pragma unroll
      for (int i = 0; i < (4 * kDconvAIt + 7) / 8; ++i) {
        const int q = wave + 8 * i;
        if (q >= d.pa) break;
        const unsigned hr = (unsigned)(q * 16 + l4);
        const unsigned g = __umul24(hr, d.m_grp) >> 20, rem = hr - __umul24(g, (unsigned)grp_rows);
        const unsigned hy = __umul24(rem, d.m_pitch) >> 20, hx = rem - __umul24(hy, (unsigned)a.pitch);
        const bool top = hy == 0, bot = hy == uR1;
        const bool on = (hr < (unsigned)a.HR) & (hx - 1u < uW) & !((top | bot) & single) & !(top & ztop0) & !(bot & zbot0);
        const int pix = (int)(__umul24(g, RW) + __umul24(hy, uW) + hx) - (int)(uW + 1u);
        const int rel = __mul24(pix, a.in_cs) + ((cpos ^ hswz<ROWPAR>((int)hr)) * EPC);
        dma16(rs0, on ? (unsigned)(rel + margin) * 2u : kDmaOob, smem + q * 1024);
      }
    }
    if (resident) {
#pragma unroll 1
      for (int c = 1; c < a.nchunks; ++c) issue_chunk(c);
      int mine = 0;
#pragma unroll
      for (int i = 0; i < W_IT8; ++i) mine += (wave + 8 * i < WP);
      w_later = (a.nchunks - 1) * mine;
    }
  }
  int paddr[TM][3];
  int out_off[TM], mask_off[TM], pool_off[TM];
  int pix_in_tile[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int cpr = a.W >= 16 ? a.W >> 4 : 1;
    const int frag = a.W >= 16 ? (2 * (wave / cpr) + j) * cpr + (wave % cpr) : wave * TM + j;
    const int pl = frag * 16 + r16;
    const int g = (int)fdiv((uint32_t)pl, a.d_rw), rem = pl - g * (a.R * a.W);
    const int y = (int)fdiv((uint32_t)rem, a.d_w), x = rem - y * a.W;
    const int hrow = g * grp_rows + y * a.pitch + x;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int hr = hrow + tx;
      paddr[j][tx] = hr * 64 + ((ch ^ hswz<ROWPAR>(hr)) << 4);
    }
    pix_in_tile[j] = pl;
    out_off[j] = pl * a.out_cs + a.out_co + n0 + 4 * TN * ch;
    mask_off[j] = pl * a.mask_cs + a.mask_co + n0 + 4 * TN * ch;
    const int pooled = g * (a.R >> 1) * (a.W >> 1) + (y >> 1) * (a.W >> 1) + (x >> 1);
    pool_off[j] = ((y | x) & 1) ? -1 : pooled * a.pool_cs + a.pool_co + n0 + 4 * TN * ch;
  }
  int wfrag[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = nmap<TN>(i, r16);
    wfrag[i] = n * 64 + ((ch ^ wswz<TN>(n)) << 4);
  }
  const int ty_bytes = a.pitch * 64;

  // fused 1x1 head (32-channel tile: TN == 2, a lane holds 8 consecutive channels of its pixel): this lane's weights
  constexpr int HC = 2;                 // head outputs at most
  float hwv[HC][8];
  float hbv[HC];
  if (TN == 2 && a.head_out != nullptr) {
#pragma unroll
    for (int co = 0; co < HC; ++co) {
      const bool live = co < a.head_cout;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = 8 * ch + q;
        const float wv = a.head_w[(live && c < a.n_real) ? co * a.n_real + c : 0];
        hwv[co][q] = (live && c < a.n_real) ? wv : 0.f;
      }
      const float bv = a.head_b[live ? co : 0];
      hbv[co] = live ? bv : 0.f;
    }
  }
  f32x4_t acc[TN][TM];
  // the accumulators start at the bias (zero for dgrad): the epilogue, which is VALU-bound (every compute wave of
  // the CU converts and stores at once, ~3,500 cycles per tile measured), saves one add per output
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{biasv[4 * i], biasv[4 * i + 1], biasv[4 * i + 2], biasv[4 * i + 3]};
  };
  // a two-tensor output: the lanes whose channels lie beyond n2 write the second tensor (a lane's run of
  // channels never straddles n2: both are multiples of 16)
  T* __restrict__ out = (a.out2 != nullptr && n0 + 4 * TN * ch >= a.n2) ? (T*)a.out2 - a.n2 - a.out_co : (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
  // ReLU mask of the tile (dgrad): fetched at the START of the tile's last stage, so the loads land
  // behind that stage's MFMAs instead of stalling the epilogue for a round trip to memory
  constexpr int MV = CPL / 8 > 0 ? CPL / 8 : 1;      // 16-byte vectors per pixel of this lane (CPL >= 8: TN >= 2)
  bf16x8_t mreg[TM][MV];
  auto prefetch_mask = [&](int tile) {
    const T* mtile = mask + (int64_t)tile * BM * a.mask_cs;
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int k = 0; k < MV; ++k) {
        if (CPL >= 8) mreg[j][k] = *(const bf16x8_t*)(mtile + mask_off[j] + 8 * k);
        else { const bf16x4_t h = *(const bf16x4_t*)(mtile + mask_off[j]); mreg[j][k] = bf16x8_t{h[0], h[1], h[2], h[3], h[0], h[1], h[2], h[3]}; }
      }
  };
  auto epilogue = [&](int tile) {
    T* otile = out + (int64_t)tile * BM * a.out_cs;
    if (mask != nullptr) {
      // dgrad through a ReLU (models/unet2d.py:30, :40, :53): the gradient survives where the stored activation is
      // positive.  Done on PAIRS -- both values rounded by one packed conversion, the pair's keep bits from two
      // packed 16-bit operations on the activations' bit patterns (0 - m, saturating, is negative exactly for
      // m > 0; its sign, smeared over the half, is the keep mask) -- 2 VALU operations per value; comparing each
      // activation as a float and selecting each value came to 4.5.
      typedef __attribute__((ext_vector_type(2))) short s16x2_t;
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        uint32_t o[CPL / 2];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = 4 * i + 2 * h;       // the pair's first channel inside this lane's run
            const s16x8_t mw = __builtin_bit_cast(s16x8_t, mreg[j][e / 8]);
            const s16x2_t m = {mw[e % 8], mw[e % 8 + 1]};
            const s16x2_t keep = __builtin_elementwise_sub_sat(s16x2_t{0, 0}, m) >> 15;
            const bf16x2_t r = {(T)acc[i][j][2 * h], (T)acc[i][j][2 * h + 1]};
            o[e / 2] = __builtin_bit_cast(uint32_t, r) & __builtin_bit_cast(uint32_t, keep);
          }
        T* dst = otile + out_off[j];
        if (CPL >= 8) {
#pragma unroll
          for (int k = 0; k < CPL / 8; ++k) *(u32x4_t*)(dst + 8 * k) = u32x4_t{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
        } else {
          typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
          *(u32x2_t*)dst = u32x2_t{o[0], o[1]};
        }
      }
      return;
    }
    // Two copies of the store loop behind ONE wave-uniform branch, the ReLU as one operation per value: written as
    // `a.relu ? fmaxf(x, 0.f) : x` it compiled to three VALU operations per value (fmaxf canonicalises its operand
    // with a max of its own, and the choice became a per-value select) in an epilogue that is VALU-bound.
    auto store_tile = [&](auto relu_c) {
      constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float v[CPL];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * i + q] = RELU ? max_raw(acc[i][j][q], 0.f) : acc[i][j][q];
          }
#ifdef MI_DCONV_X_NOSTORE
#pragma unroll
        for (int k = 0; k < CPL; k += 2) { const float pk = __builtin_bit_cast(float, bf16x2_pack(v[k], v[k + 1])); asm volatile("" ::"v"(pk)); }
#else
        store_run(otile + out_off[j], v, CPL);
#endif
        if (RELU && TN == 2 && a.head_out != nullptr) {     // (the launcher takes a head only behind a ReLU layer)
          // models/unet2d.py:89 on the values just stored: 8 channels per lane in the stand-alone kernel's order, then the
          // same butterfly over the pixel's four lanes (lane bits 4 and 5 here)
          const uint32_t p = (uint32_t)tile * BM + (uint32_t)pix_in_tile[j];
          const uint32_t b = fdiv(p, a.d_hw), rr = p - b * (uint32_t)a.head_hw;
#pragma unroll
          for (int co = 0; co < HC; ++co) {
            if (co >= a.head_cout) break;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s = fmaf((float)(T)v[q], hwv[co][q], s);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (ch == 0) a.head_out[((int64_t)b * a.head_cout + co) * a.head_hw + rr] = s + hbv[co];
          }
        }
      }
    };
    if (a.relu) store_tile(std::true_type{});
    else store_tile(std::false_type{});
    if (a.pool != nullptr) {
      // MaxPool2d(2,2) of the tile fused here (models/unet2d.py:68): the window partners are other
      // lanes of the same wave (column: lane^1; row: the wave's other fragment when W >= 16, lane^W
      // when a fragment holds two or four rows).  max commutes with the rounding to the storage type.
      T* ptile = (T*)a.pool + (int64_t)tile * (BM / 4) * a.pool_cs;
      // the ReLU goes on the window's maximum (it is monotone: the same value as the maximum of the ReLUs)
      auto val = [&](int i, int j, int q) { return acc[i][j][q]; };
      const float floor_v = a.relu ? 0.f : -__builtin_inff();
      // Only the lane of a window's even row and even column stores, and its partners sit at lane+1
      // and lane+W inside its 16-lane row: two DPP row shifts (VALU) instead of LDS-routed shuffles.
      if (a.W >= 16) {
        float m[CPL];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float tt = max_raw(val(i, 0, q), val(i, 1, q));
            m[4 * i + q] = max_raw(max_row_shl<1>(tt), floor_v);
          }
        if (pool_off[0] >= 0) store_run(ptile + pool_off[0], m, CPL);
      } else {
        // one copy of the loop per image width: a per-value choice of the shift became a branch per value
        auto pool_rows = [&](auto down) {
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            float m[CPL];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float tt = down(max_raw(val(i, j, q), floor_v));    // (the max first: the DPP source must be a VALU result)
                m[4 * i + q] = max_row_shl<1>(tt);
              }
            if (pool_off[j] >= 0) store_run(ptile + pool_off[j], m, CPL);
          }
        };
        if (a.W == 8) pool_rows([](float x) { return max_row_shl<8>(x); });
        else pool_rows([](float x) { return max_row_shl<4>(x); });
      }
    }
  };

#ifdef MI_DCONV_DIAG
  const long long r_ready = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  wait_vmcnt(w_later);              // this wave's share of weight chunk 0 and of the first pixel tile (and its bias values)
  __builtin_amdgcn_s_barrier();     // stage 0 has landed
#ifdef MI_DCONV_DIAG
  const long long r_start = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  zero_acc();
  int tile = bx, chunk = 0, aslot = 0;
#ifdef MI_DCONV_DIAG
  long long c_taps = 0, c_epi = 0, c_bar = 0, t_begin = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int s = 0; s < nstages; ++s) {
#ifdef MI_DCONV_DIAG
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
    const char* Ab = smem + aslot * d.a_slot;
    const char* Wb = smem + d.w_base + (resident ? chunk : (s & 1)) * W_BYTES;
    if (mask != nullptr && chunk == a.nchunks - 1) prefetch_mask(tile);
    // fragments of tap t+1 are read while the MFMAs of tap t issue (two register sets)
    frag_t pf[2][TM], wf[2][TN];
    auto read_tap = [&](int tap, int set) {
      const int ty = tap / 3, tx = tap % 3;
      const char* Sy = Ab + ty * ty_bytes;
      const int flip = (ROWPAR && (ty & 1)) ? 32 : 0;
#pragma unroll
      for (int j = 0; j < TM; ++j) pf[set][j] = *(const frag_t*)(Sy + (paddr[j][tx] ^ flip));
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[set][i] = *(const frag_t*)(Wb + tap * BN * 64 + wfrag[i]);
    };
#if defined(MI_DCONV_X_NOREAD)
    read_tap(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma_slab(wf[0][i], pf[0][j], acc[i][j]);
    }
#elif defined(MI_DCONV_X_NOMFMA)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      read_tap(tap, tap & 1);
#pragma unroll
      for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(pf[tap & 1][j]));
#pragma unroll
      for (int i = 0; i < TN; ++i) asm volatile("" ::"v"(wf[tap & 1][i]));
    }
    acc[0][0] = mma_slab(wf[0][0], pf[0][0], acc[0][0]);
#else
    read_tap(0, 0);
#ifndef MI_DCONV_NO_SCHED
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);     // tap 0's own reads come first
#endif
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) read_tap(tap + 1, (tap + 1) & 1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma_slab(wf[tap & 1][i], pf[tap & 1][j], acc[i][j]);
#ifndef MI_DCONV_NO_SCHED
      // The order is spelled out for the scheduler: one fragment read of the NEXT tap in front of each of this tap's
      // first MFMAs.  Left alone it folds the two register sets into one and issues every read just before its use
      // (read, wait, MFMA: six waits per tap that only the SIMD's other wave covers).
      if (tap + 1 < 9) {
#pragma unroll
        for (int k = 0; k < TM + TN; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one LDS read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
#endif
    }
#endif
#ifdef MI_DCONV_DIAG
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
#endif
    if (chunk == a.nchunks - 1) {
      epilogue(tile);
      zero_acc();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (s == 0 && w_later > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the resident weight chunks stage 0 did not need
#ifdef MI_DCONV_DIAG
    const long long t2 = (long long)__builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef MI_DCONV_DIAG
    { const long long t3 = (long long)__builtin_amdgcn_s_memtime(); c_taps += t1 - t0; c_epi += t2 - t1; c_bar += t3 - t2; }
#endif
    if (++chunk == a.nchunks) { chunk = 0; tile += d.gx; }
    if (++aslot == NA) aslot = 0;
  }
#ifdef MI_DCONV_DIAG
  if (blockIdx.x == 17 && lane == 0 && (wave == 0 || wave == 5)) {
    const long long t_end = (long long)__builtin_amdgcn_s_memtime();
    const long long r_end = (long long)__builtin_amdgcn_s_memrealtime();
    printf("dconv stamp wave %d: stages %d total %lld  taps %lld epi %lld barrier %lld (cycles, per stage %lld); x10ns: entry->ready %lld ready->start %lld loop %lld\n", wave, nstages,
           t_end - t_begin, c_taps, c_epi, c_bar, (t_end - t_begin) / nstages, r_ready - r_entry, r_start - r_ready, r_end - r_start);
  }
#endif
}

// Launches the DMA-fed kernel when the layer suits it; returns -1000 when the caller should use
// the register-staged kernel instead (an unsuitable shape).  dry_run: decide only, launch nothing.
int launch_dconv_bf16(const HconvArgs& a_in, hipStream_t s, const char** name, bool dry_run) {
  DconvArgs d{};
  d.h = a_in;
  HconvArgs& a = d.h;
  a.d_grp = make_fastdiv((a.R + 2) * a.pitch); a.d_pitch = make_fastdiv(a.pitch); a.d_rw = make_fastdiv(a.R * a.W);
  a.d_w = make_fastdiv(a.W); a.d_bands = make_fastdiv(a.bands);
  const bool rowpar = a.pitch == 8;                 // 4-wide images: the row-parity swizzle
  if (a.HR > 768 || (a.HR > kHconvMaxHalo && !rowpar)) return -1000;
  if (a.out2 != nullptr && (a.n2 % 16 != 0 || a.pool != nullptr || a.mask != nullptr)) return -1000;
  if (a.head_out != nullptr && (a.N != 32 || a.mask != nullptr || a.head_cout < 1 || a.head_cout > 2)) return -1000;
  // channel tile: 64 where that still gives every CU a workgroup, else 32 (16 for narrow layers)
  int bn = a.N % 64 == 0 ? 64 : (a.N % 32 == 0 ? 32 : 16);
  if (bn == 64 && a.ntiles * (a.N / 64) < 256 && a.ntiles * (a.N / 32) <= 512) bn = 32;
  d.pa = (a.HR + 15) / 16;
  d.a_slot = d.pa * 1024;
  for (;;) {
    const int w_bytes = 9 * bn * 64;
    // weights: resident when all chunks fit beside at least two pixel-tile slots, else two streaming slots
    d.w_resident = (int64_t)a.nchunks * w_bytes + 2 * d.a_slot <= kDconvLds;
    const int w_region = d.w_resident ? a.nchunks * w_bytes : 2 * w_bytes;
    d.na = (kDconvLds - w_region) / d.a_slot;
    if (d.na > 6) d.na = 6;
    if (d.na >= 2) break;
    if (bn == 16) return -1000;
    bn /= 2;                          // a narrower channel tile leaves room for two pixel-tile slots
  }
  d.ny = a.N / bn;
  d.w_base = d.na * d.a_slot;
  int gx = a.ntiles;
  const int cap = (256 + d.ny - 1) / d.ny;
  if (gx > cap) gx = cap;
  d.gx = gx;
  d.remap = gx % 8 == 0;
  d.d_ny = make_fastdiv(d.ny); d.d_gx = make_fastdiv(gx);
  d.m_grp = ((1u << 20) + (a.R + 2) * a.pitch - 1) / ((a.R + 2) * a.pitch);
  d.m_pitch = ((1u << 20) + a.pitch - 1) / a.pitch;
  if (dry_run) return MI_OK;                        // the caller only asks whether this kernel takes the layer
  dim3 grid(gx * d.ny);
#define MI_DCONV(TNV, RP, NAME) { if (name) *name = NAME; hipLaunchKernelGGL((dconv_kernel<TNV, RP>), grid, dim3(kDconvThreads), 0, s, d); }
  if (rowpar) {
    if (bn == 64) MI_DCONV(4, true, "dconv_256x64_rp") else if (bn == 32) MI_DCONV(2, true, "dconv_256x32_rp") else MI_DCONV(1, true, "dconv_256x16_rp")
  } else {
    if (bn == 64) MI_DCONV(4, false, "dconv_256x64") else if (bn == 32) MI_DCONV(2, false, "dconv_256x32") else MI_DCONV(1, false, "dconv_256x16")
  }
#undef MI_DCONV
  return check_hip(hipGetLastError(), "dconv launch");
}

}  // namespace mi
