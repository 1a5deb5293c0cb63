// Halo-tile weight-gradient kernel for gfx950 in fp32: dW of the reference's Conv2d(k=3, padding=1)
// layers (models/unet2d.py:28-31, :38-41, :51-54), i.e. their autograd backward w.r.t. the weights,
// for images that tile evenly into 128-pixel blocks.
//
//   part[s][n][tap*Cb + c] = sum_{p in pixel range s} dy[p][n] * x[p + tap][c]
//
// The generic kernel (wgrad_impl.h) gathers x once per tap: nine reads of every activation, at the
// rate a CU takes operands in.  Here a 512-thread workgroup owns 32 dy-channels x 32 x-channels x all
// 9 taps and walks its share of the pixel tiles; per tile it stages the dy tile [128 px][32] and ONE
// zero-ringed halo tile of x [(rows+2) x (W+2)][32] in LDS (global -> registers -> LDS, the next
// tile's loads in flight under this tile's MFMAs), and the 9 taps read their fragments from that
// halo tile at constant row offsets.  Rows are 128 bytes (32 floats): a ds_read_b32 fragment read
// serves two consecutive rows per half-wave, which never share a bank -- no swizzle.
// v_mfma_f32_16x16x4_f32 takes one float per lane and operand (k = 4 pixels per step).
// Each wave keeps 2 n-tiles x 1 c-tile x 9 taps of accumulators; the four waves that share an
// accumulator block split the tile's pixels and are summed through LDS at the end in a fixed order,
// so a workgroup writes one partial slab; the slabs of the pixel splits are summed by the
// fixed-order combine pass (bitwise repeatable, no atomics).
#include "wgrad_impl.h"

namespace mi {

namespace {

constexpr int kP = 128;                       // pixels per tile
constexpr int kMaxHalo = 392;                 // halo rows the LDS buffer holds (one row of a 128-wide image: 3 x 130)
constexpr int kXB = kMaxHalo * 128;
constexpr int kDYB = kP * 128;
constexpr int kStage = kXB + kDYB;
constexpr int kItX = (kMaxHalo * 8 + 511) / 512;   // 16-byte chunks of the halo tile per thread, at most

struct Geom { int R, G, bands, HRt, Wp; };
bool tiling(int H, int W, Geom* g) {
  if (W < 4 || W % 4 != 0 || H < 1) return false;
  const int HW = H * W;
  if (HW >= kP) {
    if (kP % W != 0) return false;
    g->R = kP / W;
    if (H % g->R != 0) return false;
    g->G = 1; g->bands = H / g->R;
  } else {
    if (kP % HW != 0) return false;
    g->R = H; g->G = kP / HW; g->bands = 1;
  }
  g->Wp = W + 2;
  g->HRt = g->G * (g->R + 2) * g->Wp;
  return g->HRt <= kMaxHalo;
}

struct Args {
  WgradArgs w;
  int R, G, bands, HRt, Wp, xcd_deal;
  FastDiv d_img, d_wp, d_rw, d_w, d_bands;     // halo rows per image, W+2, R*W, W, bands
};

template <int DUMMY>
__global__ __launch_bounds__(512) void hwgrad_f32_kernel(const Args ha) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kStage];
  const WgradArgs& a = ha.w;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int gx = a.Na / 32, blocks = gx * (a.Cb / 32);
  // id = slot*8 + xcd: the (n-block, c-block) workgroups of one pixel split sit on one XCD and share its L2 -- where
  // there are at least 8 splits to deal and no XCD gets more workgroups than it has CUs; else the ids run split by split
  int z, mem;
  if (ha.xcd_deal) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    z = (slot / blocks) * 8 + xcd; mem = slot % blocks;
  } else {
    z = blockIdx.x / blocks; mem = blockIdx.x - z * blocks;
  }
  if (z >= a.splits) return;
  const int by = mem / gx, bx = mem - by * gx;
  const int n0 = bx * 32, c0 = by * 32;
  const int K = 9 * a.Cb, W = a.Wb;
  const int tps = a.pix_per_split / kP;
  const int t_begin = z * tps, t_end = min(a.M / kP, t_begin + tps);
  if (t_begin >= t_end) return;                 // (cannot happen: the splits are sized to the tiles)
  const float* __restrict__ pdy = (const float*)a.a + a.a_co + n0;
  const float* __restrict__ px = (const float*)a.b + a.b_co + c0;
  const int64_t zoff_x = ((intptr_t)a.zero - (intptr_t)px) / 4;

  // ---- loader tables: element offsets relative to the tile's first pixel
  const int ch = t & 7;
  int dyoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) dyoff[i] = ((t >> 3) + 64 * i) * a.a_cs + ch * 4;
  int xoff[kItX];
  unsigned okmask = 0, topmask = 0, botmask = 0, inmask = 0;     // inmask: the row exists in the halo tile (it is stored)
  const int img_rows = (ha.R + 2) * ha.Wp;
#pragma unroll
  for (int i = 0; i < kItX; ++i) {
    const int row = (t >> 3) + 64 * i;
    const unsigned g = fdiv((uint32_t)row, ha.d_img), rem = (uint32_t)row - g * (uint32_t)img_rows;
    const unsigned hy = fdiv(rem, ha.d_wp), hx = rem - hy * (uint32_t)ha.Wp;
    const bool top = hy == 0, bot = hy == (unsigned)ha.R + 1u;
    // between whole images the ring rows are always zero; inside a band they are the neighbouring image rows
    const bool ok = row < ha.HRt && hx - 1u < (unsigned)W && !((top | bot) && ha.G > 1);
    xoff[i] = ((int)(g * (unsigned)(ha.R * W)) + ((int)hy - 1) * W + (int)hx - 1) * a.b_cs + ch * 4;
    okmask |= (unsigned)ok << i;
    inmask |= (unsigned)(row < ha.HRt) << i;
    topmask |= (unsigned)(ok & top) << i;
    botmask |= (unsigned)(ok & bot) << i;
  }

  u32x4_t rdy[2], rx[kItX];
  auto load_tile = [&](int tile) {
    const int64_t pix0 = (int64_t)tile * kP;
#pragma unroll
    for (int i = 0; i < 2; ++i) rdy[i] = *(const u32x4_t*)(pdy + pix0 * a.a_cs + dyoff[i]);
    const int band = tile - (int)fdiv((uint32_t)tile, ha.d_bands) * ha.bands;
    const unsigned zmask = (band == 0 ? topmask : 0u) | (band == ha.bands - 1 ? botmask : 0u);
    const unsigned live = okmask & ~zmask;
#pragma unroll
    for (int i = 0; i < kItX; ++i) {
      if ((inmask >> i) & 1u) {
        const int64_t off = ((live >> i) & 1u) ? pix0 * a.b_cs + xoff[i] : zoff_x;
        rx[i] = *(const u32x4_t*)(px + opaque_offset(off));
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* S = smem + buf * kStage;
#pragma unroll
    for (int i = 0; i < kItX; ++i)
      if ((inmask >> i) & 1u) *(u32x4_t*)(S + ((t >> 3) + 64 * i) * 128 + ch * 16) = rx[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4_t*)(S + kXB + ((t >> 3) + 64 * i) * 128 + ch * 16) = rdy[i];
  };

  // ---- wave role and fragment addresses
  const int ct = wave & 1, pg = wave >> 1;
  const int i16 = lane & 15, kq = lane >> 4;
  int hb[8];                                   // halo-tile byte address of this lane's pixel at tap (0,0), per k-step
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const unsigned p = (unsigned)(32 * pg + 4 * ks + kq);
    const unsigned g = fdiv(p, ha.d_rw), rem = p - g * (unsigned)(ha.R * W);
    const unsigned y = fdiv(rem, ha.d_w), x = rem - y * (unsigned)W;
    hb[ks] = (int)(g * (unsigned)img_rows + y * (unsigned)ha.Wp + x) * 128 + (ct * 16 + i16) * 4;
  }
  const int dyb = kXB + (32 * pg + kq) * 128 + i16 * 4;
  const int ty_bytes = ha.Wp * 128;

  f32x4_t acc[9][2], accr[2];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) { acc[tp][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[tp][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  accr[0] = accr[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool do_rows = a.rowsum != nullptr && by == 0 && ct == 0;

  auto compute = [&](int buf) {
    const char* S = smem + buf * kStage;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float fa0 = *(const float*)(S + dyb + ks * 4 * 128);
      const float fa1 = *(const float*)(S + dyb + ks * 4 * 128 + 64);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const float fb = *(const float*)(S + hb[ks] + (tp / 3) * ty_bytes + (tp % 3) * 128);
        acc[tp][0] = mma_px(fa0, fb, acc[tp][0]);
        acc[tp][1] = mma_px(fa1, fb, acc[tp][1]);
      }
      if (do_rows) {
        accr[0] = mma_px(fa0, 1.0f, accr[0]);
        accr[1] = mma_px(fa1, 1.0f, accr[1]);
      }
    }
  };

  load_tile(t_begin);
  store_tile(0);
  __syncthreads();
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int buf = (tile - t_begin) & 1;
    const bool more = tile + 1 < t_end;
    if (more) load_tile(tile + 1);
    compute(buf);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- combine the four pixel groups of every accumulator block through LDS, upper half into lower
  // half, in a fixed order; the staging buffers are free after the last barrier
  constexpr int RV = 9 * 2 * 4 + 8;            // accumulator floats per lane
  static_assert(2 * 2 * RV * 256 <= 2 * kStage, "combine buffer must fit in the staging area");
  float* red = (float*)smem;
#pragma unroll
  for (int half = 2; half >= 1; half >>= 1) {
    if (pg >= half && pg < 2 * half) {
      float* dst = red + ((pg - half) * 2 + ct) * (RV * 64) + lane;
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
      const float* src = red + (pg * 2 + ct) * (RV * 64) + lane;
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

  // ---- the workgroup's partial slab: D row = n (4*(lane>>4)+j), col = c (lane & 15)
  float* part = a.part + (int64_t)z * a.Na * K;
  const int c = c0 + ct * 16 + i16;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + nt * 16 + 4 * kq + j;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) part[(int64_t)n * K + tp * a.Cb + c] = acc[tp][nt][j];
    }
  if (do_rows && i16 == 0) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) a.rowsum[(int64_t)z * a.Na + n0 + nt * 16 + 4 * kq + j] = accr[nt][j];
  }
}

}  // namespace

// Shapes the fp32 halo-tile kernel takes: images that tile into 128-pixel blocks (bands of an image, or
// whole images), 32-channel blocks on both operands, 32-bit element offsets.
bool hwgrad_f32_applies(int B, int H, int W, int Na, int Cb, int a_cs, int b_cs) {
  Geom g;
  if (!tiling(H, W, &g)) return false;
  const int64_t M = (int64_t)B * H * W;
  if (M % kP != 0 || Na % 32 != 0 || Cb % 32 != 0) return false;
  return M * (a_cs > b_cs ? a_cs : b_cs) < (1ll << 31);
}
// Pixel splits: one round of about 256 workgroups ((Na/32) x (Cb/32) of them per split).
int hwgrad_f32_splits(int B, int H, int W, int Na, int Cb, int* pix_per_split) {
  const int blocks = (Na / 32) * (Cb / 32);
  const int ntiles = (int)((int64_t)B * H * W / kP);
  int splits = 256 / blocks;
  if (splits < 1) splits = 1;
  if (splits > ntiles) splits = ntiles;
  const int tps = (ntiles + splits - 1) / splits;
  *pix_per_split = tps * kP;
  return (ntiles + tps - 1) / tps;
}

int launch_hwgrad_f32(const WgradArgs& w, hipStream_t s, const char** name) {
  Args ha{};
  ha.w = w;
  Geom g;
  MI_REQUIRE(tiling(w.Hb, w.Wb, &g) && w.M % kP == 0 && w.pix_per_split % kP == 0 && w.Na % 32 == 0 && w.Cb % 32 == 0 &&
                 w.colsum == nullptr && w.mode == G_3x3,
             "hwgrad_f32: shape not supported");
  ha.R = g.R; ha.G = g.G; ha.bands = g.bands; ha.HRt = g.HRt; ha.Wp = g.Wp;
  ha.d_img = make_fastdiv((g.R + 2) * g.Wp); ha.d_wp = make_fastdiv(g.Wp); ha.d_rw = make_fastdiv(g.R * w.Wb);
  ha.d_w = make_fastdiv(w.Wb); ha.d_bands = make_fastdiv(g.bands);
  const int blocks = (w.Na / 32) * (w.Cb / 32);
  if (name) *name = "hwgrad_f32_32x32x9";
  ha.xcd_deal = w.splits >= 8 && (w.splits + 7) / 8 * blocks <= 32;      // no XCD (32 CUs) is dealt a second round
  const int grid = ha.xcd_deal ? (w.splits + 7) / 8 * 8 * blocks : w.splits * blocks;
  hipLaunchKernelGGL(hwgrad_f32_kernel<0>, dim3(grid), dim3(512), 0, s, ha);
  return check_hip(hipGetLastError(), "hwgrad_f32 launch");
}

}  // namespace mi
