// Position-major 3x3 convolution for the deepest levels of the U-Net (images of at most 4x4
// pixels: the last encoder level, the bottleneck and the first decoder block of
// models/unet2d.py:28-31 of the reference at 32x32 input), forward and dgrad.
//
// On such images most taps of most pixels fall on the zero padding (5 of 9 on a 2x2 image), and
// a pixel-major implicit GEMM multiplies all those zeros.  Here one workgroup owns ONE pixel
// position (y,x) of 64 consecutive images and 64 output channels, so the set of in-bounds taps is
// the same for every row of the tile and the K loop runs over the valid taps only:
//
//   out[n][y][x][co] = epi( sum_{tap valid at (y,x)} sum_c in[n][y+dy][x+dx][c] * w[co][tap*Cp + c] )
//
// The layers are short on pixels and long on K, so one workgroup per CU with a single prefetched
// stage is latency-bound.  KG groups of 4 waves split the K loop of the tile between them (each
// with its own double-buffered LDS stage, which multiplies the bytes in flight per CU by KG) and
// combine their accumulators through LDS in a fixed order before the epilogue.
//
// Workgroups are numbered so that every XCD (blockIdx.x % 8) works on one contiguous range of
// units ordered channel-tile-major: the weight slice a tile needs is then shared by the
// workgroups of one L2 rather than streamed through all eight.
#pragma once
#include "igemm_impl.h"
#include <type_traits>

namespace mi {

struct PconvArgs {
  const void* in; int in_cs, in_co;
  const void* w; int Kp, Cp;            // packed weights [N][Kp], Cp channels per tap
  int N; const float* bias; int n_real; int relu;
  void* out; int out_cs, out_co;
  const void* mask; int mask_cs, mask_co;
  int B, H, W;
  int img_tiles, n_tiles;               // ceil(B/64), N/64
  int units;                            // H*W*img_tiles*n_tiles
};

constexpr int kPconvMaxHW = 16;

template <typename T, int KG>
__global__ __launch_bounds__(256 * KG) void pconv_kernel(const PconvArgs a) {
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr int EPC = ElemTraits<T>::EPC;
  constexpr int SK = 8 * EPC;                    // k-elements per stage (two 64-byte slabs per row)
  constexpr int OP_BYTES = 64 * 128;             // one operand tile of one stage
  constexpr int STAGE = 2 * OP_BYTES;
  constexpr int RP = 80;                         // floats per row of the combine buffer
  constexpr int NT = 256 * KG;
  static_assert(2 * STAGE >= 64 * RP * 4, "combine buffer must fit in a group's staging area");
  __shared__ __attribute__((aligned(16))) char smem[KG * 2 * STAGE];

  // unit of this workgroup: XCD x (= blockIdx.x % 8) owns units [x*per, (x+1)*per)
  const int per = gridDim.x >> 3;
  const int u = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (u >= a.units) return;
  const int HW = a.H * a.W;
  const int per_n = HW * a.img_tiles;
  const int ntile = u / per_n;
  const int rem = u - ntile * per_n;
  const int pos = rem / a.img_tiles;
  const int img0 = (rem - pos * a.img_tiles) * 64;
  const int n0 = ntile * 64;
  const int y = pos / a.W, x = pos - y * a.W;
  // in-bounds taps of this position: dy in [dy0, dy0+ny), dx in [dx0, dx0+nx)
  const int dy0 = y > 0 ? -1 : 0, dy1 = y < a.H - 1 ? 1 : 0;
  const int dx0 = x > 0 ? -1 : 0, dx1 = x < a.W - 1 ? 1 : 0;
  const int nx = dx1 - dx0 + 1;
  const int ntaps = (dy1 - dy0 + 1) * nx;
  const int cps = a.Cp / SK;                     // stages per tap
  const int nstages = ntaps * cps;

  const int t = threadIdx.x, g = t >> 8, tg = t & 255, lane = t & 63, wq = tg >> 6;
  const int wm = wq & 1, wn = wq >> 1;
  const int sub = tg & 7, row0 = tg >> 3;
  const T* __restrict__ in = (const T*)a.in;
  const T* __restrict__ wp = (const T*)a.w;
  char* gs = smem + g * (2 * STAGE);

  int64_t abase[2], wbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + i * 32;
    int n = img0 + row;
    n = n < a.B ? n : a.B - 1;                   // rows past the batch re-read the last image, never stored
    abase[i] = (int64_t)n * HW * a.in_cs + a.in_co + sub * EPC;
    wbase[i] = (int64_t)(n0 + row) * a.Kp + sub * EPC;
  }

  // two register sets: the loads of the stage after next are in flight while this stage computes (one set, with
  // the load issued at the top of an iteration and stored to LDS at its end, made every iteration as long as a memory
  // round trip: ~1 us under load for 8 MFMAs per wave)
  u32x4_t ra2[2][2], rb2[2][2];
  auto load_stage = [&](int s, int set) {
    u32x4_t* ra = ra2[set];
    u32x4_t* rb = rb2[set];
    const int ti = s / cps, cc = s - ti * cps;
    const int tyi = ti / nx;
    const int dy = dy0 + tyi, dx = dx0 + (ti - tyi * nx);
    const int tap = (dy + 1) * 3 + dx + 1;
    const int64_t aoff = (int64_t)((y + dy) * a.W + x + dx) * a.in_cs + cc * SK;
    const int64_t woff = (int64_t)tap * a.Cp + cc * SK;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *(const u32x4_t*)(in + abase[i] + aoff);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = *(const u32x4_t*)(wp + wbase[i] + woff);
  };
  auto store_stage = [&](int buf, int set) {
    const u32x4_t* ra = ra2[set];
    const u32x4_t* rb = rb2[set];
    char* A = gs + buf * STAGE;
    char* Bt = A + OP_BYTES;
    const int s = sub >> 2, c = sub & 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = row0 + i * 32;
      const int o = s * (64 * 64) + row * 64 + ((c ^ (((row >> 2) & 1) << 1)) << 4);
      *(u32x4_t*)(A + o) = ra[i];
      *(u32x4_t*)(Bt + o) = rb[i];
    }
  };

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, ch = lane >> 4;
  const int frag_off = r16 * 64 + ((ch ^ (((r16 >> 2) & 1) << 1)) << 4);
  auto compute = [&](int buf) {
    const char* A = gs + buf * STAGE;
    const char* Bt = A + OP_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t pf[2], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) pf[j] = *(const frag_t*)(A + s * (64 * 64) + (wm * 2 + j) * 1024 + frag_off);
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *(const frag_t*)(Bt + s * (64 * 64) + (wn * 2 + i) * 1024 + frag_off);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mma_slab(wf[i], pf[j], acc[i][j]);
    }
  };

  // group g takes stages g, g+KG, ...; every group runs the same number of barriers
  const int niter = (nstages + KG - 1) / KG;
  int s = g;
  if (s < nstages) load_stage(s, 0);
  if (s + KG < nstages) load_stage(s + KG, 1);
  if (s < nstages) store_stage(0, 0);
  __syncthreads();
  // iteration `it` computes stage s from LDS buffer it&1, issues the loads of stage s+2*KG into the register set
  // that buffer was filled from, and parks stage s+KG (register set (it+1)&1, loaded an iteration ago) in the other buffer
  auto iteration = [&](int it, auto par) {
    constexpr int P = decltype(par)::value;            // it & 1
    if (s + 2 * KG < nstages) load_stage(s + 2 * KG, P);
    if (s < nstages) compute(P);
    if (s + KG < nstages) store_stage(1 - P, 1 - P);
    s += KG;
    __syncthreads();
  };
  for (int it = 0; it < niter; it += 2) {
    iteration(it, std::integral_constant<int, 0>{});
    if (it + 1 < niter) iteration(it + 1, std::integral_constant<int, 1>{});
  }

  // combine the KG accumulators (fixed order) through LDS; lane holds channels 4*ch..+3 of row r16
  float* red = (float*)gs;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *(f32x4_t*)(red + ((wm * 2 + j) * 16 + r16) * RP + (wn * 2 + i) * 16 + 4 * ch) = acc[i][j];
  __syncthreads();

  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
#pragma unroll
  for (int k = 0; k < 1024 / NT; ++k) {
    const int q = t + k * NT;
    const int row = q >> 4, c4 = (q & 15) * 4;
    const int n = img0 + row;
    if (n >= a.B) continue;
    f32x4_t v = *(const f32x4_t*)((const float*)smem + row * RP + c4);
#pragma unroll
    for (int gg = 1; gg < KG; ++gg) v += *(const f32x4_t*)((const float*)(smem + gg * (2 * STAGE)) + row * RP + c4);
    const int co = n0 + c4;
    if (a.bias != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += co + e < a.n_real ? a.bias[co + e] : 0.f;
    }
    if (a.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    const int64_t pix = (int64_t)n * HW + pos;
    if (mask != nullptr) {
      const f32x4_t mv = load4f(mask + pix * a.mask_cs + a.mask_co + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
    }
    store4f(out + pix * a.out_cs + a.out_co + co, v);
  }
}

template <typename T>
static int launch_pconv_t(const PconvArgs& a, hipStream_t s, const char** name) {
  const int grid = (a.units + 7) / 8 * 8;
  // few units: split K four ways inside the workgroup; otherwise two ways and two workgroups per CU
  if (a.units <= 256) {
    if (name) *name = "pconv_64x64_k4";
    hipLaunchKernelGGL((pconv_kernel<T, 4>), dim3(grid), dim3(1024), 0, s, a);
  } else {
    if (name) *name = "pconv_64x64_k2";
    hipLaunchKernelGGL((pconv_kernel<T, 2>), dim3(grid), dim3(512), 0, s, a);
  }
  return check_hip(hipGetLastError(), "pconv launch");
}

}  // namespace mi
