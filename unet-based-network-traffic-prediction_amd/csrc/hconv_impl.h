// Halo-tile 3x3 convolution kernel for gfx950: forward and dgrad of the reference's
// Conv2d(k=3, padding=1) layers (models/unet2d.py:28-31, :38-41, :51-54) when the image
// tiles evenly into 256-pixel blocks (row bands of one image, or several whole small images).
//
// Difference to the generic implicit-GEMM kernel (igemm_impl.h): the 9 taps do not re-gather
// the input from global memory.  A workgroup stages ONE zero-ringed halo tile per 64-byte
// channel chunk in LDS -- (R+2) x (W+2) pixel rows per group of R image rows -- and every tap
// reads its MFMA pixel fragments from that tile at a constant row offset
// (ty*(W+2) + tx).  Global->LDS traffic for activations drops from 9x to ~1.3x, and the
// per-stage MFMA work (9 taps x 16 tiles per wave) is long enough to cover the next stage's
// loads.  Workgroups are persistent over pixel tiles; when the layer has at most two channel
// chunks the packed weights of both stay resident in the two LDS stage buffers and are
// loaded once per workgroup.
#pragma once
#include "igemm_impl.h"

namespace mi {

struct HconvArgs {
  const void* in; int in_cs, in_co;
  const void* w; int Kp, Cp;            // packed weights [N][Kp], Cp channels per tap
  int N; const float* bias; int n_real; int relu;
  void* out; int out_cs, out_co;
  const void* mask; int mask_cs, mask_co;
  int W, R, G;                          // image width, rows per group, groups per tile (G*R*W = 256)
  int HR;                               // halo rows per tile = G*(R+2)*(W+2)
  int bands;                            // tiles per image (1 when a tile holds whole images)
  int ntiles, nchunks;
  const void* zero;                     // zero page
};

constexpr int kHconvMaxHalo = 400;      // halo rows the LDS buffer can hold (covers W <= 64)

template <typename T, int TN>
__global__ __launch_bounds__(256, 1) void hconv_kernel(const HconvArgs a) {
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr int EPC = ElemTraits<T>::EPC;
  constexpr int CK = 4 * EPC;                    // channels per 64-byte chunk
  constexpr int TM = 4, BM = 256, BN = TN * 16;
  constexpr int A_BYTES = kHconvMaxHalo * 64;
  constexpr int W_BYTES = 9 * BN * 64;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int A_IT = (kHconvMaxHalo * 4 + 255) / 256;
  constexpr int W_IT = (9 * BN * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ in = (const T*)a.in;
  const T* __restrict__ wp = (const T*)a.w;
  const int Wp2 = a.W + 2;
  const int grp_rows = (a.R + 2) * Wp2;

  // ---- loader state: this thread's halo rows (fixed for every tile)
  int rel[A_IT];          // pixel offset relative to the tile's first pixel
  int lds_a[A_IT];        // byte offset in the A buffer (multiple of 16) | kind in the low bits, or -1 (slot unused)
                          // kind: 0 interior, 1 always zero, 2 top halo row, 3 bottom halo row
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int id = t + 256 * i;
    const int hr = id >> 2, ch = id & 3;
    if (hr < a.HR) {
      const int g = hr / grp_rows, rem = hr - g * grp_rows;
      const int hy = rem / Wp2, hx = rem - hy * Wp2;
      rel[i] = g * a.R * a.W + (hy - 1) * a.W + (hx - 1);
      const int kind = (hx == 0 || hx == a.W + 1) ? 1 : (hy == 0 ? 2 : (hy == a.R + 1 ? 3 : 0));
      lds_a[i] = (hr * 64 + ((ch ^ ((hr >> 1) & 2)) << 4)) | kind;
    } else {
      rel[i] = 0; lds_a[i] = -1;
    }
  }
  const int chq = (t & 3) * EPC;   // element offset of this thread's 16-byte chunk within the 64-byte row
  // element offsets of the zero page relative to each operand's base pointer
  const int64_t zoff_in = ((intptr_t)a.zero - (intptr_t)a.in) / (int64_t)sizeof(T);
  const int64_t zoff_w = ((intptr_t)a.zero - (intptr_t)a.w) / (int64_t)sizeof(T);

  u32x4_t ra[A_IT], rw[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) rw[i] = u32x4_t{0u, 0u, 0u, 0u};
  auto load_a = [&](int tile, int chunk) {
    const int band = tile % a.bands;
    const bool top0 = band == 0, bot0 = band == a.bands - 1;
    const int64_t base = (int64_t)tile * BM;
    const int coff = a.in_co + chunk * CK + chq;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int kind = lds_a[i] & 3;   // (-1 & 3) == 3 is harmless: the slot is never stored
      const bool zero = lds_a[i] < 0 || kind == 1 || (kind == 2 && top0) || (kind == 3 && bot0);
      // padding lanes read the zero page: address select before the load, no use of the data
      // until the LDS store, so all staging loads stay in flight across the MFMA phase
      // (integer select of the OFFSET from one base pointer: a select between two pointers
      // makes the optimiser split the load into branches again)
      const int64_t off = zero ? zoff_in : (base + rel[i]) * a.in_cs + coff;
      ra[i] = *(const u32x4_t*)(in + opaque_offset(off));
    }
  };
  auto load_w = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int id = t + 256 * i, row = id >> 2;
      const int tap = row / BN, n = row - tap * BN;
      const int64_t off = row < 9 * BN ? (int64_t)(n0 + n) * a.Kp + tap * a.Cp + chunk * CK + chq : zoff_w;
      rw[i] = *(const u32x4_t*)(wp + opaque_offset(off));
    }
  };
  auto store_a = [&](int buf) {
    char* A = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      if (lds_a[i] >= 0) *(u32x4_t*)(A + (lds_a[i] & ~15)) = ra[i];
  };
  auto store_w = [&](int buf) {
    char* Wt = smem + buf * STAGE + A_BYTES;
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      const int id = t + 256 * i, row = id >> 2, ch = id & 3;
      if (row < 9 * BN) *(u32x4_t*)(Wt + row * 64 + ((ch ^ ((row >> 1) & 2)) << 4)) = rw[i];
    }
  };

  // ---- compute state: halo row of the top-left tap for this lane's pixel in each of its 4 tiles
  const int r16 = lane & 15, ch = lane >> 4;
  int hrow[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int pl = (wave * TM + j) * 16 + r16;         // pixel index inside the tile
    const int g = pl / (a.R * a.W), rem = pl - g * (a.R * a.W);
    const int y = rem / a.W, x = rem - y * a.W;
    hrow[j] = g * grp_rows + y * Wp2 + x;
  }
  const int wfrag_off = r16 * 64 + ((ch ^ ((r16 >> 1) & 2)) << 4);

  f32x4_t acc[TN][TM];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  auto compute = [&](int buf) {
    const char* A = smem + buf * STAGE;
    const char* Wt = A + A_BYTES;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int shift = (tap / 3) * Wp2 + (tap % 3);
      frag_t pf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int hr = hrow[j] + shift;
        pf[j] = *(const frag_t*)(A + hr * 64 + ((ch ^ ((hr >> 1) & 2)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *(const frag_t*)(Wt + (tap * BN + i * 16) * 64 + wfrag_off);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mma_slab(wf[i], pf[j], acc[i][j]);
    }
  };
  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
  // bias of this lane's 4 channels per tile, read once (unconditional clamped loads + select)
  f32x4_t biasv[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + i * 16 + 4 * ch + q;
      float b = 0.f;
      if (a.bias != nullptr) {
        b = a.bias[n < a.n_real ? n : 0];
        if (n >= a.n_real) b = 0.f;
      }
      biasv[i][q] = b;
    }
  }
  auto epilogue = [&](int tile) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int64_t p = (int64_t)tile * BM + (wave * TM + j) * 16 + r16;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + i * 16 + 4 * ch;
        f32x4_t v = acc[i][j];
        v += biasv[i];
        if (a.relu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
        }
        if (mask != nullptr) {
          const f32x4_t mv = load4f(mask + p * a.mask_cs + a.mask_co + n);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = mv[q] > 0.f ? v[q] : 0.f;
        }
        store4f(out + p * a.out_cs + a.out_co + n, v);
      }
    }
  };

  // ---- persistent loop over (tile, chunk) stages
  const bool w_resident = a.nchunks <= 2;   // stage parity == chunk parity: both chunks stay in LDS
  const int my_tiles = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstages = my_tiles * a.nchunks;
  if (nstages <= 0) return;
  // when weights are resident and the chunk count is odd (1), stage parity alternates while the
  // chunk does not: keep a copy in both buffers by loading during the first two stages
  int tile = blockIdx.x, chunk = 0;
  load_a(tile, 0);
  load_w(0);
  store_a(0);
  store_w(0);
  __syncthreads();
  zero_acc();
  for (int s = 0; s < nstages; ++s) {
    int ntile = tile, nchunk = chunk + 1;
    if (nchunk == a.nchunks) { nchunk = 0; ntile = tile + gridDim.x; }
    const bool more = s + 1 < nstages;
    const bool need_w = more && (!w_resident || s + 1 < 2);
    if (more) {
      load_a(ntile, nchunk);
      if (need_w) load_w(nchunk);
    }
    compute(s & 1);
    if (chunk == a.nchunks - 1) {
      epilogue(tile);
      zero_acc();
    }
    if (more) {
      store_a((s + 1) & 1);
      if (need_w) store_w((s + 1) & 1);
    }
    __syncthreads();
    tile = ntile;
    chunk = nchunk;
  }
}

// Tiling of an (H, W) image into 256-pixel tiles; returns false when the halo kernel does not apply.
inline bool hconv_tiling(int H, int W, int* R, int* G, int* bands) {
  if (W < 8 || W > 64) return false;
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
  return (*G) * (*R + 2) * (W + 2) <= kHconvMaxHalo;
}

template <typename T>
static int launch_hconv_t(const HconvArgs& a, hipStream_t s, const char** name) {
  const int bn = (a.N % 64 == 0) ? 64 : (a.N % 32 == 0) ? 32 : 16;
  const int ny = a.N / bn;
  int gx = a.ntiles;
  const int cap = (256 + ny - 1) / ny;          // one workgroup per CU
  if (gx > cap) gx = cap;
  dim3 grid(gx, ny);
  if (bn == 64) { if (name) *name = "hconv_256x64"; hipLaunchKernelGGL((hconv_kernel<T, 4>), grid, dim3(256), 0, s, a); }
  else if (bn == 32) { if (name) *name = "hconv_256x32"; hipLaunchKernelGGL((hconv_kernel<T, 2>), grid, dim3(256), 0, s, a); }
  else { if (name) *name = "hconv_256x16"; hipLaunchKernelGGL((hconv_kernel<T, 1>), grid, dim3(256), 0, s, a); }
  return check_hip(hipGetLastError(), "hconv launch");
}

}  // namespace mi
