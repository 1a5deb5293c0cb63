// 3x3 convolution (forward and dgrad, bf16 and fp32) for layers with FEW PIXELS: the deep levels of the U-Net of
// models/unet2d.py:28-31 of the reference at batch sizes of one to a few images, where a layer is
// a few hundred pixels against a megabyte or more of weights.
//
// A pixel-tiled implicit GEMM gives such a layer a handful of workgroups, each streaming its whole
// weight slice through one CU: the layer then runs at the latency of that one stream.  Here the
// work is cut the other way.  A workgroup owns 16 or 32 output channels and 16 or 32 pixels; its KW waves
// split the K range (tap, 32-channel chunk) between them, every wave loads its MFMA fragments of
// weights and of gathered pixels STRAIGHT from global memory into registers (no LDS staging: every
// byte is used by one wave only), several units deep so that a wave's whole share is in flight
// after one or two round trips, and the KW accumulators are summed through LDS in a fixed order.
//
//   out[p][n] = epi( sum_{tap} sum_c in[p + tap][c] * w[n][tap*Cp + c] )
//
// Pixels on the zero padding and units past a wave's share are fetched through a buffer
// descriptor with an out-of-range offset: such a lane reads zeros and touches no memory.
#include "igemm_impl.h"

namespace mi {

struct SconvArgs {
  const void* in; int in_cs, in_co;
  const void* w; int Kp, Cp;            // packed weights [N][Kp], Cp channels per tap
  int N; const float* bias; int n_real; int relu;
  void* out; int out_cs, out_co;
  const void* mask; int mask_cs, mask_co;
  void* pool; int pool_cs, pool_co;     // optional: also write the 2x2 max-pool of the output (W of 4, 8 or 16; 32-pixel tiles)
  int H, W, M;
  int chunks, units;                    // Cp / (channels per unit: 32 in bf16, 16 in fp32), 9*chunks
  int ups, upw;                         // units per K split (blockIdx.z), units per wave within it
  int ksplit;                           // workgroups sharing one output tile
  float* part; unsigned* count;         // ksplit > 1: partial tiles [ksplit][tiles][..] and one arrival counter per tile
  FastDiv d_hw, d_w, d_chunks;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr unsigned kOob = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sconv_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// A lane's 16 bytes are 8 (bf16) or 4 (fp32) consecutive k of one weight row / one pixel: the operand layout of one
// v_mfma_f32_16x16x32_bf16, or of four v_mfma_f32_16x16x4_f32 (element j of every lane in the j-th; both operands
// permute k the same way).  A unit of the K range is a tap's chunk of 32 (bf16) or 16 (fp32) channels.
template <typename T> __device__ __forceinline__ typename ElemTraits<T>::frag_t ld_frag(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return __builtin_bit_cast(typename ElemTraits<T>::frag_t, v);
}

template <typename T, int PF, int TN, int KW, int U>
__global__ __launch_bounds__(64 * KW) void sconv_kernel(const SconvArgs a) {
  typedef typename ElemTraits<T>::frag_t frag_t;
  constexpr unsigned EPC = ElemTraits<T>::EPC, UC = 4 * EPC, ESZ = sizeof(T);     // elements per lane, channels per unit
  __shared__ float red[KW - 1][TN * PF][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, ch = lane >> 4;
  const int n0 = blockIdx.x * (16 * TN), m0 = blockIdx.y * (16 * PF);
  const int HW = a.H * a.W;

  const unsigned wbase = ((unsigned)(n0 + r16) * (unsigned)a.Kp + (unsigned)ch * EPC) * ESZ;
  const __amdgpu_buffer_rsrc_t rin = sconv_rsrc(a.in), rw = sconv_rsrc(a.w);

  const int z1 = ((int)blockIdx.z + 1) * a.ups < a.units ? ((int)blockIdx.z + 1) * a.ups : a.units;
  const int u0 = (int)blockIdx.z * a.ups + wave * a.upw;
  const int u1 = u0 + a.upw < z1 ? u0 + a.upw : z1;
  int tap = (int)fdiv((unsigned)u0, a.d_chunks), c = u0 - tap * a.chunks;

  // The weights of the first batch are asked for before anything else: their addresses need no pixel geometry,
  // and they are the loads that come from far away (a layer's weights are read once, its few pixels by every workgroup).
  frag_t wf[U][TN];
  int wtap = tap, wc = c;
  auto load_weights = [&](int u) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
#pragma unroll
      for (int t = 0; t < TN; ++t)
        wf[i][t] = ld_frag<T>(rw, u + i < u1 ? wbase + (unsigned)(t * 16 * a.Kp + wtap * a.Cp + wc * (int)UC) * ESZ : kOob);
      if (++wc == a.chunks) { wc = 0; ++wtap; }
    }
  };
  load_weights(u0);
  __builtin_amdgcn_sched_barrier(0);

  // per pixel fragment: byte offset of this lane's pixel and the 9-bit set of in-image taps
  unsigned pbase[PF], vtaps[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int p = m0 + j * 16 + r16;
    const bool live = p < a.M;
    const unsigned pc = live ? (unsigned)p : 0u;
    const unsigned r = pc - fdiv(pc, a.d_hw) * (unsigned)HW;
    const int y = (int)fdiv(r, a.d_w), x = (int)r - y * a.W;
    unsigned vt = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (live && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) vt |= 1u << t;
    }
    vtaps[j] = vt;
    pbase[j] = (pc * (unsigned)a.in_cs + (unsigned)a.in_co + (unsigned)ch * EPC) * ESZ;
  }

  f32x4_t acc[TN * PF];
#pragma unroll
  for (int j = 0; j < TN * PF; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int u = u0; u < u1; u += U) {
    frag_t pf[U][PF];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool on = u + i < u1;
      const int ty = (tap * 11) >> 5, tx = tap - 3 * ty;              // tap / 3, tap % 3 for tap < 9
      const unsigned shift = (unsigned)(((ty - 1) * a.W + (tx - 1)) * a.in_cs + c * (int)UC) * ESZ;
#pragma unroll
      for (int j = 0; j < PF; ++j)
        pf[i][j] = ld_frag<T>(rin, (on && ((vtaps[j] >> tap) & 1u)) ? pbase[j] + shift : kOob);
      if (++c == a.chunks) { c = 0; ++tap; }
    }
    __builtin_amdgcn_sched_barrier(0);      // every load of the batch is issued before the first MFMA waits
#pragma unroll
    for (int i = 0; i < U; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < PF; ++j) acc[t * PF + j] = mma_slab(wf[i][t], pf[i][j], acc[t * PF + j]);
    if (u + U < u1) {
      __builtin_amdgcn_sched_barrier(0);
      load_weights(u + U);
    }
  }

  // K waves combine in wave order
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

  // K splits across workgroups: every workgroup leaves its partial tile in memory and takes a
  // ticket; the one that draws the last ticket sums all partials in split order (its own included,
  // so the result does not depend on which workgroup finishes last) and leaves the counter at zero
  // for the next launch.
  if (a.ksplit > 1) {
    const int tiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr int F = TN * PF * 4;
    // Partials and counters are written and read with device-scope accesses that go to the memory
    // side of the per-XCD L2s, and the ticket is drawn only after this wave's partial stores have
    // been acknowledged: a release fence here would write back the whole L2, which costs more than
    // the layer.
    float* mine = a.part + ((size_t)blockIdx.z * tiles + tile) * (F * 64) + lane;
#pragma unroll
    for (int j = 0; j < TN * PF; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __hip_atomic_store(mine + (j * 4 + q) * 64, acc[j][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(a.count + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (unsigned)a.ksplit - 1u) return;
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(a.count + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int j = 0; j < TN * PF; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float* all = a.part + (size_t)tile * (F * 64) + lane;
    for (int z = 0; z < a.ksplit; ++z) {
      const float* pz = all + (size_t)z * tiles * (F * 64);
#pragma unroll
      for (int j = 0; j < TN * PF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[j][q] += __hip_atomic_load(pz + (j * 4 + q) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // lane holds channels n..n+3 (n = n0 + 16*t + 4*(lane>>4)) of pixel (lane&15) of each fragment
  T* __restrict__ out = (T*)a.out;
  const T* __restrict__ mask = (const T*)a.mask;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n0 + 16 * t + 4 * ch;
    f32x4_t biasv{0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float b = a.bias[n + q < a.n_real ? n + q : 0];
        biasv[q] = n + q < a.n_real ? b : 0.f;
      }
    }
    f32x4_t v[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int p = m0 + j * 16 + r16;
      v[j] = acc[t * PF + j] + biasv;
      if (a.relu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = v[j][q] > 0.f ? v[j][q] : 0.f;
      }
      if (p >= a.M) continue;
      if (mask != nullptr) {
        const f32x4_t mv = load4f(mask + (int64_t)p * a.mask_cs + a.mask_co + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[j][q] = mv[q] > 0.f ? v[j][q] : 0.f;
      }
      store4f(out + (int64_t)p * a.out_cs + a.out_co + n, v[j]);
    }
    // 2x2 max-pool of the tile, which holds whole row pairs of images 4, 8 or 16 wide.  The window's other
    // column is the next lane; its other row the lane 4 or 8 further (four or two image rows per 16-pixel
    // fragment) or, 16 wide, the same lane of the tile's other fragment.  Rounding is monotonic, so the
    // maximum of the fp32 values rounds to the maximum of the stored ones.
    {
      if (a.pool != nullptr) {
        T* __restrict__ pool = (T*)a.pool;
        if (a.W == 16) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[0][q] = fmaxf(v[0][q], v[PF - 1][q]);
        } else {
          const int dist = a.W;     // 4 or 8 lanes
#pragma unroll
          for (int j = 0; j < PF; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[j][q] = fmaxf(v[j][q], __shfl_xor(v[j][q], dist));
        }
#pragma unroll
        for (int j = 0; j < (PF); ++j) {
          if (a.W == 16 && j == 1) break;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[j][q] = fmaxf(v[j][q], __shfl_xor(v[j][q], 1));
          const int p = m0 + j * 16 + r16;
          const unsigned b = fdiv((unsigned)p, a.d_hw), r = (unsigned)p - b * (unsigned)HW;
          const int y = (int)fdiv(r, a.d_w), x = (int)r - y * a.W;
          if (p < a.M && !(x & 1) && !(y & 1)) {
            const int64_t pp = ((int64_t)b * (a.H / 2) + y / 2) * (a.W / 2) + x / 2;
            store4f(pool + pp * a.pool_cs + a.pool_co + n, v[j]);
          }
        }
      }
    }
  }
}

// Image shapes whose 2x2 max-pool the few-pixel kernel writes itself: a 16-pixel fragment must hold whole row
// pairs (images 4 or 8 wide) or the two fragments of a 32-pixel tile one pair (16 wide).
bool sconv_pools(int H, int W, int M) { return (W == 4 || W == 8 || W == 16) && H % 2 == 0 && M >= 16; }

// True when the few-pixel kernel is the better choice for a bf16 3x3 layer of M pixels and N output
// channels: a pixel-tiled kernel would run fewer than 64 workgroups.
bool sconv_applies(const IgemmArgs& a, int dtype) {
  const int uc = dtype == MI_BF16 ? 32 : 16, esz = dtype == MI_BF16 ? 2 : 4;
  if (a.mode != G_3x3 || a.epi == E_SHUFFLE_BIAS) return false;
  if (a.pool != nullptr && !sconv_pools(a.Hi, a.Wi, a.M)) return false;
  if (a.Hi != a.Ho || a.Wi != a.Wo || a.Cin % uc != 0) return false;
  if ((int64_t)a.M * a.N >= sconv_limit(dtype, a.Hi * a.Wi)) return false;     // bf16: 64 tiles of 256 pixels x 32 channels
  // byte offsets are 32-bit inside the kernel
  if ((int64_t)a.M * a.in_cs * esz >= (1ll << 31) || (int64_t)a.N * a.Kp * esz >= (1ll << 31)) return false;
  return true;
}

int launch_sconv(const IgemmArgs& g, int dtype, hipStream_t s, const char** name) {
  SconvArgs a{};
  a.in = g.in; a.in_cs = g.in_cs; a.in_co = g.in_co;
  a.w = g.w; a.Kp = g.Kp; a.Cp = g.Cin;
  a.N = g.N; a.bias = g.epi == E_PLAIN ? nullptr : g.bias; a.n_real = g.n_real; a.relu = g.epi == E_BIAS_RELU;
  a.out = g.out; a.out_cs = g.out_cs; a.out_co = g.out_co;
  a.mask = g.mask; a.mask_cs = g.mask_cs; a.mask_co = g.mask_co;
  a.pool = g.pool; a.pool_cs = g.pool_cs; a.pool_co = g.pool_co;
  MI_REQUIRE(g.pool == nullptr || (g.mask == nullptr && sconv_pools(g.Hi, g.Wi, g.M) && g.pool_cs % 16 == 0 && g.pool_co % 16 == 0 && ((uintptr_t)g.pool & 15) == 0),
             "sconv: the pooled output cannot be fused for this shape");
  a.H = g.Hi; a.W = g.Wi; a.M = g.M;
  a.chunks = g.Cin / (dtype == MI_BF16 ? 32 : 16); a.units = 9 * a.chunks;
  a.d_hw = make_fastdiv((uint32_t)(g.Hi * g.Wi)); a.d_w = make_fastdiv((uint32_t)g.Wi);
  a.d_chunks = make_fastdiv((uint32_t)a.chunks);
  // workgroup tile: 16 pixels for a single tiny image; 32 pixels x 16 channels while that gives at
  // most 256 workgroups (their number matters most); 32 x 32, a third less operand traffic per
  // MFMA, beyond that
  const bool pool16 = g.pool != nullptr && g.Wi == 16;      // a row pair of a 16-wide image needs both fragments of a 32-pixel tile
  const int shape = (!pool16 && (int64_t)(g.N / 16) * ((g.M + 15) / 16) <= 256) ? 0
                    : ((int64_t)(g.N / 16) * ((g.M + 31) / 32) <= 256 || g.N % 32 != 0) ? 1 : 2;
  const int pix = shape == 0 ? 16 : 32, chn = shape == 2 ? 32 : 16;
  const dim3 tiles2(g.N / chn, (g.M + pix - 1) / pix);
  const int tiles = (int)(tiles2.x * tiles2.y);
  // With scratch memory from the caller, a long K range (8 KiB per weight row and up) is also split across
  // workgroups until about 256 of them run, each keeping at least 16 units: every CU then streams
  // a share of the weights.  Shorter ranges lose more to the exchange of partials than they gain.
  int ksplit = 1;
  if (g.kscratch != nullptr && tiles < 192 && a.units >= 128) {
    ksplit = (256 + tiles - 1) / tiles;
    if (ksplit > a.units / 16) ksplit = a.units / 16;
    const size_t per_split = (size_t)tiles * (pix / 16) * (chn / 16) * 4 * 64 * sizeof(float);
    const size_t counters = kKsplitCounterBytes;      // tiles < 192 counters of 4 bytes
    while (ksplit > 1 && counters + ksplit * per_split > g.kscratch_bytes) --ksplit;
    if (ksplit < 1) ksplit = 1;
    a.count = (unsigned*)g.kscratch;
    a.part = (float*)((char*)g.kscratch + counters);
  }
  a.ups = (a.units + ksplit - 1) / ksplit;
  ksplit = (a.units + a.ups - 1) / a.ups;
  a.ksplit = ksplit;
  // K waves: a wave's share fits one batch of loads in flight where the workgroup size allows
  const int kw = a.ups >= 128 ? 16 : a.ups >= 64 ? 8 : 4;
  a.upw = (a.ups + kw - 1) / kw;
  const dim3 grid(tiles2.x, tiles2.y, ksplit);
  static const char* const names[3][3] = {{"sconv_16x16_k4", "sconv_16x16_k8", "sconv_16x16_k16"},
                                          {"sconv_32x16_k4", "sconv_32x16_k8", "sconv_32x16_k16"},
                                          {"sconv_32x32_k4", "sconv_32x32_k8", "sconv_32x32_k16"}};
  const int ki = kw == 16 ? 2 : kw == 8 ? 1 : 0;
  if (name) *name = names[shape][ki];
#define SCONV_GO(PF, TN, KW, U)                                                                         \
  do {                                                                                                  \
    if (dtype == MI_BF16) hipLaunchKernelGGL((sconv_kernel<__bf16, PF, TN, KW, U>), grid, dim3(64 * KW), 0, s, a); \
    else hipLaunchKernelGGL((sconv_kernel<float, PF, TN, KW, U>), grid, dim3(64 * KW), 0, s, a);          \
  } while (0)
  if (shape == 0) {
    if (kw == 16) SCONV_GO(1, 1, 16, 9); else if (kw == 8) SCONV_GO(1, 1, 8, 9); else SCONV_GO(1, 1, 4, 9);
  } else if (shape == 1) {
    if (kw == 16) SCONV_GO(2, 1, 16, 8); else if (kw == 8) SCONV_GO(2, 1, 8, 9); else SCONV_GO(2, 1, 4, 9);
  } else {
    if (kw == 16) SCONV_GO(2, 2, 16, 5); else if (kw == 8) SCONV_GO(2, 2, 8, 6); else SCONV_GO(2, 2, 4, 6);
  }
#undef SCONV_GO
  return check_hip(hipGetLastError(), "sconv launch");
}

}  // namespace mi
