// Bandwidth-bound kernels of the U-Net hot path: layout changes, weight packing, 2x2
// max-pool forward/backward, the 1x1 head, MSE, bilinear resize, Adam, and the
// fixed-order combination of weight-gradient partials.  All are streaming kernels:
// 16-byte vector accesses along the contiguous (channel) axis, grid-stride loops capped
// at a few workgroups per CU, wavefront (64-lane) shuffles for reductions.
#include "igemm_impl.h"

namespace mi {

static inline int grid_for(int64_t work_items, int block = 256, int max_blocks = 256 * 8) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

// ------------------------------------------------------------------------------------------
// Layout: fp32 NCHW <-> NHWC view
// One thread per (pixel, block of 32 channels): consecutive threads walk pixels, so every
// channel-plane read is coalesced, and a thread writes its pixel's 32 channels as whole 16-byte
// pieces of one contiguous row segment.  NG = 4-channel groups of the block that hold real
// channels; all their loads are issued unconditionally (channel index clamped) before any use.
// Index math is 32-bit (64-bit division is emulated); the launcher checks the range.
template <typename T, int NG>
__device__ __forceinline__ void nchw_to_nhwc_block(const float* __restrict__ sp, T* __restrict__ dp, int nreal, int nstore,
                                                   uint32_t HW) {
  float v[NG * 4];
#pragma unroll
  for (int q = 0; q < NG * 4; ++q) v[q] = sp[(int64_t)(q < nreal ? q : nreal - 1) * HW];
#pragma unroll
  for (int q = 0; q < NG * 4; ++q) v[q] = q < nreal ? v[q] : 0.f;
  // the padded width covers every group that holds a real channel: these stores are unconditional
#pragma unroll
  for (int q = 0; q < NG * 4; q += 4) store4f(dp + q, f32x4_t{v[q], v[q + 1], v[q + 2], v[q + 3]});
#pragma unroll
  for (int q = NG * 4; q < 32; q += 4)
    if (q < nstore) store4f(dp + q, f32x4_t{0.f, 0.f, 0.f, 0.f});
}
// `tail` zero elements behind the last pixel (a multiple of 4; the overlapping-pixel layout's readable slack)
template <typename T> __device__ __forceinline__ void nhwc_zero_tail(T* dst, int64_t npix, int cs, int tail) {
  if (tail > 0 && blockIdx.x == 0 && (int)threadIdx.x * 4 < tail) store4f(dst + npix * cs + threadIdx.x * 4, f32x4_t{0.f, 0.f, 0.f, 0.f});
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int cs, int co,
                                    uint32_t npix, int C, int Cp, uint32_t HW, int tail) {
  nhwc_zero_tail(dst, (int64_t)npix, cs, tail);
  const uint32_t blocks = (Cp + 31) / 32;
  const uint32_t total = npix * blocks;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t cb = i / npix, p = i - cb * npix;
    const int c0 = (int)cb * 32;
    const uint32_t b = p / HW, r = p - b * HW;
    const float* sp = src + ((int64_t)b * C + c0) * HW + r;
    T* dp = dst + (int64_t)p * cs + co + c0;
    const int nreal = min(32, C - c0), nstore = min(32, Cp - c0);   // nreal >= 1: pad_ch never adds a whole block
    switch ((nreal + 3) / 4) {
      case 1: nchw_to_nhwc_block<T, 1>(sp, dp, nreal, nstore, HW); break;
      case 2: nchw_to_nhwc_block<T, 2>(sp, dp, nreal, nstore, HW); break;
      case 3: nchw_to_nhwc_block<T, 3>(sp, dp, nreal, nstore, HW); break;
      case 4: nchw_to_nhwc_block<T, 4>(sp, dp, nreal, nstore, HW); break;
      case 5: nchw_to_nhwc_block<T, 5>(sp, dp, nreal, nstore, HW); break;
      case 6: nchw_to_nhwc_block<T, 6>(sp, dp, nreal, nstore, HW); break;
      case 7: nchw_to_nhwc_block<T, 7>(sp, dp, nreal, nstore, HW); break;
      default: nchw_to_nhwc_block<T, 8>(sp, dp, nreal, nstore, HW); break;
    }
  }
}
template <typename T, typename I>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int cs, int co, float* __restrict__ dst,
                                    I npix, int C, I HW) {
  const I total = npix * (I)C;
  for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I c = i / npix, p = i - c * npix;
    const I b = p / HW, r = p - b * HW;
    dst[((int64_t)b * C + c) * HW + r] = to_f(src[(int64_t)p * cs + co + c]);
  }
}
int launch_nchw_to_nhwc_vec(const float* src, void* dst, int cs, int co, int B, int C, int H, int W, int Cp, int tail, int dtype, hipStream_t s);
static int launch_nchw_to_nhwc_w(const float* src, void* dst, int cs, int co, int B, int C, int H, int W, int Cp, int tail, int dtype,
                                 hipStream_t s) {
  const int64_t npix = (int64_t)B * H * W;
  if ((H * W) % 4 == 0 && ((uintptr_t)src & 15) == 0 && npix * (Cp / 4) < (1ll << 31))
    return launch_nchw_to_nhwc_vec(src, dst, cs, co, B, C, H, W, Cp, tail, dtype, s);
  const int64_t total = npix * ((Cp + 31) / 32);
  MI_REQUIRE(total < (1ll << 31), "nchw_to_nhwc: tensor too large");
  const int grid = grid_for(total, 256, 256 * 16);
  if (dtype == MI_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid), dim3(256), 0, s, src, (__bf16*)dst, cs, co, (uint32_t)npix, C, Cp, (uint32_t)(H * W), tail);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, cs, co, (uint32_t)npix, C, Cp, (uint32_t)(H * W), tail);
  return check_hip(hipGetLastError(), "nchw_to_nhwc");
}
int launch_nchw_to_nhwc(const float* src, void* dst, int cs, int co, int B, int C, int H, int W, int dtype,
                        hipStream_t s) {
  return launch_nchw_to_nhwc_w(src, dst, cs, co, B, C, H, W, pad_ch(C), 0, dtype, s);
}
int launch_nchw_to_nhwc_narrow(const float* src, void* dst, int width, int tail, int B, int C, int H, int W, int dtype, hipStream_t s) {
  MI_REQUIRE(width % 16 == 0 && width >= C && tail >= 0 && tail % 4 == 0 && tail <= 1024, "nchw_to_nhwc: bad narrow layout");
  return launch_nchw_to_nhwc_w(src, dst, width, 0, B, C, H, W, width, tail, dtype, s);
}
int launch_nhwc_to_nchw(const void* src, int cs, int co, float* dst, int B, int C, int H, int W, int dtype,
                        hipStream_t s) {
  const int64_t npix = (int64_t)B * H * W;
  const int64_t total = npix * C;
  const int grid = grid_for(total, 256, 256 * 16);
  const bool small = total < (1ll << 31);
  if (dtype == MI_BF16) {
    if (small) hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, uint32_t>), dim3(grid), dim3(256), 0, s, (const __bf16*)src, cs, co, dst, (uint32_t)npix, C, (uint32_t)(H * W));
    else hipLaunchKernelGGL((nhwc_to_nchw_kernel<__bf16, int64_t>), dim3(grid), dim3(256), 0, s, (const __bf16*)src, cs, co, dst, npix, C, (int64_t)H * W);
  } else {
    if (small) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, uint32_t>), dim3(grid), dim3(256), 0, s, (const float*)src, cs, co, dst, (uint32_t)npix, C, (uint32_t)(H * W));
    else hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, int64_t>), dim3(grid), dim3(256), 0, s, (const float*)src, cs, co, dst, npix, C, (int64_t)H * W);
  }
  return check_hip(hipGetLastError(), "nhwc_to_nchw");
}

// ------------------------------------------------------------------------------------------
// Weight packing.  One thread per packed element (the packed image is written exactly once,
// zeros included, so the buffers never need a separate memset).
__device__ __forceinline__ int unpos(int pos, int Cin, int split, int split_pad) {
  // inverse of pos(ci): padded input-channel position -> logical channel or -1 (padding)
  if (split > 0) {
    if (pos < split_pad) return pos < split ? pos : -1;
    const int c = split + (pos - split_pad);
    return c < Cin ? c : -1;
  }
  return pos < Cin ? pos : -1;
}
// Element -> source index.  The packers first compute the source index of each of a thread's
// elements (-1 = padding), then issue all loads unconditionally with the index clamped to 0,
// and only then select and store: a load inside a branch is waited for inside that branch,
// which would serialise a thread's loads.
// (element indices are 32-bit: 64-bit integer division is emulated and would dominate the kernel)
struct PackDiv { FastDiv Kp, Cp, Kp2, Np; };
__device__ __forceinline__ uint32_t pdiv(uint32_t n, const FastDiv& f) { return f.mul ? fdiv(n, f) : n / f.d; }
__device__ __forceinline__ int64_t pack_convt2x2_src(int Cin, int Cout, int Np, int Cp, int Kp, int Kp2, uint32_t i,
                                                     uint32_t nf, const PackDiv& D) {
  if (i < nf) {
    const uint32_t row = pdiv(i, D.Kp), ci = i - row * (uint32_t)Kp;
    const uint32_t g = pdiv(row, D.Np), co = row - g * (uint32_t)Np;
    return ((int)co < Cout && (int)ci < Cin) ? ((int64_t)ci * Cout + co) * 4 + g : -1;
  }
  const uint32_t j = i - nf;
  const uint32_t ci = pdiv(j, D.Kp2), k = j - ci * (uint32_t)Kp2;
  const uint32_t g = pdiv(k, D.Np), co = k - g * (uint32_t)Np;
  return ((int)ci < Cin && g < 4u && (int)co < Cout) ? ((int64_t)ci * Cout + co) * 4 + g : -1;
}
template <typename T>
__device__ __forceinline__ void pack_store(T* __restrict__ wf, T* __restrict__ wb, uint32_t i, uint32_t nf, float v) {
  if (i < nf) wf[i] = from_f<T>(v);
  else wb[i - nf] = from_f<T>(v);
}
// All layers of a network in one launch: the job table travels in the kernel arguments, a
// workgroup finds its job from the running chunk counts.
// 3x3 layers are packed by wave-sized units chosen so that a unit's sources form a compact run
// of the (Cout,Cin,3,3) parameter: the 9 taps of a source line are consumed by the same wave
// instead of by nine far-apart workgroups (each line would otherwise be fetched 9 times from L2).
//   forward image  wf[n][tap*Cp + pos]:  unit = (row n, block of 128 positions), lanes along pos
//   backward image wb[pos][tap*Np + co]: unit = (4 positions, 32 output channels), lanes along co
// Transposed-conv layers are small and keep the flat element order (1024 elements per chunk).
template <typename T>
__device__ __forceinline__ void pack_conv3x3_unit(const PackJob& q, int unit, int lane) {
  T* wf = (T*)q.wf;
  T* wb = (T*)q.wb;
  const int split_pad = pad_ch(q.split);
  const int npb = (q.Cp + 127) / 128, ncb = (q.Np + 31) / 32, npq = (q.Cp + 3) / 4;
  const int units_wf = wf ? q.Np * npb : 0, units_wb = wb ? npq * ncb : 0;
  if (unit < units_wf) {
    const int n = unit / npb, pb = unit - n * npb;
    // 18 elements per lane in two batches of 9: all loads of a batch are in flight together
    // (offsets are 32-bit: a layer has fewer than 2^31 packed elements)
    for (int e0 = 0; e0 < 9 * 128; e0 += 576) {
      int src[9], dst[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int e = e0 + k * 64 + lane;
        const int tap = e >> 7, pos = pb * 128 + (e & 127);
        src[k] = -1; dst[k] = -1;
        if (pos < q.Cp) {
          dst[k] = n * q.Kp + tap * q.Cp + pos;
          const int ci = unpos(pos, q.Cin, q.split, split_pad);
          if (n < q.Cout && ci >= 0) src[k] = (n * q.Cin + ci) * 9 + tap;
        }
      }
      float v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = q.w[src[k] >= 0 ? src[k] : 0];
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (dst[k] >= 0) wf[dst[k]] = from_f<T>(src[k] >= 0 ? v[k] : 0.f);
    }
    if (pb == 0 && lane < q.Kp - 9 * q.Cp) wf[(int64_t)n * q.Kp + 9 * q.Cp + lane] = from_f<T>(0.f);
  } else if (unit < units_wf + units_wb) {
    const int u = unit - units_wf;
    const int pq = u / ncb, cb = u - pq * ncb;
    const int co = cb * 32 + (lane & 31);
    for (int r0 = 0; r0 < 36; r0 += 18) {         // r = 9*pl + tap, two r per 64-lane pass, 9 passes per batch
      int src[9], dst[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int r = r0 + 2 * k + (lane >> 5);
        const int pl = r / 9, tap = r - pl * 9, pos = pq * 4 + pl;
        src[k] = -1; dst[k] = -1;
        if (pos < q.Cp && co < q.Np) {
          dst[k] = pos * q.Kp2 + tap * q.Np + co;
          const int ci = unpos(pos, q.Cin, q.split, split_pad);
          if (co < q.Cout && ci >= 0) src[k] = (co * q.Cin + ci) * 9 + (8 - tap);
        }
      }
      float v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = q.w[src[k] >= 0 ? src[k] : 0];
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (dst[k] >= 0) wb[dst[k]] = from_f<T>(src[k] >= 0 ? v[k] : 0.f);
    }
    if (cb == 0 && lane < q.Kp2 - 9 * q.Np) {
      for (int pl = 0; pl < 4; ++pl)
        if (pq * 4 + pl < q.Cp) wb[(int64_t)(pq * 4 + pl) * q.Kp2 + 9 * q.Np + lane] = from_f<T>(0.f);
    }
  }
}
// The same units through LDS when the layer allows it (input channels a multiple of 4, no padding
// between the two halves of a concatenated input, 16-byte aligned parameter): a unit's source is then
// one run (forward image: up to 128 channels x 9 taps of one output channel) or 32 runs of 36 floats
// (backward image: 4 channels x 9 taps of 32 output channels); the wave reads it with 16-byte loads,
// parks it in its own LDS slice and writes every destination run as 16-byte stores of 8 elements --
// instead of 4-byte loads 36 bytes apart and 2-byte stores.
constexpr int kPackLds = 1152;   // floats per wave: 128 channels x 9 taps = 32 output channels x 36
constexpr int kPackVecCo = 32;   // output channels per backward-image unit (64 measured slower: 31 against 27 us)
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v);
template <> __device__ __forceinline__ void store8<float>(float* p, const float* v) {
  *(f32x4_t*)p = f32x4_t{v[0], v[1], v[2], v[3]};
  *(f32x4_t*)(p + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
}
template <> __device__ __forceinline__ void store8<__bf16>(__bf16* p, const float* v) {
  bf16x8_t t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
  *(bf16x8_t*)p = t;
}
template <typename T>
__device__ __forceinline__ void pack_conv3x3_unit_vec(const PackJob& q, int unit, int lane, float* sh) {
  T* wf = (T*)q.wf;
  T* wb = (T*)q.wb;
  const int npb = (q.Cp + 127) / 128, ncb = (q.Np + kPackVecCo - 1) / kPackVecCo, npq = (q.Cp + 3) / 4;
  const int units_wf = wf ? q.Np * npb : 0, units_wb = wb ? npq * ncb : 0;
  if (unit < units_wf) {
    const int n = unit / npb, pb = unit - n * npb;
    const int ci0 = pb * 128;
    const int cn = n < q.Cout ? max(0, min(128, q.Cin - ci0)) : 0;      // real channels of this block
    const int cb = min(128, q.Cp - ci0);                                 // stored channels of this block
    const f32x4_t* src = (const f32x4_t*)(q.w + ((int64_t)n * q.Cin + ci0) * 9);
    const int nvec = cn * 9 / 4;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = k * 64 + lane;
      if (i < nvec) *(f32x4_t*)(sh + 4 * i) = src[i];
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    T* row = wf + (int64_t)n * q.Kp + ci0;
    for (int item = lane; item < 9 * 16; item += 64) {
      const int tap = item >> 4, p0 = (item & 15) * 8;
      if (p0 >= cb) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = p0 + e < cn ? sh[(p0 + e) * 9 + tap] : 0.f;
      store8<T>(row + tap * q.Cp + p0, v);
    }
    if (pb == 0 && lane < q.Kp - 9 * q.Cp) wf[(int64_t)n * q.Kp + 9 * q.Cp + lane] = from_f<T>(0.f);
  } else if (unit < units_wf + units_wb) {
    const int u = unit - units_wf;
    const int pq = u / ncb, cbk = u - pq * ncb;
    const int ci0 = pq * 4, co0 = cbk * kPackVecCo;
    const bool real_ci = ci0 < q.Cin;                                    // 4 channels: all real or all padding
    const int ncr = real_ci ? max(0, min(kPackVecCo, q.Cout - co0)) : 0; // real output channels of this block
    const int ncs = min(kPackVecCo, q.Np - co0);                         // stored output channels
    // 9 vectors per output channel
    for (int i = lane; i < ncr * 9; i += 64) {
      const int c = i / 9, k = i - c * 9;
      *(f32x4_t*)(sh + c * 36 + 4 * k) = *(const f32x4_t*)(q.w + ((int64_t)(co0 + c) * q.Cin + ci0) * 9 + 4 * k);
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    for (int item = lane; item < 36 * (kPackVecCo / 8); item += 64) {
      const int r = item / (kPackVecCo / 8), c8 = (item % (kPackVecCo / 8)) * 8;   // r = 9*pl + tap
      const int pl = r / 9, tap = r - pl * 9;
      if (c8 >= ncs || ci0 + pl >= q.Cp) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c8 + e < ncr ? sh[(c8 + e) * 36 + pl * 9 + (8 - tap)] : 0.f;
      store8<T>(wb + (int64_t)(ci0 + pl) * q.Kp2 + tap * q.Np + co0 + c8, v);
    }
    if (cbk == 0 && lane < q.Kp2 - 9 * q.Np) {
      for (int pl = 0; pl < 4; ++pl)
        if (ci0 + pl < q.Cp) wb[(int64_t)(ci0 + pl) * q.Kp2 + 9 * q.Np + lane] = from_f<T>(0.f);
    }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJobs J) {
  __shared__ __attribute__((aligned(16))) float sh[4][kPackLds];
  const int chunk = blockIdx.x;
  int j = 0;
  while (j < J.n - 1 && chunk >= J.job[j].chunk_end) ++j;
  const PackJob& q = J.job[j];
  const int c0 = j > 0 ? J.job[j - 1].chunk_end : 0;
  if (!q.convt) {
    const int wave = threadIdx.x >> 6;
    if (q.vec) pack_conv3x3_unit_vec<T>(q, (chunk - c0) * 4 + wave, threadIdx.x & 63, sh[wave]);
    else pack_conv3x3_unit<T>(q, (chunk - c0) * 4 + wave, threadIdx.x & 63);
    return;
  }
  const uint32_t base = (uint32_t)(chunk - c0) * 1024u + threadIdx.x;
  T* wf = (T*)q.wf;
  T* wb = (T*)q.wb;
  const uint32_t nf = wf ? 4u * q.Np * q.Kp : 0u;
  const uint32_t nb = wb ? (uint32_t)q.Cp * q.Kp2 : 0u;
  const PackDiv D{{q.mKp, 31u + q.sKp, (uint32_t)q.Kp}, {q.mCp, 31u + q.sCp, (uint32_t)q.Cp},
                  {q.mKp2, 31u + q.sKp2, (uint32_t)q.Kp2}, {q.mNp, 31u + q.sNp, (uint32_t)q.Np}};
  int64_t src[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t i = base + e * 256;
    src[e] = -1;
    if (i < nf + nb) src[e] = pack_convt2x2_src(q.Cin, q.Cout, q.Np, q.Cp, q.Kp, q.Kp2, i, nf, D);
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = q.w[src[e] >= 0 ? src[e] : 0];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t i = base + e * 256;
    if (i < nf + nb) pack_store<T>(wf, wb, i, nf, src[e] >= 0 ? v[e] : 0.f);
  }
}
int pack_jobs_add(PackJobs& J, const float* w, void* wfwd, void* wbwd, int convt, int Cout, int Cin, int split) {
  if (J.n >= kMaxPackJobs) return 0;
  PackJob& q = J.job[J.n];
  q.w = w; q.wf = wfwd; q.wb = wbwd; q.convt = convt; q.Cout = Cout; q.Cin = Cin; q.split = split;
  q.vec = 0;
  int64_t total;
  if (convt) {
    q.Np = pad_ch(Cout); q.Cp = pad_ch(Cin); q.Kp = (int)kpad(q.Cp); q.Kp2 = (int)kpad(4 * q.Np);
    total = (wfwd ? (int64_t)4 * q.Np * q.Kp : 0) + (wbwd ? (int64_t)q.Cp * q.Kp2 : 0);
  } else {
    q.Np = pad_ch(Cout); q.Cp = concat_cp(Cin, split); q.Kp = (int)kpad(9 * q.Cp); q.Kp2 = (int)kpad(9 * q.Np);
    total = (wfwd ? (int64_t)q.Np * q.Kp : 0) + (wbwd ? (int64_t)q.Cp * q.Kp2 : 0);
  }
  if (total >= (1ll << 31)) return 0;
  const FastDiv dKp = make_fastdiv(q.Kp), dCp = make_fastdiv(q.Cp), dKp2 = make_fastdiv(q.Kp2), dNp = make_fastdiv(q.Np);
  q.mKp = dKp.mul; q.sKp = (uint8_t)(dKp.shift - 31); q.mCp = dCp.mul; q.sCp = (uint8_t)(dCp.shift - 31);
  q.mKp2 = dKp2.mul; q.sKp2 = (uint8_t)(dKp2.shift - 31); q.mNp = dNp.mul; q.sNp = (uint8_t)(dNp.shift - 31);
  const int prev = J.n > 0 ? J.job[J.n - 1].chunk_end : 0;
  if (convt) {
    q.chunk_end = prev + (int)((total + 1023) / 1024);
  } else {   // wave-sized units, 4 per workgroup (see pack_conv3x3_unit and pack_conv3x3_unit_vec)
    q.vec = Cin % 4 == 0 && (split == 0 || pad_ch(split) == split) && ((uintptr_t)w & 15) == 0;
    const int co_unit = q.vec ? kPackVecCo : 32;
    const int64_t units = (wfwd ? (int64_t)q.Np * ((q.Cp + 127) / 128) : 0) +
                          (wbwd ? (int64_t)((q.Cp + 3) / 4) * ((q.Np + co_unit - 1) / co_unit) : 0);
    q.chunk_end = prev + (int)((units + 3) / 4);
  }
  ++J.n;
  return 1;
}
int launch_pack_multi(const PackJobs& J, int dtype, hipStream_t s) {
  if (J.n == 0 || J.job[J.n - 1].chunk_end == 0) return MI_OK;
  const int grid = J.job[J.n - 1].chunk_end;
  if (dtype == MI_BF16)
    hipLaunchKernelGGL(pack_multi_kernel<__bf16>, dim3(grid), dim3(256), 0, s, J);
  else
    hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(grid), dim3(256), 0, s, J);
  return check_hip(hipGetLastError(), "pack_multi");
}
int launch_pack_conv3x3(const float* w, void* wfwd, void* wbwd, int Cout, int Cin, int split, int dtype,
                        hipStream_t s) {
  if (!wfwd && !wbwd) return MI_OK;
  PackJobs J;
  J.n = 0;
  MI_REQUIRE(pack_jobs_add(J, w, wfwd, wbwd, 0, Cout, Cin, split), "pack_conv3x3: layer too large");
  return launch_pack_multi(J, dtype, s);
}
int launch_pack_convt2x2(const float* w, void* wfwd, void* wbwd, int Cin, int Cout, int dtype, hipStream_t s) {
  if (!wfwd && !wbwd) return MI_OK;
  PackJobs J;
  J.n = 0;
  MI_REQUIRE(pack_jobs_add(J, w, wfwd, wbwd, 1, Cout, Cin, 0), "pack_convt2x2: layer too large");
  return launch_pack_multi(J, dtype, s);
}

// ------------------------------------------------------------------------------------------
// 2x2 max-pool.  One thread per (window, 16-byte channel group): 8 bf16 or 4 fp32 channels.
// The index type I is 32-bit whenever the window count allows (64-bit division is emulated).
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* v) {
    const f32x4_t t = *(const f32x4_t*)p;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = t[q];
  }
  static __device__ __forceinline__ void store(float* p, const float* v) { *(f32x4_t*)p = f32x4_t{v[0], v[1], v[2], v[3]}; }
};
template <> struct Vec16<__bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __bf16* p, float* v) {
    const bf16x8_t t = *(const bf16x8_t*)p;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (float)t[q];
  }
  static __device__ __forceinline__ void store(__bf16* p, const float* v) {
    bf16x8_t t;
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = (__bf16)v[q];
    *(bf16x8_t*)p = t;
  }
};
// The same conversion for plane sizes that are multiples of 4 (every benchmark shape): one thread per
// (4 consecutive pixels, 16-byte channel group).  It reads its N channels as N float4s -- 16-byte loads,
// lanes of a group walking consecutive pixel quads of a channel plane -- and writes each pixel's N channels
// as one 16-byte store, the lanes of one pixel covering a contiguous run of its row.  Index math is 32-bit.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_vec_kernel(const float* __restrict__ src, T* __restrict__ dst, int cs, int co,
                                                               uint32_t nquads, int C, int Cp, uint32_t HW, FastDiv dG, FastDiv dHW4, int tail) {
  constexpr int N = Vec16<T>::N;
  nhwc_zero_tail(dst, (int64_t)nquads * 4, cs, tail);
  const uint32_t G = (uint32_t)Cp / N;
  const uint32_t total = nquads * G;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t pq = fdiv(i, dG), cg = i - pq * G;
    const uint32_t b = fdiv(pq, dHW4), r = (pq - b * (HW / 4)) * 4;       // image, first pixel of the quad inside its plane
    const int c0 = (int)cg * N;
    f32x4_t v[N];
    if (c0 < C) {
      const float* sp = src + ((int64_t)b * C + c0) * HW + r;
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = *(const f32x4_t*)(sp + (int64_t)(c0 + k < C ? k : 0) * HW);
#pragma unroll
      for (int k = 0; k < N; ++k)
        if (c0 + k >= C) v[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    T* dp = dst + ((int64_t)b * HW + r) * cs + co + c0;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float row[N];
#pragma unroll
      for (int k = 0; k < N; ++k) row[k] = v[k][px];
      Vec16<T>::store(dp + (int64_t)px * cs, row);
    }
  }
}
int launch_nchw_to_nhwc_vec(const float* src, void* dst, int cs, int co, int B, int C, int H, int W, int Cp, int tail, int dtype, hipStream_t s) {
  const uint32_t HW = (uint32_t)(H * W), nquads = (uint32_t)((int64_t)B * HW / 4);
  const int G = Cp / (dtype == MI_BF16 ? 8 : 4);
  const int grid = grid_for((int64_t)nquads * G, 256, 256 * 32);
  if (dtype == MI_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_vec_kernel<__bf16>, dim3(grid), dim3(256), 0, s, src, (__bf16*)dst, cs, co, nquads, C, Cp, HW, make_fastdiv(G), make_fastdiv(HW / 4), tail);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_vec_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, cs, co, nquads, C, Cp, HW, make_fastdiv(G), make_fastdiv(HW / 4), tail);
  return check_hip(hipGetLastError(), "nchw_to_nhwc");
}
template <typename T, typename I>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int xcs, int xco, T* __restrict__ y, int ycs,
                                    int yco, int B, int H, int W, int Cp) {
  constexpr int N = Vec16<T>::N;
  const I Ho = H / 2, Wo = W / 2, G = Cp / N;
  const I total = (I)B * Ho * Wo * G;
  for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I op = i / G, g = i - op * G;
    const I t1 = op / Wo, ox = op - t1 * Wo;
    const I b = t1 / Ho, oy = t1 - b * Ho;
    const T* px = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * (int64_t)xcs + xco + g * N;
    float v00[N], v01[N], v10[N], v11[N], m[N];
    Vec16<T>::load(px, v00);
    Vec16<T>::load(px + xcs, v01);
    Vec16<T>::load(px + (int64_t)W * xcs, v10);
    Vec16<T>::load(px + (int64_t)(W + 1) * xcs, v11);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      // PyTorch's update rule: take val if (val > max) or isnan(val), scanning row-major.
      float mx = v00[q];
      if (v01[q] > mx || v01[q] != v01[q]) mx = v01[q];
      if (v10[q] > mx || v10[q] != v10[q]) mx = v10[q];
      if (v11[q] > mx || v11[q] != v11[q]) mx = v11[q];
      m[q] = mx;
    }
    Vec16<T>::store(y + (int64_t)op * ycs + yco + g * N, m);
  }
}
// Backward: the gradient of a pooled value goes to the FIRST maximum of its window (row-major
// scan, PyTorch's rule); optionally adds the skip gradient already stored for the unpooled
// tensor and applies the ReLU mask of x.  EVEN: H and W even, so every window is complete and
// all loads are unconditional; otherwise the partial windows of the last row/column are
// handled too (they only pass the skip gradient through).
template <typename T, typename I, bool EVEN, bool SKIP>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, int xcs, int xco, const T* __restrict__ dyp,
                                    int pcs, int pco, const T* __restrict__ dskip, int scs, int sco,
                                    T* __restrict__ dx, int dcs, int dco, int B, int H, int W, int Cp,
                                    int relu_mask, FastDiv dG, FastDiv dWc, FastDiv dHc) {
  constexpr int N = Vec16<T>::N;
  const I Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2, G = Cp / N;
  const I total = (I)B * Hc * Wc * G;
  for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    I wdx, t1, b;
    if (sizeof(I) == 4) { wdx = (I)fdiv((uint32_t)i, dG); t1 = (I)fdiv((uint32_t)wdx, dWc); b = (I)fdiv((uint32_t)t1, dHc); }
    else { wdx = i / G; t1 = wdx / Wc; b = t1 / Hc; }
    const I g = i - wdx * G, ox = wdx - t1 * Wc, oy = t1 - b * Hc;
    const bool pooled = EVEN || (oy < Ho && ox < Wo);  // window produced a pooled output
    float xv[4][N], gv[4][N];
    bool inb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const I yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
      inb[k] = EVEN || (yy < (I)H && xx < (I)W);
      const int64_t pix = ((int64_t)b * H + yy) * W + xx;
      if (inb[k]) Vec16<T>::load(x + pix * xcs + xco + g * N, xv[k]);
      else {
#pragma unroll
        for (int q = 0; q < N; ++q) xv[k][q] = 0.f;
      }
      if (inb[k] && SKIP) Vec16<T>::load(dskip + pix * scs + sco + g * N, gv[k]);
      else {
#pragma unroll
        for (int q = 0; q < N; ++q) gv[k][q] = 0.f;
      }
    }
    if (pooled) {
      float gp[N];
      Vec16<T>::load(dyp + (((int64_t)b * Ho + oy) * Wo + ox) * pcs + pco + g * N, gp);
#pragma unroll
      for (int q = 0; q < N; ++q) {
        int arg = 0;
        float mx = xv[0][q];
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (xv[k][q] > mx || xv[k][q] != xv[k][q]) { mx = xv[k][q]; arg = k; }
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k][q] += (k == arg) ? gp[q] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!inb[k]) continue;
      if (relu_mask) {
#pragma unroll
        for (int q = 0; q < N; ++q) gv[k][q] = xv[k][q] > 0.f ? gv[k][q] : 0.f;
      }
      const I yy = 2 * oy + (k >> 1), xx = 2 * ox + (k & 1);
      Vec16<T>::store(dx + (((int64_t)b * H + yy) * W + xx) * dcs + dco + g * N, gv[k]);
    }
  }
}
template <typename T>
static void maxpool2_fwd_dispatch(int64_t total, hipStream_t s, const void* x, int xcs, int xco, void* y, int ycs, int yco,
                                  int B, int H, int W, int Cp) {
  const int grid = grid_for(total, 256, 256 * 16);
  if (total < (1ll << 31))
    hipLaunchKernelGGL((maxpool2_fwd_kernel<T, uint32_t>), dim3(grid), dim3(256), 0, s, (const T*)x, xcs, xco, (T*)y, ycs, yco, B, H, W, Cp);
  else
    hipLaunchKernelGGL((maxpool2_fwd_kernel<T, int64_t>), dim3(grid), dim3(256), 0, s, (const T*)x, xcs, xco, (T*)y, ycs, yco, B, H, W, Cp);
}
int launch_maxpool2_fwd(const void* x, int xcs, int xco, void* y, int ycs, int yco, int B, int H, int W, int C,
                        int dtype, hipStream_t s) {
  const int Cp = pad_ch(C);
  const int64_t total = (int64_t)B * (H / 2) * (W / 2) * (Cp / (dtype == MI_BF16 ? 8 : 4));
  if (total == 0) return MI_OK;
  if (dtype == MI_BF16) maxpool2_fwd_dispatch<__bf16>(total, s, x, xcs, xco, y, ycs, yco, B, H, W, Cp);
  else maxpool2_fwd_dispatch<float>(total, s, x, xcs, xco, y, ycs, yco, B, H, W, Cp);
  return check_hip(hipGetLastError(), "maxpool2_fwd");
}
template <typename T, typename I>
static void maxpool2_bwd_dispatch2(bool even, int grid, hipStream_t s, const void* x, int xcs, int xco, const void* dyp,
                                   int pcs, int pco, const void* dskip, int scs, int sco, void* dx, int dcs, int dco,
                                   int B, int H, int W, int Cp, int relu_mask) {
#define POOL_BWD(EV, SK)                                                                                               \
  hipLaunchKernelGGL((maxpool2_bwd_kernel<T, I, EV, SK>), dim3(grid), dim3(256), 0, s, (const T*)x, xcs, xco, (const T*)dyp, \
                     pcs, pco, (const T*)dskip, scs, sco, (T*)dx, dcs, dco, B, H, W, Cp, relu_mask, \
                     make_fastdiv(Cp / Vec16<T>::N), make_fastdiv((W + 1) / 2), make_fastdiv((H + 1) / 2))
  if (even && dskip) POOL_BWD(true, true);
  else if (even) POOL_BWD(true, false);
  else if (dskip) POOL_BWD(false, true);
  else POOL_BWD(false, false);
#undef POOL_BWD
}
int launch_maxpool2_bwd(const void* x, int xcs, int xco, const void* dyp, int pcs, int pco, const void* dskip,
                        int scs, int sco, void* dx, int dcs, int dco, int B, int H, int W, int C, int relu_mask,
                        int dtype, hipStream_t s) {
  const int Cp = pad_ch(C);
  const int64_t total = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (Cp / (dtype == MI_BF16 ? 8 : 4));
  const int grid = grid_for(total, 256, 256 * 16);
  const bool even = H % 2 == 0 && W % 2 == 0, small = total < (1ll << 31);
  if (dtype == MI_BF16) {
    if (small) maxpool2_bwd_dispatch2<__bf16, uint32_t>(even, grid, s, x, xcs, xco, dyp, pcs, pco, dskip, scs, sco, dx, dcs, dco, B, H, W, Cp, relu_mask);
    else maxpool2_bwd_dispatch2<__bf16, int64_t>(even, grid, s, x, xcs, xco, dyp, pcs, pco, dskip, scs, sco, dx, dcs, dco, B, H, W, Cp, relu_mask);
  } else {
    if (small) maxpool2_bwd_dispatch2<float, uint32_t>(even, grid, s, x, xcs, xco, dyp, pcs, pco, dskip, scs, sco, dx, dcs, dco, B, H, W, Cp, relu_mask);
    else maxpool2_bwd_dispatch2<float, int64_t>(even, grid, s, x, xcs, xco, dyp, pcs, pco, dskip, scs, sco, dx, dcs, dco, B, H, W, Cp, relu_mask);
  }
  return check_hip(hipGetLastError(), "maxpool2_bwd");
}

// ------------------------------------------------------------------------------------------
// 1x1 head.  Cout is tiny (1 or 12): a per-pixel dot product, bandwidth-bound.
constexpr int kHeadMaxCin = 64, kHeadMaxCout = 16;
template <typename T, int CP>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, int xcs, int xco,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ pred, int64_t npix, int HW, int Cin,
                                                        int Cout) {
  __shared__ float sw[kHeadMaxCout * CP + kHeadMaxCout];
  for (int i = threadIdx.x; i < Cout * CP; i += blockDim.x) {
    const int co = i / CP, c = i % CP;
    sw[i] = c < Cin ? w[co * Cin + c] : 0.f;
  }
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * CP + i] = bias[i];
  __syncthreads();
  constexpr int N = Vec16<T>::N;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const T* px = x + p * xcs + xco;
    float v[CP];                       // the whole pixel row first: all loads in flight together
#pragma unroll
    for (int c = 0; c < CP; c += N) Vec16<T>::load(px + c, v + c);
    // pixel counts fit 32 bits (checked by the launcher); 64-bit division is emulated
    const uint32_t b = (uint32_t)p / (uint32_t)HW, r = (uint32_t)p - b * (uint32_t)HW;
    for (int co = 0; co < Cout; ++co) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) acc = fmaf(v[c], sw[co * CP + c], acc);
      pred[((int64_t)b * Cout + co) * HW + r] = acc + sw[Cout * CP + co];
    }
  }
}
// 32-channel inputs (every U-Net here): CP/N lanes share a pixel, each loading ONE 16-byte piece of its row, so a
// wave's load instruction covers a contiguous kilobyte instead of 64 rows at a 64-byte stride; the partial dot
// products meet through lane shuffles inside the pixel's lane group.
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_quad_kernel(const T* __restrict__ x, int xcs, int xco,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ pred, uint32_t npix, FastDiv dHW, int Cin,
                                                             int Cout) {
  constexpr int CP = 32, N = Vec16<T>::N, LPP = CP / N;
  __shared__ float sw[kHeadMaxCout * CP + kHeadMaxCout];
  for (int i = threadIdx.x; i < Cout * CP; i += blockDim.x) {
    const int co = i / CP, c = i % CP;
    sw[i] = c < Cin ? w[co * Cin + c] : 0.f;
  }
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * CP + i] = bias[i];
  __syncthreads();
  const int cg = threadIdx.x % LPP;
  const uint32_t HW = dHW.d, stride = gridDim.x * (256 / LPP);
  for (uint32_t p0 = blockIdx.x * (256 / LPP) + threadIdx.x / LPP; p0 < npix; p0 += 2 * stride) {
    // two pixels per trip: both rows in flight together (the second clamped to the last pixel, its result unused)
    const uint32_t p1 = p0 + stride, q1 = p1 < npix ? p1 : npix - 1;
    float v0[N], v1[N];
    Vec16<T>::load(x + (int64_t)p0 * xcs + xco + cg * N, v0);
    Vec16<T>::load(x + (int64_t)q1 * xcs + xco + cg * N, v1);
    const uint32_t b0 = fdiv(p0, dHW), r0 = p0 - b0 * HW, b1 = fdiv(q1, dHW), r1 = q1 - b1 * HW;
    for (int co = 0; co < Cout; ++co) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int q = 0; q < N; ++q) { const float wv = sw[co * CP + cg * N + q]; a0 = fmaf(v0[q], wv, a0); a1 = fmaf(v1[q], wv, a1); }
#pragma unroll
      for (int o = 1; o < LPP; o <<= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); }
      if (cg == 0) {
        pred[((int64_t)b0 * Cout + co) * HW + r0] = a0 + sw[Cout * CP + co];
        if (p1 < npix) pred[((int64_t)b1 * Cout + co) * HW + r1] = a1 + sw[Cout * CP + co];
      }
    }
  }
}
int launch_head_fwd(const void* x, int xcs, int xco, const float* w, const float* bias, float* pred, int B, int H,
                    int W, int Cin, int Cout, int dtype, hipStream_t s) {
  MI_REQUIRE(pad_ch(Cin) <= kHeadMaxCin && Cout <= kHeadMaxCout, "head: Cin <= 64 and Cout <= 16 supported");
  const int64_t npix = (int64_t)B * H * W;
  MI_REQUIRE(npix < (1ll << 31), "head: pixel count out of range");
  const int grid = grid_for(npix, 256, 256 * 16);
  const int Cp = pad_ch(Cin);
  if (Cp == 32) {
    const int lpp = dtype == MI_BF16 ? 4 : 8;
    const int g2 = grid_for(npix * lpp / 2, 256, 256 * 16);
    if (dtype == MI_BF16)
      hipLaunchKernelGGL(head_fwd_quad_kernel<__bf16>, dim3(g2), dim3(256), 0, s, (const __bf16*)x, xcs, xco, w, bias, pred, (uint32_t)npix, make_fastdiv(H * W), Cin, Cout);
    else
      hipLaunchKernelGGL(head_fwd_quad_kernel<float>, dim3(g2), dim3(256), 0, s, (const float*)x, xcs, xco, w, bias, pred, (uint32_t)npix, make_fastdiv(H * W), Cin, Cout);
    return check_hip(hipGetLastError(), "head_fwd");
  }
#define HEAD_FWD(TT, CPV) \
  hipLaunchKernelGGL((head_fwd_kernel<TT, CPV>), dim3(grid), dim3(256), 0, s, (const TT*)x, xcs, xco, w, bias, pred, npix, H * W, Cin, Cout)
  if (dtype == MI_BF16) {
    if (Cp == 32) HEAD_FWD(__bf16, 32); else if (Cp == 48) HEAD_FWD(__bf16, 48); else HEAD_FWD(__bf16, 64);
  } else {
    if (Cp == 32) HEAD_FWD(float, 32); else if (Cp == 48) HEAD_FWD(float, 48); else HEAD_FWD(float, 64);
  }
#undef HEAD_FWD
  return check_hip(hipGetLastError(), "head_fwd");
}

// Head backward.  Phase 1 (thread = pixel): dx = (W^T dpred) * (x > 0); the pixel's dpred and
// x values are parked in LDS.  Phase 2 (thread = (co, c) pair): accumulate dW over the 256
// pixels of the tile.  Each workgroup walks several tiles and writes one partial.
constexpr int kHeadTile = 256;
constexpr int kHeadAcc = (kHeadMaxCout * kHeadMaxCin + kHeadMaxCout + 255) / 256;  // (co,c) pairs per thread
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, int xcs, int xco,
                                                        const float* __restrict__ w, const float* __restrict__ dpred,
                                                        T* __restrict__ dx, int dcs, int dco, float* __restrict__ part,
                                                        int64_t npix, int HW, int Cin, int Cp, int Cout, int relu_mask) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* sw = sh;                              // [Cout][Cp]
  float* sx = sw + Cout * Cp;                  // [kHeadTile][Cp + 1]
  float* sg = sx + kHeadTile * (Cp + 1);       // [kHeadTile][Cout]
  for (int i = threadIdx.x; i < Cout * Cp; i += blockDim.x) {
    const int co = i / Cp, c = i % Cp;
    sw[i] = c < Cin ? w[co * Cin + c] : 0.f;
  }
  const int npairs = Cout * Cp + Cout;         // dW pairs then db entries
  float accp[kHeadAcc];
#pragma unroll
  for (int i = 0; i < kHeadAcc; ++i) accp[i] = 0.f;
  __syncthreads();
  const int64_t ntiles = (npix + kHeadTile - 1) / kHeadTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p = tile * kHeadTile + threadIdx.x;
    const bool ok = p < npix;
    float g[kHeadMaxCout];
    const uint32_t b = ok ? (uint32_t)p / (uint32_t)HW : 0u, r = ok ? (uint32_t)p - b * (uint32_t)HW : 0u;
#pragma unroll
    for (int co = 0; co < kHeadMaxCout; ++co) {
      g[co] = (ok && co < Cout) ? dpred[((int64_t)b * Cout + co) * HW + r] : 0.f;
      if (co < Cout) sg[threadIdx.x * Cout + co] = g[co];
    }
    for (int c = 0; c < Cp; c += 4) {
      f32x4_t xv = {0.f, 0.f, 0.f, 0.f};
      if (ok) xv = load4f(x + p * xcs + xco + c);
      f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int co = 0; co < kHeadMaxCout; ++co)
        if (co < Cout) {
#pragma unroll
          for (int q = 0; q < 4; ++q) d[q] = fmaf(g[co], sw[co * Cp + c + q], d[q]);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (relu_mask && !(xv[q] > 0.f)) d[q] = 0.f;
        sx[threadIdx.x * (Cp + 1) + c + q] = xv[q];
      }
      if (ok) store4f(dx + p * dcs + dco + c, d);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kHeadAcc; ++k) {
      const int pr = threadIdx.x + 256 * k;
      if (pr >= npairs) continue;
      float s = 0.f;
      if (pr < Cout * Cp) {
        const int co = pr / Cp, c = pr % Cp;
        for (int i = 0; i < kHeadTile; ++i) s = fmaf(sg[i * Cout + co], sx[i * (Cp + 1) + c], s);
      } else {
        const int co = pr - Cout * Cp;
        for (int i = 0; i < kHeadTile; ++i) s += sg[i * Cout + co];
      }
      accp[k] += s;
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kHeadAcc; ++k) {
    const int pr = threadIdx.x + 256 * k;
    if (pr < npairs) part[(int64_t)blockIdx.x * npairs + pr] = accp[k];
  }
}
// Head backward for one or two output channels (the usual 1-step prediction): every thread
// keeps the dW row(s) of its own pixels in registers across a grid-stride loop; one
// wave-shuffle + LDS reduction per workgroup at the end.  Partials: [block][Cout*Cp + Cout].
template <typename T, int COUT, int CP>
__global__ __launch_bounds__(256) void head_bwd_small_kernel(const T* __restrict__ x, int xcs, int xco,
                                                              const float* __restrict__ w, const float* __restrict__ dpred,
                                                              T* __restrict__ dx, int dcs, int dco, float* __restrict__ part,
                                                              int64_t npix, int HW, int Cin, int relu_mask) {
  __shared__ float sw[COUT * CP];
  __shared__ float red[4][COUT * CP + COUT];
  for (int i = threadIdx.x; i < COUT * CP; i += 256) {
    const int co = i / CP, c = i % CP;
    sw[i] = c < Cin ? w[co * Cin + c] : 0.f;
  }
  __syncthreads();
  float acc[COUT][CP], accb[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[co][c] = 0.f;
  }
  for (int64_t p = blockIdx.x * 256ll + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const uint32_t b = (uint32_t)p / (uint32_t)HW, r = (uint32_t)p - b * (uint32_t)HW;
    float g[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { g[co] = dpred[((int64_t)b * COUT + co) * HW + r]; accb[co] += g[co]; }
    constexpr int N = Vec16<T>::N;
    float xv[CP];                      // the whole pixel row first: all loads in flight together
#pragma unroll
    for (int c = 0; c < CP; c += N) Vec16<T>::load(x + p * xcs + xco + c, xv + c);
#pragma unroll
    for (int c = 0; c < CP; c += N) {
      float d[N];
#pragma unroll
      for (int q = 0; q < N; ++q) d[q] = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int q = 0; q < N; ++q) {
          d[q] = fmaf(g[co], sw[co * CP + c + q], d[q]);
          acc[co][c + q] = fmaf(g[co], xv[c + q], acc[co][c + q]);
        }
      if (relu_mask) {
#pragma unroll
        for (int q = 0; q < N; ++q) d[q] = xv[c + q] > 0.f ? d[q] : 0.f;
      }
      Vec16<T>::store(dx + p * dcs + dco + c, d);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const float v = wave_sum(acc[co][c]);
      if (lane == 0) red[wave][co * CP + c] = v;
    }
    const float vb = wave_sum(accb[co]);
    if (lane == 0) red[wave][COUT * CP + co] = vb;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < COUT * CP + COUT; i += 256)
    part[(int64_t)blockIdx.x * (COUT * CP + COUT) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// Head backward for one or two output channels on 32-channel inputs, CP/N lanes per pixel (see head_fwd_quad_kernel):
// every lane owns one 16-byte piece of its pixels' rows -- dx piece out, dW piece accumulated in registers across a
// grid-stride loop -- and the lanes owning the same piece meet once, at the end.  Partials: [block][COUT*CP + COUT].
template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_bwd_quad_kernel(const T* __restrict__ x, int xcs, int xco,
                                                             const float* __restrict__ w, const float* __restrict__ dpred,
                                                             T* __restrict__ dx, int dcs, int dco, float* __restrict__ part,
                                                             uint32_t npix, FastDiv dHW, int Cin, int relu_mask) {
  constexpr int CP = 32, N = Vec16<T>::N, LPP = CP / N;
  __shared__ float red[4][COUT * CP + COUT];
  const int cg = threadIdx.x % LPP;
  float wr[COUT][N];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int q = 0; q < N; ++q) { const int c = cg * N + q; wr[co][q] = c < Cin ? w[co * Cin + (c < Cin ? c : 0)] : 0.f; }
  float acc[COUT][N], accb[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int q = 0; q < N; ++q) acc[co][q] = 0.f;
  }
  const uint32_t HW = dHW.d, stride = gridDim.x * (256 / LPP);
  constexpr int U = 4;                     // pixels in flight per lane
  for (uint32_t pb = blockIdx.x * (256 / LPP) + threadIdx.x / LPP; pb < npix; pb += U * stride) {
    float xv[U][N], g[U][COUT];
    uint32_t pp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t p = pb + u * stride;
      pp[u] = p < npix ? p : npix - 1;     // clamped: loaded, contributes nothing, not stored
      Vec16<T>::load(x + (int64_t)pp[u] * xcs + xco + cg * N, xv[u]);
      const uint32_t b = fdiv(pp[u], dHW), r = pp[u] - b * HW;
#pragma unroll
      for (int co = 0; co < COUT; ++co) g[u][co] = dpred[((int64_t)b * COUT + co) * HW + r];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = pb + u * stride < npix;
      float d[N];
#pragma unroll
      for (int q = 0; q < N; ++q) d[q] = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float gv = live ? g[u][co] : 0.f;
        if (cg == 0) accb[co] += gv;
#pragma unroll
        for (int q = 0; q < N; ++q) {
          d[q] = fmaf(gv, wr[co][q], d[q]);
          acc[co][q] = fmaf(gv, xv[u][q], acc[co][q]);
        }
      }
      if (relu_mask) {
#pragma unroll
        for (int q = 0; q < N; ++q) d[q] = xv[u][q] > 0.f ? d[q] : 0.f;
      }
      if (live) Vec16<T>::store(dx + (int64_t)pp[u] * dcs + dco + cg * N, d);
    }
  }
  // lanes that own the same 16-byte piece (lane % LPP) meet: xor offsets LPP .. 32, a fixed order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float v = acc[co][q];
#pragma unroll
      for (int o = LPP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane < LPP) red[wave][co * CP + lane * N + q] = v;
    }
    float vb = accb[co];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) vb += __shfl_xor(vb, o, 64);
    if (lane == 0) red[wave][COUT * CP + co] = vb;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < COUT * CP + COUT; i += 256)
    part[(int64_t)blockIdx.x * (COUT * CP + COUT) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}
// Head backward for 3..16 output channels on a 32-channel input (the multi-step prediction head):
// four threads share a pixel, each owning 8 of its input channels, so the dW rows a thread keeps
// in registers are COUTP x 8 instead of COUTP x 32 and the loads stay 16 bytes wide.  Partials have
// the layout of the kernel above.  (Measured alternatives that were slower at 128x128x64 images:
// eight threads per pixel exchanging dpred by lane shuffles, and four pixels in flight per thread;
// the strided dpred planes, not latency, bound this kernel.)
template <typename T, int COUTP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COUTP <= 12 ? 3 : 2, COUTP <= 12 ? 3 : 2))) void head_bwd_mid_kernel(const T* __restrict__ x, int xcs, int xco,
                                                            const float* __restrict__ w, const float* __restrict__ dpred,
                                                            T* __restrict__ dx, int dcs, int dco, float* __restrict__ part,
                                                            uint32_t npix, int HW, FastDiv dHW, int Cin, int Cout, int relu_mask) {
  constexpr int CP = 32;
  __shared__ __attribute__((aligned(16))) float sw[COUTP * CP];
  __shared__ float red[4][COUTP * CP + COUTP];
  for (int i = threadIdx.x; i < COUTP * CP; i += 256) {
    const int co = i / CP, c = i % CP;
    sw[i] = (co < Cout && c < Cin) ? w[co * Cin + c] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 3;                       // channels 8*cg .. 8*cg+7
  float acc[COUTP][8], accb[COUTP];
#pragma unroll
  for (int co = 0; co < COUTP; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[co][q] = 0.f;
  }
  constexpr int N = Vec16<T>::N;
  // dpred is read through a buffer resource: one 32-bit offset per pixel in a register, the channel plane as a
  // wave-uniform offset (the launcher checks that the tensor is below 2 GiB); as 64-bit pointers the COUTP plane
  // addresses and their loop-invariant parts took two dozen registers
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dpred), 0, 0x7fffffff, 0x00020000);
  for (uint32_t p = blockIdx.x * 64u + (threadIdx.x >> 2); p < npix; p += gridDim.x * 64u) {
    const uint32_t b = fdiv(p, dHW), r = p - b * (uint32_t)HW;
    const uint32_t goff = (b * (uint32_t)(Cout * HW) + r) * 4u;
    float g[COUTP];
#pragma unroll
    for (int co = 0; co < COUTP; ++co) {                // clamped plane, then select: all loads in flight
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, (int)goff, (co < Cout ? co : 0) * HW * 4, 0));
      g[co] = co < Cout ? v : 0.f;
      accb[co] += g[co];
    }
    float xv[8], d[8];
#pragma unroll
    for (int c = 0; c < 8; c += N) Vec16<T>::load(x + (int64_t)p * xcs + xco + cg * 8 + c, xv + c);
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = 0.f;
    // The weights are read from LDS in every iteration (two 16-byte broadcast reads per output channel), through an
    // offset the compiler cannot see through: hoisted out of the loop they take COUTP x 8 registers next to the COUTP x 8
    // accumulators, which left ONE wave per SIMD to cover the latency of thirteen loads per pixel.
    int woff = cg * 8;
#pragma unroll
    for (int co = 0; co < COUTP; ++co) {
      // every second channel the offset passes through an empty statement together with a running sum: the reads
      // of the next two channels cannot be issued before the sums of the last two exist (left alone, all 2 x COUTP
      // reads are issued first, into registers that are not there)
      if (co % 2 == 0) asm volatile("" : "+v"(woff), "+v"(d[0]));
      const f32x4_t* swv = (const f32x4_t*)(sw + woff);
      const f32x4_t w0 = swv[co * (CP / 4)], w1 = swv[co * (CP / 4) + 1];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        d[q] = fmaf(g[co], q < 4 ? w0[q & 3] : w1[q & 3], d[q]);
        acc[co][q] = fmaf(g[co], xv[q], acc[co][q]);
      }
    }
    if (relu_mask) {
#pragma unroll
      for (int q = 0; q < 8; ++q) d[q] = xv[q] > 0.f ? d[q] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; c += N) Vec16<T>::store(dx + (int64_t)p * dcs + dco + cg * 8 + c, d + c);
  }
  // lanes with equal (lane & 3) hold the same channel group: butterfly over the other lane bits
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int co = 0; co < COUTP; ++co) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[co][q];
#pragma unroll
      for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane < 4) red[wave][co * CP + cg * 8 + q] = v;
    }
    float vb = cg == 0 ? accb[co] : 0.f;                // the four threads of a pixel all saw its dpred
    vb = wave_sum(vb);
    if (lane == 0) red[wave][COUTP * CP + co] = vb;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cout * CP + Cout; i += 256) {
    const int src = i < Cout * CP ? i : COUTP * CP + (i - Cout * CP);
    part[(int64_t)blockIdx.x * (Cout * CP + Cout) + i] = (red[0][src] + red[1][src]) + (red[2][src] + red[3][src]);
  }
}
__global__ __launch_bounds__(64) void head_finalize_kernel(const float* __restrict__ part, int nblocks, int Cin, int Cp,
                                                            int Cout, float* __restrict__ dW, float* __restrict__ db,
                                                            float alpha, float beta) {
  // one wave per (co, c) pair: strided partial sums, then a shuffle reduction (fixed order)
  const int npairs = Cout * Cp + Cout;
  const int pr = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 64) s += part[(int64_t)i * npairs + pr];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x != 0) return;
  if (pr < Cout * Cp) {
    const int co = pr / Cp, c = pr % Cp;
    if (c < Cin) {
      float* o = dW + co * Cin + c;
      *o = alpha * s + (beta != 0.f ? beta * *o : 0.f);
    }
  } else {
    float* o = db + (pr - Cout * Cp);
    *o = alpha * s + (beta != 0.f ? beta * *o : 0.f);
  }
}
constexpr int kHeadBlocks = 768;          // partial-sum slots in the workspace; the mid kernel runs three workgroups per CU
constexpr int kHeadBlocksSmall = 512;     // the other variants: two per CU
size_t head_ws_bytes(int Cin, int Cout) { return (size_t)kHeadBlocks * (Cout * pad_ch(Cin) + Cout) * sizeof(float); }
int launch_head_bwd(const void* x, int xcs, int xco, const float* w, const float* dpred, void* dx, int dcs, int dco,
                    float* dW, float* db, float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W,
                    int Cin, int Cout, int relu_mask, int dtype, hipStream_t s) {
  MI_REQUIRE(pad_ch(Cin) <= kHeadMaxCin && Cout <= kHeadMaxCout, "head: Cin <= 64 and Cout <= 16 supported");
  if (ws_bytes < head_ws_bytes(Cin, Cout)) return fail(MI_ESIZE, "head_bwd: workspace too small");
  const int Cp = pad_ch(Cin);
  const int64_t npix = (int64_t)B * H * W;
  MI_REQUIRE(npix < (1ll << 31), "head: pixel count out of range");
  const int64_t ntiles = (npix + kHeadTile - 1) / kHeadTile;
  const int nblocks = (int)(ntiles < kHeadBlocksSmall ? ntiles : kHeadBlocksSmall);
  if (Cout <= 2 && Cp == 32) {
    int nb = (int)((npix * (dtype == MI_BF16 ? 4 : 8) / 4 + 255) / 256);      // four pixels in flight per lane
    if (nb > kHeadBlocksSmall) nb = kHeadBlocksSmall;
    if (nb < 1) nb = 1;
#define HEAD_SMALL(TT, CO, CPV)                                                                                   \
    hipLaunchKernelGGL((head_bwd_quad_kernel<TT, CO>), dim3(nb), dim3(256), 0, s, (const TT*)x, xcs, xco, w, dpred, \
                       (TT*)dx, dcs, dco, (float*)ws, (uint32_t)npix, make_fastdiv(H * W), Cin, relu_mask)
    if (dtype == MI_BF16) {
      if (Cout == 1) HEAD_SMALL(__bf16, 1, 32); else HEAD_SMALL(__bf16, 2, 32);
    } else {
      if (Cout == 1) HEAD_SMALL(float, 1, 32); else HEAD_SMALL(float, 2, 32);
    }
#undef HEAD_SMALL
    int rc2 = check_hip(hipGetLastError(), "head_bwd_small");
    if (rc2) return rc2;
    const int np2 = Cout * Cp + Cout;
    hipLaunchKernelGGL(head_finalize_kernel, dim3(np2), dim3(64), 0, s, (const float*)ws, nb, Cin, Cp, Cout, dW, db, alpha, beta);
    return check_hip(hipGetLastError(), "head_finalize");
  }
  if (Cp == 32 && npix * Cout * 4 < (1ll << 31)) {
    int nb = (int)((npix + 63) / 64);
    if (nb > kHeadBlocks) nb = kHeadBlocks;
#define HEAD_MID(TT, COP)                                                                                          \
    hipLaunchKernelGGL((head_bwd_mid_kernel<TT, COP>), dim3(nb), dim3(256), 0, s, (const TT*)x, xcs, xco, w, dpred,   \
                       (TT*)dx, dcs, dco, (float*)ws, (uint32_t)npix, H * W, make_fastdiv(H * W), Cin, Cout, relu_mask)
    if (dtype == MI_BF16) {
      if (Cout <= 4) HEAD_MID(__bf16, 4); else if (Cout <= 8) HEAD_MID(__bf16, 8); else if (Cout <= 12) HEAD_MID(__bf16, 12); else HEAD_MID(__bf16, 16);
    } else {
      if (Cout <= 4) HEAD_MID(float, 4); else if (Cout <= 8) HEAD_MID(float, 8); else if (Cout <= 12) HEAD_MID(float, 12); else HEAD_MID(float, 16);
    }
#undef HEAD_MID
    int rc2 = check_hip(hipGetLastError(), "head_bwd_mid");
    if (rc2) return rc2;
    const int np2 = Cout * Cp + Cout;
    hipLaunchKernelGGL(head_finalize_kernel, dim3(np2), dim3(64), 0, s, (const float*)ws, nb, Cin, Cp, Cout, dW, db, alpha, beta);
    return check_hip(hipGetLastError(), "head_finalize");
  }
  const size_t shm = (size_t)(Cout * Cp + kHeadTile * (Cp + 1) + kHeadTile * Cout) * sizeof(float);
  if (dtype == MI_BF16)
    hipLaunchKernelGGL(head_bwd_kernel<__bf16>, dim3(nblocks), dim3(256), shm, s, (const __bf16*)x, xcs, xco, w, dpred, (__bf16*)dx, dcs, dco, (float*)ws, npix, H * W, Cin, Cp, Cout, relu_mask);
  else
    hipLaunchKernelGGL(head_bwd_kernel<float>, dim3(nblocks), dim3(256), shm, s, (const float*)x, xcs, xco, w, dpred, (float*)dx, dcs, dco, (float*)ws, npix, H * W, Cin, Cp, Cout, relu_mask);
  int rc = check_hip(hipGetLastError(), "head_bwd");
  if (rc) return rc;
  const int npairs = Cout * Cp + Cout;
  hipLaunchKernelGGL(head_finalize_kernel, dim3(npairs), dim3(64), 0, s, (const float*)ws, nblocks, Cin, Cp, Cout, dW, db, alpha, beta);
  return check_hip(hipGetLastError(), "head_finalize");
}

// ------------------------------------------------------------------------------------------
// MSE: per-workgroup partial sums (wave shuffle -> LDS), then one small fixed-order pass.
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                           int64_t n, float* __restrict__ dpred, float coef,
                                                           const float* __restrict__ gdev, float* __restrict__ part) {
  float sq = 0.f, ab = 0.f;
  const float c = coef * (gdev ? *gdev : 1.f);
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4_t p = *(const f32x4_t*)(pred + 4 * i), y = *(const f32x4_t*)(tgt + 4 * i);
    f32x4_t d = p - y;
#pragma unroll
    for (int q = 0; q < 4; ++q) { sq = fmaf(d[q], d[q], sq); ab += fabsf(d[q]); }
    if (dpred) { d *= c; *(f32x4_t*)(dpred + 4 * i) = d; }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const int64_t i = 4 * n4 + threadIdx.x;
    const float d = pred[i] - tgt[i];
    sq = fmaf(d, d, sq); ab += fabsf(d);
    if (dpred) dpred[i] = d * c;
  }
  __shared__ float s1[4], s2[4];
  sq = wave_sum(sq); ab = wave_sum(ab);
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = sq; s2[threadIdx.x >> 6] = ab; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    part[2 * blockIdx.x + 1] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
  }
}
__global__ __launch_bounds__(64) void mse_final_kernel(const float* __restrict__ part, int nblocks, double inv_n,
                                                        float* __restrict__ out3) {
  // one wave; double accumulation keeps the reported loss within an ulp of the exact mean
  double sq = 0.0, ab = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) { sq += part[2 * i]; ab += part[2 * i + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sq += __shfl_down(sq, o, 64); ab += __shfl_down(ab, o, 64); }
  if (threadIdx.x == 0) { out3[0] = (float)(sq * inv_n); out3[1] = (float)ab; out3[2] = (float)sq; }
}
constexpr int kMseBlocks = 1024;
size_t mse_ws_bytes(int64_t) { return (size_t)kMseBlocks * 2 * sizeof(float); }
int launch_mse(const float* pred, const float* target, int64_t n, float* out3, float* dpred, float gscale,
               const float* gscale_dev, void* ws, size_t ws_bytes, hipStream_t s) {
  MI_REQUIRE(n > 0, "mse: empty input");
  MI_REQUIRE(out3 != nullptr || dpred != nullptr, "mse: neither the sums nor the gradient requested");
  if (ws_bytes < mse_ws_bytes(n)) return fail(MI_ESIZE, "mse: workspace too small");
  MI_REQUIRE(((uintptr_t)pred % 16 == 0) && ((uintptr_t)target % 16 == 0) && ((uintptr_t)dpred % 16 == 0),
             "mse: pointers must be 16-byte aligned");
  int nblocks = (int)((n / 4 + 255) / 256);
  if (nblocks < 1) nblocks = 1;
  if (nblocks > kMseBlocks) nblocks = kMseBlocks;
  const float coef = gscale * 2.0f / (float)n;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nblocks), dim3(256), 0, s, pred, target, n, dpred, coef, gscale_dev, (float*)ws);
  int rc = check_hip(hipGetLastError(), "mse_partial");
  if (rc) return rc;
  if (out3 == nullptr) return MI_OK;            // gradient only: the sums are not wanted
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, nblocks, 1.0 / (double)n, out3);
  return check_hip(hipGetLastError(), "mse_final");
}

// ------------------------------------------------------------------------------------------
// Bilinear resize, align_corners=False (half-pixel centres, source index clamped at 0).
__device__ __forceinline__ void bil_src(int o, float scale, int in, int& i0, int& i1, float& l1) {
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}
__device__ __forceinline__ float ld_any(const void* p, int64_t off, int dt) {
  return dt == MI_BF16 ? (float)((const __bf16*)p)[off] : ((const float*)p)[off];
}
__device__ __forceinline__ void st_any(void* p, int64_t off, int dt, float v) {
  if (dt == MI_BF16) ((__bf16*)p)[off] = (__bf16)v; else ((float*)p)[off] = v;
}
__global__ void bilinear_fwd_kernel(const void* __restrict__ src, int64_t ssb, int64_t ssc, int64_t ssy, int64_t ssx,
                                    int sdt, void* __restrict__ dst, int64_t dsb, int64_t dsc, int64_t dsy,
                                    int64_t dsx, int ddt, int B, int C, int Hin, int Win, int Hout, int Wout,
                                    float sh, float sw, int c_fastest) {
  const int64_t total = (int64_t)B * C * Hout * Wout;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c, ox, oy; int64_t b;
    if (c_fastest) { c = (int)(i % C); int64_t r = i / C; ox = (int)(r % Wout); r /= Wout; oy = (int)(r % Hout); b = r / Hout; }
    else { ox = (int)(i % Wout); int64_t r = i / Wout; oy = (int)(r % Hout); r /= Hout; c = (int)(r % C); b = r / C; }
    int y0, y1, x0, x1; float ly, lx;
    bil_src(oy, sh, Hin, y0, y1, ly);
    bil_src(ox, sw, Win, x0, x1, lx);
    const int64_t base = b * ssb + c * ssc;
    const float v00 = ld_any(src, base + y0 * ssy + x0 * ssx, sdt), v01 = ld_any(src, base + y0 * ssy + x1 * ssx, sdt);
    const float v10 = ld_any(src, base + y1 * ssy + x0 * ssx, sdt), v11 = ld_any(src, base + y1 * ssy + x1 * ssx, sdt);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    st_any(dst, b * dsb + c * dsc + oy * dsy + ox * dsx, ddt, v);
  }
}
// Gather form of the backward: each input element sums the output gradients that read it.
__global__ void bilinear_bwd_kernel(const void* __restrict__ dout, int64_t osb, int64_t osc, int64_t osy, int64_t osx,
                                    int odt, void* __restrict__ din, int64_t isb, int64_t isc, int64_t isy,
                                    int64_t isx, int idt, int B, int C, int Hin, int Win, int Hout, int Wout,
                                    float sh, float sw, int c_fastest) {
  const int64_t total = (int64_t)B * C * Hin * Win;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c, ix, iy; int64_t b;
    if (c_fastest) { c = (int)(i % C); int64_t r = i / C; ix = (int)(r % Win); r /= Win; iy = (int)(r % Hin); b = r / Hin; }
    else { ix = (int)(i % Win); int64_t r = i / Win; iy = (int)(r % Hin); r /= Hin; c = (int)(r % C); b = r / C; }
    // candidate output rows/cols: those whose source index lies in [iy-1, iy+1)
    int oy_lo = (int)floorf(((float)iy - 0.5f) / sh - 0.5f) - 1, oy_hi = (int)ceilf(((float)iy + 1.5f) / sh - 0.5f) + 1;
    int ox_lo = (int)floorf(((float)ix - 0.5f) / sw - 0.5f) - 1, ox_hi = (int)ceilf(((float)ix + 1.5f) / sw - 0.5f) + 1;
    if (oy_lo < 0) oy_lo = 0; if (ox_lo < 0) ox_lo = 0;
    if (oy_hi > Hout - 1) oy_hi = Hout - 1; if (ox_hi > Wout - 1) ox_hi = Wout - 1;
    float acc = 0.f;
    const int64_t base = b * osb + c * osc;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float ly;
      bil_src(oy, sh, Hin, y0, y1, ly);
      float wy = 0.f;
      if (y0 == iy) wy += 1.f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float lx;
        bil_src(ox, sw, Win, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        if (wx == 0.f) continue;
        acc = fmaf(wy * wx, ld_any(dout, base + oy * osy + ox * osx, odt), acc);
      }
    }
    st_any(din, b * isb + c * isc + iy * isy + ix * isx, idt, acc);
  }
}
int launch_bilinear_fwd(const void* src, int64_t ssb, int64_t ssc, int64_t ssy, int64_t ssx, int sdt, void* dst,
                        int64_t dsb, int64_t dsc, int64_t dsy, int64_t dsx, int ddt, int B, int C, int Hin, int Win,
                        int Hout, int Wout, hipStream_t s) {
  const int64_t total = (int64_t)B * C * Hout * Wout;
  if (total == 0) return MI_OK;
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, ssb, ssc, ssy, ssx, sdt, dst, dsb,
                     dsc, dsy, dsx, ddt, B, C, Hin, Win, Hout, Wout, sh, sw, dsc == 1 ? 1 : 0);
  return check_hip(hipGetLastError(), "bilinear_fwd");
}
int launch_bilinear_bwd(const void* dout, int64_t osb, int64_t osc, int64_t osy, int64_t osx, int odt, void* din,
                        int64_t isb, int64_t isc, int64_t isy, int64_t isx, int idt, int B, int C, int Hin, int Win,
                        int Hout, int Wout, hipStream_t s) {
  const int64_t total = (int64_t)B * C * Hin * Win;
  if (total == 0) return MI_OK;
  const float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dout, osb, osc, osy, osx, odt, din, isb,
                     isc, isy, isx, idt, B, C, Hin, Win, Hout, Wout, sh, sw, isc == 1 ? 1 : 0);
  return check_hip(hipGetLastError(), "bilinear_bwd");
}

// ------------------------------------------------------------------------------------------
// Adam over a flat range (torch.optim.Adam defaults: no weight decay, no amsgrad).
struct AdamCoef { float b1, b2, eps, step_size, inv_bc2_sqrt, gs; };
// The fused multiply-adds are spelled out and contraction is off: the plain Adam kernel and the fused
// Adam + packing kernel must round identically wherever the compiler inlines this.
__device__ __forceinline__ void adam_update1(float& p, float g, float& m, float& v, const AdamCoef& c) {
#pragma clang fp contract(off)
  const float gq = g * c.gs;
  m = fmaf(gq - m, 1.f - c.b1, m);                          // exp_avg.lerp_(grad, 1 - beta1)
  const float g2 = (1.f - c.b2) * gq * gq;
  v = fmaf(v, c.b2, g2);                                    // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  const float denom = fmaf(sqrtf(v), c.inv_bc2_sqrt, c.eps);
  p = fmaf(-c.step_size, m / denom, p);
}
__device__ __forceinline__ f32x4_t adam_update4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, const AdamCoef& c) {
  f32x4_t pv = *(f32x4_t*)p, gv = *(const f32x4_t*)g, mv = *(f32x4_t*)m, vv = *(f32x4_t*)v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float pq = pv[q], mq = mv[q], vq = vv[q];
    adam_update1(pq, gv[q], mq, vq, c);
    pv[q] = pq; mv[q] = mq; vv[q] = vq;
  }
  *(f32x4_t*)p = pv; *(f32x4_t*)m = mv; *(f32x4_t*)v = vv;
  return pv;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, const AdamCoef c) {
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    adam_update4(p + 4 * i, g + 4 * i, m + 4 * i, v + 4 * i, c);
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const int64_t i = 4 * n4 + threadIdx.x;
    float pq = p[i], mq = m[i], vq = v[i];
    adam_update1(pq, g[i], mq, vq, c);
    m[i] = mq; v[i] = vq; p[i] = pq;
  }
}
static AdamCoef adam_coef(float lr, float b1, float b2, float eps, int step, float grad_scale) {
  const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
  return AdamCoef{b1, b2, eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale};
}
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int step, float grad_scale, hipStream_t s) {
  MI_REQUIRE(step >= 1 && n >= 0, "adam: step must be >= 1");
  if (n == 0) return MI_OK;
  MI_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
             "adam: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, p, g, m, v, n,
                     adam_coef(lr, b1, b2, eps, step, grad_scale));
  return check_hip(hipGetLastError(), "adam");
}

// ------------------------------------------------------------------------------------------
// Adam + packing in one launch (see ApackJobs in common.h).  A 256-thread workgroup takes one unit:
//   3x3 conv:  8 output channels x 32 input positions x 9 taps.  The sources are 8 runs of up to 288 floats
//              of the (Cout,Cin,3,3) parameter: read with 16-byte loads (p, g, m, v), updated, written back, and
//              the new values parked in LDS; both images are then written from LDS as 16-byte stores of 8
//              elements: forward rows wf[n][tap*Cp + c] (runs along c), backward rows wb[c][tap'*Np + n] (runs
//              along n, taps reversed).  Padding rows, positions and row tails are written as zeros by the units
//              that own them, so the images never need a memset.
//   transposed conv: 4 input channels x all (co, 2x2) of the (Cin,Cout,2,2) parameter = one contiguous run.
//   plain range: 1024 consecutive elements (biases, the 1x1 head): Adam only.
constexpr int kApackRow = 292;                 // floats per staged row: 288 + 4 (16-byte aligned, spreads the banks)
constexpr int kApackRows = 8;                  // output channels per 3x3 unit (measured 32 / 16 / 8: 56.5 / 50.8 / 48.6 us per step: more, smaller units spread better over the CUs)
constexpr int kApackLds = 8192;                // staged elements: a 3x3 unit (8 x 292) or 4 input channels of a transposed conv of up to 512 outputs
template <typename T> __device__ __forceinline__ void store4(T* p, const float* v);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* v) { *(f32x4_t*)p = f32x4_t{v[0], v[1], v[2], v[3]}; }
template <> __device__ __forceinline__ void store4<__bf16>(__bf16* p, const float* v) {
  *(bf16x4_t*)p = bf16x4_t{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
}
template <typename T, bool ADAM>
__global__ __launch_bounds__(256) void apack_kernel(const ApackJobs J, float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, const AdamCoef c) {
  // the updated values are staged in the images' own type: half the LDS for bf16 images, twice the workgroups per CU
  __shared__ __attribute__((aligned(16))) T sh[kApackLds];
  int j = 0;
  while (j < J.n - 1 && (int)blockIdx.x >= J.job[j].unit_end) ++j;
  const ApackJob& q = J.job[j];
  const int unit = (int)blockIdx.x - (j > 0 ? J.job[j - 1].unit_end : 0);
  const int t = threadIdx.x;
  if (unit >= q.units_main) {                 // plain Adam range behind the weight
    if (!ADAM) return;
    const int i = (unit - q.units_main) * 1024 + 4 * t;
    if (i < q.tail_n) {
      const int64_t o = q.tail_off + i;
      adam_update4(p + o, g + o, m + o, v + o, c);
    }
    return;
  }
  T* wf = (T*)q.wf;
  T* wb = (T*)q.wb;
  if (q.kind == 0) {
    const int npb = (q.Cp + 31) / 32;
    const int nbk = unit / npb, pb = unit - nbk * npb;
    const int n0 = nbk * kApackRows, c0 = pb * 32;
    const int ns = min(kApackRows, q.Np - n0), cs = min(32, q.Cp - c0);              // stored rows / positions of this unit
    const int nr = max(0, min(kApackRows, q.Cout - n0)), cr = max(0, min(32, q.Cin - c0));   // real ones
    const int vpr = cr * 9 / 4;                                               // 16-byte vectors per source run
    // three vectors of every array in flight per thread and pass: loads first, then updates and stores
    for (int base = 0; base < nr * vpr; base += 3 * 256) {
      int64_t o[3];
      int so[3];
      f32x4_t pv[3], gv[3], mv[3], vv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int idx = base + u * 256 + t;
        const int ic = idx < nr * vpr ? idx : 0;
        const int r = ic / vpr, k = ic - r * vpr;
        o[u] = q.w_off + ((int64_t)(n0 + r) * q.Cin + c0) * 9 + 4 * k;
        so[u] = idx < nr * vpr ? r * kApackRow + 4 * k : -1;
        pv[u] = *(const f32x4_t*)(p + o[u]);
        if (ADAM) { gv[u] = *(const f32x4_t*)(g + o[u]); mv[u] = *(const f32x4_t*)(m + o[u]); vv[u] = *(const f32x4_t*)(v + o[u]); }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (so[u] < 0) continue;
        if (ADAM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pq = pv[u][e], mq = mv[u][e], vq = vv[u][e];
            adam_update1(pq, gv[u][e], mq, vq, c);
            pv[u][e] = pq; mv[u][e] = mq; vv[u][e] = vq;
          }
          *(f32x4_t*)(p + o[u]) = pv[u]; *(f32x4_t*)(m + o[u]) = mv[u]; *(f32x4_t*)(v + o[u]) = vv[u];
        }
        const float val[4] = {pv[u][0], pv[u][1], pv[u][2], pv[u][3]};
        store4<T>(sh + so[u], val);
      }
    }
    __syncthreads();
    const int pg = cs / 8;                     // 8-position groups per (row, tap)
    for (int item = t; item < ns * 9 * pg; item += 256) {
      const int p8 = item % pg, rt = item / pg, tap = rt % 9, r = rt / 9;
      float val[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (r < nr && p8 * 8 + e < cr) ? to_f<T>(sh[r * kApackRow + (p8 * 8 + e) * 9 + tap]) : 0.f;
      store8<T>(wf + (int64_t)(n0 + r) * q.Kp + tap * q.Cp + c0 + p8 * 8, val);
    }
    if (pb == 0) {                             // row tails [9*Cp, Kp)
      const int tail = q.Kp - 9 * q.Cp;
      for (int i = t; i < ns * tail; i += 256) wf[(int64_t)(n0 + i / tail) * q.Kp + 9 * q.Cp + i % tail] = from_f<T>(0.f);
    }
    const int ng = ns / 8;                     // 8-channel groups per (position, tap)
    for (int item = t; item < cs * 9 * ng; item += 256) {
      const int n8 = item % ng, pt = item / ng, tap = pt % 9, pl = pt / 9;
      float val[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (pl < cr && n8 * 8 + e < nr) ? to_f<T>(sh[(n8 * 8 + e) * kApackRow + pl * 9 + (8 - tap)]) : 0.f;
      store8<T>(wb + (int64_t)(c0 + pl) * q.Kp2 + tap * q.Np + n0 + n8 * 8, val);
    }
    if (nbk == 0) {                            // row tails [9*Np, Kp2)
      const int tail = q.Kp2 - 9 * q.Np;
      for (int i = t; i < cs * tail; i += 256) wb[(int64_t)(c0 + i / tail) * q.Kp2 + 9 * q.Np + i % tail] = from_f<T>(0.f);
    }
    return;
  }
  // transposed conv: input channels ci0 .. ci0+3 (units run over the Kp columns of the forward image)
  const int ci0 = unit * 4;
  const int rr = max(0, min(4, q.Cin - ci0));
  const int Cout = q.Cout, Np = q.Np;
  for (int idx = t; idx < rr * Cout; idx += 256) {       // one 16-byte vector = the 2x2 taps of one (ci, co)
    const int64_t o = q.w_off + ((int64_t)ci0 * Cout + idx) * 4;
    const f32x4_t pv = ADAM ? adam_update4(p + o, g + o, m + o, v + o, c) : *(const f32x4_t*)(p + o);
    const float val[4] = {pv[0], pv[1], pv[2], pv[3]};
    store4<T>(sh + 4 * idx, val);
  }
  __syncthreads();
  for (int item = t; item < 4 * Np; item += 256) {       // forward image wf[(g*Np + co)][ci]: 4 consecutive ci
    const int gq = item / Np, co = item - gq * Np;
    float val[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) val[e] = (co < Cout && e < rr) ? to_f<T>(sh[(e * Cout + co) * 4 + gq]) : 0.f;
    store4<T>(wf + (int64_t)(gq * Np + co) * q.Kp + ci0, val);
  }
  const int ng = Np / 8;
  for (int item = t; item < 4 * 4 * ng; item += 256) {   // backward image wb[ci][g*Np + co]: runs along co
    const int n8 = item % ng, pgq = item / ng, gq = pgq & 3, pl = pgq >> 2;
    if (ci0 + pl >= q.Cp) continue;
    float val[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = (pl < rr && n8 * 8 + e < Cout) ? to_f<T>(sh[(pl * Cout + n8 * 8 + e) * 4 + gq]) : 0.f;
    store8<T>(wb + (int64_t)(ci0 + pl) * q.Kp2 + gq * Np + n8 * 8, val);
  }
  const int tail = q.Kp2 - 4 * Np;
  for (int i = t; i < 4 * tail; i += 256) {
    const int pl = i / tail;
    if (ci0 + pl < q.Cp) wb[(int64_t)(ci0 + pl) * q.Kp2 + 4 * Np + i % tail] = from_f<T>(0.f);
  }
}
// Whether a layer's weight can go through the fused kernel: 16-byte aligned source runs (input channels a
// multiple of 4, no padding between the two halves of a concatenated input), the staged block fits in LDS.
bool apack_layer_ok(int convt, int Cout, int Cin, int split, int64_t w_off) {
  if (w_off % 4 != 0 || Cin % 4 != 0) return false;
  if (convt) return 4 * 4 * (int64_t)Cout <= kApackLds && pad_ch(Cout) % 16 == 0;
  return split == 0 || pad_ch(split) == split;
}
int apack_jobs_add(ApackJobs& J, int kind, int Cout, int Cin, int split, int64_t w_off, int64_t tail_off, int tail_n,
                   void* wf, void* wb) {
  if (J.n >= kMaxApackJobs) return 0;
  ApackJob& q = J.job[J.n];
  q.kind = kind; q.Cout = Cout; q.Cin = Cin; q.w_off = w_off; q.tail_off = tail_off; q.tail_n = tail_n; q.wf = wf; q.wb = wb;
  q.Np = q.Cp = q.Kp = q.Kp2 = 0; q.units_main = 0;
  if (kind == 0) {
    q.Np = pad_ch(Cout); q.Cp = concat_cp(Cin, split); q.Kp = (int)kpad(9 * q.Cp); q.Kp2 = (int)kpad(9 * q.Np);
    q.units_main = ((q.Np + kApackRows - 1) / kApackRows) * ((q.Cp + 31) / 32);
  } else if (kind == 1) {
    q.Np = pad_ch(Cout); q.Cp = pad_ch(Cin); q.Kp = (int)kpad(q.Cp); q.Kp2 = (int)kpad(4 * q.Np);
    q.units_main = q.Kp / 4;
  }
  q.unit_end = (J.n > 0 ? J.job[J.n - 1].unit_end : 0) + q.units_main + (tail_n + 1023) / 1024;
  ++J.n;
  return 1;
}
int launch_apack(const ApackJobs& J, int dtype, float* p, const float* g, float* m, float* v, float lr, float b1,
                 float b2, float eps, int step, float grad_scale, hipStream_t s) {
  MI_REQUIRE(J.n > 0 && p != nullptr, "adam+pack: nothing to do");
  const bool adam = g != nullptr;
  MI_REQUIRE(!adam || (m != nullptr && v != nullptr && step >= 1), "adam+pack: moments and a step count >= 1 required");
  MI_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) && ((uintptr_t)v % 16 == 0),
             "adam+pack: pointers must be 16-byte aligned");
  const dim3 grid(J.job[J.n - 1].unit_end);
  const AdamCoef c = adam ? adam_coef(lr, b1, b2, eps, step, grad_scale) : AdamCoef{};
  if (dtype == MI_BF16) {
    if (adam) hipLaunchKernelGGL((apack_kernel<__bf16, true>), grid, dim3(256), 0, s, J, p, g, m, v, c);
    else hipLaunchKernelGGL((apack_kernel<__bf16, false>), grid, dim3(256), 0, s, J, p, g, m, v, c);
  } else {
    if (adam) hipLaunchKernelGGL((apack_kernel<float, true>), grid, dim3(256), 0, s, J, p, g, m, v, c);
    else hipLaunchKernelGGL((apack_kernel<float, false>), grid, dim3(256), 0, s, J, p, g, m, v, c);
  }
  return check_hip(hipGetLastError(), "adam+pack");
}

// ------------------------------------------------------------------------------------------
// Fixed-order combination of weight-gradient partials into the reference's parameter layout.
// A 256-thread workgroup handles 256/G consecutive partial columns; the split range is cut in G
// interleaved groups (G = 1..16, a power of two chosen from the split count) and each thread
// keeps 4 independent sums, so a layer with few outputs and many splits still spreads over
// hundreds of workgroups with all its loads in flight at once, and a layer with few splits keeps
// every thread busy.  The combination order is fixed, hence bitwise repeatable.
__device__ __forceinline__ f32x4_t sum_splits4(const float* __restrict__ p, int64_t stride, int z0, int splits, int G) {
  // four consecutive columns per thread (16-byte loads), 8 loads in flight, two interleaved sums
  f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  int z = z0;
  for (; z + 7 * G < splits; z += 8 * G) {
    f32x4_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *(const f32x4_t*)(p + (int64_t)(z + k * G) * stride);
#pragma unroll
    for (int k = 0; k < 8; k += 2) { a0 += v[k]; a1 += v[k + 1]; }
  }
  for (; z + G < splits; z += 2 * G) {
    const f32x4_t v0 = *(const f32x4_t*)(p + (int64_t)z * stride), v1 = *(const f32x4_t*)(p + (int64_t)(z + G) * stride);
    a0 += v0; a1 += v1;
  }
  for (; z < splits; z += G) a0 += *(const f32x4_t*)(p + (int64_t)z * stride);
  return a0 + a1;
}
__device__ __forceinline__ f32x4_t combine_groups4(f32x4_t v, float* sh, int G, int cols) {
  // sh: [G groups][cols quads]; returns the total in threads 0..cols-1
  if (G == 1) return v;
  ((f32x4_t*)sh)[threadIdx.x] = v;
  __syncthreads();
  const int o = threadIdx.x & (cols - 1);
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < G; ++g) s += ((const f32x4_t*)sh)[g * cols + o];
  return s;
}
static inline int finalize_groups(int splits, int64_t outputs) {
  // as many split groups as it takes to reach ~2048 workgroups, at most 16 and at most the split count
  int g = 1;
  while (g < 16 && g < splits && outputs * (2 * g) <= 2048ll * 256) g *= 2;
  return g;
}
// The three bodies below run as workgroup `vb` of `nb` of their job (one launch serves several layers).
// conv3: one workgroup per (output channel co, block of up to 64 packed input positions).  Its 9 x 64
// partial columns are summed over the splits with 16-byte loads (the split range cut into G interleaved
// groups when the block has fewer than 256 quads, combined through LDS in a fixed order), transposed
// through LDS and written as ONE contiguous run of dW[co][ci][3][3]: the (tap, c) -> (c, tap) transposition
// as scattered 4-byte stores 36 bytes apart was what the combine pass spent its time on.
__device__ __forceinline__ void finalize_conv3_run_body(const FinJob& q, int vb, float* sh) {
  const int Cp = q.Cp, splits = q.splits;
  const int K = 9 * Cp;
  const int ny = (Cp + 63) / 64;
  const int co = vb / ny, pos0 = (vb - co * ny) * 64;
  const int ncols = min(64, Cp - pos0), qpt = ncols / 4, nq = 9 * qpt;      // quads per tap, quads in all
  int G = 256 / nq;
  if (G > splits) G = splits;
  if (G > 16) G = 16;
  const int split_pad = pad_ch(q.split);
  f32x4_t* sh4 = (f32x4_t*)sh;                  // [G][nq] quads
  float* so = sh + 4 * 256;                     // [9][64] sums
  const int g = threadIdx.x / nq, qd = threadIdx.x - g * nq;
  const int tap = qd / qpt, c4 = (qd - tap * qpt) * 4;
  if (g < G) {
    const f32x4_t v = sum_splits4(q.part + (int64_t)co * K + tap * Cp + pos0 + c4, (int64_t)q.Npad * K, g, splits, G);
    if (G > 1) sh4[g * nq + qd] = v;
    else { so[tap * 64 + c4] = v[0]; so[tap * 64 + c4 + 1] = v[1]; so[tap * 64 + c4 + 2] = v[2]; so[tap * 64 + c4 + 3] = v[3]; }
  }
  if (G > 1) {
    __syncthreads();
    if (g == 0) {
      f32x4_t v = sh4[qd];
      for (int k = 1; k < G; ++k) v += sh4[k * nq + qd];
      so[tap * 64 + c4] = v[0]; so[tap * 64 + c4 + 1] = v[1]; so[tap * 64 + c4 + 2] = v[2]; so[tap * 64 + c4 + 3] = v[3];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 9 * ncols; idx += 256) {
    const int cl = idx / 9, tp = idx - cl * 9;
    const int ci = unpos(pos0 + cl, q.Cin, q.split, split_pad);
    if (ci < 0) continue;
    float* out = q.dW + ((int64_t)co * q.Cin + ci) * 9 + tp;
    *out = q.alpha * so[tp * 64 + cl] + (q.beta != 0.f ? q.beta * *out : 0.f);
  }
  if (pos0 == 0 && threadIdx.x < 64 && q.db != nullptr) {
    // bias gradient: lanes take the splits round-robin, then a fixed-order shuffle tree
    float v = 0.f;
    if (q.sum)
      for (int z = threadIdx.x; z < splits; z += 64) v += q.sum[(int64_t)z * q.Npad + co];
    v = wave_sum(v);
    if (threadIdx.x == 0) q.db[co] = q.alpha * v + (q.beta != 0.f ? q.beta * q.db[co] : 0.f);
  }
}
// Transposed conv: partials [z][Cp][4*Npad], column sums [z][4*Npad]; Cp = padded input channels.  One workgroup
// per (input channel ci, block of 64 output channels): 4 x 64 columns in, one contiguous run of dW[ci][co][2][2] out.
__device__ __forceinline__ void finalize_convt_run_body(const FinJob& q, int vb, float* sh) {
  const int Npad = q.Npad, Cout = q.Cout, splits = q.splits;
  const int K = 4 * Npad;
  const int ny = (Npad + 63) / 64;
  const int ci = vb / ny, co0 = (vb - ci * ny) * 64;
  const int ncols = min(64, Npad - co0), qpt = ncols / 4, nq = 4 * qpt;
  int G = 256 / nq;
  if (G > splits) G = splits;
  if (G > 16) G = 16;
  f32x4_t* sh4 = (f32x4_t*)sh;
  float* so = sh + 4 * 256;                     // [4][64]
  const int g = threadIdx.x / nq, qd = threadIdx.x - g * nq;
  const int tg = qd / qpt, c4 = (qd - tg * qpt) * 4;
  if (g < G) {
    const f32x4_t v = sum_splits4(q.part + (int64_t)ci * K + tg * Npad + co0 + c4, (int64_t)q.Cp * K, g, splits, G);
    if (G > 1) sh4[g * nq + qd] = v;
    else { so[tg * 64 + c4] = v[0]; so[tg * 64 + c4 + 1] = v[1]; so[tg * 64 + c4 + 2] = v[2]; so[tg * 64 + c4 + 3] = v[3]; }
  }
  if (G > 1) {
    __syncthreads();
    if (g == 0) {
      f32x4_t v = sh4[qd];
      for (int k = 1; k < G; ++k) v += sh4[k * nq + qd];
      so[tg * 64 + c4] = v[0]; so[tg * 64 + c4 + 1] = v[1]; so[tg * 64 + c4 + 2] = v[2]; so[tg * 64 + c4 + 3] = v[3];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 4 * ncols; idx += 256) {
    const int cl = idx >> 2, tp = idx & 3;
    if (co0 + cl >= Cout) continue;
    float* out = q.dW + ((int64_t)ci * Cout + co0 + cl) * 4 + tp;
    *out = q.alpha * so[tp * 64 + cl] + (q.beta != 0.f ? q.beta * *out : 0.f);
  }
  if (ci == 0 && q.db != nullptr && threadIdx.x < ncols && co0 + threadIdx.x < Cout) {
    const int co = co0 + threadIdx.x;
    float v = 0.f;
    if (q.sum)
      for (int z = 0; z < splits; ++z)
        for (int tp = 0; tp < 4; ++tp) v += q.sum[(int64_t)z * K + tp * Npad + co];
    q.db[co] = q.alpha * v + (q.beta != 0.f ? q.beta * q.db[co] : 0.f);
  }
}
// Layers with MANY splits and few outputs (the shallow levels: 32-64 channels, up to a few hundred slabs):
// work items are quads of four consecutive partial columns, 256/G quads per workgroup, the split range cut
// into G interleaved groups -- hundreds of workgroups with all loads in flight; the scattered 4-byte stores
// do not matter at these output sizes.
__device__ __forceinline__ void finalize_conv3_body(const FinJob& q, int vb, int nb, float* sh) {
  const int Cp = q.Cp, Cout = q.Cout, G = q.G;
  const int K = 9 * Cp;
  const int split_pad = pad_ch(q.split);
  const int64_t wquads = (int64_t)Cout * K / 4, total = wquads + (Cout + 3) / 4;
  const int cols = 256 / G;
  const int o = threadIdx.x & (cols - 1), zg = threadIdx.x / cols;
  for (int64_t base = (int64_t)vb * cols; base < total; base += (int64_t)nb * cols) {
    const int64_t i4 = base + o;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    int co = 0, k = 0;
    const bool is_w = i4 < wquads, live = i4 < total;
    if (live) {
      if (is_w) {
        const uint32_t i = (uint32_t)i4 * 4u;
        co = (int)(i / (uint32_t)K); k = (int)(i - (uint32_t)co * (uint32_t)K);
        v = sum_splits4(q.part + (int64_t)co * K + k, (int64_t)q.Npad * K, zg, q.splits, G);
      } else {
        co = (int)(i4 - wquads) * 4;
        if (q.sum) v = sum_splits4(q.sum + co, q.Npad, zg, q.splits, G);
      }
    }
    const f32x4_t s = combine_groups4(v, sh, G, cols);
    if (zg == 0 && live) {
      if (is_w) {
        const int tap = k / Cp, pos = k - tap * Cp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ci = unpos(pos + e, q.Cin, q.split, split_pad);
          if (ci >= 0) {
            float* out = q.dW + ((int64_t)co * q.Cin + ci) * 9 + tap;
            *out = q.alpha * s[e] + (q.beta != 0.f ? q.beta * *out : 0.f);
          }
        }
      } else if (q.db != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < Cout) q.db[co + e] = q.alpha * s[e] + (q.beta != 0.f ? q.beta * q.db[co + e] : 0.f);
      }
    }
    if (G > 1) __syncthreads();
  }
}
__device__ __forceinline__ void finalize_convt_body(const FinJob& q, int vb, int nb, float* sh) {
  const int Npad = q.Npad, Cin = q.Cin, Cout = q.Cout, G = q.G;
  const int K = 4 * Npad;
  const int64_t wquads = (int64_t)Cin * K / 4, total = wquads + (Cout + 3) / 4;
  const int cols = 256 / G;
  const int o = threadIdx.x & (cols - 1), zg = threadIdx.x / cols;
  for (int64_t base = (int64_t)vb * cols; base < total; base += (int64_t)nb * cols) {
    const int64_t i4 = base + o;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    int ci = 0, k = 0;
    const bool is_w = i4 < wquads, live = i4 < total;
    const int co4 = (int)(i4 - wquads) * 4;
    if (live) {
      if (is_w) {
        const uint32_t i = (uint32_t)i4 * 4u;
        ci = (int)(i / (uint32_t)K); k = (int)(i - (uint32_t)ci * (uint32_t)K);
        v = sum_splits4(q.part + (int64_t)ci * K + k, (int64_t)q.Cp * K, zg, q.splits, G);
      } else if (q.sum) {
        for (int g = 0; g < 4; ++g) v += sum_splits4(q.sum + g * Npad + co4, K, zg, q.splits, G);
      }
    }
    const f32x4_t s = combine_groups4(v, sh, G, cols);
    if (zg == 0 && live) {
      if (is_w) {
        const int g = k / Npad, co = k - g * Npad;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < Cout) {
            float* out = q.dW + ((int64_t)ci * Cout + co + e) * 4 + g;
            *out = q.alpha * s[e] + (q.beta != 0.f ? q.beta * *out : 0.f);
          }
      } else if (q.db != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co4 + e < Cout) q.db[co4 + e] = q.alpha * s[e] + (q.beta != 0.f ? q.beta * q.db[co4 + e] : 0.f);
      }
    }
    if (G > 1) __syncthreads();
  }
}
// One launch for all the layers of a backward segment: a chain of small dependent launches costs
// more in dispatch gaps than the sums themselves.
__global__ __launch_bounds__(256) void finalize_multi_kernel(const FinJobs J) {
  __shared__ __attribute__((aligned(16))) float sh[4 * 256 + 9 * 64];
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.job[j].block_end) ++j;
  const FinJob& q = J.job[j];
  const int first = j ? J.job[j - 1].block_end : 0;
  const int vb = (int)blockIdx.x - first, nb = q.block_end - first;
  if (q.kind == FIN_CONV3) finalize_conv3_body(q, vb, nb, sh);
  else if (q.kind == FIN_CONVT) finalize_convt_body(q, vb, nb, sh);
  else if (q.kind == FIN_CONV3_WIDE) finalize_conv3_run_body(q, vb, sh);
  else finalize_convt_run_body(q, vb, sh);
}
static int fin_push(FinJobs& J, FinJob q, int blocks) {
  if (J.n >= kMaxFinJobs) return fail(MI_EINVAL, "finalize: too many jobs in one launch");
  q.block_end = (J.n ? J.job[J.n - 1].block_end : 0) + blocks;
  J.job[J.n++] = q;
  return MI_OK;
}
int fin_jobs_add_conv3(FinJobs& J, const float* part, const float* rowsum, int splits, int Npad, int Cp, int Cout, int Cin,
                       int split, float* dW, float* db, float alpha, float beta) {
  MI_REQUIRE((int64_t)Cout * 9 * Cp + Cout < (1ll << 31), "finalize: layer too large");
  FinJob q{};
  q.part = part; q.sum = rowsum; q.splits = splits; q.Npad = Npad; q.Cp = Cp; q.Cout = Cout; q.Cin = Cin; q.split = split;
  q.dW = dW; q.db = db; q.alpha = alpha; q.beta = beta;
  if (splits <= 16) {      // few slabs: one workgroup per contiguous output run
    q.kind = FIN_CONV3_WIDE; q.G = 1;
    return fin_push(J, q, Cout * ((Cp + 63) / 64));
  }
  const int64_t quads = (int64_t)Cout * 9 * Cp / 4 + (Cout + 3) / 4;
  q.kind = FIN_CONV3; q.G = finalize_groups(splits, quads);
  return fin_push(J, q, grid_for(quads, 256 / q.G, 8192));
}
int fin_jobs_add_convt(FinJobs& J, const float* part, const float* colsum, int splits, int Cp_in, int Npad, int Cin,
                       int Cout, float* dW, float* db, float alpha, float beta) {
  const int64_t total = (int64_t)Cin * 4 * Npad + Cout;
  MI_REQUIRE(total < (1ll << 31), "finalize: layer too large");
  FinJob q{};
  q.kind = FIN_CONVT; q.part = part; q.sum = colsum; q.splits = splits; q.Npad = Npad; q.Cp = Cp_in; q.Cout = Cout; q.Cin = Cin;
  q.dW = dW; q.db = db; q.alpha = alpha; q.beta = beta;
  if (splits <= 16) {
    q.kind = FIN_CONVT_WIDE; q.G = 1;
    return fin_push(J, q, Cin * ((Npad + 63) / 64));
  }
  const int64_t quads = (int64_t)Cin * Npad + (Cout + 3) / 4;
  q.G = finalize_groups(splits, quads);
  return fin_push(J, q, grid_for(quads, 256 / q.G, 8192));
}
int launch_finalize_multi(const FinJobs& J, hipStream_t s) {
  if (J.n == 0) return MI_OK;
  hipLaunchKernelGGL(finalize_multi_kernel, dim3(J.job[J.n - 1].block_end), dim3(256), 0, s, J);
  return check_hip(hipGetLastError(), "finalize");
}

}  // namespace mi
