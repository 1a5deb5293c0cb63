// Input pipeline on the device: the raw 1-D series stays resident in HBM and every batch is
// gathered from it, so no image tensor is ever materialised on the host or copied over PCIe.
//   reference: dataset.py:323-327 (MinMax scaling), utils/image_transform.py:46-61 (sliding
//   windows folded into images), dataset.py:176-183 (K consecutive windows per sample).
// Both kernels are HBM-bound: the batch kernel writes B*(K*H_in*W_in + H_out*W_out) floats and
// reads the overlapping windows out of L2.
#include "common.h"

namespace mi {
namespace {

constexpr int kMinmaxBlocks = 1024;

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// fmin / fmax skip NaNs, as numpy.nanmin / nanmax do in MinMaxScaler.fit.
__device__ __forceinline__ void block_minmax(double lo, double hi, double* out_lo, double* out_hi) {
  __shared__ double slo[4], shi[4];
  lo = wave_min(lo); hi = wave_max(hi);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { slo[wave] = lo; shi[wave] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    *out_lo = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    *out_hi = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
  }
}
__global__ __launch_bounds__(256) void minmax_partial_kernel(const double* __restrict__ s, int64_t n,
                                                              double* __restrict__ part) {
  double lo = __builtin_nan(""), hi = __builtin_nan("");
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = s[i];
    lo = fmin(lo, v); hi = fmax(hi, v);
  }
  block_minmax(lo, hi, part + 2 * blockIdx.x, part + 2 * blockIdx.x + 1);
}
__global__ __launch_bounds__(256) void minmax_final_kernel(const double* __restrict__ part, int nparts,
                                                            double* __restrict__ out2) {
  double lo = __builtin_nan(""), hi = __builtin_nan("");
  for (int i = threadIdx.x; i < nparts; i += 256) {
    lo = fmin(lo, part[2 * i]); hi = fmax(hi, part[2 * i + 1]);
  }
  block_minmax(lo, hi, out2, out2 + 1);
}

struct WindowArgs {
  const double* series; int64_t n;
  const int64_t* targets;
  int K, in_len, out_len, y_shift, stride;   // y_shift = in_len - overlap_len
  uint32_t nx, total;                        // elements of X, of X and Y together
  double scale, offset;
  float* x; float* y;
};
// x*scale and +offset as two rounded float64 operations: the product is made opaque to the
// optimiser so that it cannot be contracted with the addition into a fused multiply-add (which
// rounds once and differs from the host's result in the last bit now and then).
__device__ __forceinline__ float scale_to_f32(double raw, double scale, double offset) {
  double prod = raw * scale;
  asm volatile("" : "+v"(prod));
  return (float)(prod + offset);
}
// One element (VEC = 1) or four consecutive elements (VEC = 4, every row length a multiple of 4)
// per thread; the scaled value is rounded once to float32: bit-identical to scaling on the host
// and casting.
template <int VEC>
__global__ __launch_bounds__(256) void window_batch_kernel(const WindowArgs a) {
  const uint32_t nthreads = gridDim.x * 256u;
  for (uint32_t e = (blockIdx.x * 256u + threadIdx.x) * VEC; e < a.total; e += nthreads * VEC) {
    const bool is_x = e < a.nx;
    const uint32_t r = is_x ? e : e - a.nx;
    const uint32_t len = is_x ? (uint32_t)a.in_len : (uint32_t)a.out_len;
    const uint32_t row = r / len, p = r - row * len;        // row = b*K + k for X, b for Y
    const uint32_t b = is_x ? row / (uint32_t)a.K : row;
    const int64_t t = a.targets[b];
    const int64_t w = is_x ? t - (a.K - 1) + (int64_t)(row - b * (uint32_t)a.K) : t;
    const int64_t pos = w * a.stride + (is_x ? 0 : a.y_shift) + p;
    float v[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const int64_t i = pos + q;
      const bool ok = i >= 0 && i < a.n;
      const double raw = a.series[ok ? i : 0];
      v[q] = ok ? scale_to_f32(raw, a.scale, a.offset) : 0.f;
    }
    float* dst = (is_x ? a.x : a.y) + r;
    if (VEC == 4) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
    else dst[0] = v[0];
  }
}


// Overlap-average of predicted horizons back into one series (visualize_results.py:443-476 of the
// reference).  One thread per raw position: the windows whose horizon covers it are g in
// [ceil((pos - in_len - out_len + 1) / stride), floor((pos - in_len) / stride)]; slot[g] is the row
// of window g in the batch or -1.  Sums run in ascending g in float64 (the reference adds in batch
// order: equal up to the rounding of a float64 sum of a few float32 terms).
struct ReconArgs {
  const float* pred; const float* gt; const int32_t* slot;
  int64_t n_slots, total;
  int out_len, in_len, stride;
  double* pred_mean; double* gt_mean; int32_t* count;
};
__global__ __launch_bounds__(256) void reconstruct_kernel(const ReconArgs a) {
  for (int64_t pos = blockIdx.x * 256ll + threadIdx.x; pos < a.total; pos += (int64_t)gridDim.x * 256) {
    const int64_t rel = pos - a.in_len;                    // offset from the start of window 0's horizon
    double ps = 0.0, gs = 0.0;
    int n = 0;
    if (rel >= 0) {
      int64_t g_hi = rel / a.stride;
      int64_t g_lo = rel - a.out_len + 1 <= 0 ? 0 : (rel - a.out_len + a.stride) / a.stride;   // ceil((rel-out_len+1)/stride)
      if (g_hi >= a.n_slots) g_hi = a.n_slots - 1;
      for (int64_t g = g_lo; g <= g_hi; ++g) {
        const int32_t row = a.slot[g];
        if (row < 0) continue;
        const int64_t off = (int64_t)row * a.out_len + (rel - g * a.stride);
        ps += (double)a.pred[off];
        gs += (double)a.gt[off];
        ++n;
      }
    }
    a.count[pos] = n;
    a.pred_mean[pos] = n ? ps / n : 0.0;
    a.gt_mean[pos] = n ? gs / n : 0.0;
  }
}

}  // namespace
}  // namespace mi

using namespace mi;

extern "C" {

size_t mi_series_minmax_ws_bytes(void) { return (size_t)kMinmaxBlocks * 2 * sizeof(double); }

int mi_series_minmax(const double* series, int64_t n, double* out2, void* ws, size_t ws_bytes, mi_stream s) {
  MI_REQUIRE(series && out2 && ws && n > 0, "series_minmax: bad arguments");
  if (ws_bytes < mi_series_minmax_ws_bytes()) return fail(MI_ESIZE, "series_minmax: workspace too small");
  int64_t nb = (n + 255) / 256;
  if (nb > kMinmaxBlocks) nb = kMinmaxBlocks;
  hipLaunchKernelGGL(minmax_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, series, n, (double*)ws);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, (const double*)ws, (int)nb, out2);
  return check_hip(hipGetLastError(), "series_minmax");
}

int64_t mi_window_count(int64_t n, int H_in, int W_in, int H_out, int W_out, int overlap_len, int stride) {
  const int64_t in_len = (int64_t)H_in * W_in, out_len = (int64_t)H_out * W_out;
  if (H_in <= 0 || W_in <= 0 || H_out <= 0 || W_out <= 0) { fail(MI_EINVAL, "window_count: bad image size"); return -1; }
  if (overlap_len < 0 || overlap_len > (in_len < out_len ? in_len : out_len)) {
    fail(MI_EINVAL, "overlap_len must be in [0, " + std::to_string(in_len < out_len ? in_len : out_len) + "], got " +
                        std::to_string(overlap_len));
    return -1;
  }
  if (stride <= 0) { fail(MI_EINVAL, "stride must be positive"); return -1; }
  const int64_t max_start = n - (in_len + out_len - overlap_len);
  if (max_start <= 0) {
    fail(MI_EINVAL, "Time series too short for given window sizes and overlap. len=" + std::to_string(n) +
                        ", input_len=" + std::to_string(in_len) + ", output_len=" + std::to_string(out_len) +
                        ", overlap_len=" + std::to_string(overlap_len));
    return -1;
  }
  return max_start / stride + 1;
}

int mi_window_batch(const double* series, int64_t n, const int64_t* targets, int B, int K, int H_in, int W_in,
                    int H_out, int W_out, int overlap_len, int stride, double scale, double offset, float* x, float* y,
                    mi_stream s) {
  MI_REQUIRE(series && targets && x && y && B > 0 && K > 0, "window_batch: bad arguments");
  const int64_t N = mi_window_count(n, H_in, W_in, H_out, W_out, overlap_len, stride);
  if (N < 0) return MI_EINVAL;
  if (N < K) return fail(MI_EINVAL, "Not enough windows (" + std::to_string(N) + ") to build sequences with K=" +
                                        std::to_string(K));
  const int64_t in_len = (int64_t)H_in * W_in, out_len = (int64_t)H_out * W_out;
  const int64_t nx = (int64_t)B * K * in_len, total = nx + (int64_t)B * out_len;
  MI_REQUIRE(total < (1ll << 31) - 4096, "window_batch: batch too large");
  MI_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "window_batch: outputs must be 16-byte aligned");
  WindowArgs a{};
  a.series = series; a.n = n; a.targets = targets;
  a.K = K; a.in_len = (int)in_len; a.out_len = (int)out_len; a.y_shift = (int)in_len - overlap_len; a.stride = stride;
  a.nx = (uint32_t)nx; a.total = (uint32_t)total;
  a.scale = scale; a.offset = offset; a.x = x; a.y = y;
  const bool vec = in_len % 4 == 0 && out_len % 4 == 0;
  const int64_t threads = vec ? total / 4 : total;
  int64_t nb = (threads + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  if (vec) hipLaunchKernelGGL(window_batch_kernel<4>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, a);
  else hipLaunchKernelGGL(window_batch_kernel<1>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, a);
  return check_hip(hipGetLastError(), "window_batch");
}

int mi_series_reconstruct(const float* pred, const float* gt, const int32_t* slot, int64_t n_slots, int B, int in_len,
                          int out_len, int stride, int64_t total, double* pred_mean, double* gt_mean, int32_t* count,
                          mi_stream s) {
  MI_REQUIRE(pred && gt && slot && pred_mean && gt_mean && count, "series_reconstruct: null pointer");
  MI_REQUIRE(B > 0 && n_slots > 0 && in_len > 0 && out_len > 0 && stride > 0 && total > 0, "series_reconstruct: bad sizes");
  MI_REQUIRE(total >= (n_slots - 1) * stride + in_len + out_len, "series_reconstruct: output shorter than the last horizon");
  ReconArgs a{pred, gt, slot, n_slots, total, out_len, in_len, stride, pred_mean, gt_mean, count};
  int64_t nb = (total + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  hipLaunchKernelGGL(reconstruct_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, a);
  return check_hip(hipGetLastError(), "series_reconstruct");
}

}  // extern "C"
