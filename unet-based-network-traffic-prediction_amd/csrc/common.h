// Internal declarations shared by the HIP sources of libunet_tm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/unet_tm_hip.h"

namespace mi {

// Channel padding of NHWC tensors and packed weights: up to 32 channels are stored as 32 (one
// bf16 MFMA k-step, the narrowest width the halo-tile kernels take), wider tensors round up to 16.
constexpr int kPad = 16;
__host__ __device__ inline int pad_ch(int c) { return c <= 32 ? 32 : (c + kPad - 1) / kPad * kPad; }
inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
inline int elem_size(int dtype) { return dtype == MI_BF16 ? 2 : 4; }

// Exact unsigned division by a runtime constant for n < 2^31:
//   q = (uint64(n) * mul) >> shift,  mul = ceil(2^(31+l)/d), l = ceil(log2 d).
struct FastDiv {
  uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = 31 + l;
  f.mul = (uint32_t)(((1ull << f.shift) + d - 1) / d);
  f.d = d;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (uint32_t)(((uint64_t)n * f.mul) >> f.shift);
}

// Device address of 256 bytes of zeros (a file-scope __device__ array of the library, resolved
// once per device).  Loader lanes whose element is padding (conv halo, out-of-range pixel,
// channel tail) read from here instead of branching around the load or zeroing the result
// afterwards: the address select happens BEFORE the load, so nothing depends on the loaded
// data until it is written to LDS and every staging load stays in flight across the MFMA
// phase.  Kernels receive it as the `zero` field of their argument struct.
const void* zero_page();

// Makes a selected load offset opaque to the optimiser.  Without it the compiler turns
// "offset = padding ? zero_page : element; load(base + offset)" back into two branches with a
// load (and a wait) in each, which serialises the staging loads of a stage.
__device__ __forceinline__ int64_t opaque_offset(int64_t v) {
  asm("" : "+v"(v));
  return v;
}

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_hip(hipError_t e, const char* what);

#define MI_REQUIRE(cond, msg)                                   \
  do {                                                          \
    if (!(cond)) return ::mi::fail(MI_EINVAL, std::string(msg)); \
  } while (0)

// ---------------------------------------------------------------- implicit-GEMM (igemm.hip)
// out[p][n] = epilogue( sum_{tap,c} in[gather(p,tap)][c] * w[n][tap*Cin + c] )
enum GatherMode { G_3x3 = 0, G_2x2S2 = 1, G_1x1 = 2,
                  G_3x3P = 3 };  // wgrad only: 3x3 on tiny images, reduction over the in-bounds pixels of the tap
enum Epilogue { E_PLAIN = 0, E_BIAS = 1, E_BIAS_RELU = 2, E_SHUFFLE_BIAS = 3 };

struct IgemmArgs {
  const void* in; int in_cs, in_co;      // gathered operand (NHWC view)
  int B, Hi, Wi;                         // its spatial size
  int Ho, Wo;                            // output pixel grid; M = B*Ho*Wo
  int Cin;                               // channels per tap (multiple of 16)
  int taps, mode;
  const void* w; int Kp;                 // packed weights [N][Kp]
  int N;                                 // output channels incl. padding (multiple of 16)
  const float* bias; int n_real;         // bias has n_real entries (unpadded parameter)
  void* out; int out_cs, out_co;
  const void* mask; int mask_cs, mask_co;  // optional: out *= (mask > 0)
  void* pool; int pool_cs, pool_co;        // optional (3x3 forward only): also write the 2x2 max-pool of the output
  // optional (3x3, DMA-fed bf16 kernel only; anything else refuses them): the operand is TWO tensors of the same
  // channel stride -- input channels from in2_c0 on are channels 0.. of in2, output channels from out2_n0 on are
  // channels 0.. of out2.  The two halves of a 32+32-channel concat then each fill whole 64-byte pixels of their
  // own tensor instead of half of every 128-byte line of a shared one.
  const void* in2; int in2_c0;
  void* out2; int out2_n0;
  // optional: the input view's pixels OVERLAP -- in_cs is smaller than Cin, so a pixel's channels from in_cs on are
  // the next pixel's first ones (finite numbers) and must meet zero weights.  A 12-channel network input then costs
  // 16 channels of memory per pixel, not the 32 its first layer's K is padded to.  The caller guarantees Cin - in_cs
  // readable, finite elements behind the last pixel.
  int in_overlap;
  // optional (3x3 forward, DMA-fed bf16 kernel with one 32-channel tile only; anything else refuses it): the 1x1
  // head (models/unet2d.py:59, :89) applied to the layer's output in the epilogue -- head_out[b][co][y][x] =
  // head_b[co] + sum_c out[b][y][x][c] * head_w[co][c], fp32, from the values as stored (rounded to bf16), in the
  // summation order of the stand-alone head kernel, hence bit-identical to it.  head_cout <= 2.
  const float* head_w; const float* head_b; float* head_out; int head_cout;
  void* kscratch; size_t kscratch_bytes;   // optional: zero-initialised scratch that lets a layer of few pixels split K across workgroups
                                           // (the kernel leaves its counters at zero; concurrent launches must not share it)
  int epi;
  int shufC;                             // E_SHUFFLE_BIAS: N = 4*shufC
  int M;
  FastDiv divHW, divW;                   // by Ho*Wo and by Wo
  const void* zero;                      // zero page (filled in by launch_igemm)
};
int launch_igemm(const IgemmArgs& a, int dtype, hipStream_t s, const char** kernel_name = nullptr);
// Whether the 3x3 forward of this shape can write the 2x2 max-pool of its output in the same launch.
bool conv3x3_fwd_fuses_pool(int B, int H, int W, int Cin, int split, int Cout, int dtype);

// ---------------------------------------------------------------- weight gradient (wgrad.hip)
// part[z][n][k] = sum_{p in split z} a[p][n] * b[gather(p,tap)][c],  k = tap*Cb + c
// rowsum[z][n]  = sum_p a[p][n]        (if want_rowsum)
// colsum[z][k]  = sum_p b[gather][k]   (if want_colsum)
struct WgradArgs {
  const void* a; int a_cs, a_co; int Na;   // plain operand, Na channels (multiple of 16)
  const void* b; int b_cs, b_co; int Cb;   // gathered operand, Cb channels per tap
  int B, Hb, Wb;                           // spatial size of b
  int Ho, Wo;                              // pixel grid of a; M = B*Ho*Wo
  int taps, mode;
  float* part;                             // [splits][Na][K], K = taps*Cb
  float* rowsum;                           // [splits][Na] or null
  float* colsum;                           // [splits][K] or null
  int splits, pix_per_split;
  int M;
  FastDiv divHW, divW;
  const void* zero;                        // zero page (filled in by launch_wgrad)
  int gx, gy, usize, units;                // workgroup deal (filled in by the launcher): see wgrad_kernel
};
// Halo-tile weight gradient (hwgrad_bf16.hip): 3x3 convs on evenly tiling images, bf16 only.
struct HwgradArgs {
  const void* dy; int dy_cs, dy_co; int Na;
  const void* x; int x_cs, x_co; int Cb;
  const void* x2; int x2_c0;                // optional: channels from x2_c0 on are channels 0.. of x2 (same stride x_cs)
  int W, R, G, pitch, HR, bands;
  int ntiles, tiles_per_split, splits;
  float* part;      // [slabs_per_split*splits][Na][9*Cb]
  float* rowsum;    // [slabs_per_split*splits][Na] or null
  const void* zero; // zero page
  FastDiv d_grp, d_pitch, d_rw, d_w, d_bands;   // by (R+2)*pitch, pitch, R*W, W, bands (filled in by the launcher)
  unsigned m_grp, m_pitch;                      // ceil(2^20 / ((R+2)*pitch)), ceil(2^20 / pitch): 24-bit division of halo rows (launcher)
};
bool hwgrad_tiling(int H, int W, int* R, int* G, int* bands, int* pitch, int* HR);
int hwgrad_slabs_per_split(int Na);
// One layer's operands for a paired weight-gradient launch (api.hip: run_conv3x3_wgrad_pair).
struct Conv3WgradCall {
  const void* x; int xcs, xco;
  const void* x2; int x2_c0;                // optional second tensor of a two-tensor input (see IgemmArgs::in2)
  const void* dy; int ycs, yco;
  float* dW; float* db;
  void* ws; size_t ws_bytes;
  int Cin, split, Cout;
};
// wgs0 = workgroup ids of layer 0.  xcd_order: ids are dealt to the 8 XCDs so that the workgroups that read the
// same dy tiles (one n-block and pixel split, all its c-blocks) share an XCD and its L2; rot[i] rotates layer i's deal.
struct HwgradBatch { HwgradArgs layer[2]; int n, wgs0; int xcd_order; int rot[2]; };
int launch_hwgrad_bf16(const HwgradArgs* layers, int n, hipStream_t s, const char** name);
int wgrad_choose_splits(int M, int Na, int K, int dtype, int* pix_per_split);
// A 3x3 layer of fewer than this many (pixels x output channels) takes the few-pixel kernel (sconv_bf16.hip).  fp32
// goes four times further (its pixel-tiled kernel needs 31 us for one tile, whatever the layer) -- except on images of
// at most 16 pixels, where a batch that large belongs to the position-major kernel.
inline int64_t sconv_limit(int dtype, int HW) { return (dtype == MI_BF16 || HW <= 16) ? (64 << 13) : (64 << 15); }
// fp32 halo-tile weight gradient of 3x3 layers (hwgrad_f32.hip): takes the generic kernel's arguments
bool hwgrad_f32_applies(int B, int H, int W, int Na, int Cb, int a_cs, int b_cs);
int hwgrad_f32_splits(int B, int H, int W, int Na, int Cb, int* pix_per_split);
int launch_hwgrad_f32(const WgradArgs& a, hipStream_t s, const char** name);
int launch_wgrad(const WgradArgs& a, int dtype, hipStream_t s, const char** kernel_name = nullptr);
// Fixed-order combination of the partials into dW / db, several layers per launch (the table is
// passed by value in the kernel arguments).
//   conv3: dW (Cout,Cin,3,3), db from partials [z][Npad][9*Cp] / row sums [z][Npad]
//   convT: dW (Cin,Cout,2,2), db from partials [z][Cp][4*Npad] / column sums [z][4*Npad] (Cp = padded Cin)
enum FinKind { FIN_CONV3 = 0, FIN_CONVT = 1, FIN_CONV3_WIDE = 2, FIN_CONVT_WIDE = 3 };
// scratch a caller may lend to the few-pixel 3x3 kernel: arrival counters (zeroed once), then partial tiles
constexpr size_t kKsplitCounterBytes = 1024;
constexpr size_t kKsplitBytes = kKsplitCounterBytes + (2u << 20);
constexpr int kMaxFinJobs = 8;
struct FinJob {
  int kind;
  const float* part; const float* sum;
  int splits, Npad, Cp, Cout, Cin, split, G;
  float* dW; float* db;
  float alpha, beta;
  int block_end;                         // running count of workgroups up to and including this job
};
struct FinJobs { int n; FinJob job[kMaxFinJobs]; };
int fin_jobs_add_conv3(FinJobs& J, const float* part, const float* rowsum, int splits, int Npad, int Cp, int Cout,
                       int Cin, int split, float* dW, float* db, float alpha, float beta);
int fin_jobs_add_convt(FinJobs& J, const float* part, const float* colsum, int splits, int Cp_in, int Npad, int Cin,
                       int Cout, float* dW, float* db, float alpha, float beta);
int launch_finalize_multi(const FinJobs& J, hipStream_t s);

// ---------------------------------------------------------------- pointwise (pointwise.hip)
int launch_nchw_to_nhwc(const float* src, void* dst, int cs, int co, int B, int C, int H, int W,
                        int dtype, hipStream_t s);
// The same into overlapping pixels: `width` (a multiple of 16, >= C, = cs) channels are written per pixel instead of
// pad_ch(C), and `tail` zero elements behind the last pixel.
int launch_nchw_to_nhwc_narrow(const float* src, void* dst, int width, int tail, int B, int C, int H, int W, int dtype, hipStream_t s);
int launch_nhwc_to_nchw(const void* src, int cs, int co, float* dst, int B, int C, int H, int W,
                        int dtype, hipStream_t s);
int launch_pack_conv3x3(const float* w, void* wfwd, void* wbwd, int Cout, int Cin, int split, int dtype,
                        hipStream_t s);
int launch_pack_convt2x2(const float* w, void* wfwd, void* wbwd, int Cin, int Cout, int dtype, hipStream_t s);
// Packing of many layers in one launch (the table is passed by value in the kernel arguments).
constexpr int kMaxPackJobs = 32;
struct PackJob {
  const float* w; void* wf; void* wb;    // wf / wb may be null (that image is not produced)
  int convt, Cout, Cin, split, Np, Cp, Kp, Kp2;
  int vec;                               // 3x3 layers: packed through LDS with 16-byte accesses (see pack_conv3x3_unit_vec)
  int chunk_end;                         // running count of 1024-element chunks up to and including this job
  uint32_t mKp, mCp, mKp2, mNp;          // fast-division multipliers of Kp, Cp, Kp2, Np (shift 31 + s*)
  uint8_t sKp, sCp, sKp2, sNp;
};
struct PackJobs { int n; PackJob job[kMaxPackJobs]; };
int pack_jobs_add(PackJobs& J, const float* w, void* wfwd, void* wbwd, int convt, int Cout, int Cin, int split);
int launch_pack_multi(const PackJobs& J, int dtype, hipStream_t s);
int launch_maxpool2_fwd(const void* x, int xcs, int xco, void* y, int ycs, int yco, int B, int H, int W,
                        int C, int dtype, hipStream_t s);
int launch_maxpool2_bwd(const void* x, int xcs, int xco, const void* dyp, int pcs, int pco,
                        const void* dskip, int scs, int sco, void* dx, int dcs, int dco, int B, int H,
                        int W, int C, int relu_mask, int dtype, hipStream_t s);
int launch_head_fwd(const void* x, int xcs, int xco, const float* w, const float* bias, float* pred,
                    int B, int H, int W, int Cin, int Cout, int dtype, hipStream_t s);
int launch_head_bwd(const void* x, int xcs, int xco, const float* w, const float* dpred, void* dx,
                    int dcs, int dco, float* dW, float* db, float alpha, float beta, void* ws,
                    size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int relu_mask, int dtype,
                    hipStream_t s);
size_t head_ws_bytes(int Cin, int Cout);
int launch_mse(const float* pred, const float* target, int64_t n, float* out3, float* dpred, float gscale,
               const float* gscale_dev, void* ws, size_t ws_bytes, hipStream_t s);
size_t mse_ws_bytes(int64_t n);
int launch_bilinear_fwd(const void* src, int64_t ssb, int64_t ssc, int64_t ssy, int64_t ssx, int sdt,
                        void* dst, int64_t dsb, int64_t dsc, int64_t dsy, int64_t dsx, int ddt, int B, int C,
                        int Hin, int Win, int Hout, int Wout, hipStream_t s);
int launch_bilinear_bwd(const void* dout, int64_t osb, int64_t osc, int64_t osy, int64_t osx, int odt,
                        void* din, int64_t isb, int64_t isc, int64_t isy, int64_t isx, int idt, int B, int C,
                        int Hin, int Win, int Hout, int Wout, hipStream_t s);
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                float eps, int step, float grad_scale, hipStream_t s);
// Adam step and weight packing in one launch: every parameter is read once, updated, and the 3x3 / transposed
// conv weights are written into their two MFMA images from the updated values (staged in LDS per block of
// 8 output x 32 input channels), instead of an Adam pass followed by two packing passes over the masters.
constexpr int kMaxApackJobs = 32;
struct ApackJob {
  int kind;                              // 0: 3x3 conv, 1: transposed conv, 2: plain range (no images)
  int Cout, Cin, Np, Cp, Kp, Kp2;
  int units_main;                        // packing units of the weight (the tail units follow)
  int tail_n;                            // elements of the plain Adam range behind the weight (bias; whole tensor for kind 2)
  int unit_end;                          // running count of units up to and including this job
  int64_t w_off, tail_off;               // element offsets into the flat buffers
  void* wf; void* wb;
};
struct ApackJobs { int n; ApackJob job[kMaxApackJobs]; };
bool apack_layer_ok(int convt, int Cout, int Cin, int split, int64_t w_off);
int apack_jobs_add(ApackJobs& J, int kind, int Cout, int Cin, int split, int64_t w_off, int64_t tail_off, int tail_n,
                   void* wf, void* wb);
int launch_apack(const ApackJobs& J, int dtype, float* p, const float* g, float* m, float* v, float lr, float b1,
                 float b2, float eps, int step, float grad_scale, hipStream_t s);

// Packed-weight geometry helpers (elements)
inline int concat_cp(int Cin, int split) {  // padded input channels incl. concat layout
  return split > 0 ? pad_ch(split) + pad_ch(Cin - split) : pad_ch(Cin);
}
inline int64_t kpad(int64_t k) { return round_up(k, 64); }

}  // namespace mi
