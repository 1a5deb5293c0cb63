// extern "C" entry points for the single operators (see include/unet_tm_hip.h) and the
// shared error/dispatch helpers.  Host code only; the kernels live in igemm_*.hip,
// wgrad_*.hip and pointwise.hip.
#include "common.h"
#include "hconv_impl.h"
#include "pconv_impl.h"

#include <cstdio>
#include <cstring>
#include <mutex>

namespace mi {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return MI_OK;
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e;
}

int launch_igemm_f32(const IgemmArgs& a, hipStream_t s, const char** name);
int launch_igemm_bf16(const IgemmArgs& a, hipStream_t s, const char** name);
int launch_hconv_f32(const HconvArgs& a, hipStream_t s, const char** name);
int launch_hconv_bf16(const HconvArgs& a, hipStream_t s, const char** name);
int launch_dconv_bf16(const HconvArgs& a, hipStream_t s, const char** name, bool dry_run = false);
bool sconv_applies(const IgemmArgs& a, int dtype);
bool sconv_pools(int H, int W, int M);
int launch_sconv(const IgemmArgs& a, int dtype, hipStream_t s, const char** name);
int launch_pconv_f32(const PconvArgs& a, hipStream_t s, const char** name);
int launch_pconv_bf16(const PconvArgs& a, hipStream_t s, const char** name);
bool tconv_eligible(const IgemmArgs& a);
int launch_tconv_bf16(const IgemmArgs& a, hipStream_t s, const char** name);
bool rgemm_eligible(const IgemmArgs& a, int dtype);
int launch_rgemm(const IgemmArgs& a, int dtype, hipStream_t s, const char** name);
int launch_wgrad_f32(const WgradArgs& a, hipStream_t s, const char** name);
int launch_wgrad_bf16(const WgradArgs& a, hipStream_t s, const char** name);

// ---- zero page: a file-scope device array; host code must use its resolved device address
static __device__ __attribute__((aligned(256))) unsigned int mi_zero_page_storage[64];
static std::mutex g_zero_mutex;
static const void* g_zero_ptr[64] = {nullptr};
const void* zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(g_zero_mutex);
  if (g_zero_ptr[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(mi_zero_page_storage)) != hipSuccess) return nullptr;
    g_zero_ptr[dev] = p;
  }
  return g_zero_ptr[dev];
}

// ---- static scratch: the partial-sum slabs of mi_conv3x3_wgrad_static (callers that cannot allocate).
// A file-scope device array; every GPU the library runs on holds its own copy, at its own address,
// until the process ends.  In host code the array's NAME is only a stand-in: the address the kernels
// get is resolved once per device with hipGetSymbolAddress, with that device current.
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[64] = {nullptr};
static int static_scratch(void** out) {
  int dev = 0;
  int rc = check_hip(hipGetDevice(&dev), "static scratch: current device");
  if (rc) return rc;
  MI_REQUIRE(dev >= 0 && dev < 64, "static scratch: device index out of range");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    rc = check_hip(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "static scratch: hipGetSymbolAddress");
    if (rc) return rc;
    MI_REQUIRE(p != nullptr && ((uintptr_t)p & 255) == 0, "static scratch: unusable device address");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return MI_OK;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// The halo-tile geometry of a 3x3 layer, or false when the layer does not tile into 256-pixel blocks.
static bool hconv_args_for(const IgemmArgs& a, int dtype, HconvArgs* out) {
  if (!(a.mode == G_3x3 && a.epi != E_SHUFFLE_BIAS && a.M % 256 == 0 && a.Hi == a.Ho && a.Wi == a.Wo)) return false;
  const int ck = dtype == MI_BF16 ? 32 : 16;
  int R, G, bands;
  if (!(a.Cin % ck == 0 && hconv_tiling(a.Hi, a.Wi, &R, &G, &bands) && (G == 1 || a.B % G == 0))) return false;
  HconvArgs h{};
  h.in = a.in; h.in_cs = a.in_cs; h.in_co = a.in_co;
  h.w = a.w; h.Kp = a.Kp; h.Cp = a.Cin;
  h.N = a.N; h.bias = a.epi == E_PLAIN ? nullptr : a.bias; h.n_real = a.n_real; h.relu = a.epi == E_BIAS_RELU;
  h.out = a.out; h.out_cs = a.out_cs; h.out_co = a.out_co;
  h.mask = a.mask; h.mask_cs = a.mask_cs; h.mask_co = a.mask_co;
  h.pool = a.pool; h.pool_cs = a.pool_cs; h.pool_co = a.pool_co;
  h.W = a.Wi; h.R = R; h.G = G; h.pitch = (a.Wi + 2 + 7) / 8 * 8; h.HR = G * (R + 2) * h.pitch; h.bands = bands;
  h.ntiles = a.M / 256; h.nchunks = a.Cin / ck; h.zero = a.zero;
  h.in2 = a.in2; h.chunk2 = a.in2_c0 / ck; h.out2 = a.out2; h.n2 = a.out2_n0;
  h.head_w = a.head_w; h.head_b = a.head_b; h.head_out = a.head_out; h.head_cout = a.head_cout;
  h.head_hw = a.Hi * a.Wi; h.d_hw = make_fastdiv(a.Hi * a.Wi);
  *out = h;
  return true;
}
// Whether a bf16 3x3 layer (forward or dgrad arguments) runs on the DMA-fed halo-tile kernel: the one kernel
// that takes two-tensor operands (IgemmArgs::in2 / out2).  Mirrors the dispatch of launch_igemm.
bool conv3x3_takes_dconv(const IgemmArgs& a, int dtype) {
  // (a layer with the fused head never takes the few-pixel kernel: launch_igemm sends it here whatever its size)
  if (dtype != MI_BF16 || (a.head_out == nullptr && sconv_applies(a, dtype))) return false;
  HconvArgs h{};
  if (!hconv_args_for(a, dtype, &h)) return false;
  if (a.pool != nullptr && !(a.mask == nullptr && a.Hi % 2 == 0 && a.Wi % 2 == 0 && h.R % 2 == 0)) return false;
  return launch_dconv_bf16(h, nullptr, nullptr, true) == MI_OK;
}

int launch_igemm(const IgemmArgs& a_in, int dtype, hipStream_t s, const char** name) {
  IgemmArgs a = a_in;
  a.zero = zero_page();
  MI_REQUIRE(a.zero != nullptr, "igemm: cannot resolve the zero page on this device");
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_BF16, "igemm: bad dtype");
  MI_REQUIRE(a.in && a.w && a.out, "igemm: null pointer");
  MI_REQUIRE(a.M > 0 && a.M < (1 << 30), "igemm: pixel count out of range");
  MI_REQUIRE(a.N % 16 == 0 && a.Cin % 16 == 0 && a.Kp % 64 == 0 && a.Kp >= a.taps * a.Cin,
             "igemm: channel counts must be padded to 16 and K to 64");
  MI_REQUIRE(a.in_cs % 16 == 0 && a.in_co % 16 == 0 && a.out_cs % 16 == 0 && a.out_co % 16 == 0 &&
                 a.mask_cs % 16 == 0 && a.mask_co % 16 == 0,
             "igemm: view strides/offsets must be multiples of 16 elements");
  MI_REQUIRE(aligned16(a.in) && aligned16(a.w) && aligned16(a.out) && aligned16(a.mask),
             "igemm: pointers must be 16-byte aligned");
  MI_REQUIRE(a.epi == E_PLAIN || a.bias != nullptr, "igemm: bias required by this epilogue");
  MI_REQUIRE(a.epi != E_SHUFFLE_BIAS || (a.shufC % 16 == 0 && a.N == 4 * a.shufC), "igemm: bad shuffle width");
  MI_REQUIRE(a.in2 != nullptr ? (a.in_co + a.in2_c0 <= a.in_cs && a.Cin - a.in2_c0 <= a.in_cs && a.in2_c0 > 0 && a.in2_c0 < a.Cin)
                              : (a.in_overlap ? (a.in_co == 0 && a.in_cs >= 16) : a.in_co + a.Cin <= a.in_cs),
             "igemm: input channel range exceeds the view stride");
  MI_REQUIRE(a.out2 == nullptr || (a.out2_n0 > 0 && a.out2_n0 < a.N && a.out_co + a.out2_n0 <= a.out_cs && a.N - a.out2_n0 <= a.out_cs),
             "igemm: output channel range exceeds the view stride");
  const bool two_tensors = a.in2 != nullptr || a.out2 != nullptr || a.head_out != nullptr;   // operands only the DMA-fed kernel takes
  MI_REQUIRE(a.head_out == nullptr || (a.head_w && a.head_b && a.head_cout >= 1 && a.head_cout <= 2 && a.N == 32 && a.mask == nullptr &&
                                       a.epi == E_BIAS_RELU),
             "igemm: the fused 1x1 head needs a 32-channel ReLU layer and at most 2 head outputs");
  MI_REQUIRE(!two_tensors || (dtype == MI_BF16 && a.in2_c0 % 32 == 0 && a.out2_n0 % 16 == 0 && aligned16(a.in2) && aligned16(a.out2)),
             "igemm: two-tensor operands need bf16 and 32-channel-aligned halves");
  // 3x3 layers of so few pixels that a pixel-tiled kernel would leave most CUs idle: K-split kernel
  if (!two_tensors && sconv_applies(a, dtype)) return launch_sconv(a, dtype, s, name);
  // 3x3 stride-1 convolutions whose images tile evenly into 256-pixel blocks take the halo-tile
  // kernel (one LDS halo tile serves all 9 taps); everything else the generic gather kernel.
  {
    HconvArgs h{};
    if (hconv_args_for(a, dtype, &h)) {
      if (a.pool != nullptr) {
        MI_REQUIRE(a.mask == nullptr && a.Hi % 2 == 0 && a.Wi % 2 == 0 && h.R % 2 == 0 &&
                       a.pool_cs % 16 == 0 && a.pool_co % 16 == 0 && aligned16(a.pool),
                   "conv3x3_fwd: the pooled output cannot be fused for this shape");
      }
      if (dtype == MI_BF16) {      // the DMA-fed kernel where it applies, else the register-staged one
        const int rc = launch_dconv_bf16(h, s, name);
        if (rc != -1000) return rc;
      }
      MI_REQUIRE(!two_tensors, "igemm: two-tensor operands are only taken by the DMA-fed 3x3 kernel, which declines this shape");
      return dtype == MI_BF16 ? launch_hconv_bf16(h, s, name) : launch_hconv_f32(h, s, name);
    }
  }
  MI_REQUIRE(!two_tensors, "igemm: two-tensor operands are only taken by the DMA-fed 3x3 kernel, which declines this shape");
  MI_REQUIRE(a.pool == nullptr, "conv3x3_fwd: the pooled output cannot be fused for this shape");
  // other 3x3 stride-1 convolutions on images of at most 16 pixels take the position-major kernel
  // (skips the taps that fall on the padding, splits K inside the workgroup)
  if (a.mode == G_3x3 && a.epi != E_SHUFFLE_BIAS && a.Hi == a.Ho && a.Wi == a.Wo && a.Hi * a.Wi <= kPconvMaxHW &&
      a.N % 64 == 0 && a.Cin % (dtype == MI_BF16 ? 64 : 32) == 0 && a.Kp == 9 * a.Cin) {
    PconvArgs p{};
    p.in = a.in; p.in_cs = a.in_cs; p.in_co = a.in_co;
    p.w = a.w; p.Kp = a.Kp; p.Cp = a.Cin;
    p.N = a.N; p.bias = a.epi == E_PLAIN ? nullptr : a.bias; p.n_real = a.n_real; p.relu = a.epi == E_BIAS_RELU;
    p.out = a.out; p.out_cs = a.out_cs; p.out_co = a.out_co;
    p.mask = a.mask; p.mask_cs = a.mask_cs; p.mask_co = a.mask_co;
    p.B = a.B; p.H = a.Hi; p.W = a.Wi;
    p.img_tiles = (a.B + 63) / 64; p.n_tiles = a.N / 64;
    p.units = a.Hi * a.Wi * p.img_tiles * p.n_tiles;
    return dtype == MI_BF16 ? launch_pconv_bf16(p, s, name) : launch_pconv_f32(p, s, name);
  }
  // transposed convs with many pixels and a weight matrix that fits in LDS: streaming kernel
  if (dtype == MI_BF16 && tconv_eligible(a)) return launch_tconv_bf16(a, s, name);
  // the other transposed convs of few pixels and long reductions (both dtypes): fragments straight from memory, K split over waves
  if (rgemm_eligible(a, dtype) && (int64_t)a.M * a.N < (64 << 13)) return launch_rgemm(a, dtype, s, name);
  return dtype == MI_BF16 ? launch_igemm_bf16(a, s, name) : launch_igemm_f32(a, s, name);
}

bool conv3x3_fwd_fuses_pool(int B, int H, int W, int Cin, int split, int Cout, int dtype) {
  const int M = B * H * W, ck = dtype == MI_BF16 ? 32 : 16;
  // few pixels (the test of sconv_applies): the K-split kernel pools 32-pixel tiles of narrow images
  if (concat_cp(Cin, split) % (dtype == MI_BF16 ? 32 : 16) == 0 && (int64_t)M * pad_ch(Cout) < sconv_limit(dtype, H * W)) {
    if (sconv_pools(H, W, M)) return true;
    // fp32: a fused pool would send the layer to the pixel-tiled kernel, which runs such a layer on a handful of
    // workgroups (44 us against 5 + 4 for the few-pixel kernel and a pool launch); bf16 has the DMA-fed kernel for it
    if (dtype == MI_F32) return false;
  }
  int R, G, bands;
  if (M % 256 != 0 || concat_cp(Cin, split) % ck != 0 || !hconv_tiling(H, W, &R, &G, &bands) || !(G == 1 || B % G == 0))
    return false;
  return H % 2 == 0 && W % 2 == 0 && R % 2 == 0;
}

int wgrad_choose_splits(int M, int Na, int K, int dtype, int* pix_per_split) {
  const int PB = dtype == MI_BF16 ? 128 : 64;
  const int64_t tiles = (int64_t)((Na + 63) / 64) * ((K + 63) / 64);
  // enough workgroups to fill 256 CUs twice over, few enough partials for a cheap combine
  int64_t splits = (512 + tiles - 1) / tiles;
  if (splits > 256) splits = 256;
  const int64_t max_splits = (M + PB - 1) / PB;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int64_t pps = (M + splits - 1) / splits;
  pps = (pps + PB - 1) / PB * PB;
  splits = (M + pps - 1) / pps;
  *pix_per_split = (int)pps;
  return (int)splits;
}

int launch_wgrad(const WgradArgs& a_in, int dtype, hipStream_t s, const char** name) {
  WgradArgs a = a_in;
  a.zero = zero_page();
  MI_REQUIRE(a.zero != nullptr, "wgrad: cannot resolve the zero page on this device");
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_BF16, "wgrad: bad dtype");
  MI_REQUIRE(a.a && a.b && a.part, "wgrad: null pointer");
  MI_REQUIRE(a.M > 0 && a.M < (1 << 30), "wgrad: pixel count out of range");
  MI_REQUIRE(a.Na % 16 == 0 && a.Cb % 16 == 0, "wgrad: channel counts must be padded to 16");
  MI_REQUIRE(a.a_cs % 16 == 0 && a.a_co % 16 == 0 && a.b_cs % 16 == 0 && a.b_co % 16 == 0,
             "wgrad: view strides/offsets must be multiples of 16 elements");
  MI_REQUIRE(aligned16(a.a) && aligned16(a.b), "wgrad: pointers must be 16-byte aligned");
  MI_REQUIRE(a.splits >= 1 && a.pix_per_split % (dtype == MI_BF16 ? 128 : 64) == 0 &&
                 (int64_t)a.splits * a.pix_per_split >= a.M,
             "wgrad: bad pixel split");
  // fp32 3x3 layers on images that tile into 128-pixel blocks, split accordingly: one staged halo tile serves all 9 taps
  if (dtype == MI_F32 && a.mode == G_3x3 && a.colsum == nullptr && a.pix_per_split % 128 == 0 && a.Ho == a.Hb && a.Wo == a.Wb &&
      hwgrad_f32_applies(a.B, a.Hb, a.Wb, a.Na, a.Cb, a.a_cs, a.b_cs))
    return launch_hwgrad_f32(a, s, name);
  return dtype == MI_BF16 ? launch_wgrad_bf16(a, s, name) : launch_wgrad_f32(a, s, name);
}

// Shared argument builders (also used by the whole-network plan).
IgemmArgs conv3x3_fwd_args(const void* x, int xcs, int xco, const void* wf, const float* bias, void* y, int ycs,
                           int yco, int B, int H, int W, int Cin, int split, int Cout, int relu) {
  IgemmArgs a{};
  a.in = x; a.in_cs = xcs; a.in_co = xco;
  a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W;
  a.Cin = concat_cp(Cin, split);
  a.taps = 9; a.mode = G_3x3;
  a.w = wf; a.Kp = (int)kpad(9 * a.Cin);
  a.N = pad_ch(Cout);
  a.bias = bias; a.n_real = Cout;
  a.out = y; a.out_cs = ycs; a.out_co = yco;
  a.epi = relu ? E_BIAS_RELU : E_BIAS;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  return a;
}
IgemmArgs conv3x3_dgrad_args(const void* dy, int ycs, int yco, const void* wb, void* dx, int dcs, int dco,
                             const void* mask, int mcs, int mco, int B, int H, int W, int Cin, int split, int Cout) {
  IgemmArgs a{};
  a.in = dy; a.in_cs = ycs; a.in_co = yco;
  a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W;
  a.Cin = pad_ch(Cout);
  a.taps = 9; a.mode = G_3x3;
  a.w = wb; a.Kp = (int)kpad(9 * a.Cin);
  a.N = concat_cp(Cin, split);
  a.out = dx; a.out_cs = dcs; a.out_co = dco;
  a.mask = mask; a.mask_cs = mcs; a.mask_co = mco;
  a.epi = E_PLAIN;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  return a;
}
IgemmArgs convt_fwd_args(const void* x, int xcs, int xco, const void* wf, const float* bias, void* y, int ycs,
                         int yco, int B, int H, int W, int Cin, int Cout) {
  IgemmArgs a{};
  a.in = x; a.in_cs = xcs; a.in_co = xco;
  a.B = B; a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W;
  a.Cin = pad_ch(Cin);
  a.taps = 1; a.mode = G_1x1;
  a.w = wf; a.Kp = (int)kpad(a.Cin);
  a.shufC = pad_ch(Cout);
  a.N = 4 * a.shufC;
  a.bias = bias; a.n_real = Cout;
  a.out = y; a.out_cs = ycs; a.out_co = yco;
  a.epi = E_SHUFFLE_BIAS;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  return a;
}
IgemmArgs convt_dgrad_args(const void* dy, int ycs, int yco, const void* wb, void* dx, int dcs, int dco,
                           const void* mask, int mcs, int mco, int B, int H, int W, int Cin, int Cout) {
  IgemmArgs a{};
  a.in = dy; a.in_cs = ycs; a.in_co = yco;
  a.B = B; a.Hi = 2 * H; a.Wi = 2 * W; a.Ho = H; a.Wo = W;
  a.Cin = pad_ch(Cout);
  a.taps = 4; a.mode = G_2x2S2;
  a.w = wb; a.Kp = (int)kpad(4 * a.Cin);
  a.N = pad_ch(Cin);
  a.out = dx; a.out_cs = dcs; a.out_co = dco;
  a.mask = mask; a.mask_cs = mcs; a.mask_co = mco;
  a.epi = E_PLAIN;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  return a;
}

struct WgradLayout {
  int splits, pps;
  size_t part_bytes, sum_bytes;
  // halo-tile kernel (bf16, evenly tiling images): geometry and slab count
  bool halo;
  int R, G, bands, pitch, HR, tiles_per_split, slabs;
  bool posmajor;   // tiny images: reduce over the in-bounds pixels of each tap only
  int blocks;      // halo-tile kernel: workgroups per pixel split
};
// `co_blocks`: (n-block, c-block) workgroups per split of a second layer sharing the launch (0 = none)
WgradLayout conv3x3_wgrad_layout(int B, int H, int W, int Cin, int split, int Cout, int dtype, int co_blocks = 0) {
  WgradLayout l{};
  const int Np = pad_ch(Cout), Cp = concat_cp(Cin, split), K = 9 * Cp;
  const int M = B * H * W;
  if (dtype == MI_BF16 && M % 256 == 0 && Cp % 32 == 0 &&
      hwgrad_tiling(H, W, &l.R, &l.G, &l.bands, &l.pitch, &l.HR) && (l.G == 1 || B % l.G == 0)) {
    l.halo = true;
    const int ntiles = M / 256;
    const int nblk = Np % 64 == 0 ? Np / 64 : (Np + 31) / 32;
    const int blocks = nblk * (Cp / 32);                                       // (n-block, c-block) workgroups per split
    l.blocks = blocks;
    int splits = 256 / (blocks + co_blocks);                                   // one workgroup per CU, never a second round
    if (splits < 1) splits = 1;
    if (splits > ntiles) splits = ntiles;
    l.tiles_per_split = (ntiles + splits - 1) / splits;
    l.splits = (ntiles + l.tiles_per_split - 1) / l.tiles_per_split;
    l.slabs = l.splits * hwgrad_slabs_per_split(Np);
    l.part_bytes = (size_t)l.slabs * Np * K * sizeof(float);
    l.sum_bytes = (size_t)l.slabs * Np * sizeof(float);
    return l;
  }
  // fp32 on tiling images: the halo-tile kernel, with one round of workgroups (strides are checked again at the launch:
  // a view too large for its 32-bit offsets falls back to the generic kernel with the same, valid, split).  It also takes
  // the 4x4 images: all nine taps at 107 TFLOP/s beat the position-major walk over the 69 % of in-bounds products at 74-98.
  const bool halo32 = dtype == MI_F32 && hwgrad_f32_applies(B, H, W, Np, Cp, 16, 16);
  l.posmajor = H * W <= 16 && Cp % 64 == 0 && !halo32;
  l.splits = wgrad_choose_splits(M, Np, K, dtype, &l.pps);
  if (halo32) l.splits = hwgrad_f32_splits(B, H, W, Np, Cp, &l.pps);
  l.slabs = l.splits;
  l.part_bytes = (size_t)l.splits * Np * K * sizeof(float);
  l.sum_bytes = (size_t)l.splits * Np * sizeof(float);
  return l;
}
WgradLayout convt_wgrad_layout(int B, int H, int W, int Cin, int Cout, int dtype) {
  WgradLayout l{};
  const int Cp = pad_ch(Cin), K = 4 * pad_ch(Cout);
  l.splits = wgrad_choose_splits(B * H * W, Cp, K, dtype, &l.pps);
  l.part_bytes = (size_t)l.splits * Cp * K * sizeof(float);
  l.sum_bytes = (size_t)l.splits * K * sizeof(float);
  return l;
}
static int hwgrad_args(HwgradArgs& h, const WgradLayout& l, const void* x, int xcs, int xco, const void* dy, int ycs,
                       int yco, void* ws, int B, int H, int W, int Cin, int split, int Cout, const void* x2 = nullptr,
                       int x2_c0 = 0) {
  h.dy = dy; h.dy_cs = ycs; h.dy_co = yco; h.Na = pad_ch(Cout);
  h.x = x; h.x_cs = xcs; h.x_co = xco; h.Cb = concat_cp(Cin, split);
  h.x2 = x2; h.x2_c0 = x2_c0;
  MI_REQUIRE(x2 == nullptr || (x2_c0 % 32 == 0 && x2_c0 > 0 && x2_c0 < h.Cb && ((uintptr_t)x2 & 15) == 0),
             "conv3x3_wgrad: the second input tensor must start at a 32-channel boundary");
  h.W = W; h.R = l.R; h.G = l.G; h.pitch = l.pitch; h.HR = l.HR; h.bands = l.bands;
  h.ntiles = B * H * W / 256; h.tiles_per_split = l.tiles_per_split; h.splits = l.splits;
  h.part = (float*)ws;
  h.rowsum = (float*)((char*)ws + l.part_bytes);
  h.zero = zero_page();
  MI_REQUIRE(h.zero != nullptr, "conv3x3_wgrad: cannot resolve the zero page on this device");
  MI_REQUIRE(x && dy && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && xcs % 16 == 0 && xco % 16 == 0 &&
                 ycs % 16 == 0 && yco % 16 == 0, "conv3x3_wgrad: operands must be 16-byte aligned views");
  return MI_OK;
}
// Two 3x3 layers over the same pixels (the two convs of a block) in one weight-gradient launch;
// false when either layer does not take the halo-tile kernel or their channel-block widths differ.
bool conv3x3_wgrad_pairable(int B, int H, int W, int Cin0, int split0, int Cout0, int Cin1, int split1, int Cout1, int dtype) {
  const WgradLayout l0 = conv3x3_wgrad_layout(B, H, W, Cin0, split0, Cout0, dtype);
  const WgradLayout l1 = conv3x3_wgrad_layout(B, H, W, Cin1, split1, Cout1, dtype);
  return l0.halo && l1.halo && (pad_ch(Cout0) % 64 == 0) == (pad_ch(Cout1) % 64 == 0);
}
int run_conv3x3_wgrad_pair(const Conv3WgradCall& c0, const Conv3WgradCall& c1, float alpha, float beta, int B, int H, int W,
                           int dtype, hipStream_t s, const char** name, FinJobs* defer) {
  MI_REQUIRE(defer != nullptr, "conv3x3_wgrad_pair: needs a combine-job table");
  const WgradLayout s0 = conv3x3_wgrad_layout(B, H, W, c0.Cin, c0.split, c0.Cout, dtype);
  const WgradLayout s1 = conv3x3_wgrad_layout(B, H, W, c1.Cin, c1.split, c1.Cout, dtype);
  MI_REQUIRE(s0.halo && s1.halo, "conv3x3_wgrad_pair: layers must take the halo-tile kernel");
  const WgradLayout l0 = conv3x3_wgrad_layout(B, H, W, c0.Cin, c0.split, c0.Cout, dtype, s1.blocks);
  const WgradLayout l1 = conv3x3_wgrad_layout(B, H, W, c1.Cin, c1.split, c1.Cout, dtype, s0.blocks);
  if (c0.ws_bytes < l0.part_bytes + l0.sum_bytes || c1.ws_bytes < l1.part_bytes + l1.sum_bytes)
    return fail(MI_ESIZE, "conv3x3_wgrad_pair: workspace too small");
  HwgradArgs h[2]{};
  int rc = hwgrad_args(h[0], l0, c0.x, c0.xcs, c0.xco, c0.dy, c0.ycs, c0.yco, c0.ws, B, H, W, c0.Cin, c0.split, c0.Cout, c0.x2, c0.x2_c0);
  if (rc) return rc;
  rc = hwgrad_args(h[1], l1, c1.x, c1.xcs, c1.xco, c1.dy, c1.ycs, c1.yco, c1.ws, B, H, W, c1.Cin, c1.split, c1.Cout, c1.x2, c1.x2_c0);
  if (rc) return rc;
  rc = launch_hwgrad_bf16(h, 2, s, name);
  if (rc) return rc;
  rc = fin_jobs_add_conv3(*defer, h[0].part, h[0].rowsum, l0.slabs, h[0].Na, h[0].Cb, c0.Cout, c0.Cin, c0.split, c0.dW, c0.db, alpha, beta);
  if (rc) return rc;
  return fin_jobs_add_conv3(*defer, h[1].part, h[1].rowsum, l1.slabs, h[1].Na, h[1].Cb, c1.Cout, c1.Cin, c1.split, c1.dW, c1.db, alpha, beta);
}
int run_conv3x3_wgrad(const void* x, int xcs, int xco, const void* dy, int ycs, int yco, float* dW, float* db,
                      float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int split,
                      int Cout, int dtype, hipStream_t s, const char** name, FinJobs* defer) {
  const WgradLayout l = conv3x3_wgrad_layout(B, H, W, Cin, split, Cout, dtype);
  if (ws_bytes < l.part_bytes + l.sum_bytes) return fail(MI_ESIZE, "conv3x3_wgrad: workspace too small");
  if (l.halo) {
    HwgradArgs h{};
    int rc = hwgrad_args(h, l, x, xcs, xco, dy, ycs, yco, ws, B, H, W, Cin, split, Cout);
    if (rc) return rc;
    rc = launch_hwgrad_bf16(&h, 1, s, name);
    if (rc) return rc;
    FinJobs one{};
    FinJobs& J = defer ? *defer : one;
    rc = fin_jobs_add_conv3(J, h.part, h.rowsum, l.slabs, h.Na, h.Cb, Cout, Cin, split, dW, db, alpha, beta);
    return rc || defer ? rc : launch_finalize_multi(one, s);
  }
  WgradArgs a{};
  a.a = dy; a.a_cs = ycs; a.a_co = yco; a.Na = pad_ch(Cout);
  a.b = x; a.b_cs = xcs; a.b_co = xco; a.Cb = concat_cp(Cin, split);
  a.B = B; a.Hb = H; a.Wb = W; a.Ho = H; a.Wo = W;
  a.taps = 9; a.mode = l.posmajor ? G_3x3P : G_3x3;
  MI_REQUIRE(!l.posmajor || B * H * W < (1 << 27), "conv3x3_wgrad: pixel count out of range");
  a.part = (float*)ws;
  a.rowsum = (float*)((char*)ws + l.part_bytes);
  a.splits = l.splits; a.pix_per_split = l.pps;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  int rc = launch_wgrad(a, dtype, s, name);
  if (rc) return rc;
  FinJobs one{};
  FinJobs& J = defer ? *defer : one;
  rc = fin_jobs_add_conv3(J, a.part, a.rowsum, l.splits, a.Na, a.Cb, Cout, Cin, split, dW, db, alpha, beta);
  return rc || defer ? rc : launch_finalize_multi(one, s);
}
int run_convt_wgrad(const void* x, int xcs, int xco, const void* dy, int ycs, int yco, float* dW, float* db,
                    float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout,
                    int dtype, hipStream_t s, const char** name, FinJobs* defer) {
  const WgradLayout l = convt_wgrad_layout(B, H, W, Cin, Cout, dtype);
  if (ws_bytes < l.part_bytes + l.sum_bytes) return fail(MI_ESIZE, "convt2x2_wgrad: workspace too small");
  WgradArgs a{};
  a.a = x; a.a_cs = xcs; a.a_co = xco; a.Na = pad_ch(Cin);
  a.b = dy; a.b_cs = ycs; a.b_co = yco; a.Cb = pad_ch(Cout);
  a.B = B; a.Hb = 2 * H; a.Wb = 2 * W; a.Ho = H; a.Wo = W;
  a.taps = 4; a.mode = G_2x2S2;
  a.part = (float*)ws;
  a.colsum = (float*)((char*)ws + l.part_bytes);
  a.splits = l.splits; a.pix_per_split = l.pps;
  a.M = B * H * W;
  a.divHW = make_fastdiv(H * W); a.divW = make_fastdiv(W);
  int rc = launch_wgrad(a, dtype, s, name);
  if (rc) return rc;
  FinJobs one{};
  FinJobs& J = defer ? *defer : one;
  rc = fin_jobs_add_convt(J, a.part, a.colsum, l.splits, a.Na, a.Cb, Cin, Cout, dW, db, alpha, beta);
  return rc || defer ? rc : launch_finalize_multi(one, s);
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_version(void) { return 100; }
const char* mi_last_error(void) { return g_last_error.c_str(); }
int mi_pad_channels(int c) { return pad_ch(c); }

int mi_nchw_to_nhwc(const float* src, void* dst, int cs, int co, int B, int C, int H, int W, int dtype, mi_stream s) {
  MI_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
  MI_REQUIRE(cs % 16 == 0 && co % 16 == 0 && co + pad_ch(C) <= cs, "nchw_to_nhwc: bad view");
  return launch_nchw_to_nhwc(src, dst, cs, co, B, C, H, W, dtype, (hipStream_t)s);
}
int mi_nhwc_to_nchw(const void* src, int cs, int co, float* dst, int B, int C, int H, int W, int dtype, mi_stream s) {
  MI_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
  return launch_nhwc_to_nchw(src, cs, co, dst, B, C, H, W, dtype, (hipStream_t)s);
}

size_t mi_packed_conv3x3_bytes(int Cout, int Cin, int split, int dtype, int which) {
  const int Np = pad_ch(Cout), Cp = concat_cp(Cin, split);
  return (size_t)(which == 0 ? (int64_t)Np * kpad(9 * Cp) : (int64_t)Cp * kpad(9 * Np)) * elem_size(dtype);
}
size_t mi_packed_convt2x2_bytes(int Cin, int Cout, int dtype, int which) {
  const int Np = pad_ch(Cout), Cp = pad_ch(Cin);
  return (size_t)(which == 0 ? (int64_t)4 * Np * kpad(Cp) : (int64_t)Cp * kpad(4 * Np)) * elem_size(dtype);
}
int mi_pack_conv3x3(const float* w, void* wfwd, void* wbwd, int Cout, int Cin, int split, int dtype, mi_stream s) {
  MI_REQUIRE(w && Cout > 0 && Cin > 0 && split >= 0 && split < Cin, "pack_conv3x3: bad arguments");
  return launch_pack_conv3x3(w, wfwd, wbwd, Cout, Cin, split, dtype, (hipStream_t)s);
}
int mi_pack_convt2x2(const float* w, void* wfwd, void* wbwd, int Cin, int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(w && Cout > 0 && Cin > 0, "pack_convt2x2: bad arguments");
  return launch_pack_convt2x2(w, wfwd, wbwd, Cin, Cout, dtype, (hipStream_t)s);
}

int mi_conv3x3_relu_fwd(const void* x, int xcs, int xco, const void* wfwd, const float* bias, void* y, int ycs, int yco,
                        int B, int H, int W, int Cin, int split, int Cout, int relu, int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_fwd: bad shape");
  return launch_igemm(conv3x3_fwd_args(x, xcs, xco, wfwd, bias, y, ycs, yco, B, H, W, Cin, split, Cout, relu), dtype,
                      (hipStream_t)s);
}
int mi_conv3x3_dgrad(const void* dy, int ycs, int yco, const void* wbwd, void* dx, int dcs, int dco, const void* mask,
                     int mcs, int mco, int B, int H, int W, int Cin, int split, int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3x3_dgrad: bad shape");
  return launch_igemm(conv3x3_dgrad_args(dy, ycs, yco, wbwd, dx, dcs, dco, mask, mcs, mco, B, H, W, Cin, split, Cout),
                      dtype, (hipStream_t)s);
}
size_t mi_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int split, int Cout, int dtype) {
  const WgradLayout l = conv3x3_wgrad_layout(B, H, W, Cin, split, Cout, dtype);
  return l.part_bytes + l.sum_bytes;
}
int mi_conv3x3_wgrad(const void* x, int xcs, int xco, const void* dy, int ycs, int yco, float* dW, float* db,
                     float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int split,
                     int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && dW && ws, "conv3x3_wgrad: bad arguments");
  return run_conv3x3_wgrad(x, xcs, xco, dy, ycs, yco, dW, db, alpha, beta, ws, ws_bytes, B, H, W, Cin, split, Cout,
                           dtype, (hipStream_t)s, nullptr, nullptr);
}

size_t mi_conv3x3_wgrad_static_max_bytes(void) { return kStaticScratchBytes; }
int mi_conv3x3_wgrad_static(const void* x, int xcs, int xco, const void* dy, int ycs, int yco, float* dW, float* db,
                            float alpha, float beta, int B, int H, int W, int Cin, int split, int Cout, int dtype,
                            mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && dW, "conv3x3_wgrad_static: bad arguments");
  MI_REQUIRE(dtype == MI_F32 || dtype == MI_BF16, "conv3x3_wgrad_static: bad dtype");
  const size_t need = mi_conv3x3_wgrad_ws_bytes(B, H, W, Cin, split, Cout, dtype);
  if (need > kStaticScratchBytes) {
    char msg[200];
    snprintf(msg, sizeof msg, "conv3x3_wgrad_static: this shape needs %zu bytes of partial sums, the static scratch holds "
             "%zu; use mi_conv3x3_wgrad with a caller-provided workspace", need, kStaticScratchBytes);
    return fail(MI_ESIZE, msg);
  }
  void* ws = nullptr;
  int rc = static_scratch(&ws);
  if (rc) return rc;
  // every partial slab and row sum the combine reads is written by the weight-gradient kernel of
  // this same call, so nothing in the scratch has to start at zero
  return run_conv3x3_wgrad(x, xcs, xco, dy, ycs, yco, dW, db, alpha, beta, ws, kStaticScratchBytes, B, H, W, Cin, split,
                           Cout, dtype, (hipStream_t)s, nullptr, nullptr);
}

int mi_maxpool2_fwd(const void* x, int xcs, int xco, void* y, int ycs, int yco, int B, int H, int W, int C, int dtype,
                    mi_stream s) {
  MI_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "maxpool2_fwd: bad arguments");
  return launch_maxpool2_fwd(x, xcs, xco, y, ycs, yco, B, H, W, C, dtype, (hipStream_t)s);
}
int mi_maxpool2_bwd(const void* x, int xcs, int xco, const void* dyp, int pcs, int pco, const void* dskip, int scs,
                    int sco, void* dx, int dcs, int dco, int B, int H, int W, int C, int relu_mask, int dtype,
                    mi_stream s) {
  MI_REQUIRE(x && dx && B > 0 && H > 0 && W > 0 && C > 0, "maxpool2_bwd: bad arguments");
  MI_REQUIRE(dyp || (H / 2) * (W / 2) == 0, "maxpool2_bwd: pooled gradient missing");
  return launch_maxpool2_bwd(x, xcs, xco, dyp, pcs, pco, dskip, scs, sco, dx, dcs, dco, B, H, W, C, relu_mask, dtype,
                             (hipStream_t)s);
}

int mi_convt2x2_fwd(const void* x, int xcs, int xco, const void* wfwd, const float* bias, void* y, int ycs, int yco,
                    int B, int H, int W, int Cin, int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convt2x2_fwd: bad shape");
  return launch_igemm(convt_fwd_args(x, xcs, xco, wfwd, bias, y, ycs, yco, B, H, W, Cin, Cout), dtype, (hipStream_t)s);
}
int mi_convt2x2_dgrad(const void* dy, int ycs, int yco, const void* wbwd, void* dx, int dcs, int dco, const void* mask,
                      int mcs, int mco, int B, int H, int W, int Cin, int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convt2x2_dgrad: bad shape");
  return launch_igemm(convt_dgrad_args(dy, ycs, yco, wbwd, dx, dcs, dco, mask, mcs, mco, B, H, W, Cin, Cout), dtype,
                      (hipStream_t)s);
}
size_t mi_convt2x2_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dtype) {
  const WgradLayout l = convt_wgrad_layout(B, H, W, Cin, Cout, dtype);
  return l.part_bytes + l.sum_bytes;
}
int mi_convt2x2_wgrad(const void* x, int xcs, int xco, const void* dy, int ycs, int yco, float* dW, float* db,
                      float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W, int Cin, int Cout,
                      int dtype, mi_stream s) {
  MI_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && dW && ws, "convt2x2_wgrad: bad arguments");
  return run_convt_wgrad(x, xcs, xco, dy, ycs, yco, dW, db, alpha, beta, ws, ws_bytes, B, H, W, Cin, Cout, dtype,
                         (hipStream_t)s, nullptr, nullptr);
}

int mi_conv1x1_head_fwd(const void* x, int xcs, int xco, const float* w, const float* bias, float* pred, int B, int H,
                        int W, int Cin, int Cout, int dtype, mi_stream s) {
  MI_REQUIRE(x && w && bias && pred && B > 0 && H > 0 && W > 0, "head_fwd: bad arguments");
  return launch_head_fwd(x, xcs, xco, w, bias, pred, B, H, W, Cin, Cout, dtype, (hipStream_t)s);
}
size_t mi_conv1x1_head_ws_bytes(int Cin, int Cout) { return head_ws_bytes(Cin, Cout); }
int mi_conv1x1_head_bwd(const void* x, int xcs, int xco, const float* w, const float* dpred, void* dx, int dcs, int dco,
                        float* dW, float* db, float alpha, float beta, void* ws, size_t ws_bytes, int B, int H, int W,
                        int Cin, int Cout, int relu_mask, int dtype, mi_stream s) {
  MI_REQUIRE(x && w && dpred && dx && dW && db && ws && B > 0 && H > 0 && W > 0, "head_bwd: bad arguments");
  return launch_head_bwd(x, xcs, xco, w, dpred, dx, dcs, dco, dW, db, alpha, beta, ws, ws_bytes, B, H, W, Cin, Cout,
                         relu_mask, dtype, (hipStream_t)s);
}

size_t mi_mse_ws_bytes(int64_t n) { return mse_ws_bytes(n); }
int mi_mse_fwd_bwd(const float* pred, const float* target, int64_t n, float* out3, float* dpred, float gscale,
                   const float* gscale_dev, void* ws, size_t ws_bytes, mi_stream s) {
  MI_REQUIRE(pred && target && ws && (out3 || dpred), "mse: null pointer");
  return launch_mse(pred, target, n, out3, dpred, gscale, gscale_dev, ws, ws_bytes, (hipStream_t)s);
}

int mi_bilinear_fwd(const void* src, int64_t ssb, int64_t ssc, int64_t ssy, int64_t ssx, int sdt, void* dst,
                    int64_t dsb, int64_t dsc, int64_t dsy, int64_t dsx, int ddt, int B, int C, int Hin, int Win,
                    int Hout, int Wout, mi_stream s) {
  MI_REQUIRE(src && dst && B > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "bilinear_fwd: bad arguments");
  return launch_bilinear_fwd(src, ssb, ssc, ssy, ssx, sdt, dst, dsb, dsc, dsy, dsx, ddt, B, C, Hin, Win, Hout, Wout,
                             (hipStream_t)s);
}
int mi_bilinear_bwd(const void* dout, int64_t osb, int64_t osc, int64_t osy, int64_t osx, int odt, void* din,
                    int64_t isb, int64_t isc, int64_t isy, int64_t isx, int idt, int B, int C, int Hin, int Win,
                    int Hout, int Wout, mi_stream s) {
  MI_REQUIRE(dout && din && B > 0 && C > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "bilinear_bwd: bad arguments");
  return launch_bilinear_bwd(dout, osb, osc, osy, osx, odt, din, isb, isc, isy, isx, idt, B, C, Hin, Win, Hout, Wout,
                             (hipStream_t)s);
}

int mi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 int step, float grad_scale, mi_stream s) {
  MI_REQUIRE(p && g && m && v, "adam: null pointer");
  return launch_adam(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)s);
}

}  // extern "C"
