/*
 * unet_tm_hip.h -- C ABI of libunet_tm_hip.so: the U-Net traffic-matrix hot path
 * (forward, backward, loss, optimiser step) as hand-written gfx950 HIP kernels.
 *
 * Every entry point takes raw device pointers, plain ints and a hipStream_t (passed
 * as void*).  The caller owns all memory: outputs and workspaces are caller-allocated,
 * nothing is retained after a call returns, nothing is freed by the library.  Work is
 * enqueued on the given stream and no entry point waits for the GPU, with ONE exception:
 * mi_unet_plan_prepare() (or, if it was never called, the first training-mode mi_unet_forward
 * with MI_FWD_REPACK, mi_unet_backward or mi_unet_plan_side_stream of a process on a given device
 * and caller stream) picks the second stream -- the one the weight-gradient kernels and the packing
 * of the backward weight images run on -- by timing two trivial kernels, which
 * waits for the caller's stream once.  That choice -- one stream per (device, caller
 * stream), kept for the life of the process -- is the library's only process-wide state.
 * The library reads no environment variable: every switch is a plan option
 * (mi_unet_plan_set_option) that the host side sets.
 * The return value is 0 on success, a negative MI_E* code for an argument error, or a
 * positive hipError_t.  mi_last_error() returns a thread-local description of the last
 * failure.
 *
 * "Reference" below is VoVanVi/Unet-based-Network-Traffic-Prediction; each entry point
 * names the reference call site whose arithmetic it replaces.
 *
 * Activation layout.  Inside the library activations are NHWC ("pixel-major"): element
 * (b, y, x, c) of a view lives at  base[((b*H + y)*W + x) * cstride + coff + c].
 * cstride/coff let several tensors share one buffer (the skip and the upsampled half of
 * a decoder input are two channel ranges of one allocation, which is how the concat of
 * models/unet2d.py:86 costs no copy).  Channel counts are padded (mi_pad_channels: 32 up to 32
 * channels, multiples of 16 above); padding channels hold zeros.  dtype selects the storage/compute type of activations
 * and packed weights: MI_F32 (fp32 storage, exact-fp32 MFMA) or MI_BF16 (bf16 storage,
 * bf16 MFMA, fp32 accumulate).  Parameters, parameter gradients, the loss and the
 * network input/output are always fp32 in the reference's NCHW / (Cout,Cin,kh,kw) layout.
 */
#ifndef UNET_TM_HIP_H
#define UNET_TM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_F32 0
#define MI_BF16 1

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define MI_ESIZE (-2)    /* buffer / workspace too small */
#define MI_ESTATE (-3)   /* call out of order (e.g. backward before forward) */

typedef void* mi_stream;               /* hipStream_t */
typedef struct mi_unet_plan mi_unet_plan;

int mi_version(void);
const char* mi_last_error(void);
/* Stored channel count of a C-channel NHWC tensor: 32 for C <= 32, else C rounded up to 16. */
int mi_pad_channels(int c);

/* ------------------------------------------------------------------ layout helpers */
/* fp32 NCHW (B,C,H,W) -> NHWC view of `dtype` with C padded (pad channels zeroed).
 * Replaces nothing in the reference (its tensors stay NCHW, dataset.py:221-223);
 * it is the entry of the network input into the library's layout. */
int mi_nchw_to_nhwc(const float* src, void* dst, int dst_cstride, int dst_coff,
                    int B, int C, int H, int W, int dtype, mi_stream s);
/* NHWC view -> fp32 NCHW (first C channels). */
int mi_nhwc_to_nchw(const void* src, int src_cstride, int src_coff, float* dst,
                    int B, int C, int H, int W, int dtype, mi_stream s);

/* ------------------------------------------------------------------ weight packing */
/* nn.Conv2d weight (Cout,Cin,3,3) fp32 -> K-major packed operands of `dtype`:
 *   wfwd [Npad][Kp]  Npad=pad(Cout), k = tap*Cpad_in + pos(ci)           (forward)
 *   wbwd [Cpad_in][Kp2] k = (8-tap)*Npad + co   (180-degree rotated, for dgrad)
 * pos(ci) = ci for ci < split, pad(split) + (ci-split) otherwise: `split` is the
 * channel count of the first concat operand (0 = no concat).  Kp/Kp2 = K rounded up
 * to a multiple of 64 elements.  Either output may be NULL. */
int mi_pack_conv3x3(const float* w, void* wfwd, void* wbwd, int Cout, int Cin, int split,
                    int dtype, mi_stream s);
/* nn.ConvTranspose2d weight (Cin,Cout,2,2) fp32 ->
 *   wfwd [4*Npad][Kp]   row = (a*2+b)*Npad + co, k = ci                   (forward)
 *   wbwd [Cpad_in][Kp2] k = (a*2+b)*Npad + co                             (dgrad)  */
int mi_pack_convt2x2(const float* w, void* wfwd, void* wbwd, int Cin, int Cout,
                     int dtype, mi_stream s);
size_t mi_packed_conv3x3_bytes(int Cout, int Cin, int split, int dtype, int which /*0 fwd,1 bwd*/);
size_t mi_packed_convt2x2_bytes(int Cin, int Cout, int dtype, int which);

/* ------------------------------------------------------------------ A1: 3x3 conv + ReLU */
/* y = relu(conv3x3(x, W, pad 1) + b): nn.Conv2d(k=3,padding=1)+nn.ReLU,
 * models/unet2d.py:28-31, :38-41, :51-54.  Implicit GEMM on MFMA.  relu=0 omits the
 * activation.  Cin/Cout are the logical channel counts; x carries pad(Cin) (+ concat
 * padding when split>0), y receives pad(Cout) channels. */
int mi_conv3x3_relu_fwd(const void* x, int x_cstride, int x_coff, const void* wfwd, const float* bias,
                        void* y, int y_cstride, int y_coff,
                        int B, int H, int W, int Cin, int split, int Cout, int relu, int dtype, mi_stream s);
/* dx = conv3x3(dy, rot180(W)) [* (mask > 0)]: the autograd backward of the same
 * layers w.r.t. their input.  dy must already be the gradient w.r.t. the conv's
 * pre-activation output (ReLU mask applied by its producer).  If mask != NULL the
 * result is multiplied by (mask > 0) element-wise: that is the ReLU backward of the
 * PRECEDING layer (nn.ReLU(inplace=True) saves its output), fused here. */
int mi_conv3x3_dgrad(const void* dy, int dy_cstride, int dy_coff, const void* wbwd,
                     void* dx, int dx_cstride, int dx_coff,
                     const void* mask, int mask_cstride, int mask_coff,
                     int B, int H, int W, int Cin, int split, int Cout, int dtype, mi_stream s);
/* dW (Cout,Cin,3,3) and db (Cout) in fp32:  dW = alpha * sum_pixels(...) + beta * dW.
 * Deterministic: pixel-range partials in `ws`, summed in a fixed order.
 * ws must hold mi_conv3x3_wgrad_ws_bytes() bytes. */
int mi_conv3x3_wgrad(const void* x, int x_cstride, int x_coff,
                     const void* dy, int dy_cstride, int dy_coff,
                     float* dW, float* db, float alpha, float beta, void* ws, size_t ws_bytes,
                     int B, int H, int W, int Cin, int split, int Cout, int dtype, mi_stream s);
size_t mi_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int split, int Cout, int dtype);
/* The same weight gradient for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Identical to mi_conv3x3_wgrad -- same kernels, same fixed-order combine, bit-identical dW/db --
 * except that the partial sums live in a scratch array built into the library (one copy per GPU,
 * held until the process ends) instead of a caller workspace, and the call performs no allocation
 * of any kind.  Shapes whose mi_conv3x3_wgrad_ws_bytes() exceeds
 * mi_conv3x3_wgrad_static_max_bytes() (32 MiB) are rejected with MI_ESIZE.  All calls on one GPU
 * share that GPU's scratch: calls that may overlap on one GPU (different streams, different
 * threads) race; order them on one stream or with events. */
int mi_conv3x3_wgrad_static(const void* x, int x_cstride, int x_coff,
                            const void* dy, int dy_cstride, int dy_coff,
                            float* dW, float* db, float alpha, float beta,
                            int B, int H, int W, int Cin, int split, int Cout, int dtype, mi_stream s);
size_t mi_conv3x3_wgrad_static_max_bytes(void);

/* ------------------------------------------------------------------ A2: 2x2 max-pool */
/* nn.MaxPool2d(2,2), models/unet2d.py:22/:68.  Output is (B, H/2, W/2, C) (floor). */
int mi_maxpool2_fwd(const void* x, int x_cstride, int x_coff, void* y, int y_cstride, int y_coff,
                    int B, int H, int W, int C, int dtype, mi_stream s);
/* Backward of the pool fused with what surrounds it in the U-Net:
 *   dx = (route(dy_pool) + dskip) * (x > 0)
 * route() sends each pooled gradient to the FIRST maximum of its window in row-major
 * order (PyTorch's tie rule).  dskip (may be NULL) is the gradient arriving at the same
 * tensor through the skip connection (models/unet2d.py:67/:86); relu_mask!=0 applies
 * the ReLU backward of the conv that produced x.  dx may alias dskip. */
int mi_maxpool2_bwd(const void* x, int x_cstride, int x_coff, const void* dy_pool, int dyp_cstride, int dyp_coff,
                    const void* dskip, int ds_cstride, int ds_coff, void* dx, int dx_cstride, int dx_coff,
                    int B, int H, int W, int C, int relu_mask, int dtype, mi_stream s);

/* ------------------------------------------------------------------ A3: 2x2 stride-2 transposed conv */
/* nn.ConvTranspose2d(Cin,Cout,2,2), models/unet2d.py:46-48/:77.  x is (B,H,W,Cin),
 * y is (B,2H,2W,Cout) written through the (cstride,coff) view, i.e. straight into the
 * second half of the decoder's concat buffer. */
int mi_convt2x2_fwd(const void* x, int x_cstride, int x_coff, const void* wfwd, const float* bias,
                    void* y, int y_cstride, int y_coff, int B, int H, int W, int Cin, int Cout,
                    int dtype, mi_stream s);
int mi_convt2x2_dgrad(const void* dy, int dy_cstride, int dy_coff, const void* wbwd,
                      void* dx, int dx_cstride, int dx_coff,
                      const void* mask, int mask_cstride, int mask_coff,
                      int B, int H, int W, int Cin, int Cout, int dtype, mi_stream s);
int mi_convt2x2_wgrad(const void* x, int x_cstride, int x_coff, const void* dy, int dy_cstride, int dy_coff,
                      float* dW, float* db, float alpha, float beta, void* ws, size_t ws_bytes,
                      int B, int H, int W, int Cin, int Cout, int dtype, mi_stream s);
size_t mi_convt2x2_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dtype);

/* ------------------------------------------------------------------ A5: 1x1 head */
/* nn.Conv2d(features[0], out_channels, 1), models/unet2d.py:59/:89.  x NHWC view,
 * pred fp32 NCHW (B,Cout,H,W).  w is the fp32 parameter (Cout,Cin[,1,1]) itself. */
int mi_conv1x1_head_fwd(const void* x, int x_cstride, int x_coff, const float* w, const float* bias,
                        float* pred, int B, int H, int W, int Cin, int Cout, int dtype, mi_stream s);
/* dx = (W^T dpred) * (x > 0) (relu_mask), dW = alpha*sum + beta*dW, db likewise. */
int mi_conv1x1_head_bwd(const void* x, int x_cstride, int x_coff, const float* w, const float* dpred,
                        void* dx, int dx_cstride, int dx_coff, float* dW, float* db,
                        float alpha, float beta, void* ws, size_t ws_bytes,
                        int B, int H, int W, int Cin, int Cout, int relu_mask, int dtype, mi_stream s);
size_t mi_conv1x1_head_ws_bytes(int Cin, int Cout);

/* ------------------------------------------------------------------ A6: MSE loss */
/* nn.MSELoss() (train.py:498/:387) and the sums of test.py:39-43 in one pass.
 * out3 (device, 3 floats) = { mean((pred-y)^2), sum|pred-y|, sum (pred-y)^2 }.
 * If dpred != NULL also writes dpred = gscale * 2*(pred-y)/n  (gscale_dev, if not NULL,
 * is a device scalar multiplied in: the upstream gradient of the loss).
 * out3 may be NULL when dpred is not (the backward pass of the loss: gradient only, one launch).
 * ws: mi_mse_ws_bytes(n). Deterministic two-pass reduction. */
int mi_mse_fwd_bwd(const float* pred, const float* target, int64_t n, float* out3,
                   float* dpred, float gscale, const float* gscale_dev, void* ws, size_t ws_bytes, mi_stream s);
size_t mi_mse_ws_bytes(int64_t n);

/* ------------------------------------------------------------------ A8: bilinear resize */
/* F.interpolate(mode="bilinear", align_corners=False), models/unet2d.py:81-84, :92-100.
 * Generic strided 4-D views: element (b,c,y,x) at base[b*sb + c*sc + y*sy + x*sx].
 * src_dtype/dst_dtype are MI_F32/MI_BF16.  The backward is a gather (deterministic). */
int mi_bilinear_fwd(const void* src, int64_t ssb, int64_t ssc, int64_t ssy, int64_t ssx, int src_dtype,
                    void* dst, int64_t dsb, int64_t dsc, int64_t dsy, int64_t dsx, int dst_dtype,
                    int B, int C, int Hin, int Win, int Hout, int Wout, mi_stream s);
int mi_bilinear_bwd(const void* dout, int64_t osb, int64_t osc, int64_t osy, int64_t osx, int dout_dtype,
                    void* din, int64_t isb, int64_t isc, int64_t isy, int64_t isx, int din_dtype,
                    int B, int C, int Hin, int Win, int Hout, int Wout, mi_stream s);

/* ------------------------------------------------------------------ F1: Adam */
/* torch.optim.Adam(lr) step (train.py:497) over one flat fp32 range: default betas
 * handling, no weight decay / amsgrad.  step is the 1-based step count.
 * grad_scale multiplies the gradient first (1/world_size in data-parallel runs). */
int mi_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, mi_stream s);

/* ------------------------------------------------------------------ F2: input pipeline */
/* The raw series (float64, as pandas reads it: dataset.py:316-323) stays in device memory and
 * batches are gathered from it; nothing image-shaped crosses PCIe.
 * mi_series_minmax: out2 (device, 2 doubles) = { nanmin, nanmax } of the series, the statistics
 *   MinMaxScaler.fit keeps (dataset.py:326-327).  ws: mi_series_minmax_ws_bytes().
 * mi_window_count: number N of (input, target) window pairs, or -1 with the reference's
 *   ValueError text in mi_last_error() (utils/image_transform.py:27-44).
 * mi_window_batch: for b < B, with t = targets[b] the ORIGINAL window index of the sample's target
 *   (K-1 <= t < N, device int64):
 *     x[b][k][i][j] = norm(series[(t-K+1+k)*stride + i*W_in + j])                  k < K
 *     y[b][i][j]    = norm(series[t*stride + H_in*W_in - overlap_len + i*W_out + j])
 *   norm(v) = float32(v*scale + offset), two rounded float64 operations, i.e. exactly
 *   MinMaxScaler.transform followed by the float32 cast of utils/image_transform.py:63-64.
 *   x is (B,K,H_in,W_in): the K context windows of dataset.py:176-183 as channels; K = 1 gives the
 *   single windows the reference feeds its U-Net.  y is (B,1,H_out,W_out) (dataset.py:221-223).
 *   Positions outside the series (an invalid target index) read as 0. */
size_t mi_series_minmax_ws_bytes(void);
int mi_series_minmax(const double* series, int64_t n, double* out2, void* ws, size_t ws_bytes, mi_stream s);
int64_t mi_window_count(int64_t n, int H_in, int W_in, int H_out, int W_out, int overlap_len, int stride);
int mi_window_batch(const double* series, int64_t n, const int64_t* targets, int B, int K,
                    int H_in, int W_in, int H_out, int W_out, int overlap_len, int stride,
                    double scale, double offset, float* x, float* y, mi_stream s);

/* ------------------------------------------------------------------ F4: series reconstruction */
/* Overlap-average of predicted horizons into one series (visualize_results.py:443-476).  pred / gt are
 * (B, out_len) float rows; slot[g] (device int32, g < n_slots) is the row of window g or -1.  Window
 * g's horizon lies at raw positions [g*stride + in_len, + out_len), as the reference places it.  For
 * every position < total (>= (n_slots-1)*stride + in_len + out_len): count = number of horizons
 * covering it, pred_mean / gt_mean = their float64 average (0 where count is 0). */
int mi_series_reconstruct(const float* pred, const float* gt, const int32_t* slot, int64_t n_slots, int B,
                          int in_len, int out_len, int stride, int64_t total, double* pred_mean,
                          double* gt_mean, int32_t* count, mi_stream s);

/* ------------------------------------------------------------------ A7: whole-network plan */
/* The orchestration of UNet.forward (models/unet2d.py:61-102) and of its autograd
 * backward as one launch sequence per call.  A plan is immutable shape metadata. */
typedef struct mi_unet_desc {
  int in_channels, out_channels;
  int n_levels;            /* len(features), 1..8 */
  int features[8];         /* must double per level, as the reference requires */
  int batch, height, width;
  int out_h, out_w;        /* forced output size (models/unet2d.py:92-100) or 0,0 */
  int dtype;               /* MI_F32 / MI_BF16 */
  int need_input_grad;     /* also compute d(loss)/d(x) */
} mi_unet_desc;

int mi_unet_plan_create(const mi_unet_desc* desc, mi_unet_plan** out);
void mi_unet_plan_destroy(mi_unet_plan* plan);
size_t mi_unet_plan_workspace_bytes(const mi_unet_plan* plan);
/* Flat fp32 parameter buffer: parameters in construction order (downs, bottleneck,
 * ups, final_conv), each starting at a multiple of 64 elements. */
int64_t mi_unet_plan_flat_elems(const mi_unet_plan* plan);
int mi_unet_plan_num_params(const mi_unet_plan* plan);
int mi_unet_plan_param_info(const mi_unet_plan* plan, int i, char* name, int name_cap,
                            int64_t* flat_offset, int64_t shape[4], int* ndim);
/* Named intermediate tensors inside the workspace (activations "act.*", gradients
 * "grad.*"), for per-layer parity checks.  layout: 0 = NHWC of plan dtype, 1 = fp32 NCHW.
 * The channel stride of a view is whatever the plan chose: e.g. "act.x" of a network with up to 16 input
 * channels has stride 16 (its pixels are stored 16 channels wide), other activations of up to 32 channels 32. */
int mi_unet_plan_num_tensors(const mi_unet_plan* plan);
int mi_unet_plan_tensor_info(const mi_unet_plan* plan, int i, char* name, int name_cap,
                             int64_t* ws_offset_bytes, int dims[4] /*B,H,W,C*/, int* cstride,
                             int* coff, int* layout);
/* Backward runs in segments so a caller can start reducing a gradient range while the
 * rest of backward is still running.  Segment s covers flat range [lo,hi); segments
 * are numbered in execution order (head first) and their ranges descend. */
int mi_unet_plan_num_segments(const mi_unet_plan* plan);
int mi_unet_plan_segment_range(const mi_unet_plan* plan, int seg, int64_t* lo, int64_t* hi);

/* The workspace keeps state between calls (packed weights, and counters that the kernels expect
 * at zero and leave at zero): the FIRST forward on a workspace, and the first after its contents
 * were overwritten, must carry MI_FWD_REPACK, which packs the weights and initialises that state.
 * The plan remembers which workspace addresses it has initialised: a forward without MI_FWD_REPACK,
 * or a backward, on an address it has not seen returns MI_ESTATE. */
#define MI_FWD_REPACK 1      /* parameters changed since the last forward: repack weights */
#define MI_FWD_TRAIN 2       /* keep what backward needs (also packs dgrad weights) */
int mi_unet_forward(mi_unet_plan* plan, const float* x_nchw, const float* params_flat, void* ws,
                    size_t ws_bytes, float* pred_nchw, int flags, mi_stream s);
/* grads_flat = alpha * dL/dparams + beta * grads_flat over segments [seg_lo, seg_hi). */
int mi_unet_backward(mi_unet_plan* plan, const float* dpred_nchw, const float* params_flat, void* ws,
                     size_t ws_bytes, float* grads_flat, float alpha, float beta, int seg_lo, int seg_hi,
                     float* dx_nchw, mi_stream s);

/* Plan options (no counterpart in the reference).  All default to 0; set them right after
 * mi_unet_plan_create.  The Python module maps the environment variables UNET_TM_GRAPH,
 * UNET_TM_SERIAL_BACKWARD, UNET_TM_DEBUG_STREAMS and UNET_TM_PACK_ONE_STREAM onto them.
 *   MI_OPT_GRAPH            replay the plan-owned part of forward / backward from captured HIP graphs
 *                           (pays only where the host cannot keep ahead of the GPU)
 *   MI_OPT_SERIAL_BACKWARD  keep every kernel on the caller's stream (no second stream at all);
 *                           must be set before the plan's first training call
 *   MI_OPT_DEBUG_STREAMS    print the outcome of the second-stream choice on stderr
 *   MI_OPT_PACK_ONE_STREAM  pack the dgrad weight images on the caller's stream instead of beside
 *                           the forward pass
 *   MI_OPT_NO_KSPLIT        never split the K range of a few-pixel 3x3 layer across workgroups (the
 *                           split exchanges partial tiles through workspace scratch; see below) */
#define MI_OPT_GRAPH 1
#define MI_OPT_SERIAL_BACKWARD 2
#define MI_OPT_DEBUG_STREAMS 3
#define MI_OPT_PACK_ONE_STREAM 4
#define MI_OPT_NO_KSPLIT 5
int mi_unet_plan_set_option(mi_unet_plan* plan, int option, int value);
/* Where the scratch of the cross-workgroup K split lies in the workspace (total_bytes = 0: this plan
 * has none): counter_bytes of arrival counters, then partial tiles.  The kernels expect the counters
 * at zero and leave them at zero; a check can read them back between launches. */
int mi_unet_plan_ksplit_region(const mi_unet_plan* plan, size_t* ws_offset_bytes, size_t* counter_bytes,
                               size_t* total_bytes);

/* Bucketed callers (no counterpart in the reference, which is single-device): mi_unet_backward runs
 * the weight-gradient kernels on a second stream of the plan and, by default, makes the caller's
 * stream wait for them before it returns.  With defer_join on, a call over a PARTIAL segment range
 * instead leaves that stream ahead: it has waited for everything the call put on the caller's
 * stream, so a collective enqueued behind mi_unet_plan_side_stream() sees the finished gradient
 * range while the next call's dgrad chain starts without a stall.  The call whose range ends at the
 * last segment joins the two streams as usual.  mi_unet_plan_side_stream returns NULL when the plan
 * runs on one stream (profiling on, MI_OPT_SERIAL_BACKWARD, or no stream found that runs
 * concurrently with caller_stream): then there is nothing to defer.  The stream is chosen on first
 * use, which synchronises caller_stream once; mi_unet_plan_prepare makes that first use explicit
 * (call it once after creating a plan, outside any latency-critical or captured region).
 * mi_unet_plan_last_call_deferred: 1 when the last mi_unet_backward call on the plan returned with the
 * side stream ordered BEHIND everything it enqueued (reduce the finished range behind the side stream),
 * 0 when it joined the streams (the range is complete in caller_stream's order: reduce there). */
int mi_unet_plan_prepare(mi_unet_plan* plan, mi_stream caller_stream);
/* optimizer.step() (train.py:392, torch.optim.Adam of train.py:497) over ALL parameters of the plan's flat
 * layout, fused with the packing of the MFMA weight images the next forward and backward read: every
 * parameter is read once, updated (the arithmetic of mi_adam_step, bit for bit), and the 3x3 / transposed
 * conv weights are written into both images in `ws` from the updated values -- one launch instead of an
 * Adam pass plus two packing passes over the masters.  The forward calls that follow on this workspace need
 * no MI_FWD_REPACK until the parameters change by other means.  `ws` must have seen a forward with
 * MI_FWD_REPACK before (MI_ESTATE otherwise).  mi_unet_plan_adam_pack_supported: 1 when every layer of the
 * plan can go through the fused kernel (16-byte aligned weight runs: input channels a multiple of 4, no
 * padding between the halves of a concatenated input; at most 32 layers), else 0: use mi_adam_step and let
 * the next forward repack. */
int mi_unet_plan_adam_pack_supported(const mi_unet_plan* plan);
int mi_unet_adam_pack(mi_unet_plan* plan, float* params_flat, const float* grads_flat, float* exp_avg,
                      float* exp_avg_sq, void* ws, size_t ws_bytes, float lr, float beta1, float beta2, float eps,
                      int step, float grad_scale, mi_stream stream);
int mi_unet_plan_defer_join(mi_unet_plan* plan, int on);
void* mi_unet_plan_side_stream(mi_unet_plan* plan, mi_stream caller_stream);
int mi_unet_plan_last_call_deferred(const mi_unet_plan* plan);

/* Per-launch timing: when enabled, forward/backward bracket every kernel launch with
 * HIP events on the launch stream.  mi_unet_profile_read synchronises those events and
 * returns, for launch i, the kernel name, a layer label, algorithmic FLOPs and bytes,
 * and the elapsed milliseconds. */
int mi_unet_profile_enable(mi_unet_plan* plan, int on);
int mi_unet_profile_count(const mi_unet_plan* plan);
int mi_unet_profile_read(mi_unet_plan* plan, int i, char* kernel, int kcap, char* label, int lcap,
                         double* flops, double* bytes, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* UNET_TM_HIP_H */
