// Whole-network launch sequences: the orchestration of UNet.forward
// (models/unet2d.py:61-102 of the reference) and of its autograd backward, as one host
// call each.  A plan is immutable shape metadata (workspace layout, flat parameter
// layout, backward segments); all device memory is the caller's.
//
// Workspace tensors and what they correspond to in the reference forward:
//   x            network input in NHWC                       (forward argument, :61)
//   a1.l         post-ReLU output of downs[l][0..1]           (:66)
//   cat.l        decoder input of level l: channels [0,f) = post-ReLU output of
//                downs[l][2..3] = the skip (:67), [f,2f) = upsampled tensor (:77-84);
//                torch.cat (:86) therefore never copies
//   pool.l       self.pool output (:68)
//   b1, b2       bottleneck activations (:70)
//   upraw.j      ConvTranspose2d output when it still has to be resized (:81-84)
//   u1.j, u2.j   post-ReLU outputs of ups[2j+1] (:87)
//   pred0        final_conv output when a forced out_size resize follows (:92-100)
// Gradient tensors g.* mirror them; gradients of post-ReLU tensors are stored already
// multiplied by the ReLU mask (i.e. they are gradients w.r.t. the conv's pre-activation).
#include "common.h"

#include <cstring>
#include <cstdio>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace mi {

IgemmArgs conv3x3_fwd_args(const void*, int, int, const void*, const float*, void*, int, int, int, int, int, int, int, int, int);
IgemmArgs conv3x3_dgrad_args(const void*, int, int, const void*, void*, int, int, const void*, int, int, int, int, int, int, int, int);
IgemmArgs convt_fwd_args(const void*, int, int, const void*, const float*, void*, int, int, int, int, int, int, int);
IgemmArgs convt_dgrad_args(const void*, int, int, const void*, void*, int, int, const void*, int, int, int, int, int, int, int);
int run_conv3x3_wgrad(const void*, int, int, const void*, int, int, float*, float*, float, float, void*, size_t, int, int,
                      int, int, int, int, int, hipStream_t, const char**, FinJobs*);
int run_convt_wgrad(const void*, int, int, const void*, int, int, float*, float*, float, float, void*, size_t, int, int,
                    int, int, int, int, hipStream_t, const char**, FinJobs*);
bool conv3x3_wgrad_pairable(int, int, int, int, int, int, int, int, int, int);
bool conv3x3_takes_dconv(const IgemmArgs&, int);
int run_conv3x3_wgrad_pair(const Conv3WgradCall&, const Conv3WgradCall&, float, float, int, int, int, int, hipStream_t,
                           const char**, FinJobs*);

enum LayerKind { K_CONV3 = 0, K_CONVT = 1, K_CONV1 = 2 };
struct Layer {
  std::string prefix;
  int kind, cin, cout, split;
  int64_t w_off, b_off;      // element offsets into the flat fp32 parameter buffer
  size_t wf_off, wb_off;     // byte offsets of the packed operands in the workspace
  size_t part_off, part_bytes;   // the layer's own region for weight-gradient partials
};
struct Tensor {              // an allocation in the workspace
  std::string name;
  size_t off, bytes;
  int B, H, W, C, cs;        // cs = channels incl. padding (NHWC) ; for NCHW fp32 cs = C
  int layout;                // 0 NHWC plan dtype, 1 fp32 NCHW
};
struct Named {               // a view exposed for parity checks
  std::string name;
  int tensor, coff, C;
};
struct Segment { int64_t lo, hi; };
struct ProfRec {
  std::string kernel, label;
  double flops, bytes;
  hipEvent_t e0, e1;
};

}  // namespace mi

using namespace mi;

struct mi_unet_plan {
  mi_unet_desc d;
  int L;
  std::vector<int> h, w;       // spatial size per level (index L = bottleneck)
  std::vector<bool> direct;    // decoder j writes its transposed conv straight into cat
  bool split0;                 // level 0: the two halves of the concat are separate tensors ("cat.0" = skip, "up.0")
  bool x_narrow;               // the input tensor's pixels are 16 channels wide although the first layer reads 32 (they overlap)
  bool fuse_head;              // the 1x1 head rides in the epilogue of the last 3x3 conv (no launch, no re-read of its output)
  bool out_resize;
  std::vector<Layer> layers;
  std::vector<Tensor> tensors;
  std::vector<Named> named;
  std::vector<Segment> segs;
  int64_t flat_elems;
  size_t ws_bytes, scratch_off, scratch_bytes;
  size_t ksplit_off = 0, ksplit_bytes = 0;   // scratch of the few-pixel conv kernel (arrival counters first)
  bool forward_done;
  bool profile;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
  size_t events_used;
  // captured launch sequences (see run_graphed)
  struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec; int state; };  // state 0 seen once, 1 ready, -1 unusable
  std::vector<GraphEntry> graphs;
  hipStream_t cap_stream;
  int graph_mode;              // MI_OPT_GRAPH: 0 off (default), 1 on
  bool serial_backward;        // MI_OPT_SERIAL_BACKWARD: everything on the caller's stream
  bool debug_streams;          // MI_OPT_DEBUG_STREAMS: report the side-stream choice on stderr
  bool pack_one_stream;        // MI_OPT_PACK_ONE_STREAM: pack the dgrad weight images on the caller's stream too
  bool no_ksplit;              // MI_OPT_NO_KSPLIT: the few-pixel conv kernel never splits K across workgroups
  std::vector<const void*> known_ws;   // workspaces a forward with MI_FWD_REPACK has initialised (packed weights, zeroed counters)
  // second stream for the weight-gradient kernels of a backward call (see fork_side)
  hipStream_t side;
  bool side_tried, side_busy;
  bool side_owned;   // the plan created `side` itself (destroyed with the plan); otherwise it is the process-wide choice
  bool defer_join;   // partial backward calls leave the side stream un-joined (see mi_unet_plan_defer_join)
  bool last_deferred;  // the last backward call ended with the side stream ordered behind it, not joined
  std::vector<hipEvent_t> sync_pool;
  size_t sync_used;

  // name -> index, filled as tensors and layers are registered (a step looks up ~150 names)
  std::unordered_map<std::string, int> tensor_index, layer_index;
  int tensor_id(const std::string& n) const {
    const auto it = tensor_index.find(n);
    return it == tensor_index.end() ? -1 : it->second;
  }
  int layer_id(const std::string& p) const {
    const auto it = layer_index.find(p);
    return it == layer_index.end() ? -1 : it->second;
  }
};

namespace {

struct View {
  void* p;
  int cs, co;
  void* p2 = nullptr;   // a two-tensor operand: channels from c2 on are channels 0.. of p2 (same stride)
  int c2 = 0;
  bool overlap = false; // the pixels overlap: the stride is smaller than the consumer's padded channel count (the input tensor)
};

constexpr int kInputTail = 64;   // zero elements behind the last pixel of a narrow input tensor (>= 16 of any dtype)
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int add_tensor(mi_unet_plan* P, const std::string& name, int B, int H, int W, int C, int layout, size_t& cursor,
               int cs_override = 0) {
  Tensor t;
  t.name = name; t.B = B; t.H = H; t.W = W; t.C = C; t.layout = layout;
  t.cs = layout == 0 ? (cs_override ? cs_override : pad_ch(C)) : C;
  t.bytes = (size_t)B * H * W * t.cs * (layout == 0 ? elem_size(P->d.dtype) : 4);
  t.off = cursor;
  cursor = align_up(cursor + t.bytes, 256);
  P->tensors.push_back(t);
  P->tensor_index.emplace(name, (int)P->tensors.size() - 1);      // the first registration of a name wins, as a linear search would
  return (int)P->tensors.size() - 1;
}

struct Run {   // per-call context
  mi_unet_plan* P;
  char* ws;
  const float* params;
  hipStream_t s;
  int rc;
  float* head_out = nullptr;   // forward: where the last conv's epilogue writes the head's output (fused head), else null
  // weight-gradient launches of the current backward segment, issued together by flush_wgrads()
  std::vector<std::function<int(hipStream_t)>> deferred;
  struct WJob { const struct Layer* l; View x, dy; float* grads; float alpha, beta; int B, H, W; std::string label; };
  std::vector<WJob> wjobs;   // the 3x3 layers among them (two over the same pixels share one launch)
  FinJobs fin{};   // their combine jobs, one launch per segment
  View view(const std::string& name, int coff = 0) const {
    const int id = P->tensor_id(name);
    const Tensor& t = P->tensors[id];
    return View{ws + t.off, t.cs, coff};
  }
  const float* W(const Layer& l) const { return params + l.w_off; }
  const float* Bv(const Layer& l) const { return params + l.b_off; }
};

struct Scope {   // brackets one launch with events when profiling is on
  mi_unet_plan* P;
  hipStream_t s;
  int idx;
  Scope(mi_unet_plan* p, hipStream_t st, const char* label, double flops, double bytes) : P(p), s(st), idx(-1) {
    if (!P->profile) return;
    while (P->event_pool.size() < P->events_used + 2) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      P->event_pool.push_back(e);
    }
    ProfRec r;
    r.kernel = "?"; r.label = label; r.flops = flops; r.bytes = bytes;
    r.e0 = P->event_pool[P->events_used++];
    r.e1 = P->event_pool[P->events_used++];
    (void)hipEventRecord(r.e0, s);
    P->prof.push_back(r);
    idx = (int)P->prof.size() - 1;
  }
  void kernel(const char* k) { if (idx >= 0 && k) P->prof[idx].kernel = k; }
  ~Scope() { if (idx >= 0) (void)hipEventRecord(P->prof[idx].e1, s); }
};

double esz(const mi_unet_plan* P) { return (double)elem_size(P->d.dtype); }

// A layer's weight gradient needs its input and its output gradient, and nothing later in the
// backward pass reads it.  The wgrad kernels therefore run on a second stream beside the dgrad
// chain: fork_side() makes that stream wait for everything queued so far on the caller's stream
// (the producer of dy) and returns it; join_side() at the end of the call makes the caller's
// stream wait for the side stream, so in the caller's stream order the whole call is complete.
// Per-launch profiling keeps everything on one stream (the timings must not overlap), and so
// does the plan option MI_OPT_SERIAL_BACKWARD.
hipEvent_t next_sync_event(mi_unet_plan* P) {
  if (P->sync_used == P->sync_pool.size()) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    P->sync_pool.push_back(e);
  }
  return P->sync_pool[P->sync_used++];
}
// Calibration kernels for the choice of the side stream.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  int it = 0;
  while (wall_clock64() - t0 < ticks && ++it < (1 << 22)) {}
}
__global__ void noop_kernel() {}
// HIP maps streams onto a few hardware queues, and two streams on one queue run their kernels one
// after the other (which queue a new stream gets depends on what the process created before: with
// the collective library initialised first, the plain "next stream" shared the caller's queue and
// the two-stream backward ran slower than the one-stream one).  So the side stream is CHOSEN: of a
// handful of new streams, the first on which a trivial kernel finishes while a 50 us spin kernel
// still occupies the caller's stream.  If none does, backward stays on one stream.
// The choice is made once per (device, caller stream) and shared by every plan of the process (the
// one piece of process-wide state in the library: a cache of measurements, holding one stream per
// pair for the life of the process).  Rejected candidates are destroyed once the choice is made --
// not before, so that a later candidate is never handed the slot of an earlier one.
static std::mutex g_side_mutex;
static std::map<std::pair<int, hipStream_t>, hipStream_t> g_side_choice;
static hipStream_t choose_side_stream(hipStream_t main, bool debug) {
  constexpr int kCandidates = 8;
  hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
  hipStream_t chosen = nullptr;
  hipStream_t rejected[kCandidates];
  int n_rejected = 0;
  int tried = 0;
  if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&ea) == hipSuccess && hipEventCreate(&eb) == hipSuccess) {
    for (; tried < kCandidates && !chosen; ++tried) {
      hipStream_t c = nullptr;
      if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
      bool ok = hipEventRecord(e0, main) == hipSuccess;
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, main, 5000ll);      // 50 us at the 100 MHz wall clock
      ok = ok && hipEventRecord(ea, main) == hipSuccess;
      hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, c);
      ok = ok && hipEventRecord(eb, c) == hipSuccess;
      ok = ok && hipEventSynchronize(ea) == hipSuccess && hipEventSynchronize(eb) == hipSuccess;
      float ta = 0.f, tb = 0.f;
      ok = ok && hipEventElapsedTime(&ta, e0, ea) == hipSuccess && hipEventElapsedTime(&tb, e0, eb) == hipSuccess;
      if (ok && tb < 0.5f * ta) chosen = c;
      else rejected[n_rejected++] = c;
    }
  }
  for (int i = 0; i < n_rejected; ++i) (void)hipStreamDestroy(rejected[i]);   // idle: both test kernels were waited for
  (void)hipGetLastError();
  if (e0) (void)hipEventDestroy(e0);
  if (ea) (void)hipEventDestroy(ea);
  if (eb) (void)hipEventDestroy(eb);
  if (debug) fprintf(stderr, "[unet_tm] side stream: %s after %d candidates\n", chosen ? "found" : "none", tried);
  return chosen;
}
static void ensure_side(mi_unet_plan* P, hipStream_t main) {
  if (P->side_tried) return;
  P->side_tried = true;
  if (P->serial_backward) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_side_mutex);
  const auto key = std::make_pair(dev, main);
  auto it = g_side_choice.find(key);
  if (it != g_side_choice.end()) { P->side = it->second; return; }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    // first use under a capture: no measurement possible, take a plain new stream for this plan only
    (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&P->side, hipStreamNonBlocking) != hipSuccess) P->side = nullptr;
    P->side_owned = true;
    return;
  }
  P->side = choose_side_stream(main, P->debug_streams);
  g_side_choice[key] = P->side;
}
hipStream_t fork_side(Run& r) {
  mi_unet_plan* P = r.P;
  if (P->profile) return r.s;
  ensure_side(P, r.s);
  if (!P->side) return r.s;
  hipEvent_t e = next_sync_event(P);
  if (!e || hipEventRecord(e, r.s) != hipSuccess || hipStreamWaitEvent(P->side, e, 0) != hipSuccess) return r.s;
  P->side_busy = true;
  return P->side;
}
int join_side(Run& r) {
  mi_unet_plan* P = r.P;
  if (!P->side_busy) return MI_OK;
  P->side_busy = false;
  hipEvent_t e = next_sync_event(P);
  P->sync_used = 0;   // a wait refers to its event's record at the time of the call: the next call reuses the pool
  if (!e) return fail(MI_ESTATE, "joining the second stream: cannot create an event");
  int rc = check_hip(hipEventRecord(e, P->side), "joining the second stream: event record");
  if (rc == MI_OK) rc = check_hip(hipStreamWaitEvent(r.s, e, 0), "joining the second stream: stream wait");
  return rc;
}

// Bucketed caller (defer_join on, more segments to come): the side stream catches up with this call's
// main-stream work instead of the main stream waiting for the weight gradients; the caller reduces the
// finished range behind the side stream and the dgrad chain of the next call starts at once.  True when
// the side stream is now ordered behind everything the call enqueued (also when the call put nothing on
// it, as a head-only call does); false when there is no usable side stream -- the caller of this
// function then joins as usual and the range is complete in the caller's stream order.
bool defer_to_side(Run& r, int seg_hi) {
  mi_unet_plan* P = r.P;
  if (!P->defer_join || seg_hi >= (int)P->segs.size()) return false;
  if (fork_side(r) == r.s) return false;
  P->last_deferred = true;
  return true;
}

#define RUN(expr)                 \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != MI_OK) return rc__; \
  } while (0)

int do_conv_fwd(Run& r, const Layer& l, View x, View y, int B, int H, int W, const char* label, const View* pool = nullptr,
                const Layer* head = nullptr) {
  const double fl = 2.0 * B * H * W * (double)l.cout * l.cin * 9;
  const double by = (double)B * H * W * (l.cin + l.cout * (pool ? 1.25 : 1.0)) * esz(r.P);
  Scope sc(r.P, r.s, label, fl, by);
  const char* k = nullptr;
  IgemmArgs a = conv3x3_fwd_args(x.p, x.cs, x.co, r.ws + l.wf_off, r.Bv(l), y.p, y.cs, y.co, B, H, W, l.cin, l.split,
                                 l.cout, 1);
  if (pool) { a.pool = pool->p; a.pool_cs = pool->cs; a.pool_co = pool->co; }
  if (x.p2) { a.in2 = x.p2; a.in2_c0 = x.c2; }
  a.in_overlap = x.overlap;
  if (head) { a.head_w = r.W(*head); a.head_b = r.Bv(*head); a.head_out = r.head_out; a.head_cout = head->cout; }
  if (r.P->ksplit_bytes && !r.P->no_ksplit) { a.kscratch = r.ws + r.P->ksplit_off; a.kscratch_bytes = r.P->ksplit_bytes; }
  int rc = launch_igemm(a, r.P->d.dtype, r.s, &k);
  sc.kernel(k);
  return rc;
}
int do_conv_dgrad(Run& r, const Layer& l, View dy, View dx, const View* mask, int B, int H, int W, const char* label) {
  const double fl = 2.0 * B * H * W * (double)l.cout * l.cin * 9;
  const double by = (double)B * H * W * (l.cin * (mask ? 2 : 1) + l.cout) * esz(r.P);
  Scope sc(r.P, r.s, label, fl, by);
  const char* k = nullptr;
  IgemmArgs a = conv3x3_dgrad_args(dy.p, dy.cs, dy.co, r.ws + l.wb_off, dx.p, dx.cs, dx.co, mask ? mask->p : nullptr,
                                   mask ? mask->cs : 0, mask ? mask->co : 0, B, H, W, l.cin, l.split, l.cout);
  if (dx.p2) { a.out2 = dx.p2; a.out2_n0 = dx.c2; }
  if (r.P->ksplit_bytes && !r.P->no_ksplit) { a.kscratch = r.ws + r.P->ksplit_off; a.kscratch_bytes = r.P->ksplit_bytes; }
  int rc = launch_igemm(a, r.P->d.dtype, r.s, &k);
  sc.kernel(k);
  return rc;
}
int do_conv_wgrad(Run& r, const Layer& l, View x, View dy, float* grads, float alpha, float beta, int B, int H, int W,
                  const char* label) {
  r.wjobs.push_back(Run::WJob{&l, x, dy, grads, alpha, beta, B, H, W, label});
  return MI_OK;
}
static Conv3WgradCall wgrad_call(const Run& r, const Run::WJob& j) {
  const Layer& l = *j.l;
  return Conv3WgradCall{j.x.p, j.x.cs, j.x.co, j.x.p2, j.x.c2, j.dy.p, j.dy.cs, j.dy.co, j.grads + l.w_off, j.grads + l.b_off,
                        r.ws + l.part_off, l.part_bytes, l.cin, l.split, l.cout};
}
static int launch_conv_wgrads(Run& r, hipStream_t ws) {
  auto flops = [](const Run::WJob& j) { return 2.0 * j.B * j.H * j.W * (double)j.l->cout * j.l->cin * 9; };
  auto bytes = [&](const Run::WJob& j) { return (double)j.B * j.H * j.W * (j.l->cin + j.l->cout) * esz(r.P); };
  const int dt = r.P->d.dtype;
  size_t i = 0;
  while (i < r.wjobs.size()) {
    const Run::WJob& a = r.wjobs[i];
    if (i + 1 < r.wjobs.size()) {
      const Run::WJob& b = r.wjobs[i + 1];
      if (a.B == b.B && a.H == b.H && a.W == b.W && a.alpha == b.alpha && a.beta == b.beta &&
          conv3x3_wgrad_pairable(a.B, a.H, a.W, a.l->cin, a.l->split, a.l->cout, b.l->cin, b.l->split, b.l->cout, dt)) {
        Scope sc(r.P, ws, (a.label + "+" + b.l->prefix).c_str(), flops(a) + flops(b), bytes(a) + bytes(b));
        const char* k = nullptr;
        int rc = run_conv3x3_wgrad_pair(wgrad_call(r, a), wgrad_call(r, b), a.alpha, a.beta, a.B, a.H, a.W, dt, ws, &k, &r.fin);
        sc.kernel(k);
        if (rc) return rc;
        i += 2;
        continue;
      }
    }
    const Layer& l = *a.l;
    if (a.x.p2) return fail(MI_ESTATE, "unet_backward: a two-tensor input reached the unpaired weight-gradient path");
    Scope sc(r.P, ws, a.label.c_str(), flops(a), bytes(a));
    const char* k = nullptr;
    int rc = run_conv3x3_wgrad(a.x.p, a.x.cs, a.x.co, a.dy.p, a.dy.cs, a.dy.co, a.grads + l.w_off, a.grads + l.b_off, a.alpha,
                               a.beta, r.ws + l.part_off, l.part_bytes, a.B, a.H, a.W, l.cin, l.split, l.cout, dt, ws, &k, &r.fin);
    sc.kernel(k);
    if (rc) return rc;
    ++i;
  }
  r.wjobs.clear();
  return MI_OK;
}
// Issues the deferred weight-gradient launches of a segment on the side stream behind ONE fork
// (an event record + stream wait per layer costs more host time than the launches themselves),
// then ONE launch that combines the partials of all of them: every layer has its own partial
// region, so the sums need not sit between the weight-gradient kernels as dependent launches.
int flush_wgrads(Run& r, int seg) {
  if (r.deferred.empty() && r.wjobs.empty()) return MI_OK;
  hipStream_t ws = fork_side(r);
  r.fin.n = 0;
  int rc = launch_conv_wgrads(r, ws);
  if (rc) { r.wjobs.clear(); r.deferred.clear(); return rc; }
  for (auto& f : r.deferred) {
    rc = f(ws);
    if (rc) break;
  }
  r.deferred.clear();
  if (rc) return rc;
  double bytes = 0;
  for (int i = 0; i < r.fin.n; ++i) {
    const FinJob& q = r.fin.job[i];
    bytes += 4.0 * (q.kind == FIN_CONVT ? 4.0 * q.Cin * q.Cout : 9.0 * q.Cin * q.Cout) * (q.splits + 1);
  }
  Scope sc(r.P, ws, ("seg" + std::to_string(seg) + ".wgrad.reduce").c_str(), 0.0, bytes);
  sc.kernel("finalize");
  return launch_finalize_multi(r.fin, ws);
}

// Launch-sequence graphs.  A training step is ~110 dependent launches of 10-30 us kernels: issued
// one by one the host barely keeps ahead of the GPU.  The part of a forward / backward call that
// touches only plan-owned memory (workspace, packed weights) and the caller's long-lived buffers
// (flat parameters, flat gradients) is therefore captured into a HIP graph the second time the
// same arguments are seen and replayed with one launch afterwards.  The kernels that read or
// write per-call buffers (network input, prediction, loss gradient, input gradient) stay outside
// the graph, so new batches never invalidate it.  Capture runs on a private stream (the caller's
// may be the legacy default stream, which cannot capture); the side stream of the wgrad kernels
// joins the capture through its fork/join events.
// Off by default, the plan option MI_OPT_GRAPH turns it on: measured on MI355X (batch 512, 32x32) the step is
// GPU-bound, replay is no faster than eager launches on one stream (1.81 ms either way), and a
// graph loses the overlap the eager two-stream backward gets (1.83 ms against 1.70 ms).  It pays
// only where the host cannot keep ahead (small batches, slow hosts).
constexpr size_t kMaxGraphs = 12;
template <typename Body>
int run_graphed(mi_unet_plan* P, const std::vector<uint64_t>& key, hipStream_t stream, Body body) {
  if (P->profile || P->graph_mode == 0) return body(stream);
  mi_unet_plan::GraphEntry* e = nullptr;
  for (auto& g : P->graphs)
    if (g.key == key) { e = &g; break; }
  if (!e) {                                    // first sighting: run eagerly (also performs every lazy initialisation)
    if (P->graphs.size() < kMaxGraphs) P->graphs.push_back({key, nullptr, 0});
    return body(stream);
  }
  if (e->state < 0) return body(stream);
  if (e->state == 0) {
    e->state = -1;                             // stays unusable unless everything below succeeds
    if (!P->cap_stream && hipStreamCreateWithFlags(&P->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      P->cap_stream = nullptr;
      return body(stream);
    }
    if (hipStreamBeginCapture(P->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return body(stream);
    const int rc = body(P->cap_stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(P->cap_stream, &graph);
    if (rc != MI_OK || ec != hipSuccess || !graph) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      return rc != MI_OK ? rc : body(stream);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess || !exec) { (void)hipGetLastError(); return body(stream); }
    e->exec = exec;
    e->state = 1;
  }
  return check_hip(hipGraphLaunch(e->exec, stream), "graph launch");
}
enum Phase { PH_IN = 1, PH_BODY = 2, PH_OUT = 4, PH_ALL = 7 };
uint64_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

void bil_strides_nhwc(const Tensor& t, int64_t s[4]) {
  s[0] = (int64_t)t.H * t.W * t.cs; s[1] = 1; s[2] = (int64_t)t.W * t.cs; s[3] = t.cs;
}

}  // namespace

extern "C" {

int mi_unet_plan_create(const mi_unet_desc* desc, mi_unet_plan** out) {
  MI_REQUIRE(desc && out, "plan_create: null pointer");
  const mi_unet_desc& d = *desc;
  MI_REQUIRE(d.n_levels >= 1 && d.n_levels <= 8, "plan_create: n_levels must be 1..8");
  MI_REQUIRE(d.in_channels > 0 && d.out_channels > 0 && d.batch > 0 && d.height > 0 && d.width > 0,
             "plan_create: bad shape");
  MI_REQUIRE(d.dtype == MI_F32 || d.dtype == MI_BF16, "plan_create: bad dtype");
  MI_REQUIRE(d.out_channels <= 16 && pad_ch(d.features[0]) <= 64, "plan_create: head supports features[0] <= 64, out_channels <= 16");
  for (int i = 0; i < d.n_levels; ++i) {
    MI_REQUIRE(d.features[i] > 0, "plan_create: features must be positive");
    // The reference only runs when widths double (ConvTranspose2d(2f, f) consumes the
    // previous block's output, models/unet2d.py:45-48).
    if (i > 0) MI_REQUIRE(d.features[i] == 2 * d.features[i - 1], "plan_create: features must double per level");
  }
  {
    int hh = d.height, ww = d.width;
    for (int i = 0; i < d.n_levels; ++i) { hh /= 2; ww /= 2; }
    // the same failure the reference raises from max_pool2d ("Output size is too small")
    MI_REQUIRE(hh >= 1 && ww >= 1, "plan_create: input too small for this many 2x2 pools");
  }
  mi_unet_plan* P = new mi_unet_plan();
  P->d = d;
  const int L = P->L = d.n_levels;
  const int B = d.batch;
  P->forward_done = false;
  P->profile = false;
  P->events_used = 0;
  P->side = nullptr; P->side_tried = false; P->side_busy = false; P->sync_used = 0; P->defer_join = false; P->side_owned = false; P->last_deferred = false;
  P->cap_stream = nullptr; P->graph_mode = 0;
  P->serial_backward = false; P->debug_streams = false; P->pack_one_stream = false; P->no_ksplit = false;
  P->h.resize(L + 1); P->w.resize(L + 1);
  P->h[0] = d.height; P->w[0] = d.width;
  for (int l = 0; l < L; ++l) { P->h[l + 1] = P->h[l] / 2; P->w[l + 1] = P->w[l] / 2; }

  // ---- layers in construction order (models/unet2d.py:24-59)
  auto add_layer = [&](const std::string& p, int kind, int cin, int cout, int split) {
    Layer l; l.prefix = p; l.kind = kind; l.cin = cin; l.cout = cout; l.split = split;
    l.w_off = l.b_off = 0; l.wf_off = l.wb_off = 0;
    P->layers.push_back(l);
    P->layer_index.emplace(l.prefix, (int)P->layers.size() - 1);
  };
  int ch = d.in_channels;
  for (int l = 0; l < L; ++l) {
    add_layer("downs." + std::to_string(l) + ".0", K_CONV3, ch, d.features[l], 0);
    add_layer("downs." + std::to_string(l) + ".2", K_CONV3, d.features[l], d.features[l], 0);
    ch = d.features[l];
  }
  const int fb = d.features[L - 1];
  add_layer("bottleneck.0", K_CONV3, fb, 2 * fb, 0);
  add_layer("bottleneck.2", K_CONV3, 2 * fb, 2 * fb, 0);
  for (int j = 0; j < L; ++j) {
    const int f = d.features[L - 1 - j];
    add_layer("ups." + std::to_string(2 * j), K_CONVT, 2 * f, f, 0);
    add_layer("ups." + std::to_string(2 * j + 1) + ".0", K_CONV3, 2 * f, f, f);
    add_layer("ups." + std::to_string(2 * j + 1) + ".2", K_CONV3, f, f, 0);
  }
  add_layer("final_conv", K_CONV1, d.features[0], d.out_channels, 0);
  int64_t off = 0;
  for (auto& l : P->layers) {
    const int64_t nw = l.kind == K_CONV3 ? (int64_t)l.cout * l.cin * 9
                     : l.kind == K_CONVT ? (int64_t)l.cin * l.cout * 4 : (int64_t)l.cout * l.cin;
    l.w_off = off; off = round_up(off + nw, 64);
    l.b_off = off; off = round_up(off + l.cout, 64);
  }
  P->flat_elems = off;

  // ---- backward segments (execution order; flat ranges descend)
  auto seg_of = [&](int first, int last) {
    Segment s; s.lo = P->layers[first].w_off;
    s.hi = last + 1 < (int)P->layers.size() ? P->layers[last + 1].w_off : P->flat_elems;
    return s;
  };
  const int n_layers = (int)P->layers.size();
  P->segs.push_back(seg_of(n_layers - 1, n_layers - 1));                   // head
  for (int j = L - 1; j >= 0; --j) {                                       // decoder blocks, last first
    const int first = P->layer_id("ups." + std::to_string(2 * j));
    P->segs.push_back(seg_of(first, first + 2));
  }
  { const int first = P->layer_id("bottleneck.0"); P->segs.push_back(seg_of(first, first + 1)); }
  for (int l = L - 1; l >= 0; --l) {
    const int first = P->layer_id("downs." + std::to_string(l) + ".0");
    P->segs.push_back(seg_of(first, first + 1));
  }

  // ---- workspace tensors
  size_t cur = 0;
  const bool train = true;  // gradient tensors are laid out after the forward ones; an inference-only
                            // caller simply passes mi_unet_plan_workspace_bytes(plan) minus nothing: see fwd_bytes
  (void)train;
  // Level 0 with at most 32 channels: a pixel of the concat buffer would be 128 bytes, 64 per half, and every
  // kernel that touches ONE half (the transposed conv and its input gradient, max-pool backward, the skip's
  // conv) would move half-used 128-byte lines.  Where the three kernels that see BOTH halves take two-tensor
  // operands (the DMA-fed conv kernel, the halo-tile weight gradient), the halves are separate tensors.
  P->split0 = false;
  if (d.dtype == MI_BF16 && pad_ch(d.features[0]) == 32 && 2 * P->h[1] == P->h[0] && 2 * P->w[1] == P->w[0]) {
    const int f = d.features[0];
    static char dummy[16];
    IgemmArgs fa = conv3x3_fwd_args(dummy, 32, 0, dummy, (const float*)dummy, dummy, 32, 0, B, P->h[0], P->w[0], 2 * f, f, f, 1);
    IgemmArgs ga = conv3x3_dgrad_args(dummy, 32, 0, dummy, dummy, 32, 0, nullptr, 0, 0, B, P->h[0], P->w[0], 2 * f, f, f);
    fa.in2 = dummy; fa.in2_c0 = 32; ga.out2 = dummy; ga.out2_n0 = 32;
    P->split0 = conv3x3_takes_dconv(fa, d.dtype) && conv3x3_takes_dconv(ga, d.dtype) &&
                conv3x3_wgrad_pairable(B, P->h[0], P->w[0], f, 0, f, 2 * f, f, f, d.dtype);
  }
  // The 1x1 head in the epilogue of the last 3x3 conv, where that conv runs as ONE 32-channel tile of the DMA-fed kernel.
  P->fuse_head = false;
  if (d.dtype == MI_BF16 && pad_ch(d.features[0]) == 32 && d.out_channels <= 2) {
    const int f = d.features[0];
    static char dummy[16];
    IgemmArgs fa = conv3x3_fwd_args(dummy, 32, 0, dummy, (const float*)dummy, dummy, 32, 0, B, P->h[0], P->w[0], f, 0, f, 1);
    fa.head_w = fa.head_b = (const float*)dummy; fa.head_out = (float*)dummy; fa.head_cout = d.out_channels;
    P->fuse_head = conv3x3_takes_dconv(fa, d.dtype);
  }
  // The input: up to 16 channels are stored as 16-channel pixels, although the first layer's K is padded to 32 -- its
  // chunk then runs on into the next pixel, whose values meet the zero weights of the padding (IgemmArgs::in_overlap).
  // A 12-channel input costs half the writes of the layout pass and half the reads of the first layer and of its
  // weight gradient.  Behind the last pixel lie kInputTail zero elements (written by the layout pass).
  P->x_narrow = d.in_channels <= 16 && pad_ch(d.in_channels) > 16;
  add_tensor(P, "x", B, d.height, d.width, d.in_channels, 0, cur, P->x_narrow ? 16 : 0);
  if (P->x_narrow) cur += 256;
  for (int l = 0; l < L; ++l) {
    const int f = d.features[l];
    add_tensor(P, "a1." + std::to_string(l), B, P->h[l], P->w[l], f, 0, cur);
    if (l == 0 && P->split0) {
      add_tensor(P, "cat.0", B, P->h[0], P->w[0], f, 0, cur);
      add_tensor(P, "up.0", B, P->h[0], P->w[0], f, 0, cur);
    } else {
      add_tensor(P, "cat." + std::to_string(l), B, P->h[l], P->w[l], 2 * f, 0, cur, 2 * pad_ch(f));
    }
    add_tensor(P, "pool." + std::to_string(l), B, P->h[l + 1], P->w[l + 1], f, 0, cur);
  }
  add_tensor(P, "b1", B, P->h[L], P->w[L], 2 * fb, 0, cur);
  add_tensor(P, "b2", B, P->h[L], P->w[L], 2 * fb, 0, cur);
  P->direct.resize(L);
  for (int j = 0; j < L; ++j) {
    const int l = L - 1 - j, f = d.features[l];
    const int hin = P->h[l + 1], win = P->w[l + 1];
    P->direct[j] = (2 * hin == P->h[l] && 2 * win == P->w[l]);
    if (!P->direct[j]) add_tensor(P, "upraw." + std::to_string(j), B, 2 * hin, 2 * win, f, 0, cur);
    add_tensor(P, "u1." + std::to_string(j), B, P->h[l], P->w[l], f, 0, cur);
    add_tensor(P, "u2." + std::to_string(j), B, P->h[l], P->w[l], f, 0, cur);
  }
  P->out_resize = d.out_h > 0 && d.out_w > 0 && (d.out_h != d.height || d.out_w != d.width);
  if (P->out_resize) add_tensor(P, "pred0", B, d.height, d.width, d.out_channels, 1, cur);
  // packed weights
  for (auto& l : P->layers) {
    if (l.kind == K_CONV3) {
      l.wf_off = cur; cur = align_up(cur + mi_packed_conv3x3_bytes(l.cout, l.cin, l.split, d.dtype, 0), 256);
      l.wb_off = cur; cur = align_up(cur + mi_packed_conv3x3_bytes(l.cout, l.cin, l.split, d.dtype, 1), 256);
    } else if (l.kind == K_CONVT) {
      l.wf_off = cur; cur = align_up(cur + mi_packed_convt2x2_bytes(l.cin, l.cout, d.dtype, 0), 256);
      l.wb_off = cur; cur = align_up(cur + mi_packed_convt2x2_bytes(l.cin, l.cout, d.dtype, 1), 256);
    }
  }
  // gradient tensors
  if (d.need_input_grad) add_tensor(P, "g.x", B, d.height, d.width, d.in_channels, 0, cur);
  for (int l = 0; l < L; ++l) {
    const int f = d.features[l];
    add_tensor(P, "g.a1." + std::to_string(l), B, P->h[l], P->w[l], f, 0, cur);
    if (l == 0 && P->split0) {
      add_tensor(P, "g.cat.0", B, P->h[0], P->w[0], f, 0, cur);
      add_tensor(P, "g.up.0", B, P->h[0], P->w[0], f, 0, cur);
    } else {
      add_tensor(P, "g.cat." + std::to_string(l), B, P->h[l], P->w[l], 2 * f, 0, cur, 2 * pad_ch(f));
    }
    // total gradient of the skip tensor (concat half + pool path, ReLU-masked), stored densely: the
    // encoder's wgrad/dgrad then read whole rows instead of half of every g.cat row
    add_tensor(P, "g.skip." + std::to_string(l), B, P->h[l], P->w[l], f, 0, cur);
    add_tensor(P, "g.pool." + std::to_string(l), B, P->h[l + 1], P->w[l + 1], f, 0, cur);
  }
  add_tensor(P, "g.b1", B, P->h[L], P->w[L], 2 * fb, 0, cur);
  add_tensor(P, "g.b2", B, P->h[L], P->w[L], 2 * fb, 0, cur);
  for (int j = 0; j < L; ++j) {
    const int l = L - 1 - j, f = d.features[l];
    if (!P->direct[j]) add_tensor(P, "g.upraw." + std::to_string(j), B, 2 * P->h[l + 1], 2 * P->w[l + 1], f, 0, cur);
    add_tensor(P, "g.u1." + std::to_string(j), B, P->h[l], P->w[l], f, 0, cur);
    add_tensor(P, "g.u2." + std::to_string(j), B, P->h[l], P->w[l], f, 0, cur);
  }
  if (P->out_resize) add_tensor(P, "g.pred0", B, d.height, d.width, d.out_channels, 1, cur);
  // weight-gradient partials: the head's scratch, then one region per layer (the partials of a
  // backward segment are combined by one launch after all its weight-gradient kernels)
  P->scratch_off = cur; P->scratch_bytes = align_up(head_ws_bytes(d.features[0], d.out_channels), 256);
  cur += P->scratch_bytes;
  {
    // scratch for splitting K across workgroups: only when some level is small enough in pixels
    // and wide enough in channels (weight rows of 8 KiB and up) for the few-pixel conv kernel to use it
    bool any = false;
    for (int l = 0; l <= L; ++l) {
      const int f = l < L ? d.features[l] : 2 * fb;
      const int64_t M = (int64_t)B * P->h[l] * P->w[l];
      const int kmax = 9 * (l < L ? 2 * pad_ch(f) : pad_ch(f));
      if (M * pad_ch(l < L ? f : f / 2) < sconv_limit(d.dtype, P->h[l] * P->w[l]) && kmax * (int)elem_size(d.dtype) >= 8192) any = true;
    }
    if (any) { P->ksplit_off = cur; P->ksplit_bytes = kKsplitBytes; cur += kKsplitBytes; }
  }
  {
    int lvl_h, lvl_w;
    for (auto& l : P->layers) {
      // spatial size of the layer's OUTPUT grid for conv3 / INPUT grid for convT
      if (l.prefix.rfind("downs.", 0) == 0) { const int i = std::stoi(l.prefix.substr(6)); lvl_h = P->h[i]; lvl_w = P->w[i]; }
      else if (l.prefix.rfind("bottleneck.", 0) == 0) { lvl_h = P->h[L]; lvl_w = P->w[L]; }
      else if (l.prefix.rfind("ups.", 0) == 0) {
        const int idx = std::stoi(l.prefix.substr(4)); const int lv = L - 1 - idx / 2;
        if (l.kind == K_CONVT) { lvl_h = P->h[lv + 1]; lvl_w = P->w[lv + 1]; } else { lvl_h = P->h[lv]; lvl_w = P->w[lv]; }
      } else continue;
      const size_t need = l.kind == K_CONV3 ? mi_conv3x3_wgrad_ws_bytes(B, lvl_h, lvl_w, l.cin, l.split, l.cout, d.dtype)
                                            : mi_convt2x2_wgrad_ws_bytes(B, lvl_h, lvl_w, l.cin, l.cout, d.dtype);
      l.part_off = cur; l.part_bytes = align_up(need, 256);
      cur += l.part_bytes;
    }
  }
  P->ws_bytes = cur;

  // ---- named views for parity checks (names follow the reference's module paths)
  auto name_view = [&](const std::string& n, const std::string& tensor, int coff, int C) {
    const int id = P->tensor_id(tensor);
    if (id >= 0) P->named.push_back(Named{n, id, coff, C});
  };
  for (int pass = 0; pass < 2; ++pass) {
    const std::string a = pass ? "grad." : "act.", t = pass ? "g." : "";
    if (pass == 0) name_view("act.x", "x", 0, d.in_channels); else name_view("grad.x", "g.x", 0, d.in_channels);
    for (int l = 0; l < L; ++l) {
      const int f = d.features[l]; const std::string s = std::to_string(l);
      name_view(a + "downs." + s + ".1", t + "a1." + s, 0, f);
      name_view(a + "downs." + s + ".3", pass ? "g.skip." + s : "cat." + s, 0, f);
      name_view(a + "pool." + s, t + "pool." + s, 0, f);
    }
    name_view(a + "bottleneck.1", t + "b1", 0, 2 * fb);
    name_view(a + "bottleneck.3", t + "b2", 0, 2 * fb);
    for (int j = 0; j < L; ++j) {
      const int l = L - 1 - j, f = d.features[l]; const std::string s = std::to_string(j), sl = std::to_string(l);
      if (l == 0 && P->split0) {
        name_view(a + "ups." + std::to_string(2 * j), t + "up.0", 0, f);
      } else if (P->direct[j]) {
        name_view(a + "ups." + std::to_string(2 * j), t + "cat." + sl, pad_ch(f), f);
      } else {
        name_view(a + "ups." + std::to_string(2 * j), t + "upraw." + s, 0, f);
        name_view(a + "align." + s, t + "cat." + sl, pad_ch(f), f);
      }
      name_view(a + "ups." + std::to_string(2 * j + 1) + ".1", t + "u1." + s, 0, f);
      name_view(a + "ups." + std::to_string(2 * j + 1) + ".3", t + "u2." + s, 0, f);
    }
    if (P->out_resize) name_view(a + "final_conv", t + "pred0", 0, d.out_channels);
  }
  *out = P;
  return MI_OK;
}

void mi_unet_plan_destroy(mi_unet_plan* plan) {
  if (!plan) return;
  for (hipEvent_t e : plan->event_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : plan->sync_pool) (void)hipEventDestroy(e);
  if (plan->side && plan->side_owned) (void)hipStreamDestroy(plan->side);
  for (auto& g : plan->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (plan->cap_stream) (void)hipStreamDestroy(plan->cap_stream);
  delete plan;
}
size_t mi_unet_plan_workspace_bytes(const mi_unet_plan* plan) { return plan ? plan->ws_bytes : 0; }
int64_t mi_unet_plan_flat_elems(const mi_unet_plan* plan) { return plan ? plan->flat_elems : 0; }
int mi_unet_plan_num_params(const mi_unet_plan* plan) { return plan ? 2 * (int)plan->layers.size() : 0; }
int mi_unet_plan_param_info(const mi_unet_plan* plan, int i, char* name, int name_cap, int64_t* flat_offset,
                            int64_t shape[4], int* ndim) {
  MI_REQUIRE(plan && i >= 0 && i < 2 * (int)plan->layers.size(), "param_info: index out of range");
  const Layer& l = plan->layers[i / 2];
  const bool is_w = (i % 2) == 0;
  const std::string n = l.prefix + (is_w ? ".weight" : ".bias");
  if (name && name_cap > 0) { strncpy(name, n.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (flat_offset) *flat_offset = is_w ? l.w_off : l.b_off;
  if (shape && ndim) {
    if (!is_w) { shape[0] = l.cout; *ndim = 1; }
    else if (l.kind == K_CONV3) { shape[0] = l.cout; shape[1] = l.cin; shape[2] = shape[3] = 3; *ndim = 4; }
    else if (l.kind == K_CONVT) { shape[0] = l.cin; shape[1] = l.cout; shape[2] = shape[3] = 2; *ndim = 4; }
    else { shape[0] = l.cout; shape[1] = l.cin; shape[2] = shape[3] = 1; *ndim = 4; }
  }
  return MI_OK;
}
int mi_unet_plan_num_tensors(const mi_unet_plan* plan) { return plan ? (int)plan->named.size() : 0; }
int mi_unet_plan_tensor_info(const mi_unet_plan* plan, int i, char* name, int name_cap, int64_t* ws_offset_bytes,
                             int dims[4], int* cstride, int* coff, int* layout) {
  MI_REQUIRE(plan && i >= 0 && i < (int)plan->named.size(), "tensor_info: index out of range");
  const Named& v = plan->named[i];
  const Tensor& t = plan->tensors[v.tensor];
  if (name && name_cap > 0) { strncpy(name, v.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (ws_offset_bytes) *ws_offset_bytes = (int64_t)t.off;
  if (dims) { dims[0] = t.B; dims[1] = t.H; dims[2] = t.W; dims[3] = v.C; }
  if (cstride) *cstride = t.cs;
  if (coff) *coff = v.coff;
  if (layout) *layout = t.layout;
  return MI_OK;
}
int mi_unet_plan_num_segments(const mi_unet_plan* plan) { return plan ? (int)plan->segs.size() : 0; }
int mi_unet_plan_segment_range(const mi_unet_plan* plan, int seg, int64_t* lo, int64_t* hi) {
  MI_REQUIRE(plan && seg >= 0 && seg < (int)plan->segs.size(), "segment_range: index out of range");
  if (lo) *lo = plan->segs[seg].lo;
  if (hi) *hi = plan->segs[seg].hi;
  return MI_OK;
}

int mi_unet_profile_enable(mi_unet_plan* plan, int on) {
  MI_REQUIRE(plan, "profile_enable: null plan");
  plan->profile = on != 0;
  plan->prof.clear();
  plan->events_used = 0;
  return MI_OK;
}
int mi_unet_profile_count(const mi_unet_plan* plan) { return plan ? (int)plan->prof.size() : 0; }
int mi_unet_profile_read(mi_unet_plan* plan, int i, char* kernel, int kcap, char* label, int lcap, double* flops,
                         double* bytes, float* ms) {
  MI_REQUIRE(plan && i >= 0 && i < (int)plan->prof.size(), "profile_read: index out of range");
  ProfRec& r = plan->prof[i];
  if (kernel && kcap > 0) { strncpy(kernel, r.kernel.c_str(), kcap - 1); kernel[kcap - 1] = 0; }
  if (label && lcap > 0) { strncpy(label, r.label.c_str(), lcap - 1); label[lcap - 1] = 0; }
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  if (ms) {
    int rc = check_hip(hipEventSynchronize(r.e1), "profile_read: event sync");
    if (rc) return rc;
    rc = check_hip(hipEventElapsedTime(ms, r.e0, r.e1), "profile_read: elapsed");
    if (rc) return rc;
  }
  return MI_OK;
}

}  // extern "C"

namespace {
// phases: PH_IN = input layout (reads x_nchw), PH_BODY = everything on plan-owned memory,
// PH_OUT = head and forced output resize (write pred_nchw)
int forward_body(Run& r, int flags) {
  mi_unet_plan* P = r.P;
  const mi_unet_desc& d = P->d;
  const int L = P->L, B = d.batch, dt = d.dtype;
  if (flags & MI_FWD_REPACK) {
    const bool train = (flags & MI_FWD_TRAIN) != 0;
    Scope sc(P, r.s, "pack_weights", 0, 0);
    sc.kernel("pack");
    // a workspace that needs its weights packed may be new: the arrival counters of the few-pixel
    // conv kernel start at zero (the kernel itself leaves them there)
    if (P->ksplit_bytes) RUN(check_hip(hipMemsetAsync(r.ws + P->ksplit_off, 0, kKsplitCounterBytes, r.s), "zero the K-split counters"));
    // The forward images are needed at once; the backward images (dgrad operands) not before the
    // backward pass: in training they are packed on the second stream, which has nothing else to do
    // during a forward pass, and joined at the end of this call.
    hipStream_t bs = train && !P->pack_one_stream ? fork_side(r) : r.s;
    for (int pass = 0; pass < (train ? 2 : 1); ++pass) {
      hipStream_t ps = pass == 0 ? r.s : bs;
      PackJobs J;
      J.n = 0;
      for (const auto& l : P->layers) {
        if (l.kind != K_CONV3 && l.kind != K_CONVT) continue;
        if (J.n == kMaxPackJobs) { RUN(launch_pack_multi(J, dt, ps)); J.n = 0; }
        if (!pack_jobs_add(J, r.W(l), pass == 0 ? r.ws + l.wf_off : nullptr, pass == 1 ? r.ws + l.wb_off : nullptr,
                           l.kind == K_CONVT, l.cout, l.cin, l.split))
          return fail(MI_EINVAL, "unet_forward: layer too large to pack");
      }
      RUN(launch_pack_multi(J, dt, ps));
    }
  }
  View cur = r.view("x");
  cur.overlap = P->x_narrow;
  for (int l = 0; l < L; ++l) {
    const std::string s = std::to_string(l);
    const int f = d.features[l];
    const Layer& c1 = P->layers[P->layer_id("downs." + s + ".0")];
    const Layer& c2 = P->layers[P->layer_id("downs." + s + ".2")];
    View a1 = r.view("a1." + s), cat = r.view("cat." + s), pool = r.view("pool." + s);
    RUN(do_conv_fwd(r, c1, cur, a1, B, P->h[l], P->w[l], c1.prefix.c_str()));
    // the 2x2 max-pool rides in the epilogue of the level's second conv where the tile geometry allows
    const bool fused = conv3x3_fwd_fuses_pool(B, P->h[l], P->w[l], c2.cin, c2.split, c2.cout, dt);
    RUN(do_conv_fwd(r, c2, a1, cat, B, P->h[l], P->w[l], c2.prefix.c_str(), fused ? &pool : nullptr));
    if (!fused) {
      Scope sc(P, r.s, ("pool." + s).c_str(), 0, (double)B * P->h[l] * P->w[l] * f * 1.25 * esz(P));
      sc.kernel("maxpool2_fwd");
      RUN(launch_maxpool2_fwd(cat.p, cat.cs, 0, pool.p, pool.cs, 0, B, P->h[l], P->w[l], f, dt, r.s));
    }
    cur = pool;
  }
  {
    const Layer& c1 = P->layers[P->layer_id("bottleneck.0")];
    const Layer& c2 = P->layers[P->layer_id("bottleneck.2")];
    View b1 = r.view("b1"), b2 = r.view("b2");
    RUN(do_conv_fwd(r, c1, cur, b1, B, P->h[L], P->w[L], c1.prefix.c_str()));
    RUN(do_conv_fwd(r, c2, b1, b2, B, P->h[L], P->w[L], c2.prefix.c_str()));
    cur = b2;
  }
  for (int j = 0; j < L; ++j) {
    const int l = L - 1 - j, f = d.features[l], fp = pad_ch(f);
    const std::string s = std::to_string(j), sl = std::to_string(l);
    const Layer& up = P->layers[P->layer_id("ups." + std::to_string(2 * j))];
    const Layer& c1 = P->layers[P->layer_id("ups." + std::to_string(2 * j + 1) + ".0")];
    const Layer& c2 = P->layers[P->layer_id("ups." + std::to_string(2 * j + 1) + ".2")];
    const int hin = P->h[l + 1], win = P->w[l + 1];
    View cat = r.view("cat." + sl);
    const bool two = l == 0 && P->split0;               // the concat of this level is two tensors
    if (two) { cat.p2 = r.view("up.0").p; cat.c2 = fp; }
    View dst = two ? r.view("up.0") : P->direct[j] ? View{cat.p, cat.cs, fp} : r.view("upraw." + s);
    {
      const double fl = 2.0 * B * hin * win * (double)up.cin * up.cout * 4;
      Scope sc(P, r.s, up.prefix.c_str(), fl, (double)B * hin * win * (up.cin + 4.0 * up.cout) * esz(P));
      const char* k = nullptr;
      IgemmArgs a = convt_fwd_args(cur.p, cur.cs, cur.co, r.ws + up.wf_off, r.Bv(up), dst.p, dst.cs, dst.co, B, hin, win,
                                   up.cin, up.cout);
      int rc = launch_igemm(a, dt, r.s, &k);
      sc.kernel(k);
      RUN(rc);
    }
    if (!P->direct[j]) {
      const Tensor& tu = P->tensors[P->tensor_id("upraw." + s)];
      const Tensor& tc = P->tensors[P->tensor_id("cat." + sl)];
      int64_t ss[4], ds[4];
      bil_strides_nhwc(tu, ss); bil_strides_nhwc(tc, ds);
      Scope sc(P, r.s, ("align." + s).c_str(), 0, (double)B * f * esz(P) * (4.0 * hin * win + P->h[l] * P->w[l]));
      sc.kernel("bilinear_fwd");
      char* dptr = (char*)cat.p + (size_t)fp * elem_size(dt);
      RUN(launch_bilinear_fwd(dst.p, ss[0], ss[1], ss[2], ss[3], dt, dptr, ds[0], ds[1], ds[2], ds[3], dt, B, fp,
                              2 * hin, 2 * win, P->h[l], P->w[l], r.s));
    }
    View u1 = r.view("u1." + s), u2 = r.view("u2." + s);
    RUN(do_conv_fwd(r, c1, cat, u1, B, P->h[l], P->w[l], c1.prefix.c_str()));
    RUN(do_conv_fwd(r, c2, u1, u2, B, P->h[l], P->w[l], c2.prefix.c_str(), nullptr,
                    (j == L - 1 && r.head_out != nullptr) ? &P->layers.back() : nullptr));
    cur = u2;
  }
  return MI_OK;
}

int forward_phases(mi_unet_plan* P, const float* x_nchw, const float* params_flat, void* ws, float* pred_nchw,
                   int flags, hipStream_t stream, int phases) {
  const mi_unet_desc& d = P->d;
  const int L = P->L, B = d.batch, dt = d.dtype;
  Run r{P, (char*)ws, params_flat, stream, 0};
  // the 1x1 head in the last conv's epilogue: not under captured launch graphs (the prediction is a per-call
  // buffer and would become part of the replayed body)
  const bool fuse_head = P->fuse_head && P->graph_mode == 0;
  float* pred0 = P->out_resize ? (float*)(r.ws + P->tensors[P->tensor_id("pred0")].off) : pred_nchw;
  if (fuse_head) r.head_out = pred0;

  if (phases & PH_IN) {
    Scope sc(P, r.s, "input_nhwc", 0, (double)B * d.height * d.width * d.in_channels * (4 + esz(P)));
    sc.kernel("nchw_to_nhwc");
    View x = r.view("x");
    if (P->x_narrow) RUN(launch_nchw_to_nhwc_narrow(x_nchw, x.p, x.cs, kInputTail, B, d.in_channels, d.height, d.width, dt, r.s));
    else RUN(launch_nchw_to_nhwc(x_nchw, x.p, x.cs, 0, B, d.in_channels, d.height, d.width, dt, r.s));
  }
  if (phases & PH_BODY) {
    // Whatever way the body ends, the second stream (the backward weight images, if they were packed
    // beside this pass) is joined: an error half-way must not leave a fork dangling for the next call.
    const int rc = forward_body(r, flags);
    const int rj = join_side(r);
    if (rc != MI_OK) return rc;
    RUN(rj);
  }
  if (phases & PH_OUT) {
    const Layer& hd = P->layers.back();
    View cur = r.view("u2." + std::to_string(L - 1));
    if (!fuse_head) {
      Scope sc(P, r.s, "final_conv", 2.0 * B * d.height * d.width * (double)hd.cin * hd.cout,
               (double)B * d.height * d.width * (hd.cin * esz(P) + 4.0 * hd.cout));
      sc.kernel("head_fwd");
      RUN(launch_head_fwd(cur.p, cur.cs, cur.co, r.W(hd), r.Bv(hd), pred0, B, d.height, d.width, hd.cin, hd.cout, dt, r.s));
    }
    if (P->out_resize) {
      const int C = d.out_channels, H = d.height, W = d.width;
      Scope sc(P, r.s, "out_resize", 0, 4.0 * B * C * ((double)H * W + (double)d.out_h * d.out_w));
      sc.kernel("bilinear_fwd");
      RUN(launch_bilinear_fwd(pred0, (int64_t)C * H * W, (int64_t)H * W, W, 1, MI_F32, pred_nchw,
                              (int64_t)C * d.out_h * d.out_w, (int64_t)d.out_h * d.out_w, d.out_w, 1, MI_F32, B, C, H, W,
                              d.out_h, d.out_w, r.s));
    }
  }
  return MI_OK;
}
}  // namespace

extern "C" {

int mi_unet_forward(mi_unet_plan* P, const float* x_nchw, const float* params_flat, void* ws, size_t ws_bytes,
                    float* pred_nchw, int flags, mi_stream stream) {
  MI_REQUIRE(P && x_nchw && params_flat && ws && pred_nchw, "unet_forward: null pointer");
  if (ws_bytes < P->ws_bytes) return fail(MI_ESIZE, "unet_forward: workspace too small");
  MI_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)params_flat & 15) == 0, "unet_forward: workspace must be 256-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // A workspace carries state between calls (packed weights; counters the few-pixel conv kernel expects at
  // zero).  Only a forward with MI_FWD_REPACK creates it: without the flag an unknown workspace is refused,
  // because the kernels would neither fault nor complain, they would write no output at all.
  bool known = false;
  for (const void* k : P->known_ws) known = known || k == ws;
  if (!known) {
    if (!(flags & MI_FWD_REPACK))
      return fail(MI_ESTATE, "unet_forward: the first forward on a workspace must carry MI_FWD_REPACK (it packs the weights and "
                             "zeroes the workspace's counters)");
    if (P->known_ws.size() >= 64) P->known_ws.erase(P->known_ws.begin());
    P->known_ws.push_back(ws);
  }
  if (P->profile) { P->prof.clear(); P->events_used = 0; }
  RUN(forward_phases(P, x_nchw, params_flat, ws, pred_nchw, flags, s, PH_IN));
  const std::vector<uint64_t> key = {1u, (uint64_t)(uintptr_t)params_flat, (uint64_t)(uintptr_t)ws, (uint64_t)flags};
  RUN(run_graphed(P, key, s, [&](hipStream_t cs) {
    return forward_phases(P, x_nchw, params_flat, ws, pred_nchw, flags, cs, PH_BODY);
  }));
  RUN(forward_phases(P, x_nchw, params_flat, ws, pred_nchw, flags, s, PH_OUT));
  P->forward_done = true;
  return MI_OK;
}

}  // extern "C"

namespace {
// phases: PH_IN = segment 0, the head (reads dpred_nchw), PH_BODY = the other segments,
// PH_OUT = layout change of the input gradient (writes dx_nchw)
int backward_segments(Run& r, const float* dpred_nchw, float* grads, float alpha, float beta, int seg_lo, int seg_hi,
                      float* dx_nchw, int phases) {
  mi_unet_plan* P = r.P;
  const mi_unet_desc& d = P->d;
  const int L = P->L, B = d.batch, dt = d.dtype;

  for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int seg_phases = seg == 0 ? PH_IN : (seg >= L + 2 ? PH_BODY | PH_OUT : PH_BODY);
    if (!(phases & seg_phases)) continue;
    if (seg == 0) {
      // ---- head (and the forced output resize in front of it)
      const Layer& hd = P->layers.back();
      const float* dp0 = dpred_nchw;
      if (P->out_resize) {
        float* g0 = (float*)(r.ws + P->tensors[P->tensor_id("g.pred0")].off);
        const int C = d.out_channels, H = d.height, W = d.width;
        Scope sc(P, r.s, "out_resize.bwd", 0, 4.0 * B * C * ((double)H * W + (double)d.out_h * d.out_w));
        sc.kernel("bilinear_bwd");
        RUN(launch_bilinear_bwd(dpred_nchw, (int64_t)C * d.out_h * d.out_w, (int64_t)d.out_h * d.out_w, d.out_w, 1,
                                MI_F32, g0, (int64_t)C * H * W, (int64_t)H * W, W, 1, MI_F32, B, C, H, W, d.out_h,
                                d.out_w, r.s));
        dp0 = g0;
      }
      const std::string s = std::to_string(L - 1);
      View u2 = r.view("u2." + s), gu2 = r.view("g.u2." + s);
      Scope sc(P, r.s, "final_conv.bwd", 4.0 * B * d.height * d.width * (double)hd.cin * hd.cout,
               (double)B * d.height * d.width * (2.0 * hd.cin * esz(P) + 4.0 * hd.cout));
      sc.kernel("head_bwd");
      RUN(launch_head_bwd(u2.p, u2.cs, 0, r.W(hd), dp0, gu2.p, gu2.cs, 0, grads + hd.w_off, grads + hd.b_off, alpha, beta,
                          r.ws + P->scratch_off, P->scratch_bytes, B, d.height, d.width, hd.cin, hd.cout, 1, dt, r.s));
    } else if (seg <= L) {
      // ---- decoder block j (levels from the output side inwards)
      const int j = L - seg, l = L - 1 - j, f = d.features[l], fp = pad_ch(f);
      const std::string s = std::to_string(j), sl = std::to_string(l);
      const Layer& up = P->layers[P->layer_id("ups." + std::to_string(2 * j))];
      const Layer& c1 = P->layers[P->layer_id("ups." + std::to_string(2 * j + 1) + ".0")];
      const Layer& c2 = P->layers[P->layer_id("ups." + std::to_string(2 * j + 1) + ".2")];
      const int H = P->h[l], W = P->w[l], hin = P->h[l + 1], win = P->w[l + 1];
      View u1 = r.view("u1." + s), gu1 = r.view("g.u1." + s), gu2 = r.view("g.u2." + s);
      View cat = r.view("cat." + sl), gcat = r.view("g.cat." + sl);
      const bool two = l == 0 && P->split0;             // the concat of this level (and its gradient) is two tensors
      if (two) { cat.p2 = r.view("up.0").p; cat.c2 = fp; gcat.p2 = r.view("g.up.0").p; gcat.c2 = fp; }
      RUN(do_conv_wgrad(r, c2, u1, gu2, grads, alpha, beta, B, H, W, (c2.prefix + ".wgrad").c_str()));
      RUN(do_conv_dgrad(r, c2, gu2, gu1, &u1, B, H, W, (c2.prefix + ".dgrad").c_str()));
      RUN(do_conv_wgrad(r, c1, cat, gu1, grads, alpha, beta, B, H, W, (c1.prefix + ".wgrad").c_str()));
      RUN(do_conv_dgrad(r, c1, gu1, gcat, nullptr, B, H, W, (c1.prefix + ".dgrad").c_str()));
      View dyT = two ? r.view("g.up.0") : View{gcat.p, gcat.cs, fp};
      if (!P->direct[j]) {
        View gup = r.view("g.upraw." + s);
        const Tensor& tu = P->tensors[P->tensor_id("g.upraw." + s)];
        const Tensor& tc = P->tensors[P->tensor_id("g.cat." + sl)];
        int64_t us[4], cs[4];
        bil_strides_nhwc(tu, us); bil_strides_nhwc(tc, cs);
        Scope sc(P, r.s, ("align." + s + ".bwd").c_str(), 0, (double)B * f * esz(P) * (4.0 * hin * win + (double)H * W));
        sc.kernel("bilinear_bwd");
        const char* src = (const char*)gcat.p + (size_t)fp * elem_size(dt);
        RUN(launch_bilinear_bwd(src, cs[0], cs[1], cs[2], cs[3], dt, gup.p, us[0], us[1], us[2], us[3], dt, B, fp,
                                2 * hin, 2 * win, H, W, r.s));
        dyT = gup;
      }
      View prev = j == 0 ? r.view("b2") : r.view("u2." + std::to_string(j - 1));
      View gprev = j == 0 ? r.view("g.b2") : r.view("g.u2." + std::to_string(j - 1));
      {
        const Layer* upp = &up;
        Run* rp = &r;
        r.deferred.push_back([=](hipStream_t ws) -> int {
          const Layer& up = *upp;
          Run& r = *rp;
          const double fl = 2.0 * B * hin * win * (double)up.cin * up.cout * 4;
          Scope sc(P, ws, (up.prefix + ".wgrad").c_str(), fl, (double)B * hin * win * (up.cin + 4.0 * up.cout) * esz(P));
          const char* k = nullptr;
          int rc = run_convt_wgrad(prev.p, prev.cs, prev.co, dyT.p, dyT.cs, dyT.co, grads + up.w_off, grads + up.b_off,
                                   alpha, beta, r.ws + up.part_off, up.part_bytes, B, hin, win, up.cin, up.cout, dt, ws, &k,
                                   &r.fin);
          sc.kernel(k);
          return rc;
        });
      }
      {
        const double fl = 2.0 * B * hin * win * (double)up.cin * up.cout * 4;
        Scope sc(P, r.s, (up.prefix + ".dgrad").c_str(), fl, (double)B * hin * win * (2.0 * up.cin + 4.0 * up.cout) * esz(P));
        const char* k = nullptr;
        IgemmArgs a = convt_dgrad_args(dyT.p, dyT.cs, dyT.co, r.ws + up.wb_off, gprev.p, gprev.cs, gprev.co, prev.p,
                                       prev.cs, prev.co, B, hin, win, up.cin, up.cout);
        int rc = launch_igemm(a, dt, r.s, &k);
        sc.kernel(k);
        RUN(rc);
      }
    } else if (seg == L + 1) {
      // ---- bottleneck
      const Layer& c1 = P->layers[P->layer_id("bottleneck.0")];
      const Layer& c2 = P->layers[P->layer_id("bottleneck.2")];
      const int H = P->h[L], W = P->w[L];
      View b1 = r.view("b1"), gb1 = r.view("g.b1"), gb2 = r.view("g.b2");
      View pin = r.view("pool." + std::to_string(L - 1)), gpin = r.view("g.pool." + std::to_string(L - 1));
      RUN(do_conv_wgrad(r, c2, b1, gb2, grads, alpha, beta, B, H, W, (c2.prefix + ".wgrad").c_str()));
      RUN(do_conv_dgrad(r, c2, gb2, gb1, &b1, B, H, W, (c2.prefix + ".dgrad").c_str()));
      RUN(do_conv_wgrad(r, c1, pin, gb1, grads, alpha, beta, B, H, W, (c1.prefix + ".wgrad").c_str()));
      RUN(do_conv_dgrad(r, c1, gb1, gpin, nullptr, B, H, W, (c1.prefix + ".dgrad").c_str()));
    } else {
      // ---- encoder level l
      const int l = L - 1 - (seg - (L + 2)), f = d.features[l];
      const std::string s = std::to_string(l);
      const Layer& c1 = P->layers[P->layer_id("downs." + s + ".0")];
      const Layer& c2 = P->layers[P->layer_id("downs." + s + ".2")];
      const int H = P->h[l], W = P->w[l];
      View a1 = r.view("a1." + s), ga1 = r.view("g.a1." + s);
      View cat = r.view("cat." + s), gcat = r.view("g.cat." + s), gpool = r.view("g.pool." + s);
      View gskip = r.view("g.skip." + s);
      if (phases & PH_BODY) {
        {
          Scope sc(P, r.s, ("pool." + s + ".bwd").c_str(), 0, (double)B * H * W * f * 3.25 * esz(P));
          sc.kernel("maxpool2_bwd");
          RUN(launch_maxpool2_bwd(cat.p, cat.cs, 0, gpool.p, gpool.cs, 0, gcat.p, gcat.cs, 0, gskip.p, gskip.cs, 0, B, H, W,
                                  f, 1, dt, r.s));
        }
        View gy = gskip;
        RUN(do_conv_wgrad(r, c2, a1, gy, grads, alpha, beta, B, H, W, (c2.prefix + ".wgrad").c_str()));
        RUN(do_conv_dgrad(r, c2, gy, ga1, &a1, B, H, W, (c2.prefix + ".dgrad").c_str()));
        View xin = l == 0 ? r.view("x") : r.view("pool." + std::to_string(l - 1));
        RUN(do_conv_wgrad(r, c1, xin, ga1, grads, alpha, beta, B, H, W, (c1.prefix + ".wgrad").c_str()));
        if (l > 0) {
          View gxin = r.view("g.pool." + std::to_string(l - 1));
          RUN(do_conv_dgrad(r, c1, ga1, gxin, nullptr, B, H, W, (c1.prefix + ".dgrad").c_str()));
        } else if (d.need_input_grad) {
          View gx = r.view("g.x");
          RUN(do_conv_dgrad(r, c1, ga1, gx, nullptr, B, H, W, (c1.prefix + ".dgrad").c_str()));
        }
      }
      if ((phases & PH_OUT) && l == 0 && d.need_input_grad) {
        View gx = r.view("g.x");
        Scope sc(P, r.s, "input_grad_nchw", 0, (double)B * H * W * d.in_channels * (4 + esz(P)));
        sc.kernel("nhwc_to_nchw");
        RUN(launch_nhwc_to_nchw(gx.p, gx.cs, 0, dx_nchw, B, d.in_channels, H, W, dt, r.s));
      }
    }
    RUN(flush_wgrads(r, seg));
  }
  return MI_OK;
}

int backward_phases(mi_unet_plan* P, const float* dpred_nchw, const float* params_flat, void* ws, float* grads,
                    float alpha, float beta, int seg_lo, int seg_hi, float* dx_nchw, hipStream_t stream, int phases) {
  Run r{P, (char*)ws, params_flat, stream, 0};
  const int rc = backward_segments(r, dpred_nchw, grads, alpha, beta, seg_lo, seg_hi, dx_nchw, phases);
  if (rc != MI_OK) {           // an error half-way: leave no fork dangling and no queued job behind
    r.wjobs.clear(); r.deferred.clear();
    (void)join_side(r);
    return rc;
  }
  if (!(phases & PH_BODY)) return MI_OK;
  if (defer_to_side(r, seg_hi)) return MI_OK;
  return join_side(r);
}
}  // namespace

extern "C" {

int mi_unet_backward(mi_unet_plan* P, const float* dpred_nchw, const float* params_flat, void* ws, size_t ws_bytes,
                     float* grads, float alpha, float beta, int seg_lo, int seg_hi, float* dx_nchw, mi_stream stream) {
  MI_REQUIRE(P && dpred_nchw && params_flat && ws && grads, "unet_backward: null pointer");
  if (ws_bytes < P->ws_bytes) return fail(MI_ESIZE, "unet_backward: workspace too small");
  if (!P->forward_done) return fail(MI_ESTATE, "unet_backward: no forward pass has been run on this plan");
  {
    bool known = false;
    for (const void* k : P->known_ws) known = known || k == ws;
    if (!known) return fail(MI_ESTATE, "unet_backward: no forward pass has been run on this workspace");
  }
  const int nseg = (int)P->segs.size();
  MI_REQUIRE(seg_lo >= 0 && seg_hi <= nseg && seg_lo <= seg_hi, "unet_backward: bad segment range");
  MI_REQUIRE(!P->d.need_input_grad || dx_nchw || seg_hi < nseg, "unet_backward: dx buffer required");
  hipStream_t s = (hipStream_t)stream;
  P->last_deferred = false;
  if (seg_lo == 0 && seg_hi > 0)
    RUN(backward_phases(P, dpred_nchw, params_flat, ws, grads, alpha, beta, 0, 1, dx_nchw, s, PH_IN));
  const int lo = seg_lo > 1 ? seg_lo : 1;
  if (lo >= seg_hi && seg_hi > seg_lo) {     // a head-only call: nothing ran on the side stream, order it all the same
    Run r{P, (char*)ws, params_flat, s, 0};
    (void)defer_to_side(r, seg_hi);
  }
  if (lo < seg_hi) {
    const std::vector<uint64_t> key = {2u, (uint64_t)(uintptr_t)params_flat, (uint64_t)(uintptr_t)ws,
                                       (uint64_t)(uintptr_t)grads, fbits(alpha), fbits(beta), (uint64_t)lo, (uint64_t)seg_hi};
    RUN(run_graphed(P, key, s, [&](hipStream_t cs) {
      return backward_phases(P, dpred_nchw, params_flat, ws, grads, alpha, beta, lo, seg_hi, dx_nchw, cs, PH_BODY);
    }));
    RUN(backward_phases(P, dpred_nchw, params_flat, ws, grads, alpha, beta, lo, seg_hi, dx_nchw, s, PH_OUT));
  }
  return MI_OK;
}

int mi_unet_plan_set_option(mi_unet_plan* P, int option, int value) {
  MI_REQUIRE(P, "unet_plan_set_option: null plan");
  switch (option) {
    case MI_OPT_GRAPH:
      if ((value != 0) != (P->graph_mode != 0)) {      // graphs captured under the other setting are dropped
        for (auto& g : P->graphs)
          if (g.exec) (void)hipGraphExecDestroy(g.exec);
        P->graphs.clear();
      }
      P->graph_mode = value != 0;
      return MI_OK;
    case MI_OPT_SERIAL_BACKWARD:
      MI_REQUIRE(!P->side_tried || P->serial_backward == (value != 0),
                 "unet_plan_set_option: MI_OPT_SERIAL_BACKWARD must be set before the plan's first training call");
      P->serial_backward = value != 0;
      return MI_OK;
    case MI_OPT_DEBUG_STREAMS: P->debug_streams = value != 0; return MI_OK;
    case MI_OPT_PACK_ONE_STREAM: P->pack_one_stream = value != 0; return MI_OK;
    case MI_OPT_NO_KSPLIT: P->no_ksplit = value != 0; return MI_OK;
    default: return fail(MI_EINVAL, "unet_plan_set_option: unknown option");
  }
}
int mi_unet_plan_ksplit_region(const mi_unet_plan* P, size_t* ws_offset_bytes, size_t* counter_bytes, size_t* total_bytes) {
  MI_REQUIRE(P, "unet_plan_ksplit_region: null plan");
  if (ws_offset_bytes) *ws_offset_bytes = P->ksplit_off;
  if (counter_bytes) *counter_bytes = P->ksplit_bytes ? kKsplitCounterBytes : 0;
  if (total_bytes) *total_bytes = P->ksplit_bytes;
  return MI_OK;
}
int mi_unet_plan_defer_join(mi_unet_plan* P, int on) {
  MI_REQUIRE(P, "unet_plan_defer_join: null plan");
  P->defer_join = on != 0;
  return MI_OK;
}
void* mi_unet_plan_side_stream(mi_unet_plan* P, mi_stream main) {
  if (!P || P->profile) return nullptr;
  ensure_side(P, (hipStream_t)main);
  return (void*)P->side;
}
int mi_unet_plan_last_call_deferred(const mi_unet_plan* P) { return P && P->last_deferred ? 1 : 0; }
// Job table of the fused Adam + packing launch; false when some layer cannot go through it.
static bool apack_jobs(const mi_unet_plan* P, char* ws, ApackJobs& J) {
  J.n = 0;
  for (const auto& l : P->layers) {
    const int tail_n = (l.cout + 3) / 4 * 4;
    if (l.kind == K_CONV1) {
      if (!apack_jobs_add(J, 2, l.cout, l.cin, 0, l.w_off, l.w_off, (int)(l.b_off - l.w_off) + tail_n, nullptr, nullptr)) return false;
      continue;
    }
    const int convt = l.kind == K_CONVT;
    if (!apack_layer_ok(convt, l.cout, l.cin, l.split, l.w_off)) return false;
    if (!apack_jobs_add(J, convt ? 1 : 0, l.cout, l.cin, l.split, l.w_off, l.b_off, tail_n, ws ? ws + l.wf_off : nullptr,
                        ws ? ws + l.wb_off : nullptr))
      return false;
  }
  return true;
}
int mi_unet_plan_adam_pack_supported(const mi_unet_plan* P) {
  if (!P) return 0;
  ApackJobs J;
  return apack_jobs(P, nullptr, J) ? 1 : 0;
}
int mi_unet_adam_pack(mi_unet_plan* P, float* params_flat, const float* grads_flat, float* exp_avg, float* exp_avg_sq,
                      void* ws, size_t ws_bytes, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                      mi_stream stream) {
  MI_REQUIRE(P && params_flat && grads_flat && exp_avg && exp_avg_sq && ws, "unet_adam_pack: null pointer");
  if (ws_bytes < P->ws_bytes) return fail(MI_ESIZE, "unet_adam_pack: workspace too small");
  MI_REQUIRE(((uintptr_t)ws & 255) == 0, "unet_adam_pack: workspace must be 256-byte aligned");
  bool known = false;
  for (const void* k : P->known_ws) known = known || k == ws;
  if (!known) return fail(MI_ESTATE, "unet_adam_pack: no forward pass with MI_FWD_REPACK has been run on this workspace");
  ApackJobs J;
  if (!apack_jobs(P, (char*)ws, J)) return fail(MI_EINVAL, "unet_adam_pack: a layer of this plan cannot go through the fused kernel");
  return launch_apack(J, P->d.dtype, params_flat, grads_flat, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale,
                      (hipStream_t)stream);
}
int mi_unet_plan_prepare(mi_unet_plan* P, mi_stream caller_stream) {
  MI_REQUIRE(P, "unet_plan_prepare: null plan");
  ensure_side(P, (hipStream_t)caller_stream);
  return MI_OK;
}

}  // extern "C"
