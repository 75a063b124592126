// pclseg_api.hip — the engine behind include/pclseg.h: weight folding/packing, workspace,
// kernel sequencing on one HIP stream, and the C ABI.  gfx950 only; there is no CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pclseg.h"
#include "pclseg_graph.h"
#ifndef PCLSEG_SLOTS
#define PCLSEG_SLOTS 2
#endif
#include "pclseg_kernels.h"

using namespace pclseg;

namespace {

thread_local std::string g_last_error;
// set when not even the message string could be built (exception barrier); it belongs to the handle (or NULL, for
// the handle-less entry points) whose call failed — pclseg_last_error of any OTHER handle still reads that handle's own text
thread_local const char* g_static_error = nullptr;
thread_local const pclseg_handle* g_static_handle = nullptr;

constexpr double kBnEps = 1e-3;  // Keras BatchNormalization default (no epsilon= in nets/*.py)

std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return std::string(buf);
}

}  // namespace

struct pclseg_handle {
  Graph g;
  std::vector<std::vector<float>> host_w;  // Keras tensors as set by the caller
  std::vector<char> is_set;
  bool finalized = false;
  int device = 0;
  hipStream_t stream = nullptr;
  float* d_w32 = nullptr;      // exact-f32 weight fragments (only with PCLSEG_FLAG_EXACT_F32)
  _Float16* d_w16 = nullptr;   // split-f16 weight fragments
  float* d_bias = nullptr;     // folded biases
  bool exact = false;
  // Micro-batches are independent, so they are dealt round-robin to `nlanes` lanes, each with
  // its own activation arena and HIP stream: kernels of different micro-batches overlap on the
  // GPU (a memory-bound kernel of one fills the idle pipes of a latency-bound kernel of another).
  static constexpr int kMaxLanes = 8;
  static constexpr int kSlotsPerLane = PCLSEG_SLOTS;
  int nlanes = 1;
  float* d_arena_lane[kMaxLanes] = {nullptr};
  uint8_t* d_mask_lane[kMaxLanes] = {nullptr};
  hipStream_t lane_stream[kMaxLanes] = {nullptr};
  hipEvent_t ev_in = nullptr, ev_lane[kMaxLanes] = {nullptr};
  float* d_arena = nullptr;   // arena of the lane that ran the LAST micro-batch (debug reads)
  // host-mode staging (grown on demand): per-lane device slabs of one micro-batch, a per-lane
  // pinned bounce slab for pageable inputs, full-size pinned bounce buffers for pageable outputs
  struct HostLane {
    void* d_in = nullptr; size_t in_bytes = 0;
    uint8_t* d_maskin = nullptr; size_t maskin_bytes = 0;
    int32_t* d_preds = nullptr; size_t preds_bytes = 0;
    float* d_probs = nullptr; size_t probs_bytes = 0;
    float* d_logits = nullptr; size_t logits_bytes = 0;
    void* p_in = nullptr; size_t p_in_bytes = 0;
    void* p_mask = nullptr; size_t p_mask_bytes = 0;
    // ev_in: the slot's H2D copies landed; ev_done: its kernels finished; ev_out: its D2H copies finished
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    bool out_dma = false;   // the slot's last micro-batch sent results through the D2H stream (ev_out is meaningful)
    bool used = false;
  } hl[kSlotsPerLane * kMaxLanes];            // two slots per lane: micro-batch k+1 of a lane uploads while k computes
  bool host_async_pending = false;   // PCLSEG_MEM_HOST_ASYNC calls enqueued since the last pclseg_sync
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;   // dedicated copy streams (SDMA engines run beside the kernels)
  hipEvent_t ev_copy_tail = nullptr;
  int32_t* p_preds = nullptr; size_t p_preds_bytes = 0;
  float* p_probs = nullptr; size_t p_probs_bytes = 0;
  float* p_logits = nullptr; size_t p_logits_bytes = 0;
  uint8_t* p_mask = nullptr; size_t p_mask_bytes = 0;
  int last_count = 0;  // scans held by the arena after the last forward
  bool last_exact = false;  // arithmetic of the last sweep (decides how debug reads decode tensors)
  // split-f16 range guard: device word OR-ed by any kernel that split a value with |v| >= 65504
  unsigned* d_range = nullptr;
  unsigned* h_range = nullptr;   // pinned host mirror
  bool fallback = false;         // PCLSEG_FLAG_RANGE_FALLBACK: both weight sets resident
  bool force_exact = false;      // fallback handle whose folded weights cannot be split (non-finite): every sweep is exact
  // The range flag is ONE sticky word shared by every queued call and every lane, so when it fires nobody
  // knows which of the calls enqueued since the last check overflowed: a fallback handle remembers all
  // of them and pclseg_sync (or the synchronous call that observes the flag) re-runs every one in exact
  // float32, in order.  Bounded: past kMaxPending un-checked calls the repair is refused (ERR_RANGE).
  struct PendingCall {
    const float* input = nullptr; bool raw = false; const uint8_t* mask_in = nullptr; int n = 0;
    int32_t* preds = nullptr; float* probs = nullptr; float* logits = nullptr; uint8_t* mask_out = nullptr;
    int mem = PCLSEG_MEM_DEVICE;
  };
  static constexpr size_t kMaxPending = 256;
  std::vector<PendingCall> pending;
  bool pending_overflow = false;
  int unchecked_calls = 0;       // asynchronous calls enqueued since the range flag was last read
  std::string err;
};

namespace {

int fail(pclseg_handle* h, int code, const std::string& msg) {
  if (h == g_static_handle) g_static_error = nullptr;
  g_last_error = msg;
  if (h) h->err = msg;
  return code;
}

// Restores the caller's current HIP device on every exit path (a process may hold engines on
// several GPUs, and torch's current device must not change behind its back).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, e_ == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,         \
                  fmt("%s failed: %s", #expr, hipGetErrorString(e_)));                     \
  } while (0)

// ---- BatchNorm folding + fragment packing -------------------------------------------------
struct FoldIn {
  const float* kernel = nullptr;  // Keras layout
  const float* bias = nullptr;
  const float* gamma = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
};

// BN(conv(x)+b) = conv(x)*scale + shift, evaluated in float64
void fold_bn(const SubOp& su, const FoldIn& f, std::vector<double>* scale, std::vector<double>* shift) {
  scale->assign(su.cout, 1.0);
  shift->assign(su.cout, 0.0);
  for (int co = 0; co < su.cout; ++co) {
    const double b = f.bias ? (double)f.bias[co] : 0.0;
    if (f.gamma) {
      const double s = (double)f.gamma[co] / std::sqrt((double)f.var[co] + kBnEps);
      (*scale)[co] = s;
      (*shift)[co] = (b - (double)f.mean[co]) * s + (double)f.beta[co];
    } else {
      (*shift)[co] = b;
    }
  }
}

// folded weight of window tap t, input channel ci, output channel co
inline double folded_w(const Op& op, const SubOp& su, const FoldIn& f, const std::vector<double>& scale,
                       int t, int ci, int co) {
  if (ci >= op.cin_k || co >= su.cout) return 0.0;
  const int kt = su.ktap[t];
  const double k = su.deconv ? f.kernel[((size_t)kt * su.cout + co) * op.cin_k + ci]   // (1,4,Cout,Cin)
                             : f.kernel[((size_t)kt * op.cin_k + ci) * su.cout + co];  // (kh,kw,Cin,Cout)
  return k * scale[co];
}

// bias block of a sub-op: [nctp*16] folded biases, then [nctp*16] inverse weight scales (1 until a
// split-f16 packer overwrites them; the exact-f32 kernels never read them)
void pack_bias(const SubOp& su, const std::vector<double>& shift, float* bdst) {
  for (int i = 0; i < su.nctp * 16; ++i) bdst[i] = i < su.cout ? (float)shift[i] : 0.0f;
  for (int i = 0; i < su.nctp * 16; ++i) bdst[su.nctp * 16 + i] = 1.0f;
}

// ---- split-f16 weights: power-of-two pre-scale per 16-output-channel tile --------------------
// w = hi + lo with hi = f16(w), lo = f16(w - hi) carries 22 significant bits only while lo is a NORMAL
// half: |lo| <= 2^-11 |w| drops below 2^-14 (subnormal, absolute step 2^-24) as soon as |w| < 2^-3 —
// which is every weight of a trained or He-initialised layer with a large fan-in.  The folded weights
// of each tile of 16 output channels (one MFMA A fragment row block: the granularity at which the
// kernels can undo a scale from a SCALAR register) are therefore multiplied by 2^k, k chosen so that
// the tile's largest magnitude lands in [2^12, 2^13), before the split; the float32 epilogue multiplies
// the accumulator by 2^-k (one fmaf with the bias: exact, free).  Elements down to 2^-15 of the tile
// maximum keep 22 bits; below that the ABSOLUTE error is 2^-25, i.e. 2^-37 of the tile's largest weight.
// (A per-channel exponent was measured first: its 2^-k quads cost 4 VGPRs per cout tile in every
// epilogue and 1.2 % of the step — fire6 31.3 -> 34.8 us, fire13 58.5 -> 65.5 us)
constexpr int kScaleTarget = 12;   // channel max -> [2^12, 2^13): hi <= 8192, far from the f16 limit
struct ScaleStat { bool nonfinite = false; };
int scale_exponent(double maxabs) {
  if (!(maxabs > 0.0) || !std::isfinite(maxabs)) return 0;
  int e;
  (void)std::frexp(maxabs, &e);                   // maxabs = m * 2^e, m in [0.5, 1)  ->  in [2^(e-1), 2^e)
  const int k = kScaleTarget + 1 - e;             // maxabs * 2^k in [2^12, 2^13)
  return std::max(-100, std::min(100, k));        // keeps 2^-k (and 2^k) normal float32 numbers
}
inline void split_store(double w, int k, _Float16* hi_dst, _Float16* lo_dst, ScaleStat* st) {
  const float ws = (float)std::ldexp(w, k);
  if (st && !std::isfinite(ws)) st->nonfinite = true;
  const _Float16 hi = (_Float16)ws;
  if (st && !std::isfinite((float)hi)) st->nonfinite = true;   // |ws| >= 65520 rounds to inf: hi = inf, lo = NaN
  *hi_dst = hi;
  *lo_dst = (_Float16)(ws - (float)hi);
}

// exact mode: w32[((t*nc16 + c16)*nctp + ct)*256 + lane*4 + j]
//   = W[tap t][ci = 16*c16 + 4*(lane>>4) + j][co = 16*ct + (lane&15)]
void pack_w32(const Op& op, const SubOp& su, const FoldIn& f, const std::vector<double>& scale, float* dst) {
  const int taps = su.nkh * su.nkw, nc16 = (op.cin_t + 15) / 16;
  for (int t = 0; t < taps; ++t)
    for (int c16 = 0; c16 < nc16; ++c16)
      for (int ct = 0; ct < su.nctp; ++ct) {
        float* blk = dst + ((size_t)(t * nc16 + c16) * su.nctp + ct) * 256;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j)
            blk[lane * 4 + j] = (float)folded_w(op, su, f, scale, t, c16 * 16 + 4 * (lane >> 4) + j,
                                                ct * 16 + (lane & 15));
      }
}

// split-f16 mode: per LDS chunk, K-step s and 16-cout tile ct one 2-KiB block [hi|lo][lane][8];
// lane group g = lane>>4 of step s owns the (tap, 8-channel group) pair kidx = 4s + g.
// `inv` = the sub-op's inverse-scale block (bias block + nctp*16), see cout scaling above.
void pack_w16(const Op& op, const SubOp& su, const FoldIn& f, const std::vector<double>& scale, _Float16* dst,
              float* inv, ScaleStat* st = nullptr) {
  const int taps = su.nkh * su.nkw;
  const int cin8 = (op.cin_t + 7) / 8, ck8_full = op.ck16 / 8;
  const int steps_full = f16_steps_full(op, su), nchunks = f16_chunks(op);
  std::vector<int> kexp((size_t)su.nctp * 16, 0);
  for (int ct = 0; ct < su.nctp; ++ct) {
    double m = 0.0;
    for (int co = ct * 16; co < std::min(su.cout, ct * 16 + 16); ++co)
      for (int t = 0; t < taps; ++t)
        for (int ci = 0; ci < op.cin_k; ++ci) {
          const double w = std::fabs(folded_w(op, su, f, scale, t, ci, co));
          if (!(w <= m)) m = w;   // (NaN propagates into m)
        }
    if (st && !std::isfinite(m)) st->nonfinite = true;
    for (int i = 0; i < 16; ++i) kexp[ct * 16 + i] = scale_exponent(m);
  }
  for (int i = 0; i < su.nctp * 16; ++i) inv[i] = (float)std::ldexp(1.0, -kexp[i]);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int ck8 = std::min(ck8_full, cin8 - chunk * ck8_full);
    const int nk = taps * ck8;
    for (int s = 0; s < steps_full; ++s)
      for (int ct = 0; ct < su.nctp; ++ct) {
        _Float16* blk = dst + ((size_t)(chunk * steps_full + s) * su.nctp + ct) * 1024;
        for (int lane = 0; lane < 64; ++lane) {
          const int kidx = 4 * s + (lane >> 4);
          const int co = ct * 16 + (lane & 15);
          for (int j = 0; j < 8; ++j) {
            double w = 0.0;
            if (kidx < nk) {
              const int tap = kidx / ck8, c8 = kidx % ck8;
              w = folded_w(op, su, f, scale, tap, (chunk * ck8_full + c8) * 8 + j, co);
            }
            split_store(w, kexp[co], &blk[lane * 8 + j], &blk[512 + lane * 8 + j], st);
          }
        }
      }
  }
}

// fused next-squeeze fragments (conv_kernel FSQ).  After a half (sidx: 0 = expand1x1, 1 = expand3x3)
// lane (p, g) of the wave that owns cout group cg holds, per pixel, channels 4g..4g+3 of its tiles
// 2st and 2st+1: k-slot (g, j) of partial-GEMM step st is channel
//   c = co_off(half) + (cg*ntw + 2st + (j >> 2))*16 + 4g + (j & 3)      (zero if that tile does not exist).
// A-fragment lane (r = lane & 15, g = lane >> 4), element j = BN-folded squeeze weight W[c][16qt + r],
// scaled per 16-channel tile of q like every split-f16 weight (`inv`: the squeeze's inverse-scale block).
void pack_fsq(const Op& op, const FoldIn& f, const std::vector<double>& scale, _Float16* dst, float* inv,
              ScaleStat* st = nullptr) {
  const SubOp& sq = op.fsq;
  const int ncg = op.sub[0].nctp / op.ntw, ns = (op.ntw + 1) / 2, nq = sq.nctp;
  const int cx = op.sub[0].cout + op.sub[1].cout;   // channels of the (never materialised) pair output
  std::vector<int> kexp((size_t)nq * 16, 0);
  for (int qt = 0; qt < nq; ++qt) {
    double m = 0.0;
    for (int q = qt * 16; q < std::min(sq.cout, qt * 16 + 16); ++q)
      for (int c = 0; c < cx; ++c) {
        const double w = std::fabs((double)f.kernel[(size_t)c * sq.cout + q] * scale[q]);
        if (!(w <= m)) m = w;
      }
    if (st && !std::isfinite(m)) st->nonfinite = true;
    for (int i = 0; i < 16; ++i) kexp[qt * 16 + i] = scale_exponent(m);
  }
  for (int i = 0; i < nq * 16; ++i) inv[i] = (float)std::ldexp(1.0, -kexp[i]);
  for (int half = 0; half < 2; ++half)
    for (int cg = 0; cg < ncg; ++cg)
      for (int st_ = 0; st_ < ns; ++st_)
        for (int qt = 0; qt < nq; ++qt) {
          _Float16* blk = dst + ((((size_t)half * ncg + cg) * ns + st_) * nq + qt) * 1024;
          for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, gg = lane >> 4, q = qt * 16 + r;
            for (int j = 0; j < 8; ++j) {
              const int nn = 2 * st_ + (j >> 2);
              const int cl = (cg * op.ntw + nn) * 16 + 4 * gg + (j & 3);   // channel within the half
              double w = 0.0;
              if (nn < op.ntw && cl < op.sub[half].cout && q < sq.cout) {
                const int c = op.sub[half].co_off + cl;
                if (c < cx) w = (double)f.kernel[(size_t)c * sq.cout + q] * scale[q];
              }
              split_store(w, kexp[q], &blk[lane * 8 + j], &blk[512 + lane * 8 + j], st);
            }
          }
        }
}

// ---- launch helpers -------------------------------------------------------------------------
// Dynamic LDS beyond the 64 KiB default needs hipFuncAttributeMaxDynamicSharedMemorySize, which is a
// property of (function, DEVICE): a process may hold engines on several GPUs, so the "already raised"
// memo is kept per device.
hipError_t raise_lds_limit(const void* fn, size_t lds) {
  if (lds <= 64 * 1024) return hipSuccess;
  static std::mutex m;
  static std::set<std::pair<const void*, int>> raised;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev)) return e;
  std::lock_guard<std::mutex> lk(m);
  if (raised.count({fn, dev})) return hipSuccess;
  if (hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
  raised.insert({fn, dev});
  return hipSuccess;
}

template <int MTW, int NTW, int WN, bool HEAD, bool F16, bool PAIR = false, int NW = 4>
hipError_t launch_conv_epi(int epi, dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
#define PCLSEG_GO(EPI_) \
  hipLaunchKernelGGL((conv_kernel<MTW, NTW, WN, HEAD, F16, EPI_, PAIR, NW>), grid, dim3(NW * 64), lds, s, a)
  if constexpr (HEAD) { PCLSEG_GO(0); }
#ifdef PCLSEG_CAND_EXACTEPI
  else if constexpr (!F16) {   // exact mode: the two common epilogues get their own instantiation (the catch-all carries
    switch (epi) {             // registers for every optional operand and spills 260-470 B in the 4-segment shapes)
      case 0: PCLSEG_GO(0); break;
      case 1: PCLSEG_GO(1); break;
      default: PCLSEG_GO(4); break;
    }
  }
#else
  else if constexpr (!F16) { PCLSEG_GO(4); }  // exact mode: one catch-all instantiation
#endif
  else if constexpr (PAIR) {                  // FIRE expand pairs: plain, + skip add, + fused skip branch
    switch (epi) {
      case 0: PCLSEG_GO(0); break;
      case 1: PCLSEG_GO(1); break;
      case 3: if constexpr (NW == 8 && NTW == 1) { PCLSEG_GO(3); break; } else return hipErrorInvalidValue;
      default: return hipErrorInvalidValue;
    }
  } else {
    switch (epi) {
      case 0: PCLSEG_GO(0); break;
      case 1: PCLSEG_GO(1); break;
      case 2: PCLSEG_GO(2); break;
#ifdef PCLSEG_TUNING   // a fused skip branch outside a merged pair: only a tuning switch (PCLSEG_BIGTILE=0, PCLSEG_GEOM) gets here
      case 3: PCLSEG_GO(3); break;
      default: PCLSEG_GO(4); break;
#else
      default: return hipErrorInvalidValue;
#endif
    }
  }
#undef PCLSEG_GO
  return hipGetLastError();
}

template <bool HEAD, bool F16>
hipError_t launch_conv_cfg(int mtw, int ntw, int wn, int epi, dim3 grid, size_t lds, hipStream_t s,
                           const ConvArgs& a) {
#ifdef PCLSEG_TUNING   // every tile shape: PCLSEG_GEOM=<layer>=ntw,wn,mtw picks any of them
#define PCLSEG_LAUNCH(NTW_, WN_) \
  return mtw == 2 ? launch_conv_epi<2, NTW_, WN_, HEAD, F16>(epi, grid, lds, s, a) \
                  : launch_conv_epi<4, NTW_, WN_, HEAD, F16>(epi, grid, lds, s, a)
  if (wn == 2 && ntw == 2 && !HEAD) { PCLSEG_LAUNCH(2, 2); }
  else if (wn == 1 && ntw == 1) { PCLSEG_LAUNCH(1, 1); }
  else if (wn == 1 && ntw == 2) { PCLSEG_LAUNCH(2, 1); }
  else if (wn == 1 && ntw == 3) { PCLSEG_LAUNCH(3, 1); }
  else if (wn == 1 && ntw == 4 && HEAD) { PCLSEG_LAUNCH(4, 1); }
#undef PCLSEG_LAUNCH
#else
  // The shapes op_geometry (pclseg_graph.h) produces, and no others — a kernel that no plan can select is not built
  // (profiles/r05_sim_kernel_coverage.txt listed 96 such instantiations, the ones with the largest spills among them):
  //   WN = 2 or one cout tile: 4 segments per wave; WN = 1 with 2-3 tiles: 2 segments for a single sub-conv (and the
  //   head, whose tile count is ceil(NUM_CLASS / 16) <= 4), 4 for two sub-convs (transposed convolution, un-merged pair).
#define PCLSEG_LAUNCH(MTW_, NTW_, WN_) return launch_conv_epi<MTW_, NTW_, WN_, HEAD, F16>(epi, grid, lds, s, a)
  if constexpr (!HEAD) { if (wn == 2 && ntw == 2 && mtw == 4) { PCLSEG_LAUNCH(4, 2, 2); } }
  if (wn == 1 && ntw == 1 && mtw == 4) { PCLSEG_LAUNCH(4, 1, 1); }
  if (wn == 1 && ntw == 2 && mtw == 2) { PCLSEG_LAUNCH(2, 2, 1); }
  if (wn == 1 && ntw == 3 && mtw == 2) { PCLSEG_LAUNCH(2, 3, 1); }
  if constexpr (HEAD) { if (wn == 1 && ntw == 4 && mtw == 2) { PCLSEG_LAUNCH(2, 4, 1); } }
  if constexpr (!HEAD) {
    if (wn == 1 && ntw == 2 && mtw == 4) { PCLSEG_LAUNCH(4, 2, 1); }
    if (wn == 1 && ntw == 3 && mtw == 4) { PCLSEG_LAUNCH(4, 3, 1); }
  }
#undef PCLSEG_LAUNCH
#endif
  return hipErrorInvalidValue;
}

// merged FIRE expand pair (split-f16 mode): the block shapes the reference's FIRE sizes use
// (pclseg_graph.h: pair_geometry); (mtw, ntw, wn, nw)
#define PCLSEG_PAIR_CFGS(X) X(4, 2, 2, 4) X(4, 2, 1, 4) X(8, 2, 8, 8) X(4, 3, 4, 8) X(4, 2, 4, 8) X(4, 2, 2, 8) X(4, 2, 8, 8) X(2, 3, 4, 8) X(4, 1, 8, 8) X(4, 1, 2, 8)
inline bool pair_cfg_ok(const Op& op) {
  if (op.sub[0].nctp != op.sub[1].nctp || op.ck16 < op.cin_t) return false;
  if (op.fsq_fused) return true;   // its shapes are PCLSEG_FSQ_CFGS (launch_conv_fsq)
#define PCLSEG_X(M_, N_, W_, NW_) if (op.mtw == M_ && op.ntw == N_ && op.wn == W_ && op.nw == NW_) return true;
  PCLSEG_PAIR_CFGS(PCLSEG_X)
#undef PCLSEG_X
  return false;
}
// merged pair + fused next squeeze, 8 waves: (mtw, ntw, wn, nq, epi) of fire8/9 (256+256 -> 64), fire6
// (192+192 -> 48), fire7 (192+192 -> 64), fire4 (128+128 -> 32) on 64-pixel tiles, and the FIREUP pairs
// with their skip add: fire10 (128+128 -> 32, 64 px), fire11 (64+64 -> 16, 128 px), fire12 (32+32 -> 16, 256 px)
#define PCLSEG_FSQ_CFGS(X) X(4, 2, 8, 4, 0) X(2, 3, 4, 3, 0) X(2, 3, 4, 4, 0) X(4, 1, 8, 2, 0) \
                           X(4, 1, 8, 2, 1) X(4, 1, 4, 1, 1) X(4, 1, 2, 1, 1)
hipError_t launch_conv_fsq(const Op& op, int epi, dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
  // fire8 / fire9 (64 -> 256 + 256 -> 64, 4 x 16-pixel tiles): the compile-time geometry of conv_kernel GEOM 1
  static const int geom_on = tune_env("PCLSEG_FSQ_GEOM", 1);
  if (geom_on && op.mtw == 4 && op.ntw == 2 && op.wn == 8 && a.fsq_q == 64 && epi == 0 && op.cin_t == 64 && a.in_s16 &&
      a.PW == 18 && a.PH == 6 && op.ck16 == 64 && !op.up_fused) {
    auto kfn = conv_kernel<4, 2, 8, false, true, 0, true, 8, 4, 0, 0, false, 1>;
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e;
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a);
    return hipGetLastError();
  }
#ifdef PCLSEG_CAND_GEOM2
  // fire4 (32 -> 128 + 128 -> 32, 4 x 16-pixel tiles, WN = 8): conv_kernel GEOM 2
  if (op.mtw == 4 && op.ntw == 1 && op.wn == 8 && a.fsq_q == 32 && epi == 0 && op.cin_t == 32 && a.in_s16 && a.PW == 18 && a.PH == 6 &&
      op.ck16 >= 32 && !op.up_fused) {
    auto kfn = conv_kernel<4, 1, 8, false, true, 0, true, 8, 2, 0, 0, false, 2>;
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e;
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a);
    return hipGetLastError();
  }
#endif
#define PCLSEG_X(M_, N_, W_, Q_, E_) \
  if (op.mtw == M_ && op.ntw == N_ && op.wn == W_ && a.fsq_q == Q_ * 16 && epi == E_) { \
    auto kfn = conv_kernel<M_, N_, W_, false, true, E_, true, 8, Q_>; \
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e; \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a); \
    return hipGetLastError(); \
  }
  PCLSEG_FSQ_CFGS(PCLSEG_X)
#undef PCLSEG_X
  return hipErrorInvalidValue;
}

// merged FIREUP pair that up-convolves its own patch (conv_kernel UP): fire10 / fire11 / fire12 (with their
// fused next squeeze and skip add) and fire13 (fused skip branch); (mtw, ntw, wn, nq, epi, up = C/16)
#define PCLSEG_UP_CFGS(X) X(4, 1, 8, 2, 1, 4) X(4, 1, 4, 1, 1, 2) X(4, 1, 2, 1, 1, 1) X(4, 1, 2, 0, 3, 1)
hipError_t launch_conv_up(const Op& op, int epi, dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
  const int nq = op.fsq_fused ? a.fsq_q / 16 : 0;
  // fire10 (64-channel up-convolved patch, 4 x 16-pixel tiles): the compile-time geometry K loop (conv_kernel GEOM 1) —
  // with ONE cout tile per wave a K-step is only 12 MFMAs, so the generic loop's address arithmetic weighs most here
  static const int geom_on = tune_env("PCLSEG_UP_GEOM", 1);
  if (geom_on && op.mtw == 4 && op.ntw == 1 && op.wn == 8 && op.nw == 8 && nq == 2 && epi == 1 && op.cin_t == 64 &&
      a.PW == 18 && a.PH == 6 && op.ck16 == 64) {
    auto kfn = conv_kernel<4, 1, 8, false, true, 1, true, 8, 2, 4, 0, false, 1>;
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e;
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a);
    return hipGetLastError();
  }
#ifdef PCLSEG_CAND_GEOM2
  // fire11 (32-channel up-convolved patch, 8 x 16-pixel tiles, WN = 4): conv_kernel GEOM 2
  if (op.mtw == 4 && op.ntw == 1 && op.wn == 4 && op.nw == 8 && nq == 1 && epi == 1 && op.cin_t == 32 && a.PW == 18 && a.PH == 10 &&
      op.ck16 >= 32) {   // (one chunk: 40 halfs per staged pixel and plane)
    auto kfn = conv_kernel<4, 1, 4, false, true, 1, true, 8, 1, 2, 0, false, 2>;
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e;
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a);
    return hipGetLastError();
  }
#endif
#define PCLSEG_X(M_, N_, W_, Q_, E_, U_) \
  if (op.mtw == M_ && op.ntw == N_ && op.wn == W_ && op.nw == 8 && nq == Q_ && epi == E_ && op.cin_t == 16 * U_) { \
    auto kfn = conv_kernel<M_, N_, W_, false, true, E_, true, 8, Q_, U_>; \
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds)) return e; \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, s, a); \
    return hipGetLastError(); \
  }
  PCLSEG_UP_CFGS(PCLSEG_X)
#undef PCLSEG_X
  return hipErrorInvalidValue;
}

hipError_t launch_conv_pair(const Op& op, int epi, dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
#define PCLSEG_X(M_, N_, W_, NW_) \
  if (op.mtw == M_ && op.ntw == N_ && op.wn == W_ && op.nw == NW_) \
    return launch_conv_epi<M_, N_, W_, false, true, true, NW_>(epi, grid, lds, s, a);
  PCLSEG_PAIR_CFGS(PCLSEG_X)
#undef PCLSEG_X
  return hipErrorInvalidValue;
}

#ifdef PCLSEG_WITH_STAMPS
// Debug build (make stamps): on the 20th launch of the merged-pair kernel whose layer name contains
// $PCLSEG_STAMP, thread 0 of every block stamps s_memtime at the phase boundaries; the destructor (after
// the launch) prints the mean per-block duration of each phase in shader cycles.
struct StampDump {
  unsigned long long* buf = nullptr;
  unsigned blocks = 0;
  hipStream_t s;
  std::string name;
  const char* const* labels = nullptr;
  StampDump(const std::string& opname, unsigned long long** field, dim3 grid, hipStream_t s_, const char* const* labels_ = nullptr)
      : s(s_), labels(labels_) {
    static const char* want = getenv("PCLSEG_STAMP");
    static unsigned long long* dbuf = nullptr;
    static int count = 0;
    *field = nullptr;
    if (!want || opname.find(want) == std::string::npos || grid.x > 16384) return;
    if (!dbuf) (void)hipMalloc((void**)&dbuf, (size_t)16384 * 8 * 8);
    if (++count != 20) return;
    (void)hipMemsetAsync(dbuf, 0, (size_t)grid.x * 64, s);
    *field = buf = dbuf;
    blocks = grid.x;
    name = opname;
  }
  ~StampDump() {
    if (!buf) return;
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h((size_t)blocks * 8);
    (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
    // merged pairs: staged | K | epilogue or partial squeeze | K | epilogue/partial | - | slab phase;
    // other ops: staged chunk 0 | K chunk 0 | staged chunk 1 | K chunk 1 | (later chunks) | epilogue | end
    static const char* nm[8] = {"entry", "staged", "K (first)", "epilogue 1 / staged 2", "K (second)", "epilogue 2 / more chunks",
                                "loop exit / epilogue", "slab+reduce+store / end"};
    double sum[8] = {0};
    for (unsigned b = 0; b < blocks; ++b)
      for (int i = 1; i < 8; ++i) sum[i] += (double)(long long)(h[b * 8 + i] - h[b * 8 + i - 1]);
    double tot = 0;
    for (int i = 1; i < 8; ++i) tot += sum[i] / blocks;
    fprintf(stderr, "STAMPS %s: %u blocks, %.0f cycles per block\n", name.c_str(), blocks, tot);
    for (int i = 1; i < 8; ++i) fprintf(stderr, "  %-26s %8.0f cycles  %5.1f %%\n", (labels ? labels : nm)[i], sum[i] / blocks, 100.0 * sum[i] / blocks / tot);
  }
};
#endif

#ifdef PCLSEG_CAND_WIDE
// Darknet's wide 1x1 layers (BasicBlock / decoder-block conv1, float32 input, no fused operands) run on
// conv1x1_wide_kernel: launch_conv and pclseg_plan_ops both ask here, and tests/test_sim_only.py compares the plan
// with the launches the simulator sees.
bool op_is_wide_1x1(const Op& op, bool in_s16, bool has_residual, bool has_skip) {
  return op_is_flat(op) && op.nsub == 1 && op.kind == OP_CONV && op.ck16 == 64 && op.cin_t % 64 == 0 && op.cin_t >= 128 &&
         op.sub[0].nctp % 8 == 0 && op.sub[0].cout == op.sub[0].nctp * 16 && !in_s16 && !has_residual && !has_skip;
}
// -> cout tiles per wave (1 | 2) and blocks along the cout axis
void wide_1x1_geom(const Op& op, int* nt, int* ny) {
  *nt = op.sub[0].nctp % 16 == 0 ? 2 : 1;
  *ny = op.sub[0].nctp / (8 * *nt);
}
#endif

// Fill the geometry of `a` (tensor pointers already set) from `op` and launch.
// w32 / w16 / bias are the bases the sub-op offsets are relative to.
hipError_t launch_conv(const Op& op_in, int N, int H, int Win, ConvArgs a, const float* w32,
                       const _Float16* w16, const float* bias, bool exact, hipStream_t s) {
  Op op = op_in;
  if (exact && op.nw != 4) {
    // exact-f32 sweep of a graph planned for split-f16 (PCLSEG_FLAG_RANGE_FALLBACK): the 8-wave
    // block shapes exist only for the split-f16 kernels; any 4-wave shape whose cout group divides
    // the packed tile count reads the same fragments
    const int tiles = op.sub[0].nctp;   // (both halves of a pair carry the same count)
    op.nw = 4; op.mtw = 4;
    if (tiles % 4 == 0) { op.ntw = 2; op.wn = 2; }
    else if (tiles % 2 == 0) { op.ntw = 2; op.wn = 1; }
    else { op.ntw = 1; op.wn = 1; }
    op.ck32 = 32;
    while (op.ck32 > 16 && lds_bytes_f32(op, op.ck32) > 64 * 1024) op.ck32 /= 2;
  }
  a.Cin = op.cin_t;
  a.res1_mul = op.res1_mul ? 1 : 0;
  a.nsub = op.nsub;
  a.CK = exact ? op.ck32 : op.ck16;
  int ny = 0;
  for (int i = 0; i < op.nsub; ++i) {
    const SubOp& su = op.sub[i];
    ConvSub& d = a.sub[i];
    d.w32 = w32 ? w32 + su.w32_off : nullptr;
    d.w16 = w16 ? w16 + su.w16_off : nullptr;
    d.bias = bias + su.b_off;
    d.Cout = su.cout;
    d.nctp = su.nctp;
    d.ny = su.nctp / (op.ntw * op.wn);
    d.co_off = su.co_off;
    d.th0 = su.th0; d.tw0 = su.tw0; d.nkh = su.nkh; d.nkw = su.nkw;
    d.ow_off = su.ow_off;
    d.act = su.act;
    ny += d.ny;
  }
  a.ny = ny;
  a.group_major = 0;
  int wo, pl, ho, pt;
  same_pad(Win, op.pkw, op.sw, &wo, &pl);
  same_pad(H, op.pkh, 1, &ho, &pt);
  a.sw = op.sw; a.pt = pt; a.pl = op.pl_fixed >= 0 ? op.pl_fixed : pl;
  a.ow_mul = op.ow_mul;
  if (op_is_flat(op)) {
    a.N = 1; a.H = 1; a.Win = N * H * Win; a.Wconv = a.Wout = a.Win;
  } else {
    a.N = N; a.H = H; a.Win = Win;
    a.Wconv = op.ow_mul == 2 ? Win : wo;
    a.Wout = op.ow_mul == 2 ? 2 * Win : wo;
  }
  const TileGeom t = tile_geom(op);
  a.TH = t.TH; a.SEGW = t.SEGW; a.PH = t.PH; a.PW = t.PW;
  a.inv_pw = ((1 << 20) + a.PW - 1) / a.PW;
  a.tilesH = (a.H + a.TH - 1) / a.TH;
  a.tilesW = (a.Wconv + a.SEGW * 16 - 1) / (a.SEGW * 16);
  static const int use_direct = tune_env("PCLSEG_DIRECT1X1", 1);
  if (use_direct && !exact && op_is_flat(op) && op.nsub == 1 && op.kind == OP_CONV && op.cin_t % 8 == 0 &&
      op.ck16 >= 32 && !a.skx && !a.res2 && !a.in_s16 && op.sub[0].nctp <= 4) {  // squeeze-like: few couts
    // LDS-free streaming GEMM; every wave owns mtw*16 pixels x ntw*16 couts
    // one block column covers ALL couts (nctp <= 4 tiles), so the input is read exactly once
    const int ntw = op.sub[0].nctp;
    static const int splitk_min = tune_env("PCLSEG_SPLITK", 256);
    if (splitk_min > 0 && op.cin_t >= splitk_min && ntw >= 3) {
      // deep squeezes (256..512 channels -> 48/64, 64x128 pixels): the block's four waves split
      // the channels (conv1x1_direct_kernel, SPLITK): fire6/7 14.6/18.1 -> 10.6/13.7 us,
      // fire9/10 22 -> 20 us; no gain for the 32-cout layers
      const int mtw = 2;
      dim3 grid((unsigned)((a.Win + mtw * 16 - 1) / (mtw * 16)), 1u);
      const size_t lds = (size_t)4 * mtw * ntw * 1024;
      const bool res = a.res1 != nullptr;
#define PCLSEG_SK(MTW_, NTW_) \
      do { if (res) hipLaunchKernelGGL((conv1x1_direct_kernel<MTW_, NTW_, true, true>), grid, dim3(kConvThreads), lds, s, a); \
           else hipLaunchKernelGGL((conv1x1_direct_kernel<MTW_, NTW_, false, true>), grid, dim3(kConvThreads), lds, s, a); } while (0)
      switch (ntw) {
        case 3: PCLSEG_SK(2, 3); break;
        default: PCLSEG_SK(2, 4); break;
      }
#undef PCLSEG_SK
      return hipGetLastError();
    }
    const int mtw = ntw == 4 ? 1 : 2;  // 4 cout tiles x 2 segments would spill at 128 VGPRs
    const int px_per_block = 4 * mtw * 16;
    dim3 grid((unsigned)((a.Win + px_per_block - 1) / px_per_block), 1u);
    const bool res = a.res1 != nullptr;
#define PCLSEG_D(MTW_, NTW_) \
    do { if (res) hipLaunchKernelGGL((conv1x1_direct_kernel<MTW_, NTW_, true>), grid, dim3(kConvThreads), 0, s, a); \
         else hipLaunchKernelGGL((conv1x1_direct_kernel<MTW_, NTW_, false>), grid, dim3(kConvThreads), 0, s, a); } while (0)
    switch (ntw) {
      case 1: PCLSEG_D(2, 1); break;
      case 2: PCLSEG_D(2, 2); break;
      case 3: PCLSEG_D(2, 3); break;
      default: PCLSEG_D(1, 4); break;
    }
#undef PCLSEG_D
    return hipGetLastError();
  }
  size_t lds = (size_t)(exact ? lds_bytes_f32(op, op.ck32) : lds_bytes_f16(op, op.ck16));
  lds = (lds + 15) & ~(size_t)15;
  a.skw_lds_off = (int)lds;
  if (a.skx) lds += (size_t)9 * a.out_C * sizeof(float);   // fused skip branch weights behind the patch
  if (op.up_fused) {   // half-width source patch of the fused up-convolution behind that: PH x (PW/2 + 1) pixels x [hi C | lo C | pad]
    lds = (lds + 15) & ~(size_t)15;
    a.up_lds_off = (int)lds;
    lds += (size_t)a.PH * (a.PW / 2 + 1) * (2 * op.cin_t + kPadF16) * sizeof(_Float16);
  }
  if (op.fsq_fused) {   // the partial-sum slab [8 waves][mtw*16 px][Q] float32 reuses the patch's LDS
    if (exact) return hipErrorInvalidValue;
    size_t slab = (size_t)8 * op.mtw * 16 * (op.fsq.nctp * 16 + 4) * sizeof(float);   // rows padded by 4 floats
#ifdef PCLSEG_CAND_SLAB
    if (op.fsq.nctp == 4 && slab > 96 * 1024) slab = (size_t)8 * op.mtw * 16 * (2 * 16 + 4) * sizeof(float);   // two passes (conv_kernel NPASS)
#endif
    lds = std::max(lds, slab);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
  } else if (lds > 64 * 1024) return hipErrorInvalidValue;
  // (the fused-skip-branch epilogue of fire13 needs more registers than the merged kernel has left)
  const bool pair = !exact && op.pair && pair_cfg_ok(op) && (!a.skx || (op.nw == 8 && op.ntw == 1 && !a.res1)) && !a.res2;
  if (pair) {  // one block = cout group of the 3x3 half + the same group of the 1x1 half
    ny = a.sub[1].ny;
    a.ny = ny;
  }
  {
    // A layer whose packed weights fit an XCD's 4 MB L2 keeps them there under any order, so the
    // order is spent on the input patches (tile-major).  Darknet's deep layers (up to 19 MB of
    // fragments) would re-stream the weights from the Infinity Cache for every pixel tile:
    // group-major makes the blocks resident on an XCD read the same fragments together.
    double wbytes = 0;
    for (int i = 0; i < op.nsub; ++i)
      wbytes += (double)op.sub[i].nkh * op.sub[i].nkw * op.cin_t * op.sub[i].nctp * 16 * 4.0;
    static const int gm = tune_env("PCLSEG_GROUP_MAJOR", -1);
    a.group_major = gm >= 0 ? gm : (ny > 1 && wbytes > 2.0 * 1024 * 1024);
  }
#ifdef PCLSEG_CAND_WIDE
  // Darknet's wide 1x1 layers (BasicBlock / decoder-block conv1): the software-pipelined GEMM kernel
  if (!exact && w16 && op_is_wide_1x1(op, a.in_s16 != 0, a.res1 || a.res2, a.skx != nullptr)) {
    static const int wide_on = tune_env("PCLSEG_WIDE1X1", 1);
    if (wide_on) {
      int nt, ny_w;
      wide_1x1_geom(op, &nt, &ny_w);
      a.ny = ny_w;
      a.sub[0].ny = ny_w;
      if (ny_w == 1) a.group_major = 0;
      const dim3 wgrid((unsigned)(((a.Win + kW1Px - 1) / kW1Px) * ny_w));
      if (nt == 2) {
        if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&conv1x1_wide_kernel<2>), kW1Lds)) return e;
        hipLaunchKernelGGL((conv1x1_wide_kernel<2>), wgrid, dim3(512), kW1Lds, s, a);
      } else {
        if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&conv1x1_wide_kernel<1>), kW1Lds)) return e;
        hipLaunchKernelGGL((conv1x1_wide_kernel<1>), wgrid, dim3(512), kW1Lds, s, a);
      }
      return hipGetLastError();
    }
  }
#endif
  dim3 grid((unsigned)(a.N * a.tilesH * a.tilesW * ny));
#ifdef PCLSEG_WITH_STAMPS
  StampDump stamp_dump(op.name(), &a.stamps, grid, s);   // debug build: PCLSEG_STAMP=<layer name> prints its phase split
#endif
  const int epi = a.skx ? (a.res1 || a.res2 ? 4 : 3) : a.res2 ? 2 : a.res1 ? 1 : 0;
  if (op.up_fused) {   // `a.in` is the half-width squeeze tensor; Win is the up-convolved width
    if (!pair || !a.in_s16 || !w16 || (Win & 1)) return hipErrorInvalidValue;
    a.up_w16[0] = w16 + op.up[0].w16_off;
    a.up_w16[1] = w16 + op.up[1].w16_off;
    a.up_bias = bias + op.up[0].b_off;
    a.up_Win = Win / 2;
    a.up_nctp = op.up[0].nctp;
  }
  if (op.fsq_fused) {
    if (!pair) return hipErrorInvalidValue;
    a.fsq_w16 = w16 + op.fsq.w16_off;
    a.fsq_bias = bias + op.fsq.b_off;
    a.fsq_q = op.fsq.nctp * 16;
    a.fsq_ncg = op.sub[0].nctp / op.ntw;
    a.flip_bit = -1;
    return op.up_fused ? launch_conv_up(op, epi, grid, lds, s, a) : launch_conv_fsq(op, epi, grid, lds, s, a);
  }

  if (pair) {
    static const int wt = tune_env("PCLSEG_WT", 1);
    // half of the blocks of an 8-wave pair take the 1x1 half first (fire8/9/10: -1.3 .. -3.4 us; the
    // 4-wave pairs measured neutral to worse)
    static const int flip = tune_env("PCLSEG_FLIP", 3);
    a.flip_bit = op.nw == 8 ? flip : -1;
    a.wt = (wt && !a.res1 && op.nw == 4) ? 1 : 0;   // pays for the 4-wave pairs that only write (see store_quad)
    return op.up_fused ? launch_conv_up(op, epi, grid, lds, s, a) : launch_conv_pair(op, epi, grid, lds, s, a);
  }
  if (op.up_fused) return hipErrorInvalidValue;
  if (op.nw == 8) {   // Darknet's wide layers: 128 px x 128 couts, 8 waves (op_geometry)
    if (exact || op.kind == OP_HEAD || op.ntw != 2) return hipErrorInvalidValue;
    if (op.mtw == 8 && op.wn == 8 && epi <= 2) {
      // 128 px x 256 couts: the register-capped variant (two blocks per CU).  Tuning build: PCLSEG_DN_VARIANT = 0 the
      // uncapped kernel (147-226 registers, one block per CU), 1 capped, 2 loader waves (conv_kernel LW; measured: no gain)
      static const int variant = tune_env("PCLSEG_DN_VARIANT", 1);
#define PCLSEG_CAP(E_, G_) hipLaunchKernelGGL((conv_kernel<8, 2, 8, false, true, E_, false, 8, 0, 0, 0, true, G_>), grid, dim3(512), lds, s, a)
      if (variant == 1) {
        // 3x3, stride 1, every chunk a full 64 channels: the fully unrolled K loop with immediate addressing (GEOM 1)
        static const int geom_on = tune_env("PCLSEG_DN_GEOM", 1);
        const bool g1 = geom_on && op.pkh == 3 && op.pkw == 3 && op.sw == 1 && op.nsub == 1 && op.ck16 == 64 && op.cin_t % 64 == 0 &&
                        a.PW == 18 && a.PH == 10 && op.sub[0].nkh == 3 && op.sub[0].nkw == 3 && op.sub[0].th0 == 0 && op.sub[0].tw0 == 0;
        if (g1) { switch (epi) { case 0: PCLSEG_CAP(0, 1); break; case 1: PCLSEG_CAP(1, 1); break; default: PCLSEG_CAP(2, 1); break; } }
        else { switch (epi) { case 0: PCLSEG_CAP(0, 0); break; case 1: PCLSEG_CAP(1, 0); break; default: PCLSEG_CAP(2, 0); break; } }
        return hipGetLastError();
      }
#undef PCLSEG_CAP
#ifdef PCLSEG_TUNING
      const int cin8 = (op.cin_t + 7) / 8, ck8_full = op.ck16 / 8;
      if (variant == 2 && cin8 > ck8_full && epi <= 1 && !a.skx) {
        const size_t lds2 = 2 * ((lds + 15) & ~(size_t)15);
        if (lds2 <= 160 * 1024) {
          if (epi == 0) {
            auto kfn = conv_kernel<8, 2, 8, false, true, 0, false, 8, 0, 0, 4>;
            if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds2)) return e;
            hipLaunchKernelGGL(kfn, grid, dim3(12 * 64), lds2, s, a);
          } else {
            auto kfn = conv_kernel<8, 2, 8, false, true, 1, false, 8, 0, 0, 4>;
            if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn), lds2)) return e;
            hipLaunchKernelGGL(kfn, grid, dim3(12 * 64), lds2, s, a);
          }
          return hipGetLastError();
        }
      }
#endif
    }
    if (op.mtw == 4 && op.wn == 4) return launch_conv_epi<4, 2, 4, false, true, false, 8>(epi, grid, lds, s, a);
#ifdef PCLSEG_TUNING   // (the uncapped 128 px x 256 couts kernel: PCLSEG_DN_VARIANT=0)
    if (op.mtw == 8 && op.wn == 8) return launch_conv_epi<8, 2, 8, false, true, false, 8>(epi, grid, lds, s, a);
#endif
    return hipErrorInvalidValue;
  }
  if (op.kind == OP_HEAD)
    return exact ? launch_conv_cfg<true, false>(op.mtw, op.ntw, op.wn, epi, grid, lds, s, a)
                 : launch_conv_cfg<true, true>(op.mtw, op.ntw, op.wn, epi, grid, lds, s, a);
  return exact ? launch_conv_cfg<false, false>(op.mtw, op.ntw, op.wn, epi, grid, lds, s, a)
               : launch_conv_cfg<false, true>(op.mtw, op.ntw, op.wn, epi, grid, lds, s, a);
}

// 3x3 s(1,2) max-pool + 1x1 squeeze in one pass (Op::pool_fused); `Win` is the width BEFORE the pool.
template <int NTW>
hipError_t launch_pool_squeeze_n(dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
  if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&pool_squeeze_kernel<NTW>), lds)) return e;
  hipLaunchKernelGGL((pool_squeeze_kernel<NTW>), grid, dim3(kConvThreads), lds, s, a);
  return hipGetLastError();
}
hipError_t launch_pool_squeeze(const Op& op, int N, int H, int Win, ConvArgs a, const _Float16* w16,
                               const float* bias, hipStream_t s) {
  const SubOp& su = op.sub[0];
  const int C = op.cin_t;
  if (!w16 || op.nsub != 1 || C % 32 != 0 || (C & (C - 1)) != 0 || su.nctp < 1 || su.nctp > 4 || op.ck16 < 32)
    return hipErrorInvalidValue;
  int wo, pl;
  same_pad(Win, 3, 2, &wo, &pl);
  a.N = N; a.H = H; a.Win = Win; a.Wout = a.Wconv = wo; a.pl = pl;
  a.Cin = C;
  a.nsub = 1;
  ConvSub& d = a.sub[0];
  d.w32 = nullptr;
  d.w16 = w16 + su.w16_off;
  d.bias = bias + su.b_off;
  d.Cout = su.cout; d.nctp = su.nctp; d.ny = 1; d.co_off = su.co_off; d.act = su.act;
  const dim3 grid((unsigned)(N * ((H + kPoolSqRows - 1) / kPoolSqRows) * ((wo + 15) / 16)));
  const size_t lds = (size_t)kPoolSqRows * 16 * (2 * C + kPadF16) * sizeof(_Float16);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  switch (su.nctp) {
    case 1: return launch_pool_squeeze_n<1>(grid, lds, s, a);
    case 2: return launch_pool_squeeze_n<2>(grid, lds, s, a);
    case 3: return launch_pool_squeeze_n<3>(grid, lds, s, a);
    default: return launch_pool_squeeze_n<4>(grid, lds, s, a);
  }
}

// fire13's merged expand pair + conv14 + head (Op::head_fused); W = full output width.
hipError_t launch_fire_head(const Op& op, FireHeadArgs f, int N, int H, int W, const _Float16* w16, const float* bias,
                            hipStream_t s) {
  if (!w16 || !op.up_fused || op.cin_t != 16 || op.sub[0].nctp != 2 || op.sub[1].nctp != 2 || op.up[0].nctp != 1 ||
      op.hd.nctp < 1 || op.hd.nctp > 2 || (W & 15) || op.up[1].b_off != op.up[0].b_off + 32)
    return hipErrorInvalidValue;
  f.N = N; f.H = H; f.W = W;
  f.tilesH = (H + kFhTH - 1) / kFhTH;
  f.tilesW = W / kFhTW;
  f.NC = op.hd.cout;
  f.up_w16[0] = w16 + op.up[0].w16_off;
  f.up_w16[1] = w16 + op.up[1].w16_off;
  f.up_bias = bias + op.up[0].b_off;
  f.e1_w16 = w16 + op.sub[0].w16_off; f.e1_bias = bias + op.sub[0].b_off;
  f.e3_w16 = w16 + op.sub[1].w16_off; f.e3_bias = bias + op.sub[1].b_off;
  f.sk_w16 = w16 + op.skm.w16_off; f.sk_bias = bias + op.skm.b_off;
  f.hd_w16 = w16 + op.hd.w16_off; f.hd_bias = bias + op.hd.b_off;
  const dim3 grid((unsigned)(N * f.tilesH * f.tilesW));
#ifdef PCLSEG_WITH_STAMPS
  static const char* fh_labels[8] = {"entry", "source patch + barrier", "up-conv + barrier", "expand3x3 K", "expand1x1 K",
                                     "F epilogue + barrier", "conv14 K", "head epilogue"};
  StampDump stamp_dump(op.name() + "+head", &f.stamps, grid, s, fh_labels);
#endif
  if (op.hd.nctp == 1) {
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&fire_head_kernel<1>), kFhLds)) return e;
    hipLaunchKernelGGL((fire_head_kernel<1>), grid, dim3(256), kFhLds, s, f);
  } else {
    if (hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(&fire_head_kernel<2>), kFhLds)) return e;
    hipLaunchKernelGGL((fire_head_kernel<2>), grid, dim3(256), kFhLds, s, f);
  }
  return hipGetLastError();
}

hipError_t launch_cam(CamArgs c, int N, int H, int W, int C, hipStream_t s) {
  c.N = N; c.H = H; c.W = W;
  constexpr int kTH = 4, kCK = 64;  // 4 x 26 pixel tiles, 64-channel chunks, 512-thread blocks
  const int R = C / 16;
  c.tilesH = (H + kTH - 1) / kTH;
  c.tilesW = (W + kCamTW - 1) / kCamTW;
  const dim3 grid((unsigned)(N * c.tilesH * c.tilesW));
  size_t lds = (size_t)(kTH * kCamPW * kCK + kCK * R) * sizeof(float);
  if (c.sq_out) {   // fused squeeze (cam2 -> fire3): gated chunk as split-f16 behind the squeeze activations
    const int segs = (kTH * kCamTW + 15) / 16;
    lds = std::max(lds, (size_t)((kTH * kCamTW * R * 4 + 15) & ~15) + (size_t)segs * 16 * (2 * kCK + kPadF16) * sizeof(_Float16));
    const int nq = (c.sq_C + 15) / 16;
    if (C == 128 && nq == 1) hipLaunchKernelGGL((cam_kernel<128, 8, kTH, kCK, 1>), grid, dim3(8 * kCK), lds, s, c);
    else if (C == 128 && nq == 2) hipLaunchKernelGGL((cam_kernel<128, 8, kTH, kCK, 2>), grid, dim3(8 * kCK), lds, s, c);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (C == 64) hipLaunchKernelGGL((cam_kernel<64, 4, kTH, kCK>), grid, dim3(8 * kCK), lds, s, c);
  else if (C == 128) hipLaunchKernelGGL((cam_kernel<128, 8, kTH, kCK>), grid, dim3(8 * kCK), lds, s, c);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_pool(const float* in, float* out, int N, int H, int Win, int C, int kh, int kw,
                       int sw, hipStream_t s) {
  int wo, pl, ho, pt;
  same_pad(Win, kw, sw, &wo, &pl);
  same_pad(H, kh, 1, &ho, &pt);
  if (kh == 3 && kw == 3 && sw == 2) {
    constexpr int kRows = 4;
    const unsigned hbn = (unsigned)((H + kRows - 1) / kRows);
    if (hbn > 65535u || (unsigned)N > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL((maxpool3x3s2_kernel<kRows>), dim3((unsigned)((wo * (C / 4) + 255) / 256), hbn, (unsigned)N),
                       dim3(256), 0, s, in, out, N, H, Win, wo, C, pl);
    return hipGetLastError();
  }
  const size_t total = (size_t)N * H * wo * (C / 4);
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, in, out, N, H, Win, wo, C, kh, kw,
                     sw, pt, pl);
  return hipGetLastError();
}

unsigned stream_blocks(size_t items) { return (unsigned)std::min<size_t>((items + 255) / 256, 8192); }

int ensure(pclseg_handle* h, void** p, size_t* have, size_t need) {
  if (*have >= need) return PCLSEG_OK;
  if (*p) HIP_TRY(h, hipFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(h, hipMalloc(p, need));
  *have = need;
  return PCLSEG_OK;
}

// One sweep of the network over `cnt` scans already present in the arena's input tensor.
int run_ops(pclseg_handle* h, int lane, int cnt, const uint8_t* mask, int32_t* preds, float* probs,
            float* logits, bool exact) {
  const Graph& g = h->g;
  float* const arena = h->d_arena_lane[lane];
  const hipStream_t stream = h->nlanes > 1 ? h->lane_stream[lane] : h->stream;
  for (const Op& op : g.ops) {
    const TensorInfo& ti = g.tensors[op.in];
    const float* in = arena + ti.offset;
    if (op.kind == OP_POOL) {
      float* out = arena + g.tensors[op.out].offset;
      HIP_TRY(h, launch_pool(in, out, cnt, ti.H, ti.W, ti.C, op.pool_kh, op.pool_kw, op.sw, stream));
      continue;
    }
    if (op.kind == OP_CAM) {
      CamArgs c;
      memset(&c, 0, sizeof(c));
      c.x = in;
      if (op.fsq_fused) {   // cam2 -> fire3/squeeze in one kernel: op.out is the squeeze tensor
        if (exact) return fail(h, PCLSEG_ERR_STATE, "internal: fused CAM squeeze in an exact-f32 sweep");
        const TensorInfo& to = g.tensors[op.out];
        c.sq_out = arena + to.offset;
        c.sq_C = to.C;
        c.sq_s16 = to.fmt == FMT_S16 ? 1 : 0;
        c.sq_w16 = h->d_w16 + op.fsq.w16_off;
        c.sq_bias = h->d_bias + op.fsq.b_off;
        c.range_flag = h->d_range;
      } else {
        c.out = arena + g.tensors[op.out].offset;
      }
      const int C = op.cin_t, R = C / 16;
      c.w1 = h->d_bias + op.sub[0].b_off; c.b1 = c.w1 + (size_t)C * R;
      c.w2 = h->d_bias + op.sub[1].b_off; c.b2 = c.w2 + (size_t)R * C;
      HIP_TRY(h, launch_cam(c, cnt, ti.H, ti.W, C, stream));
      continue;
    }
    if (op.head_fused) {   // fire13 + conv14 + head: `in` is fire13/squeeze at half width
      if (exact || ti.fmt != FMT_S16) return fail(h, PCLSEG_ERR_STATE, "internal: fused head in an exact-f32 sweep");
      FireHeadArgs f;
      memset(&f, 0, sizeof(f));
      f.sq = reinterpret_cast<const _Float16*>(in);
      f.x8 = arena + g.tensors[op.sk_in].offset;
      f.mask = mask; f.preds = preds; f.probs = probs; f.logits = logits;
      f.none_index = g.desc.none_index;
      f.range_flag = h->d_range;
      const hipError_t le = launch_fire_head(op, f, cnt, ti.H, 2 * ti.W, h->d_w16, h->d_bias, stream);
      if (le != hipSuccess)
        return fail(h, PCLSEG_ERR_HIP, fmt("launch of '%s' + head failed (%s)", op.name().c_str(), hipGetErrorString(le)));
      continue;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.in_s16 = (!exact && ti.fmt == FMT_S16) ? 1 : 0;
    a.range_flag = exact ? nullptr : h->d_range;
    if (op.kind == OP_HEAD) {
      a.mask = mask;
      a.preds = preds;
      a.probs = probs;
      a.logits = logits;
      a.none_index = g.desc.none_index;
    } else {
      const TensorInfo& to = g.tensors[op.out];
      a.out = arena + to.offset;
      a.out_C = to.C;
      a.out_s16 = (!exact && to.fmt == FMT_S16) ? 1 : 0;
      if (op.res1 >= 0) { a.res1 = arena + g.tensors[op.res1].offset; a.res1_C = g.tensors[op.res1].C; }
      if (op.res2 >= 0) { a.res2 = arena + g.tensors[op.res2].offset; a.res2_C = g.tensors[op.res2].C; }
      if (op.sk_in >= 0) { a.skx = arena + g.tensors[op.sk_in].offset; a.skw = h->d_bias + op.sk.b_off; }
    }
    const hipError_t le = op.pool_fused
        ? (exact ? hipErrorInvalidValue : launch_pool_squeeze(op, cnt, ti.H, ti.W, a, h->d_w16, h->d_bias, stream))
        : launch_conv(op, cnt, ti.H, op.up_fused ? 2 * ti.W : ti.W, a, h->d_w32, h->d_w16, h->d_bias, exact, stream);
    if (le != hipSuccess)
      return fail(h, PCLSEG_ERR_HIP, fmt("launch of '%s' failed (%s): block shape mtw=%d ntw=%d wn=%d nw=%d, pair=%d, fused squeeze=%d",
                                         op.name().c_str(), hipGetErrorString(le), op.mtw, op.ntw, op.wn, op.nw, (int)op.pair, (int)op.fsq_fused));
  }
  return PCLSEG_OK;
}

// Is `p` page-locked host memory (hipHostMalloc / hipHostRegister / torch pin_memory)?  Only then is
// hipMemcpyAsync a true DMA that returns at once; pageable memory goes through a bounce buffer.
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  const hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

// Fence the lane streams into the caller's stream (also on error paths: nothing may still be
// running on a side stream when control returns to the caller).
int join_lanes(pclseg_handle* h) {
  if (h->nlanes <= 1) return PCLSEG_OK;
  for (int l = 0; l < h->nlanes; ++l) {
    HIP_TRY(h, hipEventRecord(h->ev_lane[l], h->lane_stream[l]));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_lane[l], 0));
  }
  return PCLSEG_OK;
}
void drain_after_error(pclseg_handle* h) {
  if (h->s_h2d) (void)hipStreamSynchronize(h->s_h2d);
  for (int l = 0; l < h->nlanes; ++l)
    if (h->lane_stream[l]) (void)hipStreamSynchronize(h->lane_stream[l]);
  if (h->s_d2h) (void)hipStreamSynchronize(h->s_d2h);
  (void)hipStreamSynchronize(h->stream);
  (void)hipGetLastError();
}

__global__ __launch_bounds__(256) void copy_bytes_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  const size_t n16 = aligned ? n >> 4 : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = t; i < n16; i += stride) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
  for (size_t k = (n16 << 4) + t; k < n; k += stride) dst[k] = src[k];   // tail (or everything, if misaligned)
}
int zero_copy_mode() {   // bit 0: inputs, bit 1: predictions
  static const int on = tune_env("PCLSEG_ZERO_COPY", 3);   // tuning aid: 0 = DMA copies
  return on;
}
bool zero_copy_enabled() { return zero_copy_mode() != 0; }

// Pageable caller buffers cross a pinned bounce slab; one CPU thread copies ~10 GB/s, which at 3 MB per
// scan caps the boundary below the GPU's rate.  Large copies are therefore split over a few helper threads
// (process-wide, created on first use, parked on a condition variable in between).
class CopyPool {
 public:
  static CopyPool& get() { static CopyPool* p = new CopyPool(); return *p; }   // (never destroyed: workers are detached)
  void copy(void* dst, const void* src, size_t bytes) {
    if (bytes < ((size_t)1 << 20) || nworkers_ == 0 || getpid() != pid_) { memcpy(dst, src, bytes); return; }   // (a forked child has no workers)
    std::lock_guard<std::mutex> one_call(call_m_);
    const int parts = nworkers_ + 1;
    const size_t chunk = ((bytes / parts) + 4095) & ~(size_t)4095;
    std::unique_lock<std::mutex> lk(m_);
    dst_ = (char*)dst; src_ = (const char*)src; bytes_ = bytes; chunk_ = chunk;
    next_ = 1; pending_ = nworkers_; ++gen_;
    lk.unlock();
    cv_.notify_all();
    memcpy(dst, src, std::min(chunk, bytes));            // the caller takes part 0
    lk.lock();
    done_.wait(lk, [&] { return pending_ == 0; });
  }
 private:
  CopyPool() {
    static const int n = tune_env("PCLSEG_COPY_THREADS", 4);   // tuning aid: 1 = caller only
    nworkers_ = std::max(0, std::min(n, 16) - 1);
    pid_ = getpid();
    for (int i = 0; i < nworkers_; ++i) std::thread([this] { run(); }).detach();
  }
  void run() {
    unsigned long seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      const int part = next_++;
      const size_t off = (size_t)part * chunk_;
      char* d = dst_; const char* s = src_; const size_t total = bytes_, chunk = chunk_;
      lk.unlock();
      if (off < total) memcpy(d + off, s + off, std::min(chunk, total - off));
      lk.lock();
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::mutex m_, call_m_;
  pid_t pid_ = 0;
  std::condition_variable cv_, done_;
  char* dst_ = nullptr; const char* src_ = nullptr;
  size_t bytes_ = 0, chunk_ = 0;
  int next_ = 0, pending_ = 0, nworkers_ = 0;
  unsigned long gen_ = 0;
};

// Host-boundary staging of one lane (PCLSEG_MEM_HOST).  Device slabs hold one micro-batch; the
// pinned bounce slab is only used when the caller's input is pageable memory.
int ensure_host_lane(pclseg_handle* h, int l, size_t in_bytes, size_t px, int NC, bool want_probs, bool want_logits) {
  pclseg_handle::HostLane& L = h->hl[l];
  int rc;
  (void)in_bytes;   // (the input and prediction slabs are allocated in sweep(), only where a micro-batch needs them)
  if ((rc = ensure(h, (void**)&L.d_maskin, &L.maskin_bytes, px))) return rc;
  if (want_probs && (rc = ensure(h, (void**)&L.d_probs, &L.probs_bytes, px * NC * sizeof(float)))) return rc;
  if (want_logits && (rc = ensure(h, (void**)&L.d_logits, &L.logits_bytes, px * NC * sizeof(float)))) return rc;
  return PCLSEG_OK;
}
int ensure_pinned(pclseg_handle* h, void** p, size_t* have, size_t need) {
  if (*have >= need) return PCLSEG_OK;
  if (*p) HIP_TRY(h, hipHostFree(*p));
  *p = nullptr;
  *have = 0;
  HIP_TRY(h, hipHostMalloc(p, need, hipHostMallocDefault));
  *have = need;
  return PCLSEG_OK;
}

// The sweep of one call: micro-batches dealt round-robin to the lanes.
//   PCLSEG_MEM_DEVICE: pointers are device memory; asynchronous on the handle's stream.
//   PCLSEG_MEM_HOST  : every micro-batch's input crosses PCIe on ITS LANE's stream (H2D of
//     micro-batch k+1 runs beside the kernels of k on another lane and the D2H of k-1), results
//     return the same way; page-locked caller buffers are DMA'd directly, pageable ones go through
//     pinned bounce buffers (input: per-lane slab, copied by the CPU while the GPU works on earlier
//     micro-batches; outputs: one full-size pinned buffer, copied out after the final fence).
int sweep(pclseg_handle* h, const float* input, bool raw, const uint8_t* mask_in, int n, int32_t* preds,
          float* probs, float* logits, uint8_t* mask_out, int mem, bool exact) {
  const Graph& g = h->g;
  const size_t HW = (size_t)g.desc.height * g.desc.width;
  const int NC = g.desc.num_class;
  const int cin = raw ? 5 : 6;
  const bool host = mem == PCLSEG_MEM_HOST || mem == PCLSEG_MEM_HOST_ASYNC;
  const bool host_async = mem == PCLSEG_MEM_HOST_ASYNC;
  const bool multi = h->nlanes > 1;

  // Micro-batches: at most g.micro_batch scans each, their number rounded up to a multiple of the
  // lane count and the scans spread evenly, so every lane gets the same work (32 scans, 3 lanes:
  // 4,4,4,4,4,3,3,3,3 instead of eight 4s dealt 3/3/2).
  int nmb = (n + g.micro_batch - 1) / g.micro_batch;
  if (multi && nmb > 1) nmb = std::min(n, ((nmb + h->nlanes - 1) / h->nlanes) * h->nlanes);
  const int mb_lo = n / nmb, mb_extra = n % nmb;  // the first mb_extra micro-batches take one more
  const int mb_max = mb_lo + (mb_extra ? 1 : 0);

  bool in_pinned = false, mask_pinned = false;
  int32_t* o_preds = preds; float* o_probs = probs; float* o_logits = logits; uint8_t* o_mask = mask_out;
  const int nslots = std::min(pclseg_handle::kSlotsPerLane * h->nlanes, nmb);
  if (host) {
    int rc;
    if (!h->s_h2d) {
      HIP_TRY(h, hipStreamCreateWithFlags(&h->s_h2d, hipStreamNonBlocking));
      HIP_TRY(h, hipStreamCreateWithFlags(&h->s_d2h, hipStreamNonBlocking));
      HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copy_tail, hipEventDisableTiming));
    }
    in_pinned = is_pinned_host(input);
    mask_pinned = raw || is_pinned_host(mask_in);
    for (int l = 0; l < nslots; ++l) {
      pclseg_handle::HostLane& L = h->hl[l];
      if ((rc = ensure_host_lane(h, l, (size_t)mb_max * HW * cin * sizeof(float), (size_t)mb_max * HW, NC,
                                 probs != nullptr, logits != nullptr))) return rc;
      if (!in_pinned && (rc = ensure_pinned(h, (void**)&L.p_in, &L.p_in_bytes, (size_t)mb_max * HW * cin * sizeof(float)))) return rc;
      if (!mask_pinned && (rc = ensure_pinned(h, (void**)&L.p_mask, &L.p_mask_bytes, (size_t)mb_max * HW))) return rc;
      if (!L.ev_in) {
        HIP_TRY(h, hipEventCreateWithFlags(&L.ev_in, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&L.ev_out, hipEventDisableTiming));
      }
      if (!host_async && !h->host_async_pending) L.used = false;   // nothing of an earlier call is in flight
    }
    if (host_async) {
      if (!in_pinned || !mask_pinned || !is_pinned_host(preds) || (probs && !is_pinned_host(probs)) ||
          (logits && !is_pinned_host(logits)) || (mask_out && !is_pinned_host(mask_out)))
        return fail(h, PCLSEG_ERR_BAD_ARG, "PCLSEG_MEM_HOST_ASYNC needs page-locked host buffers "
                                           "(pclseg_host_alloc / hipHostMalloc / hipHostRegister)");
      h->host_async_pending = true;
    }
    // pageable outputs: full-size pinned bounce, copied out after the fence
    if (!is_pinned_host(preds)) {
      if ((rc = ensure_pinned(h, (void**)&h->p_preds, &h->p_preds_bytes, (size_t)n * HW * sizeof(int32_t)))) return rc;
      o_preds = h->p_preds;
    }
    if (probs && !is_pinned_host(probs)) {
      if ((rc = ensure_pinned(h, (void**)&h->p_probs, &h->p_probs_bytes, (size_t)n * HW * NC * sizeof(float)))) return rc;
      o_probs = h->p_probs;
    }
    if (logits && !is_pinned_host(logits)) {
      if ((rc = ensure_pinned(h, (void**)&h->p_logits, &h->p_logits_bytes, (size_t)n * HW * NC * sizeof(float)))) return rc;
      o_logits = h->p_logits;
    }
    if (mask_out && !is_pinned_host(mask_out)) {
      if ((rc = ensure_pinned(h, (void**)&h->p_mask, &h->p_mask_bytes, (size_t)n * HW))) return rc;
      o_mask = h->p_mask;
    }
  }

  const bool need_dma_out = probs != nullptr || logits != nullptr || mask_out != nullptr;
  // A kernel that reads over PCIe is bound by the link (≈ 50 GB/s), not by the GPU: it gets one small
  // block per CU, so it leaves the wave slots to the other lanes' kernels while it waits
  static const unsigned zc_blocks = tune_env("PCLSEG_ZC_BLOCKS", 16u);
  NormArgs na;
  for (int i = 0; i < 5; ++i) { na.mean[i] = g.desc.mean[i]; na.std[i] = g.desc.std[i]; }
  if (multi && !host) {  // lanes start after everything already queued on the caller's stream (device inputs);
                         // host inputs depend on nothing there, and consecutive host-async calls must overlap
    HIP_TRY(h, hipEventRecord(h->ev_in, h->stream));
    for (int l = 0; l < h->nlanes; ++l) HIP_TRY(h, hipStreamWaitEvent(h->lane_stream[l], h->ev_in, 0));
  }
  int s0 = 0;
  for (int mbi = 0; mbi < nmb; ++mbi) {
    const int cnt = mb_lo + (mbi < mb_extra ? 1 : 0);
    const size_t P = (size_t)cnt * HW;
    const int lane = mbi % h->nlanes;
    const hipStream_t stream = multi ? h->lane_stream[lane] : h->stream;
    pclseg_handle::HostLane& L = h->hl[host ? mbi % (pclseg_handle::kSlotsPerLane * h->nlanes) : lane];   // slot s always serves lane s % nlanes
    float* d_lidar8 = h->d_arena_lane[lane] + g.tensors[g.t_input].offset;
    const float* d_in = input + (size_t)s0 * HW * cin;
    const uint8_t* d_mask_in = mask_in ? mask_in + (size_t)s0 * HW : nullptr;
    int32_t* d_preds = preds + (size_t)s0 * HW;
    float* d_probs = probs ? probs + (size_t)s0 * HW * NC : nullptr;
    float* d_logits = logits ? logits + (size_t)s0 * HW * NC : nullptr;
    uint8_t* d_mask_out = mask_out ? mask_out + (size_t)s0 * HW : nullptr;
    bool zc_in = false, zc_preds = false;   // this micro-batch's input / predictions cross PCIe inside kernels
    if (host) {
      int rc0;
      const float* src = d_in;
      const uint8_t* msrc = d_mask_in;
      if (!in_pinned || !mask_pinned) {
        if (L.used) HIP_TRY(h, hipEventSynchronize(L.ev_in));   // the bounce slab's previous reader / upload has left it
        if (!in_pinned) { CopyPool::get().copy(L.p_in, d_in, P * cin * sizeof(float)); src = (const float*)L.p_in; }
        if (!mask_pinned) { memcpy(L.p_mask, d_mask_in, P); msrc = (const uint8_t*)L.p_mask; }
      }
      // With ROCm 7.2 a hipMemcpyAsync queued behind an UNSATISFIED cross-stream event wait blocks the calling
      // thread until the dependency resolves (measured in steady state: 350 us per upload, 200 us per
      // download), which starves the other lanes: mean kernels in flight 1.2 instead of 2.4.  Page-locked
      // host memory is mapped into the GPU's address space, so wherever a DMA command would block, a kernel
      // on the lane's own stream moves the bytes instead: a small coalesced copy kernel brings the scans in
      // (ordered by the lane itself: no event), and the head writes the int32 predictions straight into the
      // caller's buffer.  An upload whose slot is already free still goes through the DMA engine (it then
      // runs ahead of the lane, beside the kernels).  Optional float outputs (probs / logits / mask) always
      // leave through the DMA stream.
      void *zin = nullptr, *zmask = nullptr, *zpreds = nullptr;
      const int zmode = zero_copy_mode();
      const bool slot_busy = L.used && hipEventQuery(L.ev_done) != hipSuccess;
      (void)hipGetLastError();
      zc_in = (zmode & 1) && (host_async || slot_busy) &&
              hipHostGetDevicePointer(&zin, const_cast<float*>(src), 0) == hipSuccess &&
              (raw || hipHostGetDevicePointer(&zmask, const_cast<uint8_t*>(msrc), 0) == hipSuccess);
      zc_preds = (zmode & 2) && hipHostGetDevicePointer(&zpreds, o_preds + (size_t)s0 * HW, 0) == hipSuccess;
      (void)hipGetLastError();
      if ((rc0 = ensure(h, (void**)&L.d_in, &L.in_bytes, (size_t)mb_max * HW * cin * sizeof(float)))) return rc0;
      if (!zc_preds && (rc0 = ensure(h, (void**)&L.d_preds, &L.preds_bytes, (size_t)mb_max * HW * sizeof(int32_t)))) return rc0;
      if (zc_in) {
        // one small block per few CUs: the copy is bound by the link (~50 GB/s), and a large grid would
        // hold every wave slot of the chip while it waits (16 blocks: 93 % of the device-resident rate for
        // enqueue-only calls, 256 blocks: 83 %).  16 bytes per lane, fully coalesced — the normalise kernel's
        // own 20-byte-stride reads would cross PCIe as small partial requests at a quarter of the link rate.
        hipLaunchKernelGGL(copy_bytes_kernel, dim3(zc_blocks), dim3(256), 0, stream, (const uint8_t*)zin, (uint8_t*)L.d_in, P * cin * sizeof(float));
        HIP_TRY(h, hipGetLastError());
        if (!raw) {
          hipLaunchKernelGGL(copy_bytes_kernel, dim3(zc_blocks), dim3(256), 0, stream, (const uint8_t*)zmask, L.d_maskin, P);
          HIP_TRY(h, hipGetLastError());
        }
        if (!in_pinned || !mask_pinned) HIP_TRY(h, hipEventRecord(L.ev_in, stream));   // the bounce slab has been read
      } else {
        // upload on the H2D stream into this micro-batch's slot (two slots per lane, so the upload of the
        // lane's NEXT micro-batch runs while this one computes)
        if (L.used) HIP_TRY(h, hipStreamWaitEvent(h->s_h2d, L.ev_done, 0));   // slot's previous kernels have read it
        HIP_TRY(h, hipMemcpyAsync(L.d_in, src, P * cin * sizeof(float), hipMemcpyHostToDevice, h->s_h2d));
        if (!raw) HIP_TRY(h, hipMemcpyAsync(L.d_maskin, msrc, P, hipMemcpyHostToDevice, h->s_h2d));
        HIP_TRY(h, hipEventRecord(L.ev_in, h->s_h2d));
        HIP_TRY(h, hipStreamWaitEvent(stream, L.ev_in, 0));
      }
      if (L.used && L.out_dma) HIP_TRY(h, hipStreamWaitEvent(stream, L.ev_out, 0));   // slot's previous results have left
      d_in = (const float*)L.d_in;
      d_mask_in = L.d_maskin;
      d_preds = zc_preds ? (int32_t*)zpreds : L.d_preds;
      d_probs = probs ? L.d_probs : nullptr;
      d_logits = logits ? L.d_logits : nullptr;
      d_mask_out = raw ? L.d_maskin : nullptr;   // raw mode: the slot's own mask slab (copied out below)
    }
    const uint8_t* mask_mb;
    if (raw) {
      uint8_t* mdst = d_mask_out ? d_mask_out : h->d_mask_lane[lane];
      hipLaunchKernelGGL(normalize_kernel<8>, dim3(stream_blocks(P)), dim3(256), 0, stream, d_in, d_lidar8, mdst, P, na);
      HIP_TRY(h, hipGetLastError());
      mask_mb = mdst;
    } else {
      hipLaunchKernelGGL(pad6to8_kernel, dim3(stream_blocks(P)), dim3(256), 0, stream, d_in, d_lidar8, P);
      HIP_TRY(h, hipGetLastError());
      mask_mb = d_mask_in;
    }
    int rc = run_ops(h, lane, cnt, mask_mb, d_preds, d_probs, d_logits, exact);
    if (rc) return rc;
    if (host) {   // results leave on the D2H stream once the slot's kernels are done
      HIP_TRY(h, hipEventRecord(L.ev_done, stream));
      L.out_dma = !zc_preds || need_dma_out;
      L.used = true;
    }
    if (host && L.out_dma) {
      HIP_TRY(h, hipStreamWaitEvent(h->s_d2h, L.ev_done, 0));
      if (!zc_preds) HIP_TRY(h, hipMemcpyAsync(o_preds + (size_t)s0 * HW, d_preds, P * sizeof(int32_t), hipMemcpyDeviceToHost, h->s_d2h));
      if (probs) HIP_TRY(h, hipMemcpyAsync(o_probs + (size_t)s0 * HW * NC, d_probs, P * NC * sizeof(float), hipMemcpyDeviceToHost, h->s_d2h));
      if (logits) HIP_TRY(h, hipMemcpyAsync(o_logits + (size_t)s0 * HW * NC, d_logits, P * NC * sizeof(float), hipMemcpyDeviceToHost, h->s_d2h));
      if (mask_out) HIP_TRY(h, hipMemcpyAsync(o_mask + (size_t)s0 * HW, mask_mb, P, hipMemcpyDeviceToHost, h->s_d2h));
      HIP_TRY(h, hipEventRecord(L.ev_out, h->s_d2h));
    }
    h->last_count = cnt;
    h->last_exact = exact;
    h->d_arena = h->d_arena_lane[lane];
    s0 += cnt;
  }
  int rc = join_lanes(h);  // the caller's stream continues only after every lane has drained
  if (rc) return rc;
  if (host) {
    HIP_TRY(h, hipEventRecord(h->ev_copy_tail, h->s_d2h));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_copy_tail, 0));
    if (host_async) return PCLSEG_OK;   // pclseg_sync waits (and reports the range guard)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->host_async_pending = false;
    if (o_preds != preds) CopyPool::get().copy(preds, o_preds, (size_t)n * HW * sizeof(int32_t));
    if (probs && o_probs != probs) CopyPool::get().copy(probs, o_probs, (size_t)n * HW * NC * sizeof(float));
    if (logits && o_logits != logits) CopyPool::get().copy(logits, o_logits, (size_t)n * HW * NC * sizeof(float));
    if (mask_out && o_mask != mask_out) memcpy(mask_out, o_mask, (size_t)n * HW);
  }
  return PCLSEG_OK;
}

// Read (and clear) the sticky split-f16 range flag; the handle's stream must be idle.
int take_range_flag(pclseg_handle* h, bool* fired) {
  *fired = false;
  if (h->exact || !h->d_range) return PCLSEG_OK;
  HIP_TRY(h, hipMemcpy(h->h_range, h->d_range, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (*h->h_range) {
    *fired = true;
    HIP_TRY(h, hipMemset(h->d_range, 0, sizeof(unsigned)));
  }
  return PCLSEG_OK;
}

const char* kRangeMsg =
    "split-f16 range exceeded: an activation reached |v| >= 65504, the results of this call are not valid; "
    "create the handle with PCLSEG_FLAG_EXACT_F32 or PCLSEG_FLAG_RANGE_FALLBACK";

void clear_pending(pclseg_handle* h) {
  h->pending.clear();
  h->pending_overflow = false;
  h->unchecked_calls = 0;
}

// The range guard fired on a fallback handle: re-run EVERY asynchronous call enqueued since the flag was
// last read with exact float32 products, oldest first (their buffers are the caller's and must still be
// valid, as for any asynchronous call that has not been synchronised).  Leaves the stream idle.
int repair_pending(pclseg_handle* h) {
  if (h->pending_overflow) {
    const size_t cap = pclseg_handle::kMaxPending;
    clear_pending(h);
    return fail(h, PCLSEG_ERR_RANGE, std::string(kRangeMsg) + fmt(" (more than %zu calls were enqueued without a pclseg_sync: "
                                                                  "the earliest cannot be repaired)", cap));
  }
  std::vector<pclseg_handle::PendingCall> calls;
  calls.swap(h->pending);
  clear_pending(h);
  for (const pclseg_handle::PendingCall& c : calls) {
    const int rc = sweep(h, c.input, c.raw, c.mask_in, c.n, c.preds, c.probs, c.logits, c.mask_out, c.mem, true);
    if (rc) { drain_after_error(h); return rc; }
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->host_async_pending = false;
  return PCLSEG_OK;
}

int forward_impl(pclseg_handle* h, const float* input, bool raw, const uint8_t* mask_in, int n,
                 int32_t* preds, float* probs, float* logits, uint8_t* mask_out, int mem) {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  if (!h->finalized) return fail(h, PCLSEG_ERR_STATE, "forward called before pclseg_finalize");
  if (!input || !preds) return fail(h, PCLSEG_ERR_BAD_ARG, "input and preds must not be NULL");
  if (!raw && !mask_in) return fail(h, PCLSEG_ERR_BAD_ARG, "mask must not be NULL");
  if (n <= 0) return fail(h, PCLSEG_ERR_BAD_ARG, fmt("n must be positive, got %d", n));
  if (mem != PCLSEG_MEM_HOST && mem != PCLSEG_MEM_DEVICE && mem != PCLSEG_MEM_HOST_ASYNC)
    return fail(h, PCLSEG_ERR_BAD_ARG, fmt("unknown mem %d", mem));
  DeviceGuard guard(h->device);
  int rc = sweep(h, input, raw, mask_in, n, preds, probs, logits, mask_out, mem, h->exact || h->force_exact);
  if (rc) { drain_after_error(h); return rc; }
  if (mem != PCLSEG_MEM_HOST) {   // asynchronous: pclseg_sync reports / repairs a range overflow
    if (h->unchecked_calls < INT32_MAX) ++h->unchecked_calls;   // saturates: handles that never call pclseg_sync
    if (h->fallback && !h->force_exact) {
      if (h->pending.size() < pclseg_handle::kMaxPending) {
        pclseg_handle::PendingCall c;
        c.mem = mem;
        c.input = input; c.raw = raw; c.mask_in = mask_in; c.n = n;
        c.preds = preds; c.probs = probs; c.logits = logits; c.mask_out = mask_out;
        h->pending.push_back(c);
      } else {
        h->pending_overflow = true;
      }
    }
    return PCLSEG_OK;
  }
  // synchronous call: the stream is idle, so every earlier asynchronous call has finished too and the
  // flag covers all of them
  bool fired = false;
  if ((rc = take_range_flag(h, &fired))) return rc;
  if (!fired) { clear_pending(h); return PCLSEG_OK; }
  if (!h->fallback) {
    const int earlier = h->unchecked_calls;
    clear_pending(h);
    return fail(h, PCLSEG_ERR_RANGE, earlier ? std::string(kRangeMsg) + fmt(" (raised by this call or one of the %d asynchronous "
                                                                            "calls enqueued before it: all of their outputs are suspect)", earlier)
                                             : std::string(kRangeMsg));
  }
  if ((rc = repair_pending(h))) return rc;
  rc = sweep(h, input, raw, mask_in, n, preds, probs, logits, mask_out, mem, true);   // exact float32
  if (rc) drain_after_error(h);
  return rc;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

// pack + upload the parameters of one stand-alone op and launch it on the default stream
int run_single_op(Op* op, const FoldIn* folds, int n, int h, int w, ConvArgs a, int math) {
  if (math != PCLSEG_MATH_F16X3 && math != PCLSEG_MATH_F32)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, fmt("unknown math mode %d", math));
  const bool exact = (math == PCLSEG_MATH_F32);
  op_geometry(op);
  size_t nw32 = 0, nw16 = 0, nb = 0;
  for (int i = 0; i < op->nsub; ++i) {
    SubOp& su = op->sub[i];
    su.w32_off = nw32; nw32 += sub_w32_floats(*op, su);
    su.w16_off = nw16; nw16 += sub_w16_halfs(*op, su);
    su.b_off = nb; nb += (size_t)sub_bias_floats(su);
  }
  std::vector<float> hb(nb), hw32(exact ? nw32 : 0);
  std::vector<_Float16> hw16(exact ? 0 : nw16);
  for (int i = 0; i < op->nsub; ++i) {
    const SubOp& su = op->sub[i];
    std::vector<double> scale, shift;
    fold_bn(su, folds[i], &scale, &shift);
    pack_bias(su, shift, hb.data() + su.b_off);
    if (exact) pack_w32(*op, su, folds[i], scale, hw32.data() + su.w32_off);
    else pack_w16(*op, su, folds[i], scale, hw16.data() + su.w16_off, hb.data() + su.b_off + su.nctp * 16);
  }
  DevBuf dw, db;
  const size_t wbytes = exact ? nw32 * sizeof(float) : nw16 * sizeof(_Float16);
  HIP_TRY(nullptr, hipMalloc(&dw.p, wbytes));
  HIP_TRY(nullptr, hipMalloc(&db.p, nb * sizeof(float)));
  HIP_TRY(nullptr, hipMemcpy(dw.p, exact ? (const void*)hw32.data() : (const void*)hw16.data(), wbytes,
                             hipMemcpyHostToDevice));
  HIP_TRY(nullptr, hipMemcpy(db.p, hb.data(), nb * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(nullptr, launch_conv(*op, n, h, w, a, exact ? (const float*)dw.p : nullptr,
                               exact ? nullptr : (const _Float16*)dw.p, (const float*)db.p, exact, nullptr));
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
}

// ---- packed parameter blob (pclseg_export_packed / pclseg_import_packed) ------------------------
// [PackedHeader][bias floats][split-f16 fragments][exact-f32 fragments]: exactly the three device
// arrays pclseg_finalize uploads, so a rank that receives the blob (one RCCL broadcast) is ready after
// one device-to-device copy — no BatchNorm folding, no repacking, no Keras tensors on that rank.
struct PackedHeader {
  uint32_t magic, version;
  int32_t arch, height, width, num_class, output_stride;
  uint32_t math;            // bit 0: exact-f32 fragments present, bit 1: split-f16 fragments present, bit 2: force_exact
  int64_t n_bias, n_w16, n_w32;
  uint64_t plan_hash;       // block shapes + fragment offsets of every op: a blob only fits the plan that made it
};
constexpr uint32_t kPackedMagic = 0x50434c50u;   // "PLCP"

uint64_t plan_hash(const Graph& g) {
  uint64_t hsh = 1469598103934665603ull;
  auto mix = [&](int64_t v) { for (int i = 0; i < 8; ++i) { hsh ^= (uint64_t)(v >> (8 * i)) & 0xffu; hsh *= 1099511628211ull; } };
  mix(PCLSEG_VERSION);
  for (const Op& op : g.ops) {
    mix(op.kind); mix(op.ntw); mix(op.wn); mix(op.mtw); mix(op.nw); mix(op.ck16); mix(op.ck32);
    mix(op.pair); mix(op.fsq_fused); mix(op.up_fused); mix(op.pool_fused); mix(op.head_fused);
    if (op.head_fused) { mix(op.hd.w16_off); mix(op.hd.b_off); mix(op.skm.w16_off); mix(op.skm.b_off); }
    for (int i = 0; i < op.nsub; ++i) { mix(op.sub[i].w16_off); mix(op.sub[i].w32_off); mix(op.sub[i].b_off); mix(op.sub[i].nctp); }
    if (op.fsq_fused) { mix(op.fsq.w16_off); mix(op.fsq.b_off); }
    if (op.up_fused) for (int i = 0; i < 2; ++i) { mix(op.up[i].w16_off); mix(op.up[i].b_off); }
    if (op.sk_in >= 0) mix(op.sk.b_off);
  }
  return hsh;
}

PackedHeader packed_header(const pclseg_handle* h) {
  PackedHeader ph;
  memset(&ph, 0, sizeof(ph));
  ph.magic = kPackedMagic; ph.version = PCLSEG_VERSION;
  ph.arch = h->g.desc.arch; ph.height = h->g.desc.height; ph.width = h->g.desc.width;
  ph.num_class = h->g.desc.num_class; ph.output_stride = h->g.desc.output_stride;
  ph.math = (h->d_w32 ? 1u : 0u) | (h->d_w16 ? 2u : 0u) | (h->force_exact ? 4u : 0u);
  ph.n_bias = h->g.packed_bias_floats;
  ph.n_w16 = h->d_w16 ? h->g.packed16_halfs : 0;
  ph.n_w32 = h->d_w32 ? h->g.packed32_floats : 0;
  ph.plan_hash = plan_hash(h->g);
  return ph;
}
size_t packed_bytes(const PackedHeader& ph) {
  return sizeof(PackedHeader) + (size_t)ph.n_bias * 4 + (size_t)ph.n_w16 * 2 + (size_t)ph.n_w32 * 4;
}


// ---- exception barrier of the C ABI (include/pclseg.h: "no exceptions cross the boundary") --------
// Every extern "C" entry point is a function-try-block that ends in PCLSEG_CATCH: whatever the host-side
// code throws (std::bad_alloc from the BatchNorm folding / fragment packing of 53 M parameters, a
// std::length_error from a vector sized by a hostile desc, anything else) becomes a status code and a
// message in pclseg_last_error instead of std::terminate in the caller's process.

int on_exception(const pclseg_handle* ch) noexcept {
  pclseg_handle* h = const_cast<pclseg_handle*>(ch);
  int code = PCLSEG_ERR_INTERNAL;
  const char* what = "unknown C++ exception";
  char buf[256];
  try { throw; }
  catch (const std::bad_alloc&) { code = PCLSEG_ERR_OOM; what = "out of host memory (std::bad_alloc)"; }
  catch (const std::length_error& e) {
    code = PCLSEG_ERR_OOM;
    snprintf(buf, sizeof(buf), "allocation size out of range (std::length_error: %s)", e.what());
    what = buf;
  }
  catch (const std::exception& e) {
    snprintf(buf, sizeof(buf), "internal error (C++ exception: %s)", e.what());
    what = buf;
  }
  catch (...) {}
  try {
    return fail(h, code, what);
  } catch (...) {
    g_static_error = code == PCLSEG_ERR_OOM ? "out of host memory (std::bad_alloc)" : "internal error (C++ exception)";
    g_static_handle = ch;
    return code;
  }
}
#define PCLSEG_CATCH(h) catch (...) { return on_exception(h); }

// pclseg_create: the half-built handle is released on every exit path, also when something throws
struct HandleGuard {
  pclseg_handle* h;
  ~HandleGuard() { if (h) (void)pclseg_destroy(h); }
};

}  // namespace

// =============================================================================== C ABI
extern "C" {

int pclseg_version(void) { return PCLSEG_VERSION; }

#ifndef PCLSEG_SRC_SHA
#define PCLSEG_SRC_SHA "unknown"
#endif
const char* pclseg_build_sha(void) { return PCLSEG_SRC_SHA; }

const char* pclseg_last_error(const pclseg_handle* h) try {
  if (g_static_error && h == g_static_handle) return g_static_error;
  if (h) return h->err.c_str();
  return g_last_error.c_str();
} catch (...) { return "pclseg_last_error: internal error"; }

int pclseg_plan(const pclseg_desc* desc, pclseg_plan_info* out) try {
  if (!out) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "out is NULL");
  Graph g;
  int rc = build_graph(desc, &g);
  if (rc) return fail(nullptr, rc, g.error);
  out->num_ops = (int)g.ops.size();
  out->num_weights = (int)g.weights.size();
  out->num_tensors = (int)g.tensors.size();
  out->micro_batch = g.micro_batch;
  out->num_params = g.num_params;
  out->alg_macs_per_scan = g.alg_macs;
  out->alg_bytes_per_scan = g.alg_bytes;
  out->workspace_bytes = g.arena_floats * (int64_t)sizeof(float);
  const bool exact = (desc->flags & PCLSEG_FLAG_EXACT_F32) != 0;
  out->packed_weight_bytes = (exact ? g.packed32_floats * 4 : g.packed16_halfs * 2) + g.packed_bias_floats * 4;
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_plan_ops(const pclseg_desc* desc, char* buf, size_t cap) try {
  if (!buf || !cap) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "buf is NULL");
  Graph g;
  int rc = build_graph(desc, &g);
  if (rc) return fail(nullptr, rc, g.error);
  std::string out;
  for (const Op& op : g.ops) {
    std::string nm = op.name();
    if (op.kind == OP_CONV && op.nsub == 2 && !op.sub[0].deconv) nm = nm.substr(0, nm.find('/')) + "/expand";
    if (op.kind == OP_CAM) nm = nm.substr(0, nm.find('/'));
    if (op.pool_fused) nm = "pool+" + nm;
    if (op.up_fused) nm = "up+" + nm;
    if (op.fsq_fused) nm += "+" + op.fsq.name;
    if (op.head_fused) nm += "+" + op.hd.name + "+head";
    if (op.kind == OP_HEAD) nm += "+head";
    // multiply-accumulates of the launch per scan (every fused piece included)
    const TensorInfo& ti = g.tensors[op.in];
    int64_t macs = 0;
    if (op.kind == OP_CAM) {
      macs = (int64_t)ti.H * ti.W * 2 * op.cin_t * (op.cin_t / 16);
      if (op.fsq_fused) macs += (int64_t)ti.H * ti.W * op.cin_t * op.fsq.cout;
    } else if (op.kind != OP_POOL) {
      int wo, pl;
      same_pad(ti.W, op.pkw, op.sw, &wo, &pl);
      int64_t hw = (int64_t)ti.H * wo;                       // output pixels of the main convolution
      if (op.pool_fused) { same_pad(ti.W, 3, 2, &wo, &pl); hw = (int64_t)ti.H * wo; }
      if (op.up_fused) {                                      // `in` is at half width
        macs += (int64_t)ti.H * ti.W * 2 * 2 * op.cin_t * op.up[0].cout;
        hw = (int64_t)ti.H * ti.W * 2;
      }
      if (op.ow_mul == 2) hw = (int64_t)ti.H * ti.W;          // transposed conv: each parity sub-conv covers Win columns
      for (int i = 0; i < op.nsub; ++i) macs += hw * op.sub[i].nkh * op.sub[i].nkw * op.cin_k * op.sub[i].cout;
      if (op.sk_in >= 0) macs += hw * 6 * op.sk.cout;
      if (op.fsq_fused) macs += hw * (op.sub[0].cout + op.sub[1].cout) * op.fsq.cout;
      if (op.head_fused) macs += hw * 9 * 64 * op.hd.cout;
    }
    // static launch resources of the split-f16 plan (what decides which kernels can share a CU): dynamic LDS
    // bytes per block, threads per block, blocks per scan — the same expressions the launchers use
    int64_t lds = 0, threads = 256, blocks = 0;
    if (op.kind == OP_CAM) {
      const int R = op.cin_t / 16;
      lds = (int64_t)(4 * kCamPW * 64 + 64 * R) * 4;
      if (op.fsq_fused) lds = std::max<int64_t>(lds, ((4 * kCamTW * R * 4 + 15) & ~15) + (int64_t)((4 * kCamTW + 15) / 16) * 16 * (2 * 64 + kPadF16) * 2);
      threads = 512;
      blocks = (int64_t)((ti.H + 3) / 4) * ((ti.W + kCamTW - 1) / kCamTW);
    } else if (op.kind == OP_POOL) {
      blocks = ((int64_t)ti.H * ti.W * ti.C / 4 + 255) / 256 / 4;
    } else if (op.head_fused) {
      lds = kFhLds;
      blocks = (int64_t)((ti.H + kFhTH - 1) / kFhTH) * (2 * ti.W / kFhTW);
    } else if (op.pool_fused) {
      int wo2, pl2;
      same_pad(ti.W, 3, 2, &wo2, &pl2);
      lds = (int64_t)kPoolSqRows * 16 * (2 * op.cin_t + kPadF16) * 2;
      blocks = (int64_t)((ti.H + kPoolSqRows - 1) / kPoolSqRows) * ((wo2 + 15) / 16);
    } else {
      const bool direct = op_is_flat(op) && op.nsub == 1 && op.kind == OP_CONV && op.cin_t % 8 == 0 && op.ck16 >= 32 &&
                          op.sk_in < 0 && op.res2 < 0 && g.tensors[op.in].fmt != FMT_S16 && op.sub[0].nctp <= 4;
      const TileGeom t = tile_geom(op);
      const int Wc = op.up_fused ? 2 * ti.W : ti.W;
      int wo2, pl2;
      same_pad(Wc, op.pkw, op.sw, &wo2, &pl2);
      const int wconv = op.ow_mul == 2 ? Wc : wo2;
      threads = op.nw * 64;
#ifdef PCLSEG_CAND_WIDE
      if (!direct && op_is_wide_1x1(op, g.tensors[op.in].fmt == FMT_S16, op.res1 >= 0 || op.res2 >= 0, op.sk_in >= 0)) {
        int nt, ny_w;
        wide_1x1_geom(op, &nt, &ny_w);
        lds = kW1Lds;
        threads = 512;
        blocks = (((int64_t)ti.H * ti.W + kW1Px - 1) / kW1Px) * ny_w;
      } else
#endif
      if (direct) {
        const bool splitk = op.cin_t >= 256 && op.sub[0].nctp >= 3;
        lds = splitk ? (int64_t)4 * 2 * op.sub[0].nctp * 1024 : 0;
        const int px = splitk ? 32 : 4 * (op.sub[0].nctp == 4 ? 1 : 2) * 16;
        blocks = ((int64_t)ti.H * ti.W + px - 1) / px;
      } else {
        lds = (lds_bytes_f16(op, op.ck16) + 15) & ~(int64_t)15;
        if (op.sk_in >= 0) lds += (int64_t)9 * g.tensors[op.out].C * 4;
        if (op.up_fused) lds = ((lds + 15) & ~(int64_t)15) + (int64_t)t.PH * (t.PW / 2 + 1) * (2 * op.cin_t + kPadF16) * 2;
        if (op.fsq_fused) {
          int64_t slab = (int64_t)8 * op.mtw * 16 * (op.fsq.nctp * 16 + 4) * 4;
#ifdef PCLSEG_CAND_SLAB
          if (op.fsq.nctp == 4 && slab > 96 * 1024) slab = (int64_t)8 * op.mtw * 16 * (2 * 16 + 4) * 4;
#endif
          lds = std::max<int64_t>(lds, slab);
        }
        int ny = 0;
        for (int i = 0; i < op.nsub; ++i) ny += op.sub[i].nctp / (op.ntw * op.wn);
        if (op.pair) ny = op.sub[1].nctp / (op.ntw * op.wn);
        blocks = op_is_flat(op) ? (((int64_t)ti.H * ti.W + t.SEGW * 16 - 1) / (t.SEGW * 16)) * std::max(ny, 1)
                                : (int64_t)((ti.H + t.TH - 1) / t.TH) * ((wconv + t.SEGW * 16 - 1) / (t.SEGW * 16)) * std::max(ny, 1);
      }
    }
    out += nm + "\t" + std::to_string(macs) + "\t" + std::to_string(lds) + "\t" + std::to_string(threads) + "\t" + std::to_string(blocks) + "\n";
  }
  if (out.size() + 1 > cap) return fail(nullptr, PCLSEG_ERR_BAD_ARG, fmt("buffer of %zu bytes, the op list needs %zu", cap, out.size() + 1));
  memcpy(buf, out.c_str(), out.size() + 1);
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_create(const pclseg_desc* desc, pclseg_handle** out) try {
  if (!out) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (!desc) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "desc is NULL");
  pclseg_handle* h = new pclseg_handle();
  HandleGuard hg{h};            // releases the half-built handle on every early exit, thrown or returned
  int rc = build_graph(desc, &h->g);
  if (rc) return fail(nullptr, rc, h->g.error);
#ifndef PCLSEG_TUNING
  // The shipped dispatch tables hold the fused skip-branch epilogue (nets/SqueezeSegV2.py:293,319) for merged expand
  // pairs on 8-wave blocks only (launch_conv: `pair`; launch_conv_epi has no epi 3/4 outside a pair).  A plan that
  // demotes such a pair (op_geometry: patch in more than one chunk; pair_cfg_ok) is refused HERE, with a message,
  // instead of surfacing as hipErrorInvalidValue at the first forward call.  (Exact mode has a catch-all epilogue.)
  if (!(desc->flags & PCLSEG_FLAG_EXACT_F32))
    for (const Op& op : h->g.ops)
      if (op.kind == OP_CONV && op.sk_in >= 0 && !op.head_fused &&
          !(op.pair && pair_cfg_ok(op) && op.nw == 8 && op.ntw == 1 && op.res1 < 0 && op.res2 < 0))
        return fail(nullptr, PCLSEG_ERR_BAD_SHAPE,
                    fmt("layer %s: the fused skip branch needs a merged expand pair on 8-wave blocks; this plan gives "
                        "(pair %d, waves %d, cout tiles per wave %d)", op.name().c_str(), (int)op.pair, op.nw, op.ntw));
#endif
  h->device = desc->device;
  h->host_w.resize(h->g.weights.size());
  h->is_set.assign(h->g.weights.size(), 0);
  auto bail = [&](int code, const std::string& msg) { return fail(nullptr, code, msg); };
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return bail(PCLSEG_ERR_HIP, fmt("no HIP device available (%s); this engine has no CPU fallback",
                                    e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
  if (desc->device < 0 || desc->device >= ndev)
    return bail(PCLSEG_ERR_BAD_ARG, fmt("device %d out of range (have %d)", desc->device, ndev));
  DeviceGuard guard(desc->device);
  {
    int cur = -1;
    if ((e = hipGetDevice(&cur)) != hipSuccess || cur != desc->device)
      return bail(PCLSEG_ERR_HIP, fmt("hipSetDevice(%d): %s", desc->device, hipGetErrorString(e)));
  }
  const size_t arena_bytes = (size_t)h->g.arena_floats * sizeof(float);
  const size_t mask_bytes = (size_t)h->g.micro_batch * desc->height * desc->width;
  {
    int lanes = debug_env("PCLSEG_LANES", 3);  // tuning override
    if (desc->flags & PCLSEG_FLAG_KEEP_ACTIVATIONS) lanes = 1;  // debug reads need one arena
    h->nlanes = std::max(1, std::min(lanes, (int)pclseg_handle::kMaxLanes));
  }
  for (int l = 0; l < h->nlanes; ++l) {
    if ((e = hipMalloc((void**)&h->d_arena_lane[l], arena_bytes)) != hipSuccess)
      return bail(e == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,
                  fmt("hipMalloc(%zu B activation arena): %s", arena_bytes, hipGetErrorString(e)));
    (void)hipMemset(h->d_arena_lane[l], 0, arena_bytes);
    if ((e = hipMalloc((void**)&h->d_mask_lane[l], mask_bytes)) != hipSuccess)
      return bail(PCLSEG_ERR_HIP, fmt("hipMalloc(mask): %s", hipGetErrorString(e)));
    if (h->nlanes > 1) {
      if ((e = hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking)) != hipSuccess ||
          (e = hipEventCreateWithFlags(&h->ev_lane[l], hipEventDisableTiming)) != hipSuccess)
        return bail(PCLSEG_ERR_HIP, fmt("stream/event creation: %s", hipGetErrorString(e)));
    }
  }
  if (h->nlanes > 1 && (e = hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming)) != hipSuccess)
    return bail(PCLSEG_ERR_HIP, fmt("event creation: %s", hipGetErrorString(e)));
  h->d_arena = h->d_arena_lane[0];
  h->exact = (desc->flags & PCLSEG_FLAG_EXACT_F32) != 0;
  h->fallback = !h->exact && (desc->flags & PCLSEG_FLAG_RANGE_FALLBACK) != 0;
  e = hipSuccess;
  if (h->exact || h->fallback) e = hipMalloc((void**)&h->d_w32, (size_t)h->g.packed32_floats * sizeof(float));
  if (e == hipSuccess && !h->exact) e = hipMalloc((void**)&h->d_w16, (size_t)h->g.packed16_halfs * sizeof(_Float16));
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias, (size_t)h->g.packed_bias_floats * sizeof(float));
  if (e == hipSuccess && !h->exact) {
    e = hipMalloc((void**)&h->d_range, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(h->d_range, 0, sizeof(unsigned));
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_range, sizeof(unsigned), hipHostMallocDefault);
  }
  if (e != hipSuccess)
    return bail(e == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,
                fmt("hipMalloc(parameters): %s", hipGetErrorString(e)));
  // a dead handle's exception-barrier text must not speak for a new handle the allocator placed at the same address
  if (g_static_handle == h) { g_static_error = nullptr; g_static_handle = nullptr; }
  hg.h = nullptr;
  *out = h;
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_destroy(pclseg_handle* h) try {
  if (!h) return PCLSEG_OK;
  DeviceGuard guard(h->device);
  for (int l = 0; l < pclseg_handle::kMaxLanes; ++l) {
    if (h->lane_stream[l]) { (void)hipStreamSynchronize(h->lane_stream[l]); (void)hipStreamDestroy(h->lane_stream[l]); }
    if (h->ev_lane[l]) (void)hipEventDestroy(h->ev_lane[l]);
    if (h->d_arena_lane[l]) (void)hipFree(h->d_arena_lane[l]);
    if (h->d_mask_lane[l]) (void)hipFree(h->d_mask_lane[l]);
  }
  if (h->ev_in) (void)hipEventDestroy(h->ev_in);
  for (int l = 0; l < pclseg_handle::kSlotsPerLane * pclseg_handle::kMaxLanes; ++l) {
    pclseg_handle::HostLane& L = h->hl[l];
    void* dev[] = {L.d_in, L.d_maskin, L.d_preds, L.d_probs, L.d_logits};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (L.p_in) (void)hipHostFree(L.p_in);
    if (L.p_mask) (void)hipHostFree(L.p_mask);
    for (hipEvent_t e : {L.ev_in, L.ev_done, L.ev_out}) if (e) (void)hipEventDestroy(e);
  }
  if (h->s_h2d) { (void)hipStreamSynchronize(h->s_h2d); (void)hipStreamDestroy(h->s_h2d); }
  if (h->s_d2h) { (void)hipStreamSynchronize(h->s_d2h); (void)hipStreamDestroy(h->s_d2h); }
  if (h->ev_copy_tail) (void)hipEventDestroy(h->ev_copy_tail);
  void* pinned[] = {h->p_preds, h->p_probs, h->p_logits, h->p_mask, h->h_range};
  for (void* p : pinned) if (p) (void)hipHostFree(p);
  void* bufs[] = {h->d_w32, h->d_w16, h->d_bias, h->d_range};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  if (g_static_handle == h) { g_static_error = nullptr; g_static_handle = nullptr; }
  delete h;
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_num_weights(const pclseg_handle* h) { return h ? (int)h->g.weights.size() : PCLSEG_ERR_BAD_ARG; }

int pclseg_weight_info(const pclseg_handle* h, int index, char* name, size_t name_cap,
                       int64_t shape[4], int* ndim) try {
  if (!h || index < 0 || index >= (int)h->g.weights.size())
    return fail(const_cast<pclseg_handle*>(h), PCLSEG_ERR_BAD_ARG, "bad weight index");
  const WeightInfo& w = h->g.weights[index];
  if (name && name_cap) snprintf(name, name_cap, "%s", w.name.c_str());
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = w.shape[i];
  if (ndim) *ndim = w.ndim;
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_set_weight(pclseg_handle* h, const char* keras_path, const float* data,
                      const int64_t* shape, int ndim) try {
  if (!h || !keras_path || !data || !shape) return fail(h, PCLSEG_ERR_BAD_ARG, "NULL argument");
  if (h->finalized) return fail(h, PCLSEG_ERR_STATE, "set_weight after finalize");
  auto it = h->g.weight_index.find(keras_path);
  if (it == h->g.weight_index.end())
    return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("model has no tensor named '%s'", keras_path));
  const WeightInfo& w = h->g.weights[it->second];
  bool ok = (ndim == w.ndim);
  for (int i = 0; ok && i < ndim; ++i) ok = (shape[i] == w.shape[i]);
  if (!ok) {
    std::string got = "(", want = "(";
    for (int i = 0; i < ndim && i < 8; ++i) got += std::to_string(shape[i]) + (i + 1 < ndim ? "," : "");
    for (int i = 0; i < w.ndim; ++i) want += std::to_string(w.shape[i]) + (i + 1 < w.ndim ? "," : "");
    return fail(h, PCLSEG_ERR_BAD_SHAPE,
                fmt("tensor '%s' has shape %s), expected %s)", keras_path, got.c_str(), want.c_str()));
  }
  h->host_w[it->second].assign(data, data + w.numel());
  h->is_set[it->second] = 1;
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_finalize(pclseg_handle* h) try {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  if (h->finalized) return fail(h, PCLSEG_ERR_STATE, "finalize called twice");
  for (size_t i = 0; i < h->is_set.size(); ++i)
    if (!h->is_set[i])
      return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("tensor '%s' was never set", h->g.weights[i].name.c_str()));
  auto W = [&](const std::string& name) -> const float* {
    auto it = h->g.weight_index.find(name);
    return it == h->g.weight_index.end() ? nullptr : h->host_w[it->second].data();
  };
  const bool want32 = h->exact || h->fallback, want16 = !h->exact;
  std::vector<float> w32(want32 ? (size_t)h->g.packed32_floats : 0, 0.0f);
  std::vector<_Float16> w16(want16 ? (size_t)h->g.packed16_halfs : 0, (_Float16)0.0f);
  std::vector<float> bias((size_t)h->g.packed_bias_floats, 0.0f);
  ScaleStat stat;   // non-finite folded weights seen by the split-f16 packers
  for (const Op& op : h->g.ops) {
    if (op.kind == OP_POOL) continue;
    if (op.sk_in >= 0) {  // fused skip branch: [8][C] folded 1x1 weights (rows >= Cin zero) + [C] bias
      const SubOp& su = op.sk;
      FoldIn f;
      f.kernel = W(su.name + "/kernel"); f.bias = W(su.name + "/bias");
      f.gamma = W(su.bn + "/gamma"); f.beta = W(su.bn + "/beta");
      f.mean = W(su.bn + "/moving_mean"); f.var = W(su.bn + "/moving_variance");
      if (!f.kernel || !f.bias || !f.gamma || !f.beta || !f.mean || !f.var)
        return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
      std::vector<double> scale, shift;
      fold_bn(su, f, &scale, &shift);
      const int cin = (int)h->g.weights[h->g.weight_index[su.name + "/kernel"]].shape[2];
      float* dst = bias.data() + su.b_off;
      for (int ci = 0; ci < 8; ++ci)
        for (int co = 0; co < su.cout; ++co)
          dst[(size_t)ci * su.cout + co] =
              ci < cin ? (float)((double)f.kernel[(size_t)ci * su.cout + co] * scale[co]) : 0.0f;
      for (int co = 0; co < su.cout; ++co) dst[(size_t)8 * su.cout + co] = (float)shift[co];
    }
    if (op.fsq_fused) {
      const SubOp& su = op.fsq;
      FoldIn f;
      f.kernel = W(su.name + "/kernel"); f.bias = W(su.name + "/bias");
      f.gamma = W(su.bn + "/gamma"); f.beta = W(su.bn + "/beta");
      f.mean = W(su.bn + "/moving_mean"); f.var = W(su.bn + "/moving_variance");
      if (!f.kernel || !f.bias || !f.gamma || !f.beta || !f.mean || !f.var)
        return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
      std::vector<double> scale, shift;
      fold_bn(su, f, &scale, &shift);
      pack_bias(su, shift, bias.data() + su.b_off);
      if (op.kind == OP_CAM) {   // cam_kernel SQ: the fragments of a plain 1x1 conv over C channels, 64-channel chunks
        Op as_conv;
        as_conv.cin_t = as_conv.cin_k = op.cin_t;
        as_conv.ck16 = 64;
        pack_w16(as_conv, su, f, scale, w16.data() + su.w16_off, bias.data() + su.b_off + su.nctp * 16, &stat);
      } else {
        pack_fsq(op, f, scale, w16.data() + su.w16_off, bias.data() + su.b_off + su.nctp * 16, &stat);
      }
    }
    if (op.head_fused) {   // conv14 as one 64-channel chunk (fire_head_kernel reads (tap, 8-channel group) pairs 4 st + g)
      if (!want16) return fail(h, PCLSEG_ERR_STATE, "internal: fused head in an exact-f32 plan");
      const SubOp& su = op.hd;
      FoldIn f;
      f.kernel = W(su.name + "/kernel"); f.bias = W(su.name + "/bias");
      if (!f.kernel || !f.bias)
        return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
      Op as_head;
      as_head.kind = OP_HEAD;
      as_head.cin_t = as_head.cin_k = 64;
      as_head.ck16 = 64;
      std::vector<double> scale, shift;
      fold_bn(su, f, &scale, &shift);
      pack_bias(su, shift, bias.data() + su.b_off);
      pack_w16(as_head, su, f, scale, w16.data() + su.w16_off, bias.data() + su.b_off + su.nctp * 16, &stat);
      // the skip branch (conv1_skip + bn1_skip, 6 -> 64) as ONE K-step of split-f16 fragments: its 8 (padded)
      // input channels are K-group 0, the other three lane groups carry zeros
      const SubOp& sm = op.skm;
      FoldIn fs;
      fs.kernel = W(sm.name + "/kernel"); fs.bias = W(sm.name + "/bias");
      fs.gamma = W(sm.bn + "/gamma"); fs.beta = W(sm.bn + "/beta");
      fs.mean = W(sm.bn + "/moving_mean"); fs.var = W(sm.bn + "/moving_variance");
      if (!fs.kernel || !fs.bias || !fs.gamma || !fs.beta || !fs.mean || !fs.var)
        return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", sm.name.c_str()));
      Op as_skip;
      as_skip.cin_t = 8;
      as_skip.cin_k = (int)h->g.weights[h->g.weight_index[sm.name + "/kernel"]].shape[2];
      as_skip.ck16 = 16;
      fold_bn(sm, fs, &scale, &shift);
      pack_bias(sm, shift, bias.data() + sm.b_off);
      pack_w16(as_skip, sm, fs, scale, w16.data() + sm.w16_off, bias.data() + sm.b_off + sm.nctp * 16, &stat);
    }
    if (op.up_fused) {
      if (!want16) return fail(h, PCLSEG_ERR_STATE, "internal: fused up-convolution in an exact-f32 plan");
      for (int i = 0; i < 2; ++i) {
        const SubOp& su = op.up[i];
        FoldIn f;
        f.kernel = W(su.name + "/kernel"); f.bias = W(su.name + "/bias");
        if (!f.kernel || !f.bias)
          return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
        std::vector<double> scale, shift;
        fold_bn(su, f, &scale, &shift);
        pack_bias(su, shift, bias.data() + su.b_off);
        pack_w16(op, su, f, scale, w16.data() + su.w16_off, bias.data() + su.b_off + su.nctp * 16, &stat);
      }
    }
    for (int i = 0; i < op.nsub; ++i) {
      const SubOp& su = op.sub[i];
      FoldIn f;
      if (op.kind == OP_CAM) {  // plain row-major [Cin][Cout] * scale, then shift
        f.kernel = W(su.name + "/kernel"); f.bias = W(su.name + "/bias");
        f.gamma = W(su.bn + "/gamma"); f.beta = W(su.bn + "/beta");
        f.mean = W(su.bn + "/moving_mean"); f.var = W(su.bn + "/moving_variance");
        if (!f.kernel || !f.bias || !f.gamma || !f.beta || !f.mean || !f.var)
          return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
        std::vector<double> scale, shift;
        fold_bn(su, f, &scale, &shift);
        const int cin = i == 0 ? op.cin_t : op.cin_t / 16;
        float* dst = bias.data() + su.b_off;
        for (int ci = 0; ci < cin; ++ci)
          for (int co = 0; co < su.cout; ++co)
            dst[(size_t)ci * su.cout + co] = (float)((double)f.kernel[(size_t)ci * su.cout + co] * scale[co]);
        for (int co = 0; co < su.cout; ++co) dst[(size_t)cin * su.cout + co] = (float)shift[co];
        continue;
      }
      f.kernel = W(su.name + "/kernel");
      f.bias = su.has_bias ? W(su.name + "/bias") : nullptr;
      if (!su.bn.empty()) {
        f.gamma = W(su.bn + "/gamma");
        f.beta = W(su.bn + "/beta");
        f.mean = W(su.bn + "/moving_mean");
        f.var = W(su.bn + "/moving_variance");
      }
      if (!f.kernel || (su.has_bias && !f.bias) || (!su.bn.empty() && !(f.gamma && f.beta && f.mean && f.var)))
        return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", su.name.c_str()));
      std::vector<double> scale, shift;
      fold_bn(su, f, &scale, &shift);
      pack_bias(su, shift, bias.data() + su.b_off);
      if (want32) pack_w32(op, su, f, scale, w32.data() + su.w32_off);
      if (want16) pack_w16(op, su, f, scale, w16.data() + su.w16_off, bias.data() + su.b_off + su.nctp * 16, &stat);
    }
  }
  if (want16 && stat.nonfinite) {
    // inf / NaN after BatchNorm folding: float32 arithmetic would propagate it, the hi/lo split cannot
    // (hi = inf, lo = inf - inf = NaN)
    if (!h->fallback)
      return fail(h, PCLSEG_ERR_RANGE, "a BatchNorm-folded weight is not finite: the split-f16 fragments cannot carry it; "
                                       "create the handle with PCLSEG_FLAG_EXACT_F32 or PCLSEG_FLAG_RANGE_FALLBACK");
    h->force_exact = true;
  }
  DeviceGuard guard(h->device);
  if (want32) HIP_TRY(h, hipMemcpy(h->d_w32, w32.data(), w32.size() * sizeof(float), hipMemcpyHostToDevice));
  if (want16) HIP_TRY(h, hipMemcpy(h->d_w16, w16.data(), w16.size() * sizeof(_Float16), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  h->finalized = true;
  // the Keras-layout copies are no longer needed
  for (auto& v : h->host_w) std::vector<float>().swap(v);
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_packed_size(const pclseg_handle* h, size_t* bytes) try {
  if (!h || !bytes) return fail(const_cast<pclseg_handle*>(h), PCLSEG_ERR_BAD_ARG, "NULL argument");
  *bytes = packed_bytes(packed_header(h));
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_export_packed(pclseg_handle* h, void* dst, size_t capacity, int mem) try {
  if (!h || !dst) return fail(h, PCLSEG_ERR_BAD_ARG, "NULL argument");
  if (!h->finalized) return fail(h, PCLSEG_ERR_STATE, "export_packed before pclseg_finalize");
  if (mem != PCLSEG_MEM_HOST && mem != PCLSEG_MEM_DEVICE) return fail(h, PCLSEG_ERR_BAD_ARG, fmt("unknown mem %d", mem));
  const PackedHeader ph = packed_header(h);
  if (capacity < packed_bytes(ph))
    return fail(h, PCLSEG_ERR_BAD_ARG, fmt("buffer of %zu bytes, the packed parameters need %zu", capacity, packed_bytes(ph)));
  DeviceGuard guard(h->device);
  const hipMemcpyKind head = mem == PCLSEG_MEM_HOST ? hipMemcpyHostToHost : hipMemcpyHostToDevice;
  const hipMemcpyKind body = mem == PCLSEG_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  char* p = (char*)dst;
  HIP_TRY(h, hipMemcpy(p, &ph, sizeof(ph), head));
  p += sizeof(ph);
  HIP_TRY(h, hipMemcpy(p, h->d_bias, (size_t)ph.n_bias * 4, body));
  p += (size_t)ph.n_bias * 4;
  if (ph.n_w16) HIP_TRY(h, hipMemcpy(p, h->d_w16, (size_t)ph.n_w16 * 2, body));
  p += (size_t)ph.n_w16 * 2;
  if (ph.n_w32) HIP_TRY(h, hipMemcpy(p, h->d_w32, (size_t)ph.n_w32 * 4, body));
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_import_packed(pclseg_handle* h, const void* src, size_t bytes, int mem) try {
  if (!h || !src) return fail(h, PCLSEG_ERR_BAD_ARG, "NULL argument");
  if (h->finalized) return fail(h, PCLSEG_ERR_STATE, "import_packed on a finalized handle");
  if (mem != PCLSEG_MEM_HOST && mem != PCLSEG_MEM_DEVICE) return fail(h, PCLSEG_ERR_BAD_ARG, fmt("unknown mem %d", mem));
  if (bytes < sizeof(PackedHeader)) return fail(h, PCLSEG_ERR_BAD_ARG, "blob shorter than its header");
  DeviceGuard guard(h->device);
  PackedHeader got;
  HIP_TRY(h, hipMemcpy(&got, src, sizeof(got), mem == PCLSEG_MEM_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost));
  PackedHeader want = packed_header(h);
  want.math = (want.math & 3u) | (got.math & 4u);
  if (got.magic != kPackedMagic) return fail(h, PCLSEG_ERR_BAD_ARG, "not a packed-parameter blob (bad magic)");
  if (memcmp(&got, &want, sizeof(got)) != 0)
    return fail(h, PCLSEG_ERR_BAD_SHAPE,
                fmt("packed blob does not fit this handle: blob arch %d %dx%d NC %d stride %d math %u version %u "
                    "(%lld bias, %lld f16, %lld f32 scalars, plan %016llx); handle arch %d %dx%d NC %d stride %d math %u version %u "
                    "(%lld, %lld, %lld, plan %016llx); debug switches resolved in THIS process: %s "
                    "(the exporting process must run the same build with the same PCLSEG_FUSE_* environment)",
                    got.arch, got.height, got.width, got.num_class, got.output_stride, got.math, got.version,
                    (long long)got.n_bias, (long long)got.n_w16, (long long)got.n_w32, (unsigned long long)got.plan_hash,
                    want.arch, want.height, want.width, want.num_class, want.output_stride, want.math, want.version,
                    (long long)want.n_bias, (long long)want.n_w16, (long long)want.n_w32, (unsigned long long)want.plan_hash,
                    DebugSwitches::text().c_str()));
  if (bytes < packed_bytes(got)) return fail(h, PCLSEG_ERR_BAD_ARG, "blob truncated");
  const hipMemcpyKind body = mem == PCLSEG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const char* p = (const char*)src + sizeof(got);
  HIP_TRY(h, hipMemcpy(h->d_bias, p, (size_t)got.n_bias * 4, body));
  p += (size_t)got.n_bias * 4;
  if (got.n_w16) HIP_TRY(h, hipMemcpy(h->d_w16, p, (size_t)got.n_w16 * 2, body));
  p += (size_t)got.n_w16 * 2;
  if (got.n_w32) HIP_TRY(h, hipMemcpy(h->d_w32, p, (size_t)got.n_w32 * 4, body));
  h->force_exact = (got.math & 4u) != 0;
  h->finalized = true;
  for (auto& v : h->host_w) std::vector<float>().swap(v);
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_set_stream(pclseg_handle* h, void* hip_stream) try {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  h->stream = (hipStream_t)hip_stream;
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_sync(pclseg_handle* h) try {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->host_async_pending = false;
  bool fired = false;
  int rc = take_range_flag(h, &fired);
  if (rc) return rc;
  if (!fired) { clear_pending(h); return PCLSEG_OK; }
  if (!h->fallback) { clear_pending(h); return fail(h, PCLSEG_ERR_RANGE, kRangeMsg); }
  return repair_pending(h);
} PCLSEG_CATCH(h)

void* pclseg_host_alloc(size_t bytes) try {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    g_last_error = fmt("hipHostMalloc(%zu) failed", bytes);
    return nullptr;
  }
  return p;
} catch (...) { (void)on_exception(nullptr); return nullptr; }

int pclseg_host_free(void* p) try {
  if (!p) return PCLSEG_OK;
  HIP_TRY(nullptr, hipHostFree(p));
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_forward(pclseg_handle* h, const float* lidar, const uint8_t* mask, int n, int32_t* preds,
                   float* probs, float* logits, int mem) try {
  return forward_impl(h, lidar, false, mask, n, preds, probs, logits, nullptr, mem);
} PCLSEG_CATCH(h)

int pclseg_forward_raw(pclseg_handle* h, const float* scans, int n, int32_t* preds, float* probs,
                       float* logits, uint8_t* mask_out, int mem) try {
  return forward_impl(h, scans, true, nullptr, n, preds, probs, logits, mask_out, mem);
} PCLSEG_CATCH(h)

int pclseg_num_tensors(const pclseg_handle* h) { return h ? (int)h->g.tensors.size() : PCLSEG_ERR_BAD_ARG; }

int pclseg_tensor_info(const pclseg_handle* h, int index, char* name, size_t name_cap, int64_t shape[4]) try {
  if (!h || index < 0 || index >= (int)h->g.tensors.size())
    return fail(const_cast<pclseg_handle*>(h), PCLSEG_ERR_BAD_ARG, "bad tensor index");
  const TensorInfo& t = h->g.tensors[index];
  if (name && name_cap) snprintf(name, name_cap, "%s", t.name.c_str());
  if (shape) { shape[0] = h->last_count; shape[1] = t.H; shape[2] = t.W; shape[3] = t.C; }
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

int pclseg_read_tensor(pclseg_handle* h, int index, float* host_out, size_t capacity_floats) try {
  if (!h || !host_out || index < 0 || index >= (int)h->g.tensors.size())
    return fail(h, PCLSEG_ERR_BAD_ARG, "bad argument to read_tensor");
  const TensorInfo& t = h->g.tensors[index];
  const size_t nfl = (size_t)h->last_count * t.scan_floats();
  if (capacity_floats < nfl) return fail(h, PCLSEG_ERR_BAD_ARG, "host buffer too small");
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(host_out, h->d_arena + t.offset, nfl * sizeof(float), hipMemcpyDeviceToHost));
  if (t.fmt == FMT_S16 && !h->last_exact) {
    // split-f16 pair format: per pixel C halfs hi then C halfs lo in the same 4*C bytes -> hi + lo
    const size_t npx = nfl / t.C;
    std::vector<_Float16> px(2 * (size_t)t.C);
    for (size_t i = 0; i < npx; ++i) {
      memcpy(px.data(), host_out + i * t.C, 4 * (size_t)t.C);
      for (int c = 0; c < t.C; ++c) host_out[i * t.C + c] = (float)px[c] + (float)px[t.C + c];
    }
  }
  return PCLSEG_OK;
} PCLSEG_CATCH(h)

// ---- single-operator entry points
int pclseg_op_normalize(const float* scans, int n, int h, int w, const double mean[5],
                        const double std[5], float* lidar6, uint8_t* mask) try {
  if (!scans || !lidar6 || !mean || !std || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_normalize");
  NormArgs na;
  for (int i = 0; i < 5; ++i) { na.mean[i] = mean[i]; na.std[i] = std[i]; }
  const size_t P = (size_t)n * h * w;
  hipLaunchKernelGGL(normalize_kernel<6>, dim3(stream_blocks(P)), dim3(256), 0, nullptr, scans, lidar6,
                     mask, P, na);
  HIP_TRY(nullptr, hipGetLastError());
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_op_conv2d(const float* x, int n, int h, int w, int cin, const float* kernel, int kh, int kw,
                     int cout, int stride_w, const float* bias, const float* bn_gamma,
                     const float* bn_beta, const float* bn_mean, const float* bn_var, int act,
                     const float* residual, float* y, int math) try {
  if (!x || !kernel || !y || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_conv2d");
  if (cin % 4 || cout % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin and Cout must be multiples of 4");
  if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3)) || (stride_w != 1 && stride_w != 2) || act < 0 || act > 3)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "unsupported kernel size / stride / activation");
  Op op;
  op.kind = OP_CONV;
  op.cin_t = op.cin_k = cin;
  op.pkh = kh; op.pkw = kw; op.sw = stride_w;
  op.sub[0].cout = cout;
  op.sub[0].nkh = kh; op.sub[0].nkw = kw;
  op.sub[0].act = act;
  FoldIn f;
  f.kernel = kernel; f.bias = bias;
  f.gamma = bn_gamma; f.beta = bn_beta; f.mean = bn_mean; f.var = bn_var;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = y; a.out_C = cout;
  if (residual) { a.res1 = residual; a.res1_C = cout; }
  return run_single_op(&op, &f, n, h, w, a, math);
} PCLSEG_CATCH(nullptr)

int pclseg_op_conv2d_transpose(const float* x, int n, int h, int w, int cin, const float* kernel,
                               int cout, const float* bias, const float* bn_gamma,
                               const float* bn_beta, const float* bn_mean, const float* bn_var,
                               int act, float* y, int math) try {
  if (!x || !kernel || !y || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_conv2d_transpose");
  if (cin % 4 || cout % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin and Cout must be multiples of 4");
  if (act < 0 || act > 3) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "unsupported activation");
  Op op;
  op.kind = OP_CONV;
  op.cin_t = op.cin_k = cin;
  op.pkh = 1; op.pkw = 3; op.sw = 1; op.pl_fixed = 1; op.ow_mul = 2;
  op.nsub = 2;
  FoldIn f[2];
  for (int parity = 0; parity < 2; ++parity) {
    SubOp& s = op.sub[parity];
    s.deconv = true;
    s.cout = cout;
    s.nkh = 1; s.nkw = 2; s.tw0 = parity;
    s.ktap[0] = parity == 0 ? 3 : 2;
    s.ktap[1] = parity == 0 ? 1 : 0;
    s.ow_off = parity;
    s.act = act;
    f[parity].kernel = kernel; f[parity].bias = bias;
    f[parity].gamma = bn_gamma; f[parity].beta = bn_beta; f[parity].mean = bn_mean; f[parity].var = bn_var;
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = y; a.out_C = cout;
  return run_single_op(&op, f, n, h, w, a, math);
} PCLSEG_CATCH(nullptr)

int pclseg_op_max_pool(const float* x, int n, int h, int w, int c, int k, int stride_w, float* y) try {
  if (!x || !y || n <= 0 || h <= 0 || w <= 0 || k <= 0 || (stride_w != 1 && stride_w != 2))
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_max_pool");
  if (c % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "C must be a multiple of 4");
  if (k > 3 && stride_w == 1) {
    // separable: rows (1xk) then columns (kx1) — bit-identical to the k x k window (max is associative)
    DevBuf tmp;
    HIP_TRY(nullptr, hipMalloc(&tmp.p, (size_t)n * h * w * c * sizeof(float)));
    HIP_TRY(nullptr, launch_pool(x, (float*)tmp.p, n, h, w, c, 1, k, 1, nullptr));
    HIP_TRY(nullptr, launch_pool((const float*)tmp.p, y, n, h, w, c, k, 1, 1, nullptr));
    HIP_TRY(nullptr, hipDeviceSynchronize());
    return PCLSEG_OK;
  }
  HIP_TRY(nullptr, launch_pool(x, y, n, h, w, c, k, k, stride_w, nullptr));
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_op_head(const float* x, const uint8_t* mask, int n, int h, int w, int cin,
                   const float* kernel, const float* bias, int num_class, int none_index,
                   int32_t* preds, float* probs, float* logits, int math) try {
  if (!x || !mask || !kernel || !bias || !preds || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_head");
  if (cin % 4 || num_class < 2 || num_class > 64)
    return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin must be a multiple of 4 and num_class in [2,64]");
  Op op;
  op.kind = OP_HEAD;
  op.cin_t = op.cin_k = cin;
  op.pkh = op.pkw = 3; op.sw = 1;
  op.sub[0].cout = num_class;
  op.sub[0].nkh = op.sub[0].nkw = 3;
  FoldIn f;
  f.kernel = kernel; f.bias = bias;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x;
  a.mask = mask; a.preds = preds; a.probs = probs; a.logits = logits; a.none_index = none_index;
  return run_single_op(&op, &f, n, h, w, a, math);
} PCLSEG_CATCH(nullptr)

int pclseg_op_split_f16_roundtrip(const float* kernel, int kh, int kw, int cin, int cout, double* recon,
                                  int32_t* exponents) try {
  if (!kernel || !recon || cin <= 0 || cout <= 0 || !((kh == 1 && kw == 1) || (kh == 3 && kw == 3)))
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_split_f16_roundtrip");
  if (cin % 4 || cout % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin and Cout must be multiples of 4");
  Op op;
  op.kind = OP_CONV;
  op.cin_t = op.cin_k = cin;
  op.pkh = kh; op.pkw = kw; op.sw = 1;
  SubOp& su = op.sub[0];
  su.cout = cout; su.nkh = kh; su.nkw = kw;
  op_geometry(&op);
  FoldIn f;
  f.kernel = kernel;
  std::vector<double> scale, shift;
  fold_bn(su, f, &scale, &shift);
  std::vector<_Float16> w16((size_t)sub_w16_halfs(op, su));
  std::vector<float> inv((size_t)su.nctp * 16);
  ScaleStat st;
  pack_w16(op, su, f, scale, w16.data(), inv.data(), &st);
  if (st.nonfinite) return fail(nullptr, PCLSEG_ERR_RANGE, "non-finite weight");
  // walk the fragments exactly as the kernels address them and undo the scale
  const int taps = kh * kw, cin8 = (cin + 7) / 8, ck8_full = op.ck16 / 8;
  const int steps_full = f16_steps_full(op, su), nchunks = f16_chunks(op);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int ck8 = std::min(ck8_full, cin8 - chunk * ck8_full), nk = taps * ck8;
    for (int s = 0; s < steps_full; ++s)
      for (int ct = 0; ct < su.nctp; ++ct) {
        const _Float16* blk = w16.data() + ((size_t)(chunk * steps_full + s) * su.nctp + ct) * 1024;
        for (int lane = 0; lane < 64; ++lane) {
          const int kidx = 4 * s + (lane >> 4), co = ct * 16 + (lane & 15);
          if (kidx >= nk || co >= cout) continue;
          const int tap = kidx / ck8, c8 = kidx % ck8;
          for (int j = 0; j < 8; ++j) {
            const int ci = (chunk * ck8_full + c8) * 8 + j;
            if (ci >= cin) continue;
            recon[((size_t)tap * cin + ci) * cout + co] =
                ((double)(float)blk[lane * 8 + j] + (double)(float)blk[512 + lane * 8 + j]) * (double)inv[co];
          }
        }
      }
  }
  if (exponents)
    for (int co = 0; co < cout; ++co) exponents[co] = -std::ilogb(inv[co]);
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_op_confusion_matrix(const int32_t* labels, const int32_t* preds, size_t count,
                               int num_class, int64_t* cm, void* hip_stream) try {
  if (!labels || !preds || !cm || num_class < 1 || num_class > 64)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_confusion_matrix");
  if (count == 0) return PCLSEG_OK;
  const unsigned blocks = (unsigned)std::min<size_t>((count + 4095) / 4096, 1024);
  hipLaunchKernelGGL(confusion_kernel, dim3(blocks), dim3(256), (size_t)num_class * num_class * sizeof(unsigned int),
                     (hipStream_t)hip_stream, labels, preds, count, num_class,
                     reinterpret_cast<unsigned long long*>(cm));
  HIP_TRY(nullptr, hipGetLastError());
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_op_project_ex(const pclseg_proj_desc* d, const float* points, int point_stride, size_t m,
                         const int32_t* ring, const float* depth, const int32_t* labels,
                         const int32_t* label_lut, int lut_size, float* image, int32_t* proj_idx,
                         uint64_t* scratch, void* hip_stream) try {
  if (!d || !points || !image || !scratch || d->h <= 0 || d->w <= 0 || m > 0x7ffffffeull || point_stride < 4)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_project_ex");
  if (d->row_mode < 0 || d->row_mode > 1 || d->col_mode < 0 || d->col_mode > 1 || d->winner < 0 || d->winner > 1 ||
      d->out_channels < 5 || d->out_channels > 7)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "op_project_ex: unknown mode or out_channels not in [5,7]");
  if (d->row_mode == PCLSEG_PROJ_ROW_RING && !ring)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "op_project_ex: ring rows need the ring index array");
  if (label_lut && lut_size <= 0) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "op_project_ex: empty label table");
  ProjArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.H = d->h; pa.W = d->w;
  pa.row_mode = d->row_mode; pa.col_mode = d->col_mode; pa.winner = d->winner; pa.out_c = d->out_channels;
  if (d->row_mode == PCLSEG_PROJ_ROW_FOV) {
    const double up = (double)d->fov_up / 180.0 * M_PI, down = (double)d->fov_down / 180.0 * M_PI;
    const double fov = std::fabs(down) + std::fabs(up);
    if (!(fov > 0.0)) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "field of view must be positive");
    pa.fdown = (float)std::fabs(down); pa.ffov = (float)fov;
  }
  pa.fpi = (float)M_PI;
  if (d->col_mode == PCLSEG_PROJ_COL_FRONT) {
    pa.left_phi = d->left_phi;
    pa.dphi = (d->right_phi + d->left_phi) / (double)d->w;
    if (!(pa.dphi > 0.0)) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "front window must have a positive width");
  }
  pa.stride = point_stride;
  pa.ring = ring; pa.depth = depth; pa.labels = labels; pa.lut = label_lut; pa.lut_size = lut_size;
  pa.empty = d->empty;
  hipStream_t s = (hipStream_t)hip_stream;
  const int npix = d->h * d->w;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(scratch);
  hipLaunchKernelGGL(proj_init_kernel, dim3(stream_blocks((size_t)npix)), dim3(256), 0, s, keys, npix,
                     d->winner == PCLSEG_PROJ_NEAREST ? ~0ull : 0ull);
  if (m) hipLaunchKernelGGL(proj_scatter_kernel, dim3(stream_blocks(m)), dim3(256), 0, s, points, m, keys, pa);
  hipLaunchKernelGGL(proj_gather_kernel, dim3(stream_blocks((size_t)npix)), dim3(256), 0, s, points, keys,
                     npix, image, proj_idx, pa);
  HIP_TRY(nullptr, hipGetLastError());
  return PCLSEG_OK;
} PCLSEG_CATCH(nullptr)

int pclseg_op_project(const float* points, size_t m, int h, int w, float fov_up, float fov_down,
                      float empty, float* image5, int32_t* proj_idx, uint64_t* scratch,
                      void* hip_stream) try {
  pclseg_proj_desc d;
  memset(&d, 0, sizeof(d));
  d.h = h; d.w = w;
  d.row_mode = PCLSEG_PROJ_ROW_FOV; d.col_mode = PCLSEG_PROJ_COL_FULL; d.winner = PCLSEG_PROJ_NEAREST;
  d.out_channels = 5;
  d.fov_up = fov_up; d.fov_down = fov_down; d.empty = empty;
  return pclseg_op_project_ex(&d, points, 4, m, nullptr, nullptr, nullptr, nullptr, 0, image5, proj_idx,
                              scratch, hip_stream);
} PCLSEG_CATCH(nullptr)

}  // extern "C"
