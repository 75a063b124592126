// pclseg_api.hip — the engine behind include/pclseg.h: weight folding/packing, workspace,
// kernel sequencing on one HIP stream, and the C ABI.  gfx950 only; there is no CPU path.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/pclseg.h"
#include "pclseg_graph.h"
#include "pclseg_kernels.h"

using namespace pclseg;

namespace {

thread_local std::string g_last_error;

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
  float* d_params = nullptr;
  float* d_arena = nullptr;
  uint8_t* d_mask = nullptr;  // micro-batch mask when the caller gives none
  // host-mode staging (grown on demand)
  void* d_stage_in = nullptr;   size_t stage_in_bytes = 0;
  uint8_t* d_stage_mask = nullptr; size_t stage_mask_bytes = 0;
  int32_t* d_stage_preds = nullptr; size_t stage_preds_bytes = 0;
  float* d_stage_probs = nullptr; size_t stage_probs_bytes = 0;
  float* d_stage_logits = nullptr; size_t stage_logits_bytes = 0;
  int last_count = 0;  // scans held by the arena after the last forward
  std::string err;
};

namespace {

int fail(pclseg_handle* h, int code, const std::string& msg) {
  g_last_error = msg;
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, e_ == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,         \
                  fmt("%s failed: %s", #expr, hipGetErrorString(e_)));                     \
  } while (0)

// ---- BatchNorm folding + fragment packing -------------------------------------------------
// wpk[((t*nc16 + c16)*nctp + ct)*256 + lane*4 + j] = K[tap t][ci = 16*c16 + 4*(lane>>4) + j]
//                                                     [co = 16*ct + (lane&15)] * bn_scale[co]
struct FoldIn {
  const float* kernel = nullptr;  // Keras layout
  const float* bias = nullptr;
  const float* gamma = nullptr, *beta = nullptr, *mean = nullptr, *var = nullptr;
};

void pack_op(const Op& op, const FoldIn& f, float* wdst, float* bdst) {
  const int taps = op.kh * op.kw;
  const int cin = op.cin_k, cout = op.cout;
  std::vector<double> scale(cout, 1.0), shift(cout, 0.0);
  for (int co = 0; co < cout; ++co) {
    double b = f.bias ? (double)f.bias[co] : 0.0;
    if (f.gamma) {
      const double s = (double)f.gamma[co] / std::sqrt((double)f.var[co] + kBnEps);
      scale[co] = s;
      shift[co] = (b - (double)f.mean[co]) * s + (double)f.beta[co];
    } else {
      shift[co] = b;
    }
  }
  for (int i = 0; i < op.nctp * 16; ++i) bdst[i] = i < cout ? (float)shift[i] : 0.0f;
  for (int t = 0; t < taps; ++t) {
    for (int c16 = 0; c16 < op.nc16; ++c16)
      for (int ct = 0; ct < op.nctp; ++ct) {
        float* blk = wdst + ((size_t)(t * op.nc16 + c16) * op.nctp + ct) * 256;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int ci = c16 * 16 + 4 * (lane >> 4) + j;
            const int co = ct * 16 + (lane & 15);
            float v = 0.0f;
            if (ci < cin && co < cout) {
              double k;
              if (op.kind == OP_DECONV) {
                // Keras Conv2DTranspose kernel (1,4,Cout,Cin); o = 2i + k - 1:
                //   even o = 2j:   x[j-1]*K[3] (tap 0, patch col j-1) + x[j]*K[1] (tap 1)
                //   odd  o = 2j+1: x[j]*K[2]   (tap 0, patch col j)   + x[j+1]*K[0] (tap 1)
                const int parity = op.sw;
                const int kk = parity == 0 ? (t == 0 ? 3 : 1) : (t == 0 ? 2 : 0);
                k = f.kernel[((size_t)kk * cout + co) * cin + ci];
              } else {
                k = f.kernel[((size_t)t * cin + ci) * cout + co];
              }
              v = (float)(k * scale[co]);
            }
            blk[lane * 4 + j] = v;
          }
      }
  }
}

// ---- launch helpers -------------------------------------------------------------------------
struct ConvGeom {
  int N, H, Win;  // input tensor
};

template <bool HEAD>
hipError_t launch_conv_nt(int nt, dim3 grid, size_t lds, hipStream_t s, const ConvArgs& a) {
  switch (nt) {
    case 1: hipLaunchKernelGGL((conv_mfma_kernel<1, HEAD>), grid, dim3(kConvThreads), lds, s, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_kernel<2, HEAD>), grid, dim3(kConvThreads), lds, s, a); break;
    case 3: hipLaunchKernelGGL((conv_mfma_kernel<3, HEAD>), grid, dim3(kConvThreads), lds, s, a); break;
    case 4: hipLaunchKernelGGL((conv_mfma_kernel<4, HEAD>), grid, dim3(kConvThreads), lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// Fill geometry fields of `a` (pointers and channel bookkeeping already set) and launch.
hipError_t launch_conv(const Op& op, int N, int H, int Win, ConvArgs a, hipStream_t s) {
  a.Cin = op.cin_t;
  a.nc16 = op.nc16;
  a.Cout = op.cout;
  a.nctp = op.nctp;
  a.act = op.act;
  a.res1_mul = op.res1_mul ? 1 : 0;
  const bool flat = (op.kind == OP_CONV && op.kh == 1 && op.kw == 1 && op.sw == 1);
  if (op.kind == OP_DECONV) {
    a.KH = 1; a.KW = 2; a.sw = 1; a.pt = 0; a.pl = (op.sw == 0) ? 1 : 0;
    a.ow_mul = 2; a.ow_off = op.sw;
    a.N = N; a.H = H; a.Win = Win; a.Wconv = Win; a.Wout = 2 * Win;
  } else {
    int wo, pl, ho, pt;
    same_pad(Win, op.kw, op.sw, &wo, &pl);
    same_pad(H, op.kh, 1, &ho, &pt);
    a.KH = op.kh; a.KW = op.kw; a.sw = op.sw; a.pt = pt; a.pl = pl;
    a.ow_mul = 1; a.ow_off = 0;
    if (flat) {
      a.N = 1; a.H = 1; a.Win = N * H * Win; a.Wconv = a.Wout = a.Win;
    } else {
      a.N = N; a.H = H; a.Win = Win; a.Wconv = a.Wout = wo;
    }
  }
  if (flat) { a.TH = 1; a.SEGW = kSegsPerBlock; }
  else { a.TH = kSegsPerBlock; a.SEGW = 1; }
  a.PH = a.TH + a.KH - 1;
  a.PW = (a.SEGW * 16 - 1) * a.sw + a.KW;
  a.tilesH = (a.H + a.TH - 1) / a.TH;
  a.tilesW = (a.Wconv + a.SEGW * 16 - 1) / (a.SEGW * 16);
  const int cinp = a.nc16 * 16;
  const int CS = (cinp < kChunk ? cinp : kChunk) + 4;
  const size_t lds = (size_t)a.PH * a.PW * CS * sizeof(float);
  dim3 grid((unsigned)(a.N * a.tilesH * a.tilesW), (unsigned)(a.nctp / op.nt));
  if (op.kind == OP_HEAD) return launch_conv_nt<true>(op.nt, grid, lds, s, a);
  return launch_conv_nt<false>(op.nt, grid, lds, s, a);
}

hipError_t launch_pool(const float* in, float* out, int N, int H, int Win, int C, int k, int sw,
                       hipStream_t s) {
  int wo, pl, ho, pt;
  same_pad(Win, k, sw, &wo, &pl);
  same_pad(H, k, 1, &ho, &pt);
  const size_t total = (size_t)N * H * wo * (C / 4);
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, in, out, N, H, Win, wo, C, k, sw,
                     pt, pl);
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
int run_ops(pclseg_handle* h, int cnt, const uint8_t* mask, int32_t* preds, float* probs,
            float* logits) {
  const Graph& g = h->g;
  const float* P = h->d_params;
  for (const Op& op : g.ops) {
    const TensorInfo& ti = g.tensors[op.in];
    const float* in = h->d_arena + ti.offset;
    if (op.kind == OP_POOL) {
      float* out = h->d_arena + g.tensors[op.out].offset;
      HIP_TRY(h, launch_pool(in, out, cnt, ti.H, ti.W, ti.C, op.kh, op.sw, h->stream));
      continue;
    }
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.wpk = P + op.w_off;
    a.bias = P + op.b_off;
    if (op.kind == OP_HEAD) {
      a.mask = mask;
      a.preds = preds;
      a.probs = probs;
      a.logits = logits;
      a.none_index = g.desc.none_index;
    } else {
      const TensorInfo& to = g.tensors[op.out];
      a.out = h->d_arena + to.offset;
      a.out_C = to.C;
      a.co_off = op.co_off;
      if (op.res1 >= 0) { a.res1 = h->d_arena + g.tensors[op.res1].offset; a.res1_C = g.tensors[op.res1].C; }
      if (op.res2 >= 0) { a.res2 = h->d_arena + g.tensors[op.res2].offset; a.res2_C = g.tensors[op.res2].C; }
    }
    HIP_TRY(h, launch_conv(op, cnt, ti.H, ti.W, a, h->stream));
  }
  return PCLSEG_OK;
}

int forward_impl(pclseg_handle* h, const float* input, bool raw, const uint8_t* mask_in, int n,
                 int32_t* preds, float* probs, float* logits, uint8_t* mask_out, int mem) {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  if (!h->finalized) return fail(h, PCLSEG_ERR_STATE, "forward called before pclseg_finalize");
  if (!input || !preds) return fail(h, PCLSEG_ERR_BAD_ARG, "input and preds must not be NULL");
  if (!raw && !mask_in) return fail(h, PCLSEG_ERR_BAD_ARG, "mask must not be NULL");
  if (n <= 0) return fail(h, PCLSEG_ERR_BAD_ARG, fmt("n must be positive, got %d", n));
  if (mem != PCLSEG_MEM_HOST && mem != PCLSEG_MEM_DEVICE)
    return fail(h, PCLSEG_ERR_BAD_ARG, fmt("unknown mem %d", mem));
  HIP_TRY(h, hipSetDevice(h->device));
  const Graph& g = h->g;
  const size_t HW = (size_t)g.desc.height * g.desc.width;
  const int NC = g.desc.num_class;
  const int cin = raw ? 5 : 6;

  const float* d_in = input;
  const uint8_t* d_mask_in = mask_in;
  int32_t* d_preds = preds;
  float* d_probs = probs;
  float* d_logits = logits;
  uint8_t* d_mask_out = mask_out;
  if (mem == PCLSEG_MEM_HOST) {
    int rc;
    if ((rc = ensure(h, &h->d_stage_in, &h->stage_in_bytes, n * HW * cin * sizeof(float)))) return rc;
    if ((rc = ensure(h, (void**)&h->d_stage_preds, &h->stage_preds_bytes, n * HW * sizeof(int32_t)))) return rc;
    if (!raw || mask_out)
      if ((rc = ensure(h, (void**)&h->d_stage_mask, &h->stage_mask_bytes, n * HW))) return rc;
    if (probs && (rc = ensure(h, (void**)&h->d_stage_probs, &h->stage_probs_bytes, n * HW * NC * sizeof(float)))) return rc;
    if (logits && (rc = ensure(h, (void**)&h->d_stage_logits, &h->stage_logits_bytes, n * HW * NC * sizeof(float)))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage_in, input, n * HW * cin * sizeof(float), hipMemcpyHostToDevice, h->stream));
    d_in = (const float*)h->d_stage_in;
    if (!raw) {
      HIP_TRY(h, hipMemcpyAsync(h->d_stage_mask, mask_in, n * HW, hipMemcpyHostToDevice, h->stream));
      d_mask_in = h->d_stage_mask;
    }
    d_preds = h->d_stage_preds;
    d_probs = probs ? h->d_stage_probs : nullptr;
    d_logits = logits ? h->d_stage_logits : nullptr;
    d_mask_out = mask_out ? h->d_stage_mask : nullptr;
  }

  float* d_lidar8 = h->d_arena + g.tensors[g.t_input].offset;
  NormArgs na;
  for (int i = 0; i < 5; ++i) { na.mean[i] = g.desc.mean[i]; na.std[i] = g.desc.std[i]; }
  for (int s0 = 0; s0 < n; s0 += g.micro_batch) {
    const int cnt = std::min(g.micro_batch, n - s0);
    const size_t P = (size_t)cnt * HW;
    const uint8_t* mask_mb;
    if (raw) {
      uint8_t* mdst = d_mask_out ? d_mask_out + (size_t)s0 * HW : h->d_mask;
      hipLaunchKernelGGL(normalize_kernel<8>, dim3(stream_blocks(P)), dim3(256), 0, h->stream,
                         d_in + (size_t)s0 * HW * 5, d_lidar8, mdst, P, na);
      HIP_TRY(h, hipGetLastError());
      mask_mb = mdst;
    } else {
      hipLaunchKernelGGL(pad6to8_kernel, dim3(stream_blocks(P)), dim3(256), 0, h->stream,
                         d_in + (size_t)s0 * HW * 6, d_lidar8, P);
      HIP_TRY(h, hipGetLastError());
      mask_mb = d_mask_in + (size_t)s0 * HW;
    }
    int rc = run_ops(h, cnt, mask_mb, d_preds + (size_t)s0 * HW,
                     d_probs ? d_probs + (size_t)s0 * HW * NC : nullptr,
                     d_logits ? d_logits + (size_t)s0 * HW * NC : nullptr);
    if (rc) return rc;
    h->last_count = cnt;
  }
  if (mem == PCLSEG_MEM_HOST) {
    HIP_TRY(h, hipMemcpyAsync(preds, d_preds, n * HW * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (probs) HIP_TRY(h, hipMemcpyAsync(probs, d_probs, n * HW * NC * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (logits) HIP_TRY(h, hipMemcpyAsync(logits, d_logits, n * HW * NC * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (mask_out) HIP_TRY(h, hipMemcpyAsync(mask_out, d_mask_out, n * HW, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return PCLSEG_OK;
}

// geometry of a stand-alone op (single-operator entry points)
void op_geometry(Op* op) {
  op->nc16 = (op->cin_t + 15) / 16;
  const int nct = (op->cout + 15) / 16;
  op->nt = (op->kind == OP_HEAD) ? nct : choose_nt(nct);
  op->nctp = ((nct + op->nt - 1) / op->nt) * op->nt;
}

struct DevBuf {
  float* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

// pack + upload the parameters of one stand-alone op; w at dev.p, bias at dev.p + nw
int upload_op(const Op& op, const FoldIn& f, DevBuf* dev, size_t* nw_out) {
  const size_t nw = (size_t)op.kh * op.kw * op.nc16 * op.nctp * 256;
  const size_t nb = (size_t)op.nctp * 16;
  std::vector<float> host(nw + nb);
  pack_op(op, f, host.data(), host.data() + nw);
  HIP_TRY(nullptr, hipMalloc((void**)&dev->p, (nw + nb) * sizeof(float)));
  HIP_TRY(nullptr, hipMemcpy(dev->p, host.data(), (nw + nb) * sizeof(float), hipMemcpyHostToDevice));
  *nw_out = nw;
  return PCLSEG_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int pclseg_version(void) { return PCLSEG_VERSION; }

const char* pclseg_last_error(const pclseg_handle* h) {
  if (h) return h->err.c_str();
  return g_last_error.c_str();
}

int pclseg_plan(const pclseg_desc* desc, pclseg_plan_info* out) {
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
  out->packed_weight_bytes = g.packed_floats * (int64_t)sizeof(float);
  return PCLSEG_OK;
}

int pclseg_create(const pclseg_desc* desc, pclseg_handle** out) {
  if (!out) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  pclseg_handle* h = new pclseg_handle();
  int rc = build_graph(desc, &h->g);
  if (rc) {
    std::string msg = h->g.error;
    delete h;
    return fail(nullptr, rc, msg);
  }
  h->device = desc->device;
  h->host_w.resize(h->g.weights.size());
  h->is_set.assign(h->g.weights.size(), 0);
  auto bail = [&](int code, const std::string& msg) {
    pclseg_destroy(h);
    return fail(nullptr, code, msg);
  };
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return bail(PCLSEG_ERR_HIP, fmt("no HIP device available (%s); this engine has no CPU fallback",
                                    e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
  if (desc->device < 0 || desc->device >= ndev)
    return bail(PCLSEG_ERR_BAD_ARG, fmt("device %d out of range (have %d)", desc->device, ndev));
  if ((e = hipSetDevice(desc->device)) != hipSuccess)
    return bail(PCLSEG_ERR_HIP, fmt("hipSetDevice(%d): %s", desc->device, hipGetErrorString(e)));
  const size_t arena_bytes = (size_t)h->g.arena_floats * sizeof(float);
  if ((e = hipMalloc((void**)&h->d_arena, arena_bytes)) != hipSuccess)
    return bail(e == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,
                fmt("hipMalloc(%zu B activation arena): %s", arena_bytes, hipGetErrorString(e)));
  (void)hipMemset(h->d_arena, 0, arena_bytes);
  const size_t mask_bytes = (size_t)h->g.micro_batch * desc->height * desc->width;
  if ((e = hipMalloc((void**)&h->d_mask, mask_bytes)) != hipSuccess)
    return bail(PCLSEG_ERR_HIP, fmt("hipMalloc(mask): %s", hipGetErrorString(e)));
  if ((e = hipMalloc((void**)&h->d_params, (size_t)h->g.packed_floats * sizeof(float))) != hipSuccess)
    return bail(e == hipErrorOutOfMemory ? PCLSEG_ERR_OOM : PCLSEG_ERR_HIP,
                fmt("hipMalloc(parameters): %s", hipGetErrorString(e)));
  *out = h;
  return PCLSEG_OK;
}

int pclseg_destroy(pclseg_handle* h) {
  if (!h) return PCLSEG_OK;
  if (h->d_arena || h->d_params) (void)hipSetDevice(h->device);
  void* bufs[] = {h->d_arena, h->d_params, h->d_mask, h->d_stage_in, h->d_stage_mask,
                  h->d_stage_preds, h->d_stage_probs, h->d_stage_logits};
  for (void* p : bufs)
    if (p) (void)hipFree(p);
  delete h;
  return PCLSEG_OK;
}

int pclseg_num_weights(const pclseg_handle* h) { return h ? (int)h->g.weights.size() : PCLSEG_ERR_BAD_ARG; }

int pclseg_weight_info(const pclseg_handle* h, int index, char* name, size_t name_cap,
                       int64_t shape[4], int* ndim) {
  if (!h || index < 0 || index >= (int)h->g.weights.size())
    return fail(const_cast<pclseg_handle*>(h), PCLSEG_ERR_BAD_ARG, "bad weight index");
  const WeightInfo& w = h->g.weights[index];
  if (name && name_cap) snprintf(name, name_cap, "%s", w.name.c_str());
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = w.shape[i];
  if (ndim) *ndim = w.ndim;
  return PCLSEG_OK;
}

int pclseg_set_weight(pclseg_handle* h, const char* keras_path, const float* data,
                      const int64_t* shape, int ndim) {
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
}

int pclseg_finalize(pclseg_handle* h) {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  if (h->finalized) return fail(h, PCLSEG_ERR_STATE, "finalize called twice");
  for (size_t i = 0; i < h->is_set.size(); ++i)
    if (!h->is_set[i])
      return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("tensor '%s' was never set", h->g.weights[i].name.c_str()));
  auto W = [&](const std::string& name) -> const float* {
    auto it = h->g.weight_index.find(name);
    return it == h->g.weight_index.end() ? nullptr : h->host_w[it->second].data();
  };
  std::vector<float> blob((size_t)h->g.packed_floats, 0.0f);
  for (const Op& op : h->g.ops) {
    if (op.kind == OP_POOL) continue;
    FoldIn f;
    f.kernel = W(op.name + "/kernel");
    f.bias = op.has_bias ? W(op.name + "/bias") : nullptr;
    if (!op.bn.empty()) {
      f.gamma = W(op.bn + "/gamma");
      f.beta = W(op.bn + "/beta");
      f.mean = W(op.bn + "/moving_mean");
      f.var = W(op.bn + "/moving_variance");
    }
    if (!f.kernel || (op.has_bias && !f.bias) || (!op.bn.empty() && !(f.gamma && f.beta && f.mean && f.var)))
      return fail(h, PCLSEG_ERR_MISSING_WEIGHT, fmt("internal: parameters of '%s' not found", op.name.c_str()));
    pack_op(op, f, blob.data() + op.w_off, blob.data() + op.b_off);
  }
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpy(h->d_params, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
  h->finalized = true;
  // the Keras-layout copies are no longer needed
  for (auto& v : h->host_w) std::vector<float>().swap(v);
  return PCLSEG_OK;
}

int pclseg_set_stream(pclseg_handle* h, void* hip_stream) {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  h->stream = (hipStream_t)hip_stream;
  return PCLSEG_OK;
}

int pclseg_sync(pclseg_handle* h) {
  if (!h) return fail(nullptr, PCLSEG_ERR_BAD_ARG, "handle is NULL");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return PCLSEG_OK;
}

int pclseg_forward(pclseg_handle* h, const float* lidar, const uint8_t* mask, int n, int32_t* preds,
                   float* probs, float* logits, int mem) {
  return forward_impl(h, lidar, false, mask, n, preds, probs, logits, nullptr, mem);
}

int pclseg_forward_raw(pclseg_handle* h, const float* scans, int n, int32_t* preds, float* probs,
                       float* logits, uint8_t* mask_out, int mem) {
  return forward_impl(h, scans, true, nullptr, n, preds, probs, logits, mask_out, mem);
}

int pclseg_num_tensors(const pclseg_handle* h) { return h ? (int)h->g.tensors.size() : PCLSEG_ERR_BAD_ARG; }

int pclseg_tensor_info(const pclseg_handle* h, int index, char* name, size_t name_cap, int64_t shape[4]) {
  if (!h || index < 0 || index >= (int)h->g.tensors.size())
    return fail(const_cast<pclseg_handle*>(h), PCLSEG_ERR_BAD_ARG, "bad tensor index");
  const TensorInfo& t = h->g.tensors[index];
  if (name && name_cap) snprintf(name, name_cap, "%s", t.name.c_str());
  if (shape) { shape[0] = h->last_count; shape[1] = t.H; shape[2] = t.W; shape[3] = t.C; }
  return PCLSEG_OK;
}

int pclseg_read_tensor(pclseg_handle* h, int index, float* host_out, size_t capacity_floats) {
  if (!h || !host_out || index < 0 || index >= (int)h->g.tensors.size())
    return fail(h, PCLSEG_ERR_BAD_ARG, "bad argument to read_tensor");
  const TensorInfo& t = h->g.tensors[index];
  const size_t nfl = (size_t)h->last_count * t.scan_floats();
  if (capacity_floats < nfl) return fail(h, PCLSEG_ERR_BAD_ARG, "host buffer too small");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemcpy(host_out, h->d_arena + t.offset, nfl * sizeof(float), hipMemcpyDeviceToHost));
  return PCLSEG_OK;
}

// ---- single-operator entry points
int pclseg_op_normalize(const float* scans, int n, int h, int w, const double mean[5],
                        const double std[5], float* lidar6, uint8_t* mask) {
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
}

int pclseg_op_conv2d(const float* x, int n, int h, int w, int cin, const float* kernel, int kh, int kw,
                     int cout, int stride_w, const float* bias, const float* bn_gamma,
                     const float* bn_beta, const float* bn_mean, const float* bn_var, int act,
                     const float* residual, float* y) {
  if (!x || !kernel || !y || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_conv2d");
  if (cin % 4 || cout % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin and Cout must be multiples of 4");
  if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3)) || (stride_w != 1 && stride_w != 2) || act < 0 || act > 3)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "unsupported kernel size / stride / activation");
  Op op;
  op.kind = OP_CONV;
  op.cin_t = op.cin_k = cin;
  op.cout = cout;
  op.kh = kh; op.kw = kw; op.sw = stride_w;
  op.act = act;
  op_geometry(&op);
  FoldIn f;
  f.kernel = kernel; f.bias = bias;
  f.gamma = bn_gamma; f.beta = bn_beta; f.mean = bn_mean; f.var = bn_var;
  DevBuf dev;
  size_t nw;
  int rc = upload_op(op, f, &dev, &nw);
  if (rc) return rc;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.wpk = dev.p; a.bias = dev.p + nw; a.out = y; a.out_C = cout;
  if (residual) { a.res1 = residual; a.res1_C = cout; }
  HIP_TRY(nullptr, launch_conv(op, n, h, w, a, nullptr));
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
}

int pclseg_op_conv2d_transpose(const float* x, int n, int h, int w, int cin, const float* kernel,
                               int cout, const float* bias, const float* bn_gamma,
                               const float* bn_beta, const float* bn_mean, const float* bn_var,
                               int act, float* y) {
  if (!x || !kernel || !y || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_conv2d_transpose");
  if (cin % 4 || cout % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin and Cout must be multiples of 4");
  for (int parity = 0; parity < 2; ++parity) {
    Op op;
    op.kind = OP_DECONV;
    op.cin_t = op.cin_k = cin;
    op.cout = cout;
    op.kh = 1; op.kw = 2; op.sw = parity;
    op.act = act;
    op_geometry(&op);
    FoldIn f;
    f.kernel = kernel; f.bias = bias;
    f.gamma = bn_gamma; f.beta = bn_beta; f.mean = bn_mean; f.var = bn_var;
    DevBuf dev;
    size_t nw;
    int rc = upload_op(op, f, &dev, &nw);
    if (rc) return rc;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = x; a.wpk = dev.p; a.bias = dev.p + nw; a.out = y; a.out_C = cout;
    HIP_TRY(nullptr, launch_conv(op, n, h, w, a, nullptr));
    HIP_TRY(nullptr, hipDeviceSynchronize());
  }
  return PCLSEG_OK;
}

int pclseg_op_max_pool(const float* x, int n, int h, int w, int c, int k, int stride_w, float* y) {
  if (!x || !y || n <= 0 || h <= 0 || w <= 0 || k <= 0 || (stride_w != 1 && stride_w != 2))
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_max_pool");
  if (c % 4) return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "C must be a multiple of 4");
  HIP_TRY(nullptr, launch_pool(x, y, n, h, w, c, k, stride_w, nullptr));
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
}

int pclseg_op_head(const float* x, const uint8_t* mask, int n, int h, int w, int cin,
                   const float* kernel, const float* bias, int num_class, int none_index,
                   int32_t* preds, float* probs, float* logits) {
  if (!x || !mask || !kernel || !bias || !preds || n <= 0 || h <= 0 || w <= 0)
    return fail(nullptr, PCLSEG_ERR_BAD_ARG, "bad argument to op_head");
  if (cin % 4 || num_class < 2 || num_class > 64)
    return fail(nullptr, PCLSEG_ERR_BAD_SHAPE, "Cin must be a multiple of 4 and num_class in [2,64]");
  Op op;
  op.kind = OP_HEAD;
  op.cin_t = op.cin_k = cin;
  op.cout = num_class;
  op.kh = op.kw = 3; op.sw = 1;
  op_geometry(&op);
  FoldIn f;
  f.kernel = kernel; f.bias = bias;
  DevBuf dev;
  size_t nw;
  int rc = upload_op(op, f, &dev, &nw);
  if (rc) return rc;
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.wpk = dev.p; a.bias = dev.p + nw;
  a.mask = mask; a.preds = preds; a.probs = probs; a.logits = logits; a.none_index = none_index;
  HIP_TRY(nullptr, launch_conv(op, n, h, w, a, nullptr));
  HIP_TRY(nullptr, hipDeviceSynchronize());
  return PCLSEG_OK;
}

}  // extern "C"
