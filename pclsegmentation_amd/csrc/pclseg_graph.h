// pclseg_graph.h — host-side graph description of the three networks (no HIP in here).
//
// Builds, from a pclseg_desc, the ordered operator list, the activation tensors with their
// lifetimes, the Keras weight inventory and the workspace plan.  The same structure drives
// pclseg_plan (CPU-only) and the device engine.
//
// reference graphs: nets/SqueezeSegV2.py:285-325 (CAM :66-70, FIRE :123-127, FIREUP :191-199),
// nets/Darknet.py:279-314 (BasicBlock :54-66, EncoderLayer :96-103, DecoderLayer :130-138,
// stride logic :158-181/:215-231, skip bookkeeping :263-277).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pclseg.h"

namespace pclseg {

// Experiment switches.  The shipped library carries only the decisions (the defaults below are the
// measured best); `make tuning` (-DPCLSEG_TUNING) builds libpclseg_tuning.so, in which
// the same names are read from the environment for A/B runs (scripts/ab_prof.sh).
#ifdef PCLSEG_TUNING
inline int tune_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int tune_env(const char*, int dflt) { return dflt; }
#endif

// The debug switches the SHIPPED library honours (PCLSEG_LANES and the PCLSEG_FUSE_* set).  They are
// read in ONE place, which also remembers what each resolved to: a stray variable in a rank's environment changes
// that rank's plan, and the error that reports it (pclseg_import_packed's plan mismatch) prints the resolved set.
struct DebugSwitches {
  static std::vector<std::pair<std::string, int>>& seen() { static std::vector<std::pair<std::string, int>> v; return v; }
  static std::string text();
};
inline std::mutex& debug_env_mutex();
inline std::string DebugSwitches::text() {
    std::lock_guard<std::mutex> lock(debug_env_mutex());
    std::string t;
    for (const auto& kv : seen()) t += (t.empty() ? "" : " ") + kv.first + "=" + std::to_string(kv.second);
    return t.empty() ? "none read yet" : t;
}
inline std::mutex& debug_env_mutex() { static std::mutex m; return m; }
inline int debug_env(const char* name, int dflt) {
  const char* v = getenv(name);
  const int r = v ? atoi(v) : dflt;
  std::lock_guard<std::mutex> lock(debug_env_mutex());   // (handles may be created from several threads at once)
  for (auto& kv : DebugSwitches::seen()) if (kv.first == name) { kv.second = r; return r; }
  DebugSwitches::seen().emplace_back(name, r);
  return r;
}

enum OpKind { OP_CONV = 0, OP_POOL = 2, OP_HEAD = 3, OP_CAM = 4 };

struct WeightInfo {
  std::string name;
  int ndim;
  int64_t shape[4];
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

enum TensorFmt { FMT_F32 = 0, FMT_S16 = 1 };  // S16: split-f16 pair format, see pclseg_kernels.h

struct TensorInfo {
  std::string name;
  int H, W, C;
  int fmt = FMT_F32;
  int def_op = -1, last_op = -1;
  int64_t offset = -1;  // floats, into the activation arena
  int64_t scan_floats() const { return (int64_t)H * W * C; }
};

// One Keras layer's share of a launch: a window of taps inside the staged input patch.
struct SubOp {
  std::string name;  // Keras layer path, e.g. "fire2/expand3x3"
  std::string bn;    // Keras BatchNormalization path, "" = none
  bool has_bias = true;
  bool deconv = false;    // Keras kernel layout (1,4,Cout,Cin) instead of (kh,kw,Cin,Cout)
  int cout = 0, co_off = 0;
  int th0 = 0, tw0 = 0, nkh = 1, nkw = 1;  // tap window inside the patch
  int ktap[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};  // window tap -> tap index of the Keras kernel
  int ow_off = 0;
  int act = 0;
  // packed parameters (offsets into the device blobs; w16 in halfs)
  int nctp = 0;
  int64_t w32_off = 0, w16_off = 0, b_off = 0;
};

struct Op {
  int kind = OP_CONV;
  int in = -1, out = -1, res1 = -1, res2 = -1;
  bool res1_mul = false;
  int cin_t = 0;  // channels of the input tensor (8 for the zero-padded network input)
  int cin_k = 0;  // Cin of the Keras kernel (6 for the network input)
  int pkh = 1, pkw = 1;  // tap extent of the staged patch
  int sw = 1;            // W stride of the convolution
  int pl_fixed = -1;     // >= 0: left padding of the patch (transposed conv), else TF SAME
  int ow_mul = 1;        // output column = j*ow_mul + sub.ow_off (2 for transposed conv)
  int nsub = 1;
  SubOp sub[2];
  int ntw = 1, wn = 1;   // kernel template: 16-cout tiles per wave, wave columns per block
  int mtw = 4;           // kernel template: 16-pixel segments per wave (2, 4 or 8)
  int nw = 4;            // kernel template: waves per block (4 or 8)
  int ck16 = 64;         // channels per LDS pass, split-f16 mode
  int ck32 = 32;         // channels per LDS pass, exact-f32 mode
  int pool_kh = 1, pool_kw = 1;
  bool pair = false;     // FIRE expand pair run as merged blocks (conv_kernel PAIR), split-f16 mode
  // 1x1 conv whose input is max-pooled 3x3 / strides (1,2) on the fly (pool_squeeze_kernel): `in` is
  // the tensor BEFORE the pool, the pooled tensor is never written
  bool pool_fused = false;
  // FIREUP expand pair that up-convolves its own input patch (conv_kernel UP): `in` is the module's
  // squeeze output at half the width, `up[parity]` are the transposed conv's two 2-tap sub-convs
  bool up_fused = false;
  SubOp up[2];
  // fused squeeze of the NEXT FIRE module (conv_kernel FSQ): this op writes fireN+1/squeeze instead
  // of its own output; `fsq` names the Keras tensors, fsq.w16_off / b_off locate the packed fragments
  bool fsq_fused = false;
  SubOp fsq;
  // optional fused skip branch (SqueezeSegV2 conv1_skip/bn1_skip): BN(conv1x1(sk_in)) is added
  // in the epilogue; `sk` names its Keras tensors, sk.b_off locates [8][C] weights + [C] bias
  int sk_in = -1;
  SubOp sk;
  // fire13 -> conv14 -> head in one kernel (fire_head_kernel): this expand pair (up_fused, fused skip branch)
  // also runs the segmentation head that is its only reader; `hd` is the head's 3x3 conv, the pair's own
  // output tensor is never materialised and the op writes predictions / probabilities / logits
  bool head_fused = false;
  SubOp hd;
  SubOp skm;   // head_fused: the skip branch as split-f16 fragments (K = 8 input channels in lane group 0), 4 cout tiles
  std::string name() const { return sub[0].name; }
};

struct Graph {
  pclseg_desc desc;
  int micro_batch = 1;
  std::vector<Op> ops;
  std::vector<TensorInfo> tensors;
  std::vector<WeightInfo> weights;
  std::map<std::string, int> weight_index;
  int t_input = -1;  // lidar8
  int64_t arena_floats = 0;  // per micro-batch
  int64_t alg_macs = 0, alg_bytes = 0, num_params = 0;
  int64_t packed32_floats = 0, packed16_halfs = 0, packed_bias_floats = 0;
  std::string error;
};

inline void same_pad(int size, int k, int s, int* out, int* before) {
  *out = (size + s - 1) / s;
  int total = std::max((*out - 1) * s + k - size, 0);
  *before = total / 2;
}

// ---- block / LDS geometry shared by the planner and the launcher
struct TileGeom { int TH, SEGW, PH, PW; };

inline bool op_is_flat(const Op& op) {
  return op.kind == OP_CONV && op.pkh == 1 && op.pkw == 1 && op.sw == 1 && op.ow_mul == 1;
}

// A block covers (nw/wn)*mtw segments of 16 pixels: 1x1 convs walk the flattened N*H*W pixel row,
// everything else takes an 8-row tile, 16, 32 or 64 columns wide.
inline TileGeom tile_geom(const Op& op) {
  TileGeom t;
  const int S = (op.nw / op.wn) * op.mtw;  // segments per block: 4, 8, 16 or 32
  if (op_is_flat(op)) { t.TH = 1; t.SEGW = S; }
  else if (S >= 8) { t.TH = 8; t.SEGW = S / 8; }
  else { t.TH = S; t.SEGW = 1; }
  t.PH = t.TH + op.pkh - 1;
  t.PW = (t.SEGW * 16 - 1) * op.sw + op.pkw;
  return t;
}

// halfs per staged pixel in split-f16 mode: the chunk's float4 quads rounded up to a power of
// two (the staging loop indexes with shifts), plus padding
inline int f16_csh(int cin_t, int ck) {
  const int cin8 = (cin_t + 7) / 8;
  const int qmax = std::min(cin8, ck / 8) * 2;
  int qs = 2;
  while (qs < qmax) qs *= 2;
  return qs * 4 + 8;
}
inline int64_t lds_bytes_f16(const Op& op, int ck) {
  const TileGeom t = tile_geom(op);
  return (int64_t)2 * t.PH * t.PW * f16_csh(op.cin_t, ck) * 2;
}
inline int64_t lds_bytes_f32(const Op& op, int ck) {
  const TileGeom t = tile_geom(op);
  const int cinp = ((op.cin_t + 15) / 16) * 16;
  return (int64_t)t.PH * t.PW * (std::min(cinp, ck) + 4) * 4;
}

// Block shape of a merged FIRE expand pair (split-f16 mode).  e = couts of each half.
//   e >= 128 (the W/8 and W/16 layers, matrix-core / latency bound): 8-wave blocks with large
//     register tiles — 128 pixels x all or half of the couts per block, patch staged once.
//   e <= 64  (the high-resolution layers, bandwidth bound): 4-wave blocks, many per CU.
inline void pair_geometry(Op* op) {
  static const int big = tune_env("PCLSEG_BIGTILE", 1);
  const int nct = (op->sub[1].cout + 15) / 16;
  if (!big) return;
  // measured per layer (one-lane us, 4-wave vs 8-wave blocks): fire4/5 27.4/31.0 vs 29.7/31.5,
  // fire6/7 28.1/28.2 vs 30.3/29.0, fire8/9 42.5/43.1 vs 41.7/41.9, fire10 53.5 vs 46.3: big tiles pay
  // from 256 couts per half, and for the 128-cout FIREUP pair (fire10)
  if (op->sk_in >= 0 && nct == 2) { op->nw = 8; op->wn = 2; op->ntw = 1; op->mtw = 4; return; }   // fire13: 256 px x (32 + 32),
                                                   // the fused skip-branch epilogue needs the 8-wave register budget
  if (nct == 16) { op->nw = 8; op->wn = 8; op->ntw = 2; op->mtw = 8; }                        // 128 px x 256 couts
  else if (nct == 8 && (op->res1 >= 0 || big == 2)) { op->nw = 8; op->wn = 4; op->ntw = 2; op->mtw = 4; }  // 128 px x 128 couts
  else if (nct == 12 && big == 2) { op->nw = 8; op->wn = 4; op->ntw = 3; op->mtw = 4; }       // 128 px x 192 couts
}

// Kernel configuration + packed-parameter geometry of one op (graph ops and stand-alone ops).
inline void op_geometry(Op* op) {
  if (op->kind == OP_POOL || op->kind == OP_CAM) return;
  const int nct = (op->sub[0].cout + 15) / 16;
  if (op->kind == OP_HEAD) { op->wn = 1; op->ntw = nct; }
  else if (nct % 4 == 0) { op->ntw = 2; op->wn = 2; }
  else if (nct % 3 == 0) { op->ntw = 3; op->wn = 1; }
  else if (nct % 2 == 0) { op->ntw = 2; op->wn = 1; }
  else if (nct == 1) { op->ntw = 1; op->wn = 1; }
  else { op->ntw = 2; op->wn = 2; }
  // WN = 2 -> 4 segments per wave (128-pixel blocks).  WN = 1: 256-pixel blocks for the FIRE expand
  // pairs and the 16-cout ops (fire12/13: 47 -> 45 us, 78 -> 70 us), 128-pixel blocks (smaller LDS
  // patch, more co-resident blocks) for the rest (head, 32-cout up-convolutions).
  op->mtw = (op->wn == 1 && op->nsub == 1 && op->ntw != 1) ? 2 : 4;
  op->nw = 4;
  if (op->pair) pair_geometry(op);
  // Darknet's wide layers (couts a multiple of 128, single sub-conv, 3x3 and 1x1): 8-wave blocks on the
  // same 128-pixel tile with 128 or 256 couts per block — the patch is staged once per 128/256 couts
  // instead of once per 64, and a weight fragment feeds 8 pixel segments instead of 4.  Measured
  // (Darknet-53 64x2048 / Darknet-21 32x1024 scans/s): 4-wave 334 / 2449, 128 couts 351 / 2578,
  // 256 couts where they divide 360 / 2657, the 1x1 layers too 364 / 2675.  (tuning aid: 0..3)
  static const int dn8 = tune_env("PCLSEG_DN8", 3);
  static const int dn_up = tune_env("PCLSEG_DN_UP8", 1);   // the decoder's transposed convs (two parity sub-convs) too
  if (dn8 && op->kind == OP_CONV && (op->nsub == 1 || (dn_up && op->nsub == 2 && op->sub[0].deconv)) && !op->pair && nct % 8 == 0 && (dn8 >= 3 || !op_is_flat(*op)) && op->sk_in < 0) {
    op->nw = 8; op->wn = 4; op->ntw = 2; op->mtw = 4;
    if (dn8 >= 2 && nct % 16 == 0) { op->wn = 8; op->mtw = 8; }   // 128 px x 256 couts
  }
  static const int geom_only = tune_env("PCLSEG_FSQ_GEOM_ONLY", 0);   // debug
  if (op->fsq_fused || (geom_only && op->pair && (nct == 16 || nct == 12 || nct == 8) && op->res1 < 0)) {   // 8 waves on a 64-pixel tile, all couts of both halves in the block
    op->nw = 8;
    if (nct == 16) { op->wn = 8; op->ntw = 2; op->mtw = 4; }        // 64 px x (256 + 256) couts
    else if (nct == 12) { op->wn = 4; op->ntw = 3; op->mtw = 2; }   // 64 px x (192 + 192)
    else if (nct == 8) { op->wn = 8; op->ntw = 1; op->mtw = 4; }    // 64 px x (128 + 128)
    else if (nct == 4) { op->wn = 4; op->ntw = 1; op->mtw = 4; }    // 128 px x (64 + 64)
    else { op->wn = 2; op->ntw = 1; op->mtw = 4; }                  // 256 px x (32 + 32)
  }
#ifdef PCLSEG_TUNING
  if (!op->fsq_fused) if (const char* ov = getenv("PCLSEG_GEOM")) {  // "subname=ntw,wn,mtw[,nw];subname=..."
    const std::string key = op->sub[0].name + "=";
    const char* hit = strstr(ov, key.c_str());
    int a = 0, b = 0, c = 0, d = 4;
    if (hit && (hit == ov || hit[-1] == ';')) {
      const int got = sscanf(hit + key.size(), "%d,%d,%d,%d", &a, &b, &c, &d);
      if (got >= 3) { op->ntw = a; op->wn = b; op->mtw = c; op->nw = got == 4 ? d : 4; }
    }
  }
#endif
  const int group = op->ntw * op->wn;
  for (int i = 0; i < op->nsub; ++i) {
    const int n = (op->sub[i].cout + 15) / 16;
    op->sub[i].nctp = ((n + group - 1) / group) * group;
  }
  const int64_t budget = 64 * 1024;  // default dynamic-LDS limit per block
  // f16 mode: 40 KiB keeps four blocks per CU resident (160 KiB LDS); worth a second channel chunk
  // from 64 input channels up (head 77 -> 68 us), not for the 48-channel squeezes (35 -> 36 us).
  // Merged FIRE pairs and 8-wave blocks take the whole patch in one chunk.
  static const int64_t b64 = tune_env("PCLSEG_LDS_BUDGET64", 40 * 1024);   // tuning aid
  const int64_t budget16 = (op->cin_t >= 64 && !op->pair && op->nw == 4) ? b64 : budget;
  op->ck16 = 64;
  while (op->ck16 > 16 && lds_bytes_f16(*op, op->ck16) > budget16) op->ck16 /= 2;
  op->ck32 = 32;
  while (op->ck32 > 16 && lds_bytes_f32(*op, op->ck32) > budget) op->ck32 /= 2;
  if (op->pair && op->ck16 < op->cin_t) op->pair = false;   // merged pairs need the patch in one chunk
}

// A sub-op's block in the bias blob: [nctp*16] folded biases, then [nctp*16] per-output-channel
// inverse weight scales 2^-k of the split-f16 fragments (pclseg_api.hip: cout_exponents).
inline int64_t sub_bias_floats(const SubOp& s) { return (int64_t)2 * s.nctp * 16; }
inline int64_t sub_w32_floats(const Op& op, const SubOp& s) {
  return (int64_t)s.nkh * s.nkw * ((op.cin_t + 15) / 16) * s.nctp * 256;
}
// fused next-squeeze fragments: [half][cout group][K-step of two cout tiles][16-q tile][hi|lo][lane][8]
inline int64_t fsq_w16_halfs(const Op& op) {
  const int ncg = op.sub[0].nctp / op.ntw, ns = (op.ntw + 1) / 2, nq = (op.fsq.cout + 15) / 16;
  return (int64_t)2 * ncg * ns * nq * 1024;
}
inline int f16_steps_full(const Op& op, const SubOp& s) { return (s.nkh * s.nkw * (op.ck16 / 8) + 3) / 4; }
inline int f16_chunks(const Op& op) { return ((op.cin_t + 7) / 8 + op.ck16 / 8 - 1) / (op.ck16 / 8); }
inline int64_t sub_w16_halfs(const Op& op, const SubOp& s) {
  return (int64_t)f16_chunks(op) * f16_steps_full(op, s) * s.nctp * 1024;
}

class GraphBuilder {
 public:
  explicit GraphBuilder(Graph* g) : g_(g) {}

  int tensor(const std::string& name, int H, int W, int C) {
    TensorInfo t;
    t.name = name;
    t.H = H;
    t.W = W;
    t.C = C;
    g_->tensors.push_back(t);
    return (int)g_->tensors.size() - 1;
  }

  void add_weight(const std::string& name, std::vector<int64_t> shape) {
    WeightInfo w;
    w.name = name;
    w.ndim = (int)shape.size();
    for (int i = 0; i < 4; ++i) w.shape[i] = i < w.ndim ? shape[i] : 1;
    g_->weight_index[name] = (int)g_->weights.size();
    g_->num_params += w.numel();
    g_->weights.push_back(w);
  }
  void add_bn(const std::string& p, int c) {
    add_weight(p + "/gamma", {c});
    add_weight(p + "/beta", {c});
    add_weight(p + "/moving_mean", {c});
    add_weight(p + "/moving_variance", {c});
  }

  void touch(int t, int op) {
    if (t < 0) return;
    TensorInfo& ti = g_->tensors[t];
    if (ti.def_op < 0) ti.def_op = op;
    ti.last_op = std::max(ti.last_op, op);
  }

  // declare the Keras tensors of a Conv2D layer and return its SubOp (full kh x kw window)
  SubOp conv_sub(const std::string& name, int kh, int kw, int cin_k, int cout, bool bias,
                 const std::string& bn, int act, int co_off) {
    SubOp s;
    s.name = name;
    s.bn = bn;
    s.has_bias = bias;
    s.cout = cout;
    s.co_off = co_off;
    s.nkh = kh;
    s.nkw = kw;
    s.act = act;
    add_weight(name + "/kernel", {kh, kw, cin_k, cout});
    if (bias) add_weight(name + "/bias", {cout});
    if (!bn.empty()) add_bn(bn, cout);
    return s;
  }

  // Conv2D SAME (+bias) (+BN) (+act), optional residuals, optional channel-slice output.
  // `out` < 0 creates the output tensor; returns the output tensor id.
  int conv(const std::string& name, int in, int kh, int kw, int cout, int sw, bool bias,
           const std::string& bn, int act, int out = -1, int co_off = 0, int res1 = -1,
           bool res1_mul = false, int res2 = -1, int cin_k = -1) {
    const TensorInfo ti = g_->tensors[in];
    Op op;
    op.kind = OP_CONV;
    op.in = in;
    op.cin_t = ti.C;
    op.cin_k = cin_k < 0 ? ti.C : cin_k;
    op.pkh = kh;
    op.pkw = kw;
    op.sw = sw;
    op.res1 = res1;
    op.res1_mul = res1_mul;
    op.res2 = res2;
    op.sub[0] = conv_sub(name, kh, kw, op.cin_k, cout, bias, bn, act, co_off);
    int wo, pl;
    same_pad(ti.W, kw, sw, &wo, &pl);
    if (out < 0) out = tensor(name, ti.H, wo, cout);
    op.out = out;
    g_->alg_macs += (int64_t)ti.H * wo * kh * kw * op.cin_k * cout;
    push(op);
    return out;
  }

  // FIRE / FIREUP expand stage: relu(bn(1x1)) || relu(bn(3x3)) over the same input, written to
  // the two channel slices of `out` (tf.concat), one launch, one staged patch.
  int expand_pair(const std::string& p, int in, int e1, int e3, int out, int skip) {
    const TensorInfo ti = g_->tensors[in];
    Op op;
    op.kind = OP_CONV;
    op.in = in;
    op.out = out;
    op.cin_t = op.cin_k = ti.C;
    op.pkh = op.pkw = 3;
    op.sw = 1;
    op.res1 = skip;
    op.nsub = 2;
    op.sub[0] = conv_sub(p + "/expand1x1", 1, 1, ti.C, e1, true, p + "/expand1x1_bn", 1, 0);
    op.sub[0].th0 = op.sub[0].tw0 = 1;  // centre tap of the 3x3 patch
    op.sub[1] = conv_sub(p + "/expand3x3", 3, 3, ti.C, e3, true, p + "/expand3x3_bn", 1, e1);
    g_->alg_macs += (int64_t)ti.H * ti.W * ti.C * (e1 + 9 * e3);
    push(op);
    return (int)g_->ops.size() - 1;
  }

  // Conv2DTranspose (1,4)/(1,2) SAME (+bias) (+BN) (+act): o = 2i + k - 1, so
  //   even o = 2j  : x[j-1]*K[3] + x[j]*K[1]      odd o = 2j+1 : x[j]*K[2] + x[j+1]*K[0]
  // = two 2-tap sub-convs over the patch {x[j-1], x[j], x[j+1]}, one launch.
  int deconv(const std::string& name, int in, int cout, const std::string& bn, int act) {
    const TensorInfo ti = g_->tensors[in];
    int out = tensor(name, ti.H, ti.W * 2, cout);
    add_weight(name + "/kernel", {1, 4, cout, ti.C});
    add_weight(name + "/bias", {cout});
    if (!bn.empty()) add_bn(bn, cout);
    g_->alg_macs += (int64_t)ti.H * ti.W * 4 * ti.C * cout;
    Op op;
    op.kind = OP_CONV;
    op.in = in;
    op.out = out;
    op.cin_t = op.cin_k = ti.C;
    op.pkh = 1;
    op.pkw = 3;
    op.sw = 1;
    op.pl_fixed = 1;
    op.ow_mul = 2;
    op.nsub = 2;
    for (int parity = 0; parity < 2; ++parity) {
      SubOp& s = op.sub[parity];
      s.name = name;
      s.bn = bn;
      s.has_bias = true;
      s.deconv = true;
      s.cout = cout;
      s.nkh = 1;
      s.nkw = 2;
      s.tw0 = parity;
      s.ktap[0] = parity == 0 ? 3 : 2;
      s.ktap[1] = parity == 0 ? 1 : 0;
      s.ow_off = parity;
      s.act = act;
    }
    push(op);
    return out;
  }

  // Whole Context Aggregation Module as one fused kernel (C = 64 or 128).
  int cam_fused(const std::string& p, int x) {
    const TensorInfo ti = g_->tensors[x];
    const int C = ti.C, R = C / 16;
    Op op;
    op.kind = OP_CAM;
    op.in = x;
    op.cin_t = op.cin_k = C;
    op.nsub = 2;
    op.sub[0] = conv_sub(p + "/squeeze", 1, 1, C, R, true, p + "/squeeze_bn", 1, 0);
    op.sub[1] = conv_sub(p + "/excitation", 1, 1, R, C, true, p + "/excitation_bn", 3, 0);
    op.out = tensor(p, ti.H, ti.W, C);
    g_->alg_macs += (int64_t)ti.H * ti.W * 2 * C * R;
    push(op);
    return op.out;
  }

  // MaxPool k x k, strides (1, sw), SAME (the graphs only use 3x3 s2: nets/SqueezeSegV2.py:295,301,305;
  // CAM's 7x7 s1 pool lives inside cam_kernel).
  int pool(const std::string& name, int in, int k, int sw) {
    const TensorInfo ti = g_->tensors[in];
    int wo, pl;
    same_pad(ti.W, k, sw, &wo, &pl);
    Op op;
    op.kind = OP_POOL;
    op.sub[0].name = name;
    op.in = in;
    op.pool_kh = op.pool_kw = k;
    op.sw = sw;
    op.cin_t = op.cin_k = ti.C;
    op.sub[0].cout = ti.C;
    op.out = tensor(name, ti.H, wo, ti.C);
    push(op);
    return op.out;
  }

  // Can pool_squeeze_kernel take this pool + squeeze?  (power-of-two channel count, all couts in one block)
  static bool pool_squeeze_ok(int C, int cout) { return C >= 32 && C <= 512 && (C & (C - 1)) == 0 && cout <= 64; }

  // The last op is the 3x3 s(1,2) pool that produced `pooled`, and this 1x1 conv (+BN+ReLU) is its only
  // reader: replace the pool by one op that pools while it loads (nets/SqueezeSegV2.py:295-296,301-302,305-306).
  int pooled_squeeze(const std::string& name, int pooled, int cout, const std::string& bn) {
    const Op pool_op = g_->ops.back();
    g_->ops.pop_back();
    const TensorInfo tp = g_->tensors[pooled];
    g_->tensors[pooled].def_op = g_->tensors[pooled].last_op = -1;   // never materialised
    Op op;
    op.kind = OP_CONV;
    op.in = pool_op.in;
    op.cin_t = op.cin_k = tp.C;
    op.pool_fused = true;
    op.sub[0] = conv_sub(name, 1, 1, tp.C, cout, true, bn, 1, 0);
    op.out = tensor(name, tp.H, tp.W, cout);
    g_->alg_macs += (int64_t)tp.H * tp.W * tp.C * cout;
    push(op);
    return op.out;
  }

  void head(const std::string& name, int in, int num_class) {
    const TensorInfo ti = g_->tensors[in];
    Op op;
    op.kind = OP_HEAD;
    op.in = in;
    op.cin_t = op.cin_k = ti.C;
    op.pkh = op.pkw = 3;
    op.sw = 1;
    op.sub[0] = conv_sub(name, 3, 3, ti.C, num_class, true, "", 0, 0);
    g_->alg_macs += (int64_t)ti.H * ti.W * 9 * ti.C * num_class;
    push(op);
  }

  // module-granular traffic accounting: floats read / written per scan
  void module_bytes(int64_t read_floats, int64_t write_floats, int64_t extra_bytes = 0) {
    g_->alg_bytes += 4 * (read_floats + write_floats) + extra_bytes;
  }
  int64_t fl(int t, int c_override = -1) const {
    const TensorInfo& ti = g_->tensors[t];
    return (int64_t)ti.H * ti.W * (c_override < 0 ? ti.C : c_override);
  }

 private:
  void push(const Op& op) {
    int idx = (int)g_->ops.size();
    g_->ops.push_back(op);
    touch(op.in, idx);
    touch(op.out, idx);
    touch(op.res1, idx);
    touch(op.res2, idx);
  }
  Graph* g_;
};

// ---- SqueezeSegV2 (reference: nets/SqueezeSegV2.py:285-325)
inline void build_squeezesegv2(Graph* g) {
  GraphBuilder b(g);
  const int H = g->desc.height, W = g->desc.width, NC = g->desc.num_class;
  const int x_in = b.tensor("input", H, W, 8);
  g->t_input = x_in;

  auto cam = [&](const std::string& p, int x) {   // C is 64 (cam1) or 128 (cam2, cam3): cam_kernel
    const int out = b.cam_fused(p, x);
    b.module_bytes(b.fl(x), b.fl(out));
    return out;
  };
  int last_expand = -1;
  // fireN's output (for the FIREUP modules: after its skip add) feeds ONLY fireN+1's squeeze for
  // N = 4, 6, 7, 8, 9 (:302-312) and N = 10, 11, 12 (:313-318): there the
  // squeeze is computed by fireN's expand blocks and the expand output is never written (split-f16
  // arithmetic only; every intermediate stays observable with KEEP_ACTIVATIONS, which disables it)
  static const int fuse_env = debug_env("PCLSEG_FUSE_SQ", 1);
  static const int fuse_pool = debug_env("PCLSEG_FUSE_POOL", 1);   // pool -> squeeze in one kernel
  static const int fuse_cam = debug_env("PCLSEG_FUSE_CAM", 1);   // cam2 -> fire3/squeeze in one kernel
  static const int fuse_keep = debug_env("PCLSEG_FUSE_KEEP", 0);   // debug
  const bool fuse = fuse_env && !(g->desc.flags & (PCLSEG_FLAG_EXACT_F32 | PCLSEG_FLAG_RANGE_FALLBACK)) &&
                    (fuse_keep || !(g->desc.flags & PCLSEG_FLAG_KEEP_ACTIVATIONS));
  auto fire = [&](const std::string& p, int x, int sq_c, int e1, int e3, bool up, int skip,
                  int64_t skip_floats = 0, bool fuse_prev = false) {
    int s;
    static const int fuse_mask = debug_env("PCLSEG_FUSE_MASK", 255);   // debug: bit per fusion
    const int fuse_bit = p == "fire5" ? 1 : p == "fire7" ? 2 : p == "fire8" ? 4 : p == "fire9" ? 8 : p == "fire10" ? 16 :
                         p == "fire11" ? 32 : p == "fire12" ? 64 : 128;
    if (fuse_prev && fuse && (fuse_mask & fuse_bit) && last_expand >= 0 && g->ops[last_expand].out == x) {
      const TensorInfo tx = g->tensors[x];
      Op& pe = g->ops[last_expand];
      pe.fsq = b.conv_sub(p + "/squeeze", 1, 1, tx.C, sq_c, true, p + "/squeeze_bn", 1, 0);
      pe.fsq_fused = true;
      s = b.tensor(p + "/squeeze", tx.H, tx.W, sq_c);
      pe.out = s;
      b.touch(s, last_expand);
      g->tensors[x].def_op = g->tensors[x].last_op = -1;   // never materialised
      g->alg_macs += (int64_t)tx.H * tx.W * tx.C * sq_c;
    } else if (fuse_prev && fuse && fuse_cam && !g->ops.empty() && g->ops.back().kind == OP_CAM && g->ops.back().out == x &&
               g->tensors[x].C == 128 && sq_c <= 32) {
      // cam2 -> fire3 (:297-298): the gated tile never leaves the CAM block, which writes fire3/squeeze
      const TensorInfo tx = g->tensors[x];
      Op& pc = g->ops.back();
      pc.fsq = b.conv_sub(p + "/squeeze", 1, 1, tx.C, sq_c, true, p + "/squeeze_bn", 1, 0);
      pc.fsq_fused = true;
      s = b.tensor(p + "/squeeze", tx.H, tx.W, sq_c);
      pc.out = s;
      b.touch(s, (int)g->ops.size() - 1);
      g->tensors[x].def_op = g->tensors[x].last_op = -1;   // never materialised
      g->alg_macs += (int64_t)tx.H * tx.W * tx.C * sq_c;
    } else if (fuse && fuse_pool && !g->ops.empty() && g->ops.back().kind == OP_POOL && g->ops.back().out == x &&
               g->ops.back().pool_kh == 3 && g->ops.back().sw == 2 && GraphBuilder::pool_squeeze_ok(g->tensors[x].C, sq_c)) {
      s = b.pooled_squeeze(p + "/squeeze", x, sq_c, p + "/squeeze_bn");   // pool1/3/5 feed only this squeeze
    } else {
      s = b.conv(p + "/squeeze", x, 1, 1, sq_c, 1, true, p + "/squeeze_bn", 1);
    }
    const int64_t in_floats = b.fl(x);
    if (up) s = b.deconv(p + "/upconv", s, sq_c, "", 1);  // ReLU, no BN (:194)
    const int out = b.tensor(p, g->tensors[s].H, g->tensors[s].W, e1 + e3);
    last_expand = b.expand_pair(p, s, e1, e3, out, skip);
    b.module_bytes(in_floats + (skip >= 0 ? b.fl(skip) : skip_floats), b.fl(out));
    return out;
  };
  auto pool = [&](const std::string& p, int x) {
    const int out = b.pool(p, x, 3, 2);
    b.module_bytes(b.fl(x), b.fl(out));
    return out;
  };

  int x = b.conv("conv1", x_in, 3, 3, 64, 2, true, "bn1", 1, -1, 0, -1, false, -1, 6);  // :289
  b.module_bytes(b.fl(x_in, 6), b.fl(x));
  const int cam1 = cam("cam1", x);                                                          // :291
  // conv1_skip + bn1_skip (:293) are not materialised: the 1x1 conv of the 6-channel input is
  // evaluated inside fire13's epilogue (:319).  ALG_BYTES still counts the reference's module.
  const SubOp skip_sub = b.conv_sub("conv1_skip", 1, 1, 6, 64, true, "bn1_skip", 0, 0);
  g->alg_macs += (int64_t)H * W * 6 * 64;
  b.module_bytes(b.fl(x_in, 6), (int64_t)H * W * 64);
  x = pool("pool1", cam1);                                                                  // :295
  x = fire("fire2", x, 16, 64, 64, false, -1);
  x = cam("cam2", x);
  x = fire("fire3", x, 16, 64, 64, false, -1, 0, true);
  const int cam3 = cam("cam3", x);                                                          // :299
  x = pool("pool3", cam3);                                                                  // :301
  x = fire("fire4", x, 32, 128, 128, false, -1);
  const int fire5 = fire("fire5", x, 32, 128, 128, false, -1, 0, true);                     // :303
  x = pool("pool5", fire5);                                                                 // :305
  x = fire("fire6", x, 48, 192, 192, false, -1);
  x = fire("fire7", x, 48, 192, 192, false, -1, 0, true);
  x = fire("fire8", x, 64, 256, 256, false, -1, 0, true);
  x = fire("fire9", x, 64, 256, 256, false, -1, 0, true);                                   // :309
  x = fire("fire10", x, 64, 128, 128, true, fire5, 0, true);                                // :312-313
  x = fire("fire11", x, 32, 64, 64, true, cam3, 0, true);                                   // :314-315
  x = fire("fire12", x, 16, 32, 32, true, cam1, 0, true);                                   // :316-317
  x = fire("fire13", x, 16, 32, 32, true, -1, (int64_t)H * W * 64, true);                   // :318-319
  g->ops[last_expand].sk_in = x_in;
  g->ops[last_expand].sk = skip_sub;
  b.touch(x_in, last_expand);
  b.head("conv14", x, NC);                                                                  // :323-325
  b.module_bytes(b.fl(x), (int64_t)H * W, (int64_t)H * W);  // + mask 1 B/px, int32 preds out
}

// ---- Darknet-21/53 (reference: nets/Darknet.py:279-314)
inline void darknet_strides(int output_stride, int enc[5], int dec[5]) {
  for (int i = 0; i < 5; ++i) enc[i] = dec[i] = 2;
  int cur = 32;
  if (output_stride <= cur) {
    for (int i = 0; i < 5; ++i) {  // reversed(encoder_strides)
      if (cur != output_stride) {
        if (enc[4 - i] == 2) { cur /= 2; enc[4 - i] = 1; }
        if (cur == output_stride) break;
      }
    }
  }
  cur = 32;
  for (int i = 0; i < 5; ++i) {
    if (cur != output_stride) {
      if (dec[i] == 2) { cur /= 2; dec[i] = 1; }
      if (cur == output_stride) break;
    }
  }
}

inline void build_darknet(Graph* g, int num_layers) {
  GraphBuilder b(g);
  const int H = g->desc.height, W = g->desc.width, NC = g->desc.num_class;
  static const int blocks21[5] = {1, 1, 2, 2, 1}, blocks53[5] = {1, 2, 8, 8, 4};
  const int* nb = num_layers == 21 ? blocks21 : blocks53;
  int enc_s[5], dec_s[5];
  darknet_strides(g->desc.output_stride, enc_s, dec_s);
  const int x_in = b.tensor("input", H, W, 8);
  g->t_input = x_in;
  const int LR = 2;  // LeakyReLU(0.1)

  // BasicBlock: x + lrelu(bn2(conv3x3(lrelu(bn1(conv1x1(x)))))) (+ decoder skip)
  auto block = [&](const std::string& p, int x, int mid, int c, int skip) {
    const int m = b.conv(p + "/conv1", x, 1, 1, mid, 1, false, p + "/bn1", LR);
    const int y = b.conv(p + "/conv2", m, 3, 3, c, 1, false, p + "/bn2", LR, -1, 0, x, false, skip);
    b.module_bytes(b.fl(x) + (skip >= 0 ? b.fl(skip) : 0), b.fl(y));
    return y;
  };

  int x = b.conv("conv1", x_in, 3, 3, 32, 1, false, "bn1", LR, -1, 0, -1, false, -1, 6);  // :288-290
  b.module_bytes(b.fl(x_in, 6), b.fl(x));
  static const int enc_planes[5][2] = {{32, 64}, {64, 128}, {128, 256}, {256, 512}, {512, 1024}};
  std::vector<int> skips;  // inputs of the W-shrinking encoder layers (:263-269)
  for (int i = 0; i < 5; ++i) {
    const std::string p = "enc" + std::to_string(i + 1);
    if (enc_s[i] == 2) skips.push_back(x);
    int y = b.conv(p + "/conv1", x, 3, 3, enc_planes[i][1], enc_s[i], false, p + "/bn1", LR);
    b.module_bytes(b.fl(x), b.fl(y));
    for (int j = 0; j < nb[i]; ++j)
      y = block(p + "/residual_" + std::to_string(j), y, enc_planes[i][0], enc_planes[i][1], -1);
    x = y;
  }
  static const int dec_planes[5][2] = {{1024, 512}, {512, 256}, {256, 128}, {128, 64}, {64, 32}};
  for (int k = 0; k < 5; ++k) {
    const std::string p = "dec" + std::to_string(5 - k);
    int y;
    int skip = -1;
    if (dec_s[k] == 2) {
      y = b.deconv(p + "/upconv1", x, dec_planes[k][1], p + "/bn1", LR);
      skip = skips.back();  // matching encoder input (:271-277)
      skips.pop_back();
    } else {
      y = b.conv(p + "/conv1", x, 3, 3, dec_planes[k][1], 1, true, p + "/bn1", LR);
    }
    b.module_bytes(b.fl(x), b.fl(y));
    x = block(p + "/block", y, dec_planes[k][0], dec_planes[k][1], skip);
  }
  b.head("head", x, NC);                                                                    // :312-314
  b.module_bytes(b.fl(x), (int64_t)H * W, (int64_t)H * W);
}

// ---- liveness-based first-fit placement of activation tensors in one arena
inline void plan_workspace(Graph* g) {
  struct Live { int64_t off, size; int last; };
  std::vector<Live> live;
  const bool keep = (g->desc.flags & PCLSEG_FLAG_KEEP_ACTIVATIONS) != 0;
  std::vector<int> order(g->tensors.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b2) {
    return g->tensors[a].def_op < g->tensors[b2].def_op;
  });
  int64_t top = 0;
  for (int ti : order) {
    TensorInfo& t = g->tensors[ti];
    const int64_t size = ((t.scan_floats() * g->micro_batch + 63) / 64) * 64;  // 256-B granules
    if (keep) {
      t.offset = top;
      top += size;
      continue;
    }
    // the network input is written before op 0 (def_op == 0 as a reader): treat def as -1
    const int def = (ti == g->t_input) ? -1 : t.def_op;
    std::vector<Live> still;
    for (const Live& l : live)
      if (l.last >= def) still.push_back(l);
    live.swap(still);
    std::sort(live.begin(), live.end(), [](const Live& a, const Live& b2) { return a.off < b2.off; });
    int64_t off = 0;
    for (const Live& l : live) {
      if (off + size <= l.off) break;
      off = std::max(off, l.off + l.size);
    }
    t.offset = off;
    live.push_back({off, size, t.last_op});
    top = std::max(top, off + size);
  }
  g->arena_floats = top;
}

// Tensor formats (split-f16 mode only).  A conv output whose ONLY reader is the main input of one
// other convolution — FIRE squeeze / up-convolution outputs, Darknet's 1x1 bottlenecks — is kept in
// the split-f16 pair format: the producer splits once, the reader's LDS staging is a plain copy.
// Everything a residual / skip / pool / CAM / the caller touches stays float32.  A FIRE expand pair
// whose input is in that format runs as merged blocks (conv_kernel PAIR).
inline void assign_formats(Graph* g) {
  if (g->desc.flags & PCLSEG_FLAG_EXACT_F32) return;
  static const int s16 = tune_env("PCLSEG_S16", 1);   // tuning aid: 0 = all float32
  static const int pair = tune_env("PCLSEG_PAIR", 1);
  if (!s16) return;
  const int nt = (int)g->tensors.size();
  std::vector<int> readers(nt, 0), other(nt, 0), producer(nt, -1), reader_op(nt, -1);
  for (size_t i = 0; i < g->ops.size(); ++i) {
    const Op& op = g->ops[i];
    if (op.in >= 0) {
      if (op.kind == OP_CONV || op.kind == OP_HEAD) { readers[op.in]++; reader_op[op.in] = (int)i; }
      else other[op.in]++;
    }
    if (op.res1 >= 0) other[op.res1]++;
    if (op.res2 >= 0) other[op.res2]++;
    if (op.sk_in >= 0) other[op.sk_in]++;
    if (op.kind == OP_CONV && op.out >= 0) producer[op.out] = (int)i;
  }
  for (int t = 0; t < nt; ++t) {
    if (t == g->t_input || producer[t] < 0 || readers[t] != 1 || other[t] != 0) continue;
    const Op& prod = g->ops[producer[t]];
    const Op& rd = g->ops[reader_op[t]];
    if (prod.fsq_fused) { g->tensors[t].fmt = FMT_S16; continue; }   // the fused squeeze writes split-f16 only
    bool narrow = prod.nsub == 1 ? (prod.pkh == 1 && prod.pkw == 1) : (prod.ow_mul == 2);  // 1x1 conv or up-conv
    if (s16 >= 2 && rd.kind == OP_HEAD) narrow = true;   // tuning aid: the head's input too
    if (!narrow || g->tensors[t].C % 8 != 0) continue;
    if (op_is_flat(rd) && rd.nsub == 1) continue;   // LDS-free 1x1 readers split in registers anyway
    g->tensors[t].fmt = FMT_S16;
  }
  for (const Op& op : g->ops)   // the squeeze a CAM block computes (cam_kernel SQ) feeds an expand pair's staging
    if (op.kind == OP_CAM && op.fsq_fused && readers[op.out] == 1 && other[op.out] == 0 && g->tensors[op.out].C % 8 == 0)
      g->tensors[op.out].fmt = FMT_S16;
  if (pair)
    for (Op& op : g->ops)
      if (op.kind == OP_CONV && op.nsub == 2 && op.ow_mul == 1 && op.pkh == 3 && op.sub[0].nkh == 1 &&
          op.sub[0].cout == op.sub[1].cout && g->tensors[op.in].fmt == FMT_S16)
        op.pair = true;
}

// Recompute every tensor's first / last op from the op list (after ops were removed or rewired).
inline void recompute_lifetimes(Graph* g) {
  for (TensorInfo& t : g->tensors) t.def_op = t.last_op = -1;
  auto touch = [&](int t, int i) {
    if (t < 0) return;
    TensorInfo& ti = g->tensors[t];
    if (ti.def_op < 0) ti.def_op = i;
    ti.last_op = std::max(ti.last_op, i);
  };
  for (size_t i = 0; i < g->ops.size(); ++i) {
    const Op& op = g->ops[i];
    touch(op.in, (int)i); touch(op.out, (int)i); touch(op.res1, (int)i); touch(op.res2, (int)i); touch(op.sk_in, (int)i);
  }
}

// SqueezeSegV2's FIREUP modules (nets/SqueezeSegV2.py:191-199): squeeze -> Conv2DTranspose -> expand pair.
// Where the pair runs as one merged block that holds every cout of its pixels (fire10/11/12 with their
// fused next squeeze, fire13 with its fused skip branch) and both tensors are split-f16, the transposed
// conv moves into the pair's staging (conv_kernel UP) and its launch and its output tensor disappear.
inline void fuse_upconvs(Graph* g) {
  static const int on = debug_env("PCLSEG_FUSE_UP", 1);
  static const int fuse_keep = debug_env("PCLSEG_FUSE_KEEP", 0);   // debug (tests/fused_worker.py)
  if (!on || (g->desc.flags & (PCLSEG_FLAG_EXACT_F32 | PCLSEG_FLAG_RANGE_FALLBACK))) return;
  if ((g->desc.flags & PCLSEG_FLAG_KEEP_ACTIVATIONS) && !fuse_keep) return;
  std::vector<int> readers(g->tensors.size(), 0);
  for (const Op& op : g->ops) {
    if (op.in >= 0) readers[op.in]++;
    if (op.res1 >= 0) readers[op.res1]++;
    if (op.res2 >= 0) readers[op.res2]++;
    if (op.sk_in >= 0) readers[op.sk_in]++;
  }
  bool changed = false;
  for (size_t i = 0; i + 1 < g->ops.size(); ++i) {
    const Op& d = g->ops[i];
    Op& e = g->ops[i + 1];
    const bool is_deconv = d.kind == OP_CONV && d.ow_mul == 2 && d.nsub == 2 && d.sub[0].deconv && d.res1 < 0 && d.sk_in < 0;
    if (!is_deconv || e.kind != OP_CONV || e.in != d.out || readers[d.out] != 1) continue;
    const int C = d.cin_t;
    if (!e.pair || !(e.fsq_fused || e.sk_in >= 0) || d.sub[0].cout != C || (C != 16 && C != 32 && C != 64)) continue;
    if (g->tensors[d.in].fmt != FMT_S16 || g->tensors[d.out].fmt != FMT_S16) continue;
    e.up_fused = true;
    e.up[0] = d.sub[0];
    e.up[1] = d.sub[1];
    e.in = d.in;
    g->ops.erase(g->ops.begin() + (long)i);
    changed = true;
  }
  if (changed) recompute_lifetimes(g);
}

// SqueezeSegV2's tail (nets/SqueezeSegV2.py:318-325): fire13's expand pair + skip branch feeds only conv14,
// whose logits feed only the head.  With the up-convolution already inside the pair (fuse_upconvs) the three
// run as ONE kernel that keeps fire13's 64-channel output in LDS (fire_head_kernel): 16 + 32 + 32 -> 64
// channels, up to 32 classes.
inline void fuse_head(Graph* g) {
  static const int on = debug_env("PCLSEG_FUSE_HEAD", 1);
  const size_t n = g->ops.size();
  if (!on || n < 2) return;
  Op& e = g->ops[n - 2];
  const Op& hd = g->ops[n - 1];
  if (hd.kind != OP_HEAD || e.kind != OP_CONV || hd.in != e.out || !e.pair || !e.up_fused || e.fsq_fused || e.sk_in < 0 ||
      e.res1 >= 0 || e.res2 >= 0 || e.nsub != 2) return;
  if (e.cin_t != 16 || e.sub[0].cout != 32 || e.sub[1].cout != 32 || e.sk.cout != 64 || hd.cin_t != 64 ||
      hd.sub[0].cout > 32 || hd.pkh != 3 || hd.pkw != 3) return;
  for (size_t i = 0; i + 2 < n; ++i)   // fire13's output must have no other reader
    if (g->ops[i].in == e.out || g->ops[i].res1 == e.out || g->ops[i].res2 == e.out) return;
  e.head_fused = true;
  e.hd = hd.sub[0];
  e.out = -1;
  g->ops.pop_back();
  recompute_lifetimes(g);
}

inline int resolve_micro_batch(const pclseg_desc& d) {
  if (d.micro_batch > 0) return d.micro_batch;
  const int64_t px = (int64_t)d.height * d.width;
  int64_t mb = (int64_t)(1 << 19) / std::max<int64_t>(px, 1);
  return (int)std::min<int64_t>(std::max<int64_t>(mb, 1), 16);
}

// Validate desc and build the graph.  Returns a pclseg_status.
inline int build_graph(const pclseg_desc* d, Graph* g) {
  if (!d) { g->error = "desc is NULL"; return PCLSEG_ERR_BAD_ARG; }
  g->desc = *d;
  if (d->arch < 0 || d->arch > PCLSEG_ARCH_DARKNET53) {
    g->error = "unknown arch " + std::to_string(d->arch);
    return PCLSEG_ERR_BAD_ARG;
  }
  if (d->height <= 0 || d->width <= 0) {
    g->error = "height and width must be positive";
    return PCLSEG_ERR_BAD_SHAPE;
  }
  if (d->num_class < 2 || d->num_class > 64) {
    g->error = "num_class must be in [2, 64]";
    return PCLSEG_ERR_BAD_ARG;
  }
  if (d->none_index < 0 || d->none_index >= d->num_class) {
    g->error = "none_index out of range";
    return PCLSEG_ERR_BAD_ARG;
  }
  for (int i = 0; i < 5; ++i)
    if (!(d->std[i] > 0.0) || !std::isfinite(d->mean[i])) {
      g->error = "std must be positive and mean finite";
      return PCLSEG_ERR_BAD_ARG;
    }
  int down = 16;  // four stride-2 stages along W, undone by four x2 transposed convs + skips
  if (d->arch != PCLSEG_ARCH_SQUEEZESEGV2) {
    if (d->output_stride != 8 && d->output_stride != 16 && d->output_stride != 32) {
      g->error = "output_stride must be 8, 16 or 32";
      return PCLSEG_ERR_BAD_ARG;
    }
    down = d->output_stride;
  }
  if (d->width % down != 0) {
    g->error = "width " + std::to_string(d->width) + " is not divisible by " + std::to_string(down) +
               " (encoder/decoder skip shapes would not match)";
    return PCLSEG_ERR_BAD_SHAPE;
  }
  g->micro_batch = resolve_micro_batch(*d);
  if (d->arch == PCLSEG_ARCH_SQUEEZESEGV2) build_squeezesegv2(g);
  else build_darknet(g, d->arch == PCLSEG_ARCH_DARKNET21 ? 21 : 53);
  assign_formats(g);
  if (d->arch == PCLSEG_ARCH_SQUEEZESEGV2) { fuse_upconvs(g); fuse_head(g); }
  // packed-parameter geometry: exact-f32 fragments, split-f16 fragments, biases
  for (Op& op : g->ops) {
    if (op.kind == OP_POOL) continue;
    if (op.sk_in >= 0) {  // [8][C] + [C]
      op.sk.b_off = g->packed_bias_floats;
      g->packed_bias_floats += (int64_t)9 * op.sk.cout;
    }
    if (op.kind == OP_CAM) {  // plain [C][R]+[R] and [R][C]+[C] float32 blocks
      const int C = op.cin_t, R = C / 16;
      if (op.fsq_fused) {     // + the fused squeeze as 1x1 fragments [C/32 steps][tiles][hi|lo][lane][8]
        op.fsq.nctp = (op.fsq.cout + 15) / 16;
        op.fsq.w16_off = g->packed16_halfs;
        g->packed16_halfs += (int64_t)(C / 32) * op.fsq.nctp * 1024;
        op.fsq.b_off = g->packed_bias_floats;
        g->packed_bias_floats += sub_bias_floats(op.fsq);
      }
      op.sub[0].b_off = g->packed_bias_floats;
      g->packed_bias_floats += (int64_t)C * R + R;
      op.sub[1].b_off = g->packed_bias_floats;
      g->packed_bias_floats += (int64_t)R * C + C;
      continue;
    }
    op_geometry(&op);
    if (op.fsq_fused) {
      if (!op.pair || op.nsub != 2 || op.sub[0].nctp != op.sub[1].nctp || op.sub[0].nctp != op.ntw * op.wn) {
        g->error = "internal: fused squeeze on an op that cannot run as one merged block (" + op.name() + ")";
        return PCLSEG_ERR_BAD_ARG;
      }
      op.fsq.nctp = (op.fsq.cout + 15) / 16;
      op.fsq.w16_off = g->packed16_halfs;
      g->packed16_halfs += fsq_w16_halfs(op);
      op.fsq.b_off = g->packed_bias_floats;
      g->packed_bias_floats += sub_bias_floats(op.fsq);
    }
    if (op.head_fused) {   // conv14's fragments, packed with ONE 64-channel chunk: 18 K-steps x nctp tiles
      op.hd.nctp = (op.hd.cout + 15) / 16;
      op.hd.w16_off = g->packed16_halfs;
      g->packed16_halfs += (int64_t)18 * op.hd.nctp * 1024;
      op.hd.b_off = g->packed_bias_floats;
      g->packed_bias_floats += sub_bias_floats(op.hd);
      op.skm = op.sk;                     // one K-step x 4 cout tiles
      op.skm.nctp = (op.sk.cout + 15) / 16;
      op.skm.w16_off = g->packed16_halfs;
      g->packed16_halfs += (int64_t)op.skm.nctp * 1024;
      op.skm.b_off = g->packed_bias_floats;
      g->packed_bias_floats += sub_bias_floats(op.skm);
    }
    if (op.up_fused) {   // the transposed conv's two parities, packed like any 2-tap sub-conv of this op
      for (int i = 0; i < 2; ++i) {
        SubOp& su = op.up[i];
        su.nctp = (su.cout + 15) / 16;
        su.w16_off = g->packed16_halfs;
        g->packed16_halfs += sub_w16_halfs(op, su);
        su.b_off = g->packed_bias_floats;
        g->packed_bias_floats += sub_bias_floats(su);
      }
    }
    for (int i = 0; i < op.nsub; ++i) {
      SubOp& su = op.sub[i];
      su.w32_off = g->packed32_floats;
      g->packed32_floats += sub_w32_floats(op, su);
      su.w16_off = g->packed16_halfs;
      g->packed16_halfs += sub_w16_halfs(op, su);
      su.b_off = g->packed_bias_floats;
      g->packed_bias_floats += sub_bias_floats(su);
    }
  }
  plan_workspace(g);
  return PCLSEG_OK;
}

}  // namespace pclseg
