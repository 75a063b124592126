// pclseg_kernels.h — CDNA4 (gfx950) device kernels of the forward pass.
//
// Layout: every activation is float32 NHWC with C a multiple of 4, so one lane moves one
// 16-byte channel quad and a run of pixels along W is one contiguous span in HBM.
//
// conv_kernel is the workhorse: an implicit-GEMM convolution on the matrix cores with the
// zero-padded input halo patch staged in LDS and the BatchNorm-folded weights streamed from L2
// in a pre-packed per-lane fragment order.  Two arithmetic modes:
//   F16X3 = false  v_mfma_f32_16x16x4_f32: exact float32 (bit-identical to an fmaf chain).
//   F16X3 = true   v_mfma_f32_16x16x32_f16 on split operands: every float32 value v is carried
//                  as hi = f16(v), lo = f16(v - hi) — 22 significant bits while lo is a normal half
//                  (|v| >= 2^-3), an absolute 2^-25 below; weights are pre-scaled per 16-cout tile so
//                  that all of them sit in the 22-bit regime (pclseg_api.hip: scale_exponent) — and a
//                  product is hi*hi + hi*lo + lo*hi with float32 accumulation: float32-class accuracy
//                  (|logit error| 1-4e-5 through the whole network) at 3/16 of the f32 MFMA cost.
// One launch covers
//   Conv2D 3x3 / 1x1, strides (1,1) and (1,2), TF "SAME" padding      (SURVEY.md K2,K3,K4)
//   Conv2DTranspose (1,4)/(1,2): both output parities as two 2-tap sub-convs     (K5)
//   FIRE's expand1x1 || expand3x3 pair as two sub-convs over one staged patch
//   fused epilogues: +bias(BN) -> relu / leaky(0.1) / sigmoid -> (*gate | +residual) -> +skip,
//   written into a channel slice of the destination (tf.concat for free)         (K8)
//   and the segmentation head: 3x3 conv -> [softmax] -> argmax -> mask            (K9)
//
// GEMM orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel].  With the 16x16 lane maps
// (A: row = lane&15; B: col = lane&15; D: col = lane&15, row = 4*(lane>>4)+reg) a lane ends up
// holding 4 CONSECUTIVE output channels of ONE pixel, so the epilogue is 16-byte loads/stores
// along the channel axis.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// CANDIDATE kernel variants: written in rounds 4-5, bit-identical to the shipped code paths on the functional simulator,
// but never run on an MI355X — the shipped library (plain `make`) does not contain them: its device kernels are
// instruction-identical to the set round 3 verified and measured on hardware (scripts/kernel_isa_diff.py ad2e081).
// `make candidates` (-DPCLSEG_CAND) builds build/libpclseg_cand.so with all of them for the hardware A/B of
// scripts/gpu_all.sh; each has its own switch for per-component builds (make variant NAME=tail EXTRA=-DPCLSEG_CAND_TAIL).
// A candidate moves out of this block (and its #else branch is deleted) only when it is bit-identical AND not slower
// on the device.
#ifdef PCLSEG_CAND
#define PCLSEG_CAND_TAIL 1   // fire_head_kernel: batched up-convolution, prefetched epilogue operands, pipelined conv14 sweep
#define PCLSEG_CAND_CAM 1    // cam_kernel SQ: fused-squeeze fragments requested one gate pass ahead
#define PCLSEG_CAND_SLAB 1   // conv_kernel FSQ: fire8/9's partial-sum slab in two passes (70 KB instead of 136 KB)
#define PCLSEG_CAND_WIDE 1   // conv1x1_wide_kernel for Darknet's wide 1x1 layers
#define PCLSEG_CAND_KPIPE 1  // conv_kernel GEOM 1, fire8/9's 64-pixel merged pairs: fragment reads one K-step ahead of the MFMAs
#define PCLSEG_CAND_GEOM2 1  // conv_kernel GEOM 2: compile-time geometry K loop for the 32-channel merged pairs (fire4, fire11)
#define PCLSEG_CAND_EXACTEPI 1  // exact-float32 mode: own instantiations of the plain and the residual epilogue (pclseg_api.hip)
#endif

namespace pclseg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_SIGMOID = 3 };

constexpr int kConvThreads = 256;  // 4 waves, one per SIMD
constexpr int kPadF32 = 4;         // floats of per-pixel LDS padding (bank spread)
constexpr int kPadF16 = 8;         // halfs of per-pixel LDS padding
constexpr int kStageBatch = 4;     // global loads a thread keeps in flight while staging

// A sub-convolution: a window of taps inside the staged patch, with its own weights / bias /
// activation / destination channel slice / output column phase.
struct ConvSub {
  const float* w32;     // exact mode: [tap][c16][ct][lane][4] floats
  const _Float16* w16;  // f16x3 mode: [chunk-step][ct][hi|lo][lane][8] halfs
  const float* bias;    // [nctp*16]
  int Cout, nctp, ny;   // ny = blocks along grid.y = nctp / (NTW*WN)
  int co_off;           // destination channel offset
  int th0, tw0, nkh, nkw;  // tap window inside the patch
  int ow_off;           // output column = j*ow_mul + ow_off
  int act;
};

struct ConvArgs {
  const float* in;    // [N,H,Win,Cin]
  float* out;         // [N,H,Wout,out_C]
  const float* res1;  // optional, [N,H,Wout,res1_C], read at co_off + co
  const float* res2;  // optional
  const float* skx;   // optional fused skip branch: BN(conv1x1(skx)) added after the activation;
  const float* skw;   //   skx [N,H,Wout,8], skw = [8][out_C] folded weights then [out_C] bias
  const uint8_t* mask;  // head only
  int32_t* preds;       // head only
  float* probs;         // head only, optional
  float* logits;        // head only, optional
  int N, H, Win, Wout, Wconv;
  int Cin, out_C, res1_C, res2_C;
  int sw, pt, pl, ow_mul;
  int TH, SEGW, PH, PW, tilesH, tilesW;
  int res1_mul, none_index;
  int CK;      // channels staged per LDS pass (f16x3: 64/32/16, exact: 32/16)
  int inv_pw;  // ceil(2^20 / PW): pixel index -> patch row by multiply-shift
  int nsub;
  int ny;      // blocks per pixel tile (all sub-convs' cout groups)
  int group_major;  // block order, see conv_kernel
  int in_s16;   // input tensor is in split-f16 pair format (see below), else float32
  int out_s16;  // write the output in split-f16 pair format
  unsigned* range_flag;  // sticky: set when a value that is split to f16 hi/lo has |v| >= 65504
#ifdef PCLSEG_WITH_STAMPS
  unsigned long long* stamps;   // debug build (make stamps): s_memtime at the phase boundaries, [block][8]
#endif
  int wt;                // write-through output stores (see store_quad)
  // fused squeeze of the NEXT FIRE module (conv_kernel FSQ): packed fragments, bias, couts, cout groups
  const _Float16* fsq_w16;
  const float* fsq_bias;
  int fsq_q, fsq_ncg;
  // fused FIREUP transposed convolution (conv_kernel UP): `in` is the module's squeeze output at HALF the
  // width (split-f16), the block deconvolves its patch while staging; packed fragments per output parity
  const _Float16* up_w16[2];
  const float* up_bias;
  int up_Win, up_nctp;
  int up_lds_off;        // byte offset of the source-patch copy in dynamic LDS
  int flip_bit;          // merged pairs: blocks with this bit of blockIdx.x set run the 1x1 half first (-1: none)
  int skw_lds_off;       // byte offset of the fused skip branch's [9][out_C] weights in dynamic LDS
  ConvSub sub[2];
};

// Split-f16 pair format ("S16") of an activation tensor [N,H,W,C], C % 8 == 0: per pixel C halfs
// hi = f16(v) followed by C halfs lo = f16(v - hi) — 4*C bytes per pixel, exactly the float32
// footprint.  Tensors whose only reader is the LDS staging of one convolution (FIRE squeeze and
// up-convolution outputs, Darknet's 1x1 bottlenecks) are kept in this form: the producer splits
// every element once in its epilogue and the consumer's staging becomes a plain 16-byte copy
// instead of a convert that every cout-group block of a tile would repeat.
constexpr float kF16Max = 65504.0f;

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)(v[e] - (float)hi[e]);
  }
}
// Output stores.  A plain store leaves the line dirty in the XCD's L2 until the end-of-kernel
// write-back; `sc1` writes through.  Measured per layer: write-through takes
// 2-3.5 us (10-12 %) off the FIRE expand pairs that only write (fire2-fire7), and ADDS 6-8 us to the
// ones that also read a skip tensor (fire10-fire13) — hence a per-launch switch (ConvArgs::wt).
__device__ __forceinline__ void store_quad(float* dst, const f32x4 v, const bool write_through) {
#ifndef PCLSEG_SIM
  if (write_through) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
  else
#endif
    *reinterpret_cast<f32x4*>(dst) = v;
}

// Accumulator -> pre-activation value.  Split-f16 weights are packed with a power-of-two scale 2^k per
// 16-output-channel tile (pclseg_api.hip: scale_exponent), so the accumulator holds 2^k times the
// convolution sum: one fmaf with the tile's 2^-k and the bias undoes it exactly (a power of two never
// rounds).  The tile index is wave-uniform, so 2^-k comes through the scalar cache into an SGPR
// (constant address space: the parameter blob is never written while a kernel runs) — no vector
// register, no vector-memory instruction.
#ifndef PCLSEG_SIM
typedef __attribute__((address_space(4))) const float* const_f32_ptr;
#else
typedef const float* const_f32_ptr;
#endif
__device__ __forceinline__ float sload(const float* p) { return *(const_f32_ptr)(p); }
__device__ __forceinline__ f32x4 fma4(const f32x4 a, const float s, const f32x4 b) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = fmaf(a[e], s, b[e]);
  return r;
}

// Workgroup barrier that leaves this wave's global loads in flight: __syncthreads() drains vmcnt too, which
// would turn every prefetch issued before it into a stall at it.  LDS traffic is ordered by lgkmcnt.
#ifndef PCLSEG_SIM
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else   // (functional simulator, sim/: tests only)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#endif

__device__ __forceinline__ float absmax4(float m, const f32x4 v) {
  m = fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1])));
  return fmaxf(m, fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// Blocks are dealt round-robin over the 8 XCDs (private L2 each).  Remap the linear block id so
// every XCD works on one CONTIGUOUS range of logical ids: neighbouring tiles (shared halo) and the
// cout groups of one tile (same input patch) then run back to back on the same L2.  Bijective
// for any grid size; a pure speed choice, results never depend on it.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// Activation of a channel quad.  `act` is wave-uniform, so this is three scalar branches around
// four-wide bodies (a per-element switch with a precise expf/division unrolled over every
// accumulator register made the epilogue two thirds of the kernel's code).  The sigmoid runs on the
// transcendental unit (v_exp_f32 + v_rcp_f32, ~1 ulp each).
__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
  if (act == ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
  } else if (act == ACT_LRELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.1f * v[e];
  } else if (act == ACT_SIGMOID) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      v[e] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v[e]));
  }
  return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_LRELU: return v > 0.0f ? v : 0.1f * v;
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// Segmentation head of one pixel (reference: nets/SegmentationNetwork.py:58-69): [softmax ->] argmax ->
// mask.  lv[nn] holds this lane's 4 logits of cout tile nn (channels nn*16 + 4g .. +3); the NTW tiles
// hold all NUM_CLASS logits of a pixel across the 4 lanes {p, p+16, p+32, p+48}.  A NaN or +inf logit
// makes every softmax probability of the pixel NaN, and tf.argmax over all-NaN probabilities yields
// index 0: such pixels get class 0, never an out-of-range id.  Must be called from convergent code.
template <int NTW>
__device__ __forceinline__ void head_finish(const f32x4 (&lv)[NTW], const bool valid, const size_t pix, const int g,
                                            const int NC, const uint8_t mask_px, int32_t* preds, float* probs,
                                            float* logits, const int none_index) {
  // (written with selects, not per-element branches: `co < NC` differs per lane group, and every divergent
  // `if` costs a saveexec / branch / restore triple in a kernel that is bound by instruction issue)
  float val[NTW * 4];
  float best = -INFINITY;
  int bi = 0;
  bool bad = false;
#pragma unroll
  for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = nn * 16 + g * 4 + i;
      const bool in = co < NC;
      const float v = in ? lv[nn][i] : -INFINITY;      // (classes beyond NC never win and never count as "bad")
      val[nn * 4 + i] = v;
      bad = bad || (in && !(v < INFINITY));             // NaN or +inf
      const bool up = v > best;
      best = up ? v : best;
      bi = up ? co : bi;
    }
  if (logits) {   // (uniform)
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = nn * 16 + g * 4 + i;
        if (valid && co < NC) logits[pix * NC + co] = lv[nn][i];
      }
  }
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float ov = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    const bool take = ov > best || (ov == best && oi < bi);
    best = take ? ov : best;
    bi = take ? oi : bi;
  }
  {
    int b = bad ? 1 : 0;
    b |= __shfl_xor(b, 16);
    b |= __shfl_xor(b, 32);
    bad = b != 0;
  }
  if (probs) {   // (uniform)
    // softmax materialised: exp(x - max) / sum, and the argmax is taken over the
    // probabilities exactly as the reference does (lowest index wins ties).
    float sum = 0.f;
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (nn * 16 + g * 4 + i < NC) sum += expf(val[nn * 4 + i] - best);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    float pbest = -1.f;
    int pbi = 0;
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = nn * 16 + g * 4 + i;
        if (co < NC) {
          const float pr = bad ? NAN : expf(val[nn * 4 + i] - best) / sum;
          if (valid) probs[pix * NC + co] = pr;
          if (pr > pbest) { pbest = pr; pbi = co; }
        }
      }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float ov = __shfl_xor(pbest, off);
      const int oi = __shfl_xor(pbi, off);
      if (ov > pbest || (ov == pbest && oi < pbi)) { pbest = ov; pbi = oi; }
    }
    bi = pbi;
  }
  if (bad) bi = 0;
  if (g == 0 && valid) preds[pix] = mask_px ? bi : none_index;
}

// Block = 4 waves arranged WM x WN (WM = 4/WN): wave (wm, wn) owns MTW (2 or 4) pixel segments
// {wm*MTW ..} and NTW 16-cout tiles {wn*NTW ..}.  WN = 2 halves the weight fragments each
// wave streams from L2 (the block shares one 64-cout group), WN = 1 keeps all couts of a pixel in
// one wave (needed by the head's argmax) on a 256-pixel block.
// EPI: which epilogue operands this instantiation carries registers for:
//   0 none, 1 res1, 2 res1 + res2, 3 fused skip branch (skx), 4 all of them.
// PAIR (split-f16 mode, FIRE expand pair, single channel chunk): the block computes cout group `by`
// of the 3x3 sub-conv and then the SAME cout group of the 1x1 sub-conv from the one staged patch
// (the 1x1 is the centre tap), so the pair needs half the blocks and stages half as often.
// NW = waves per block: 4 (256 threads, four co-resident blocks per CU) or 8 (512 threads, large
// register tiles, two waves per SIMD that alternate between LDS reads and MFMA bursts): with
// NW = 8 a block covers (8/WN)*MTW pixel segments x WN*NTW cout tiles per sub-conv, so the deep
// FIRE layers stage each patch once for 128-256 output channels instead of once per 64.
// FSQ = NQ > 0 (merged pair, 8 waves): the block does NOT write the pair's output.  It applies the NEXT
// FIRE module's squeeze (1x1 conv C -> 16*NQ + BN + ReLU, nets/SqueezeSegV2.py:96-104,123-124) to its
// tile and writes only that (split-f16): the 12-17 MB/scan tensors between fire4..fire10 are never
// materialised.  The block holds all C channels of its pixels, split over the waves' cout groups:
//   - a wave's bias+ReLU'd accumulators ARE an MFMA B operand as they stand — lane (p, g) holds
//     channels {4g..4g+3} of tile 2t and of tile 2t+1, eight k-values of a 32-deep step — once the
//     squeeze weights are packed in that channel order (host: pack_fsq); no data movement;
//   - each wave accumulates the partial squeeze over ITS channels (both halves), the partials of the
//     WN waves that share a pixel group are summed through LDS in a fixed order (deterministic), then
//     bias + ReLU + split + store.
// UP = C/16 > 0 (merged pair of a FIREUP module, C = squeeze channels): the module's Conv2DTranspose
// (1,4)/(1,2) + ReLU (nets/SqueezeSegV2.py:176-183,194) runs inside the staging: instead of copying an
// up-convolved patch from memory, the waves compute it — 16 patch pixels of one output parity per MFMA
// tile, K = 2 taps x C straight from the half-width squeeze tensor — and write it to LDS in the staged
// split-f16 layout.  Four launches and the four up-convolved tensors disappear.
// LW = loader waves (8-wave single-conv blocks of Darknet's wide layers, one block per CU): LW extra waves do
// nothing but stage the NEXT channel chunk's patch into a second LDS buffer while the NW compute waves run the
// current chunk's K loop — one barrier per chunk, no staging in the compute waves, and the loaders' memory
// round trips are counted in their own vmcnt.  Measured need (s_memtime stamps, round 3): in the 512 -> 1024
// 3x3 layers a chunk's K loop runs at 99 % of the matrix rate (28 k cycles) but the staging between two K
// loops takes 14.5 k cycles with idle matrix cores — every block of the chip stages at the same moment, one
// block per CU (147-226 registers), so nothing overlaps it.  Prefetching the next chunk into registers of the
// compute waves does not help (+0.4 %): vmcnt retires in order, so the K loop's first weight-fragment wait
// also waits for the prefetch.  Capping the registers at 128 for two blocks per CU spills (+3.7 % only, and
// the two blocks stage in lockstep anyway).
// OCC128: 8-wave single-conv variant whose register allocation is capped at 128 (two blocks per CU).
// GEOM = 1 (with OCC128, 8 x 16-pixel tiles): the geometry of Darknet's heavy layers — 3x3, stride 1, full
// 64-channel chunks (patch 10 x 18 pixels, 72 halfs per pixel and plane) — is a compile-time constant, so
// the K loop is fully unrolled and EVERY activation-fragment address is one per-lane base register plus an
// immediate: the generic loop spends ~19 vector adds, ~15 register moves and ~15 waits per K-step on them
// beside 48 MFMAs, and at two to four waves per SIMD that issue traffic is what held its K loop at 72 % of
// the matrix rate (s_memtime stamps, round 3).
template <int MTW, int NTW, int WN, bool HEAD, bool F16X3, int EPI, bool PAIR = false, int NW = 4, int FSQ = 0, int UP = 0, int LW = 0, bool OCC128 = false, int GEOM = 0>
__global__ __launch_bounds__((NW + LW) * 64, LW ? 3 : ((NW == 8 && !OCC128) ? 2 : 4)) void conv_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int kThreads = NW * 64;
  const int tid = threadIdx.x;
#ifdef PCLSEG_WITH_STAMPS
  auto stamp = [&](int i) { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime(); };
#else
  auto stamp = [](int) {};
#endif
  stamp(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % WN, wm = wave / WN;
  const int p = lane & 15;  // pixel within the 16-pixel segment
  const int g = lane >> 4;  // k-group (operands) / cout quad (accumulator)

  // Each XCD works on a contiguous range of logical ids.  tile-major (id = tile * ny + group): the
  // cout groups of one tile run back to back and share the staged input patch in L2 (activation-
  // heavy layers).  group-major (id = group * ntiles + tile): the ~128 blocks resident on an XCD
  // stream the SAME weight fragments, which then stay in its 4 MB L2 (weight-heavy layers).
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int tile, by;
  if (a.group_major) {
    const int ntiles = gridDim.x / a.ny;
    by = lid / ntiles;
    tile = lid - by * ntiles;
  } else {
    tile = lid / a.ny;
    by = lid - tile * a.ny;
  }
  const int twi = tile % a.tilesW;
  tile /= a.tilesW;
  const int thi = tile % a.tilesH;
  const int n = tile / a.tilesH;
  const int h0 = thi * a.TH;
  const int w0 = twi * (a.SEGW * 16);
  const int si = PAIR ? 1 : ((a.nsub > 1 && by >= a.sub[0].ny) ? 1 : 0);
  const ConvSub& S = a.sub[si];
  const int ct0 = (by - ((!PAIR && si) ? a.sub[0].ny : 0)) * (NTW * WN) + wn * NTW;

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* in_n = a.in + (size_t)n * a.H * a.Win * a.Cin;
  const int npix = a.PH * a.PW;
  float vmax = 0.f;  // largest |value| this thread split to f16 hi/lo (range guard)

  // this wave's segments: (row, column-segment) inside the tile
  int seg_r[MTW], seg_q[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int seg = wm * MTW + m;
    seg_r[m] = seg / a.SEGW;
    seg_q[m] = seg - seg_r[m] * a.SEGW;
  }

  // ---------------------------------------------------------------------------- epilogue
  // The fused skip branch's folded 1x1 weights + bias ([9][out_C] floats) live in LDS behind the
  // patch: fetched once per block, read in the epilogue on the LDS counter (not behind stores).
  constexpr bool kSkip = !HEAD && (EPI == 3 || EPI == 4);
  const float* skw_lds = reinterpret_cast<const float*>(smem_raw + a.skw_lds_off);
  if constexpr (kSkip) if (a.skx) {
    f32x4* dst = reinterpret_cast<f32x4*>(smem_raw + a.skw_lds_off);
    for (int i = tid; i < (9 * a.out_C) >> 2; i += kThreads)
      dst[i] = *reinterpret_cast<const f32x4*>(a.skw + i * 4);
  }
  // (mlo, mhi: the segments to finish — the register-capped 8-segment variant does it in two halves)
  auto epilogue = [&](const ConvSub& E, f32x4 (&ac)[MTW][NTW], const int mlo = 0, const int mhi = MTW) {
    // opaque copies of the lane coordinates: everything the epilogue derives from them is computed
    // HERE, not hoisted above the K loop where it would cost registers for its whole duration
    int p = lane & 15, g = lane >> 4;
#ifndef PCLSEG_SIM
    asm volatile("" : "+v"(p), "+v"(g));
#endif
    f32x4 bv[NTW];
    float iv[F16X3 ? NTW : 1];   // (split-f16) inverse weight scale of each cout tile, scalar
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) {
      bv[nn] = *reinterpret_cast<const f32x4*>(E.bias + (ct0 + nn) * 16 + g * 4);
      if constexpr (F16X3) iv[nn] = sload(E.bias + (E.nctp + ct0 + nn) * 16);
    }
    auto pre = [&](const f32x4 acv, const int nn) -> f32x4 {
      if constexpr (F16X3) return fma4(acv, iv[nn], bv[nn]); else return acv + bv[nn];
    };
    if constexpr (!HEAD) {
      // vmcnt retires loads AND stores in issue order, so a residual load issued after a store
      // would wait for that store's acknowledgement: tile after tile, the epilogue would pay a full
      // memory round trip each.  Hence two passes: every residual / skip operand of the whole
      // accumulator tile is fetched first, then all tiles are finished and stored back to back.
      constexpr bool kR1 = EPI == 1 || EPI == 2 || EPI == 4;
      constexpr bool kR2 = EPI == 2 || EPI == 4;
      constexpr bool kSk = EPI == 3 || EPI == 4;
      f32x4 r1[kR1 ? MTW : 1][NTW], r2[kR2 ? MTW : 1][NTW], sx0[kSk ? MTW : 1], sx1[kSk ? MTW : 1];
      size_t pixm[MTW];
      bool validm[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        if (m < mlo || m >= mhi) continue;
        const int oh = h0 + seg_r[m];
        const int j = w0 + seg_q[m] * 16 + p;
        validm[m] = (oh < a.H) && (j < a.Wconv);
        pixm[m] = ((size_t)n * a.H + oh) * a.Wout + (j * a.ow_mul + E.ow_off);
      }
      if constexpr (kR1) if (a.res1) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int nn = 0; nn < NTW; ++nn) {
            if (m < mlo || m >= mhi) continue;
            const int co = (ct0 + nn) * 16 + g * 4;
            const bool ok = validm[m] && co < E.Cout;
            r1[m][nn] = *reinterpret_cast<const f32x4*>(ok ? a.res1 + pixm[m] * a.res1_C + E.co_off + co : a.res1);
          }
      }
      if constexpr (kR2) if (a.res2) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int nn = 0; nn < NTW; ++nn) {
            if (m < mlo || m >= mhi) continue;
            const int co = (ct0 + nn) * 16 + g * 4;
            const bool ok = validm[m] && co < E.Cout;
            r2[m][nn] = *reinterpret_cast<const f32x4*>(ok ? a.res2 + pixm[m] * a.res2_C + E.co_off + co : a.res2);
          }
      }
      if constexpr (kSk) if (a.skx) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if (m < mlo || m >= mhi) continue;
          const float* xp = validm[m] ? a.skx + pixm[m] * 8 : a.skx;
          sx0[m] = *reinterpret_cast<const f32x4*>(xp);
          sx1[m] = *reinterpret_cast<const f32x4*>(xp + 4);
        }
      }
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn) {
        const int co = (ct0 + nn) * 16 + g * 4;
        // SqueezeSegV2's conv1_skip + bn1_skip (nets/SqueezeSegV2.py:293,319) evaluated here from
        // the 8-channel network input instead of round-tripping a 64-channel tensor: this lane's
        // 8x4 weight block + bias (9 quads, from LDS) is shared by its MTW pixels
        f32x4 skwv[kSk ? 9 : 1];
        if constexpr (kSk) if (a.skx && co < E.Cout) {
#pragma unroll
          for (int c = 0; c < 9; ++c)
            skwv[c] = *reinterpret_cast<const f32x4*>(skw_lds + E.co_off + co + c * a.out_C);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          if (m < mlo || m >= mhi) continue;
          if (validm[m] && co < E.Cout) {
            f32x4 v = act4(pre(ac[m][nn], nn), E.act);
            if constexpr (kR1) if (a.res1) v = a.res1_mul ? v * r1[m][nn] : v + r1[m][nn];
            if constexpr (kR2) if (a.res2) v += r2[m][nn];
            if constexpr (kSk) if (a.skx) {
              f32x4 z = skwv[8];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = fmaf(sx0[m][c], skwv[c][e], z[e]);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = fmaf(sx1[m][c], skwv[4 + c][e], z[e]);
              }
              v += z;
            }
            if (F16X3 && a.out_s16) {
              f16x4 hi, lo;
              split4(v, hi, lo);
              vmax = absmax4(vmax, v);
              _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + pixm[m] * (size_t)(2 * a.out_C) + E.co_off + co;
              *reinterpret_cast<f16x4*>(o16) = hi;
              *reinterpret_cast<f16x4*>(o16 + a.out_C) = lo;
            } else {
              store_quad(a.out + pixm[m] * a.out_C + E.co_off + co, v, a.wt != 0);
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < MTW; ++m) {
        const int oh = h0 + seg_r[m];
        const int j = w0 + seg_q[m] * 16 + p;
        const bool valid = (oh < a.H) && (j < a.Wconv);
        const int ow = j * a.ow_mul + E.ow_off;
        const size_t pix = ((size_t)n * a.H + oh) * a.Wout + ow;
        f32x4 lv[NTW];
#pragma unroll
        for (int nn = 0; nn < NTW; ++nn) lv[nn] = pre(ac[m][nn], nn);
        head_finish<NTW>(lv, valid, pix, g, E.Cout, valid ? a.mask[pix] : (uint8_t)0, a.preds, a.probs, a.logits, a.none_index);
      }
    }
  };

  if constexpr (!F16X3) {
    // ------------------------------------------------------------ exact float32 matrix cores
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int nc16 = (a.Cin + 15) >> 4;
    const int cinp = nc16 * 16;
    const int CS = (cinp < a.CK ? cinp : a.CK) + kPadF32;  // floats per patch pixel
    const int ntaps = S.nkh * S.nkw;
    for (int c0 = 0; c0 < cinp; c0 += a.CK) {
      const int ckp = (cinp - c0) < a.CK ? (cinp - c0) : a.CK;
      const int cqn = ckp >> 2;
      if (c0) __syncthreads();
      // cqn is 4 or 8: thread -> (fixed channel quad, strided pixels); loads are issued in
      // batches of kStageBatch so their latencies overlap
      const int lq = cqn == 8 ? 3 : 2;
      const int sq = tid & (cqn - 1);
      const int spstep = kThreads >> lq;
      const int c = c0 + sq * 4;
      const bool cok = c < a.Cin;
      for (int pb = tid >> lq; pb < npix; pb += kStageBatch * spstep) {
        f32x4 v[kStageBatch];
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k) {
          const int pix = pb + k * spstep;
          const int pr = (int)(((unsigned)pix * (unsigned)a.inv_pw) >> 20);
          const int pc = pix - pr * a.PW;
          const int h = h0 - a.pt + pr;
          const int w = w0 * a.sw - a.pl + pc;
          const bool ok = cok && pix < npix && h >= 0 && h < a.H && w >= 0 && w < a.Win;
          const float* src = ok ? in_n + ((size_t)h * a.Win + w) * a.Cin + c : in_n;
          const f32x4 t = *reinterpret_cast<const f32x4*>(src);
          v[k] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < kStageBatch; ++k) {
          const int pix = pb + k * spstep;
          if (pix < npix) *reinterpret_cast<f32x4*>(smem + pix * CS + sq * 4) = v[k];
        }
      }
      __syncthreads();
      const int nsubk = ckp >> 4;
      for (int t = 0; t < ntaps; ++t) {
        const int ti = t / S.nkw;
        const int th = S.th0 + ti;
        const int tw = S.tw0 + (t - ti * S.nkw);
        for (int sk = 0; sk < nsubk; ++sk) {
          const int c16 = (c0 >> 4) + sk;
          const float* wb = S.w32 + ((size_t)(t * nc16 + c16) * S.nctp + ct0) * 256 + lane * 4;
          f32x4 wv[NTW];
#pragma unroll
          for (int nn = 0; nn < NTW; ++nn) wv[nn] = *reinterpret_cast<const f32x4*>(wb + nn * 256);
          f32x4 xv[MTW];
#pragma unroll
          for (int m = 0; m < MTW; ++m) {
            const int pcol = (seg_q[m] * 16 + p) * a.sw + tw;
            const int prow = seg_r[m] + th;
            xv[m] = *reinterpret_cast<const f32x4*>(smem + (prow * a.PW + pcol) * CS + sk * 16 + g * 4);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
              for (int nn = 0; nn < NTW; ++nn)
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nn][j], xv[m][j], acc[m][nn], 0, 0, 0);
        }
      }
    }
    epilogue(S, acc);
  } else {
    // ------------------------------------------------------------ split-f16 matrix cores
    // LDS: two planes (hi, lo) of [PH*PW][CSh] halfs.  K runs over (tap, 8-channel group)
    // pairs: lane group g of K-step s owns pair kidx = 4s + g, so taps with few channels
    // (Cin = 16, 48) still fill the 32-deep MFMA.
    _Float16* sm = reinterpret_cast<_Float16*>(smem_raw);
    const int cin8 = (a.Cin + 7) >> 3;
    const int ck8_full = a.CK >> 3;
    // staged 16-byte units per pixel: next power of two of the widest chunk (zero-filled), so the
    // staging index math is shifts only.  float32 input: a unit is a float4 channel quad that
    // becomes 4 hi + 4 lo halfs; split-f16 input: a unit is 8 hi or 8 lo halfs, copied as is
    // (the first half of the unit indices addresses the hi plane, the second the lo plane)
    const int qmax = (cin8 < ck8_full ? cin8 : ck8_full) * 2;
    int lq = 1;
    while ((1 << lq) < qmax) ++lq;
    const int qs = 1 << lq;
    const int CSh = qs * 4 + kPadF16;  // halfs per patch pixel
    const int plane = npix * CSh;      // halfs per plane
    int pixoff[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
      pixoff[m] = (seg_r[m] * a.PW + (seg_q[m] * 16 + p) * a.sw) * CSh;

    // staging threads: all of the block, or (LW > 0) the LW loader waves behind the NW compute waves
    constexpr int kStageThreads = LW ? LW * 64 : kThreads;
    const int stid = LW ? tid - kThreads : tid;
    const int sq = stid & (qs - 1);        // this thread's unit (fixed: kStageThreads % qs == 0)
    const int spix0 = stid >> lq;
    const int spstep = kStageThreads >> lq;
    const int hbase = h0 - a.pt, wbase = w0 * a.sw - a.pl;

    // stage the channel chunk [c8_0, c8_0 + ck8) (8-channel groups) of the patch into LDS
    auto stage = [&](const int c8_0, const int ck8, _Float16* const sm) {   // (sm: the LDS buffer to fill)
      // loads a thread keeps in flight: loader waves have nothing else to do and a chunk's whole share
      // (<= 12 units of a 10 x 18 patch x 64 channels over 256 threads) must be ONE round trip, not three
      constexpr int kB = LW ? 12 : kStageBatch;
      if (a.in_s16) {
        const int hq = qs >> 1;
        const int pl_sel = sq >= hq ? 1 : 0;
        const int c8u = sq & (hq - 1);
        const bool cok = c8u < ck8;
        const _Float16* in16 = reinterpret_cast<const _Float16*>(in_n);  // [H,W][hi Cin | lo Cin]
        const int coff = pl_sel * a.Cin + (c8_0 + c8u) * 8;
        _Float16* dbase = sm + pl_sel * plane + c8u * 8;
        for (int pb = spix0; pb < npix; pb += kB * spstep) {
          f16x8 v[kB];
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int pix = pb + k * spstep;
            const int pr = (int)(((unsigned)pix * (unsigned)a.inv_pw) >> 20);
            const int pc = pix - pr * a.PW;
            const int h = hbase + pr;
            const int w = wbase + pc;
            const bool ok = cok && pix < npix && h >= 0 && h < a.H && w >= 0 && w < a.Win;
            const _Float16* src = ok ? in16 + ((size_t)h * a.Win + w) * (size_t)(2 * a.Cin) + coff : in16;
            const f16x8 t = *reinterpret_cast<const f16x8*>(src);
            v[k] = ok ? t : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
          }
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int pix = pb + k * spstep;
            if (pix < npix && cok) *reinterpret_cast<f16x8*>(dbase + pix * CSh) = v[k];
          }
        }
      } else {
        const int c = c8_0 * 8 + sq * 4;
        const bool cok = (c < a.Cin) && (sq < ck8 * 2);
        for (int pb = spix0; pb < npix; pb += kB * spstep) {
          f32x4 v[kB];
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int pix = pb + k * spstep;
            const int pr = (int)(((unsigned)pix * (unsigned)a.inv_pw) >> 20);
            const int pc = pix - pr * a.PW;
            const int h = hbase + pr;
            const int w = wbase + pc;
            const bool ok = cok && pix < npix && h >= 0 && h < a.H && w >= 0 && w < a.Win;
            const float* src = ok ? in_n + ((size_t)h * a.Win + w) * a.Cin + c : in_n;
            const f32x4 t = *reinterpret_cast<const f32x4*>(src);
            v[k] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int pix = pb + k * spstep;
            f16x4 hi, lo;
            split4(v[k], hi, lo);
            vmax = absmax4(vmax, v[k]);
            if (pix < npix) {
              _Float16* dst = sm + pix * CSh + sq * 4;
              *reinterpret_cast<f16x4*>(dst) = hi;
              *reinterpret_cast<f16x4*>(dst + plane) = lo;
            }
          }
        }
      }
    };

    // FIREUP: up-convolve the patch from the half-width squeeze tensor (see UP above).  Output column w
    // = 2j + parity reads x[j - 1 + parity] (tap 0) and x[j + parity] (tap 1) with the kernel taps the
    // packed fragments of that parity carry; pixels outside the image are the expand conv's zero padding.
    auto stage_up = [&]() {
      // (1) the source patch — PH rows x (PW/2 + 1) half-width columns, zero outside the image — is copied
      //     to LDS behind the main patch, 16-byte units, consecutive lanes on consecutive bytes;
      // (2) wave <-> (output parity, 16-cout tile): its 2 x UP weight fragments are fetched once (in flight
      //     during (1)) and stay in registers while it walks the 16-pixel units of that parity, B operands
      //     from the LDS copy (units are dealt to the NSL waves that share a pair when 2*UP < NW).
      constexpr int NCT = UP > 0 ? UP : 1;
      constexpr int C = 16 * NCT, ck8 = C >> 3, nsteps = NCT;   // (host-checked: a.Cin == 16 * UP)
      constexpr int NPAIR = 2 * NCT, NSL = NW / NPAIR > 0 ? NW / NPAIR : 1;
      constexpr int SS = 2 * C + kPadF16;               // halfs per source pixel: [hi C | lo C | pad]
      constexpr int UPP = 2 * C / 8;                    // 16-byte units per source pixel (power of two)
      const int pairi = wave % NPAIR, slice = wave / NPAIR;
      const int parity = pairi / NCT, ct = pairi - parity * NCT;
      const int npc2 = a.PW >> 1;                       // patch columns of one parity (PW is even)
      const int SC = npc2 + 1;                          // source columns: j0 .. j0 + npc2
      const int j0 = (wbase - 1) >> 1;                  // = w0/2 - 1 (wbase = w0 - 1 is odd)
      _Float16* src_lds = reinterpret_cast<_Float16*>(smem_raw + a.up_lds_off);
      f16x8 wh[nsteps], wl[nsteps];
      {
        const _Float16* wq = a.up_w16[parity] + ct * 1024 + lane * 8;
#pragma unroll
        for (int s = 0; s < nsteps; ++s) {
          wh[s] = *reinterpret_cast<const f16x8*>(wq + (size_t)s * a.up_nctp * 1024);
          wl[s] = *reinterpret_cast<const f16x8*>(wq + (size_t)s * a.up_nctp * 1024 + 512);
        }
      }
      // (each parity's sub-conv has its own [bias | inverse scale] block, pclseg_graph.h: sub_bias_floats)
      const float* ubp = a.up_bias + (parity * 2 * a.up_nctp + ct) * 16 + g * 4;
      const f32x4 ub = *reinterpret_cast<const f32x4*>(ubp);
      const float ui = sload(a.up_bias + (parity * 2 * a.up_nctp + a.up_nctp + ct) * 16);
      {
        const _Float16* in16 = reinterpret_cast<const _Float16*>(a.in) + (size_t)n * a.H * a.up_Win * (size_t)(2 * C);
        const int nunits = a.PH * SC * UPP;
        const int inv_sc = (65536 + SC - 1) / SC;
        for (int i = tid; i < nunits; i += kThreads) {
          const int px = i / UPP, un = i - px * UPP;    // UPP is a power of two
          const int r = __mul24(px, inv_sc) >> 16, c = px - __mul24(r, SC);
          const int h = hbase + r, col = j0 + c;
          const bool ok = h >= 0 && h < a.H && col >= 0 && col < a.up_Win;
          f16x8 v = *reinterpret_cast<const f16x8*>(ok ? in16 + ((size_t)h * a.up_Win + col) * (size_t)(2 * C) + un * 8 : in16);
          if (!ok) v = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
          *reinterpret_cast<f16x8*>(src_lds + px * SS + un * 8) = v;
        }
      }
      __syncthreads();
      const int per = a.PH * npc2, upp = (per + 15) >> 4;
      const int inv_np = (65536 + npc2 - 1) / npc2;
      if (slice < NSL)
      for (int u = slice; u < upp; u += NSL) {
        const int l = u * 16 + p;
        const bool lv = l < per;
        const int pr = lv ? __mul24(l, inv_np) >> 16 : 0;
        const int k2 = lv ? l - __mul24(pr, npc2) : 0;
        const int pc = 2 * k2 + ((parity ^ wbase) & 1);
        const int h = hbase + pr, w = wbase + pc;
        const bool pv = lv && h >= 0 && h < a.H && w >= 0 && w < a.Win;
        // source column of tap 0 relative to j0: (w >> 1) - 1 + parity - j0
        const int c0 = ((w >> 1) - 1 + parity) - j0;
        f32x4 au = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < nsteps; ++s) {
          const int kidx = 4 * s + g;
          const int tap = kidx >= ck8 ? 1 : 0;
          const _Float16* sp = src_lds + (pr * SC + c0 + tap) * SS + (kidx - tap * ck8) * 8;
          const f16x8 xh = *reinterpret_cast<const f16x8*>(sp);
          const f16x8 xl = *reinterpret_cast<const f16x8*>(sp + C);
          au = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], xh, au, 0, 0, 0);
          au = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xl, au, 0, 0, 0);
          au = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xh, au, 0, 0, 0);
        }
        if (lv) {
          f32x4 v = fma4(au, ui, ub);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = pv ? fmaxf(v[e], 0.0f) : 0.0f;
          vmax = absmax4(vmax, v);
          f16x4 hi, lo;
          split4(v, hi, lo);
          _Float16* d = sm + (pr * a.PW + pc) * CSh + ct * 16 + g * 4;
          *reinterpret_cast<f16x4*>(d) = hi;
          *reinterpret_cast<f16x4*>(d + plane) = lo;
        }
      }
    };

    auto kloop = [&](const ConvSub& K, const int chunk, const int ck8, const _Float16* const sm) {   // (sm: the LDS buffer to read)
#ifdef PCLSEG_CAND_GEOM2
      // GEOM = 2 (round 4): the same compile-time geometry for the merged pairs with a 32-channel patch — fire4 (64-pixel
      // tiles, WN = 8) and fire11 (128-pixel tiles, WN = 4: wave (wm, wn) owns tile rows 4 wm .. 4 wm + 3) — whose K loop is
      // 9 K-steps of 12 MFMAs: one step = one tap (lane group g = channels 8 g ..), every fragment address one per-lane
      // base plus an immediate.  Their generic loop spends 13 vector instructions of address arithmetic per step and
      // waits after every second read.
      if constexpr (GEOM == 2) {
        static_assert(MTW == 4 && NTW == 1 && NW == 8 && PAIR, "GEOM 2: 4 x 16-pixel row segments per wave, one cout tile");
        constexpr int kPW = 18, kCS = 40, kRows = (NW / WN) * MTW, kPlane = (kRows + 2) * kPW * kCS;   // (host-checked against a.PW / CSh / plane)
        const unsigned wstep = (unsigned)K.nctp * 1024u;
        const unsigned lane8 = (unsigned)lane * 8u;
        const _Float16* xb = sm + ((wm * MTW) * kPW + p) * kCS + g * 8;   // segment m = tile row wm * 4 + m
        const bool is3 = K.nkh == 3;
        const int nst = is3 ? 9 : 1;
        const _Float16* wb = K.w16 + ((size_t)(chunk * nst) * K.nctp + ct0) * 1024;   // scalar
        f16x8 wh[2], wl[2];
        auto load_w = [&](const int st, const int slot) {
          wh[slot] = *reinterpret_cast<const f16x8*>(wb + (size_t)st * wstep + lane8);
          wl[slot] = *reinterpret_cast<const f16x8*>(wb + (size_t)st * wstep + 512 + lane8);
        };
        auto gstep = [&](const int off, const int slot) {
          f16x8 xh[4], xl[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            xh[m] = *reinterpret_cast<const f16x8*>(xb + off + m * (kPW * kCS));
            xl[m] = *reinterpret_cast<const f16x8*>(xb + off + m * (kPW * kCS) + kPlane);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot], xh[m], acc[m][0], 0, 0, 0);
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], xl[m], acc[m][0], 0, 0, 0);
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot], xh[m], acc[m][0], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        };
        load_w(0, 0);
        if (is3) {
          load_w(1, 1);
#pragma unroll
          for (int st = 0; st < 9; ++st) {
            const int ti = st / 3, tj = st - 3 * ti;
            gstep((ti * kPW + tj) * kCS, st & 1);
            if (st + 2 < 9) load_w(st + 2, st & 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          gstep((1 * kPW + 1) * kCS, 0);
        }
        return;
      }
#endif
      if constexpr (GEOM == 1) {
        static_assert((MTW == 8 || MTW == 4) && WN == 8 && NW == 8, "GEOM 1: MTW x 16-pixel tiles (one segment per tile row) of the 8-wave blocks");
        constexpr int kPW = 18, kCS = 72, kPlane = (MTW + 2) * 18 * 72;   // (host-checked against a.PW / CSh / plane)
        const unsigned wstep = (unsigned)K.nctp * 1024u;
        const unsigned lane8 = (unsigned)lane * 8u;
        const _Float16* xb = sm + p * kCS + g * 8;   // segment m = tile row m: + m * kPW * kCS; lane group g = channel group 4 (s & 1) + g
        f16x8 wh[2][NTW], wl[2][NTW];
        // one K-step: `off` = the tap's patch offset + the step's channel half, compile-time after unrolling
        auto gstep = [&](const int off, const int slot) {
#pragma unroll
          for (int m0 = 0; m0 < MTW; m0 += 4) {
            f16x8 xh[4], xl[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              xh[m] = *reinterpret_cast<const f16x8*>(xb + off + (m0 + m) * (kPW * kCS));
              xl[m] = *reinterpret_cast<const f16x8*>(xb + off + (m0 + m) * (kPW * kCS) + kPlane);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int nn = 0; nn < NTW; ++nn) {
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot][nn], xh[m], acc[m0 + m][nn], 0, 0, 0);
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][nn], xl[m], acc[m0 + m][nn], 0, 0, 0);
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][nn], xh[m], acc[m0 + m][nn], 0, 0, 0);
              }
            // (hundreds of straight-line MFMAs invite the scheduler to hoist every read of the chunk to the top —
            // 1.6 KB of scratch per lane; nothing may cross a half-step.  With four waves per SIMD a wave owns the
            // matrix pipe a quarter of the time, so its own read latency needs no software pipelining.)
            __builtin_amdgcn_sched_barrier(0);
          }
        };
#ifdef PCLSEG_CAND_KPIPE
        // (round 4) the merged pairs' 64-pixel blocks (fire8/9: one block per CU, two waves per SIMD, 186
        // registers): hipcc emits a K-step as `4 reads, wait, 12 NTW MFMAs` — every step starts with an LDS round
        // trip.  Here the 8 fragment reads of step st + 1 are issued BEFORE the MFMAs of step st (two register
        // sets, scheduling-group barriers keep the order), as in conv1x1_wide_kernel.
        // (NTW >= 2 only: fire10's kernel — one cout tile per wave — sits at 120 registers with two blocks per CU,
        // the second fragment set would cost it that: 154)
        if constexpr (MTW == 4 && NTW >= 2 && !OCC128) {
          f16x8 xh[2][4], xl[2][4];
          auto rd = [&](const int off, const int set) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              xh[set][m] = *reinterpret_cast<const f16x8*>(xb + off + m * (kPW * kCS));
              xl[set][m] = *reinterpret_cast<const f16x8*>(xb + off + m * (kPW * kCS) + kPlane);
            }
          };
          auto mm = [&](const int set, const int slot) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
              for (int nn = 0; nn < NTW; ++nn) {
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[slot][nn], xh[set][m], acc[m][nn], 0, 0, 0);
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][nn], xl[set][m], acc[m][nn], 0, 0, 0);
                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[slot][nn], xh[set][m], acc[m][nn], 0, 0, 0);
              }
          };
          const bool is3 = !PAIR || K.nkh == 3;
          const int nst = is3 ? 18 : 2;
          const _Float16* wb = K.w16 + ((size_t)(chunk * nst) * K.nctp + ct0) * 1024;   // scalar
          auto load_w = [&](const int st, const int slot) {
            const _Float16* wp = wb + (size_t)st * wstep;
#pragma unroll
            for (int nn = 0; nn < NTW; ++nn) {
              wh[slot][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + lane8);
              wl[slot][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512 + lane8);
            }
          };
          load_w(0, 0);
          load_w(1, 1);
          if (is3) {
            rd(0, 0);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
              if (st + 1 < 18) {
                const int tap = (st + 1) >> 1, ti = tap / 3, tj = tap - 3 * ti;
                rd((ti * kPW + tj) * kCS + ((st + 1) & 1) * 32, (st + 1) & 1);
              }
              mm(st & 1, st & 1);
              if (st + 1 < 18) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 12 * NTW, 0);
              __builtin_amdgcn_sched_barrier(0);
              if (st + 2 < 18) load_w(st + 2, st & 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {            // 1x1 half of a merged pair: the centre tap, 2 K-steps
            rd((1 * kPW + 1) * kCS, 0);
            rd((1 * kPW + 1) * kCS + 32, 1);
            mm(0, 0);
            mm(1, 1);
          }
          return;
        }
#endif
        if (!PAIR || K.nkh == 3) {   // 3x3: 18 K-steps, tap s / 2 (single convs are always 3x3 here: host-checked)
          constexpr int kSteps = 18;
          const _Float16* wb = K.w16 + ((size_t)(chunk * kSteps) * K.nctp + ct0) * 1024;   // scalar
          auto load_w = [&](const int st, const int slot) {
            const _Float16* wp = wb + (size_t)st * wstep;
#pragma unroll
            for (int nn = 0; nn < NTW; ++nn) {
              wh[slot][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + lane8);
              wl[slot][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512 + lane8);
            }
          };
          load_w(0, 0);
          load_w(1, 1);
#pragma unroll
          for (int st = 0; st < kSteps; ++st) {
            const int tap = st >> 1, ti = tap / 3, tj = tap - 3 * ti;
            gstep((ti * kPW + tj) * kCS + (st & 1) * 32, st & 1);
            if (st + 2 < kSteps) load_w(st + 2, st & 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {            // 1x1 half of a merged pair: the centre tap, 2 K-steps
          const _Float16* wb = K.w16 + ((size_t)(chunk * 2) * K.nctp + ct0) * 1024;
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            const _Float16* wp = wb + (size_t)st * wstep;
#pragma unroll
            for (int nn = 0; nn < NTW; ++nn) {
              wh[st][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + lane8);
              wl[st][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512 + lane8);
            }
          }
#pragma unroll
          for (int st = 0; st < 2; ++st) gstep((1 * kPW + 1) * kCS + st * 32, st);
        }
        return;
      }
        const int ntaps = K.nkh * K.nkw;
        const int steps_full = (ntaps * ck8_full + 3) >> 2;
        const int nk = ntaps * ck8;
        const int nsteps = (nk + 3) >> 2;
        const int inv_ck8 = (65536 + ck8 - 1) / ck8;
        const int inv_kw = (65536 + K.nkw - 1) / K.nkw;
        const int origin = (K.th0 * a.PW + K.tw0) * CSh;
        const _Float16* wbase16 = K.w16 + ((size_t)(chunk * steps_full) * K.nctp + ct0) * 1024 + lane * 8;
        const int wstep = K.nctp * 1024;  // halfs per K-step of packed fragments
        auto load_w = [&](int s, f16x8 (&wh)[NTW], f16x8 (&wl)[NTW]) {
          const int sc = s < nsteps ? s : nsteps - 1;   // trailing refills are unused
          const _Float16* wp = wbase16 + (unsigned)__mul24(sc, wstep);
#pragma unroll
          for (int nn = 0; nn < NTW; ++nn) {
            wh[nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024);
            wl[nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512);
          }
        };
        auto step = [&](int s, const f16x8 (&wh)[NTW], const f16x8 (&wl)[NTW]) {
          int kidx = 4 * s + g;
          if (kidx >= nk) kidx = 0;  // padded K: its weights are zero, read any valid data
          // 24-bit multiplies (full rate; a 32-bit v_mul_lo_u32 issues at a quarter of it)
          const int tap = __mul24(kidx, inv_ck8) >> 16;
          const int c8 = kidx - __mul24(tap, ck8);
          const int ti = __mul24(tap, inv_kw) >> 16;
          const int koff = origin + __mul24(__mul24(ti, a.PW) + (tap - __mul24(ti, K.nkw)), CSh) + c8 * 8;
          // (8-segment register tiles read their activation fragments in two halves: 64 registers of
          // fragments at once is what pushed those kernels past 128 registers, i.e. to one block per CU)
          constexpr int MB = (MTW == 8 && OCC128) ? 4 : MTW;
#pragma unroll
          for (int m0 = 0; m0 < MTW; m0 += MB) {
            f16x8 xh[MB], xl[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
              xh[m] = *reinterpret_cast<const f16x8*>(sm + pixoff[m0 + m] + koff);
              xl[m] = *reinterpret_cast<const f16x8*>(sm + plane + pixoff[m0 + m] + koff);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
              for (int nn = 0; nn < NTW; ++nn) {
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nn], xh[m], acc[m0 + m][nn], 0, 0, 0);
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nn], xl[m], acc[m0 + m][nn], 0, 0, 0);
                acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nn], xh[m], acc[m0 + m][nn], 0, 0, 0);
              }
            if (MB < MTW && m0 == 0) asm volatile("" ::: "memory");
          }
        };
        // Weight fragments stream from L2 through a 2-deep register ring: the loads of step s+2
        // are issued as soon as step s has consumed its slot.
        f16x8 whA[NTW], wlA[NTW], whB[NTW], wlB[NTW];
        load_w(0, whA, wlA);
        load_w(1, whB, wlB);
        for (int s0 = 0; s0 < nsteps; s0 += 2) {
          step(s0, whA, wlA);
          load_w(s0 + 2, whA, wlA);
          if (s0 + 1 < nsteps) {
            step(s0 + 1, whB, wlB);
            load_w(s0 + 3, whB, wlB);
          }
        }
      };
    if constexpr (PAIR) {
      // one chunk by construction (host): the 3x3 half, then the 1x1 half (the centre tap of the
      // same staged patch) through the same accumulators — ONE instance of the K loop and of the
      // epilogue, run twice (code size: a kernel that does not fit the instruction cache pays for it
      // on every launch)
      if constexpr (UP > 0) stage_up(); else stage(0, cin8, sm);
      __syncthreads();
      stamp(1);
      // Every block of a launch starts at the same time; if all of them ran 3x3 -> store -> 1x1 ->
      // store in the same order, the whole chip would alternate between a matrix-core phase (HBM
      // idle) and a store burst (matrix cores idle).  Half of the blocks therefore take the halves in
      // the opposite order, so one half's stores coincide with the other half's K loop.
      const int flip = a.flip_bit >= 0 ? (int)((blockIdx.x >> a.flip_bit) & 1u) : 0;
      f32x4 acc2[FSQ > 0 ? MTW : 1][FSQ > 0 ? FSQ : 1];
      if constexpr (FSQ > 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int qt = 0; qt < FSQ; ++qt) acc2[m][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma nounroll
      for (int half = 0; half < 2; ++half) {
        const int sidx = (1 - half) ^ flip;
        const ConvSub& K = a.sub[sidx];
        if (half) {
#pragma unroll
          for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int nn = 0; nn < NTW; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        kloop(K, 0, cin8, sm);
        stamp(2 + 2 * half);
        if constexpr (FSQ == 0) {
          epilogue(K, acc);
          stamp(3 + 2 * half);
        } else {
          // this half's channels -> partial squeeze sums.  K-step st of the partial GEMM covers the
          // wave's cout tiles 2st and 2st+1: lane (p, g) contributes k = (g, j): j < 4 -> channel
          // 4g+j of tile 2st, j >= 4 -> channel 4g+j-4 of tile 2st+1 (zero if NTW is odd and it is missing)
          constexpr int NS = (NTW + 1) / 2;
          f32x4 bv[NTW];
          float iv[NTW];
#pragma unroll
          for (int nn = 0; nn < NTW; ++nn) {
            bv[nn] = *reinterpret_cast<const f32x4*>(K.bias + (ct0 + nn) * 16 + g * 4);
            iv[nn] = sload(K.bias + (K.nctp + ct0 + nn) * 16);
          }
          const int cg = ct0 / NTW;   // this wave's cout group within the half
#pragma unroll
          for (int st = 0; st < NS; ++st) {
            const _Float16* ap = a.fsq_w16 + ((((size_t)sidx * a.fsq_ncg + cg) * NS + st) * FSQ) * 1024 + lane * 8;
            f16x8 ah[FSQ], al[FSQ];
#pragma unroll
            for (int qt = 0; qt < FSQ; ++qt) {
              ah[qt] = *reinterpret_cast<const f16x8*>(ap + qt * 1024);
              al[qt] = *reinterpret_cast<const f16x8*>(ap + qt * 1024 + 512);
            }
            // FIREUP pairs add a skip tensor to the pair output before the next squeeze reads it
            // (nets/SqueezeSegV2.py:313-317): fetch this step's residual quads up front
            constexpr bool kRes = EPI == 1;
            f32x4 rq[kRes ? MTW : 1][2];
            if constexpr (kRes) {
#pragma unroll
              for (int m = 0; m < MTW; ++m) {
                const int oh = h0 + seg_r[m];
                const int j = w0 + seg_q[m] * 16 + p;
                const bool valid = a.res1 && (oh < a.H) && (j < a.Wconv);
                const float* rp = a.res1 + (((size_t)n * a.H + oh) * a.Wout + j) * a.res1_C + K.co_off + g * 4;
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                  const int nn = 2 * st + t2 < NTW ? 2 * st + t2 : 2 * st;
                  const bool ok = valid && (2 * st + t2 < NTW) && ((ct0 + nn) * 16 + g * 4 < K.Cout);
                  const f32x4 t = *reinterpret_cast<const f32x4*>(ok ? rp + (ct0 + nn) * 16 : a.res1 ? a.res1 : a.in);
                  rq[m][t2] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
              }
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
              f32x4 v0 = act4(fma4(acc[m][2 * st], iv[2 * st], bv[2 * st]), K.act);
              const int n1 = 2 * st + 1 < NTW ? 2 * st + 1 : 2 * st;   // (resolved by the unroller)
              f32x4 v1 = act4(fma4(acc[m][n1], iv[n1], bv[n1]), K.act);
              if constexpr (kRes) { v0 += rq[m][0]; v1 += rq[m][1]; }
              if (2 * st + 1 >= NTW) v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
              vmax = absmax4(absmax4(vmax, v0), v1);
              f16x4 h0v, l0v, h1v, l1v;
              split4(v0, h0v, l0v);
              split4(v1, h1v, l1v);
              const f16x8 bh = (f16x8){h0v[0], h0v[1], h0v[2], h0v[3], h1v[0], h1v[1], h1v[2], h1v[3]};
              const f16x8 bl = (f16x8){l0v[0], l0v[1], l0v[2], l0v[3], l1v[0], l1v[1], l1v[2], l1v[3]};
#pragma unroll
              for (int qt = 0; qt < FSQ; ++qt) {
                acc2[m][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[qt], bh, acc2[m][qt], 0, 0, 0);
                acc2[m][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[qt], bl, acc2[m][qt], 0, 0, 0);
                acc2[m][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[qt], bh, acc2[m][qt], 0, 0, 0);
              }
            }
          }
          stamp(3 + 2 * half);
        }
      }
      stamp(6);
      if constexpr (FSQ > 0) {
        // partial sums -> LDS slab [wave][MTW*16 px][Q] (the patch is dead once every wave is here),
        // then a fixed-order sum over the WN waves of each pixel group, squeeze bias + ReLU, split, store
        // Slab rows are padded by 4 floats: with Q a multiple of 16 the 8 consecutive pixels one
        // ds_write_b128 lane group covers would otherwise all start in the same bank (rows 64..256 B
        // apart: an 8- to 16-way conflict that made this phase 29 % of the block).
#ifdef PCLSEG_CAND_SLAB
        constexpr int Q = FSQ * 16, PXW = MTW * 16;
        // (candidate, -DPCLSEG_CAND) the 136 KB slab of fire8/9 goes through LDS in TWO passes of half the squeeze
        // tiles: 70 KB per block, so that a block of another lane's memory-bound kernel can share the CU
        // (profiles/r04_ssv2_3lane_counters.txt: a 136 KB block is placed only on an EMPTY CU)
        constexpr int NPASS = (FSQ == 4 && NW * PXW * (Q + 4) * 4 > 96 * 1024) ? 2 : 1;
        constexpr int FQ = FSQ / NPASS, QP = FQ * 16, QS = QP + 4;
        float* slab = reinterpret_cast<float*>(smem_raw);
        constexpr int WMc = NW / WN, QQ = QP / 4;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          __syncthreads();
#pragma unroll
          for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int qt = 0; qt < FQ; ++qt)
              *reinterpret_cast<f32x4*>(slab + ((size_t)(wave * PXW + m * 16 + p) * QS + qt * 16 + 4 * g)) = acc2[m][pass * FQ + qt];
          __syncthreads();
          for (int idx = tid; idx < WMc * PXW * QQ; idx += kThreads) {
            const int px = idx / QQ, qq = idx - px * QQ;
            const int wmi = px / PXW, pl = px - wmi * PXW;
            const int qg = pass * QP + qq * 4;   // squeeze channel of this quad
            f32x4 sum = *reinterpret_cast<const f32x4*>(slab + ((size_t)((wmi * WN) * PXW + pl) * QS + qq * 4));
#pragma unroll
            for (int w = 1; w < WN; ++w)
              sum += *reinterpret_cast<const f32x4*>(slab + ((size_t)((wmi * WN + w) * PXW + pl) * QS + qq * 4));
            sum = fma4(sum, a.fsq_bias[Q + (qg & ~15)], *reinterpret_cast<const f32x4*>(a.fsq_bias + qg));
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[e] = fmaxf(sum[e], 0.0f);
            const int seg = wmi * MTW + (pl >> 4), pp = pl & 15;
            const int sr = seg / a.SEGW;
            const int oh = h0 + sr, j = w0 + (seg - sr * a.SEGW) * 16 + pp;
            if (oh < a.H && j < a.Wconv) {
              f16x4 hi, lo;
              split4(sum, hi, lo);
              vmax = absmax4(vmax, sum);
              _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + (((size_t)n * a.H + oh) * a.Wout + j) * (size_t)(2 * Q) + qg;
              *reinterpret_cast<f16x4*>(o16) = hi;
              *reinterpret_cast<f16x4*>(o16 + Q) = lo;
            }
          }
        }
      }
#else
        constexpr int Q = FSQ * 16, PXW = MTW * 16, QS = Q + 4;
        float* slab = reinterpret_cast<float*>(smem_raw);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
          for (int qt = 0; qt < FSQ; ++qt)
            *reinterpret_cast<f32x4*>(slab + ((size_t)(wave * PXW + m * 16 + p) * QS + qt * 16 + 4 * g)) = acc2[m][qt];
        __syncthreads();
        constexpr int WMc = NW / WN, QQ = Q / 4;
        for (int idx = tid; idx < WMc * PXW * QQ; idx += kThreads) {
          const int px = idx / QQ, qq = idx - px * QQ;
          const int wmi = px / PXW, pl = px - wmi * PXW;
          f32x4 sum = *reinterpret_cast<const f32x4*>(slab + ((size_t)((wmi * WN) * PXW + pl) * QS + qq * 4));
#pragma unroll
          for (int w = 1; w < WN; ++w)
            sum += *reinterpret_cast<const f32x4*>(slab + ((size_t)((wmi * WN + w) * PXW + pl) * QS + qq * 4));
          sum = fma4(sum, a.fsq_bias[Q + ((qq * 4) & ~15)], *reinterpret_cast<const f32x4*>(a.fsq_bias + qq * 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) sum[e] = fmaxf(sum[e], 0.0f);
          const int seg = wmi * MTW + (pl >> 4), pp = pl & 15;
          const int sr = seg / a.SEGW;
          const int oh = h0 + sr, j = w0 + (seg - sr * a.SEGW) * 16 + pp;
          if (oh < a.H && j < a.Wconv) {
            f16x4 hi, lo;
            split4(sum, hi, lo);
            vmax = absmax4(vmax, sum);
            _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + (((size_t)n * a.H + oh) * a.Wout + j) * (size_t)(2 * Q) + qq * 4;
            *reinterpret_cast<f16x4*>(o16) = hi;
            *reinterpret_cast<f16x4*>(o16 + Q) = lo;
          }
        }
      }
#endif
    } else if constexpr (LW > 0) {
      // two LDS buffers; loaders fill buffer (c + 1) & 1 while the compute waves read buffer c & 1.  Barrier
      // #c (c = 0 .. nchunks - 1) is passed by the loaders after staging chunk c and by the compute waves
      // before its K loop — i.e. after the K loop of chunk c - 1, which frees buffer (c + 1) & 1 for chunk c + 1.
      const int nchunks = (cin8 + ck8_full - 1) / ck8_full;
      const int buf_halfs = (2 * plane + 7) & ~7;
      if (wave >= NW) {
        // A loader shares its SIMD with two compute waves that keep the matrix pipe and half of the vector issue
        // slots busy, and as the YOUNGEST wave it loses every arbitration: without priority a chunk's staging
        // took 38 k cycles beside a K loop (11 k alone) and the compute waves waited for it.
        __builtin_amdgcn_s_setprio(3);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
          const int c8_0 = chunk * ck8_full;
          const int ck8 = (cin8 - c8_0) < ck8_full ? (cin8 - c8_0) : ck8_full;
          stage(c8_0, ck8, sm + (chunk & 1) * buf_halfs);
#ifdef PCLSEG_WITH_STAMPS
          if (a.stamps && tid == kThreads && (chunk == 1 || chunk == 2)) a.stamps[(size_t)blockIdx.x * 8 + 4 + chunk] = __builtin_amdgcn_s_memtime();
#endif
          __syncthreads();
        }
        if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
        return;
      }
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c8_0 = chunk * ck8_full;
        const int ck8 = (cin8 - c8_0) < ck8_full ? (cin8 - c8_0) : ck8_full;
        lds_barrier();
        if (chunk == 0) stamp(1);
        if (chunk == 1) stamp(3);
        kloop(S, chunk, ck8, sm + (chunk & 1) * buf_halfs);
        if (chunk == 0) stamp(2);
        if (chunk == 1) stamp(4);
      }
    } else {
      int chunk = 0;
      for (int c8_0 = 0; c8_0 < cin8; c8_0 += ck8_full, ++chunk) {
        const int ck8 = (cin8 - c8_0) < ck8_full ? (cin8 - c8_0) : ck8_full;
        if (chunk) __syncthreads();
        stage(c8_0, ck8, sm);
        __syncthreads();
        stamp(chunk < 2 ? 1 + 2 * chunk : 5);
        kloop(S, chunk, ck8, sm);
        stamp(chunk < 2 ? 2 + 2 * chunk : 5);
      }
      if (chunk < 2) { stamp(3); stamp(4); }
    }
    if constexpr (!PAIR) {
      if constexpr (LW == 0) stamp(5);
      if constexpr (OCC128 && MTW == 8) {
        epilogue(S, acc, 0, 4);
        asm volatile("" ::: "memory");
        epilogue(S, acc, 4, 8);
      } else {
        epilogue(S, acc);
      }
      if constexpr (LW == 0) stamp(6);
    }
    stamp(7);
    if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
  }
}

// ---- fire13 -> conv14 -> head in one kernel (SqueezeSegV2's tail, nets/SqueezeSegV2.py:276-282,318-325)
//   fire13/upconv (Conv2DTranspose (1,4)/(1,2) + ReLU)  ->  expand1x1 || expand3x3 (+BN+ReLU)
//   -> + bn1_skip(conv1_skip(input))  ->  [dropout: identity]  ->  conv14 3x3 -> [softmax] -> argmax -> mask
// fire13's 64-channel full-resolution output (33.5 MB per scan written, 35.5 MB re-read by the head: 16 %
// of all HBM-side traffic of the network and two of its three longest launches) never leaves the CU: a
// 4-wave block owns an 8 x 16 pixel tile of PREDICTIONS and keeps, in LDS,
//   U  the up-convolved squeeze (16 ch) on the tile + 2-pixel halo, 12 x 20 px, split-f16 [hi 16|lo 16|pad]
//   F  fire13's output (64 ch, after the skip add) on the tile + 1-pixel halo, 10 x 18 px = 12 segments of
//      16 pixels (flattened), split-f16 [hi 64|lo 64|pad]; pixels outside the image are ZERO (conv14's
//      padding), not fire13 evaluated on padding
// 72 KB per block: two blocks per CU, so one block's staging / epilogues overlap the other's K loops.
//   phase 0  half-width squeeze patch (12 x 12 source pixels) -> LDS (aliases F)
//   phase 1  up-convolution: wave <-> (output parity, half of the 16-pixel units), K = 2 taps x 16
//   phase 2  expand pair on the 12 F segments (3 per wave, weight fragments fetched once per wave):
//            3x3 half 5 K-steps + 1x1 half 1 K-step, bias/ReLU, skip branch (one more K-step) from the 8-channel network
//            input (32 B per pixel from L2), zero outside the image, split -> F
//   phase 3  conv14 on the 8 tile rows (2 per wave): 18 K-steps over (tap, 8-channel group) pairs of F,
//            packed head fragments streamed through a 2-deep ring, then head_finish
// The halo costs 1.41x fire13's matrix work (180 px computed for 128) - fire13 is 1/4 of the block's MFMAs.
// Only 9 MB per scan cross the memory system (squeeze tensor, raw input, mask, predictions).
struct FireHeadArgs {
  const _Float16* sq;      // fire13/squeeze [N,H,W/2][hi 16 | lo 16]
  const float* x8;         // network input [N,H,W,8] (skip branch)
  const uint8_t* mask;     // [N,H,W]
  int32_t* preds;
  float* probs;            // optional
  float* logits;           // optional
  int N, H, W, tilesH, tilesW, NC, none_index;
  const _Float16* up_w16[2];   // transposed conv, one 1-step fragment set per output parity
  const float* up_bias;        // [parity][bias 16 | inv 16]
  const _Float16* e1_w16;      // expand1x1 (1 K-step used), e3: expand3x3 (5 K-steps); 2 cout tiles each
  const _Float16* e3_w16;
  const float* e1_bias;        // [bias 32 | inv 32]
  const float* e3_bias;
  const _Float16* sk_w16;      // skip branch conv1_skip + bn1_skip: one K-step (8 input channels = K-group 0) x 4 cout tiles
  const float* sk_bias;        // [bias 64 | inv 64]
  const _Float16* hd_w16;      // conv14: 18 K-steps x NCT tiles (packed with a 64-channel chunk)
  const float* hd_bias;        // [bias NCT*16 | inv NCT*16]
  unsigned* range_flag;
#ifdef PCLSEG_WITH_STAMPS
  unsigned long long* stamps;
#endif
};

constexpr int kFhTH = 8, kFhTW = 16;                 // prediction tile
constexpr int kFhFW = kFhTW + 2, kFhFH = kFhTH + 2;   // F region 10 x 18
constexpr int kFhUW = kFhTW + 4, kFhUH = kFhTH + 4;   // U region 12 x 20
constexpr int kFhSW = kFhUW / 2 + 2;                  // source columns 12
constexpr int kFhCSU = 2 * 16 + kPadF16;              // halfs per U / source pixel
constexpr int kFhCSF = 2 * 64 + kPadF16;              // halfs per F pixel
constexpr int kFhFSeg = (kFhFH * kFhFW + 15) / 16;    // 12 segments
constexpr int kFhLdsF = kFhFSeg * 16 * kFhCSF * 2;    // bytes
constexpr int kFhLdsU = kFhUH * kFhUW * kFhCSU * 2;
constexpr int kFhLds = kFhLdsF + kFhLdsU;
static_assert(kFhUH * kFhSW * kFhCSU * 2 <= kFhLdsF, "the source patch aliases F");
static_assert(kFhFSeg == 12, "3 F segments per wave");

// What was measured on the way (s_memtime stamps per phase, one lane, 4 scans):
//   v1  weights streamed per wave through a 2-deep ring, operands fetched where used, __syncthreads():
//       136 us = the two separate launches; 32 k cycles per block for 5.4 k cycles of matrix work
//   v2  this kernel: every operand requested at entry, barriers that do not drain vmcnt, expand fragments
//       resident, conv14's K-steps dealt to the waves: 121 us
//   v3  persistent blocks (2 per CU) with the next tile's patch requested a phase ahead: 161 us, and 3-lane
//       throughput 5700 instead of 6390 scans/s — resident blocks hold 144 KB of LDS per CU for the whole
//       launch and shut the other lanes' kernels out; the prefetched operands spilled to scratch
//   v4  8 waves per block at 128 registers (4 waves per SIMD; phase 2 split by segment group x cout half, phase 3
//       by K group x cout tile), built twice — with the VALU skip branch (206 us) and with the skip branch on
//       the matrix cores and scheduling barriers around the staggered loads (228 us): the allocation never got
//       below 128 registers without 200 bytes of scratch per lane, and every use of scratch in this kernel
//       has cost a factor, not a percentage
//   v5  v2 + the skip branch as one more K-step on the matrix cores (-400 vector instructions per wave): 119 us
// The block is bound by instruction issue, not by any pipe: ~2000 vector + ~1100 scalar + ~250 LDS
// instructions per wave and tile beside 351 MFMAs, at two waves per SIMD.
template <int NCT>
__global__ __launch_bounds__(256, 2) void fire_head_kernel(const FireHeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* F = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* S = F;                                                       // source patch (dead before F is written)
  _Float16* U = reinterpret_cast<_Float16*>(smem_raw + kFhLdsF);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const unsigned lane8 = (unsigned)lane * 8u;   // fragment addresses: base in SGPRs + this shared lane offset
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % a.tilesW;
  tile /= a.tilesW;
  const int thi = tile % a.tilesH;
  const int n = tile / a.tilesH;
  const int h0 = thi * kFhTH, w0 = twi * kFhTW;
  const int W2 = a.W >> 1;
  float vmax = 0.f;
#ifdef PCLSEG_WITH_STAMPS
  auto stamp = [&](int i) { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + i] = __builtin_amdgcn_s_memtime(); };
#else
  auto stamp = [](int) {};
#endif
  stamp(0);

  // ------------------------------------------------ entry: every long-latency operand is requested NOW
  // Two blocks share a CU and each is a chain of short phases; a phase that began with a memory round
  // trip would expose it.
  const int parity = wave & 1;
  // (1) the half-width squeeze patch, 12 x 12 source pixels x 4 sixteen-byte units = 576 units
  constexpr int kSrcUnits = kFhUH * kFhSW * 4, kSrcPer = (kSrcUnits + 255) / 256;
  f16x8 sv[kSrcPer];
  {
    const _Float16* sqn = a.sq + (size_t)n * a.H * W2 * 32;
    const int j0 = (w0 >> 1) - 2;
#pragma unroll
    for (int k = 0; k < kSrcPer; ++k) {
      const int i = tid + k * 256;
      const int px = (i < kSrcUnits ? i : 0) >> 2, un = i & 3;
      const int r = px / kFhSW, c = px - r * kFhSW;
      const int h = h0 - 2 + r, col = j0 + c;
      const bool ok = i < kSrcUnits && h >= 0 && h < a.H && col >= 0 && col < W2;
      sv[k] = *reinterpret_cast<const f16x8*>(ok ? sqn + ((size_t)h * W2 + col) * 32 + un * 8 : sqn);
      if (!ok) sv[k] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // (2) this wave's up-conv fragments (one K-step, one cout tile) of its output parity
  const f16x8 uwh = *reinterpret_cast<const f16x8*>(a.up_w16[parity] + lane8);
  const f16x8 uwl = *reinterpret_cast<const f16x8*>(a.up_w16[parity] + 512 + lane8);
  const f32x4 ub = *reinterpret_cast<const f32x4*>(a.up_bias + parity * 32 + g * 4);
  const float ui = sload(a.up_bias + parity * 32 + 16);
  // (3) the skip branch's raw-input addresses of this wave's 3 F segments
  int uoff[3], foff[3];
  bool fimg[3];
  const float* sxp[3];
  {
    const float* x8n = a.x8 + (size_t)n * a.H * a.W * 8;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int l = (wave * 3 + m) * 16 + p;
      const int lc = l < kFhFH * kFhFW ? l : kFhFH * kFhFW - 1;   // (lanes past the region repeat its last pixel)
      foff[m] = lc * kFhCSF;
      const int r = lc / kFhFW, c = lc - r * kFhFW;
      uoff[m] = (r * kFhUW + c) * kFhCSU;
      const int h = h0 - 1 + r, w = w0 - 1 + c;
      fimg[m] = h >= 0 && h < a.H && w >= 0 && w < a.W;
      sxp[m] = fimg[m] ? x8n + ((size_t)h * a.W + w) * 8 : x8n;
    }
  }
  // (4) the expand3x3 fragments: 5 K-steps x 2 cout tiles, hi + lo = 80 registers (the 1x1 half's 16, the
  //     skip input's 24 and the head bias follow as the 3x3 steps retire their fragments: a register
  //     budget of 256 per wave does not hold everything at once)
  f16x8 e3h[5][2], e3l[5][2], e1h[2], e1l[2];
#pragma unroll
  for (int st = 0; st < 5; ++st)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      e3h[st][t] = *reinterpret_cast<const f16x8*>(a.e3_w16 + (st * 2 + t) * 1024 + lane8);
      e3l[st][t] = *reinterpret_cast<const f16x8*>(a.e3_w16 + (st * 2 + t) * 1024 + 512 + lane8);
    }
  f32x4 sx0[3], sx1[3];
  f32x4 hbv[NCT];
  float hiv[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t) hiv[t] = sload(a.hd_bias + (NCT + t) * 16);
  // (5) this lane's mask bytes (output rows 2 wave, 2 wave + 1, column p)
  bool ovalid[2];
  size_t opix[2];
  uint8_t omask[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int oh = h0 + wave * 2 + m, ow = w0 + p;
    ovalid[m] = oh < a.H && ow < a.W;
    opix[m] = ((size_t)n * a.H + oh) * a.W + ow;
    omask[m] = a.mask[ovalid[m] ? opix[m] : (size_t)0];   // (clamped address instead of a branch; unused when !ovalid)
  }

  // ---------------------------------------------------------------- phase 0: patch and skip weights -> LDS
#pragma unroll
  for (int k = 0; k < kSrcPer; ++k) {
    const int i = tid + k * 256;
    if (i < kSrcUnits) *reinterpret_cast<f16x8*>(S + (i >> 2) * kFhCSU + (i & 3) * 8) = sv[k];
  }
  lds_barrier();
  stamp(1);

  // ---------------------------------------------------------------- phase 1: up-convolution -> U
  // output column w = 2j + parity reads x[j - 1 + parity] (tap 0) and x[j + parity] (tap 1); U column pc
  // is image column w0 - 2 + pc (w0 - 2 is even, so pc has the parity of w)
  {
#ifdef PCLSEG_CAND_TAIL
    constexpr int PER = kFhUH * (kFhUW / 2), UNITS = (PER + 15) / 16, NU = (UNITS + 1) / 2, NB = 2;
    static_assert(NU % NB == 0, "units per wave in batches of NB");
    // (the wave's NU units in batches of NB: the batch's fragment reads, then its NB independent 3-MFMA chains,
    // then the epilogues — as a rolled loop every unit was a serial chain read -> 3 dependent MFMAs -> split ->
    // write; all NU at once needs 48 registers this phase does not have: 32 B of scratch)
#pragma nounroll
    for (int k0 = 0; k0 < NU; k0 += NB) {
      f16x8 xh[NB], xl[NB];
      int dst[NB];
      bool pvv[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int u = 2 * (k0 + k) + (wave >> 1);
        const int l = u * 16 + p;
        const int lc = l < PER ? l : PER - 1;
        const int pr = lc / (kFhUW / 2), k2 = lc - pr * (kFhUW / 2);
        const int pc = 2 * k2 + parity;
        const int h = h0 - 2 + pr, w = w0 - 2 + pc;
        pvv[k] = h >= 0 && h < a.H && w >= 0 && w < a.W;
        const int tap = g >> 1, c8 = g & 1;
        const _Float16* sp = S + (pr * kFhSW + k2 + parity + tap) * kFhCSU + c8 * 8;
        xh[k] = *reinterpret_cast<const f16x8*>(sp);
        xl[k] = *reinterpret_cast<const f16x8*>(sp + 16);
        dst[k] = (pr * kFhUW + pc) * kFhCSU + g * 4;
      }
      f32x4 au[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) au[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwl, xh[k], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NB; ++k) au[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwh, xl[k], au[k], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NB; ++k) au[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwh, xh[k], au[k], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NB; ++k) {   // (lanes past the last pixel were clamped onto it: they compute and store ITS value again — no branch)
        f32x4 v = fma4(au[k], ui, ub);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pvv[k] ? fmaxf(v[e], 0.0f) : 0.0f;
        vmax = absmax4(vmax, v);
        f16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<f16x4*>(U + dst[k]) = hi;
        *reinterpret_cast<f16x4*>(U + dst[k] + 16) = lo;
      }
    }
  }
#else
    constexpr int PER = kFhUH * (kFhUW / 2), UNITS = (PER + 15) / 16;
#pragma nounroll
    for (int u0 = 0; u0 < UNITS; u0 += 2) {
      const int u = u0 + (wave >> 1);
      const int l = u * 16 + p;
      const int lc = l < PER ? l : PER - 1;
      const int pr = lc / (kFhUW / 2), k2 = lc - pr * (kFhUW / 2);
      const int pc = 2 * k2 + parity;
      const int h = h0 - 2 + pr, w = w0 - 2 + pc;
      const bool pv = h >= 0 && h < a.H && w >= 0 && w < a.W;
      const int tap = g >> 1, c8 = g & 1;
      const _Float16* sp = S + (pr * kFhSW + k2 + parity + tap) * kFhCSU + c8 * 8;
      const f16x8 xh = *reinterpret_cast<const f16x8*>(sp);
      const f16x8 xl = *reinterpret_cast<const f16x8*>(sp + 16);
      f32x4 au = (f32x4){0.f, 0.f, 0.f, 0.f};
      au = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwl, xh, au, 0, 0, 0);
      au = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwh, xl, au, 0, 0, 0);
      au = __builtin_amdgcn_mfma_f32_16x16x32_f16(uwh, xh, au, 0, 0, 0);
      {   // (lanes past the last pixel were clamped onto it: they compute and store ITS value again — no branch)
        f32x4 v = fma4(au, ui, ub);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pv ? fmaxf(v[e], 0.0f) : 0.0f;
        vmax = absmax4(vmax, v);
        f16x4 hi, lo;
        split4(v, hi, lo);
        _Float16* d = U + (pr * kFhUW + pc) * kFhCSU + g * 4;
        *reinterpret_cast<f16x4*>(d) = hi;
        *reinterpret_cast<f16x4*>(d + 16) = lo;
      }
    }
  }
#endif
  lds_barrier();   // U complete (and every wave is done reading S: F may be overwritten)
  stamp(2);

  // ---------------------------------------------------------------- phase 2: expand pair + skip -> F
  // head fragments of THIS wave's K-steps {wave, wave + 4, ...} (phase 3) are requested as the expand
  // fragments' registers come free, in flight during the F epilogue
  constexpr int kHdSteps = 5;   // ceil(18 / 4); step 16 + wave exists for waves 0 and 1 only
  f16x8 hwh[kHdSteps][NCT], hwl[kHdSteps][NCT];
  auto load_hw = [&](const int i) {
    const int st = wave + 4 * i < 18 ? wave + 4 * i : wave;   // (waves 2, 3: the fifth slot is unused)
    const _Float16* wp = a.hd_w16 + st * (NCT * 1024);         // (scalar)
#pragma unroll
    for (int tt = 0; tt < NCT; ++tt) {
      hwh[i][tt] = *reinterpret_cast<const f16x8*>(wp + tt * 1024 + lane8);
      hwl[i][tt] = *reinterpret_cast<const f16x8*>(wp + tt * 1024 + 512 + lane8);
    }
  };
  {
    f32x4 acc[3][4];   // tiles 0,1: expand1x1 (channels 0..31), tiles 2,3: expand3x3 (32..63)
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // expand3x3: K pairs (tap, 8-channel group) = 18 -> 5 steps; lane group g of step s owns pair 4s + g
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      int kidx = 4 * st + g;
      if (kidx >= 18) kidx = 0;   // padded K: zero weights, any valid address
      const int tap = kidx >> 1, c8 = kidx & 1;
      const int ti = (tap * 11) >> 5;   // tap / 3 for tap < 9
      const int koff = (ti * kFhUW + (tap - 3 * ti)) * kFhCSU + c8 * 8;
      f16x8 xh[3], xl[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        xh[m] = *reinterpret_cast<const f16x8*>(U + uoff[m] + koff);
        xl[m] = *reinterpret_cast<const f16x8*>(U + uoff[m] + koff + 16);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[m][2 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e3l[st][t], xh[m], acc[m][2 + t], 0, 0, 0);
          acc[m][2 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e3h[st][t], xl[m], acc[m][2 + t], 0, 0, 0);
          acc[m][2 + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e3h[st][t], xh[m], acc[m][2 + t], 0, 0, 0);
        }
      asm volatile("" ::: "memory");   // (keeps the compiler from hoisting the loads below back to the kernel entry)
      if (st == 0) {          // the registers of step 0's fragments are free: the 1x1 half's fragments
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          e1h[t] = *reinterpret_cast<const f16x8*>(a.e1_w16 + t * 1024 + lane8);
          e1l[t] = *reinterpret_cast<const f16x8*>(a.e1_w16 + t * 1024 + 512 + lane8);
        }
      } else if (st == 1) {   // the skip branch's raw input of segments 0 and 1
        sx0[0] = *reinterpret_cast<const f32x4*>(sxp[0]); sx1[0] = *reinterpret_cast<const f32x4*>(sxp[0] + 4);
        sx0[1] = *reinterpret_cast<const f32x4*>(sxp[1]); sx1[1] = *reinterpret_cast<const f32x4*>(sxp[1] + 4);
      } else if (st == 2) {   // ... of segment 2, and the head's bias
        sx0[2] = *reinterpret_cast<const f32x4*>(sxp[2]); sx1[2] = *reinterpret_cast<const f32x4*>(sxp[2] + 4);
#pragma unroll
        for (int t = 0; t < NCT; ++t) hbv[t] = *reinterpret_cast<const f32x4*>(a.hd_bias + t * 16 + g * 4);
      }
    }
    stamp(3);
    // expand1x1: the centre tap, 2 channel groups -> 1 step (lane groups 2, 3 carry zero weights)
    {
      const int koff = (1 * kFhUW + 1) * kFhCSU + (g & 1) * 8;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const f16x8 xh = *reinterpret_cast<const f16x8*>(U + uoff[m] + koff);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(U + uoff[m] + koff + 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e1l[t], xh, acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e1h[t], xl, acc[m][t], 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(e1h[t], xh, acc[m][t], 0, 0, 0);
        }
      }
    }
    stamp(4);
    asm volatile("" ::: "memory");
#ifdef PCLSEG_CAND_TAIL
    // (round 4) the epilogue's own operands — the skip branch's fragments and the two bias quads of each cout tile —
    // are requested HERE, together, into the registers the expand fragments have just left; they were fetched inside
    // the per-tile loop right in front of their use: four L2 round trips in a row per block
    f16x8 kwhA[4], kwlA[4];
    f32x4 bvA[4], kbA[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kwhA[t] = *reinterpret_cast<const f16x8*>(a.sk_w16 + t * 1024 + lane8);
      kwlA[t] = *reinterpret_cast<const f16x8*>(a.sk_w16 + t * 1024 + 512 + lane8);
      bvA[t] = *reinterpret_cast<const f32x4*>((t < 2 ? a.e1_bias : a.e3_bias) + (t & 1) * 16 + g * 4);
      kbA[t] = *reinterpret_cast<const f32x4*>(a.sk_bias + t * 16 + g * 4);
    }
    load_hw(0);                           // (the other steps' follow the epilogue: register budget)
    f16x8 sxh[3], sxl[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = g == 0 ? sx0[m][e] : 0.0f, x1 = g == 0 ? sx1[m][e] : 0.0f;
        const _Float16 h0v = (_Float16)x0, h1v = (_Float16)x1;
        sxh[m][e] = h0v; sxh[m][4 + e] = h1v;
        sxl[m][e] = (_Float16)(x0 - (float)h0v); sxl[m][4 + e] = (_Float16)(x1 - (float)h1v);
      }
      vmax = absmax4(absmax4(vmax, sx0[m]), sx1[m]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* bb = t < 2 ? a.e1_bias : a.e3_bias;
      const int tt = t & 1, co = t * 16 + g * 4;
      const float iv = sload(bb + 32 + tt * 16);
      const float ki = sload(a.sk_bias + 64 + t * 16);
      f32x4 z[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        z[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwlA[t], sxh[m], z[m], 0, 0, 0);
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwhA[t], sxl[m], z[m], 0, 0, 0);
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwhA[t], sxh[m], z[m], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        f32x4 v = fma4(acc[m][t], iv, bvA[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        v += fma4(z[m], ki, kbA[t]);
        if (!fimg[m]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        vmax = absmax4(vmax, v);
        f16x4 hi, lo;
        split4(v, hi, lo);
        _Float16* d = F + foff[m] + co;
        *reinterpret_cast<f16x4*>(d) = hi;
        *reinterpret_cast<f16x4*>(d + 64) = lo;
      }
    }
    asm volatile("" ::: "memory");
    load_hw(1); load_hw(2); load_hw(3); load_hw(4);
  }
#else
    load_hw(0); load_hw(1); load_hw(2);   // (the last two steps' follow the epilogue: register budget)
    // bias + ReLU, + skip branch (nets/SqueezeSegV2.py:293,319), zero outside the image, split -> F.
    // The skip branch (1x1 conv of the 8-channel input) runs on the matrix cores too: K = 8 is one quarter
    // of a 32-deep step (lane group 0), three MFMAs per (segment, cout tile) instead of 32 FMAs + 9 LDS reads —
    // this kernel is bound by vector-instruction issue, the matrix pipe has room.
    f16x8 sxh[3], sxl[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x0 = g == 0 ? sx0[m][e] : 0.0f, x1 = g == 0 ? sx1[m][e] : 0.0f;
        const _Float16 h0v = (_Float16)x0, h1v = (_Float16)x1;
        sxh[m][e] = h0v; sxh[m][4 + e] = h1v;
        sxl[m][e] = (_Float16)(x0 - (float)h0v); sxl[m][4 + e] = (_Float16)(x1 - (float)h1v);
      }
      vmax = absmax4(absmax4(vmax, sx0[m]), sx1[m]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* bb = t < 2 ? a.e1_bias : a.e3_bias;
      const int tt = t & 1, co = t * 16 + g * 4;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bb + tt * 16 + g * 4);
      const float iv = sload(bb + 32 + tt * 16);
      const f16x8 kwh = *reinterpret_cast<const f16x8*>(a.sk_w16 + t * 1024 + lane8);
      const f16x8 kwl = *reinterpret_cast<const f16x8*>(a.sk_w16 + t * 1024 + 512 + lane8);
      const f32x4 kb = *reinterpret_cast<const f32x4*>(a.sk_bias + t * 16 + g * 4);
      const float ki = sload(a.sk_bias + 64 + t * 16);
      f32x4 z[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        z[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwl, sxh[m], z[m], 0, 0, 0);
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwh, sxl[m], z[m], 0, 0, 0);
        z[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kwh, sxh[m], z[m], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        f32x4 v = fma4(acc[m][t], iv, bv);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        v += fma4(z[m], ki, kb);
        if (!fimg[m]) v = (f32x4){0.f, 0.f, 0.f, 0.f};
        vmax = absmax4(vmax, v);
        f16x4 hi, lo;
        split4(v, hi, lo);
        _Float16* d = F + foff[m] + co;
        *reinterpret_cast<f16x4*>(d) = hi;
        *reinterpret_cast<f16x4*>(d + 64) = lo;
      }
    }
    asm volatile("" ::: "memory");
    load_hw(3); load_hw(4);
  }
#endif
  lds_barrier();
  stamp(5);

  // ---------------------------------------------------------------- phase 3: conv14, K split over the waves
  // The 18 K-steps (pair 4 st + g = tap st / 2, channel group 4 (st & 1) + g) are dealt to the 4 waves;
  // a wave keeps ITS steps' fragments in registers and sweeps all 8 tile rows with them, so conv14's 72 KB
  // of fragments enter the block once (a wave that owned rows would stream all of them per 2 rows: 288 KB
  // per block through a 64 B/clk vector L1, more than the matrix work takes).  The four partial sums of a
  // row meet in LDS and are added in a fixed order (deterministic).
  {
    f32x4 acc[8][NCT];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int t = 0; t < NCT; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fbase = p * kFhCSF + g * 8;
#ifdef PCLSEG_CAND_TAIL
    // Software-pipelined over the (K-step, row) groups: the two fragment reads of group k + 1 are issued BEFORE the
    // six MFMAs of group k and land while those run (hipcc, left alone, emitted `2 reads, wait, 6 MFMAs` per
    // group: ~100 cycles of LDS latency in front of every 96 cycles of matrix work, 36 times per wave).  A
    // scheduling barrier per group keeps that order; steps 0-3 exist for every wave (32 groups), step 4 for waves
    // 0 and 1 only (wave-uniform branch, its own prologue).
    auto frag_off = [&](const int st, const int r) {
      const int tap = st >> 1, ti = (tap * 11) >> 5, tj = tap - 3 * ti;
      return fbase + (ti * kFhFW + tj) * kFhCSF + (st & 1) * 32 + r * (kFhFW * kFhCSF);
    };
    auto sweep = [&](const int i0, const int i1) {
      f16x8 xh = *reinterpret_cast<const f16x8*>(F + frag_off(wave + 4 * i0, 0));
      f16x8 xl = *reinterpret_cast<const f16x8*>(F + frag_off(wave + 4 * i0, 0) + 64);
#pragma unroll
      for (int i = i0; i < i1; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          f16x8 nh = xh, nl = xl;
          const bool more = r < 7 || i + 1 < i1;
          if (more) {
            const int off = r < 7 ? frag_off(wave + 4 * i, r + 1) : frag_off(wave + 4 * (i + 1), 0);
            nh = *reinterpret_cast<const f16x8*>(F + off);
            nl = *reinterpret_cast<const f16x8*>(F + off + 64);
          }
#pragma unroll
          for (int t = 0; t < NCT; ++t) {
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwl[i][t], xh, acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh[i][t], xl, acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh[i][t], xh, acc[r][t], 0, 0, 0);
          }
          // (order inside the group: the two DS reads FIRST, then the MFMAs — left to itself the scheduler sinks
          // the reads to two MFMAs before their use to save registers)
          if (more) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * NCT, 0);
          __builtin_amdgcn_sched_barrier(0);
          xh = nh; xl = nl;
        }
      }
    };
    sweep(0, 4);
    if (wave + 16 < 18) sweep(4, 5);   // wave-uniform
#else
#pragma unroll
    for (int i = 0; i < kHdSteps; ++i) {
      const int st = wave + 4 * i;
      if (st < 18) {   // wave-uniform
        const int tap = st >> 1, ti = (tap * 11) >> 5, tj = tap - 3 * ti;
        const int koff = fbase + (ti * kFhFW + tj) * kFhCSF + (st & 1) * 32;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const f16x8 xh = *reinterpret_cast<const f16x8*>(F + koff + r * (kFhFW * kFhCSF));
          const f16x8 xl = *reinterpret_cast<const f16x8*>(F + koff + r * (kFhFW * kFhCSF) + 64);
#pragma unroll
          for (int t = 0; t < NCT; ++t) {
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwl[i][t], xh, acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh[i][t], xl, acc[r][t], 0, 0, 0);
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hwh[i][t], xh, acc[r][t], 0, 0, 0);
          }
        }
      }
    }
#endif
    stamp(6);
    // partial sums -> LDS [wave][row][tile][lane] (F and U are dead once every wave is here)
    f32x4* part = reinterpret_cast<f32x4*>(smem_raw);
    static_assert(4 * 8 * 2 * 64 * 16 <= kFhLdsF + kFhLdsU, "the partial sums alias F and U");
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int t = 0; t < NCT; ++t) part[((wave * 8 + r) * NCT + t) * 64 + lane] = acc[r][t];
    lds_barrier();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x4 lv[NCT];
#pragma unroll
      for (int t = 0; t < NCT; ++t) {
        f32x4 sum = part[((0 * 8 + wave * 2 + m) * NCT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += part[((w * 8 + wave * 2 + m) * NCT + t) * 64 + lane];
        lv[t] = fma4(sum, hiv[t], hbv[t]);
      }
      head_finish<NCT>(lv, ovalid[m], opix[m], g, a.NC, omask[m], a.preds, a.probs, a.logits, a.none_index);
    }
  }
  stamp(7);
  if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
}

#ifdef PCLSEG_CAND_WIDE
// ---- Wide 1x1 convolutions (Darknet's BasicBlock / decoder-block conv1: 128-1024 -> 64-1024 channels), split-f16.
// Reference: nets/Darknet.py:34-43 (conv1 1x1 + bn1 + LeakyReLU of BasicBlock).
// conv_kernel runs them as `stage a 64-channel chunk -> barrier -> 2 K-steps -> barrier`: a chunk's matrix work is
// 96 MFMAs per wave (1.5 k cycles) between two memory round trips that nothing overlaps (3x3 layers amortise the
// same per-chunk cost over 18 K-steps): 13-31 % of the matrix rate on 1.7 ms of Darknet-53's 10.5 ms micro-batch.
// This kernel is the same GEMM (D[cout][pixel], packed fragments of conv_kernel, 64-channel chunks) as a software
// pipeline in which NO wave waits for a memory round trip in steady state:
//   - 8 waves, 128 pixels x 8*NT cout tiles per block; wave w owns cout tiles {NT w ..} for all 8 pixel segments;
//   - activations: the float4 units of chunk c + 1 are requested (4 per thread) BEFORE the K loop of chunk c, split
//     to hi/lo and written to the OTHER LDS buffer after it: TWO buffers, so ONE barrier per chunk (the buffer
//     written after K loop c was last read in K loop c - 1, before the barrier of that iteration);
//   - weights: each K-step's fragments are re-requested for chunk c + 1 as soon as their MFMAs of chunk c are issued;
//   - fragment reads run one half-step ahead of the MFMAs (two register sets, scheduling-group barriers);
//   - barriers are `s_waitcnt lgkmcnt(0); s_barrier`: they leave the prefetches in flight.
// One block per CU (<= 256 registers, 68 KB of LDS).  Flat pixels: a.Win = N*H*W, a.Cin % 64 == 0, float32 input,
// no residual operands.
constexpr int kW1Px = 128, kW1CS = 2 * 64 + kPadF16;                 // pixels per block; halfs per staged pixel [hi 64 | lo 64 | pad]
constexpr int kW1Buf = kW1Px * kW1CS;                                // halfs per LDS buffer
constexpr int kW1Lds = 2 * kW1Buf * 2;                               // bytes
template <int NT>
__global__ __launch_bounds__(512, 2) void conv1x1_wide_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* sm = reinterpret_cast<_Float16*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const ConvSub& S = a.sub[0];
  const int total = a.Win;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  int tile, by;
  if (a.group_major) {
    const int ntiles = gridDim.x / a.ny;
    by = lid / ntiles;
    tile = lid - by * ntiles;
  } else {
    tile = lid / a.ny;
    by = lid - tile * a.ny;
  }
  const int pix0 = tile * kW1Px;
  const int ct0 = (by * 8 + wave) * NT;
  const int nch = a.Cin >> 6;
  float vmax = 0.f;

  // staging: thread <-> (channel quad u of the chunk, pixels sp0 + 32 k): a wave covers 4 pixels x 256 contiguous bytes
  const int u = tid & 15, sp0 = tid >> 4;
  const float* src[4];
  bool sok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = pix0 + sp0 + 32 * k;
    sok[k] = px < total;
    src[k] = a.in + (size_t)(sok[k] ? px : 0) * a.Cin + u * 4;
  }
  auto load_acts = [&](const int c, f32x4 (&v)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src[k] + c * 64);
  };
  auto store_acts = [&](f32x4 (&v)[4], _Float16* const buf) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!sok[k]) v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f16x4 hi, lo;
      split4(v[k], hi, lo);
      vmax = absmax4(vmax, v[k]);
      _Float16* d = buf + (sp0 + 32 * k) * kW1CS + u * 4;
      *reinterpret_cast<f16x4*>(d) = hi;
      *reinterpret_cast<f16x4*>(d + 64) = lo;
    }
  };
  const _Float16* wbase = S.w16 + (size_t)ct0 * 1024 + lane * 8;
  const unsigned wstep = (unsigned)S.nctp * 1024u;
  // weight fragments: ONE register set per K-step of a chunk; a set is re-requested for the NEXT chunk as soon as its
  // K-step's MFMAs are issued (a chunk's matrix work = 1.5 k cycles per wave x 2 waves per SIMD covers the L2 trip)
  f16x8 wh[2][NT], wl[2][NT];
  auto load_w = [&](const int c, const int st) {
    const _Float16* wp = wbase + (size_t)(2 * c + st) * wstep;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      wh[st][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024);
      wl[st][nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512);
    }
  };
  f32x4 acc[8][NT];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int xoff = p * kW1CS + g * 8;
  // A chunk's K loop = 4 half-steps (K-step st = h / 2, segments 4 (h & 1) ..): the 8 fragment reads of half-step
  // h + 1 are issued BEFORE the 12 NT MFMAs of half-step h (two register sets), pinned by scheduling-group barriers:
  // left alone the scheduler emits `2 reads, wait, 6 MFMAs` and sinks every prefetch to just before its use.
  auto kloop = [&](const _Float16* const buf, const int cnext) {   // cnext: the chunk whose fragments to request next
    f16x8 xh[2][4], xl[2][4];
    auto rd = [&](const int h, const int slot) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const _Float16* xp = buf + xoff + (4 * (h & 1) + m) * (16 * kW1CS) + (h >> 1) * 32;
        xh[slot][m] = *reinterpret_cast<const f16x8*>(xp);
        xl[slot][m] = *reinterpret_cast<const f16x8*>(xp + 64);
      }
    };
    rd(0, 0);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int st = h >> 1, m0 = 4 * (h & 1), slot = h & 1;
      if (h + 1 < 4) rd(h + 1, slot ^ 1);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
          acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[st][nn], xh[slot][m], acc[m0 + m][nn], 0, 0, 0);
          acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st][nn], xl[slot][m], acc[m0 + m][nn], 0, 0, 0);
          acc[m0 + m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[st][nn], xh[slot][m], acc[m0 + m][nn], 0, 0, 0);
        }
      if (h + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12 * NT, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (h & 1) {                          // K-step st is done with its fragments
        load_w(cnext, st);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: chunk 0 staged, its fragments requested
  f32x4 v[4];
  load_acts(0, v);
  load_w(0, 0);
  load_w(0, 1);
  store_acts(v, sm);
  lds_barrier();
  // ---- steady state.  vmcnt retires in issue order, so a request is waited for as soon as anything issued AFTER
  // it is needed: the activations of chunk c + 1 go out before the K loop of chunk c (whose own fragments are
  // older) and are consumed right after it — one K loop (2 waves per SIMD x 96 MFMAs) of cover, with no younger
  // request needed in between; the fragments of chunk c + 1 go out inside that K loop, behind them.
  // (every prefetch is unconditional — past the last chunk it re-requests chunk nch - 1 and nobody reads it: a
  // branch around a request makes the compiler's wait-count pass join two histories and wait for everything)
  const int last = nch - 1;
  for (int c = 0; c < nch; ++c) {
    const int cn = c + 1 < nch ? c + 1 : last;
    _Float16* const cur = sm + (c & 1) * kW1Buf;
    _Float16* const nxt = sm + ((c + 1) & 1) * kW1Buf;
    load_acts(cn, v);
    __builtin_amdgcn_sched_barrier(0);      // (the requests stay HERE, ahead of the K loop they are hidden behind)
    kloop(cur, cn);
    if (c + 1 < nch) {
      store_acts(v, nxt);                   // (last read two K loops ago, with a barrier in between)
      lds_barrier();
    }
  }

  // ---- epilogue: bias (+ inverse weight scale), activation, float32 or split-f16 store
  {
    f32x4 bv[NT];
    float iv[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      bv[nn] = *reinterpret_cast<const f32x4*>(S.bias + (ct0 + nn) * 16 + g * 4);
      iv[nn] = sload(S.bias + (S.nctp + ct0 + nn) * 16);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int px = pix0 + m * 16 + p;
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) {
        const int co = (ct0 + nn) * 16 + g * 4;
        if (px < total && co < S.Cout) {
          const f32x4 v = act4(fma4(acc[m][nn], iv[nn], bv[nn]), S.act);
          if (a.out_s16) {
            f16x4 hi, lo;
            split4(v, hi, lo);
            vmax = absmax4(vmax, v);
            _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + (size_t)px * (size_t)(2 * a.out_C) + S.co_off + co;
            *reinterpret_cast<f16x4*>(o16) = hi;
            *reinterpret_cast<f16x4*>(o16 + a.out_C) = lo;
          } else {
            *reinterpret_cast<f32x4*>(a.out + (size_t)px * a.out_C + S.co_off + co) = v;
          }
        }
      }
    }
  }
  if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
}
#endif  // PCLSEG_CAND_WIDE

// ---- 1x1 convolutions without LDS (split-f16 mode).
// A 1x1 conv has no halo, so staging its input through LDS only buys the hi/lo split and costs two
// block barriers per channel chunk plus the LDS footprint.  Here every wave is an independent
// streaming GEMM: lane (p, g) loads the 8 channels {32t + 8g ..} of pixel p straight from
// global memory (4 lanes cover 128 contiguous bytes of a pixel), splits them to f16 hi/lo in
// registers and feeds the MFMAs; the operands of K-step t+1 are in flight while step t computes.
// No barriers, no LDS -> occupancy is bounded by registers only.  Uses the same packed weight
// fragments as conv_kernel (K-step t of the packed array covers channel groups 4t .. 4t+3 when
// the packing chunk is 32 or 64 channels).  Requires Cin % 8 == 0.
// SPLITK (deep layers: many channels, few pixels): the four waves of a block share one run of
// MTW*16 pixels and split the channels into four contiguous quarters, then add their partial
// accumulators through LDS — four times the waves per pixel for the memory system to work with,
// and each weight fragment is fetched once per MTW*16 pixels instead of once per wave.
template <int MTW, int NTW, bool RES, bool SPLITK = false>
__global__ __launch_bounds__(kConvThreads, (SPLITK && MTW * NTW == 8) ? 3 : 4) void conv1x1_direct_kernel(const ConvArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, g = lane >> 4;
  const ConvSub& S = a.sub[0];
  const int total = a.Win;  // flattened pixel count (N = H = 1 view)
  const int pix0 = SPLITK ? (int)blockIdx.x * (MTW * 16) : ((int)blockIdx.x * 4 + wave) * (MTW * 16);
  if (pix0 >= total) return;
  const int ct0 = blockIdx.y * NTW;
  const int cin8 = a.Cin >> 3;
  const int nsteps = (cin8 + 3) >> 2;

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* xptr[MTW];
  bool pvalid[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const int px = pix0 + m * 16 + p;
    pvalid[m] = px < total;
    xptr[m] = a.in + (size_t)(pvalid[m] ? px : 0) * a.Cin + g * 8;
  }
  const _Float16* wbase = S.w16 + (size_t)ct0 * 1024 + lane * 8;

  f32x4 xa[MTW], xb[MTW];       // raw float32 operands of the step being prefetched
  f16x8 wh[NTW], wl[NTW];
  auto load_step = [&](int t) {
    const bool gok = (4 * t + g) < cin8;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      const float* src = (gok && pvalid[m]) ? xptr[m] + t * 32 : a.in;
      xa[m] = *reinterpret_cast<const f32x4*>(src);
      xb[m] = *reinterpret_cast<const f32x4*>(src + 4);
      if (!(gok && pvalid[m])) { xa[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; xb[m] = xa[m]; }
    }
    const _Float16* wp = wbase + (size_t)t * S.nctp * 1024;
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) {
      wh[nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024);
      wl[nn] = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512);
    }
  };

  int t_lo = 0, t_hi = nsteps;
  if constexpr (SPLITK) {
    const int per = (nsteps + 3) >> 2;
    t_lo = wave * per;
    t_hi = t_lo + per < nsteps ? t_lo + per : nsteps;
  }
  float vmax = 0.f;  // range guard: largest |value| split to f16 hi/lo by this thread
  // one place for the output store: float32 quad, or split-f16 pair format (see ConvArgs)
  auto store_out = [&](size_t px, int co, const f32x4 v) {
    if (a.out_s16) {
      f16x4 hi, lo;
      split4(v, hi, lo);
      vmax = absmax4(vmax, v);
      _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + px * (size_t)(2 * a.out_C) + S.co_off + co;
      *reinterpret_cast<f16x4*>(o16) = hi;
      *reinterpret_cast<f16x4*>(o16 + a.out_C) = lo;
    } else {
      *reinterpret_cast<f32x4*>(a.out + px * a.out_C + S.co_off + co) = v;
    }
  };
  if (t_lo < t_hi) load_step(t_lo);
  for (int t = t_lo; t < t_hi; ++t) {
    // split this step's activations, keep its weights, then refill the raw registers
    f16x8 xh[MTW], xl[MTW], ch[NTW], cl[NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      vmax = absmax4(absmax4(vmax, xa[m]), xb[m]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const _Float16 h0 = (_Float16)xa[m][e];
        const _Float16 h1 = (_Float16)xb[m][e];
        xh[m][e] = h0;
        xh[m][4 + e] = h1;
        xl[m][e] = (_Float16)(xa[m][e] - (float)h0);
        xl[m][4 + e] = (_Float16)(xb[m][e] - (float)h1);
      }
    }
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) { ch[nn] = wh[nn]; cl[nn] = wl[nn]; }
    if (t + 1 < t_hi) load_step(t + 1);
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn) {
        acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[nn], xh[m], acc[m][nn], 0, 0, 0);
        acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[nn], xl[m], acc[m][nn], 0, 0, 0);
        acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[nn], xh[m], acc[m][nn], 0, 0, 0);
      }
  }

  if constexpr (SPLITK) {
    // partial accumulators -> LDS [wave][tile][lane]; tile t is finished by wave t & 3
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
    constexpr int NT = MTW * NTW;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn) red[(wave * NT + m * NTW + nn) * 64 + lane] = acc[m][nn];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn) {
        constexpr int dummy = 0; (void)dummy;
        const int tile = m * NTW + nn;
        if ((tile & 3) != wave) continue;
        f32x4 v = red[tile * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[(w * NT + tile) * 64 + lane];
        const int co = (ct0 + nn) * 16 + g * 4;
        if (pvalid[m] && co < S.Cout) {
          const size_t px = (size_t)(pix0 + m * 16 + p);
          v = act4(fma4(v, sload(S.bias + (S.nctp + ct0 + nn) * 16), *reinterpret_cast<const f32x4*>(S.bias + co)), S.act);
          if constexpr (RES) if (a.res1) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(a.res1 + px * a.res1_C + S.co_off + co);
            v = a.res1_mul ? v * r : v + r;
          }
          store_out(px, co, v);
        }
      }
    if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
    return;
  }

  // epilogue (two passes: residual operands first, then the stores; see conv_kernel)
  f32x4 r1[RES ? MTW : 1][NTW];
  if constexpr (RES) if (a.res1) {
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn) {
        const int co = (ct0 + nn) * 16 + g * 4;
        const bool ok = pvalid[m] && co < S.Cout;
        r1[m][nn] = *reinterpret_cast<const f32x4*>(
            ok ? a.res1 + (size_t)(pix0 + m * 16 + p) * a.res1_C + S.co_off + co : a.res1);
      }
  }
#pragma unroll
  for (int nn = 0; nn < NTW; ++nn) {
    const int co = (ct0 + nn) * 16 + g * 4;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(S.bias + co);
    const float iv = sload(S.bias + (S.nctp + ct0 + nn) * 16);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
      if (pvalid[m] && co < S.Cout) {
        f32x4 v = act4(fma4(acc[m][nn], iv, bv), S.act);
        if constexpr (RES) if (a.res1) v = a.res1_mul ? v * r1[m][nn] : v + r1[m][nn];
        store_out((size_t)(pix0 + m * 16 + p), co, v);
      }
    }
  }
  if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
}

// ---- 3x3 / strides (1,2) SAME max-pool fused into the 1x1 squeeze that is its only reader
// (pool1 -> fire2, pool3 -> fire4, pool5 -> fire6; nets/SqueezeSegV2.py:295-306).  The pooled tensor is
// never written.  A block owns 16 output columns x 4 rows x all C channels:
//   pool phase   thread <-> (column, channel quad) exactly as maxpool3x3s2_kernel: 6 rows x 3 columns of
//                16-byte loads, consecutive lanes on consecutive channels (a quarter-wave covers whole
//                cache lines; loading in the MFMA operand layout instead — 16 lanes on 16 different
//                pixels — issues 8x the cache-line requests and was bound by the L1 tag rate), column max
//                then row max in registers, split to f16 hi/lo, 8-byte LDS writes [pixel][hi C | lo C];
//   GEMM phase   wave w <-> output row w: B operands are two ds_read_b128 per 32-channel step, weights the
//                packed fragments of conv1x1_direct_kernel, then bias + ReLU + split-f16 store.
// Out-of-image taps are clamped onto an in-window pixel (a duplicate never changes a max).
constexpr int kPoolSqRows = 4;
template <int NTW>
__global__ __launch_bounds__(kConvThreads) void pool_squeeze_kernel(const ConvArgs a) {
  constexpr int ROWS = kPoolSqRows;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  _Float16* xs = reinterpret_cast<_Float16*>(smem_raw);   // [ROWS*16 px][2*C + kPadF16]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvSub& S = a.sub[0];
  const int C = a.Cin, c4n = C >> 2, pstride = 2 * C + kPadF16;
  const int cgn = (a.Wout + 15) >> 4, rbn = (a.H + ROWS - 1) / ROWS;
  int unit = xcd_remap(blockIdx.x, gridDim.x);
  const int cg = unit % cgn;
  unit /= cgn;
  const int h0 = (unit % rbn) * ROWS;
  const int n = unit / rbn;
  const float* base = a.in + (size_t)n * a.H * a.Win * C;
  int rowoff[ROWS + 2];
#pragma unroll
  for (int r = 0; r < ROWS + 2; ++r) {
    const int hh = h0 - 1 + r;
    rowoff[r] = (hh < 0 ? 0 : hh >= a.H ? a.H - 1 : hh) * a.Win * C;
  }
  float vmax = 0.f;
  const int lc4 = 31 - __builtin_clz(c4n);   // c4n is a power of two (host-checked)
  for (int item = tid; item < 16 * c4n; item += kConvThreads) {
    const int cl = item >> lc4, c4 = item & (c4n - 1);
    const int wo = cg * 16 + cl;
    int col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ww = wo * 2 - a.pl + j;
      col[j] = (ww < 0 ? 0 : ww >= a.Win ? a.Win - 1 : ww) * C + c4 * 4;
    }
    f32x4 rm[ROWS + 2];
#pragma unroll
    for (int r = 0; r < ROWS + 2; ++r) {
      const float* row = base + rowoff[r];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + col[0]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + col[1]);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(row + col[2]);
#pragma unroll
      for (int e = 0; e < 4; ++e) rm[r][e] = fmaxf(fmaxf(v0[e], v1[e]), v2[e]);
    }
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(rm[m][e], rm[m + 1][e]), rm[m + 2][e]);
      vmax = absmax4(vmax, v);
      f16x4 hi, lo;
      split4(v, hi, lo);
      _Float16* d = xs + (m * 16 + cl) * pstride + c4 * 4;
      *reinterpret_cast<f16x4*>(d) = hi;
      *reinterpret_cast<f16x4*>(d + C) = lo;
    }
  }
  __syncthreads();

  const int p = lane & 15, g = lane >> 4;
  f32x4 acc[NTW];
#pragma unroll
  for (int nn = 0; nn < NTW; ++nn) acc[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const _Float16* xrow = xs + (wave * 16 + p) * pstride + g * 8;
  const _Float16* wbase = S.w16 + lane * 8;
  const int nsteps = C >> 5;
  for (int t = 0; t < nsteps; ++t) {
    const f16x8 xh = *reinterpret_cast<const f16x8*>(xrow + t * 32);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(xrow + t * 32 + C);
    const _Float16* wp = wbase + (size_t)t * S.nctp * 1024;
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) {
      const f16x8 wh = *reinterpret_cast<const f16x8*>(wp + nn * 1024);
      const f16x8 wl = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512);
      acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[nn], 0, 0, 0);
      acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[nn], 0, 0, 0);
      acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[nn], 0, 0, 0);
    }
  }
  const int wo = cg * 16 + p, h = h0 + wave;
  if (wo < a.Wout && h < a.H) {
    const size_t px = ((size_t)n * a.H + h) * a.Wout + wo;
#pragma unroll
    for (int nn = 0; nn < NTW; ++nn) {
      const int co = nn * 16 + g * 4;
      if (co >= S.Cout) continue;
      const f32x4 v = act4(fma4(acc[nn], sload(S.bias + (S.nctp + nn) * 16), *reinterpret_cast<const f32x4*>(S.bias + co)), S.act);
      if (a.out_s16) {
        f16x4 hi, lo;
        split4(v, hi, lo);
        vmax = absmax4(vmax, v);
        _Float16* o16 = reinterpret_cast<_Float16*>(a.out) + px * (size_t)(2 * a.out_C) + S.co_off + co;
        *reinterpret_cast<f16x4*>(o16) = hi;
        *reinterpret_cast<f16x4*>(o16 + a.out_C) = lo;
      } else {
        *reinterpret_cast<f32x4*>(a.out + px * a.out_C + S.co_off + co) = v;
      }
    }
  }
  if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
}

// ---- MaxPool kh x kw, strides (1, sw), TF SAME (padding never wins)            (K6, K7)
// Generic fallback (stand-alone op API and shapes the fused kernels do not cover).
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in,
                                                      float* __restrict__ out, int N, int H, int Win,
                                                      int Wout, int C, int kh, int kw, int sw, int pt,
                                                      int pl) {
  const int c4n = C >> 2;
  const size_t total = (size_t)N * H * Wout * c4n;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    size_t pix = idx / c4n;
    const int wo = (int)(pix % Wout);
    pix /= Wout;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    f32x4 m = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < kh; ++i) {
      const int hh = h - pt + i;
      if (hh < 0 || hh >= H) continue;
      const float* row = in + ((size_t)n * H + hh) * Win * C + c4 * 4;
      for (int j = 0; j < kw; ++j) {
        const int ww = wo * sw - pl + j;
        if (ww < 0 || ww >= Win) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)ww * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *reinterpret_cast<f32x4*>(out + idx * 4) = m;
  }
}

// ---- Context Aggregation Module, one kernel (reference: nets/SqueezeSegV2.py:30-70)
//   out = x * sigmoid(BN(W2 . relu(BN(W1 . maxpool7x7_s1_SAME(x)))))          C -> C/16 -> C
// Everything is float32 on the VALU (the two 1x1 convs are 2*C*C/16 MACs per pixel).
struct CamArgs {
  const float* x;   // [N,H,W,C]
  float* out;       // [N,H,W,C]
  const float* w1;  // [C][R]  BN-folded squeeze weights
  const float* b1;  // [R]
  const float* w2;  // [R][C]  BN-folded excitation weights
  const float* b2;  // [C]
  int N, H, W, tilesH, tilesW;
  // fused squeeze of the FIRE module that is this CAM's only reader (cam_kernel SQ > 0): out = nullptr,
  // the gated tile goes through LDS to the matrix cores and only the squeeze output is written
  const _Float16* sq_w16;   // packed 1x1 fragments [C/32 steps][SQ tiles][hi|lo][lane][8]
  const float* sq_bias;     // [SQ*16]
  float* sq_out;            // [N,H,W,sq_C] float32, or split-f16 pair format when sq_s16
  int sq_C, sq_s16;
  unsigned* range_flag;
};

// A block owns a TH x 26 pixel tile; its (TH+6) x 32 halo patch is swept in CK-channel chunks.
// Thread (pc, q) owns patch column pc and channel quad q: it loads the column's TH+6 rows straight
// into registers (CK*4 contiguous bytes per pixel), KEEPS the TH tile rows for the gate pass,
// takes the 7-tall column max in registers and publishes only those TH values to LDS; after one
// barrier the 7-wide row max is read back from LDS and folded into the squeeze sums, which are
// reduce-scattered over the pixel's CK/4 quad lanes.  x is read from memory exactly once; the
// 7x7 max is separable and associative, so it is bit-identical to the 49-tap window.
constexpr int kCamTW = 26, kCamPW = 32;

// Cross-lane sum over the 16 lanes of a pixel on the DPP network (one VALU instruction per exchange, no
// LDS round trip as with ds_bpermute): recursive halving — mirror / half-mirror / quad-perm pairings, each
// step keeps half of the values and adds the partner's copy of them.  To keep every step the SAME register
// pattern in every lane (no per-lane selects), lane q stores partial sum number v in slot v ^ L,
// L = q & (R-1): a partner is always lane L ^ mask, so "my slot s pairs with its slot s ^ mask".
// The caller fills p[s] with the partial of squeeze channel s ^ L (it reads its weights in that order).
// Returns, in every lane, the total of squeeze channel q & (R-1); must be called from convergent code.
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int R>
__device__ __forceinline__ float cam_reduce_scatter(const float (&p)[R]) {
  static_assert(R == 4 || R == 8, "unsupported CAM geometry");
  constexpr int kRowMirror = 0x140, kHalfMirror = 0x141, kXor2 = 0x4E, kXor1 = 0xB1;
  float a1[R];      // lanes q and 15-q (L ^ (R-1)) combined: all R values stay
#pragma unroll
  for (int i = 0; i < R; ++i) a1[i] = p[i] + dpp_read<kRowMirror>(p[i ^ (R - 1)]);
  float k4[4];
  if constexpr (R == 8) {     // lanes u and 7-u (L ^ 7): keep slots 0..3
#pragma unroll
    for (int i = 0; i < 4; ++i) k4[i] = a1[i] + dpp_read<kHalfMirror>(a1[7 - i]);
  } else {                    // four values over eight lanes: one more full exchange (L ^ 3)
#pragma unroll
    for (int i = 0; i < 4; ++i) k4[i] = a1[i] + dpp_read<kHalfMirror>(a1[3 - i]);
  }
  const float k20 = k4[0] + dpp_read<kXor2>(k4[2]);   // L ^ 2: keep slots 0, 1
  const float k21 = k4[1] + dpp_read<kXor2>(k4[3]);
  return k20 + dpp_read<kXor1>(k21);                  // L ^ 1: keep slot 0 = value L
}

// SQ > 0: the module's output feeds only the next FIRE squeeze (cam2 -> fire3, nets/SqueezeSegV2.py:297-298):
// the gate pass writes each 64-channel chunk of the gated tile as split-f16 into LDS (the dead column-max
// buffer), wave w < 7 accumulates pixel segment w x SQ cout tiles on the matrix cores, and the block
// writes bias + ReLU of that instead of the 128-channel tensor.
template <int C, int R, int TH, int CK, int SQ = 0>
__global__ __launch_bounds__(8 * CK, 4) void cam_kernel(const CamArgs a) {
  constexpr int TW = kCamTW, PW = kCamPW, PH = TH + 6, NCH = C / CK;
  constexpr int QP = CK / 4, LQ = 4;  // channel quads per chunk = lanes per pixel = one DPP row
  static_assert(QP == 16, "cam_reduce_scatter works on one 16-lane DPP row per pixel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* colmax = reinterpret_cast<float*>(smem_raw);  // [TH][PW][CK]
  float* w1c = colmax + TH * PW * CK;                  // [CK][R]
  const int tid = threadIdx.x;
  const int pc = tid >> LQ, q = tid & (QP - 1);
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int twi = tile % a.tilesW;
  tile /= a.tilesW;
  const int thi = tile % a.tilesH;
  const int n = tile / a.tilesH;
  const int h0 = thi * TH, w0 = twi * TW;
  const float* xn = a.x + (size_t)n * a.H * a.W * C;
  const int w = w0 - 3 + pc;
  const bool wok = w >= 0 && w < a.W;
  const bool interior = pc >= 3 && pc < 3 + TW;
  // out-of-image taps are clamped onto the nearest image pixel: it lies inside every 7x7 window that
  // would have seen the padding, and a duplicate never changes a max (no per-load select)
  const int wcl = w < 0 ? 0 : w >= a.W ? a.W - 1 : w;

  f32x4 v[PH];
  auto load_chunk = [&](int chunk) {
#pragma unroll
    for (int pr = 0; pr < PH; ++pr) {
      const int h = h0 - 3 + pr;
      const int hcl = h < 0 ? 0 : h >= a.H ? a.H - 1 : h;
      v[pr] = *reinterpret_cast<const f32x4*>(xn + ((size_t)hcl * a.W + wcl) * C + chunk * CK + q * 4);
    }
  };

  f32x4 xs_reg[NCH][TH];  // this thread's tile pixels (rows h0.., column w), kept for the gate
  float sp[TH];
#pragma unroll
  for (int r = 0; r < TH; ++r) sp[r] = 0.f;

  load_chunk(0);
#pragma unroll
  for (int chunk = 0; chunk < NCH; ++chunk) {
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      xs_reg[chunk][r] = v[3 + r];
    }
    {  // 7-tall running max: triples shared between neighbouring outputs (v_max3_f32)
      f32x4 t3[PH - 2];
#pragma unroll
      for (int i = 0; i < PH - 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) t3[i][e] = fmaxf(fmaxf(v[i][e], v[i + 1][e]), v[i + 2][e]);
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(t3[r][e], t3[r + 3][e]), v[r + 6][e]);
        *reinterpret_cast<f32x4*>(colmax + (r * PW + pc) * CK + q * 4) = m;
      }
    }
    if (tid < CK * R / 4)
      *reinterpret_cast<f32x4*>(w1c + tid * 4) =
          *reinterpret_cast<const f32x4*>(a.w1 + (size_t)chunk * CK * R + tid * 4);
    __syncthreads();
    if (chunk + 1 < NCH) load_chunk(chunk + 1);
    {  // all lanes run the row pass (exterior columns on a clamped window, results unused) so
       // the cross-lane reduction below is never inside divergent control flow
      const int pcc = interior ? pc - 3 : 0;
      float w1r[4][R];   // slot s = squeeze channel s ^ (q & (R-1)), see cam_reduce_scatter
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int sl = 0; sl < R; ++sl) w1r[e][sl] = w1c[(q * 4 + e) * R + (sl ^ (q & (R - 1)))];
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const float* src = colmax + (r * PW + pcc) * CK + q * 4;
        f32x4 m = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int j = 1; j < 7; ++j) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(src + j * CK);
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], t[e]);
        }
        float p[R];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
          float t = m[0] * w1r[0][rr];
#pragma unroll
          for (int e = 1; e < 4; ++e) t = fmaf(m[e], w1r[e][rr], t);
          p[rr] = t;
        }
        sp[r] += cam_reduce_scatter<R>(p);
      }
    }
    __syncthreads();
  }

  // squeeze output: + bias, ReLU -> s[TH*TW][R] (aliases colmax; the loop ended on a barrier)
  float* s_lds = colmax;
  if (interior && q < R) {
    const int rq = q;
    const float b = a.b1[rq];
#pragma unroll
    for (int r = 0; r < TH; ++r) s_lds[(r * TW + pc - 3) * R + rq] = fmaxf(sp[r] + b, 0.f);
  }
  __syncthreads();

  // gate: out = x * sigmoid(W2 . s + b2) on the register-resident tile pixels
  const bool active = interior && wok;
  float* outn = a.out + (size_t)n * a.H * a.W * C;
  // fused squeeze: gated chunk -> LDS [TH*TW px (+ pad to 16)][CK hi | CK lo | pad] behind s_lds
  constexpr int kSegs = (TH * TW + 15) / 16, kXS = 2 * CK + kPadF16;
  static_assert(SQ == 0 || kSegs <= 8, "one pixel segment per wave");
  _Float16* xs = reinterpret_cast<_Float16*>(smem_raw + ((TH * TW * R * 4 + 15) & ~15));
  const int lane = tid & 63, wave = tid >> 6;
  f32x4 acc[SQ > 0 ? SQ : 1];
#pragma unroll
  for (int nn = 0; nn < (SQ > 0 ? SQ : 1); ++nn) acc[nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float vmax = 0.f;
#ifdef PCLSEG_CAND_CAM
  // fused squeeze: a chunk's weight fragments are requested one gate pass AHEAD of their MFMAs (they were fetched
  // right in front of them: NCH * CK/32 exposed L2 round trips per block, in a kernel whose blocks are short
  // chains of dependent phases; all chunks at once costs 16 more registers than the 128 this kernel may use)
  constexpr int kSqSteps = CK / 32, kSqT = SQ > 0 ? SQ : 1;
  f16x8 swh[kSqSteps][kSqT], swl[kSqSteps][kSqT];
  auto load_sqw = [&](const int chunk) {
#pragma unroll
    for (int t = 0; t < kSqSteps; ++t)
#pragma unroll
      for (int nn = 0; nn < kSqT; ++nn) {
        const _Float16* wp = a.sq_w16 + ((size_t)(chunk * kSqSteps + t) * kSqT + nn) * 1024 + lane * 8;
        swh[t][nn] = *reinterpret_cast<const f16x8*>(wp);
        swl[t][nn] = *reinterpret_cast<const f16x8*>(wp + 512);
      }
  };
  if constexpr (SQ > 0) load_sqw(0);
#endif
#pragma unroll
  for (int chunk = 0; chunk < NCH; ++chunk) {
    if (active) {
      f32x4 w2r[R];
#pragma unroll
      for (int rr = 0; rr < R; ++rr)
        w2r[rr] = *reinterpret_cast<const f32x4*>(a.w2 + rr * C + chunk * CK + q * 4);
      const f32x4 bz = *reinterpret_cast<const f32x4*>(a.b2 + chunk * CK + q * 4);
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const int h = h0 + r;
        if (SQ > 0 || h < a.H) {
          f32x4 z = bz;
#pragma unroll
          for (int rr = 0; rr < R; rr += 4) {
            const f32x4 sv = *reinterpret_cast<const f32x4*>(s_lds + (r * TW + pc - 3) * R + rr);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int e = 0; e < 4; ++e) z[e] = fmaf(sv[i], w2r[rr + i][e], z[e]);
          }
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)  // sigmoid on the transcendental unit (v_exp_f32 + v_rcp_f32, ~1 ulp each)
            o[e] = xs_reg[chunk][r][e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z[e]));
          if constexpr (SQ > 0) {
            if (h >= a.H) o = (f32x4){0.f, 0.f, 0.f, 0.f};   // (rows below the image hold clamped duplicates; never stored)
            f16x4 hi, lo;
            split4(o, hi, lo);
            vmax = absmax4(vmax, o);
            _Float16* d = xs + (r * TW + pc - 3) * kXS + q * 4;
            *reinterpret_cast<f16x4*>(d) = hi;
            *reinterpret_cast<f16x4*>(d + CK) = lo;
          } else {
            *reinterpret_cast<f32x4*>(outn + ((size_t)h * a.W + w) * C + chunk * CK + q * 4) = o;
          }
        }
      }
    }
    if constexpr (SQ > 0) {
      __syncthreads();
      if (wave < kSegs) {
        const int p = lane & 15, g = lane >> 4;
        const _Float16* xrow = xs + (wave * 16 + p) * kXS + g * 8;
#pragma unroll
        for (int t = 0; t < CK / 32; ++t) {
          const f16x8 xh = *reinterpret_cast<const f16x8*>(xrow + t * 32);
          const f16x8 xl = *reinterpret_cast<const f16x8*>(xrow + t * 32 + CK);
#ifdef PCLSEG_CAND_CAM
#pragma unroll
          for (int nn = 0; nn < SQ; ++nn) {
            const f16x8 wh = swh[t][nn], wl = swl[t][nn];
#else
          const _Float16* wp = a.sq_w16 + ((size_t)(chunk * (CK / 32) + t) * SQ) * 1024 + lane * 8;
#pragma unroll
          for (int nn = 0; nn < SQ; ++nn) {
            const f16x8 wh = *reinterpret_cast<const f16x8*>(wp + nn * 1024);
            const f16x8 wl = *reinterpret_cast<const f16x8*>(wp + nn * 1024 + 512);
#endif
            acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[nn], 0, 0, 0);
            acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[nn], 0, 0, 0);
            acc[nn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[nn], 0, 0, 0);
          }
        }
      }
#ifdef PCLSEG_CAND_CAM
      if (chunk + 1 < NCH) {
        asm volatile("" ::: "memory");   // (the next chunk's fragments: requested AFTER this chunk's MFMAs have read theirs)
        load_sqw(chunk + 1);
        lds_barrier();                   // (orders LDS only: a __syncthreads() would drain vmcnt and wait for the request right here)
      }
#else
      if (chunk + 1 < NCH) __syncthreads();
#endif
    }
  }
  if constexpr (SQ > 0) {
    const int p = lane & 15, g = lane >> 4;
    const int i = wave * 16 + p;          // tile pixel of this lane's accumulator column
    const int r = i / TW, c = i - r * TW;
    const int h = h0 + r, wc = w0 + c;
    if (wave < kSegs && i < TH * TW && h < a.H && wc < a.W) {
      const size_t px = ((size_t)n * a.H + h) * a.W + wc;
#pragma unroll
      for (int nn = 0; nn < SQ; ++nn) {
        const int co = nn * 16 + g * 4;
        if (co >= a.sq_C) continue;
        f32x4 v = fma4(acc[nn], sload(a.sq_bias + (SQ + nn) * 16), *reinterpret_cast<const f32x4*>(a.sq_bias + co));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
        if (a.sq_s16) {
          f16x4 hi, lo;
          split4(v, hi, lo);
          vmax = absmax4(vmax, v);
          _Float16* o16 = reinterpret_cast<_Float16*>(a.sq_out) + px * (size_t)(2 * a.sq_C) + co;
          *reinterpret_cast<f16x4*>(o16) = hi;
          *reinterpret_cast<f16x4*>(o16 + a.sq_C) = lo;
        } else {
          *reinterpret_cast<f32x4*>(a.sq_out + px * a.sq_C + co) = v;
        }
      }
    }
    if (vmax >= kF16Max && a.range_flag) atomicOr(a.range_flag, 1u);
  }
}

// 3x3 stride (1,2) pool (K7), the only shape SqueezeSegV2 keeps as its own pass.  A thread owns one
// channel quad of ROWS vertically adjacent outputs: (ROWS+2) x 3 branch-free loads, all in flight
// together (out-of-image taps are clamped onto an in-window pixel: a duplicate never changes a max).
template <int ROWS>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ in,
                                                           float* __restrict__ out, int N, int H,
                                                           int Win, int Wout, int C, int pl) {
  // grid = (ceil(Wout * C/4 / 256), row blocks, N): no integer division per thread (there is no
  // hardware divide: one costs ~25 VALU instructions, more than the 40 max3 this thread does)
  const int c4n = C >> 2;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= Wout * c4n) return;
  const int wo = (c4n & (c4n - 1)) == 0 ? x >> (31 - __builtin_clz(c4n)) : x / c4n;
  const int c4 = x - wo * c4n;
  const int n = blockIdx.z;
  const int h0 = blockIdx.y * ROWS;
  int col[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int ww = wo * 2 - pl + j;
    col[j] = (ww < 0 ? 0 : ww >= Win ? Win - 1 : ww) * C + c4 * 4;
  }
  const float* base = in + (size_t)n * H * Win * C;
  f32x4 rm[ROWS + 2];
#pragma unroll
  for (int r = 0; r < ROWS + 2; ++r) {
    const int hh = h0 - 1 + r;
    const float* row = base + (size_t)(hh < 0 ? 0 : hh >= H ? H - 1 : hh) * Win * C;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(row + col[0]);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(row + col[1]);
    const f32x4 v2 = *reinterpret_cast<const f32x4*>(row + col[2]);
#pragma unroll
    for (int e = 0; e < 4; ++e) rm[r][e] = fmaxf(fmaxf(v0[e], v1[e]), v2[e]);
  }
  float* o = out + (((size_t)n * H + h0) * Wout + wo) * C + c4 * 4;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    if (h0 + i < H) {
      f32x4 m;
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(rm[i][e], rm[i + 1][e]), rm[i + 2][e]);
      *reinterpret_cast<f32x4*>(o + (size_t)i * Wout * C) = m;
    }
  }
}

// ---- normalise + depth mask (reference: inference.py:50-62), float64 arithmetic like NumPy   (K1)
// scans [P,5] -> lidar [P,CO] (CO = 6 for the caller-visible tensor, 8 = zero-padded network
// input) and mask [P].
struct NormArgs {
  double mean[5];
  double std[5];
};

template <int CO>
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ scans,
                                                        float* __restrict__ lidar,
                                                        uint8_t* __restrict__ mask, size_t P,
                                                        const NormArgs na) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P;
       i += (size_t)gridDim.x * blockDim.x) {
    const float* s = scans + i * 5;
    float o[CO];
    const bool valid = s[4] > 0.0f;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const double v = ((double)s[c] - na.mean[c]) / na.std[c];
      o[c] = valid ? (float)v : 0.0f;
    }
    o[5] = valid ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 6; c < CO; ++c) o[c] = 0.0f;
#pragma unroll
    for (int c = 0; c < CO; ++c) lidar[i * CO + c] = o[c];
    if (mask) mask[i] = valid ? 1 : 0;
  }
}

// caller-provided lidar [P,6] -> network input [P,8]
__global__ __launch_bounds__(256) void pad6to8_kernel(const float* __restrict__ lidar6,
                                                      float* __restrict__ lidar8, size_t P) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P;
       i += (size_t)gridDim.x * blockDim.x) {
    const float* s = lidar6 + i * 6;
    f32x4 a = (f32x4){s[0], s[1], s[2], s[3]};
    f32x4 b = (f32x4){s[4], s[5], 0.f, 0.f};
    *reinterpret_cast<f32x4*>(lidar8 + i * 8) = a;
    *reinterpret_cast<f32x4*>(lidar8 + i * 8 + 4) = b;
  }
}

// ---- confusion matrix for the evaluation metrics (reference: eval.py:41-48 ->
// tf.metrics.MeanIoU.update_state; utils/util.py:64-79).  cm[label][pred] += 1 over all pixels,
// exact integer counts: per-block histogram in LDS, then one 64-bit atomic per non-zero cell.
// Entries whose label or prediction is outside [0, NC) are ignored.
__global__ __launch_bounds__(256) void confusion_kernel(const int32_t* __restrict__ labels,
                                                        const int32_t* __restrict__ preds, size_t count,
                                                        int NC, unsigned long long* __restrict__ cm) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned int* hist = reinterpret_cast<unsigned int*>(smem_raw);
  const int cells = NC * NC;
  for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (size_t)gridDim.x * blockDim.x) {
    const int l = labels[i], q = preds[i];
    if (l >= 0 && l < NC && q >= 0 && q < NC) atomicAdd(&hist[l * NC + q], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cells; i += blockDim.x) {
    const unsigned int v = hist[i];
    if (v) atomicAdd(&cm[i], (unsigned long long)v);
  }
}

// ---- spherical projection: point cloud -> range image (the step before the network).
// Three reference variants share one scatter / gather pair:
//   rows   FOV : py = floor((1 - (asin(z/depth) + |fov_down|) / fov) * H), clamped
//                (dataset_convert/laserscan_semantic_kitti.py:106-166, laserscan_nuscenes.py:226-288)
//          RING: py = H - 1 - ring_index          (laserscan_nuscenes.py:191-223,
//                preprocessing/convert_validation_pcd_to_npy.py:147-153)
//   cols   FULL : px = floor(0.5 * (-atan2(y,x)/pi + 1) * W), clamped (float32 arithmetic in NumPy's
//                 operation order, contraction disabled; atan2/asin in float64, rounded once)
//          FRONT: px = (int)((left_phi - atan2(y,x)) / ((right_phi+left_phi)/W)) in float64,
//                 truncated toward zero; points outside [0, W) are dropped
//                 (convert_validation_pcd_to_npy.py:120-137)
//   winner NEAREST: the reference sorts by decreasing depth and scatters, so the nearest point wins:
//                 one 64-bit atomicMin on (depth bits << 32) | index (positive floats order like their
//                 bit patterns; equal depths -> lowest index)
//          LAST : plain fancy-index assignment in input order, the LAST point of a pixel wins:
//                 atomicMax on index + 1 (0 = empty)
// then one gather per pixel: x, y, z, remission, depth [, label through an optional look-up table
// (the converters' learning_map) [, mask = depth > 0]].  Points at the origin are skipped in FOV mode.
struct ProjArgs {
  int H, W;
  int row_mode, col_mode, winner, out_c;
  float fpi, fdown, ffov;  // float32(pi), float32(|fov_down| rad), float32(fov rad)
  double left_phi, dphi;   // FRONT columns
  int stride;              // floats per point (x, y, z at 0..2, remission at 3)
  const int32_t* ring;     // RING rows
  const float* depth;      // optional per-point depth (else the float32 norm)
  const int32_t* labels;   // optional
  const int32_t* lut;      // optional label look-up table
  int lut_size;
  float empty;
};

__global__ __launch_bounds__(256) void proj_init_kernel(unsigned long long* __restrict__ keys, int n,
                                                        unsigned long long v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) keys[i] = v;
}

__global__ __launch_bounds__(256) void proj_scatter_kernel(const float* __restrict__ pts, size_t m,
                                                           unsigned long long* __restrict__ keys,
                                                           const ProjArgs a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
    const float* pt = pts + i * a.stride;
    const float x = pt[0], y = pt[1], z = pt[2];
    float depth;
    int ix, iy;
    {
#pragma clang fp contract(off)   // NumPy does not fuse multiply-add: keep every rounding
      const float d2 = (x * x + y * y) + z * z;
      depth = a.depth ? a.depth[i] : sqrtf(d2);
      if (a.col_mode == 0) {
        const float yaw = -(float)atan2((double)y, (double)x);
        float px = 0.5f * (yaw / a.fpi + 1.0f);
        px = floorf(px * (float)a.W);
        ix = (int)fmaxf(0.0f, fminf((float)(a.W - 1), px));
      } else {
        const double phi = atan2((double)y, (double)x);
        const double c = (a.left_phi - phi) / a.dphi;
        if (!(c > -2147483648.0 && c < 2147483648.0)) continue;
        ix = (int)c;                                   // astype(int): truncation toward zero
        if (ix < 0 || ix >= a.W) continue;             // np.delete of the out-of-window points
      }
      if (a.row_mode == 0) {
        if (!(depth > 0.0f)) continue;
        const float pitch = (float)asin((double)(z / depth));
        float py = 1.0f - (pitch + a.fdown) / a.ffov;
        py = floorf(py * (float)a.H);
        iy = (int)fmaxf(0.0f, fminf((float)(a.H - 1), py));
      } else {
        iy = a.H - 1 - a.ring[i];
        if (iy < 0) iy += a.H;                         // NumPy negative index wraps once
        if (iy < 0 || iy >= a.H) continue;             // (the reference would raise IndexError)
      }
    }
    unsigned long long* cell = &keys[(size_t)iy * a.W + ix];
    if (a.winner == 0)
      atomicMin(cell, ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned long long)(unsigned)i);
    else
      atomicMax(cell, (unsigned long long)i + 1ull);
  }
}

__global__ __launch_bounds__(256) void proj_gather_kernel(const float* __restrict__ pts,
                                                          const unsigned long long* __restrict__ keys,
                                                          int npix, float* __restrict__ image,
                                                          int32_t* __restrict__ proj_idx, const ProjArgs a) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    const unsigned long long key = keys[i];
    float o[7] = {a.empty, a.empty, a.empty, a.empty, a.empty, 0.0f, 0.0f};
    int idx = -1;
    const bool hit = a.winner == 0 ? key != ~0ull : key != 0ull;
    if (hit) {
      idx = a.winner == 0 ? (int)(unsigned)(key & 0xffffffffull) : (int)(key - 1ull);
      const float* pt = pts + (size_t)idx * a.stride;
      o[0] = pt[0]; o[1] = pt[1]; o[2] = pt[2]; o[3] = pt[3];
      if (a.depth) o[4] = a.depth[idx];
      else if (a.winner == 0) o[4] = __uint_as_float((unsigned)(key >> 32));
      else {
#pragma clang fp contract(off)
        o[4] = sqrtf((pt[0] * pt[0] + pt[1] * pt[1]) + pt[2] * pt[2]);
      }
      o[6] = o[4] > 0.0f ? 1.0f : 0.0f;
    }
    if (a.out_c > 5) {
      int l = (hit && a.labels) ? a.labels[idx] : 0;   // empty pixels carry label 0 before the map
      if (a.lut) l = (l >= 0 && l < a.lut_size) ? a.lut[l] : -1;
      o[5] = (float)l;
    }
    for (int c = 0; c < a.out_c; ++c) image[(size_t)i * a.out_c + c] = o[c];
    if (proj_idx) proj_idx[i] = idx;
  }
}

}  // namespace pclseg
