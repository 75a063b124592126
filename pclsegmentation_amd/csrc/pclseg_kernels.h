// pclseg_kernels.h — CDNA4 (gfx950) device kernels of the forward pass.
//
// Layout: every activation is float32 NHWC with C a multiple of 4, so one lane moves one
// 16-byte channel quad and a run of pixels along W is one contiguous span in HBM.
//
// conv_mfma_kernel is the workhorse: an implicit-GEMM convolution on the f32-input matrix
// cores (v_mfma_f32_16x16x4_f32 — exact float32, bit-identical to an fmaf chain), with the
// input halo patch staged once per channel chunk in LDS and the (BatchNorm-folded) weights
// streamed from L2 in a pre-packed per-lane fragment order.  It covers
//   Conv2D 3x3 / 1x1, strides (1,1) and (1,2), TF "SAME" padding      (SURVEY.md K2,K3,K4)
//   Conv2DTranspose (1,4)/(1,2) as two 2-tap convs, one per output parity      (K5)
//   fused epilogues: +bias(BN) -> relu / leaky(0.1) / sigmoid -> (*gate | +residual) -> +skip,
//   writing into a channel slice of the destination (tf.concat for free)         (K8)
//   and the segmentation head: 3x3 conv -> [softmax] -> argmax -> mask            (K9)
//
// GEMM orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel].  With the 16x16x4 lane
// maps (A: row = lane&15, k = lane>>4; B: k = lane>>4, col = lane&15; D: col = lane&15,
// row = 4*(lane>>4)+reg) a lane ends up holding 4 CONSECUTIVE output channels of ONE pixel, so
// the epilogue is 16-byte loads/stores along the channel axis.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pclseg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_SIGMOID = 3 };

constexpr int kConvThreads = 256;  // 4 waves, one per SIMD
constexpr int kMT = 2;             // 16-pixel segments per wave
constexpr int kSegsPerBlock = 4 * kMT;
constexpr int kChunk = 32;         // input channels staged per LDS pass

struct ConvArgs {
  const float* in;    // [N,H,Win,Cin]
  const float* wpk;   // packed weights [tap][c16][ct][lane][4]
  const float* bias;  // [nctp*16]
  float* out;         // [N,H,Wout,out_C], written at channel offset co_off
  const float* res1;  // optional, [N,H,Wout,res1_C], read at co_off + co
  const float* res2;  // optional
  const uint8_t* mask;  // head only
  int32_t* preds;       // head only
  float* probs;         // head only, optional
  float* logits;        // head only, optional
  int N, H, Win, Wout, Wconv;
  int Cin, nc16, Cout, nctp;
  int out_C, co_off, res1_C, res2_C;
  int KH, KW, sw, pt, pl, ow_mul, ow_off;
  int TH, SEGW, PH, PW, tilesH, tilesW;
  int act, res1_mul, none_index;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_LRELU: return v > 0.0f ? v : 0.1f * v;
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <int NT, bool HEAD>
__global__ __launch_bounds__(kConvThreads) void conv_mfma_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15;  // pixel within the 16-pixel segment
  const int g = lane >> 4;  // k-group (operands) / cout quad (accumulator)

  int tile = blockIdx.x;
  const int twi = tile % a.tilesW;
  tile /= a.tilesW;
  const int thi = tile % a.tilesH;
  const int n = tile / a.tilesH;
  const int h0 = thi * a.TH;
  const int w0 = twi * (a.SEGW * 16);
  const int ct0 = blockIdx.y * NT;

  f32x4 acc[kMT][NT];
#pragma unroll
  for (int m = 0; m < kMT; ++m)
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int cinp = a.nc16 * 16;
  const int CS = (cinp < kChunk ? cinp : kChunk) + 4;  // floats per patch pixel (+4: bank spread)
  const float* in_n = a.in + (size_t)n * a.H * a.Win * a.Cin;
  const int ntaps = a.KH * a.KW;

  // this wave's two segments: (row, column-segment) inside the tile
  int seg_r[kMT], seg_q[kMT];
#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int seg = wave * kMT + m;
    seg_r[m] = seg / a.SEGW;
    seg_q[m] = seg - seg_r[m] * a.SEGW;
  }

  for (int c0 = 0; c0 < cinp; c0 += kChunk) {
    const int ckp = (cinp - c0) < kChunk ? (cinp - c0) : kChunk;
    const int cqn = ckp >> 2;
    if (c0) __syncthreads();
    // ---- stage the zero-padded input patch [PH][PW][ckp] into LDS
    const int total = a.PH * a.PW * cqn;
    for (int idx = tid; idx < total; idx += kConvThreads) {
      const int cq = idx % cqn;
      const int pix = idx / cqn;
      const int pc = pix % a.PW;
      const int pr = pix / a.PW;
      const int h = h0 - a.pt + pr;
      const int w = w0 * a.sw - a.pl + pc;
      const int c = c0 + cq * 4;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (h >= 0 && h < a.H && w >= 0 && w < a.Win && c < a.Cin)
        v = *reinterpret_cast<const f32x4*>(in_n + ((size_t)h * a.Win + w) * a.Cin + c);
      *reinterpret_cast<f32x4*>(smem + (pr * a.PW + pc) * CS + cq * 4) = v;
    }
    __syncthreads();

    // ---- matrix-core loop over taps x 16-channel sub-chunks
    const int nsub = ckp >> 4;
    for (int t = 0; t < ntaps; ++t) {
      const int th = t / a.KW;
      const int tw = t - th * a.KW;
      for (int sub = 0; sub < nsub; ++sub) {
        const int c16 = (c0 >> 4) + sub;
        const float* wb = a.wpk + ((size_t)(t * a.nc16 + c16) * a.nctp + ct0) * 256 + lane * 4;
        f32x4 wv[NT];
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) wv[nn] = *reinterpret_cast<const f32x4*>(wb + nn * 256);
        f32x4 xv[kMT];
#pragma unroll
        for (int m = 0; m < kMT; ++m) {
          const int pcol = (seg_q[m] * 16 + p) * a.sw + tw;
          const int prow = seg_r[m] + th;
          xv[m] = *reinterpret_cast<const f32x4*>(smem + (prow * a.PW + pcol) * CS + sub * 16 + g * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int m = 0; m < kMT; ++m)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
              acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nn][j], xv[m][j], acc[m][nn], 0, 0, 0);
      }
    }
  }

  // ---- epilogue
  f32x4 bv[NT];
#pragma unroll
  for (int nn = 0; nn < NT; ++nn)
    bv[nn] = *reinterpret_cast<const f32x4*>(a.bias + (ct0 + nn) * 16 + g * 4);

#pragma unroll
  for (int m = 0; m < kMT; ++m) {
    const int oh = h0 + seg_r[m];
    const int j = w0 + seg_q[m] * 16 + p;
    const bool valid = (oh < a.H) && (j < a.Wconv);
    const int ow = j * a.ow_mul + a.ow_off;
    const size_t pix = ((size_t)n * a.H + oh) * a.Wout + ow;

    if constexpr (!HEAD) {
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) {
        const int co = (ct0 + nn) * 16 + g * 4;
        if (valid && co < a.Cout) {
          f32x4 v = acc[m][nn] + bv[nn];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = apply_act(v[i], a.act);
          if (a.res1) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(a.res1 + pix * a.res1_C + a.co_off + co);
            v = a.res1_mul ? v * r : v + r;
          }
          if (a.res2) v += *reinterpret_cast<const f32x4*>(a.res2 + pix * a.res2_C + a.co_off + co);
          *reinterpret_cast<f32x4*>(a.out + pix * a.out_C + a.co_off + co) = v;
        }
      }
    } else {
      // segmentation head (reference: nets/SegmentationNetwork.py:58-69).  The NT tiles hold
      // all NUM_CLASS logits of a pixel across the 4 lanes {p, p+16, p+32, p+48}.
      const int NC = a.Cout;
      float val[NT * 4];
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int nn = 0; nn < NT; ++nn)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = nn * 16 + g * 4 + i;
          const float v = acc[m][nn][i] + bv[nn][i];
          val[nn * 4 + i] = v;
          if (co < NC) {
            if (a.logits && valid) a.logits[pix * NC + co] = v;
            if (v > best) { best = v; bi = co; }
          }
        }
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (a.probs) {
        // softmax materialised: exp(x - max) / sum, and the argmax is taken over the
        // probabilities exactly as the reference does (lowest index wins ties).
        float s = 0.f;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (nn * 16 + g * 4 + i < NC) s += expf(val[nn * 4 + i] - best);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        float pbest = -1.f;
        int pbi = 0x7fffffff;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = nn * 16 + g * 4 + i;
            if (co < NC) {
              const float pr = expf(val[nn * 4 + i] - best) / s;
              if (valid) a.probs[pix * NC + co] = pr;
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
      if (g == 0 && valid) a.preds[pix] = a.mask[pix] ? bi : a.none_index;
    }
  }
}

// ---- MaxPool k x k, strides (1, sw), TF SAME (padding never wins)            (K6, K7)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ in,
                                                      float* __restrict__ out, int N, int H, int Win,
                                                      int Wout, int C, int k, int sw, int pt, int pl) {
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
    for (int i = 0; i < k; ++i) {
      const int hh = h - pt + i;
      if (hh < 0 || hh >= H) continue;
      const float* row = in + ((size_t)n * H + hh) * Win * C + c4 * 4;
      for (int j = 0; j < k; ++j) {
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

}  // namespace pclseg
