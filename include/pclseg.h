/*
 * pclseg.h — C ABI of the MI355X-native forward-pass engine (libpclseg.so).
 *
 * The reference has no FFI layer: its operator API for this path is the Keras call
 *     probabilities, predictions = model([lidar, mask])
 * (reference: pcl_segmentation/inference.py:75, eval.py:47, utils/callbacks.py:53) on a
 * model built by load_model_config (utils/args_loader.py:52-55) or loaded with
 * tf.keras.models.load_model (inference.py:39).  The entry points below are what a
 * ctypes binding for that call needs; each cites the reference lines it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions cross the boundary: every entry point is a
 *     function-try-block; std::bad_alloc / std::length_error become PCLSEG_ERR_OOM, anything else
 *     PCLSEG_ERR_INTERNAL (tests/test_host.py::test_allocation_failure_is_a_status_not_an_abort).
 *   - Every function returns int: 0 = PCLSEG_OK, negative = pclseg_status.  The text of the
 *     last failure is available from pclseg_last_error().
 *   - The caller owns every buffer it passes.  `mem` says where the caller's buffers live:
 *     PCLSEG_MEM_HOST (the library copies over PCIe itself) or PCLSEG_MEM_DEVICE (pointers
 *     are HIP device pointers on the handle's device; nothing is copied) or
 *     PCLSEG_MEM_HOST_ASYNC (page-locked host buffers, asynchronous).
 *   - One handle = one device + one stream; a handle is not thread-safe, independent
 *     handles are.  Forward calls are asynchronous on the handle's stream in
 *     PCLSEG_MEM_DEVICE mode (call pclseg_sync or synchronise the stream) and synchronous on
 *     return in PCLSEG_MEM_HOST mode.  Host-mode INPUTS (PCLSEG_MEM_HOST / _HOST_ASYNC) must be
 *     complete on the CPU side when the call is made: the uploads run on the engine's own copy
 *     streams and lanes and are NOT ordered behind work queued on the handle's stream (only
 *     device-memory inputs are).
 *   - All activations are float32 NHWC; predictions are int32.
 */
#ifndef PCLSEG_H_
#define PCLSEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCLSEG_VERSION 300 /* major*10000 + minor*100 + patch */

typedef struct pclseg_handle pclseg_handle;

typedef enum pclseg_status {
  PCLSEG_OK = 0,
  PCLSEG_ERR_BAD_ARG = -1,        /* null pointer, unknown enum, n <= 0 ...                */
  PCLSEG_ERR_BAD_SHAPE = -2,      /* W % 16 != 0, tensor shape mismatch (Keras: ValueError) */
  PCLSEG_ERR_MISSING_WEIGHT = -3, /* finalize with an unset tensor / unknown Keras path    */
  PCLSEG_ERR_HIP = -4,            /* HIP runtime error (incl. "no GPU")                     */
  PCLSEG_ERR_OOM = -5,
  PCLSEG_ERR_STATE = -6,          /* forward before finalize, set_weight after finalize     */
  PCLSEG_ERR_RANGE = -7,          /* split-f16 mode: an activation reached |v| >= 65504 (f16 range);
                                     the outputs of the offending call are not valid            */
  PCLSEG_ERR_INTERNAL = -8        /* a C++ exception other than an allocation failure was stopped at the
                                     boundary (allocation failures are PCLSEG_ERR_OOM); message in
                                     pclseg_last_error.  Nothing ever propagates into the caller.   */
} pclseg_status;

/* model_map keys of the reference (utils/args_loader.py:36-40); darknet21/53 differ only in
 * NUM_LAYERS (nets/Darknet.py:142-145). */
typedef enum pclseg_arch {
  PCLSEG_ARCH_SQUEEZESEGV2 = 0,
  PCLSEG_ARCH_DARKNET21 = 1,
  PCLSEG_ARCH_DARKNET53 = 2
} pclseg_arch;

typedef enum pclseg_mem {
  PCLSEG_MEM_HOST = 0,       /* host buffers (pinned or pageable); the call returns when the outputs are there */
  PCLSEG_MEM_DEVICE = 1,     /* device buffers; asynchronous on the handle's stream */
  PCLSEG_MEM_HOST_ASYNC = 2  /* PAGE-LOCKED host buffers; the call only enqueues (uploads, kernels, downloads),
                                pclseg_sync waits: consecutive calls overlap, the GPU never idles while the
                                CPU enqueues the next batch */
} pclseg_mem;

/* desc.flags */
#define PCLSEG_FLAG_KEEP_ACTIVATIONS 1u /* debug: no workspace aliasing, so every
                                           intermediate can be read back after a forward */
#define PCLSEG_FLAG_EXACT_F32 2u        /* run every convolution on the f32-input matrix cores
                                           (bit-exact float32 products) instead of the default
                                           split-f16 products (see pclseg_math) */

#define PCLSEG_FLAG_RANGE_FALLBACK 4u   /* split-f16 mode: keep the exact-f32 weight fragments resident
                                           too and, when the range guard fires, transparently re-run
                                           the offending call with exact float32 products instead of
                                           returning PCLSEG_ERR_RANGE */

/* Arithmetic of the convolutions.  Activations, accumulation and all outputs are float32 in
 * both modes.
 *   PCLSEG_MATH_F16X3: each float32 operand v is split hi = f16(v), lo = f16(v - hi) and a
 *     product is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16.  The pair carries 22 significant
 *     bits while lo is a normal half, i.e. for |v| >= 2^-3; below that the ABSOLUTE error is 2^-25.
 *     Weights are therefore pre-scaled by a power of two per tile of 16 output channels (largest |w|
 *     of the tile -> [2^12, 2^13), undone exactly in the float32 epilogue), so every weight down to 2^-15
 *     of its tile's maximum keeps 22 bits; activations are split as they are (O(1) values after
 *     BatchNorm: absolute error 2^-25 where |v| < 1/8).  Measured: logits within ~1e-5 of the float64
 *     oracle on all three networks (bench.py `parity_check`).  Default.  Requires |activation| < 65504: every kernel that
 *     splits a value checks it, a violation sets a sticky flag on the device and the call that
 *     observes it (a PCLSEG_MEM_HOST forward, or pclseg_sync after PCLSEG_MEM_DEVICE forwards)
 *     returns PCLSEG_ERR_RANGE — or repairs the call, see PCLSEG_FLAG_RANGE_FALLBACK.
 *   PCLSEG_MATH_F32: v_mfma_f32_16x16x4_f32, exact float32 products. */
typedef enum pclseg_math { PCLSEG_MATH_F16X3 = 0, PCLSEG_MATH_F32 = 1 } pclseg_math;

/* Everything the forward pass reads from a reference config (configs/ *.py: ZENITH_LEVEL,
 * AZIMUTH_LEVEL, NUM_CLASS, CLASSES.index("None"), INPUT_MEAN, INPUT_STD, OUTPUT_STRIDE). */
typedef struct pclseg_desc {
  int32_t arch;          /* pclseg_arch */
  int32_t height;        /* ZENITH_LEVEL  */
  int32_t width;         /* AZIMUTH_LEVEL; must be divisible by 16 (four stride-2 stages) */
  int32_t num_class;     /* NUM_CLASS */
  int32_t none_index;    /* CLASSES.index("None"): prediction written where mask is false */
  int32_t output_stride; /* Darknet OUTPUT_STRIDE (only 16 is used by the reference configs;
                            8/16/32 accepted); ignored for SqueezeSegV2 */
  int32_t device;        /* HIP device ordinal */
  int32_t micro_batch;   /* scans swept through the network per kernel sequence; 0 = auto.
                            Any n may be passed to forward; it is processed in chunks. */
  uint32_t flags;
  double mean[5];        /* INPUT_MEAN (x, y, z, intensity, depth) */
  double std[5];         /* INPUT_STD */
} pclseg_desc;

/* CPU-only description of what a desc builds (no GPU needed). */
typedef struct pclseg_plan_info {
  int32_t num_ops;            /* kernel launches per micro-batch */
  int32_t num_weights;        /* Keras tensors the model owns */
  int32_t num_tensors;        /* activation tensors */
  int32_t micro_batch;        /* resolved micro-batch */
  int64_t num_params;         /* scalars in all Keras tensors */
  int64_t alg_macs_per_scan;  /* multiply-accumulates of every Conv2D / Conv2DTranspose */
  int64_t alg_bytes_per_scan; /* module-granular fp32 traffic, no weights */
  int64_t workspace_bytes;    /* activation arena for one micro-batch */
  int64_t packed_weight_bytes;
} pclseg_plan_info;

int pclseg_version(void);
/* First 16 hex digits of the sha256 over the sources this binary was built from (csrc/pclseg_kernels.h,
 * pclseg_graph.h, pclseg_api.hip, include/pclseg.h, concatenated in that order; "unknown" when built outside
 * the Makefile).  bench.py compares it with the sources beside it: a profile or traffic figure is only
 * attributed to a binary whose hash matches. */
const char* pclseg_build_sha(void);

/* Last error text of `h` (or of the last failed handle-less call when h is NULL).
 * Never returns NULL. */
const char* pclseg_last_error(const pclseg_handle* h);

/* Validate a desc and describe the graph it builds.  Runs without a GPU. */
int pclseg_plan(const pclseg_desc* desc, pclseg_plan_info* out);

/* Names of the kernel launches a desc plans, in launch order, one per line ("conv1", "cam1",
 * "pool+fire2/squeeze", "fire4/expand+fire5/squeeze", "enc3/residual_1/conv2", ...), each followed by a TAB and
 * the multiply-accumulates per scan of that launch, then TAB-separated the static launch resources of the
 * split-f16 plan — dynamic LDS bytes per block, threads per block, blocks per scan: labels, FLOP counts and
 * co-residency limits for per-operator profiles (the pre-processing launch that precedes them is not in the
 * list).  Runs without a GPU. */
int pclseg_plan_ops(const pclseg_desc* desc, char* buf, size_t cap);

/* Build the model graph on desc->device: replaces constructing the Keras model
 * (SqueezeSegV2(mc) nets/SqueezeSegV2.py:220-283; Darknet(mc) nets/Darknet.py:149-260).
 * Fails with PCLSEG_ERR_HIP when no GPU is present: there is no CPU fallback. */
int pclseg_create(const pclseg_desc* desc, pclseg_handle** out);
int pclseg_destroy(pclseg_handle* h);

/* Weight inventory, in layer-construction order, named by Keras attribute path
 * ("fire2/squeeze/kernel", "enc3/residual_1/bn2/moving_variance", ...). */
int pclseg_num_weights(const pclseg_handle* h);
int pclseg_weight_info(const pclseg_handle* h, int index, char* name, size_t name_cap,
                       int64_t shape[4], int* ndim);

/* Bind one Keras tensor (host pointer, float32, Keras layout: conv (kh,kw,Cin,Cout),
 * transposed conv (1,4,Cout,Cin), vectors (C,)).  Replaces load_model's variable restore
 * (inference.py:39).  The data is copied. */
int pclseg_set_weight(pclseg_handle* h, const char* keras_path, const float* data,
                      const int64_t* shape, int ndim);

/* Fold BatchNorm (eps = 1e-3, inference form) into the conv weights, repack for the MFMA
 * kernels and upload.  Must be called once, after every tensor has been set. */
int pclseg_finalize(pclseg_handle* h);

/* Multi-GPU start-up (one process per GPU, the scans of a batch sharded over the ranks; the reference has
 * no device story beyond inference.py:116-118).  After pclseg_finalize the folded, packed parameters are
 * three device arrays; export copies them (behind a small header) into ONE caller buffer — host or device,
 * `mem` = PCLSEG_MEM_HOST / PCLSEG_MEM_DEVICE — which rank 0 broadcasts once (RCCL over xGMI); the other
 * ranks call import on a freshly created handle of the SAME desc INSTEAD of set_weight + finalize: one
 * device copy, no BatchNorm folding or repacking on their host cores.  import fails with
 * PCLSEG_ERR_BAD_SHAPE when the blob was made for a different desc, arithmetic mode or library version. */
int pclseg_packed_size(const pclseg_handle* h, size_t* bytes);
int pclseg_export_packed(pclseg_handle* h, void* dst, size_t capacity, int mem);
int pclseg_import_packed(pclseg_handle* h, const void* src, size_t bytes, int mem);

/* Use `hip_stream` (a hipStream_t) for all subsequent work; NULL = the legacy default stream. */
int pclseg_set_stream(pclseg_handle* h, void* hip_stream);
/* Wait for the handle's stream.  Returns PCLSEG_ERR_RANGE if a split-f16 kernel since the last
 * check saw an out-of-range activation: the flag is one sticky word, so EVERY asynchronous call
 * enqueued since the previous check is suspect.  With PCLSEG_FLAG_RANGE_FALLBACK the handle remembers
 * those calls (up to 256) and re-runs all of them, oldest first, in exact float32 and returns
 * PCLSEG_OK.  A PCLSEG_MEM_HOST forward that observes a flag raised by earlier asynchronous calls does the
 * same (or says so in its PCLSEG_ERR_RANGE message).
 * COMPLETION RULE for PCLSEG_FLAG_RANGE_FALLBACK handles: pclseg_sync (or a PCLSEG_MEM_HOST forward) is the
 * ONLY point at which an asynchronous call is complete.  The repair re-reads the caller's INPUT pointers and
 * re-writes its OUTPUT pointers, so until that point every buffer of every un-synced call must stay allocated
 * and its inputs unmodified — synchronising the HIP stream yourself does NOT release them (without the flag
 * it does: nothing is replayed).  More than 256 un-synced calls: a range overflow is reported as
 * PCLSEG_ERR_RANGE instead of repaired. */
int pclseg_sync(pclseg_handle* h);

/* Page-locked host memory for the PCLSEG_MEM_HOST boundary (the reference hands NumPy arrays to
 * Keras, inference.py:71-75).  Buffers from here (or any hipHostMalloc / hipHostRegister /
 * torch pin_memory buffer) are read and written by the GPU directly (DMA uploads that run ahead of
 * the kernels, or copy kernels / direct prediction writes over PCIe where a DMA command would stall the
 * calling thread) and overlapped with compute; pageable buffers are accepted too and go through the
 * library's own pinned bounce buffers. */
void* pclseg_host_alloc(size_t bytes);
int pclseg_host_free(void* p);

/* The reference-shaped call: model([lidar, mask]) -> (probabilities, predictions)
 * (nets/SegmentationNetwork.py:55-69, nets/SqueezeSegV2.py:285-325, nets/Darknet.py:279-314).
 *   lidar  float32 [n,H,W,6]  normalised, channel 5 = mask as 0/1
 *   mask   uint8   [n,H,W]    0 = no point
 *   preds  int32   [n,H,W]    out, required
 *   probs  float32 [n,H,W,NC] out, optional (NULL: softmax is not materialised and the
 *                             argmax is taken over the logits, which is the same argmax)
 *   logits float32 [n,H,W,NC] out, optional */
int pclseg_forward(pclseg_handle* h, const float* lidar, const uint8_t* mask, int n,
                   int32_t* preds, float* probs, float* logits, int mem);

/* Same, starting from raw scans: does the caller-side pre-processing of the reference on the
 * device (inference.py:50-62; data_loader/data_loader.py:156-171): mask = depth > 0,
 * (x - mean)/std in float64, invalid pixels zeroed, mask appended as 6th channel.
 *   scans    float32 [n,H,W,5]  x, y, z, intensity, depth
 *   mask_out uint8   [n,H,W]    out, optional */
int pclseg_forward_raw(pclseg_handle* h, const float* scans, int n, int32_t* preds,
                       float* probs, float* logits, uint8_t* mask_out, int mem);

/* Debug: activation tensors of the LAST micro-batch (meaningful with
 * PCLSEG_FLAG_KEEP_ACTIVATIONS).  shape = {micro-batch scans held, H, W, C}. */
int pclseg_num_tensors(const pclseg_handle* h);
int pclseg_tensor_info(const pclseg_handle* h, int index, char* name, size_t name_cap,
                       int64_t shape[4]);
int pclseg_read_tensor(pclseg_handle* h, int index, float* host_out, size_t capacity_floats);

/* ---- single-operator entry points (device pointers, handle-less, default stream, synchronous).
 * They run the same kernels the graph uses and exist for operator-level parity tests. */

/* inference.py:50-62.  scans [n,H,W,5] -> lidar [n,H,W,6], mask [n,H,W]. */
int pclseg_op_normalize(const float* scans, int n, int h, int w, const double mean[5],
                        const double std[5], float* lidar6, uint8_t* mask);

/* Conv2D SAME, strides (1, stride_w), kernel (kh,kw,Cin,Cout) host pointer, optional bias and
 * BatchNorm (host pointers, NULL = absent), activation 0 none / 1 relu / 2 leaky(0.1) /
 * 3 sigmoid, optional residual added after the activation.  x, residual, y: device, NHWC;
 * Cin and Cout multiples of 4.  math: pclseg_math. */
int pclseg_op_conv2d(const float* x, int n, int h, int w, int cin, const float* kernel, int kh,
                     int kw, int cout, int stride_w, const float* bias, const float* bn_gamma,
                     const float* bn_beta, const float* bn_mean, const float* bn_var, int act,
                     const float* residual, float* y, int math);

/* Conv2DTranspose kernel (1,4), strides (1,2), SAME: kernel (1,4,Cout,Cin) host pointer.
 * y [n,h,2w,Cout]. */
int pclseg_op_conv2d_transpose(const float* x, int n, int h, int w, int cin, const float* kernel,
                               int cout, const float* bias, const float* bn_gamma,
                               const float* bn_beta, const float* bn_mean, const float* bn_var,
                               int act, float* y, int math);

/* MaxPool k x k, strides (1, stride_w), SAME (padding never wins). C multiple of 4. */
int pclseg_op_max_pool(const float* x, int n, int h, int w, int c, int k, int stride_w, float* y);

/* conv 3x3 Cin->NC + bias, then nets/SegmentationNetwork.py:58-69.  kernel/bias host pointers;
 * preds required; probs/logits optional device pointers. */
int pclseg_op_head(const float* x, const uint8_t* mask, int n, int h, int w, int cin,
                   const float* kernel, const float* bias, int num_class, int none_index,
                   int32_t* preds, float* probs, float* logits, int math);

/* CPU only (no GPU needed): what the split-f16 matrix-core path multiplies by.  Packs a Keras
 * Conv2D kernel (kh,kw,Cin,Cout; 1x1 or 3x3) exactly as pclseg_finalize does in PCLSEG_MATH_F16X3 mode —
 * per tile of 16 output channels a power-of-two scale 2^k that puts the tile's largest |w| into
 * [2^12, 2^13), then hi = f16(w 2^k), lo = f16(w 2^k - hi) — and writes, in the same Keras layout, the value
 * the fragments represent: recon = (hi + lo) 2^-k (float64, exact).  exponents (optional, [Cout]) receives k.
 * Reference arithmetic this stands in for: TensorFlow float32 Conv2D (requirements.txt:1). */
int pclseg_op_split_f16_roundtrip(const float* kernel, int kh, int kw, int cin, int cout, double* recon,
                                  int32_t* exponents);

/* Evaluation metrics (the row after the forward pass: eval.py:41-58, utils/util.py:64-79,
 * tf.metrics.MeanIoU as used in nets/SegmentationNetwork.py:52).  Accumulates the confusion
 * matrix cm[label][pred] += 1 over `count` pixels into a device int64 [num_class, num_class]
 * buffer (not cleared: call repeatedly to accumulate over a dataset, like update_state).
 * labels / preds / cm: device pointers; entries outside [0, num_class) are ignored.
 * Asynchronous on `hip_stream` (NULL = default stream). */
int pclseg_op_confusion_matrix(const int32_t* labels, const int32_t* preds, size_t count,
                               int num_class, int64_t* cm, void* hip_stream);

/* Spherical projection, the step before the network (dataset_convert/laserscan_semantic_kitti.py:
 * 106-166, LaserScan.do_range_projection; used by dataset_convert/semantic_kitti.py:150-179).
 *   points   float32 [m,4]   x, y, z, remission                          (device)
 *   image5   float32 [H,W,5] x, y, z, remission, depth of the NEAREST point of each pixel;
 *                            pixels without a point hold `empty` (-1 like LaserScan's attributes,
 *                            0 like the .npy files the converter writes)     (device, out)
 *   proj_idx int32   [H,W]   index of the winning point, -1 = none          (device, out, optional)
 *   scratch  uint64  [H*W]   work space                                     (device)
 * fov_up / fov_down in degrees.  Asynchronous on `hip_stream`. */
int pclseg_op_project(const float* points, size_t m, int h, int w, float fov_up, float fov_down,
                      float empty, float* image5, int32_t* proj_idx, uint64_t* scratch,
                      void* hip_stream);

/* The other projection variants of the reference's converters, same scatter / gather pair:
 *   row_mode  PCLSEG_PROJ_ROW_FOV  row from the elevation angle (fov_up / fov_down, degrees)
 *             PCLSEG_PROJ_ROW_RING row = H-1-ring_index, laserscan_nuscenes.py:191-223
 *                                  (do_range_projection_ring), convert_validation_pcd_to_npy.py:147-153
 *   col_mode  PCLSEG_PROJ_COL_FULL  360 degree azimuth, laserscan_*.py
 *             PCLSEG_PROJ_COL_FRONT column = (int)((left_phi - atan2(y,x)) / ((right_phi+left_phi)/W)),
 *                                  points outside the window dropped (convert_validation_pcd_to_npy.py:
 *                                  120-137; radians)
 *   winner    PCLSEG_PROJ_NEAREST  nearest point of a pixel wins (the reference's depth-sorted scatter)
 *             PCLSEG_PROJ_LAST     last point in input order wins (plain fancy-index assignment)
 *   out_channels 5: x,y,z,remission,depth; 6: + label (SemLaserScan.do_label_projection and the
 *             converters' learning_map, dataset_convert/semantic_kitti.py:150-179, nu_dataset.py:157-167;
 *             empty pixels carry label_lut[0], or 0 without a table); 7: + mask = depth > 0.
 *   points    float32 [m, point_stride], x,y,z at 0..2, remission at 3; ring int32 [m] (ring rows);
 *   depth     optional float32 [m] (NULL: float32 norm of xyz); labels optional int32 [m];
 *   label_lut optional int32 [lut_size]: label -> train id (out of range -> -1).
 *   image float32 [H,W,out_channels], proj_idx int32 [H,W] (optional), scratch uint64 [H*W]: device. */
enum { PCLSEG_PROJ_ROW_FOV = 0, PCLSEG_PROJ_ROW_RING = 1 };
enum { PCLSEG_PROJ_COL_FULL = 0, PCLSEG_PROJ_COL_FRONT = 1 };
enum { PCLSEG_PROJ_NEAREST = 0, PCLSEG_PROJ_LAST = 1 };
typedef struct pclseg_proj_desc {
  int32_t h, w;
  int32_t row_mode, col_mode, winner;
  int32_t out_channels;
  float fov_up, fov_down; /* degrees, PCLSEG_PROJ_ROW_FOV */
  float empty;            /* value of x,y,z,remission,depth where no point landed */
  double left_phi, right_phi; /* radians, PCLSEG_PROJ_COL_FRONT */
} pclseg_proj_desc;
int pclseg_op_project_ex(const pclseg_proj_desc* desc, const float* points, int point_stride, size_t m,
                         const int32_t* ring, const float* depth, const int32_t* labels,
                         const int32_t* label_lut, int lut_size, float* image, int32_t* proj_idx,
                         uint64_t* scratch, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PCLSEG_H_ */
