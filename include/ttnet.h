/*
 * ttnet.h -- C ABI of the MI355X-native TTNet inference path (libttnet.so).
 *
 * This is the drop-in boundary for ONE path of Anonymousijcai2024ttnet/scale_imagenet: the
 * eval-mode forward() of the TTNet ImageNet classifiers.  The reference has no FFI (it is
 * 100 % PyTorch), so each entry point below names the reference interface it replaces
 * (file:line under the upstream repository) and INTEGRATION.md shows the ctypes stub a
 * maintainer adds to the reference's nn.Module to call it.
 *
 * Conventions
 *   - plain C types only; device pointers are raw HIP device addresses; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream).
 *   - every function returns 0 on success or a negative ttnet_status; the message for the
 *     last failure on the calling thread is ttnet_last_error().
 *   - a plan is bound to one device; calls on one plan are serialised by the caller;
 *     different plans may run concurrently on different streams.
 *   - device memory is allocated only inside ttnet_plan_create / _finalize / _set_lanes
 *     (weights, truth tables, one activation workspace for `max_batch` images per lane);
 *     ttnet_forward never synchronises.  It instantiates a hipGraph of its own the third time
 *     a batch size is seen (and replays it afterwards); when the caller's stream is itself
 *     capturing, it records plain launches into the caller's graph instead.
 *   - there is no CPU fallback: without a HIP device every compute entry point fails.
 *
 * Packed activation layouts (all little-endian, LSB first)
 *   rows  ("RP"): uint64 [N][C][H]      bit x of word (n,c,y) is pixel (y,x); W <= 60
 *   chans ("CP"): uint16 [N][C/16][H][W] bit k of word (n,q,y,x) is channel 16q+k
 */
#ifndef TTNET_H
#define TTNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ttnet_plan ttnet_plan;
typedef struct ttnet_comm ttnet_comm;

typedef enum ttnet_status {
  TTNET_OK = 0,
  TTNET_E_INVALID = -1,      /* bad argument (NULL, shape, unknown key, batch > max_batch) */
  TTNET_E_STATE = -2,        /* call out of order (forward before finalize, missing tensor) */
  TTNET_E_HIP = -3,          /* a HIP runtime call failed; message carries hipGetErrorString */
  TTNET_E_UNSUPPORTED = -4,  /* geometry this build has no kernel for */
  TTNET_E_NOMEM = -5,
  TTNET_E_RANGE = -6         /* an earlier forward met a value outside the fp16 x 2 operand split (see ttnet_forward) */
} ttnet_status;

typedef enum ttnet_dtype { TTNET_F32 = 0, TTNET_I64 = 1, TTNET_U8 = 2, TTNET_U16 = 3, TTNET_U64 = 4 } ttnet_dtype;

typedef enum ttnet_variant {
  TTNET_SMALL = 0,   /* models/TT_general_imagenet_v2_small.py:151  TT_vf_19lv3_imgnet_small  */
  TTNET_XSMALL = 1,  /* models/TT_general_imagenet_v2_xsmall.py:151 TT_vf_19lv3_imgnet_xsmall */
  TTNET_FULL = 2,    /* models/TT_general_imagenet_v2.py:139        TT_vf_19lv3_imgnet        */
  TTNET_VALEXNET = 3 /* models/TT_FHE_XSMALL_vAlexnet.py:585        TT_FHE_XSMALL_vAlexnet (CIFAR 32x32;
                        nfilter / tfilter / layers are ignored, as the reference's constructor ignores them) */
} ttnet_variant;

/* The constructor arguments of the reference model (args.nfilter / tfilter / layers,
 * models/TT_general_imagenet_v2_small.py:154-181; main.py:47-50) plus what the reference
 * discovers with a dry run on torch.rand(1,3,224,224) (:199-207): the input size.
 *
 * Which (variant, p = nfilter * tfilter, layers) the reference constructs and this library builds
 * (everything else: ttnet_plan_create returns TTNET_E_UNSUPPORTED with the reason in ttnet_last_error):
 *   TTNET_SMALL    reference: any p for which nn.Conv2d accepts groups = int(C / 16) for C = p, 2p, 4p .. and 4C
 *                  (:28-76; e.g. every multiple of 16, but also p = 40 with a fan-in of 20), layers 0..4.
 *                  built: p in {16, 32, .., 128} (fan-in 16: the truth-table kernels; the stem kernel holds one, two or
 *                  four 32-channel M-tiles), layers 0..2; layers 3 / 4 (stride-1 blocks) at p = 64.  Not built: p > 128,
 *                  p % 16 != 0.
 *   TTNET_XSMALL   reference: any p with p % 4 == 0 ..., layers 0..4.  built: p in {16, 32, .., 128} (its depthwise tables are
 *                  striped by 16 channels), layers 0..2 (the stride-1 first blocks of layers 3 / 4 exist for TTNET_SMALL only).
 *   TTNET_FULL     reference: p = 60 and the other p for which int(4C / 30) divides 4C (p = 64 does NOT construct,
 *                  SURVEY 2 #2), layers 0..3 (its --layers 2 falls through the reference's own branch-padding rules at the 9 x 9
 *                  input of the fourth block, TT_general_imagenet_v2.py:98-128).  built: those p <= 64, layers 0..1.
 *   TTNET_VALEXNET fixed geometry.
 * image_h = image_w = 224 (32 for TTNET_VALEXNET): the reference's branch-padding rules are keyed by the widths that
 * 224 x 224 produces (:98-139); other input sizes fall through them in the reference and are refused here. */
typedef struct ttnet_net_desc {
  int32_t variant;     /* ttnet_variant */
  int32_t nfilter;     /* main.py:47, default 8 */
  int32_t tfilter;     /* main.py:48, default 8 */
  int32_t layers;      /* main.py:50, default 1 */
  int32_t image_h;     /* 224 */
  int32_t image_w;     /* 224 */
  int32_t max_batch;   /* workspace is sized for this many images per forward */
  int32_t reserved;
} ttnet_net_desc;

/* Replaces TT_vf_19lv3_imgnet_small.__init__ / make_small_network
 * (models/TT_general_imagenet_v2_small.py:154-203): fixes the geometry, allocates device
 * storage for the 174 state tensors and the activation workspace. */
int ttnet_plan_create(const ttnet_net_desc *desc, int device, ttnet_plan **out);

/* Replaces nn.Module.load_state_dict (main.py:220-222): hands the plan one state_dict
 * entry under its reference key, e.g. "features.4.Block_conv1.conv1.weight".  A leading
 * "module." (DataParallel / DDP checkpoints, main.py:181-192) is accepted and stripped.
 * `ptr` may be a host or a device pointer (on_device != 0); the bytes are copied.
 * num_batches_tracked and grad_scale entries are accepted and ignored (eval mode never
 * reads them).  Unknown keys and shape / dtype mismatches are TTNET_E_INVALID, as strict
 * load_state_dict would raise. */
int ttnet_plan_set_tensor(ttnet_plan *plan, const char *key, const void *ptr,
                          const int64_t *shape, int ndim, int dtype, int on_device);

/* Derived state, rebuilt after the tensors change: folds every BatchNorm, enumerates every
 * binarised Block_TT into its truth table on the GPU (float64, exact erf; the enumeration
 * convention of Block_TT.get_TT_block_all_filter, models/TT_FHE_SMALL.py:322-342), builds
 * the float table of the last block, permutes lin1 to the feature order of the device
 * kernels.  Fails with TTNET_E_STATE if a required tensor was never set. */
int ttnet_plan_finalize(ttnet_plan *plan, void *stream);

/* Replaces SeqBinModelHelper.forward (models/model_utils/netbin.py:703-708), i.e.
 * `outputs = model(inputs)` at main.py:261 in eval mode under no_grad.
 *   x_dev      float32 [n,3,image_h,image_w] NCHW, contiguous, on the plan's device
 *   logits_dev float32 [n,n_classes]  (1000; 10 for TTNET_VALEXNET)
 * Asynchronous on `stream`.  From the third call with the same n the launches are replayed from a
 * hipGraph captured on a private stream (the input / logits pointers are patched per call);
 * TTNET_NO_GRAPH=1 in the environment keeps plain launches.  Results are identical either way.
 * Setting a tensor, a table or finalizing drops every captured graph (they are re-captured).
 * Range.  The float stages run on the 16-bit matrix cores with every float32 operand carried as two
 * fp16 terms after a power-of-two prescale (x16 for activations): inputs, pooled features and
 * classifier activations must satisfy |v| < 4094 (the reference's float32 path has no such limit;
 * normalised ImageNet inputs span [-2.2, 2.7]).  A value outside the range, or a NaN, raises a sticky
 * flag from inside the kernel; since the forward is asynchronous, it is the NEXT call on the plan
 * (forward, read_stage) that fails with TTNET_E_RANGE, and keeps failing until
 * ttnet_plan_query("range_overflow") has read and cleared the flag.
 * Alignment.  x_dev must be 16-byte aligned (the stem reads it with 16-byte buffer loads; any tensor
 * torch allocates is): TTNET_E_INVALID otherwise.  The uint8 input of ttnet_forward_u8 needs 4 bytes. */
int ttnet_forward(ttnet_plan *plan, const float *x_dev, int64_t n, float *logits_dev, void *stream);

/* Batches in flight.  A plan starts with one lane = one set of activation buffers; lanes share
 * the weights and truth tables.  ttnet_plan_set_lanes grows the plan to `lanes` sets (1..16),
 * and ttnet_forward_lane runs a forward on one of them: forwards on different lanes may be in
 * flight together on different streams (the ramp and tail of one batch's kernels are filled by
 * another's; this is how the eval loop of main.py:255-275 is pipelined, evaluate.py).  Calls on
 * one plan are still issued from one thread at a time; a lane must not be reused before the
 * forward issued on it has finished or been ordered before the new one by its stream.
 * ttnet_forward is lane 0; ttnet_read_stage reads the lane used last. */
int ttnet_plan_set_lanes(ttnet_plan *plan, int lanes);
int ttnet_forward_lane(ttnet_plan *plan, int lane, const float *x_dev, int64_t n, float *logits_dev, void *stream);

/* SURVEY 8(f) N1 -- the tail of the input pipeline fused into the stem.  x is the decoder's
 * uint8 image, HWC: uint8 [n][image_h][image_w][3]; the library applies ToTensor (/255) and
 * Normalize(mean, std) (utils/preprocess.py:104-108; the ImageNet constants by default,
 * ttnet_plan_set_input_norm replaces them) in front of the stem's average pool.  Same result as
 * ttnet_forward on the normalised float32 tensor up to stem near ties (the stem works on the exact integer byte
 * sums, with the normalisation folded into its weights: two matrix products per output instead of the three the
 * float32 input needs).  Not available for TTNET_VALEXNET. */
int ttnet_plan_set_input_norm(ttnet_plan *plan, const float *mean3, const float *std3);
int ttnet_forward_u8(ttnet_plan *plan, int lane, const uint8_t *x_nhwc_dev, int64_t n, float *logits_dev, void *stream);

/* SURVEY 8(f) N1, the part in front of ttnet_forward_u8: transforms.Resize(resize) followed by
 * transforms.CenterCrop(crop) of the eval transform (utils/preprocess.py:104-105; 256 / 224 there) on a batch of
 * n decoded images of one size, uint8 HWC [n][h][w][3] -> uint8 [n][crop][crop][3], both on the device.
 * Resize is torchvision's resize of a PIL image, i.e. Pillow's bilinear resampling with antialiasing in
 * 8-bit fixed point (horizontal pass, then vertical); the output size and crop offsets follow
 * torchvision.transforms.functional (shorter side -> resize, longer side int(resize * long / short);
 * offsets int(round((size - crop) / 2.0))).  Not bound to a plan.  Asynchronous on `stream` (one kernel; the first call with a
 * geometry uploads its coefficient tables synchronously and keeps them for the life of the process).  src must be 16-byte
 * aligned, dst 4-byte aligned, crop * 3 a multiple of 4, n <= 65535.  Byte-identical to Pillow 12.x on the committed fixture tests/golden/ref_resize.npz
 * (Pillow's own outputs for seeded images of nine geometries); torchvision is not importable where this is
 * built, so its output-size and crop-offset rules are restated. */
int ttnet_resize_center_crop_u8(const uint8_t *src_hwc_dev, int64_t n, int h, int w, int resize, int crop,
                                uint8_t *dst_hwc_dev, void *stream);

/* Same, starting from the binarised stem output (features[3], netbin.py:193) given as
 * row-packed bits uint64 [n][p][56]; used by the parity tests to separate the integer
 * gate path (bit exact) from the float stem (exact except at near ties). */
int ttnet_forward_from_stem_bits(ttnet_plan *plan, const uint64_t *rows_dev, int64_t n,
                                 float *logits_dev, void *stream);

/* Parity taps: copies a stage of the LAST forward on this plan to `dst` (host pointer
 * unless on_device).  Stages: "features.3", "features.4", "features.5" (row-packed uint64
 * [n][C][H]), "features.<b>.out1".."out4" (row-packed, after the branch padding),
 * "flatten" (float32 [n][fcsize] in the reference's C-major order), "stem.pre"
 * is not kept.  Synchronises `stream`.  Replaces the reference's debug attributes
 * Block_TT.input_layer / output_layer (models/TT_FHE_SMALL.py:310,319). */
int ttnet_read_stage(ttnet_plan *plan, const char *stage, int64_t n, void *dst, size_t dst_bytes,
                     int on_device, void *stream);

/* Truth-table export / import in the reference's canonical order
 * (Block_TT.get_TT_block_all_filter, models/TT_FHE_SMALL.py:322-342): for Block_TT `name`
 * (e.g. "features.4.Block_conv3"), `bits` is uint8 [groups][2^n][cout_per_group] with
 * entry index = the pattern read MSB first over (c_in_group, kh, kw).  get copies the
 * table the plan built; set replaces it (e.g. with truth tables published alongside a
 * checkpoint, README.md:22).  For the last (float) block the element type is float32. */
int ttnet_plan_get_table(ttnet_plan *plan, const char *name, void *dst_host, size_t dst_bytes);
int ttnet_plan_set_table(ttnet_plan *plan, const char *name, const void *src_host, size_t src_bytes);

/* Integer facts about the plan: "fcsize", "n_classes", "n_state_tensors", "max_batch",
 * "near_ties:<block_tt name>" (entries with |pre-activation| < 1e-5 found while building
 * that table), "table_bytes", "workspace_bytes", "graph_replays" (forwards replayed from a
 * captured hipGraph so far), "graphs_enabled" (0: ttnet_last_error() then says why), "graph_captures",
 * "graph_drops", "graphs_cached", "lanes", "range_overflow" (synchronises; 1 if a forward since the last
 * query left the fp16 x 2 range, and clears the flag). */
int ttnet_plan_query(ttnet_plan *plan, const char *what, int64_t *out);

/* Device time of the kernels of the last forward, measured with HIP events on the stream
 * the kernels were launched on (enable with ttnet_plan_set_profiling).  names/ms are
 * arrays of capacity `cap`; returns the number of entries.  Synchronises. */
int ttnet_plan_set_profiling(ttnet_plan *plan, int enabled);
int ttnet_plan_last_timings(ttnet_plan *plan, const char **names, float *ms, int cap);

void ttnet_plan_destroy(ttnet_plan *plan);

/* Multi-GPU: the path shards by image (eval BatchNorm uses running statistics; nothing
 * crosses samples).  The only exchange is the gather of logits that
 * torch.nn.DataParallel.gather performs on GPU 0 in the reference (main.py:192).  One
 * process per GPU; `unique_id` is the 128-byte ncclUniqueId from ttnet_comm_unique_id on
 * rank 0, distributed by the caller (e.g. through its torch.distributed store). */
int ttnet_comm_unique_id(void *id128);
int ttnet_comm_create(const void *id128, int rank, int world, int device, ttnet_comm **out);
int ttnet_allgather_logits(ttnet_comm *comm, const float *local_dev, int64_t n_local, int64_t n_classes,
                           float *all_dev, void *stream);
void ttnet_comm_destroy(ttnet_comm *comm);

const char *ttnet_last_error(void);
const char *ttnet_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TTNET_H */
