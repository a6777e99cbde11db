// libttnet.so -- plan object and the C ABI of include/ttnet.h.
//
// Host-side counterpart of the reference's model object (construction, load_state_dict,
// forward dispatch; models/TT_general_imagenet_v2_small.py:151-207,
// models/model_utils/netbin.py:703-708).  No torch types, no CPU compute path: every
// arithmetic step of forward() is a HIP kernel from stem.hip / gate.hip / head.hip, and the
// derived tables are built by lut_build.hip.  Host code here only folds BatchNorm
// parameters (a few thousand scalars, float64) and moves bytes.

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <memory>

#include <mutex>
#include <unordered_map>

#include "ttnet_common.h"

namespace ttnet {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

constexpr double kBnEps = 1e-5;   // nn.BatchNorm2d / BatchNorm1d default

struct Tensor {
  std::vector<int64_t> shape;
  int dtype = TTNET_F32;
  void *dev = nullptr;
  size_t bytes = 0;
  bool set = false;
  bool required = true;
};

struct BlockTT {
  BlockGeom g;
  std::vector<uint8_t> perm;     // internal index bit p -> canonical input column
  uint8_t *perm_dev = nullptr;
  double *s1 = nullptr, *t1 = nullptr, *s2 = nullptr, *t2 = nullptr;
  void *table = nullptr;         // internal layout
  unsigned *near_dev = nullptr;
  int64_t near_ties = -1;
  bool user_table = false;
};

struct MultiHead {
  std::string name;
  int C = 0, H = 0, W = 0, Ho = 0, Wo = 0, off34 = 0, stride = 2;
  uint64_t *c3_tmp = nullptr;    // full variant: conv3 output at input resolution, before the majority pool
  bool last = false;
  BlockTT c1, c2, c3, cf;
  uint16_t *o[4] = {nullptr, nullptr, nullptr, nullptr};
  // block-fused path (gate_fused.hip): table images; the branch dwords of a last block
  void *img_dw = nullptr, *img_c3 = nullptr;
  uint32_t *idx = nullptr;
};

struct Timing {
  const char *name;
  hipEvent_t e0, e1;
};

size_t dtype_size(int dt) {
  switch (dt) {
    case TTNET_F32: return 4;
    case TTNET_I64: return 8;
    case TTNET_U8: return 1;
    case TTNET_U16: return 2;
    case TTNET_U64: return 8;
  }
  return 0;
}

}  // namespace
}  // namespace ttnet

using namespace ttnet;

struct ttnet_plan {
  ttnet_net_desc desc{};
  int device = 0;
  int p = 0;
  int n_classes = 1000, inter = 1000, fcsize = 0;
  int featC = 0, featPP = 0;        // last block: channels, pooled pixels per channel
  std::string head;
  std::vector<MultiHead> blocks;
  std::map<std::string, Tensor> tensors;
  std::vector<std::string> key_order;
  bool finalized = false;
  bool xs = false;                  // x-small variant: row-packed branch tensors, gate_xs.hip kernels
  bool full = false;                // full variant (fan-in 30): direct float64 evaluation, gate_full.hip
  bool va = false;                  // CIFAR vAlexnet variant, gate_va.hip
  bool fused = false;               // TT-small with stride-2 blocks only: one launch per block (gate_fused.hip), activations
                                    // as compact rows; TTNET_GATE_UNFUSED=1 keeps the two-launch kernels of gate.hip
  uint32_t *tap = nullptr;          // fused path: branch dwords of a non-last block, filled on demand by ttnet_read_stage
  size_t tap_elems = 0;
  uint64_t *va_y = nullptr;         // vAlexnet: the concatenated block output [n][256][11] rows
  float *va_scale = nullptr, *va_shift = nullptr;   // vAlexnet stem BatchNorm folded
  float *last_float = nullptr;      // full variant: relu'd output of the last block before AvgPool2d
  uint32_t *full_fix = nullptr;     // full variant: [64] counters + the pixel lists of one grouped 1x1 block (gate_full.hip)
  size_t full_fix_cap = 0;          // list entries
  float *full_gel = nullptr;        // full variant: GELU tables of the fast kernels (launch_full_gelu_tables), shared by all lanes

  // stem
  uint16_t *stem_wt = nullptr;      // fp16 x 2 split weights, fragment order
  float *stem_init = nullptr;       // accumulator start values (folded BN shift), 64 floats
  uint32_t *norm_tab = nullptr;     // uint8 input: centres and border corrections (stem_split_weights_u8)
  uint16_t *stem_wt_u8 = nullptr;   // uint8 input: split weights with the normalisation folded in
  float *stem_init_u8 = nullptr;
  float in_mean[3] = {0.485f, 0.456f, 0.406f}, in_std[3] = {0.229f, 0.224f, 0.225f};   // utils/preprocess.py:107-108
  // activations: x_rp[i] / x_cp[i] = input of block i
  std::vector<uint64_t *> x_rp;
  std::vector<uint16_t *> x_cp;
  uint16_t *feat = nullptr;         // features as two fp16 planes in lin1 fragment order
  // head
  float *w1p = nullptr;             // scratch for the permuted lin1 weights (finalize only)
  uint16_t *w1f = nullptr;          // lin1 weights, two fp16 planes in fragment order
  float *bn_scale = nullptr, *bn_shift = nullptr, *part = nullptr;
  uint16_t *mid_frag = nullptr;     // lin2's A operand (head_mid_kernel), rows padded to 64
  uint16_t *w2f = nullptr;          // lin2 weights, split planes in fragment order, K padded to 16
  float lin2_inv = 1.f;             // 1 / (weight prescale x activation prescale)
  size_t part_elems = 0;
  size_t table_bytes = 0, workspace_bytes = 0;
  int64_t last_n = 0;

  bool profiling = false;
  // captured forward per batch size (hipGraph): one launch instead of ~11 on the host side
  struct GraphEntry {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t first = nullptr, last = nullptr;     // the kernels that read x / write the logits
    // own copies of those two kernels' launch parameters (argument values in 8-byte slots)
    hipKernelNodeParams first_p{}, last_p{};
    int first_nargs = 0;
    uint64_t first_argv[12] = {}, last_argv[12] = {};
    void *first_args[12] = {}, *last_args[12] = {};
    const void *x = nullptr;
    void *out = nullptr;
  };
  // A lane = one set of activation buffers (+ the graphs captured over them).  Lanes share the
  // weights and truth tables; forwards on different lanes may be in flight at the same time on
  // different streams (ttnet_forward_lane).  The workspace pointers above always mirror the
  // lane `cur`.
  struct Lane {
    std::vector<uint64_t *> x_rp;
    std::vector<uint16_t *> x_cp;
    std::vector<std::array<uint16_t *, 4>> o;
    std::vector<uint64_t *> c3_tmp;
    std::vector<uint32_t *> idx;
    uint64_t *va_y = nullptr;
    float *last_float = nullptr, *part = nullptr;
    uint32_t *full_fix = nullptr;
    uint16_t *feat = nullptr, *mid_frag = nullptr;
    std::map<int64_t, GraphEntry> graphs;
    std::map<int64_t, int> eager_calls;
    int64_t last_n = 0;
  };
  std::vector<Lane> lanes;
  int cur = 0;
  hipStream_t cap_stream = nullptr;
  bool graphs_ok = getenv("TTNET_NO_GRAPH") == nullptr;   // plain launches only when set (debugging)
  int64_t graph_replays = 0;
  int64_t graph_captures = 0, graph_drops = 0;
  std::string graph_off_reason;     // why graphs_ok went false (query "graphs_enabled" + ttnet_last_error)
  // sticky range flag: one word of host-mapped memory that kernels set when a value leaves the
  // range of the fp16 x 2 split (ttnet_common.h: split_out_of_range)
  uint32_t *range_host = nullptr, *range_dev = nullptr;
  std::vector<Timing> timings;
  size_t timing_used = 0;

  std::vector<void *> owned;        // everything hipMalloc'ed, freed in destroy
};

int ttnet::ensure_dynamic_lds(const void *kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return TTNET_OK;
  static std::mutex mu;
  static std::unordered_map<const void *, size_t> done;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find(kernel);
  if (it != done.end() && it->second >= bytes) return TTNET_OK;
  TT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done[kernel] = bytes;
  return TTNET_OK;
}

namespace {

// Captured graphs bake in by-value kernel arguments derived from the weights (lin2's 1/prescale) and
// the device addresses of tables: whenever a tensor, a table or the finalized state changes they are
// all dropped (after a device synchronisation -- a replay may still be in flight) and re-captured
// from the third forward on.
int invalidate_graphs(ttnet_plan *pl);

template <typename T>
int dev_alloc(ttnet_plan *pl, T **out, size_t count, bool zero, size_t *account = nullptr) {
  void *ptr = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
  hipError_t e = hipMalloc(&ptr, bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return TTNET_E_NOMEM;
  }
  if (zero) TT_HIP(hipMemset(ptr, 0, bytes));
  pl->owned.push_back(ptr);
  if (account) *account += bytes;
  *out = (T *)ptr;
  return TTNET_OK;
}

void add_tensor(ttnet_plan *pl, const std::string &key, std::vector<int64_t> shape, int dtype, bool required) {
  Tensor t;
  t.shape = std::move(shape);
  t.dtype = dtype;
  size_t n = 1;
  for (auto d : t.shape) n *= (size_t)d;
  t.bytes = n * dtype_size(dtype);
  t.required = required;
  pl->tensors[key] = t;
  pl->key_order.push_back(key);
}

void add_bn(ttnet_plan *pl, const std::string &prefix, int c) {
  add_tensor(pl, prefix + ".weight", {c}, TTNET_F32, true);
  add_tensor(pl, prefix + ".bias", {c}, TTNET_F32, true);
  add_tensor(pl, prefix + ".running_mean", {c}, TTNET_F32, true);
  add_tensor(pl, prefix + ".running_var", {c}, TTNET_F32, true);
  add_tensor(pl, prefix + ".num_batches_tracked", {}, TTNET_I64, false);
}

// every Block_TT of the plan (vAlexnet's block has no convf: its entry keeps an empty name)
std::vector<BlockTT *> all_block_tts(ttnet_plan *pl) {
  std::vector<BlockTT *> v;
  for (auto &mh : pl->blocks)
    for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3, &mh.cf})
      if (!b->g.name.empty()) v.push_back(b);
  return v;
}

void add_block_tt(ttnet_plan *pl, const BlockGeom &g) {
  const int mid = 8 * g.in_planes;
  add_tensor(pl, g.name + ".conv1.weight", {mid, g.cin_g(), g.kh, g.kw}, TTNET_F32, true);
  add_bn(pl, g.name + ".bn1", mid);
  add_tensor(pl, g.name + ".conv2.weight", {g.out_planes, mid / g.groups, 1, 1}, TTNET_F32, true);
  add_bn(pl, g.name + ".bn2", g.out_planes);
  add_tensor(pl, g.name + ".act.grad_scale", {}, TTNET_F32, false);
}

BlockGeom make_geom(const std::string &name, int in_planes, int out_planes, int kh, int kw, int stride, int pad,
                    int groups, bool last) {
  BlockGeom g;
  g.name = name; g.in_planes = in_planes; g.out_planes = out_planes; g.kh = kh; g.kw = kw;
  g.stride = stride; g.pad = pad; g.groups = groups; g.last = last;
  return g;
}

// Geometry of the network (mirrors make_small_network,
// models/TT_general_imagenet_v2_small.py:159-203, and the shape-keyed branch padding of
// the block forward, :98-139).
// TT_FHE_XSMALL_vAlexnet (models/TT_FHE_XSMALL_vAlexnet.py:585-660): fixed geometry
int build_geometry_valexnet(ttnet_plan *pl) {
  const ttnet_net_desc &d = pl->desc;
  if (d.image_h != 32 || d.image_w != 32) {
    set_error("vAlexnet takes 32x32 inputs (got %dx%d)", d.image_h, d.image_w);
    return TTNET_E_UNSUPPORTED;
  }
  if (d.max_batch < 1) {
    set_error("max_batch must be positive");
    return TTNET_E_INVALID;
  }
  pl->va = true;
  pl->p = 64;
  pl->n_classes = 10; pl->inter = 100; pl->fcsize = 256 * 11 * 11;
  pl->head = "features.7";
  for (const char *pre : {"VGG_Model16_0", "features.0"}) {      // one conv module registered under two names
    add_tensor(pl, std::string(pre) + ".weight", {64, 3, 3, 3}, TTNET_F32, std::string(pre) == "features.0");
    add_tensor(pl, std::string(pre) + ".bias", {64}, TTNET_F32, std::string(pre) == "features.0");
  }
  add_bn(pl, "features.2", 64);
  add_tensor(pl, "features.4.grad_scale", {}, TTNET_F32, false);
  MultiHead mh;
  mh.name = "features.5";
  mh.C = 64; mh.H = 10; mh.W = 10; mh.Ho = 11; mh.Wo = 11; mh.stride = 1; mh.last = true;
  mh.c1.g = make_geom("features.5.Block_conv1", 64, 64, 3, 2, 1, 1, 64, false);
  mh.c2.g = make_geom("features.5.Block_conv2", 64, 64, 2, 3, 1, 1, 64, false);
  mh.c3.g = make_geom("features.5.Block_conv3", 64, 64, 1, 1, 1, 0, 8, false);
  for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3}) {
    add_block_tt(pl, b->g);
    b->perm.resize(b->g.nbits());
    for (int q = 0; q < b->g.nbits(); ++q) b->perm[q] = (uint8_t)q;
  }
  pl->blocks.push_back(mh);
  add_tensor(pl, "features.7.lin1.weight", {pl->inter, pl->fcsize}, TTNET_F32, true);
  add_bn(pl, "features.7.BN2", pl->inter);
  add_tensor(pl, "features.7.lin2.weight", {pl->n_classes, pl->inter}, TTNET_F32, true);
  add_tensor(pl, "features.7.lin2.bias", {pl->n_classes}, TTNET_F32, true);
  return TTNET_OK;
}

int build_geometry(ttnet_plan *pl) {
  const ttnet_net_desc &d = pl->desc;
  if (d.variant == TTNET_VALEXNET) return build_geometry_valexnet(pl);
  int kh = 4, kw = 4, pad = 2, gsize = 16;
  if (d.variant == TTNET_SMALL) {
  } else if (d.variant == TTNET_XSMALL) {
    kh = kw = 2; pad = 1; gsize = 4;
    pl->xs = true;
  } else if (d.variant == TTNET_FULL) {
    gsize = 30; pad = 3;               // kernels (6,5) / (5,6): set per branch below
    pl->full = true;
  } else {
    set_error("unknown variant %d", d.variant);
    return TTNET_E_INVALID;
  }
  if (d.image_h != 224 || d.image_w != 224) {
    set_error("only 224x224 inputs are supported (got %dx%d)", d.image_h, d.image_w);
    return TTNET_E_UNSUPPORTED;
  }
  if (d.nfilter < 1 || d.tfilter < 1 || d.max_batch < 1) {
    set_error("nfilter, tfilter and max_batch must be positive");
    return TTNET_E_INVALID;
  }
  const int p = d.nfilter * d.tfilter;
  pl->p = p;
  // The reference constructs any p whose group counts divide its channel counts (TT_general_imagenet_v2_small.py:
  // 165-167, :28-76); a fan-in of 16 (the truth tables of the small variant) needs p % 16 == 0, and the depthwise
  // tables of both table variants are striped by 16 channels.  Built: p in {16, 32, .., 128} for the table variants
  // (one, two or four 32-channel M-tiles in the stem kernel), p <= 64 for the full variant; anything else is refused here.
  if (p > 128 || (pl->full && p > 64) || (!pl->full && p % 16 != 0)) {
    set_error("p = nfilter*tfilter = %d: built for p <= 128 with p %% 16 == 0 (fan-in 16 / tables striped by 16 channels; full variant: p <= 64)", p);
    return TTNET_E_UNSUPPORTED;
  }
  if (d.layers >= 3 && p != 64) {
    set_error("--layers %d at p = %d: the stride-1 blocks (two-launch gate kernels, channel-word layout) are built for p = 64", d.layers, p);
    return TTNET_E_UNSUPPORTED;
  }
  std::vector<int> cfg, strides;
  switch (d.layers) {   // TT_general_imagenet_v2_small.py:172-181; a bare entry is a stride-1 block
    case 0: cfg = {p, 2 * p}; strides = {2, 2}; break;
    case 1: cfg = {p, 2 * p, 4 * p}; strides = {2, 2, 2}; break;
    case 2: cfg = {p, 2 * p, 4 * p, 8 * p}; strides = {2, 2, 2, 2}; break;
    case 3: cfg = {p, 2 * p, 4 * p, 8 * p}; strides = {1, 2, 2, 2}; break;
    case 4: cfg = {p, 2 * p, 2 * p, 4 * p, 8 * p}; strides = {1, 2, 1, 2, 2}; break;
    default:
      set_error("--layers %d: the reference defines 0..4", d.layers);
      return TTNET_E_UNSUPPORTED;
  }
  if (d.layers >= 3 && (pl->xs || pl->full)) {
    set_error("--layers %d (stride-1 blocks) is built for the small variant only", d.layers);
    return TTNET_E_UNSUPPORTED;
  }
  add_tensor(pl, "features.1.weight", {p, 3, 7, 7}, TTNET_F32, true);
  add_bn(pl, "features.2", p);
  add_tensor(pl, "features.3.grad_scale", {}, TTNET_F32, false);

  int h = 56, w = 56, in_planes = p;
  for (size_t i = 0; i < cfg.size(); ++i) {
    MultiHead mh;
    mh.name = "features." + std::to_string(4 + i);
    const int out_planes = cfg[i];
    mh.last = (out_planes == cfg.back());
    mh.C = in_planes; mh.H = h; mh.W = w;
    const int stride = strides[i];
    mh.stride = stride;
    int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
    if (pl->full) {
      // models/TT_general_imagenet_v2.py:98-128: conv1 is (6,5), conv2 (5,6); at 29x29 they come
      // out 15x16 / 16x15 and are padded (bottom / right) to 16x16, out3/out4 by (0,2,0,2)
      if (w == 56) { mh.off34 = 1; ho = wo = 29; }
      else if (w == 29) { mh.off34 = 0; ho = wo = 16; }
      else if (w == 16) { mh.off34 = 0; ho = wo = 9; }
      else {
        set_error("%s: no branch-padding rule for width %d (full variant)", mh.name.c_str(), w);
        return TTNET_E_UNSUPPORTED;
      }
    } else {
    // branch padding keyed by the input width (:98-139): out3/out4 are floor(h/2) wide
    if (w == 56) mh.off34 = 1;                       // pad0 = ZeroPad2d((1,0,1,0))
    else if (w == 29 || w == 15 || w == 8 || w == 16 || w == 30 || w == 57 || w == 58) mh.off34 = 0;   // pad2 = (0,1,0,1)
    else {
      set_error("%s: no branch-padding rule for width %d", mh.name.c_str(), w);
      return TTNET_E_UNSUPPORTED;
    }
    if (h / stride + 1 != ho || w / stride + 1 != wo || h != w) {
      set_error("%s: branch shapes do not line up (%dx%d -> %dx%d)", mh.name.c_str(), h, w, ho, wo);
      return TTNET_E_UNSUPPORTED;
    }
    }
    mh.Ho = ho; mh.Wo = wo;
    if (in_planes % gsize) {
      set_error("in_channels must be divisible by groups (in_planes=%d, group size %d)", in_planes, gsize);
      return TTNET_E_INVALID;
    }
    const int kh1 = pl->full ? 6 : kh, kw1 = pl->full ? 5 : kw, kh2 = pl->full ? 5 : kh, kw2 = pl->full ? 6 : kw;
    mh.c1.g = make_geom(mh.name + ".Block_conv1", in_planes, in_planes, kh1, kw1, stride, pad, in_planes, false);
    mh.c2.g = make_geom(mh.name + ".Block_conv2", in_planes, in_planes, kh2, kw2, stride, pad, in_planes, false);
    mh.c3.g = make_geom(mh.name + ".Block_conv3", in_planes, in_planes, 1, 1, 1, 0, in_planes / gsize, false);
    const int cf_out = mh.last ? 4 * in_planes : 2 * out_planes;
    mh.cf.g = make_geom(mh.name + ".Block_convf", 4 * in_planes, cf_out, 1, 1, 1, 0, 4 * in_planes / gsize, mh.last);
    add_block_tt(pl, mh.c1.g);
    add_block_tt(pl, mh.c2.g);
    add_block_tt(pl, mh.c3.g);
    add_tensor(pl, mh.name + ".act.grad_scale", {}, TTNET_F32, false);
    add_block_tt(pl, mh.cf.g);
    // internal index orders (table variants only)
    if (!pl->full)
    for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3}) {
      b->perm.resize(b->g.nbits());
      for (int q = 0; q < b->g.nbits(); ++q) b->perm[q] = (uint8_t)q;
    }
    // convf group = gsize/4 channels x 4 branches; reference interleave is channel 4c+branch (:144-147),
    // internal index bit = (gsize/4)*branch + channel-in-group
    const int nch = gsize / 4;
    if (!pl->full) {
      mh.cf.perm.resize(gsize);
      for (int br = 0; br < 4; ++br)
        for (int cl = 0; cl < nch; ++cl) mh.cf.perm[nch * br + cl] = (uint8_t)(4 * cl + br);
    }
    pl->blocks.push_back(mh);
    h = ho; w = wo;
    in_planes = 2 * out_planes;
  }
  if (d.variant == TTNET_SMALL && getenv("TTNET_GATE_UNFUSED") == nullptr) {
    pl->fused = true;
    for (const MultiHead &mh : pl->blocks)
      pl->fused = pl->fused && fused_block_supported(mh.C, mh.H, mh.Ho, mh.stride, mh.c1.g.pad, mh.c1.g.kh, mh.c1.g.kw);
  }
  const MultiHead &lb = pl->blocks.back();
  pl->featC = lb.cf.g.out_planes;
  pl->featPP = (lb.Ho / 2) * (lb.Wo / 2);
  pl->fcsize = pl->featC * pl->featPP;
  pl->head = "features." + std::to_string(4 + cfg.size() + 2);
  add_tensor(pl, pl->head + ".lin1.weight", {pl->inter, pl->fcsize}, TTNET_F32, true);
  add_bn(pl, pl->head + ".BN2", pl->inter);
  add_tensor(pl, pl->head + ".lin2.weight", {pl->n_classes, pl->inter}, TTNET_F32, true);
  add_tensor(pl, pl->head + ".lin2.bias", {pl->n_classes}, TTNET_F32, true);
  return TTNET_OK;
}

// Activation workspace of one lane, into the plan's current workspace pointers.  Everything is
// zeroed once: the branch-padding borders, the k padding of lin2's A operand and the rows of the
// lin1 operand beyond the batch are never written again.
int alloc_workspace(ttnet_plan *pl) {
  const int nb = pl->desc.max_batch;
  size_t *ws = &pl->workspace_bytes;
  const int kpad = (pl->inter + 15) / 16 * 16;
  if (pl->va) {
    pl->x_rp.assign(1, nullptr);
    pl->x_cp.assign(1, nullptr);
    TT_TRY(dev_alloc(pl, &pl->x_rp[0], (size_t)nb * 64 * 10, true, ws));
    TT_TRY(dev_alloc(pl, &pl->va_y, (size_t)nb * 256 * 11, true, ws));
  } else {
    pl->x_rp.assign(pl->blocks.size(), nullptr);
    pl->x_cp.assign(pl->blocks.size(), nullptr);
    for (size_t i = 0; i < pl->blocks.size(); ++i) {
      MultiHead &mh = pl->blocks[i];
      if (pl->fused) {
        // one activation layout: rows of uint64 / uint32 / uint16 words by width (the stem's output stays uint64)
        const size_t bytes = (size_t)nb * mh.C * mh.H * (i == 0 ? 8 : row_bytes(mh.W));
        TT_TRY(dev_alloc(pl, &pl->x_rp[i], (bytes + 7) / 8, true, ws));
        TT_TRY(dev_alloc(pl, &pl->x_cp[i], 4, true, ws));
        for (int b = 0; b < 4; ++b) TT_TRY(dev_alloc(pl, &mh.o[b], 8, true, ws));
        mh.idx = nullptr;
        if (mh.last) TT_TRY(dev_alloc(pl, &mh.idx, (size_t)nb * (mh.C / 8) * mh.Ho * mh.Wo, true, ws));
        continue;
      }
      TT_TRY(dev_alloc(pl, &pl->x_rp[i], (size_t)nb * mh.C * mh.H, true, ws));
      TT_TRY(dev_alloc(pl, &pl->x_cp[i], pl->full ? 8 : (size_t)nb * mh.H * mh.W * (mh.C / 16), true, ws));
      if (pl->full) TT_TRY(dev_alloc(pl, &mh.c3_tmp, (size_t)nb * mh.C * mh.H, true, ws));
      for (int b = 0; b < 4; ++b) {
        const size_t words16 = (size_t)nb * mh.Ho * mh.Wo * (mh.C / 16), rows64 = (size_t)nb * mh.C * mh.Ho;
        TT_TRY(dev_alloc(pl, &mh.o[b], (pl->xs || pl->full) ? rows64 * 4 : words16, true, ws));
      }
    }
    if (pl->full) {
      const MultiHead &lb = pl->blocks.back();
      TT_TRY(dev_alloc(pl, &pl->last_float, (size_t)nb * lb.cf.g.out_planes * lb.Ho * lb.Wo, true, ws));
      size_t cap = 0;                                  // pixels x groups of the largest binarised 1x1 block
      for (const MultiHead &mh : pl->blocks) {
        cap = std::max(cap, (size_t)mh.c3.g.groups * mh.H * mh.W);
        if (!mh.last) cap = std::max(cap, (size_t)mh.cf.g.groups * mh.Ho * mh.Wo);
      }
      pl->full_fix_cap = cap * (size_t)nb;
      TT_TRY(dev_alloc(pl, &pl->full_fix, 64 + pl->full_fix_cap, true, ws));
      if (!pl->full_gel) {
        TT_TRY(dev_alloc(pl, &pl->full_gel, full_gelu_tables_elems(), false, ws));
        TT_TRY(launch_full_gelu_tables(pl->full_gel, nullptr));
        TT_HIP(hipDeviceSynchronize());
      }
    }
  }
  const int nb_pad = (nb + 255) / 256 * 256;          // the lin1 GEMM walks whole 256-row tiles
  TT_TRY(dev_alloc(pl, &pl->feat, frag_elems(nb_pad, pl->fcsize), true, ws));
  TT_TRY(dev_alloc(pl, &pl->mid_frag, frag_elems((nb + 63) / 64 * 64, kpad), true, ws));
  size_t pe = 0;
  for (int n = 1; n <= nb; n = n < 256 ? 256 : n + 256) pe = std::max(pe, gemm_f16x2_part_elems(std::min(n, nb), pl->inter, pl->fcsize / 16));
  pe = std::max(pe, gemm_f16x2_part_elems(nb, pl->inter, pl->fcsize / 16));
  pl->part_elems = pe;
  TT_TRY(dev_alloc(pl, &pl->part, pe, false, ws));
  return TTNET_OK;
}

void store_lane(ttnet_plan *pl, ttnet_plan::Lane &l) {
  l.x_rp = pl->x_rp;
  l.x_cp = pl->x_cp;
  l.o.resize(pl->blocks.size());
  l.c3_tmp.resize(pl->blocks.size());
  l.idx.resize(pl->blocks.size());
  for (size_t i = 0; i < pl->blocks.size(); ++i) {
    for (int b = 0; b < 4; ++b) l.o[i][b] = pl->blocks[i].o[b];
    l.c3_tmp[i] = pl->blocks[i].c3_tmp;
    l.idx[i] = pl->blocks[i].idx;
  }
  l.va_y = pl->va_y;
  l.last_float = pl->last_float;
  l.full_fix = pl->full_fix;
  l.part = pl->part;
  l.feat = pl->feat;
  l.mid_frag = pl->mid_frag;
}

void load_lane(ttnet_plan *pl, const ttnet_plan::Lane &l) {
  pl->x_rp = l.x_rp;
  pl->x_cp = l.x_cp;
  for (size_t i = 0; i < pl->blocks.size(); ++i) {
    for (int b = 0; b < 4; ++b) pl->blocks[i].o[b] = l.o[i][b];
    pl->blocks[i].c3_tmp = l.c3_tmp[i];
    pl->blocks[i].idx = l.idx[i];
  }
  pl->va_y = l.va_y;
  pl->last_float = l.last_float;
  pl->full_fix = l.full_fix;
  pl->part = l.part;
  pl->feat = l.feat;
  pl->mid_frag = l.mid_frag;
}

void switch_lane(ttnet_plan *pl, int k) {
  if (k == pl->cur) return;
  load_lane(pl, pl->lanes[k]);
  pl->cur = k;
}

int allocate(ttnet_plan *pl) {
  size_t *tb = &pl->table_bytes;
  TT_HIP(hipHostMalloc((void **)&pl->range_host, sizeof(uint32_t), hipHostMallocMapped));
  *pl->range_host = 0u;
  TT_HIP(hipHostGetDevicePointer((void **)&pl->range_dev, pl->range_host, 0));
  const int kpad = (pl->inter + 15) / 16 * 16;
  for (auto &kv : pl->tensors) TT_TRY(dev_alloc(pl, (uint8_t **)&kv.second.dev, kv.second.bytes, true));
  if (pl->va) {
    TT_TRY(dev_alloc(pl, &pl->va_scale, 64, false));
    TT_TRY(dev_alloc(pl, &pl->va_shift, 64, false));
  } else {
    TT_TRY(dev_alloc(pl, &pl->stem_wt, stem_split_weights_elems(), false));
    TT_TRY(dev_alloc(pl, &pl->stem_init, 64, false));
    TT_TRY(dev_alloc(pl, &pl->norm_tab, stem_u8_table_elems(), true));
    TT_TRY(dev_alloc(pl, &pl->stem_wt_u8, stem_split_weights_elems(), true));
    TT_TRY(dev_alloc(pl, &pl->stem_init_u8, 64, true));
    TT_TRY(dev_alloc(pl, &pl->w1p, (size_t)pl->inter * pl->fcsize, false));
  }
  for (auto &mh : pl->blocks) {
    for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3, &mh.cf}) {
      if (pl->va && b == &mh.cf) continue;            // vAlexnet has no Block_convf
      const BlockGeom &g = b->g;
      if (!pl->full) {
        TT_TRY(dev_alloc(pl, (uint8_t **)&b->table, g.table_bytes(), true, tb));
        TT_TRY(dev_alloc(pl, &b->perm_dev, b->perm.size(), false));
        TT_HIP(hipMemcpy(b->perm_dev, b->perm.data(), b->perm.size(), hipMemcpyHostToDevice));
      }
      TT_TRY(dev_alloc(pl, &b->s1, (size_t)8 * g.in_planes, false));
      TT_TRY(dev_alloc(pl, &b->t1, (size_t)8 * g.in_planes, false));
      TT_TRY(dev_alloc(pl, &b->s2, g.out_planes, false));
      TT_TRY(dev_alloc(pl, &b->t2, g.out_planes, false));
      TT_TRY(dev_alloc(pl, &b->near_dev, 1, true));
    }
  }
  if (pl->fused)
    for (auto &mh : pl->blocks) {
      TT_TRY(dev_alloc(pl, (uint8_t **)&mh.img_dw, (size_t)mh.C * 16384, false, tb));
      TT_TRY(dev_alloc(pl, (uint8_t **)&mh.img_c3, (size_t)(mh.C / 8) * 65536, false, tb));
    }
  TT_TRY(dev_alloc(pl, &pl->w1f, frag_elems(pl->va ? 128 : (pl->inter + 127) / 128 * 128, pl->fcsize), false));
  TT_TRY(dev_alloc(pl, &pl->bn_scale, pl->inter, false));
  TT_TRY(dev_alloc(pl, &pl->bn_shift, pl->inter, false));
  TT_TRY(dev_alloc(pl, &pl->w2f, frag_elems((pl->n_classes + 63) / 64 * 64, kpad), true));
  // lane 0
  TT_TRY(alloc_workspace(pl));
  pl->lanes.resize(1);
  store_lane(pl, pl->lanes[0]);
  pl->cur = 0;
  return TTNET_OK;
}

int fetch(const Tensor &t, std::vector<float> &host) {
  host.resize(t.bytes / 4);
  TT_HIP(hipMemcpy(host.data(), t.dev, t.bytes, hipMemcpyDeviceToHost));
  return TTNET_OK;
}

// eval-mode BatchNorm as y = x*scale + shift, folded in float64
int fold_bn(ttnet_plan *pl, const std::string &prefix, std::vector<double> &scale, std::vector<double> &shift) {
  std::vector<float> w, b, m, v;
  TT_TRY(fetch(pl->tensors[prefix + ".weight"], w));
  TT_TRY(fetch(pl->tensors[prefix + ".bias"], b));
  TT_TRY(fetch(pl->tensors[prefix + ".running_mean"], m));
  TT_TRY(fetch(pl->tensors[prefix + ".running_var"], v));
  scale.resize(w.size());
  shift.resize(w.size());
  for (size_t i = 0; i < w.size(); ++i) {
    scale[i] = (double)w[i] / sqrt((double)v[i] + kBnEps);
    shift[i] = (double)b[i] - (double)m[i] * scale[i];
  }
  return TTNET_OK;
}

// uint8 input: the stem's weights with ToTensor + Normalize folded in (stem.hip, U8).  From the loaded weights and the
// plan's mean / std: at finalize and again whenever ttnet_plan_set_input_norm changes them.
int prepare_stem_u8(ttnet_plan *pl) {
  std::vector<float> w;
  TT_TRY(fetch(pl->tensors["features.1.weight"], w));
  std::vector<uint16_t> wf(stem_split_weights_elems());
  std::vector<uint32_t> tab(stem_u8_table_elems(), 0u);
  std::vector<double> sc, sh;
  TT_TRY(fold_bn(pl, "features.2", sc, sh));
  float init[64];
  if (!stem_split_weights_u8(w.data(), sc.data(), sh.data(), pl->p, pl->in_mean, pl->in_std, wf.data(), init, tab.data())) {
    set_error("stem (uint8 input): the folded BatchNorm shift of features.2 is outside the range of the split operands");
    return TTNET_E_UNSUPPORTED;
  }
  TT_HIP(hipMemcpy(pl->stem_wt_u8, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
  TT_HIP(hipMemcpy(pl->stem_init_u8, init, sizeof(init), hipMemcpyHostToDevice));
  TT_HIP(hipMemcpy(pl->norm_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  return TTNET_OK;
}


int upload_f32(float *dst, const std::vector<double> &src) {
  std::vector<float> f(src.begin(), src.end());
  TT_HIP(hipMemcpy(dst, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  return TTNET_OK;
}

int build_table(ttnet_plan *pl, BlockTT &b, hipStream_t s) {
  std::vector<double> s1, t1, s2, t2;
  TT_TRY(fold_bn(pl, b.g.name + ".bn1", s1, t1));
  TT_TRY(fold_bn(pl, b.g.name + ".bn2", s2, t2));
  TT_HIP(hipMemcpy(b.s1, s1.data(), s1.size() * 8, hipMemcpyHostToDevice));
  TT_HIP(hipMemcpy(b.t1, t1.data(), t1.size() * 8, hipMemcpyHostToDevice));
  TT_HIP(hipMemcpy(b.s2, s2.data(), s2.size() * 8, hipMemcpyHostToDevice));
  TT_HIP(hipMemcpy(b.t2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
  if (b.user_table || pl->full) return TTNET_OK;
  TT_HIP(hipMemsetAsync(b.near_dev, 0, sizeof(unsigned), s));
  LutBuildArgs a{};
  a.w1 = (const float *)pl->tensors[b.g.name + ".conv1.weight"].dev;
  a.w2 = (const float *)pl->tensors[b.g.name + ".conv2.weight"].dev;
  a.s1 = b.s1; a.t1 = b.t1; a.s2 = b.s2; a.t2 = b.t2;
  a.perm = b.perm_dev;
  a.groups = b.g.groups; a.n = b.g.nbits(); a.mid_g = b.g.mid_g(); a.cout_g = b.g.cout_g(); a.last = b.g.last;
  a.table = b.table;
  a.near_ties = b.near_dev;
  return launch_lut_build(a, s);
}

void begin_timing(ttnet_plan *pl, const char *name, hipStream_t s) {
  if (!pl->profiling) return;
  if (pl->timing_used == pl->timings.size()) {
    Timing t{name, nullptr, nullptr};
    (void)hipEventCreate(&t.e0);
    (void)hipEventCreate(&t.e1);
    pl->timings.push_back(t);
  }
  pl->timings[pl->timing_used].name = name;
  (void)hipEventRecord(pl->timings[pl->timing_used].e0, s);
}
void end_timing(ttnet_plan *pl, hipStream_t s) {
  if (!pl->profiling) return;
  (void)hipEventRecord(pl->timings[pl->timing_used].e1, s);
  pl->timing_used++;
}

#define TT_TIMED(pl, name, s, expr) \
  do {                              \
    begin_timing(pl, name, s);      \
    int r__ = (expr);               \
    end_timing(pl, s);              \
    if (r__ != TTNET_OK) return r__; \
  } while (0)

GateBlockArgs gate_args(ttnet_plan *pl, size_t i, int n) {
  MultiHead &mh = pl->blocks[i];
  GateBlockArgs a{};
  a.n = n; a.C = mh.C; a.H = mh.H; a.W = mh.W; a.Ho = mh.Ho; a.Wo = mh.Wo; a.off34 = mh.off34;
  a.kh1 = mh.c1.g.kh; a.kw1 = mh.c1.g.kw; a.kh2 = mh.c2.g.kh; a.kw2 = mh.c2.g.kw;
  a.stride = mh.c1.g.stride; a.pad = mh.c1.g.pad;
  a.cf_bits = mh.cf.g.cout_g();
  a.x_rp = pl->x_rp[i]; a.x_cp = pl->x_cp[i];
  a.t_dw1 = (const uint8_t *)mh.c1.table; a.t_dw2 = (const uint8_t *)mh.c2.table;
  a.t_c3 = (const uint16_t *)mh.c3.table;
  a.o1 = mh.o[0]; a.o2 = mh.o[1]; a.o3 = mh.o[2]; a.o4 = mh.o[3];
  return a;
}

static const char *kS1Names[] = {"gate_stage1.f4", "gate_stage1.f5", "gate_stage1.f6", "gate_stage1.f7"};
static const char *kPfNames[] = {"gate_pf.f4", "gate_pf.f5", "gate_pf.f6", "gate_pf.f7"};

// One block of the full variant: direct float64 evaluation (gate_full.hip)
int run_full_block(ttnet_plan *pl, size_t i, int n, hipStream_t s) {
  MultiHead &mh = pl->blocks[i];
  uint64_t *o64[4] = {(uint64_t *)mh.o[0], (uint64_t *)mh.o[1], (uint64_t *)mh.o[2], (uint64_t *)mh.o[3]};
  auto wts = [&](const BlockTT &b, const char *leaf) { return (const float *)pl->tensors[b.g.name + leaf].dev; };
  for (int br = 0; br < 2; ++br) {
    const BlockTT &b = br ? mh.c2 : mh.c1;
    FullDwArgs a{};
    a.n = n; a.C = mh.C; a.H = mh.H; a.W = mh.W;
    a.kh = b.g.kh; a.kw = b.g.kw; a.stride = b.g.stride; a.pad = b.g.pad;
    a.ho = (mh.H + 2 * a.pad - a.kh) / a.stride + 1;
    a.wo = (mh.W + 2 * a.pad - a.kw) / a.stride + 1;
    a.Ho = mh.Ho; a.pad_t = 0; a.pad_l = 0;          // out1 / out2 only ever get bottom / right zero padding
    a.x_rp = pl->x_rp[i];
    a.w1 = wts(b, ".conv1.weight"); a.w2 = wts(b, ".conv2.weight");
    a.s1 = b.s1; a.t1 = b.t1; a.s2 = b.s2; a.t2 = b.t2;
    a.out = o64[br];
    a.gel = pl->full_gel;
    if (pl->full_fix) {                                // (the list area is shared with the 1x1 blocks: launches are ordered)
      a.fix_count = pl->full_fix;
      a.fix_list = pl->full_fix + 64;
      a.fix_cap = (uint32_t)std::min<size_t>(pl->full_fix_cap, 0x7FFFFFFFu);
    }
    static const char *const kDw[2][4] = {{"full.conv1.f4", "full.conv1.f5", "full.conv1.f6", "full.conv1.f7"},
                                         {"full.conv2.f4", "full.conv2.f5", "full.conv2.f6", "full.conv2.f7"}};
    TT_TIMED(pl, kDw[br][std::min<size_t>(i, 3)], s, launch_full_dw(a, s));
  }
  {
    const BlockTT &b = mh.c3;
    FullPwArgs a{};
    a.n = n; a.H = mh.H; a.W = mh.W;
    a.groups = b.g.groups; a.cin = b.g.cin_g(); a.mid = b.g.mid_g(); a.cout = b.g.cout_g(); a.Cout = b.g.out_planes;
    a.Csrc = mh.C; a.interleaved = 0; a.src[0] = pl->x_rp[i];
    a.w1 = wts(b, ".conv1.weight"); a.w2 = wts(b, ".conv2.weight");
    a.s1 = b.s1; a.t1 = b.t1; a.s2 = b.s2; a.t2 = b.t2;
    a.out_rp = mh.c3_tmp; a.out_float = nullptr;
    a.gel = pl->full_gel ? pl->full_gel + full_gelu_tables_elems() / 2 : nullptr;
    a.fix_count = pl->full_fix; a.fix_list = pl->full_fix ? pl->full_fix + 64 : nullptr; a.range_flag = pl->range_dev;
    static const char *const kC3[4] = {"full.conv3.f4", "full.conv3.f5", "full.conv3.f6", "full.conv3.f7"};
    TT_TIMED(pl, kC3[std::min<size_t>(i, 3)], s, launch_full_pw(a, s));
    TT_TIMED(pl, "full.maj3", s,
             launch_rp_majority(mh.c3_tmp, o64[2], n, mh.C, mh.H, mh.W, mh.Ho, mh.off34, mh.off34, s));
    TT_TIMED(pl, "full.maj4", s,
             launch_rp_majority(pl->x_rp[i], o64[3], n, mh.C, mh.H, mh.W, mh.Ho, mh.off34, mh.off34, s));
  }
  {
    const BlockTT &b = mh.cf;
    FullPwArgs a{};
    a.n = n; a.H = mh.Ho; a.W = mh.Wo;
    a.groups = b.g.groups; a.cin = b.g.cin_g(); a.mid = b.g.mid_g(); a.cout = b.g.cout_g(); a.Cout = b.g.out_planes;
    a.Csrc = mh.C; a.interleaved = 1;
    for (int k = 0; k < 4; ++k) a.src[k] = o64[k];
    a.w1 = wts(b, ".conv1.weight"); a.w2 = wts(b, ".conv2.weight");
    a.s1 = b.s1; a.t1 = b.t1; a.s2 = b.s2; a.t2 = b.t2;
    a.gel = pl->full_gel ? pl->full_gel + full_gelu_tables_elems() / 2 : nullptr;
    a.fix_count = pl->full_fix; a.fix_list = pl->full_fix ? pl->full_fix + 64 : nullptr; a.range_flag = pl->range_dev;
    if (mh.last) {
      a.out_rp = nullptr; a.out_float = pl->last_float;
      TT_TIMED(pl, "full.convf_last", s, launch_full_pw(a, s));
      TT_TIMED(pl, "full.pool", s, launch_full_pool_split(pl->last_float, pl->feat, n, b.g.out_planes, mh.Ho, mh.Wo, pl->range_dev, s));
    } else {
      a.out_rp = pl->x_rp[i + 1]; a.out_float = nullptr;
      static const char *const kCf[4] = {"full.convf.f4", "full.convf.f5", "full.convf.f6", "full.convf.f7"};
      TT_TIMED(pl, kCf[std::min<size_t>(i, 3)], s, launch_full_pw(a, s));
    }
  }
  return TTNET_OK;
}

// split lin2.weight into w2f (finalize)
int prepare_lin2(ttnet_plan *pl, const std::string &key, hipStream_t s) {
  std::vector<float> w2;
  TT_TRY(fetch(pl->tensors[key], w2));
  const float ws2 = weight_prescale(w2.data(), w2.size());
  pl->lin2_inv = 1.0f / (ws2 * ACT_PRESCALE);
  const int kpad = (pl->inter + 15) / 16 * 16;
  return launch_split_to_frag((const float *)pl->tensors[key].dev, pl->w2f, pl->n_classes, kpad, (pl->n_classes + 63) / 64 * 64, ws2, s,
                              pl->inter, pl->inter);
}

// Classifier_scale from the features in pl->feat
int run_head(ttnet_plan *pl, int n, float *logits, int polynomial, const std::string &head, hipStream_t s) {
  const int s1 = gemm_f16x2_splits(n, pl->inter, pl->fcsize / 16);
  TT_TIMED(pl, "head.lin1", s, launch_gemm_f16x2(pl->feat, pl->w1f, pl->part, n, pl->inter, pl->fcsize, s1, s));
  TT_TIMED(pl, polynomial ? "head.bn_poly" : "head.bn", s,
           launch_head_mid(pl->part, s1, pl->bn_scale, pl->bn_shift, pl->mid_frag, n, pl->inter, polynomial, pl->range_dev, s));
  TT_TIMED(pl, "head.lin2", s,
           launch_lin2_f16x2(pl->mid_frag, pl->w2f, (const float *)pl->tensors[head + ".lin2.bias"].dev, pl->lin2_inv, logits, n,
                             pl->n_classes, pl->inter, s));
  return TTNET_OK;
}

// vAlexnet: block + Flatten + Classifier_scale from the stem bits in x_rp[0]
int run_va_tail(ttnet_plan *pl, int n, float *logits, hipStream_t s) {
  MultiHead &mh = pl->blocks[0];
  TT_TIMED(pl, "va.block", s, launch_va_block(pl->x_rp[0], mh.c1.table, mh.c2.table, mh.c3.table, pl->va_y, n, s));
  TT_TIMED(pl, "va.flatten", s, launch_va_feat(pl->va_y, pl->feat, n, s));
  TT_TRY(run_head(pl, n, logits, 0, "features.7", s));
  pl->last_n = n;
  return TTNET_OK;
}

int run_from_blocks(ttnet_plan *pl, int n, float *logits, hipStream_t s) {
  for (size_t i = 0; i < pl->blocks.size(); ++i) {
    MultiHead &mh = pl->blocks[i];
    GateBlockArgs a = gate_args(pl, i, n);
    if (pl->full) {
      TT_TRY(run_full_block(pl, i, n, s));
      continue;
    }
    if (pl->xs) {
      uint64_t *const o64[4] = {(uint64_t *)mh.o[0], (uint64_t *)mh.o[1], (uint64_t *)mh.o[2], (uint64_t *)mh.o[3]};
      TT_TIMED(pl, kS1Names[i], s, launch_xs_branches(a, mh.c3.table, o64, s));
      if (!mh.last)
        TT_TIMED(pl, kPfNames[i], s,
                 launch_xs_pf(n, mh.C, mh.Ho, mh.Wo, mh.cf.g.cout_g(), o64, mh.cf.table, pl->x_rp[i + 1], s));
      else
        TT_TIMED(pl, "gate_last", s,
                 launch_xs_last(n, mh.C, mh.Ho, mh.Wo, mh.cf.g.cout_g(), o64, (const float *)mh.cf.table, pl->feat, pl->range_dev, s));
      continue;
    }
    if (pl->fused) {
      static const char *kBlkNames[] = {"gate_block.f4", "gate_block.f5", "gate_block.f6", "gate_block.f7"};
      FusedBlockArgs f{};
      f.n = n; f.C = mh.C; f.H = mh.H; f.Ho = mh.Ho; f.off34 = mh.off34; f.last = mh.last ? 1 : 0;
      f.x = pl->x_rp[i]; f.img_c3 = mh.img_c3; f.img_dw = mh.img_dw;
      f.t_cf = mh.last ? nullptr : (const uint8_t *)mh.cf.table;
      f.y = mh.last ? nullptr : (void *)pl->x_rp[i + 1];
      f.idx = mh.last ? mh.idx : nullptr;
      TT_TIMED(pl, kBlkNames[std::min<size_t>(i, 3)], s, launch_gate_block(f, s));
      if (mh.last) TT_TIMED(pl, "gate_last", s, launch_gate_last(a, (const float *)mh.cf.table, pl->feat, pl->range_dev, s, mh.idx));
      continue;
    }
    TT_TIMED(pl, kS1Names[i], s, launch_gate_stage1(a, s));
    if (!mh.last) {
      TT_TIMED(pl, kPfNames[i], s,
               launch_gate_pf(a, (const uint8_t *)mh.cf.table, pl->x_cp[i + 1], pl->x_rp[i + 1], s));
    } else {
      TT_TIMED(pl, "gate_last", s, launch_gate_last(a, (const float *)mh.cf.table, pl->feat, pl->range_dev, s));
    }
  }
  TT_TRY(run_head(pl, n, logits, 1, pl->head, s));
  pl->last_n = n;
  return TTNET_OK;
}

// The range flag is raised by a kernel, i.e. asynchronously: the forward that overflowed has already
// returned TTNET_OK.  Every later call on the plan fails until the flag is read (and cleared) with
// ttnet_plan_query("range_overflow"); ttnet_read_stage checks it after its own synchronisation.
int check_range(ttnet_plan *pl) {
  if (pl->range_host && *(volatile uint32_t *)pl->range_host) {
    set_error("an earlier forward on this plan met an activation outside the range of the fp16 x 2 operand split "
              "(|input| or |feature| >= 4094, or NaN): its logits are invalid; ttnet_plan_query(\"range_overflow\") "
              "reads and clears the flag");
    return TTNET_E_RANGE;
  }
  return TTNET_OK;
}

int check_ready(ttnet_plan *pl, const void *in, int64_t n, const void *out) {
  if (!pl || !in || !out) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  if (!pl->finalized) {
    set_error("ttnet_forward before ttnet_plan_finalize");
    return TTNET_E_STATE;
  }
  if (n < 1 || n > pl->desc.max_batch) {
    set_error("batch %lld outside [1, max_batch=%d]", (long long)n, pl->desc.max_batch);
    return TTNET_E_INVALID;
  }
  return check_range(pl);
}

BlockTT *find_block(ttnet_plan *pl, const char *name) {
  for (BlockTT *b : all_block_tts(pl))
    if (b->g.name == name) return b;
  return nullptr;
}

// canonical index (pattern read MSB first over (c,kh,kw), TT_FHE_SMALL.py:330-334) of an
// internal index
inline uint32_t canonical_index(const BlockTT &b, uint32_t idx) {
  const int n = b.g.nbits();
  uint32_t ci = 0;
  for (int p = 0; p < n; ++p)
    if ((idx >> p) & 1u) ci |= 1u << (n - 1 - b.perm[p]);
  return ci;
}

}  // namespace

extern "C" {

const char *ttnet_last_error(void) { return g_err; }
const char *ttnet_version(void) { return "ttnet-mi355x 0.1 (gfx950)"; }

int ttnet_plan_create(const ttnet_net_desc *desc, int device, ttnet_plan **out) {
  if (!desc || !out) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
    set_error("no HIP device: libttnet has no CPU path");
    return TTNET_E_HIP;
  }
  if (device < 0 || device >= count) {
    set_error("device %d out of range (%d devices)", device, count);
    return TTNET_E_INVALID;
  }
  TT_HIP(hipSetDevice(device));
  std::unique_ptr<ttnet_plan> pl(new ttnet_plan());
  pl->desc = *desc;
  pl->device = device;
  int st = build_geometry(pl.get());
  if (st == TTNET_OK) st = allocate(pl.get());
  if (st != TTNET_OK) {
    for (void *ptr : pl->owned) (void)hipFree(ptr);
    return st;
  }
  *out = pl.release();
  return TTNET_OK;
}

int ttnet_plan_set_tensor(ttnet_plan *pl, const char *key, const void *ptr, const int64_t *shape, int ndim,
                          int dtype, int on_device) {
  if (!pl || !key || !ptr || (ndim > 0 && !shape)) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  std::string k(key);
  if (k.rfind("module.", 0) == 0) k = k.substr(7);   // DataParallel / DDP checkpoints (main.py:181-192)
  auto it = pl->tensors.find(k);
  if (it == pl->tensors.end()) {
    set_error("unexpected key in state_dict: %s", key);
    return TTNET_E_INVALID;
  }
  Tensor &t = it->second;
  bool ok = (dtype == t.dtype) && (ndim == (int)t.shape.size());
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == t.shape[i];
  if (!ok) {
    set_error("size mismatch for %s", key);
    return TTNET_E_INVALID;
  }
  TT_HIP(hipSetDevice(pl->device));
  TT_TRY(invalidate_graphs(pl));
  TT_HIP(hipMemcpy(t.dev, ptr, t.bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  t.set = true;
  pl->finalized = false;
  // new parameters for a Block_TT invalidate a table injected with ttnet_plan_set_table
  for (BlockTT *b : all_block_tts(pl))
    if (k.compare(0, b->g.name.size() + 1, b->g.name + ".") == 0) b->user_table = false;
  return TTNET_OK;
}

int ttnet_plan_finalize(ttnet_plan *pl, void *stream) {
  if (!pl) {
    set_error("null plan");
    return TTNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  TT_HIP(hipSetDevice(pl->device));
  TT_TRY(invalidate_graphs(pl));
  for (auto &k : pl->key_order) {
    const Tensor &t = pl->tensors[k];
    if (t.required && !t.set) {
      set_error("missing key in state_dict: %s", k.c_str());
      return TTNET_E_STATE;
    }
  }
  if (pl->va) {
    std::vector<double> sc, sh;
    TT_TRY(fold_bn(pl, "features.2", sc, sh));
    TT_TRY(upload_f32(pl->va_scale, sc));
    TT_TRY(upload_f32(pl->va_shift, sh));
    for (BlockTT *b : all_block_tts(pl)) TT_TRY(build_table(pl, *b, s));
    TT_TRY(fold_bn(pl, "features.7.BN2", sc, sh));
    float ws1 = 1.f;
    {
      std::vector<float> w1;
      TT_TRY(fetch(pl->tensors["features.7.lin1.weight"], w1));
      ws1 = weight_prescale(w1.data(), w1.size());
    }
    for (double &v : sc) v /= (double)ws1 * ACT_PRESCALE;      // operand prescales (powers of two) out of lin1's result
    TT_TRY(upload_f32(pl->bn_scale, sc));
    TT_TRY(upload_f32(pl->bn_shift, sh));
    TT_TRY(launch_split_to_frag((const float *)pl->tensors["features.7.lin1.weight"].dev, pl->w1f, pl->inter, pl->fcsize, 128, ws1,
                                s));
    TT_TRY(prepare_lin2(pl, "features.7.lin2.weight", s));
    TT_HIP(hipStreamSynchronize(s));
    for (BlockTT *b : all_block_tts(pl)) {
      if (b->user_table) continue;
      unsigned v = 0;
      TT_HIP(hipMemcpy(&v, b->near_dev, sizeof(v), hipMemcpyDeviceToHost));
      b->near_ties = v;
    }
    pl->finalized = true;
    return TTNET_OK;
  }
  // stem: BN scale folded into the weights, which are split into two prescaled fp16 planes in MFMA
  // fragment order; BN shift as the weights of one more k-row (stem.hip)
  {
    std::vector<float> w;
    TT_TRY(fetch(pl->tensors["features.1.weight"], w));
    std::vector<uint16_t> wf(stem_split_weights_elems());
    std::vector<double> sc, sh;
    TT_TRY(fold_bn(pl, "features.2", sc, sh));
    float init[64];
    if (!stem_split_weights(w.data(), sc.data(), sh.data(), pl->p, wf.data(), init)) {
      set_error("stem: the folded BatchNorm shift of features.2 is outside the range of the split operands");
      return TTNET_E_UNSUPPORTED;
    }
    TT_HIP(hipMemcpy(pl->stem_wt, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
    TT_HIP(hipMemcpy(pl->stem_init, init, sizeof(init), hipMemcpyHostToDevice));
    TT_TRY(prepare_stem_u8(pl));
  }
  for (auto &mh : pl->blocks) {
    for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3, &mh.cf}) TT_TRY(build_table(pl, *b, s));
    if (pl->fused) TT_TRY(launch_fused_images(mh.c1.table, mh.c2.table, mh.c3.table, mh.C, mh.img_dw, mh.img_c3, s));
  }
  {
    std::vector<double> sc, sh;
    TT_TRY(fold_bn(pl, pl->head + ".BN2", sc, sh));
    float ws1 = 1.f;
    {
      std::vector<float> w1;
      TT_TRY(fetch(pl->tensors[pl->head + ".lin1.weight"], w1));
      ws1 = weight_prescale(w1.data(), w1.size());
    }
    for (double &v : sc) v /= (double)ws1 * ACT_PRESCALE;      // operand prescales (powers of two) out of lin1's result
    TT_TRY(upload_f32(pl->bn_scale, sc));
    TT_TRY(upload_f32(pl->bn_shift, sh));
    TT_TRY(launch_permute_lin1((const float *)pl->tensors[pl->head + ".lin1.weight"].dev, pl->w1p, pl->inter,
                               pl->featC / 16, pl->featPP, s));
    TT_TRY(launch_split_to_frag(pl->w1p, pl->w1f, pl->inter, pl->fcsize, (pl->inter + 127) / 128 * 128, ws1, s));
    TT_TRY(prepare_lin2(pl, pl->head + ".lin2.weight", s));
  }
  TT_HIP(hipStreamSynchronize(s));
  for (auto &mh : pl->blocks)
    for (BlockTT *b : {&mh.c1, &mh.c2, &mh.c3, &mh.cf}) {
      if (b->user_table || pl->full) continue;
      unsigned v = 0;
      TT_HIP(hipMemcpy(&v, b->near_dev, sizeof(v), hipMemcpyDeviceToHost));
      b->near_ties = v;
    }
  pl->finalized = true;
  return TTNET_OK;
}

namespace {

int forward_eager(ttnet_plan *pl, const void *x_dev, bool u8, int64_t n, float *logits_dev, hipStream_t s) {
  pl->timing_used = 0;
  if (pl->va) {
    if (u8) {
      set_error("uint8 input is not implemented for the vAlexnet variant");
      return TTNET_E_UNSUPPORTED;
    }
    TT_TIMED(pl, "va.stem", s,
             launch_va_stem((const float *)x_dev, (const float *)pl->tensors["features.0.weight"].dev,
                            (const float *)pl->tensors["features.0.bias"].dev, pl->va_scale, pl->va_shift, pl->x_rp[0],
                            (int)n, s));
    return run_va_tail(pl, (int)n, logits_dev, s);
  }
  TT_TIMED(pl, "stem", s,
           launch_stem(x_dev, u8, pl->norm_tab, u8 ? pl->stem_wt_u8 : pl->stem_wt, u8 ? pl->stem_init_u8 : pl->stem_init, pl->x_rp[0], (pl->full || pl->fused || pl->xs) ? nullptr : pl->x_cp[0], (int)n,
                       pl->p, pl->range_dev, s, (pl->lanes.size() >= 2 && !u8 && !pl->full) ? 128 : 256));      // (stem.hip: half the CUs for float32 input with batches in flight)
  return run_from_blocks(pl, (int)n, logits_dev, s);
}

// The forward is a fixed chain of ~11 launches whose host cost (~20 us each) equals the device
// time at batch 256, so from the third call with a given batch size on it is replayed as a
// hipGraph captured on a private stream.  Only two pointers change between calls: the input
// (argument 0 of the first kernel) and the logits (argument 4 of lin2, the last kernel); they
// are patched into the instantiated graph when they differ from the previous call.
constexpr int kLastKernelArgs = 8, kLastKernelOutArg = 4;

void drop_graph(ttnet_plan::GraphEntry &e) {
  if (e.exec) (void)hipGraphExecDestroy(e.exec);
  if (e.graph) (void)hipGraphDestroy(e.graph);
  e = ttnet_plan::GraphEntry{};
}

int invalidate_graphs(ttnet_plan *pl) {
  bool any = false;
  for (auto &l : pl->lanes) any = any || !l.graphs.empty();
  if (any) TT_HIP(hipDeviceSynchronize());
  for (auto &l : pl->lanes) {
    for (auto &kv : l.graphs) {
      drop_graph(kv.second);
      pl->graph_drops++;
    }
    l.graphs.clear();
    l.eager_calls.clear();
  }
  return TTNET_OK;
}

void graphs_off(ttnet_plan *pl, const char *why) {
  pl->graphs_ok = false;
  pl->graph_off_reason = why;
  (void)hipGetLastError();
}

// Copy a kernel node's launch parameters into storage we own.  sizes[i] = byte size of argument i.
// (The arrays returned by hipGraphKernelNodeGetParams belong to the node: they are read once,
// here, and never handed back to the runtime.)
bool own_params(hipGraphNode_t node, const int *sizes, int nargs, hipKernelNodeParams &p, uint64_t *argv, void **args) {
  hipKernelNodeParams q{};
  if (hipGraphKernelNodeGetParams(node, &q) != hipSuccess || !q.kernelParams || q.extra) return false;
  for (int i = 0; i < nargs; ++i) {
    if (!q.kernelParams[i]) return false;
    argv[i] = 0;
    memcpy(&argv[i], q.kernelParams[i], (size_t)sizes[i]);
    args[i] = &argv[i];
  }
  p = q;
  p.kernelParams = args;
  p.extra = nullptr;
  return true;
}

// *status: what forward_eager returned inside the capture (a caller error -- bad argument, range flag -- is reported to
// the caller as such and does not turn graph replay off for the plan; only a failure of the capture machinery does)
bool capture_forward(ttnet_plan *pl, const void *x_dev, bool u8, int64_t n, float *logits_dev, ttnet_plan::GraphEntry &e, int *status) {
  *status = TTNET_OK;
  if (!pl->cap_stream && hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking) != hipSuccess) return false;
  if (hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  const int r = forward_eager(pl, x_dev, u8, n, logits_dev, pl->cap_stream);
  hipGraph_t g = nullptr;
  const hipError_t ee = hipStreamEndCapture(pl->cap_stream, &g);
  if (r != TTNET_OK || ee != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    *status = r;
    return false;
  }
  e.graph = g;
  if (hipGraphInstantiate(&e.exec, g, nullptr, nullptr, 0) != hipSuccess) return false;
  // a single chain: walk from the root to the leaf
  hipGraphNode_t node = nullptr;
  size_t cnt = 1;
  if (hipGraphGetRootNodes(g, &node, &cnt) != hipSuccess || cnt != 1) return false;
  e.first = node;
  for (int guard = 0; guard < 1000; ++guard) {
    size_t nd = 0;
    if (hipGraphNodeGetDependentNodes(node, nullptr, &nd) != hipSuccess) return false;
    if (nd == 0) break;
    if (nd != 1) return false;
    hipGraphNode_t next = nullptr;
    if (hipGraphNodeGetDependentNodes(node, &next, &nd) != hipSuccess) return false;
    node = next;
  }
  e.last = node;
  hipGraphNodeType t0, t1;
  if (hipGraphNodeGetType(e.first, &t0) != hipSuccess || hipGraphNodeGetType(e.last, &t1) != hipSuccess ||
      t0 != hipGraphNodeTypeKernel || t1 != hipGraphNodeTypeKernel || e.first == e.last)
    return false;
  const int *first_sizes = nullptr;                               // stem_pc_kernel's arguments, from the file that declares it
  const int first_n = stem_kernel_arg_sizes(&first_sizes);
  static const int first_sizes_va[7] = {8, 8, 8, 8, 8, 8, 4};     // va_stem_kernel(x, w, bias, scale, shift, rp, n)
  static const int last_sizes[8] = {8, 8, 8, 4, 8, 4, 4, 4};      // lin2_f16x2_kernel(A, B, bias, inv, out, M, N, KS)
  e.first_nargs = pl->va ? 7 : first_n;
  if (e.first_nargs > 12) return false;
  if (!own_params(e.first, pl->va ? first_sizes_va : first_sizes, e.first_nargs, e.first_p, e.first_argv, e.first_args) ||
      !own_params(e.last, last_sizes, kLastKernelArgs, e.last_p, e.last_argv, e.last_args))
    return false;
  // the two slots that will be patched must hold exactly the pointers this capture ran with
  if (e.first_argv[0] != (uint64_t)(uintptr_t)x_dev || e.last_argv[kLastKernelOutArg] != (uint64_t)(uintptr_t)logits_dev ||
      e.last_argv[5] != (uint64_t)n)
    return false;
  e.x = x_dev;
  e.out = logits_dev;
  return true;
}

}  // namespace

int ttnet_plan_set_lanes(ttnet_plan *pl, int lanes) {
  if (!pl || lanes < 1 || lanes > 16) {
    set_error("set_lanes: %d outside [1,16]", lanes);
    return TTNET_E_INVALID;
  }
  (void)hipSetDevice(pl->device);
  const int keep = pl->cur;
  if ((int)pl->lanes.size() < lanes && pl->lanes.size() == 1) TT_TRY(invalidate_graphs(pl));      // (the stem's grid depends on lanes >= 2)
  while ((int)pl->lanes.size() < lanes) {
    // allocate a fresh workspace into the plan's pointers, file it as a new lane, restore
    pl->lanes[pl->cur].last_n = pl->last_n;
    TT_TRY(alloc_workspace(pl));
    pl->lanes.emplace_back();
    store_lane(pl, pl->lanes.back());
    load_lane(pl, pl->lanes[keep]);
  }
  return TTNET_OK;
}

namespace {
int forward_impl(ttnet_plan *pl, int lane, const void *x_dev, bool u8, int64_t n, float *logits_dev, void *stream) {
  TT_TRY(check_ready(pl, x_dev, n, logits_dev));
  // The input contract of ttnet.h (16-byte aligned float32, 4-byte aligned uint8: the stem reads it with 16 / 12-byte
  // buffer loads) is checked HERE, in front of the replay, the capture and the plain path alike: a cached graph
  // only has its first argument re-pointed and would otherwise take any pointer.
  if (!pl->va && ((uintptr_t)x_dev & (u8 ? 3u : 15u)) != 0) {
    set_error("forward: the input must be %d-byte aligned", u8 ? 4 : 16);
    return TTNET_E_INVALID;
  }
  if (((uintptr_t)logits_dev & 3u) != 0) {
    set_error("forward: the logits buffer must be 4-byte aligned");
    return TTNET_E_INVALID;
  }
  if (lane < 0 || lane >= (int)pl->lanes.size()) {
    set_error("forward: lane %d but the plan has %d (ttnet_plan_set_lanes)", lane, (int)pl->lanes.size());
    return TTNET_E_INVALID;
  }
  if (lane != pl->cur) {
    pl->lanes[pl->cur].last_n = pl->last_n;
    switch_lane(pl, lane);
    pl->last_n = pl->lanes[lane].last_n;
  }
  ttnet_plan::Lane &L = pl->lanes[lane];
  hipStream_t s = (hipStream_t)stream;
  if (pl->profiling || !pl->graphs_ok) return forward_eager(pl, x_dev, u8, n, logits_dev, s);
  const int64_t key = 2 * n + (u8 ? 1 : 0);               // one graph per (batch size, input kind)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return forward_eager(pl, x_dev, u8, n, logits_dev, s);   // the caller is capturing us into a graph of their own
  auto it = L.graphs.find(key);
  if (it == L.graphs.end()) {
    if (++L.eager_calls[key] <= 2) return forward_eager(pl, x_dev, u8, n, logits_dev, s);   // warm: attributes, lazy module load
    ttnet_plan::GraphEntry e;
    int cap_status = TTNET_OK;
    if (!capture_forward(pl, x_dev, u8, n, logits_dev, e, &cap_status)) {
      drop_graph(e);
      if (cap_status != TTNET_OK) {          // the forward itself refused the call: the caller's error, graphs stay on
        --L.eager_calls[key];
        return cap_status;
      }
      graphs_off(pl, "capture or instantiation of the forward failed");   // stay on plain launches
      return forward_eager(pl, x_dev, u8, n, logits_dev, s);
    }
    pl->graph_captures++;
    if (L.graphs.size() >= 8) {                                // bound the cache: drop the smallest batch size
      // its last replay may still be running on a stream this call knows nothing about
      TT_HIP(hipDeviceSynchronize());
      drop_graph(L.graphs.begin()->second);
      L.graphs.erase(L.graphs.begin());
      pl->graph_drops++;
    }
    it = L.graphs.emplace(key, e).first;
  }
  ttnet_plan::GraphEntry &e = it->second;
  bool ok = true;
  if (e.x != x_dev) {
    e.first_argv[0] = (uint64_t)(uintptr_t)x_dev;
    e.first_p.kernelParams = e.first_args;          // (the entry may have been moved since capture)
    for (int i = 0; i < e.first_nargs; ++i) e.first_args[i] = &e.first_argv[i];
    ok = hipGraphExecKernelNodeSetParams(e.exec, e.first, &e.first_p) == hipSuccess;
    e.x = x_dev;
  }
  if (ok && e.out != logits_dev) {
    e.last_argv[kLastKernelOutArg] = (uint64_t)(uintptr_t)logits_dev;
    e.last_p.kernelParams = e.last_args;
    for (int i = 0; i < kLastKernelArgs; ++i) e.last_args[i] = &e.last_argv[i];
    ok = hipGraphExecKernelNodeSetParams(e.exec, e.last, &e.last_p) == hipSuccess;
    e.out = logits_dev;
  }
  if (ok) ok = hipGraphLaunch(e.exec, s) == hipSuccess;
  if (!ok) {
    drop_graph(e);
    L.graphs.erase(it);
    graphs_off(pl, "hipGraphExecKernelNodeSetParams / hipGraphLaunch failed");
    return forward_eager(pl, x_dev, u8, n, logits_dev, s);
  }
  pl->last_n = n;
  pl->timing_used = 0;
  pl->graph_replays++;
  return TTNET_OK;
}
}  // namespace

int ttnet_forward_lane(ttnet_plan *pl, int lane, const float *x_dev, int64_t n, float *logits_dev, void *stream) {
  return forward_impl(pl, lane, x_dev, false, n, logits_dev, stream);
}

int ttnet_forward(ttnet_plan *pl, const float *x_dev, int64_t n, float *logits_dev, void *stream) {
  return forward_impl(pl, 0, x_dev, false, n, logits_dev, stream);
}

int ttnet_forward_u8(ttnet_plan *pl, int lane, const uint8_t *x_nhwc_dev, int64_t n, float *logits_dev, void *stream) {
  return forward_impl(pl, lane, x_nhwc_dev, true, n, logits_dev, stream);
}

int ttnet_plan_set_input_norm(ttnet_plan *pl, const float *mean3, const float *std3) {
  if (!pl || !mean3 || !std3 || pl->va) {
    set_error("set_input_norm: null argument (or the vAlexnet variant, which has no uint8 path)");
    return TTNET_E_INVALID;
  }
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] > 0.f)) {
      set_error("set_input_norm: std[%d] = %g", c, (double)std3[c]);
      return TTNET_E_INVALID;
    }
    pl->in_mean[c] = mean3[c];
    pl->in_std[c] = std3[c];
  }
  (void)hipSetDevice(pl->device);
  if (!pl->finalized) return TTNET_OK;                 // (finalize folds them into the uint8 stem's weights)
  TT_HIP(hipDeviceSynchronize());                      // no forward may be reading the buffers that are rewritten in place
  return prepare_stem_u8(pl);
}

int ttnet_forward_from_stem_bits(ttnet_plan *pl, const uint64_t *rows_dev, int64_t n, float *logits_dev,
                                 void *stream) {
  TT_TRY(check_ready(pl, rows_dev, n, logits_dev));
  hipStream_t s = (hipStream_t)stream;
  pl->timing_used = 0;
  const MultiHead &b0 = pl->blocks[0];
  TT_HIP(hipMemcpyAsync(pl->x_rp[0], rows_dev, (size_t)n * b0.C * b0.H * 8, hipMemcpyDeviceToDevice, s));
  if (pl->va) return run_va_tail(pl, (int)n, logits_dev, s);
  if (!pl->full && !pl->xs && !pl->fused) TT_TRY(launch_rp_to_cp(pl->x_rp[0], pl->x_cp[0], (int)n, b0.C, b0.H, b0.W, s));
  return run_from_blocks(pl, (int)n, logits_dev, s);
}

int ttnet_read_stage(ttnet_plan *pl, const char *stage, int64_t n, void *dst, size_t dst_bytes, int on_device,
                     void *stream) {
  if (!pl || !stage || !dst) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  if (n < 1 || n > pl->last_n) {
    set_error("read_stage: n=%lld but the last forward ran %lld images", (long long)n, (long long)pl->last_n);
    return TTNET_E_STATE;
  }
  hipStream_t s = (hipStream_t)stream;
  const std::string st(stage);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  auto copy_out = [&](const void *src, size_t bytes) -> int {
    if (dst_bytes != bytes) {
      set_error("read_stage(%s): destination is %zu bytes, stage is %zu", stage, dst_bytes, bytes);
      return TTNET_E_INVALID;
    }
    TT_HIP(hipMemcpyAsync(dst, src, bytes, kind, s));
    TT_HIP(hipStreamSynchronize(s));
    return check_range(pl);
  };
  if (pl->va) {
    if (st == "features.4") return copy_out(pl->x_rp[0], (size_t)n * 64 * 10 * 8);
    if (st == "features.5") return copy_out(pl->va_y, (size_t)n * 256 * 11 * 8);
    if (st == "flatten") {
      float *tmp = nullptr;
      const size_t elems = (size_t)n * pl->fcsize;
      TT_HIP(hipMalloc((void **)&tmp, elems * 4));
      int r = launch_va_frag_to_flat(pl->feat, tmp, (int)n, s);
      if (r == TTNET_OK) r = copy_out(tmp, elems * 4);
      (void)hipFree(tmp);
      return r;
    }
    set_error("unknown stage %s", stage);
    return TTNET_E_INVALID;
  }
  for (size_t i = 0; i < pl->blocks.size(); ++i) {
    MultiHead &mh = pl->blocks[i];
    const std::string in_name = i == 0 ? std::string("features.3") : pl->blocks[i - 1].name;
    if (st == in_name && pl->fused && i > 0) {          // compact rows -> the uint64 rows of the ABI
      const size_t rows = (size_t)n * mh.C * mh.H;
      uint64_t *tmp = nullptr;
      TT_HIP(hipMalloc((void **)&tmp, rows * 8));
      int r = launch_widen_rows(pl->x_rp[i], tmp, rows, mh.W, s);
      if (r == TTNET_OK) r = copy_out(tmp, rows * 8);
      (void)hipFree(tmp);
      return r;
    }
    if (st == in_name) return copy_out(pl->x_rp[i], (size_t)n * mh.C * mh.H * 8);
    for (int b = 0; b < 4; ++b) {
      if (st == mh.name + ".out" + std::to_string(b + 1) && pl->fused) {
        // the branch tensors never reach HBM on the fused path: a last block's dwords are its output; for
        // the others the block is run once more on its (still resident) input with the tap buffer attached
        const size_t words = (size_t)n * mh.C * mh.Ho, dwords = (size_t)pl->desc.max_batch * (mh.C / 8) * mh.Ho * mh.Wo;
        const uint32_t *src = mh.idx;
        if (!mh.last) {
          if (pl->tap_elems < dwords) {
            uint32_t *t = nullptr;
            TT_TRY(dev_alloc(pl, &t, dwords, true));
            pl->tap = t;
            pl->tap_elems = dwords;
          }
          FusedBlockArgs f{};
          f.n = (int)n; f.C = mh.C; f.H = mh.H; f.Ho = mh.Ho; f.off34 = mh.off34; f.last = 0;
          f.x = pl->x_rp[i]; f.img_c3 = mh.img_c3; f.img_dw = mh.img_dw; f.t_cf = (const uint8_t *)mh.cf.table;
          f.y = pl->x_rp[i + 1]; f.idx = pl->tap;
          TT_TRY(launch_gate_block(f, s));
          src = pl->tap;
        }
        uint64_t *tmp = nullptr;
        TT_HIP(hipMalloc((void **)&tmp, words * 8));
        int r = launch_branch_rows(src, tmp, (int)n, mh.C, mh.Ho, b, s);
        if (r == TTNET_OK) r = copy_out(tmp, words * 8);
        (void)hipFree(tmp);
        return r;
      }
      if (st == mh.name + ".out" + std::to_string(b + 1)) {
        const size_t words = (size_t)n * mh.C * mh.Ho;
        if (pl->xs || pl->full) return copy_out(mh.o[b], words * 8);
        uint64_t *tmp = nullptr;
        TT_HIP(hipMalloc((void **)&tmp, words * 8));
        int r = launch_cp_to_rp(mh.o[b], tmp, (int)n, mh.C, mh.Ho, mh.Wo, s);
        if (r == TTNET_OK) r = copy_out(tmp, words * 8);
        (void)hipFree(tmp);
        return r;
      }
    }
  }
  if (st == "flatten") {
    float *tmp = nullptr;
    const size_t elems = (size_t)n * pl->fcsize;
    TT_HIP(hipMalloc((void **)&tmp, elems * 4));
    int r = launch_frag_to_reference_order(pl->feat, tmp, (int)n, pl->featC / 16, pl->featPP, s);
    if (r == TTNET_OK) r = copy_out(tmp, elems * 4);
    (void)hipFree(tmp);
    return r;
  }
  set_error("unknown stage %s", stage);
  return TTNET_E_INVALID;
}

int ttnet_plan_get_table(ttnet_plan *pl, const char *name, void *dst_host, size_t dst_bytes) {
  if (!pl || !name || !dst_host) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  BlockTT *b = find_block(pl, name);
  if (!b) {
    set_error("no Block_TT named %s", name);
    return TTNET_E_INVALID;
  }
  if (pl->full) {
    set_error("the full variant (fan-in 30) has no truth tables: 2^30 entries per output bit");
    return TTNET_E_UNSUPPORTED;
  }
  if (!pl->finalized) {
    set_error("get_table before finalize");
    return TTNET_E_STATE;
  }
  const BlockGeom &g = b->g;
  const size_t entries = (size_t)1 << g.nbits();
  const size_t need = (size_t)g.groups * entries * g.cout_g() * (g.last ? 4 : 1);
  if (dst_bytes != need) {
    set_error("get_table(%s): destination is %zu bytes, table is %zu", name, dst_bytes, need);
    return TTNET_E_INVALID;
  }
  std::vector<uint8_t> raw(g.table_bytes());
  TT_HIP(hipMemcpy(raw.data(), b->table, raw.size(), hipMemcpyDeviceToHost));
  const int cg = g.cout_g(), eb = g.entry_bits();
  const size_t words_1 = entries >= 32 ? entries / 32 : 1;   // striped 1-bit layout: [grp/16][w][grp%16]
  const uint32_t *raw32 = (const uint32_t *)raw.data();
  for (int grp = 0; grp < g.groups; ++grp)
    for (uint32_t idx = 0; idx < entries; ++idx) {
      const size_t ci = canonical_index(*b, idx);
      if (g.last) {
        memcpy((float *)dst_host + ((size_t)grp * entries + ci) * cg,
               (const float *)raw.data() + ((size_t)grp * entries + idx) * cg, (size_t)cg * 4);
        continue;
      }
      uint32_t bits;
      if (eb == 1) bits = (raw32[((size_t)(grp >> 4) * words_1 + (idx >> 5)) * 16 + (grp & 15)] >> (idx & 31)) & 1u;
      else if (eb == 8) bits = raw[(size_t)grp * entries + idx];
      else bits = ((const uint16_t *)raw.data())[(size_t)grp * entries + idx];
      uint8_t *d = (uint8_t *)dst_host + ((size_t)grp * entries + ci) * cg;
      for (int o = 0; o < cg; ++o) d[o] = (bits >> o) & 1u;
    }
  return TTNET_OK;
}

int ttnet_plan_set_table(ttnet_plan *pl, const char *name, const void *src_host, size_t src_bytes) {
  if (!pl || !name || !src_host) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  BlockTT *b = find_block(pl, name);
  if (!b) {
    set_error("no Block_TT named %s", name);
    return TTNET_E_INVALID;
  }
  if (pl->full) {
    set_error("the full variant (fan-in 30) has no truth tables: 2^30 entries per output bit");
    return TTNET_E_UNSUPPORTED;
  }
  const BlockGeom &g = b->g;
  const size_t entries = (size_t)1 << g.nbits();
  const size_t need = (size_t)g.groups * entries * g.cout_g() * (g.last ? 4 : 1);
  if (src_bytes != need) {
    set_error("set_table(%s): source is %zu bytes, table is %zu", name, src_bytes, need);
    return TTNET_E_INVALID;
  }
  std::vector<uint8_t> raw(g.table_bytes(), 0);
  const int cg = g.cout_g(), eb = g.entry_bits();
  const size_t words_1 = entries >= 32 ? entries / 32 : 1;
  uint32_t *raw32 = (uint32_t *)raw.data();
  for (int grp = 0; grp < g.groups; ++grp)
    for (uint32_t idx = 0; idx < entries; ++idx) {
      const size_t ci = canonical_index(*b, idx);
      if (g.last) {
        memcpy((float *)raw.data() + ((size_t)grp * entries + idx) * cg,
               (const float *)src_host + ((size_t)grp * entries + ci) * cg, (size_t)cg * 4);
        continue;
      }
      const uint8_t *sp = (const uint8_t *)src_host + ((size_t)grp * entries + ci) * cg;
      uint32_t bits = 0;
      for (int o = 0; o < cg; ++o) bits |= (uint32_t)(sp[o] & 1u) << o;
      if (eb == 1) raw32[((size_t)(grp >> 4) * words_1 + (idx >> 5)) * 16 + (grp & 15)] |= bits << (idx & 31);
      else if (eb == 8) raw[(size_t)grp * entries + idx] = (uint8_t)bits;
      else ((uint16_t *)raw.data())[(size_t)grp * entries + idx] = (uint16_t)bits;
    }
  TT_HIP(hipSetDevice(pl->device));
  TT_TRY(invalidate_graphs(pl));
  TT_HIP(hipMemcpy(b->table, raw.data(), raw.size(), hipMemcpyHostToDevice));
  b->user_table = true;
  b->near_ties = -1;
  if (pl->fused)                            // the kernels read images derived from the conv1 / conv2 / conv3 tables
    for (auto &mh : pl->blocks)
      if (b == &mh.c1 || b == &mh.c2 || b == &mh.c3) {
        TT_TRY(launch_fused_images(mh.c1.table, mh.c2.table, mh.c3.table, mh.C, mh.img_dw, mh.img_c3, nullptr));
        TT_HIP(hipDeviceSynchronize());
      }
  return TTNET_OK;
}

int ttnet_plan_query(ttnet_plan *pl, const char *what, int64_t *out) {
  if (!pl || !what || !out) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  const std::string w(what);
  if (w == "fcsize") *out = pl->fcsize;
  else if (w == "n_classes") *out = pl->n_classes;
  else if (w == "n_state_tensors") *out = (int64_t)pl->key_order.size();
  else if (w == "max_batch") *out = pl->desc.max_batch;
  else if (w == "table_bytes") *out = (int64_t)pl->table_bytes;
  else if (w == "workspace_bytes") *out = (int64_t)pl->workspace_bytes;
  else if (w == "p") *out = pl->p;
  else if (w == "graph_replays") *out = pl->graph_replays;
  else if (w == "graphs_enabled") {
    *out = pl->graphs_ok ? 1 : 0;
    if (!pl->graphs_ok)
      set_error("graphs disabled: %s", pl->graph_off_reason.empty() ? "TTNET_NO_GRAPH is set" : pl->graph_off_reason.c_str());   // readable through ttnet_last_error
  }
  else if (w == "graph_captures") *out = pl->graph_captures;
  else if (w == "graph_drops") *out = pl->graph_drops;
  else if (w == "graphs_cached") {
    int64_t c = 0;
    for (auto &l : pl->lanes) c += (int64_t)l.graphs.size();
    *out = c;
  }
  else if (w == "range_overflow") {          // read and clear (synchronises the device: the flag is raised by kernels)
    TT_HIP(hipSetDevice(pl->device));
    TT_HIP(hipDeviceSynchronize());
    *out = *(volatile uint32_t *)pl->range_host ? 1 : 0;
    *pl->range_host = 0u;
  }
  else if (w == "lanes") *out = (int64_t)pl->lanes.size();
  else if (w == "full_listed_pw" || w == "full_listed_dw") {      // full variant: (pixel, group) pairs / outputs sent to float64 so far (current lane)
    uint32_t v[2] = {0, 0};
    if (pl->full_fix) {
      TT_HIP(hipSetDevice(pl->device));
      TT_HIP(hipDeviceSynchronize());
      TT_HIP(hipMemcpy(v, pl->full_fix + 62, sizeof(v), hipMemcpyDeviceToHost));
    }
    *out = v[w == "full_listed_dw" ? 1 : 0];
  }
  else if (w.rfind("near_ties:", 0) == 0) {
    BlockTT *b = find_block(pl, w.c_str() + 10);
    if (!b) {
      set_error("no Block_TT named %s", w.c_str() + 10);
      return TTNET_E_INVALID;
    }
    *out = b->near_ties;
  } else {
    set_error("unknown query %s", what);
    return TTNET_E_INVALID;
  }
  return TTNET_OK;
}

int ttnet_plan_set_profiling(ttnet_plan *pl, int enabled) {
  if (!pl) {
    set_error("null plan");
    return TTNET_E_INVALID;
  }
  pl->profiling = enabled != 0;
  pl->timing_used = 0;
  return TTNET_OK;
}

int ttnet_plan_last_timings(ttnet_plan *pl, const char **names, float *ms, int cap) {
  if (!pl || !names || !ms) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  int k = 0;
  for (size_t i = 0; i < pl->timing_used && k < cap; ++i, ++k) {
    if (hipEventSynchronize(pl->timings[i].e1) != hipSuccess) {
      set_error("hipEventSynchronize failed");
      return TTNET_E_HIP;
    }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, pl->timings[i].e0, pl->timings[i].e1);
    names[k] = pl->timings[i].name;
    ms[k] = t;
  }
  return k;
}

void ttnet_plan_destroy(ttnet_plan *pl) {
  if (!pl) return;
  (void)hipSetDevice(pl->device);
  for (auto &t : pl->timings) {
    (void)hipEventDestroy(t.e0);
    (void)hipEventDestroy(t.e1);
  }
  for (auto &l : pl->lanes)
    for (auto &kv : l.graphs) {
      if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
      if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
  if (pl->cap_stream) (void)hipStreamDestroy(pl->cap_stream);
  if (pl->range_host) (void)hipHostFree(pl->range_host);
  for (void *ptr : pl->owned) (void)hipFree(ptr);
  delete pl;
}

// ---- logits all-gather over RCCL (xGMI) ------------------------------------------------------
// librccl is resolved at first use so that a process which already carries an RCCL (e.g.
// the one inside PyTorch-ROCm) keeps exactly one copy.

struct ttnet_comm {
  void *nccl = nullptr;
  int rank = 0, world = 1, device = 0;
};

namespace {
struct Id128 {
  char b[128];   // ncclUniqueId, passed by value
};
struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.lib) return TTNET_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("cannot load librccl: %s", dlerror());
    return TTNET_E_UNSUPPORTED;
  }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
    set_error("librccl lacks an expected symbol");
    return TTNET_E_UNSUPPORTED;
  }
  g_rccl.lib = h;
  return TTNET_OK;
}

int rccl_check(int r, const char *what) {
  if (r == 0) return TTNET_OK;
  set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
  return TTNET_E_HIP;
}
}  // namespace

int ttnet_comm_unique_id(void *id128) {
  if (!id128) {
    set_error("null argument");
    return TTNET_E_INVALID;
  }
  TT_TRY(load_rccl());
  return rccl_check(g_rccl.GetUniqueId(id128), "ncclGetUniqueId");
}

int ttnet_comm_create(const void *id128, int rank, int world, int device, ttnet_comm **out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
    set_error("bad argument to ttnet_comm_create");
    return TTNET_E_INVALID;
  }
  TT_TRY(load_rccl());
  TT_HIP(hipSetDevice(device));
  std::unique_ptr<ttnet_comm> c(new ttnet_comm());
  c->rank = rank; c->world = world; c->device = device;
  Id128 id;
  memcpy(id.b, id128, 128);
  TT_TRY(rccl_check(g_rccl.CommInitRank(&c->nccl, world, id, rank), "ncclCommInitRank"));
  *out = c.release();
  return TTNET_OK;
}

int ttnet_allgather_logits(ttnet_comm *comm, const float *local_dev, int64_t n_local, int64_t n_classes,
                           float *all_dev, void *stream) {
  if (!comm || !local_dev || !all_dev || n_local < 1 || n_classes < 1) {
    set_error("bad argument to ttnet_allgather_logits");
    return TTNET_E_INVALID;
  }
  // ncclFloat32 == 7
  return rccl_check(g_rccl.AllGather(local_dev, all_dev, (size_t)(n_local * n_classes), 7, comm->nccl,
                                     (hipStream_t)stream),
                    "ncclAllGather");
}

void ttnet_comm_destroy(ttnet_comm *comm) {
  if (!comm) return;
  if (comm->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm->nccl);
  delete comm;
}

}  // extern "C"
