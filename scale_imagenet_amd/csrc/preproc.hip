// The eval input transform in front of ttnet_forward_u8, on the GPU:
//   transforms.Resize(256) -> transforms.CenterCrop(224)      (utils/preprocess.py:104-105, main.py:208)
// on decoded uint8 HWC images; ToTensor + Normalize (:106-108) are fused into the stem (stem.hip).
//
// torchvision's Resize of a PIL image is Pillow's Image.resize(..., BILINEAR): a separable
// convolution whose support widens with the down-scaling factor (antialiasing), carried out on uint8
// with 22-bit fixed-point coefficients, horizontal pass first, the intermediate image rounded to
// uint8 (Pillow, src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc).  This file restates that arithmetic exactly
// (integer for integer); the coefficient tables are built on the host in float64 like Pillow's, the
// two passes run in one kernel that computes only what the centre crop keeps (the intermediate image lives in LDS).
//
// Parity: pinned to Pillow 12.x -- tests/golden/ref_resize.npz holds Pillow's own outputs for seeded images of
// nine geometries (oracle/gen_golden.py resize, run in the build container where Pillow imports) and
// tests/test_gpu_preprocess.py compares these kernels with it byte for byte.  torchvision is not importable:
// its output-size and crop-offset rules are restated.

#include <math.h>

#include <algorithm>

#include <mutex>
#include <vector>

#include "ttnet_common.h"

namespace ttnet {

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

struct Coeffs {
  int ksize = 0;
  std::vector<int> bounds;      // [out][2]: first input index, tap count
  std::vector<int> kk;          // [out][ksize] fixed point
};

// Resample.c: precompute_coeffs (bilinear: support 1) + normalize_coeffs_8bpc
Coeffs precompute(int in_size, int out_size) {
  Coeffs c;
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  c.ksize = (int)ceil(support) * 2 + 1;
  c.bounds.resize((size_t)out_size * 2);
  c.kk.assign((size_t)out_size * c.ksize, 0);
  std::vector<double> k(c.ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double w = t < 1.0 ? 1.0 - t : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      c.kk[(size_t)xx * c.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    }
    c.bounds[2 * xx] = xmin;
    c.bounds[2 * xx + 1] = xmax;
  }
  return c;
}

// clip8 of Resample.c: (v >> 22) saturated to 0..255.  Written as clamp-then-shift: from the shift-then-clamp form hipcc 7.2
// selects gfx950's v_ashr_pk_u8_i32 for two of the four bytes of a packed dword and ORs the other two into the upper half of its
// result, which is not zero there -- bytes 2 and 3 of some dwords came out wrong (tests/test_gpu_preprocess.py caught it).
__device__ inline uint32_t clip8(int v) {
  const int hi = (256 << kPrecisionBits) - 1;
  v = v < 0 ? 0 : (v > hi ? hi : v);
  return (uint32_t)v >> kPrecisionBits;
}

// One launch for both passes (round 3).  A workgroup makes TY output rows of one image: it stages the input rows those rows
// depend on, chunk by chunk, as 16-byte pieces in LDS (only the columns the kept output columns depend on; a thread's loads of a
// chunk are issued together), runs the horizontal pass of a chunk into an LDS image of the intermediate rows -- rounded to
// uint8 exactly as Pillow stores its intermediate image -- and then the vertical pass from that image, four output bytes per
// thread and store.  Both passes loop over the table's tap count (uniform; coefficients beyond a window's own count are zero)
// with the four bytes of a dword side by side, so that eight LDS reads are in flight per step.  Input bytes are read once (plus
// the few rows two neighbouring tiles share), the intermediate image never leaves the chip, nothing is allocated or waited for
// on the host: the call is asynchronous on its stream.  (Round 2: two kernels with a byte per thread, a hipMalloc'ed
// intermediate image, a table upload and a stream synchronisation per call: 615 us for 256 images of 375 x 500; now see
// tools/preproc_bench.py.)
constexpr int RS_TY = 16, RS_CHUNK = 16, RS_THREADS = 256, RS_STAGE_LOADS = 6;      // (a chunk: at most RS_CHUNK rows and RS_STAGE_LOADS x RS_THREADS pieces)
struct ResizeArgs {
  const uint8_t *src;
  uint8_t *dst;
  const int *bh, *kh, *bv, *kv;      // bounds [out][2] and coefficients [out][ksize] of the two passes (device)
  int n, h, w, crop, x0, y0, ksh, ksv;
  int c0, cw;                        // first input column and number of input columns the kept output columns depend on
  int rmax;                          // most intermediate rows a tile needs
  int row_q;                         // 16-byte pieces per staged input row (cw * 3 bytes + alignment slack)
  int chunk;                         // input rows staged at a time
};
// KSH / KSV: the tap counts of the two tables when they are one of 1, 3, 5, 7, 9 (the loops unroll and the byte reads of a
// window get immediate offsets); 0 = any count, at run time.
template <int KSH, int KSV>
__global__ __launch_bounds__(RS_THREADS) void resize_crop_kernel(ResizeArgs a) {
  extern __shared__ __align__(16) uint8_t lds[];
  const int im = blockIdx.y, ty = blockIdx.x, tid = threadIdx.x;
  const int ksh = KSH ? KSH : a.ksh, ksv = KSV ? KSV : a.ksv;
  const int ow = a.crop * 3, owq = ow >> 2;                // bytes / dwords per output (and intermediate) row
  const float inv_owq = 1.0f / (float)owq, inv_rowq = 1.0f / (float)a.row_q, inv_crop = 1.0f / (float)a.crop;
  const int ya = a.y0 + ty * RS_TY, rows_out = min(RS_TY, a.crop - ty * RS_TY);
  const int r0 = a.bv[2 * ya], rend = a.bv[2 * (ya + rows_out - 1)] + a.bv[2 * (ya + rows_out - 1) + 1], R = rend - r0;
  // LDS: [staged input chunk (16-byte aligned)][intermediate rows][tables of both passes]
  uint4 *s_in = (uint4 *)lds;                                                  // [chunk][row_q]
  uint8_t *s_mid = lds + (size_t)a.chunk * a.row_q * 16;                       // [rmax][ow]
  int *s_xb = (int *)(s_mid + (size_t)a.rmax * ow);                            // [crop] byte offset of a window's first tap  (ow % 4 == 0)
  int *s_kh = s_xb + a.crop;                                                   // [crop][ksh]
  int *s_bv = s_kh + a.crop * ksh;                                             // [RS_TY] first intermediate row of a window
  int *s_kv = s_bv + RS_TY;                                                    // [RS_TY][ksv]
  for (int i = tid; i < a.crop; i += RS_THREADS) s_xb[i] = (a.bh[2 * (a.x0 + i)] - a.c0) * 3;
  for (int i = tid; i < a.crop * ksh; i += RS_THREADS) s_kh[i] = a.kh[(size_t)a.x0 * ksh + i];
  for (int i = tid; i < rows_out; i += RS_THREADS) s_bv[i] = a.bv[2 * (ya + i)] - r0;
  for (int i = tid; i < rows_out * ksv; i += RS_THREADS) s_kv[i] = a.kv[(size_t)ya * ksv + i];
  const size_t total = (size_t)a.n * a.h * a.w * 3, whole_q = total >> 4;
  const uint32_t pitch = (uint32_t)a.w * 3u;               // bytes per input row
  for (int rc = 0; rc < R; rc += a.chunk) {
    const int nr = min(a.chunk, R - rc);
    const size_t first0 = (((size_t)im * a.h + r0 + rc) * a.w + a.c0) * 3;                   // byte address of (first row of the chunk, c0, channel 0)
    const uint32_t al0 = (uint32_t)(first0 & 15);
    __syncthreads();                                       // the previous chunk has been consumed (and the tables are in place)
    // stage: 16-byte pieces from the piece that holds byte (row, c0); all of a thread's loads first
    uint4 v[RS_STAGE_LOADS];
#pragma unroll
    for (int u = 0; u < RS_STAGE_LOADS; ++u) {
      const int i = tid + RS_THREADS * u;
      v[u] = make_uint4(0u, 0u, 0u, 0u);
      if (i < nr * a.row_q) {
        const int rr = (int)(((float)i + 0.5f) * inv_rowq), d = i - rr * a.row_q;
        const size_t q = ((first0 + (size_t)((uint32_t)rr * pitch)) >> 4) + d;
        if (q < whole_q) v[u] = ((const uint4 *)a.src)[q];
        else if (q == whole_q) {                            // (never a byte beyond the buffer: its last, partial piece byte by byte)
          uint32_t t[4] = {0u, 0u, 0u, 0u};
          for (size_t b = 16 * whole_q; b < total; ++b) t[(b & 15) >> 2] |= (uint32_t)a.src[b] << (8 * (b & 3));
          v[u] = make_uint4(t[0], t[1], t[2], t[3]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RS_STAGE_LOADS; ++u) {
      const int i = tid + RS_THREADS * u;
      if (i < nr * a.row_q) s_in[i] = v[u];
    }
    __syncthreads();
    // horizontal pass: one intermediate pixel (three bytes) per thread and step.  Taps beyond a window's own count carry a
    // zero coefficient, and what they read still lies inside the LDS allocation (the staged chunk is followed by s_mid).
    for (int i = tid; i < nr * a.crop; i += RS_THREADS) {
      const int rr = (int)(((float)i + 0.5f) * inv_crop), xc = i - rr * a.crop;
      const uint8_t *p = (const uint8_t *)s_in + (uint32_t)rr * (uint32_t)(a.row_q * 16) + ((al0 + (uint32_t)rr * pitch) & 15u) + s_xb[xc];
      const int *k = s_kh + xc * ksh;
      int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
      // (24-bit multiplies: a byte times a coefficient below 2^23 -- the full-rate v_mad_i32_i24, not the quarter-rate 32-bit one)
      if constexpr (KSH != 0) {
#pragma unroll
        for (int x = 0; x < KSH; ++x) {
          const int kx = k[x];
          a0 += __mul24((int)p[3 * x], kx);
          a1 += __mul24((int)p[3 * x + 1], kx);
          a2 += __mul24((int)p[3 * x + 2], kx);
        }
      } else {
#pragma nounroll
        for (int x = 0; x < ksh; ++x) {
          const int kx = k[x];
          a0 += __mul24((int)p[3 * x], kx);
          a1 += __mul24((int)p[3 * x + 1], kx);
          a2 += __mul24((int)p[3 * x + 2], kx);
        }
      }
      uint8_t *o = s_mid + (size_t)(rc + rr) * ow + 3 * xc;
      o[0] = (uint8_t)clip8(a0);
      o[1] = (uint8_t)clip8(a1);
      o[2] = (uint8_t)clip8(a2);
    }
  }
  __syncthreads();
  // vertical pass: four consecutive output bytes per thread, one dword store
  for (int i = tid; i < rows_out * owq; i += RS_THREADS) {
    const int yc = (int)(((float)i + 0.5f) * inv_owq), q = i - yc * owq;
    const int ymin = s_bv[yc];
    const int *k = s_kv + yc * ksv;
    int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0, s3 = s0;
    auto tap = [&](int y) {
      const uint32_t v = ((const uint32_t *)(s_mid + (size_t)min(ymin + y, R - 1) * ow))[q];
      const int kvy = k[y];
      s0 += __mul24((int)(v & 255u), kvy);
      s1 += __mul24((int)((v >> 8) & 255u), kvy);
      s2 += __mul24((int)((v >> 16) & 255u), kvy);
      s3 += __mul24((int)(v >> 24), kvy);
    };
    if constexpr (KSV != 0) {
#pragma unroll
      for (int y = 0; y < KSV; ++y) tap(y);
    } else {
#pragma nounroll
      for (int y = 0; y < ksv; ++y) tap(y);
    }
    const uint32_t word = clip8(s0) | (clip8(s1) << 8) | (clip8(s2) << 16) | (clip8(s3) << 24);
    ((uint32_t *)(a.dst + ((size_t)im * a.crop + ty * RS_TY + yc) * ow))[q] = word;
  }
}

int round_half_even(double v) { return (int)nearbyint(v); }     // Python's round(), default rounding mode

}  // namespace

}  // namespace ttnet

using namespace ttnet;

// Device copies of the coefficient tables, per geometry and device: built once, kept for the life of the process (a few KB each).
namespace {
struct ResizeGeo {
  int *tab = nullptr;
  size_t nb_h = 0, nk_h = 0, nb_v = 0, nk_v = 0;
  int ksh = 0, ksv = 0, nw = 0, nh = 0, x0 = 0, y0 = 0, c0 = 0, cw = 0, rmax = 0;
};
std::mutex g_geo_mutex;
std::vector<std::pair<std::vector<int>, ResizeGeo>> g_geos;      // key: device, h, w, resize, crop
Coeffs identity(int size) {
  Coeffs c;
  c.ksize = 1;
  c.bounds.resize((size_t)size * 2);
  c.kk.assign((size_t)size, 1 << kPrecisionBits);
  for (int i = 0; i < size; ++i) { c.bounds[2 * i] = i; c.bounds[2 * i + 1] = 1; }
  return c;
}
}  // namespace

extern "C" int ttnet_resize_center_crop_u8(const uint8_t *src_dev, int64_t n, int h, int w, int resize, int crop, uint8_t *dst_dev,
                                           void *stream) {
  if (!src_dev || !dst_dev || n < 1 || h < 1 || w < 1 || resize < 1 || crop < 1 || n > 65535) {
    set_error("resize_center_crop: bad argument");
    return TTNET_E_INVALID;
  }
  if (((uintptr_t)src_dev & 15) || ((uintptr_t)dst_dev & 3) || (crop * 3) % 4) {
    set_error("resize_center_crop: the input must be 16-byte aligned, the output 4-byte aligned and crop * 3 a multiple of 4");
    return TTNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  int dev = 0;
  TT_HIP(hipGetDevice(&dev));
  ResizeGeo g;
  {
    std::lock_guard<std::mutex> lock(g_geo_mutex);
    const std::vector<int> key = {dev, h, w, resize, crop};
    bool found = false;
    for (auto &kv : g_geos)
      if (kv.first == key) { g = kv.second; found = true; break; }
    if (!found) {
      // torchvision.transforms.functional.resize with an int size: the shorter side becomes `resize`,
      // the longer one int(resize * long / short); an image whose shorter side already matches is kept
      int nw = w, nh = h;
      if (!((w <= h && w == resize) || (h <= w && h == resize))) {
        if (w <= h) { nw = resize; nh = (int)((double)resize * h / w); }
        else { nh = resize; nw = (int)((double)resize * w / h); }
      }
      if (nw < crop || nh < crop) {
        set_error("resize_center_crop: %dx%d resized to %dx%d is smaller than the %d crop (torchvision would pad)", w, h, nw, nh, crop);
        return TTNET_E_UNSUPPORTED;
      }
      // CenterCrop: int(round((H - crop) / 2.0)); a pass Pillow skips (size unchanged) is the identity table: the same bytes
      g.nw = nw; g.nh = nh;
      g.y0 = round_half_even((nh - crop) / 2.0);
      g.x0 = round_half_even((nw - crop) / 2.0);
      const Coeffs ch = nw != w ? precompute(w, nw) : identity(w), cv = nh != h ? precompute(h, nh) : identity(h);
      g.ksh = ch.ksize; g.ksv = cv.ksize;
      g.c0 = ch.bounds[2 * g.x0];
      g.cw = ch.bounds[2 * (g.x0 + crop - 1)] + ch.bounds[2 * (g.x0 + crop - 1) + 1] - g.c0;
      for (int t0 = 0; t0 < crop; t0 += RS_TY) {
        const int last = g.y0 + std::min(crop, t0 + RS_TY) - 1;
        g.rmax = std::max(g.rmax, cv.bounds[2 * last] + cv.bounds[2 * last + 1] - cv.bounds[2 * (g.y0 + t0)]);
      }
      g.nb_h = ch.bounds.size(); g.nk_h = ch.kk.size(); g.nb_v = cv.bounds.size(); g.nk_v = cv.kk.size();
      std::vector<int> host(g.nb_h + g.nk_h + g.nb_v + g.nk_v);
      std::copy(ch.bounds.begin(), ch.bounds.end(), host.begin());
      std::copy(ch.kk.begin(), ch.kk.end(), host.begin() + g.nb_h);
      std::copy(cv.bounds.begin(), cv.bounds.end(), host.begin() + g.nb_h + g.nk_h);
      std::copy(cv.kk.begin(), cv.kk.end(), host.begin() + g.nb_h + g.nk_h + g.nb_v);
      TT_HIP(hipMalloc((void **)&g.tab, host.size() * sizeof(int)));
      TT_HIP(hipMemcpy(g.tab, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));      // (synchronous: once per geometry)
      g_geos.emplace_back(key, g);
    }
  }
  ResizeArgs a{};
  a.src = src_dev; a.dst = dst_dev;
  a.bh = g.tab; a.kh = g.tab + g.nb_h; a.bv = g.tab + g.nb_h + g.nk_h; a.kv = g.tab + g.nb_h + g.nk_h + g.nb_v;
  a.n = (int)n; a.h = h; a.w = w; a.crop = crop; a.x0 = g.x0; a.y0 = g.y0; a.ksh = g.ksh; a.ksv = g.ksv;
  a.c0 = g.c0; a.cw = g.cw; a.rmax = g.rmax;
  a.row_q = (g.cw * 3 + 15 + 15) / 16;                    // cw * 3 bytes starting up to 15 bytes into the first piece
  a.chunk = std::min(RS_CHUNK, (RS_STAGE_LOADS * RS_THREADS) / a.row_q);
  const size_t lds = (size_t)a.chunk * a.row_q * 16 + (size_t)g.rmax * crop * 3 +
                     (size_t)(crop + crop * g.ksh + RS_TY + RS_TY * g.ksv) * sizeof(int);
  if (lds > 160 * 1024 || a.chunk < 1) {
    set_error("resize_center_crop: %dx%d -> crop %d needs %zu bytes of LDS per workgroup", w, h, crop, lds);
    return TTNET_E_UNSUPPORTED;
  }
  auto launch = [&](auto kernel) -> int {
    TT_TRY(ensure_dynamic_lds((const void *)kernel, lds));
    hipLaunchKernelGGL(kernel, dim3((crop + RS_TY - 1) / RS_TY, (unsigned)n), dim3(RS_THREADS), lds, s, a);
    return TTNET_OK;
  };
  // the common pairs of tap counts get unrolled loops (the aspect is kept, so both passes have the same scale and count)
  if (g.ksh == 1 && g.ksv == 1) TT_TRY(launch(resize_crop_kernel<1, 1>));
  else if (g.ksh == 3 && g.ksv == 3) TT_TRY(launch(resize_crop_kernel<3, 3>));
  else if (g.ksh == 5 && g.ksv == 5) TT_TRY(launch(resize_crop_kernel<5, 5>));
  else if (g.ksh == 7 && g.ksv == 7) TT_TRY(launch(resize_crop_kernel<7, 7>));
  else if (g.ksh == 9 && g.ksv == 9) TT_TRY(launch(resize_crop_kernel<9, 9>));
  else TT_TRY(launch(resize_crop_kernel<0, 0>));
  TT_HIP(hipGetLastError());
  return TTNET_OK;
}
