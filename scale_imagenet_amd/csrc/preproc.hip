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
// two passes run as kernels that compute only what the centre crop keeps.
//
// Parity: pinned to Pillow 12.x -- tests/golden/ref_resize.npz holds Pillow's own outputs for seeded images of
// nine geometries (oracle/gen_golden.py resize, run in the build container where Pillow imports) and
// tests/test_gpu_preprocess.py compares these kernels with it byte for byte.  torchvision is not importable:
// its output-size and crop-offset rules are restated.

#include <math.h>

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

__device__ inline uint8_t clip8(int v) {
  v >>= kPrecisionBits;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: rows [row0, row0 + rows) of the input, output columns [x0, x0 + cols) of the resized width
__global__ void resample_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ mid, const int *__restrict__ bounds,
                                  const int *__restrict__ kk, int ksize, int n, int h, int w, int row0, int rows, int x0, int cols) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * rows * cols * 3) return;
  const int ch = (int)(t % 3), xc = (int)((t / 3) % cols), r = (int)((t / (3 * (size_t)cols)) % rows), im = (int)(t / (3 * (size_t)cols * rows));
  const int xx = x0 + xc, xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
  const uint8_t *in = src + (((size_t)im * h + row0 + r) * w + xmin) * 3 + ch;
  const int *k = kk + (size_t)xx * ksize;
  int ss = 1 << (kPrecisionBits - 1);
  for (int x = 0; x < cnt; ++x) ss += (int)in[(size_t)x * 3] * k[x];
  mid[t] = clip8(ss);
}
// vertical pass over the intermediate rows: output rows [y0, y0 + crop) of the resized height
__global__ void resample_v_kernel(const uint8_t *__restrict__ mid, uint8_t *__restrict__ dst, const int *__restrict__ bounds,
                                  const int *__restrict__ kk, int ksize, int n, int row0, int rows, int cols, int y0, int crop_h) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * crop_h * cols * 3) return;
  const int e = (int)(t % (3 * (size_t)cols)), yc = (int)((t / (3 * (size_t)cols)) % crop_h), im = (int)(t / (3 * (size_t)cols * crop_h));
  const int yy = y0 + yc, ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
  const uint8_t *in = mid + ((size_t)im * rows + (ymin - row0)) * cols * 3 + e;
  const int *k = kk + (size_t)yy * ksize;
  int ss = 1 << (kPrecisionBits - 1);
  for (int y = 0; y < cnt; ++y) ss += (int)in[(size_t)y * cols * 3] * k[y];
  dst[t] = clip8(ss);
}
__global__ void crop_only_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int n, int h, int w, int y0, int x0,
                                 int crop) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)n * crop * crop * 3) return;
  const int e = (int)(t % (3 * (size_t)crop)), yc = (int)((t / (3 * (size_t)crop)) % crop), im = (int)(t / (3 * (size_t)crop * crop));
  dst[t] = src[(((size_t)im * h + y0 + yc) * w + x0) * 3 + e];
}

int round_half_even(double v) { return (int)nearbyint(v); }     // Python's round(), default rounding mode

}  // namespace

}  // namespace ttnet

using namespace ttnet;

extern "C" int ttnet_resize_center_crop_u8(const uint8_t *src_dev, int64_t n, int h, int w, int resize, int crop, uint8_t *dst_dev,
                                           void *stream) {
  if (!src_dev || !dst_dev || n < 1 || h < 1 || w < 1 || resize < 1 || crop < 1 || n > (1 << 20)) {
    set_error("resize_center_crop: bad argument");
    return TTNET_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
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
  // CenterCrop: int(round((H - crop) / 2.0))
  const int y0 = round_half_even((nh - crop) / 2.0), x0 = round_half_even((nw - crop) / 2.0);
  const bool need_h = nw != w, need_v = nh != h;
  if (!need_h && !need_v) {
    const size_t t = (size_t)n * crop * crop * 3;
    hipLaunchKernelGGL(crop_only_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src_dev, dst_dev, (int)n, h, w, y0, x0, crop);
    TT_HIP(hipGetLastError());
    return TTNET_OK;
  }
  const Coeffs ch = precompute(w, nw), cv = precompute(h, nh);
  // input rows the kept output rows depend on
  int row0 = 0, rows = h;
  if (need_v) {
    row0 = cv.bounds[2 * y0];
    const int last = y0 + crop - 1;
    rows = cv.bounds[2 * last] + cv.bounds[2 * last + 1] - row0;
  } else {
    row0 = y0;
    rows = crop;
  }
  int *d_tab = nullptr;
  const size_t nb_h = ch.bounds.size(), nk_h = ch.kk.size(), nb_v = cv.bounds.size(), nk_v = cv.kk.size();
  std::vector<int> host(nb_h + nk_h + nb_v + nk_v);
  std::copy(ch.bounds.begin(), ch.bounds.end(), host.begin());
  std::copy(ch.kk.begin(), ch.kk.end(), host.begin() + nb_h);
  std::copy(cv.bounds.begin(), cv.bounds.end(), host.begin() + nb_h + nk_h);
  std::copy(cv.kk.begin(), cv.kk.end(), host.begin() + nb_h + nk_h + nb_v);
  uint8_t *mid = nullptr;
  const size_t mid_bytes = (size_t)n * rows * crop * 3;
  TT_HIP(hipMalloc((void **)&d_tab, host.size() * sizeof(int)));
  if (hipMalloc((void **)&mid, mid_bytes) != hipSuccess) {
    (void)hipFree(d_tab);
    set_error("resize_center_crop: hipMalloc(%zu) failed", mid_bytes);
    return TTNET_E_NOMEM;
  }
  int st = TTNET_OK;
  do {
    if (hipMemcpyAsync(d_tab, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) { st = TTNET_E_HIP; break; }
    const int *bh = d_tab, *kh = d_tab + nb_h, *bv = d_tab + nb_h + nk_h, *kv = d_tab + nb_h + nk_h + nb_v;
    const uint8_t *vin = mid;
    if (need_h) {
      const size_t t = (size_t)n * rows * crop * 3;
      hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src_dev, mid, bh, kh, ch.ksize, (int)n, h, w,
                         row0, rows, x0, crop);
    } else {            // only a vertical pass: gather the kept columns of the needed rows
      const size_t t = (size_t)n * rows * crop * 3;
      // (rows x crop window at (row0, x0): crop_only_kernel with a rectangular window = two calls' worth of
      // index arithmetic; the horizontal kernel with identity coefficients would round the same bytes)
      std::vector<int> idb((size_t)nw * 2), idk((size_t)nw, 1 << kPrecisionBits);
      for (int i = 0; i < nw; ++i) { idb[2 * i] = i; idb[2 * i + 1] = 1; }
      int *d_id = nullptr;
      if (hipMalloc((void **)&d_id, (idb.size() + idk.size()) * sizeof(int)) != hipSuccess) { st = TTNET_E_NOMEM; break; }
      (void)hipMemcpyAsync(d_id, idb.data(), idb.size() * sizeof(int), hipMemcpyHostToDevice, s);
      (void)hipMemcpyAsync(d_id + idb.size(), idk.data(), idk.size() * sizeof(int), hipMemcpyHostToDevice, s);
      hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, src_dev, mid, d_id, d_id + idb.size(), 1,
                         (int)n, h, w, row0, rows, x0, crop);
      (void)hipStreamSynchronize(s);
      (void)hipFree(d_id);
    }
    if (need_v) {
      const size_t t = (size_t)n * crop * crop * 3;
      hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, vin, dst_dev, bv, kv, cv.ksize, (int)n, row0,
                         rows, crop, y0, crop);
    } else {
      if (hipMemcpyAsync(dst_dev, mid, mid_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { st = TTNET_E_HIP; break; }
    }
    if (hipGetLastError() != hipSuccess) { st = TTNET_E_HIP; break; }
  } while (0);
  // the tables and the intermediate image are freed once the stream has consumed them
  (void)hipStreamSynchronize(s);
  (void)hipFree(mid);
  (void)hipFree(d_tab);
  if (st != TTNET_OK) set_error("resize_center_crop: a HIP call failed");
  return st;
}
