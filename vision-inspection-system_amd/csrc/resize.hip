// Row f3: the service-side bicubic resample of the decoded frame, on the GPU, BIT-EXACT
// with Pillow's ImagingResample for 8-bit images - the resampler behind the HF Qwen2-VL image processor's
// resize(resample=BICUBIC) (TF:models/qwen2_vl/image_processing_qwen2_vl.py:62-89,:165-198 via
// transformers.image_transforms.resize -> PIL.Image.resize), which the host did before (image_processing.py).
//
// Pillow's algorithm (Pillow 12.2, src/libImaging/Resample.c; restated, not copied): two separable passes,
// horizontal first, with a uint8 intermediate image.  For every output coordinate the host precomputes (in
// float64, same operation order as the C code) a window [xmin, xmin+n) and n normalised filter weights,
// converted to fixed point with 22 fractional bits (round half away from zero).  A pass then computes, per
// channel,  clip8((2^21 + sum_i pixel[xmin+i] * k[i]) >> 22)  in 32-bit integer arithmetic - exactly what
// these kernels do, so the result does not depend on the device at all.
// Integer/byte work, HBM-trivial (3 MB in, 2.9 MB out): one thread per output pixel, coalesced 3-byte
// neighbours, coefficient rows read through the scalar/L1 path.
#include "common.hip.h"

#define RS_PRECISION_BITS 22

__device__ __forceinline__ uint8_t rs_clip8(int v) {
  v >>= RS_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [in_h][in_w][3] -> tmp [in_h][out_w][3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                       int in_h, int in_w, int out_w, const int* __restrict__ kk,
                                                       const int* __restrict__ bounds, int ksize) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)in_h * out_w) return;
  const int row = (int)(idx / out_w), xo = (int)(idx - (long long)row * out_w);
  const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int* k = kk + (size_t)xo * ksize;
  const uint8_t* p = src + ((size_t)row * in_w + xmin) * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int c = k[x];
    s0 += p[3 * x] * c;
    s1 += p[3 * x + 1] * c;
    s2 += p[3 * x + 2] * c;
  }
  uint8_t* o = tmp + idx * 3;
  o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
}

// vertical pass: tmp [in_h][out_w][3] -> dst [out_h][out_w][3]
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                       int out_h, int out_w, const int* __restrict__ kk,
                                                       const int* __restrict__ bounds, int ksize) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)out_h * out_w) return;
  const int yo = (int)(idx / out_w), x = (int)(idx - (long long)yo * out_w);
  const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int* k = kk + (size_t)yo * ksize;
  const uint8_t* p = tmp + ((size_t)ymin * out_w + x) * 3;
  const size_t stride = (size_t)out_w * 3;
  int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int c = k[y];
    s0 += p[0] * c;
    s1 += p[1] * c;
    s2 += p[2] * c;
    p += stride;
  }
  uint8_t* o = dst + idx * 3;
  o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
}

// kx/bx: horizontal coefficients int32 [out_w][ksx] and windows int32 [out_w][2] = {first input column, count};
// ky/by likewise for rows.  tmp: u8 [in_h][out_w][3].  All pointers are device pointers.
extern "C" int vis_resize_rgb_u8(const void* src, void* tmp, void* dst, int in_h, int in_w, int out_h, int out_w,
                                 const void* kx, const void* bx, int ksx, const void* ky, const void* by, int ksy,
                                 hipStream_t stream) {
  if (!src || !tmp || !dst || !kx || !bx || !ky || !by) return VIS_ERR_ARG;
  if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || ksx <= 0 || ksy <= 0) return VIS_ERR_ARG;
  if ((long long)in_h * out_w > (1LL << 31) || (long long)out_h * out_w > (1LL << 31)) return VIS_ERR_ARG;
  if (((uintptr_t)kx | (uintptr_t)bx | (uintptr_t)ky | (uintptr_t)by) & 3) return VIS_ERR_ARG;
  vis_clear_error();
  const long long nh = (long long)in_h * out_w, nv = (long long)out_h * out_w;
  hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)src,
                     (uint8_t*)tmp, in_h, in_w, out_w, (const int*)kx, (const int*)bx, ksx);
  hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, stream, (const uint8_t*)tmp,
                     (uint8_t*)dst, out_h, out_w, (const int*)ky, (const int*)by, ksy);
  return vis_check_launch();
}
