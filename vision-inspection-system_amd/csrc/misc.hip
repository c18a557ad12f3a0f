// K1 front half and K12 glue for gfx950: patch extraction (u8 frame -> normalised
// bf16 patch rows), embedding row gather, image-token row scatter.
// All HBM-bound row movers; 16-byte accesses per lane.
#include "common.hip.h"

// ---------------------------------------------------------------------------
// vis_patchify_u8: the resized RGB frame [H][W][3] u8 (H, W multiples of 28)
// becomes pixel_values rows in exactly the layout of
// TF:models/qwen2_vl/image_processing_pil_qwen2_vl.py:156-190 (patchify):
//   row  p = ((gh/2)*(GW/2) + gw/2)*4 + (gh%2)*2 + (gw%2)      (2x2 merge groups)
//   col  f = ((c*2 + t)*14 + ph)*14 + pw, t = temporal copy (both frames equal)
// with rescale 1/255 and CLIP mean/std normalisation fused in, rounded to bf16,
// and the row zero-padded from 1176 to ld_out (a multiple of the GEMM K-step)
// so the patch-embed conv (TF:...modeling_qwen2_vl.py:251-274) is one K2 GEMM.
struct PatchArgs {
  const uint8_t* img;
  bf16_t* out;
  int H, W, ld_out, row0;
  float mean[3], istd[3];
};

__global__ __launch_bounds__(256) void patchify_u8_kernel(PatchArgs p) {
  constexpr int P = 14, F = 3 * 2 * P * P;  // 1176
  const int GW = p.W / P;
  const int chunks = p.ld_out >> 3;
  const int np = (p.H / P) * GW;
  const long long total = (long long)np * chunks;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int patch = (int)(it / chunks), ch = (int)(it - (long long)patch * chunks);
    const int grp = patch >> 2, mh = (patch >> 1) & 1, mw = patch & 1;
    const int gh = (grp / (GW >> 1)) * 2 + mh, gw = (grp % (GW >> 1)) * 2 + mw;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = ch * 8 + e;
      float v = 0.f;
      if (f < F) {
        const int c = f / (2 * P * P);
        const int rem = f - c * (2 * P * P);
        const int pp = rem % (P * P);  // temporal index dropped: both frames are the same image
        const int ph = pp / P, pw = pp - ph * P;
        const int y = gh * P + ph, x = gw * P + pw;
        const float u = (float)p.img[((size_t)y * p.W + x) * 3 + c];
        v = (u * (1.0f / 255.0f) - p.mean[c]) * p.istd[c];
      }
      o[e] = v;
    }
    *(u32x4*)(p.out + (size_t)(p.row0 + patch) * p.ld_out + ch * 8) = pack8(o);
  }
}

extern "C" int vis_patchify_u8(const void* img, void* out, int H, int W, int ld_out, int row0, const float* mean,
                               const float* stdv, hipStream_t stream) {
  if (!img || !out || !mean || !stdv || H <= 0 || W <= 0) return VIS_ERR_ARG;
  if (H % 28 != 0 || W % 28 != 0 || ld_out % 8 != 0 || ld_out < 1176 || row0 < 0) return VIS_ERR_ARG;
  if ((uintptr_t)out & 15) return VIS_ERR_ARG;
  PatchArgs p;
  p.img = (const uint8_t*)img; p.out = (bf16_t*)out; p.H = H; p.W = W; p.ld_out = ld_out; p.row0 = row0;
  for (int c = 0; c < 3; ++c) {
    if (!(stdv[c] > 0.f)) return VIS_ERR_ARG;
    p.mean[c] = mean[c];
    p.istd[c] = 1.0f / stdv[c];
  }
  const long long total = (long long)(H / 14) * (W / 14) * (ld_out / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  vis_clear_error();
  hipLaunchKernelGGL(patchify_u8_kernel, dim3(blocks), dim3(256), 0, stream, p);
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// vis_gather_rows:  out[i][:] = table[ids[i]][:]       (embedding lookup; ids on device)
// vis_scatter_rows: dst[idx[i]][:] = src[i][:]         (masked_scatter of the merger output
//                   into the text embeddings, TF:...modeling_qwen2_vl.py:1144-1200)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ table,
                                                          const int* __restrict__ ids, bf16_t* __restrict__ out,
                                                          int n, int D, int n_table) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= n_table ? n_table - 1 : id);
  const bf16_t* src = table + (size_t)id * D;
  bf16_t* dst = out + (size_t)row * D;
  for (int c = lane; c < (D >> 3); c += 64) *(u32x4*)(dst + c * 8) = *(const u32x4*)(src + c * 8);
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src,
                                                           const int* __restrict__ idx, bf16_t* __restrict__ dst,
                                                           int n, int D, int n_dst) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  const int t = idx[row];
  if (t < 0 || t >= n_dst) return;
  const bf16_t* s = src + (size_t)row * D;
  bf16_t* d = dst + (size_t)t * D;
  for (int c = lane; c < (D >> 3); c += 64) *(u32x4*)(d + c * 8) = *(const u32x4*)(s + c * 8);
}

extern "C" int vis_gather_rows(const void* table, const void* ids, void* out, int n, int D, int n_table,
                               hipStream_t stream) {
  vis_clear_error();
  if (!table || !ids || !out || n <= 0 || D <= 0 || D % 8 != 0 || n_table <= 0) return VIS_ERR_ARG;
  if (((uintptr_t)table | (uintptr_t)out) & 15) return VIS_ERR_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (const bf16_t*)table,
                     (const int*)ids, (bf16_t*)out, n, D, n_table);
  return vis_check_launch();
}

extern "C" int vis_scatter_rows(const void* src, const void* idx, void* dst, int n, int D, int n_dst,
                                hipStream_t stream) {
  vis_clear_error();
  if (!src || !idx || !dst || n <= 0 || D <= 0 || D % 8 != 0 || n_dst <= 0) return VIS_ERR_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return VIS_ERR_ARG;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (const bf16_t*)src,
                     (const int*)idx, (bf16_t*)dst, n, D, n_dst);
  return vis_check_launch();
}

extern "C" int vis_abi_version(void) { return 3; }   // 2: V^T columns in k-slot order; 3: VIS_ERR_UNSUPPORTED, r04/r05 entry points

// ---------------------------------------------------------------------------
// Row f2 (mllama): vis_patchify_tiles_u8 - the resized RGB frame [H][W][3] u8, zero-padded on the right/bottom to
// the tile canvas (TF:models/mllama/image_processing_pil_mllama.py pad(): the padding is applied to RAW pixels,
// i.e. before rescale/normalise, so padded pixels become (0 - mean) / std), split into tiles_h x tiles_w tiles of
// `tile` x `tile` pixels (split_to_tiles_np) and cut into 14x14 patches in the Conv2d feature order
// f = (c*14 + ph)*14 + pw (TF:models/mllama/modeling_mllama.py:833-840,:912-914).  Output row of tile t, patch
// (py, px):  t * tokens_per_tile + 1 + py * (tile/14) + px   (row 0 of every tile is the CLS slot and is left
// untouched, like the rows of absent tiles - the caller zero-fills the buffer).
struct TilePatchArgs {
  const uint8_t* img;
  bf16_t* out;
  int H, W, tiles_h, tiles_w, tile, ld_out;
  float mean[3], istd[3];
};

__global__ __launch_bounds__(256) void patchify_tiles_u8_kernel(TilePatchArgs p) {
  constexpr int P = 14, F = 3 * P * P;  // 588
  const int G = p.tile / P;             // patches per tile side
  const int chunks = p.ld_out >> 3;
  const int per_tile = G * G;
  const long long total = (long long)p.tiles_h * p.tiles_w * per_tile * chunks;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int ch = (int)(it % chunks);
    const int pidx = (int)(it / chunks);
    const int t = pidx / per_tile, pp = pidx - t * per_tile;
    const int ty = t / p.tiles_w, tx = t - ty * p.tiles_w;
    const int py = pp / G, px = pp - py * G;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = ch * 8 + e;
      float v = 0.f;
      if (f < F) {
        const int c = f / (P * P), rem = f - c * (P * P);
        const int ph = rem / P, pw = rem - ph * P;
        const int y = ty * p.tile + py * P + ph, x = tx * p.tile + px * P + pw;
        const float u = (y < p.H && x < p.W) ? (float)p.img[((size_t)y * p.W + x) * 3 + c] : 0.f;
        v = (u * (1.0f / 255.0f) - p.mean[c]) * p.istd[c];
      }
      o[e] = v;
    }
    const size_t row = (size_t)t * (per_tile + 1) + 1 + pp;
    *(u32x4*)(p.out + row * p.ld_out + ch * 8) = pack8(o);
  }
}

extern "C" int vis_patchify_tiles_u8(const void* img, void* out, int H, int W, int tiles_h, int tiles_w, int tile,
                                     int ld_out, const float* mean, const float* stdv, hipStream_t stream) {
  if (!img || !out || !mean || !stdv || H <= 0 || W <= 0 || tiles_h <= 0 || tiles_w <= 0) return VIS_ERR_ARG;
  if (tile <= 0 || tile % 14 != 0 || H > tiles_h * tile || W > tiles_w * tile) return VIS_ERR_ARG;
  if (ld_out % 8 != 0 || ld_out < 588 || ((uintptr_t)out & 15)) return VIS_ERR_ARG;
  TilePatchArgs p;
  p.img = (const uint8_t*)img; p.out = (bf16_t*)out; p.H = H; p.W = W; p.tiles_h = tiles_h; p.tiles_w = tiles_w;
  p.tile = tile; p.ld_out = ld_out;
  for (int c = 0; c < 3; ++c) {
    if (!(stdv[c] > 0.f)) return VIS_ERR_ARG;
    p.mean[c] = mean[c];
    p.istd[c] = 1.0f / stdv[c];
  }
  const long long total = (long long)tiles_h * tiles_w * (tile / 14) * (tile / 14) * (ld_out / 8);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  vis_clear_error();
  hipLaunchKernelGGL(patchify_tiles_u8_kernel, dim3(blocks), dim3(256), 0, stream, p);
  return vis_check_launch();
}

// vis_add_rows_bf16: x[i][:] += table[idx[i]][:]  (f32 add, one rounding) - the per-tile embedding added to every
// token of a tile (mllama post_tile_positional_embedding, TF:models/mllama/modeling_mllama.py:102-122,:958-961)
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                       const int* __restrict__ idx, int n, int D, int ldx,
                                                       int n_table) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n) return;
  int id = idx[row];
  id = id < 0 ? 0 : (id >= n_table ? n_table - 1 : id);
  bf16_t* xr = x + (size_t)row * ldx;
  const bf16_t* tr = table + (size_t)id * D;
  for (int c = lane; c < (D >> 3); c += 64) {
    float a[8], b[8];
    unpack8(*(const u32x4*)(xr + c * 8), a);
    unpack8(*(const u32x4*)(tr + c * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    *(u32x4*)(xr + c * 8) = pack8(a);
  }
}

extern "C" int vis_add_rows_bf16(void* x, const void* table, const void* idx, int n, int D, int ldx, int n_table,
                                 hipStream_t stream) {
  if (!x || !table || !idx || n <= 0 || D <= 0 || D % 8 != 0 || ldx % 8 != 0 || ldx < D || n_table <= 0)
    return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)table) & 15) return VIS_ERR_ARG;
  vis_clear_error();
  hipLaunchKernelGGL(add_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (bf16_t*)x, (const bf16_t*)table,
                     (const int*)idx, n, D, ldx, n_table);
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// Row f4: the image-quality pre-check's pixel statistics (reference:
// src/safety/image_quality.py:42-56,:118-127 - cv2.cvtColor(BGR2GRAY), cv2.Laplacian(gray, CV_64F).var(),
// np.mean(gray)) as one HBM-bound pass over the decoded RGB frame.
// PARITY UNPINNED: the reference computes these with OpenCV, which is not in this image and has no fixtures in the
// reference; what is restated here is OpenCV's published 8-bit algorithm:
//   gray = (4899 R + 9617 G + 1868 B + 8192) >> 14                  (RGB2GRAY fixed-point coefficients, shift 14)
//   lap  = g[y-1][x] + g[y+1][x] + g[y][x-1] + g[y][x+1] - 4 g[y][x] (ksize = 1 aperture, BORDER_REFLECT_101)
// Outputs are EXACT integer sums (int64 atomics: order-independent): stats[0] = sum gray, stats[1] = sum lap,
// stats[2] = sum lap^2; the host forms mean and variance in float64.
__device__ __forceinline__ int iq_gray(const uint8_t* __restrict__ img, int W, int y, int x) {
  const uint8_t* p = img + ((size_t)y * W + x) * 3;
  return (4899 * p[0] + 9617 * p[1] + 1868 * p[2] + 8192) >> 14;
}

__global__ __launch_bounds__(256) void image_stats_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                          long long* __restrict__ stats) {
  long long sg = 0, sl = 0, sq = 0;
  const long long total = (long long)H * W;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int y = (int)(it / W), x = (int)(it - (long long)y * W);
    const int yu = (y == 0) ? min(1, H - 1) : y - 1, yd = (y == H - 1) ? max(H - 2, 0) : y + 1;
    const int xl = (x == 0) ? min(1, W - 1) : x - 1, xr = (x == W - 1) ? max(W - 2, 0) : x + 1;
    const int c = iq_gray(img, W, y, x);
    const int lap = iq_gray(img, W, yu, x) + iq_gray(img, W, yd, x) + iq_gray(img, W, y, xl) + iq_gray(img, W, y, xr) - 4 * c;
    sg += c;
    sl += lap;
    sq += (long long)lap * lap;
  }
  __shared__ long long red[3][256];
  red[0][threadIdx.x] = sg; red[1][threadIdx.x] = sl; red[2][threadIdx.x] = sq;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicAdd((unsigned long long*)&stats[threadIdx.x], (unsigned long long)red[threadIdx.x][0]);
}

// stats: device int64[3], zeroed by this call before the kernel runs
extern "C" int vis_image_stats_u8(const void* img, int H, int W, void* stats, hipStream_t stream) {
  if (!img || !stats || H <= 0 || W <= 0 || ((uintptr_t)stats & 7)) return VIS_ERR_ARG;
  vis_clear_error();
  if (hipMemsetAsync(stats, 0, 3 * sizeof(long long), stream) != hipSuccess) return VIS_ERR_LAUNCH;
  const long long total = (long long)H * W;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(image_stats_kernel, dim3(blocks), dim3(256), 0, stream, (const uint8_t*)img, H, W, (long long*)stats);
  return vis_check_launch();
}
