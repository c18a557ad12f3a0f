// GPU half of the service-side JPEG decode.  The host (csrc/jpeg_host.c) has parsed the
// markers and Huffman-decoded the scan into quantised DCT coefficients; here:
//   jpeg_idct_kernel   : one thread per 8x8 block - dequantise, libjpeg's integer "islow" inverse DCT
//                        (jidctint.c jpeg_idct_islow: 13-bit constants, 2 extra bits between the passes), +128, clamp;
//                        writes the component planes (uint8, whole MCUs).
//   jpeg_rgb_kernel    : one thread per output pixel quad - "fancy" triangle chroma upsampling (jdsample.c
//                        h2v1_fancy_upsample / h2v2_fancy_upsample; replication when the chroma plane is <= 2 samples
//                        wide, as jinit_upsampler selects) and the 16-bit fixed-point YCbCr -> RGB of jdcolor.c.
// Every intermediate is an integer: the output equals libjpeg-turbo's (= PIL's) bit for bit (tests/test_jpeg_gpu.py against
// PIL and against oracle/jpeg_ref.py).  HBM-bound byte work: 128 B of coefficients in, 64 B of samples out per block;
// 1.5 B in, 3 B out per pixel (4:2:0).
#include "common.hip.h"

struct JpegArgs {
  const int16_t* coeffs;   // [total_blocks][64], natural order, quantised
  const int* qt;           // [3][64]
  uint8_t* planes;         // Y plane, then Cb, Cr (each bh*8 rows of bw*8 bytes)
  uint8_t* rgb;            // [H][W][3]
  int W, H, ncomp, hs, vs;
  int bw_y, bh_y, bw_c, bh_c, dw_c, dh_c;
};

#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

// one 8-point pass of jpeg_idct_islow; SHIFT = 11 (columns) or 18 (rows)
template <int SHIFT>
__device__ __forceinline__ void jpeg_idct8(const int (&in)[8], int (&out)[8]) {
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * JF_0_541196100;
  int tmp2 = z1 - z3 * JF_1_847759065;
  int tmp3 = z1 + z2 * JF_0_765366865;
  z2 = in[0]; z3 = in[4];
  int tmp0 = (z2 + z3) << 13;
  int tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * JF_1_175875602;
  tmp0 *= JF_0_298631336; tmp1 *= JF_2_053119869; tmp2 *= JF_3_072711026; tmp3 *= JF_1_501321110;
  z1 *= -JF_0_899976223; z2 *= -JF_2_562915447;
  z3 = z3 * -JF_1_961570560 + z5;
  z4 = z4 * -JF_0_390180644 + z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  constexpr int R = 1 << (SHIFT - 1);
  out[0] = (tmp10 + tmp3 + R) >> SHIFT; out[7] = (tmp10 - tmp3 + R) >> SHIFT;
  out[1] = (tmp11 + tmp2 + R) >> SHIFT; out[6] = (tmp11 - tmp2 + R) >> SHIFT;
  out[2] = (tmp12 + tmp1 + R) >> SHIFT; out[5] = (tmp12 - tmp1 + R) >> SHIFT;
  out[3] = (tmp13 + tmp0 + R) >> SHIFT; out[4] = (tmp13 - tmp0 + R) >> SHIFT;
}

__global__ __launch_bounds__(128) void jpeg_idct_kernel(JpegArgs p) {
  __shared__ int qts[3 * 64];
  for (int i = threadIdx.x; i < 3 * 64; i += 128) qts[i] = p.qt[i];
  __syncthreads();
  const int ny = p.bw_y * p.bh_y, nc = p.bw_c * p.bh_c;
  const int total = ny + (p.ncomp == 3 ? 2 * nc : 0);
  const int b = blockIdx.x * 128 + threadIdx.x;
  if (b >= total) return;
  int comp = 0, local = b, bw = p.bw_y;
  uint8_t* plane = p.planes;
  if (b >= ny) {
    comp = 1 + (b - ny) / nc;
    local = (b - ny) % nc;
    bw = p.bw_c;
    plane = p.planes + (size_t)ny * 64 + (size_t)(comp - 1) * nc * 64;
  }
  const int by = local / bw, bx = local - by * bw;
  const int* q = qts + comp * 64;
  // the block: 8 x 16-byte loads, row r of the block in regs v[r][0..7]
  int ws[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const u32x4 raw = *(const u32x4*)(p.coeffs + (size_t)b * 64 + r * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ws[r][2 * e] = (int)(int16_t)(raw[e] & 0xffffu) * q[r * 8 + 2 * e];
      ws[r][2 * e + 1] = (int)(int16_t)(raw[e] >> 16) * q[r * 8 + 2 * e + 1];
    }
  }
  // pass 1: columns
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = ws[r][c];
    jpeg_idct8<11>(in, out);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r][c] = out[r];
  }
  // pass 2: rows, range limit (libjpeg's table index is masked to 10 bits), +128
  uint8_t* dst = plane + ((size_t)by * 8) * (bw * 8) + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int out[8];
    jpeg_idct8<18>(ws[r], out);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int v = ((out[c] + 512) & 1023) - 512 + 128;
      v = min(max(v, 0), 255);
      if (c < 4) lo |= (uint32_t)v << (8 * c);
      else hi |= (uint32_t)v << (8 * (c - 4));
    }
    *(u32x2*)(dst + (size_t)r * (bw * 8)) = (u32x2){lo, hi};
  }
}

// chroma sample for output pixel (y, x) of a plane with real size dh x dw and row stride ld
template <int HS, int VS>
__device__ __forceinline__ int jpeg_chroma(const uint8_t* pl, int ld, int dw, int dh, int y, int x) {
  if constexpr (HS == 1) {
    return pl[(size_t)y * ld + x];
  } else if constexpr (VS == 1) {   // h2v1
    const int cx = x >> 1;
    const uint8_t* row = pl + (size_t)y * ld;
    if (dw <= 2) return row[cx];
    const int cur = row[cx];
    if (x & 1) return (cx == dw - 1) ? cur : (3 * cur + row[cx + 1] + 2) >> 2;
    return (cx == 0) ? cur : (3 * cur + row[cx - 1] + 1) >> 2;
  } else {                          // h2v2
    const int cx = x >> 1, cy = y >> 1;
    if (dw <= 2) return pl[(size_t)cy * ld + cx];
    const int fy = (y & 1) ? min(cy + 1, dh - 1) : max(cy - 1, 0);
    const uint8_t* r0 = pl + (size_t)cy * ld;
    const uint8_t* r1 = pl + (size_t)fy * ld;
    const int nx = (x & 1) ? min(cx + 1, dw - 1) : max(cx - 1, 0);
    const int cs = 3 * r0[cx] + r1[cx], ns = 3 * r0[nx] + r1[nx];
    return (3 * cs + ns + ((x & 1) ? 7 : 8)) >> 4;
  }
}

template <int HS, int VS>
__global__ __launch_bounds__(256) void jpeg_rgb_kernel(JpegArgs p) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= p.W) return;
  const int ldy = p.bw_y * 8, ldc = p.bw_c * 8;
  const uint8_t* py = p.planes;
  const int yy = py[(size_t)y * ldy + x];
  uint8_t* o = p.rgb + ((size_t)y * p.W + x) * 3;
  if (p.ncomp == 1) {
    o[0] = o[1] = o[2] = (uint8_t)yy;
    return;
  }
  const uint8_t* pcb = py + (size_t)p.bw_y * p.bh_y * 64;
  const uint8_t* pcr = pcb + (size_t)p.bw_c * p.bh_c * 64;
  const int cb = jpeg_chroma<HS, VS>(pcb, ldc, p.dw_c, p.dh_c, y, x) - 128;
  const int cr = jpeg_chroma<HS, VS>(pcr, ldc, p.dw_c, p.dh_c, y, x) - 128;
  // jdcolor.c build_ycc_rgb_table: FIX(x) = (int)(x * 65536 + 0.5), ONE_HALF = 32768, arithmetic right shifts
  const int r = yy + ((91881 * cr + 32768) >> 16);
  const int b = yy + ((116130 * cb + 32768) >> 16);
  const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
  o[0] = (uint8_t)min(max(r, 0), 255);
  o[1] = (uint8_t)min(max(g, 0), 255);
  o[2] = (uint8_t)min(max(b, 0), 255);
}

extern "C" int vis_jpeg_to_rgb(const void* coeffs, const void* qt, void* planes, void* rgb, int width, int height,
                               int ncomp, int hs, int vs, int bw_y, int bh_y, int bw_c, int bh_c, int dw_c, int dh_c,
                               hipStream_t stream) {
  if (!coeffs || !qt || !planes || !rgb || width <= 0 || height <= 0) return VIS_ERR_ARG;
  if (ncomp != 1 && ncomp != 3) return VIS_ERR_ARG;
  if (bw_y <= 0 || bh_y <= 0 || bw_y * 8 < width || bh_y * 8 < height) return VIS_ERR_ARG;
  if (ncomp == 3) {
    if (!((hs == 1 && vs == 1) || (hs == 2 && vs == 1) || (hs == 2 && vs == 2))) return VIS_ERR_ARG;
    if (bw_c <= 0 || bh_c <= 0 || dw_c <= 0 || dh_c <= 0 || dw_c > bw_c * 8 || dh_c > bh_c * 8) return VIS_ERR_ARG;
    if (dw_c * hs < width || dh_c * vs < height) return VIS_ERR_ARG;   // every output pixel has a chroma sample
  }
  if (((uintptr_t)coeffs & 15) || ((uintptr_t)planes & 7)) return VIS_ERR_ARG;
  JpegArgs p;
  p.coeffs = (const int16_t*)coeffs; p.qt = (const int*)qt; p.planes = (uint8_t*)planes; p.rgb = (uint8_t*)rgb;
  p.W = width; p.H = height; p.ncomp = ncomp; p.hs = hs; p.vs = vs;
  p.bw_y = bw_y; p.bh_y = bh_y; p.bw_c = bw_c; p.bh_c = bh_c; p.dw_c = dw_c; p.dh_c = dh_c;
  const int total = bw_y * bh_y + (ncomp == 3 ? 2 * bw_c * bh_c : 0);
  vis_clear_error();
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((total + 127) / 128), dim3(128), 0, stream, p);
  const dim3 grid((width + 255) / 256, height), block(256);
  if (ncomp == 1 || (hs == 1 && vs == 1))
    hipLaunchKernelGGL((jpeg_rgb_kernel<1, 1>), grid, block, 0, stream, p);
  else if (vs == 1)
    hipLaunchKernelGGL((jpeg_rgb_kernel<2, 1>), grid, block, 0, stream, p);
  else
    hipLaunchKernelGGL((jpeg_rgb_kernel<2, 2>), grid, block, 0, stream, p);
  return vis_check_launch();
}
