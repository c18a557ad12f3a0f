/* Host half of the service-side JPEG decode (vision-inspection-system_amd/csrc/jpeg_host.c -> libvis_jpeg_host.so, plain C,
 * no GPU).  Replaces, together with vis_jpeg_idct / vis_jpeg_to_rgb of include/vis_hip.h, the libjpeg call behind the
 * reference's data-URI images (request side: src/agents/vlm_inspector.py:46-88 writes the JPEG; the service decodes it).
 * Marker parsing + Huffman decoding only: quantised DCT coefficients out, everything after that runs on the GPU.
 *
 * Supported: baseline / 8-bit extended-sequential Huffman JPEG with one interleaved scan; 1 component (grey) or 3
 * (YCbCr: luma 1x1, 2x1 or 2x2, chroma 1x1); restart markers.  Everything else returns VIS_JPEG_UNSUPPORTED (-1) and the
 * caller decodes with PIL; damaged data returns VIS_JPEG_CORRUPT (-2). */
#ifndef VIS_JPEG_HOST_H
#define VIS_JPEG_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int width, height, ncomp;
  int hs[3], vs[3];      /* sampling factors */
  int bw[3], bh[3];      /* 8x8 blocks per row / column of each component plane (whole MCUs) */
  int dw[3], dh[3];      /* real size of each component in samples (libjpeg's downsampled_width / _height) */
  int mcus_x, mcus_y;
  int restart_interval;
  int total_blocks;      /* sum of bw * bh: vis_jpeg_decode_coeffs writes total_blocks * 64 int16 */
  int sos_offset;
  uint16_t qt[3][64];    /* per component, natural (row-major) order */
  uint8_t dc_tab[3], ac_tab[3];
  uint8_t huff_counts[2][4][16];
  uint8_t huff_syms[2][4][256];
  uint8_t huff_present[2][4];
} VisJpegInfo;

int vis_jpeg_info_size(void);                                               /* sizeof(VisJpegInfo), for bindings */
int vis_jpeg_probe(const uint8_t* data, size_t n, VisJpegInfo* info);      /* 0, or -1 unsupported, -2 corrupt */
/* coeffs: component planes back to back, blocks row-major, 64 int16 per block in natural order, NOT dequantised */
int vis_jpeg_decode_coeffs(const uint8_t* data, size_t n, const VisJpegInfo* info, int16_t* coeffs);

#ifdef __cplusplus
}
#endif
#endif
