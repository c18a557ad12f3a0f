// Host half of the service-side JPEG decode ("GPU-side preprocessing: JPEG decode -> ...").
//
// The reference's agents send every image as a baseline JPEG (q85, 4:2:0, optimised Huffman tables:
// src/agents/vlm_inspector.py:46-88) and its service decodes it with libjpeg.  Entropy decoding is a serial bit
// stream and stays on a host core (one image per ingest-pool thread); everything after it - dequantisation, the 8x8
// inverse DCT, chroma upsampling, YCbCr -> RGB - is data parallel and runs on the GPU (csrc/jpeg.hip), bit-exact with
// libjpeg-turbo's default decoder (integer "islow" IDCT, "fancy" triangle upsampling, 16-bit fixed-point colour
// tables), i.e. with what PIL returns for the same bytes.
//
// This file: marker parsing + Huffman decoding of baseline (SOF0 / 8-bit SOF1) JPEGs with one interleaved scan,
// 1 component (grey) or 3 components (YCbCr; luma 1x1, 2x1 or 2x2, chroma 1x1), restart markers included.
// Anything else (progressive, arithmetic coding, CMYK, Adobe RGB, multi-scan, 12-bit, exotic sampling) returns
// VIS_JPEG_UNSUPPORTED and the caller decodes that image with PIL instead.
//
// C ABI (include/vis_jpeg_host.h); built by `make` / __graft_entry__.build() with gcc into libvis_jpeg_host.so.
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define VIS_JPEG_OK 0
#define VIS_JPEG_UNSUPPORTED (-1)
#define VIS_JPEG_CORRUPT (-2)

typedef struct {
  int width, height, ncomp;
  int hs[3], vs[3];        // sampling factors per component
  int bw[3], bh[3];        // blocks per row / column of each component plane (padded to whole MCUs)
  int dw[3], dh[3];        // real ("downsampled") size of each component in samples
  int mcus_x, mcus_y;
  int restart_interval;
  int total_blocks;        // sum over components of bw * bh
  int sos_offset;          // byte offset of the entropy-coded data
  uint16_t qt[3][64];      // quantisation table of each component, natural (row-major) order
  uint8_t dc_tab[3], ac_tab[3];
  // Huffman tables as transmitted (counts + symbols), indexed [class 0 = DC / 1 = AC][id 0..3]
  uint8_t huff_counts[2][4][16];
  uint8_t huff_syms[2][4][256];
  uint8_t huff_present[2][4];
} VisJpegInfo;

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int vis_jpeg_probe(const uint8_t* d, size_t n, VisJpegInfo* info) {
  if (!d || !info || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return VIS_JPEG_CORRUPT;
  memset(info, 0, sizeof(*info));
  uint16_t qtabs[4][64];
  int qt_present[4] = {0, 0, 0, 0};
  int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0};
  int have_sof = 0, adobe_transform = -1, have_jfif = 0;
  size_t i = 2;
  for (;;) {
    if (i + 4 > n) return VIS_JPEG_CORRUPT;
    if (d[i] != 0xFF) return VIS_JPEG_CORRUPT;
    while (i < n && d[i] == 0xFF) ++i;  // fill bytes
    if (i >= n) return VIS_JPEG_CORRUPT;
    const int m = d[i++];
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;  // standalone markers
    if (m == 0xD9) return VIS_JPEG_CORRUPT;                              // EOI before any scan
    if (i + 2 > n) return VIS_JPEG_CORRUPT;
    const int L = be16(d + i);
    if (L < 2 || i + (size_t)L > n) return VIS_JPEG_CORRUPT;
    const uint8_t* p = d + i + 2;
    const int len = L - 2;
    if (m == 0xDB) {  // DQT
      int o = 0;
      while (o < len) {
        const int pq = p[o] >> 4, tq = p[o] & 15;
        ++o;
        if (tq > 3 || pq > 1) return VIS_JPEG_CORRUPT;
        if (o + 64 * (pq + 1) > len) return VIS_JPEG_CORRUPT;
        for (int k = 0; k < 64; ++k) {
          const int v = pq ? be16(p + o + 2 * k) : p[o + k];
          qtabs[tq][ZIGZAG[k]] = (uint16_t)v;
        }
        o += 64 * (pq + 1);
        qt_present[tq] = 1;
      }
    } else if (m == 0xC4) {  // DHT
      int o = 0;
      while (o < len) {
        if (o + 17 > len) return VIS_JPEG_CORRUPT;
        const int tc = p[o] >> 4, th = p[o] & 15;
        if (tc > 1 || th > 3) return VIS_JPEG_CORRUPT;
        int total = 0;
        for (int k = 0; k < 16; ++k) {
          info->huff_counts[tc][th][k] = p[o + 1 + k];
          total += p[o + 1 + k];
        }
        if (total > 256 || o + 17 + total > len) return VIS_JPEG_CORRUPT;
        memcpy(info->huff_syms[tc][th], p + o + 17, (size_t)total);
        info->huff_present[tc][th] = 1;
        o += 17 + total;
      }
    } else if (m == 0xC0 || m == 0xC1) {  // baseline / extended sequential, Huffman
      if (have_sof || len < 6) return VIS_JPEG_CORRUPT;
      if (p[0] != 8) return VIS_JPEG_UNSUPPORTED;  // 12-bit samples
      info->height = be16(p + 1);
      info->width = be16(p + 3);
      info->ncomp = p[5];
      if (info->width <= 0 || info->height <= 0) return VIS_JPEG_UNSUPPORTED;  // height 0 = DNL marker form
      if (info->ncomp != 1 && info->ncomp != 3) return VIS_JPEG_UNSUPPORTED;
      if (len < 6 + 3 * info->ncomp) return VIS_JPEG_CORRUPT;
      for (int c = 0; c < info->ncomp; ++c) {
        comp_id[c] = p[6 + 3 * c];
        info->hs[c] = p[7 + 3 * c] >> 4;
        info->vs[c] = p[7 + 3 * c] & 15;
        comp_tq[c] = p[8 + 3 * c];
        if (comp_tq[c] > 3) return VIS_JPEG_CORRUPT;
      }
      have_sof = 1;
    } else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return VIS_JPEG_UNSUPPORTED;  // progressive, lossless, arithmetic, hierarchical
    } else if (m == 0xCC) {
      return VIS_JPEG_UNSUPPORTED;  // arithmetic conditioning
    } else if (m == 0xDD) {  // DRI
      if (len < 2) return VIS_JPEG_CORRUPT;
      info->restart_interval = be16(p);
    } else if (m == 0xE0) {
      if (len >= 5 && !memcmp(p, "JFIF", 5)) have_jfif = 1;
    } else if (m == 0xEE) {
      if (len >= 12 && !memcmp(p, "Adobe", 5)) adobe_transform = p[11];
    } else if (m == 0xDA) {  // SOS
      if (!have_sof || len < 1) return VIS_JPEG_CORRUPT;
      const int ns = p[0];
      if (ns != info->ncomp) return VIS_JPEG_UNSUPPORTED;  // one interleaved scan only
      if (len < 1 + 2 * ns + 3) return VIS_JPEG_CORRUPT;
      for (int s = 0; s < ns; ++s) {
        if (p[1 + 2 * s] != comp_id[s]) return VIS_JPEG_UNSUPPORTED;  // scan order = frame order
        info->dc_tab[s] = p[2 + 2 * s] >> 4;
        info->ac_tab[s] = p[2 + 2 * s] & 15;
        if (info->dc_tab[s] > 3 || info->ac_tab[s] > 3) return VIS_JPEG_CORRUPT;
        if (!info->huff_present[0][info->dc_tab[s]] || !info->huff_present[1][info->ac_tab[s]]) return VIS_JPEG_CORRUPT;
      }
      if (p[1 + 2 * ns] != 0 || p[2 + 2 * ns] != 63 || p[3 + 2 * ns] != 0) return VIS_JPEG_UNSUPPORTED;
      info->sos_offset = (int)(i + (size_t)L);
      break;
    }
    i += (size_t)L;
  }
  // colour space: libjpeg's rules (jdapimin.c default_decompress_parms) reduced to the two cases handled here
  if (info->ncomp == 3) {
    if (adobe_transform == 0) return VIS_JPEG_UNSUPPORTED;  // Adobe RGB
    if (!have_jfif && adobe_transform < 0 && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B')
      return VIS_JPEG_UNSUPPORTED;
    if (info->hs[1] != 1 || info->vs[1] != 1 || info->hs[2] != 1 || info->vs[2] != 1) return VIS_JPEG_UNSUPPORTED;
    const int h = info->hs[0], v = info->vs[0];
    if (!((h == 1 && v == 1) || (h == 2 && v == 1) || (h == 2 && v == 2))) return VIS_JPEG_UNSUPPORTED;
  } else {
    info->hs[0] = info->vs[0] = 1;  // a single-component scan is never interleaved: one block per MCU
  }
  const int hmax = info->hs[0], vmax = info->vs[0];
  info->mcus_x = (info->width + 8 * hmax - 1) / (8 * hmax);
  info->mcus_y = (info->height + 8 * vmax - 1) / (8 * vmax);
  info->total_blocks = 0;
  for (int c = 0; c < info->ncomp; ++c) {
    if (!qt_present[comp_tq[c]]) return VIS_JPEG_CORRUPT;
    memcpy(info->qt[c], qtabs[comp_tq[c]], sizeof(info->qt[c]));
    info->bw[c] = info->mcus_x * info->hs[c];
    info->bh[c] = info->mcus_y * info->vs[c];
    info->dw[c] = (info->width * info->hs[c] + hmax - 1) / hmax;
    info->dh[c] = (info->height * info->vs[c] + vmax - 1) / vmax;
    info->total_blocks += info->bw[c] * info->bh[c];
  }
  // beyond ~89 M luma samples (PIL's own decompression-bomb limit) let PIL decide what to do with the file
  if ((long long)info->total_blocks * 64 > (1LL << 27)) return VIS_JPEG_UNSUPPORTED;
  return VIS_JPEG_OK;
}

// ---- canonical Huffman decoding (ITU T.81 Annex F.2.2.3) with an 9-bit first-level table
#define FAST_BITS 9
typedef struct {
  uint16_t fast[1 << FAST_BITS];  // (length << 8) | symbol, 0 = longer than FAST_BITS
  int32_t maxcode[18];            // largest code of each length (left-justified to 16 bits), -1 = none
  int32_t valoff[17];             // symbol index = code + valoff[length]
  const uint8_t* syms;
} HuffDec;

static int build_huff(HuffDec* h, const uint8_t* counts, const uint8_t* syms) {
  uint16_t codes[256];
  uint8_t sizes[256];
  int n = 0;
  for (int len = 1; len <= 16; ++len)
    for (int k = 0; k < counts[len - 1]; ++k) sizes[n++] = (uint8_t)len;
  int code = 0, si = n ? sizes[0] : 0, k = 0;
  while (k < n) {
    while (k < n && sizes[k] == si) codes[k++] = (uint16_t)code++;
    if (code - 1 >= (1 << si) && k) return -1;  // over-subscribed
    code <<= 1;
    ++si;
  }
  memset(h->fast, 0, sizeof(h->fast));
  h->syms = syms;
  int p = 0;
  for (int len = 1; len <= 16; ++len) {
    if (counts[len - 1]) {
      h->valoff[len] = p - codes[p];
      p += counts[len - 1];
      h->maxcode[len] = ((int32_t)codes[p - 1] + 1) << (16 - len);  // exclusive upper bound, 16-bit aligned
    } else {
      h->valoff[len] = 0;
      h->maxcode[len] = -1;
    }
  }
  h->maxcode[17] = 0x7fffffff;
  for (int s = 0; s < n; ++s) {
    const int len = sizes[s];
    if (len <= FAST_BITS) {
      const int first = codes[s] << (FAST_BITS - len), cnt = 1 << (FAST_BITS - len);
      for (int j = 0; j < cnt; ++j) h->fast[first + j] = (uint16_t)((len << 8) | syms[s]);
    }
  }
  return 0;
}

typedef struct {
  const uint8_t* d;
  size_t pos, n;
  uint64_t acc;   // bits left-justified? no: right-justified, `cnt` valid low bits
  int cnt;
  int marker;     // a marker (not FF00) or the end of the data was reached: feed zero bits from here on
  int fake;       // how many of the `cnt` bits (always the lowest ones) are such made-up zero bits
} Bits;

static void fill(Bits* b) {
  while (b->cnt <= 56) {
    int byte = 0;
    if (!b->marker && b->pos < b->n) {
      byte = b->d[b->pos];
      if (byte == 0xFF) {
        if (b->pos + 1 < b->n && b->d[b->pos + 1] == 0x00) {
          b->pos += 2;
        } else {
          b->marker = 1;  // leave pos at the FF
          byte = 0;
        }
      } else {
        b->pos += 1;
      }
    } else {
      b->marker = 1;
    }
    b->acc = (b->acc << 8) | (uint64_t)byte;
    b->cnt += 8;
    if (b->marker) b->fake += 8;
  }
}

// True when the decoder has consumed bits that were never in the file: the scan ended (marker or end of data) in the
// middle of an MCU.  libjpeg pads such a scan with zero bits and warns; here the image goes back to the caller as
// VIS_JPEG_CORRUPT and PIL decides what a damaged file means (it raises "image file is truncated").
static inline int overran(const Bits* b) { return b->cnt < b->fake; }

static inline int peek(Bits* b, int nbits) { return (int)((b->acc >> (b->cnt - nbits)) & ((1u << nbits) - 1)); }
static inline void drop(Bits* b, int nbits) { b->cnt -= nbits; }

static inline int decode_sym(Bits* b, const HuffDec* h) {
  if (b->cnt < 16) fill(b);
  const int f = h->fast[peek(b, FAST_BITS)];
  if (f) {
    drop(b, f >> 8);
    return f & 0xff;
  }
  const int32_t c16 = peek(b, 16);
  for (int len = FAST_BITS + 1; len <= 16; ++len) {
    if (h->maxcode[len] >= 0 && c16 < h->maxcode[len]) {
      const int code = c16 >> (16 - len);
      const int idx = code + h->valoff[len];
      if (idx < 0 || idx > 255) return -1;
      drop(b, len);
      return h->syms[idx];
    }
  }
  return -1;
}

static inline int receive_extend(Bits* b, int s) {
  if (s == 0) return 0;
  if (b->cnt < s) fill(b);
  const int v = peek(b, s);
  drop(b, s);
  return (v < (1 << (s - 1))) ? v - (1 << s) + 1 : v;
}

// coeffs: [component 0 blocks (row-major over bh x bw)] [component 1 ...] [component 2 ...], 64 int16 each, natural
// order, NOT dequantised (the GPU multiplies by info->qt).  Blocks are zero-filled first.
int vis_jpeg_decode_coeffs(const uint8_t* d, size_t n, const VisJpegInfo* info, int16_t* coeffs) {
  if (!d || !info || !coeffs || info->sos_offset <= 0 || (size_t)info->sos_offset > n) return VIS_JPEG_CORRUPT;
  HuffDec dc[3], ac[3];
  size_t base[3];
  size_t off = 0;
  for (int c = 0; c < info->ncomp; ++c) {
    if (build_huff(&dc[c], info->huff_counts[0][info->dc_tab[c]], info->huff_syms[0][info->dc_tab[c]])) return VIS_JPEG_CORRUPT;
    if (build_huff(&ac[c], info->huff_counts[1][info->ac_tab[c]], info->huff_syms[1][info->ac_tab[c]])) return VIS_JPEG_CORRUPT;
    base[c] = off;
    off += (size_t)info->bw[c] * info->bh[c];
  }
  memset(coeffs, 0, off * 64 * sizeof(int16_t));
  Bits b;
  b.d = d; b.pos = (size_t)info->sos_offset; b.n = n; b.acc = 0; b.cnt = 0; b.marker = 0; b.fake = 0;
  int pred[3] = {0, 0, 0};
  int until_restart = info->restart_interval, next_rst = 0;
  for (int my = 0; my < info->mcus_y; ++my) {
    for (int mx = 0; mx < info->mcus_x; ++mx) {
      if (info->restart_interval && until_restart == 0) {
        // byte-align, expect RSTn, reset predictors
        if (b.cnt - b.fake >= 8) return VIS_JPEG_CORRUPT;  // whole unread bytes between the interval's last MCU and RSTn
        b.acc = 0; b.cnt = 0; b.fake = 0;
        // (the reader may not have seen the marker yet if the segment ended on a byte boundary: it is then the next
        // unread byte - anything else in front of it is damage)
        if (b.pos + 1 >= b.n || b.d[b.pos] != 0xFF || b.d[b.pos + 1] != (0xD0 + next_rst)) return VIS_JPEG_CORRUPT;
        b.pos += 2;
        b.marker = 0;
        next_rst = (next_rst + 1) & 7;
        pred[0] = pred[1] = pred[2] = 0;
        until_restart = info->restart_interval;
      }
      for (int c = 0; c < info->ncomp; ++c) {
        for (int v = 0; v < info->vs[c]; ++v) {
          for (int h = 0; h < info->hs[c]; ++h) {
            const int by = my * info->vs[c] + v, bx = mx * info->hs[c] + h;
            int16_t* blk = coeffs + (base[c] + (size_t)by * info->bw[c] + bx) * 64;
            const int t = decode_sym(&b, &dc[c]);
            if (t < 0 || t > 11) return VIS_JPEG_CORRUPT;
            pred[c] += receive_extend(&b, t);
            if (pred[c] < -32768 || pred[c] > 32767) return VIS_JPEG_CORRUPT;  // no valid stream leaves the int16 range
            blk[0] = (int16_t)pred[c];
            int k = 1;
            while (k < 64) {
              const int rs = decode_sym(&b, &ac[c]);
              if (rs < 0) return VIS_JPEG_CORRUPT;
              const int r = rs >> 4, s = rs & 15;
              if (s == 0) {
                if (r != 15) break;  // EOB
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) return VIS_JPEG_CORRUPT;
              blk[ZIGZAG[k]] = (int16_t)receive_extend(&b, s);
              ++k;
            }
          }
        }
      }
      if (overran(&b)) return VIS_JPEG_CORRUPT;  // the data ended (or a marker came) inside this MCU
      if (info->restart_interval) --until_restart;
    }
  }
  // After the last MCU: < 8 padding bits, then EOI (fill FF bytes allowed in front of it).  Anything else - extra
  // bytes, another marker, no EOI at all - is a damaged or unusual file: PIL decides.
  if (b.cnt - b.fake >= 8) return VIS_JPEG_CORRUPT;
  size_t q = b.pos;
  if (q >= n || d[q] != 0xFF) return VIS_JPEG_CORRUPT;
  while (q + 1 < n && d[q + 1] == 0xFF) ++q;
  if (q + 1 >= n || d[q + 1] != 0xD9) return VIS_JPEG_CORRUPT;
  return VIS_JPEG_OK;
}

int vis_jpeg_info_size(void) { return (int)sizeof(VisJpegInfo); }
