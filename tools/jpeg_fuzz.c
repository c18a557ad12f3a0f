// ASan / UBSan fuzz driver for the host half of the JPEG decode (csrc/jpeg_host.c), CPU only.
//   tools/jpeg_fuzz.sh [iterations]     builds this with -fsanitize=address,undefined and runs it over PIL-written seeds
// Every mutated file is copied into an exact-size heap block (so an over-read is an ASan finding), probed and decoded.
// Mutations: bit flips, byte overwrites, truncation, byte deletion, marker insertion (RSTn / EOI / SOS / FFxx), splices.
#include <stdio.h>
#include <stdlib.h>
#include "../vision-inspection-system_amd/csrc/jpeg_host.c"

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); }

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: jpeg_fuzz ITERATIONS seed.jpg...\n"); return 2; }
  const long iters = atol(argv[1]);
  const int nseed = argc - 2;
  uint8_t** seed = malloc(sizeof(*seed) * nseed);
  size_t* slen = malloc(sizeof(*slen) * nseed);
  for (int i = 0; i < nseed; ++i) {
    FILE* f = fopen(argv[i + 2], "rb");
    if (!f) { perror(argv[i + 2]); return 2; }
    fseek(f, 0, SEEK_END); slen[i] = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    seed[i] = malloc(slen[i]);
    if (fread(seed[i], 1, slen[i], f) != slen[i]) return 2;
    fclose(f);
  }
  long ok = 0, unsupported = 0, corrupt_probe = 0, corrupt_scan = 0;
  for (long it = 0; it < iters; ++it) {
    const int s = rnd() % nseed;
    size_t n = slen[s];
    uint8_t* w = malloc(n + 64);
    memcpy(w, seed[s], n);
    const int nmut = 1 + rnd() % 4;
    for (int m = 0; m < nmut && n > 4; ++m) {
      const size_t at = rnd() % n;
      switch (rnd() % 7) {
        case 0: w[at] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 1: w[at] = (uint8_t)rnd(); break;
        case 2: n = at < 4 ? 4 : at; break;                                        // truncate
        case 3: memmove(w + at, w + at + 1, n - at - 1); --n; break;                // delete a byte
        case 4: if (n + 2 <= slen[s] + 60) { memmove(w + at + 2, w + at, n - at); w[at] = 0xFF;
                  static const uint8_t mk[] = {0xD0, 0xD3, 0xD7, 0xD9, 0xDA, 0xC4, 0xDB, 0x00, 0xFF};
                  w[at + 1] = mk[rnd() % sizeof(mk)]; n += 2; } break;              // insert a marker
        case 5: { const size_t from = rnd() % n, len = 1 + rnd() % 32;              // splice
                  for (size_t k = 0; k < len && at + k < n && from + k < n; ++k) w[at + k] = w[from + k]; } break;
        default: w[at] = 0xFF; break;
      }
    }
    uint8_t* exact = malloc(n);
    memcpy(exact, w, n);
    free(w);
    VisJpegInfo info;
    const int p = vis_jpeg_probe(exact, n, &info);
    if (p == VIS_JPEG_OK) {
      int16_t* c = malloc((size_t)info.total_blocks * 64 * sizeof(int16_t));
      const int d = vis_jpeg_decode_coeffs(exact, n, &info, c);
      if (d == VIS_JPEG_OK) ++ok; else ++corrupt_scan;
      free(c);
    } else if (p == VIS_JPEG_UNSUPPORTED) ++unsupported; else ++corrupt_probe;
    free(exact);
  }
  for (int i = 0; i < nseed; ++i) free(seed[i]);
  free(seed); free(slen);
  printf("%ld mutated files: %ld decoded, %ld refused in the scan, %ld refused in the headers, %ld unsupported; no sanitizer finding\n",
         iters, ok, corrupt_scan, corrupt_probe, unsupported);
  return 0;
}
