// K10 / K11 / K12: the per-token decode path on gfx950 - HBM-bound by design.
//
//  * vis_gemv_bf16: y = act(W x + bias) + R with an optional fused RMSNorm of x
//    (K3 folded into the prologue).  Weights are streamed exactly once,
//    16 B per lane straight to VGPRs (no LDS round trip for a read-once
//    operand), x lives in LDS, products via v_dot2c_f32_bf16, f32 accumulate.
//    Algorithmic bytes per launch: N*K*2 (weights) - the roofline denominator
//    of the whole decode phase (14.14 GB per generated token at 7B).
//  * vis_decode_attn (+ combine): GQA attention of one query token over the KV
//    cache, split over the context so the 4 KV heads still fill the chip.
//  * vis_argmax_f32: greedy next-token pick, appends to the device-side token
//    list and advances the device-side step counter, so a whole decode step
//    needs no host round trip and sits in one replayable hipGraph.
// Reference semantics: TF:models/qwen2_vl/modeling_qwen2_vl.py:96-110 (norm),
// :459-466 (MLP), :501-556 (attention), greedy = argmax over lm_head logits.
#include "common.hip.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

enum { GV_ACT_NONE = 0, GV_ACT_SWIGLU = 3 };

struct GemvArgs {
  const bf16_t* x;
  const bf16_t* W;
  const bf16_t* bias;
  const bf16_t* R;
  const bf16_t* norm_w;  // non-null: x <- rmsnorm(x) * norm_w before the product
  void* y;
  int N, K, ldw;
  int act, out_f32;
  int outs_per_block;
  float eps;
};

// NB: __builtin_bit_cast(bf16x2, v[i]) on a vector ELEMENT is miscompiled by hipcc 7.2 (always
// element 0); extract the bf16 pairs with shufflevector from a whole-vector bit_cast instead.
__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
  const bf16x8 wv = __builtin_bit_cast(bf16x8, w), xv = __builtin_bit_cast(bf16x8, x);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 0, 1), __builtin_shufflevector(xv, xv, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 2, 3), __builtin_shufflevector(xv, xv, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 4, 5), __builtin_shufflevector(xv, xv, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(wv, wv, 6, 7), __builtin_shufflevector(xv, xv, 6, 7), acc, false);
  return acc;
}

// every wave produces outputs in pairs (two weight rows in flight per lane)
__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemvArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch = p.K >> 3;

  // ---- stage x (optionally RMS-normalised) into LDS
  if (p.norm_w) {
    float ss = 0.f;
    for (int c = tid; c < nch; c += 256) {
      float f[8];
      unpack8(*(const u32x4*)(p.x + c * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += f[e] * f[e];
    }
    ss = wave_sum(ss);
    __shared__ float red[4];
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)p.K + p.eps);
    for (int c = tid; c < nch; c += 256) {
      float f[8], w[8], o[8];
      unpack8(*(const u32x4*)(p.x + c * 8), f);
      unpack8(*(const u32x4*)(p.norm_w + c * 8), w);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bf2f(f2bf(f[e] * rstd)) * w[e];
      *(u32x4*)(xs + c * 8) = pack8(o);
    }
  } else {
    for (int c = tid; c < nch; c += 256) *(u32x4*)(xs + c * 8) = *(const u32x4*)(p.x + c * 8);
  }
  __syncthreads();

  const bool swiglu = (p.act == GV_ACT_SWIGLU);
  const int n_out = swiglu ? (p.N >> 1) : p.N;
  const int o_begin = blockIdx.x * p.outs_per_block;
  const int o_end = min(o_begin + p.outs_per_block, n_out);

  // swiglu: one output per wave-iteration from rows (gate, up); else two outputs
  const int step = swiglu ? 4 : 8;
  for (int o = o_begin + (swiglu ? wave : 2 * wave); o < o_end; o += step) {
    int r0, r1;
    if (swiglu) {
      r0 = ((o >> 4) << 5) + (o & 15);  // gate row in the 16-interleaved weight
      r1 = r0 + 16;                      // matching up row
    } else {
      r0 = o;
      r1 = min(o + 1, p.N - 1);
    }
    const bf16_t* w0 = p.W + (size_t)r0 * p.ldw;
    const bf16_t* w1 = p.W + (size_t)r1 * p.ldw;
    float a0 = 0.f, a1 = 0.f;
    int c = lane;
    for (; c + 192 < nch; c += 256) {
      u32x4 wa[4], wb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        wa[u] = *(const u32x4*)(w0 + (c + 64 * u) * 8);
        wb[u] = *(const u32x4*)(w1 + (c + 64 * u) * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u32x4 xv = *(const u32x4*)(xs + (c + 64 * u) * 8);
        a0 = dot8(wa[u], xv, a0);
        a1 = dot8(wb[u], xv, a1);
      }
    }
    for (; c < nch; c += 64) {
      const u32x4 wa = *(const u32x4*)(w0 + c * 8);
      const u32x4 wb = *(const u32x4*)(w1 + c * 8);
      const u32x4 xv = *(const u32x4*)(xs + c * 8);
      a0 = dot8(wa, xv, a0);
      a1 = dot8(wb, xv, a1);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
      if (swiglu) {
        const float v = a0 / (1.0f + __expf(-a0)) * a1;
        ((bf16_t*)p.y)[o] = f2bf(v);
      } else {
        float v0 = a0, v1 = a1;
        if (p.bias) { v0 += bf2f(p.bias[o]); v1 += bf2f(p.bias[r1]); }
        if (p.R) { v0 += bf2f(p.R[o]); v1 += bf2f(p.R[r1]); }
        if (p.out_f32) {
          ((float*)p.y)[o] = v0;
          if (o + 1 < o_end) ((float*)p.y)[o + 1] = v1;
        } else {
          ((bf16_t*)p.y)[o] = f2bf(v0);
          if (o + 1 < o_end) ((bf16_t*)p.y)[o + 1] = f2bf(v1);
        }
      }
    }
  }
}

extern "C" int vis_gemv_bf16(const void* x, const void* W, const void* bias, const void* R, const void* norm_w,
                             void* y, int N, int K, int ldw, int act, int out_f32, float eps,
                             hipStream_t stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VIS_ERR_ARG;
  if (K % 8 != 0 || ldw % 8 != 0 || K * 2 > 60 * 1024) return VIS_ERR_ARG;
  if (act != GV_ACT_NONE && act != GV_ACT_SWIGLU) return VIS_ERR_ARG;
  if (act == GV_ACT_SWIGLU && (N % 32 != 0 || bias || R || out_f32)) return VIS_ERR_ARG;
  if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)norm_w) & 15) return VIS_ERR_ARG;
  GemvArgs p;
  p.x = (const bf16_t*)x; p.W = (const bf16_t*)W; p.bias = (const bf16_t*)bias; p.R = (const bf16_t*)R;
  p.norm_w = (const bf16_t*)norm_w; p.y = y;
  p.N = N; p.K = K; p.ldw = ldw; p.act = act; p.out_f32 = out_f32; p.eps = eps;
  const int n_out = (act == GV_ACT_SWIGLU) ? N / 2 : N;
  // >= 8 outputs per workgroup (4 waves x 2 rows); aim for ~2048 workgroups on big matrices
  int opb = 8;
  while (opb < 64 && (n_out + opb - 1) / opb > 2048) opb *= 2;
  p.outs_per_block = opb;
  const int blocks = (n_out + opb - 1) / opb;
  vis_clear_error();
  hipLaunchKernelGGL(gemv_bf16_kernel, dim3(blocks), dim3(256), (size_t)K * 2, stream, p);
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// Decode attention: grid (Hkv, nsplit).  ctx = slot_base + *step + 1 keys.
// Partial results (unnormalised o, running max m in log2 domain, sum l) go to
// a workspace and are merged by decode_attn_combine_kernel.
#define DA_MAXG 8
#define DA_MAXKEYS 1024  // keys per split held in LDS

struct DecAttnArgs {
  const bf16_t* q;        // [Hq][128] rotated query of the new token
  const bf16_t* k_cache;  // [Hkv][cache_tokens][128]
  const bf16_t* v_cache;  // [Hkv][cache_tokens][128]
  const int* step_ptr;
  float* part_o;          // [Hq][nsplit][128]
  float* part_ml;         // [Hq][nsplit][2]
  int Hq, Hkv, cache_tokens, slot_base, nsplit;
  float scale_log2;
};

__global__ __launch_bounds__(256) void decode_attn_kernel(DecAttnArgs p) {
  constexpr int HD = 128;
  __shared__ float sc[DA_MAXG][DA_MAXKEYS];
  __shared__ float red[4][DA_MAXG][HD];
  __shared__ float ml[DA_MAXG][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, kq = lane >> 4;
  const int hkv = blockIdx.x, split = blockIdx.y;
  const int G = p.Hq / p.Hkv;
  const int ctx = min(p.slot_base + *p.step_ptr + 1, p.cache_tokens);
  int per = (ctx + p.nsplit - 1) / p.nsplit;
  per = min((per + 15) & ~15, DA_MAXKEYS);
  const int ks = split * per;
  const int ke = min(ks + per, ctx);
  const int nk = max(ke - ks, 0);

  const bf16_t* Kh = p.k_cache + (size_t)hkv * p.cache_tokens * HD;
  const bf16_t* Vh = p.v_cache + (size_t)hkv * p.cache_tokens * HD;

  // phase 1: scores.  16 lanes per key (8 dims each), 16 keys per block iteration
  float qreg[DA_MAXG][8];
#pragma unroll
  for (int g = 0; g < DA_MAXG; ++g) {
    if (g < G) unpack8(*(const u32x4*)(p.q + (size_t)(hkv * G + g) * HD + sub * 8), qreg[g]);
  }
  for (int kk = wave * 4 + kq; kk < nk; kk += 16) {
    float kf[8];
    unpack8(*(const u32x4*)(Kh + (size_t)(ks + kk) * HD + sub * 8), kf);
#pragma unroll
    for (int g = 0; g < DA_MAXG; ++g) {
      if (g < G) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += kf[e] * qreg[g][e];
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (sub == g) sc[g][kk] = s * p.scale_log2;
      }
    }
  }
  __syncthreads();

  // phase 2: per-head softmax statistics over this split (one wave per head)
  for (int g = wave; g < G; g += 4) {
    float mx = -1.0e30f;
    for (int i = lane; i < nk; i += 64) mx = fmaxf(mx, sc[g][i]);
    mx = wave_max(mx);
    float ls = 0.f;
    for (int i = lane; i < nk; i += 64) {
      const float e = exp2f(sc[g][i] - mx);
      sc[g][i] = e;
      ls += e;
    }
    ls = wave_sum(ls);
    if (lane == 0) { ml[g][0] = mx; ml[g][1] = ls; }
  }
  __syncthreads();

  // phase 3: o[g][d] = sum_k p[g][k] V[k][d]; 16 lanes per V row, 4 keys per wave iteration
  float acc[DA_MAXG][8];
#pragma unroll
  for (int g = 0; g < DA_MAXG; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
  for (int kk = wave * 4 + kq; kk < nk; kk += 16) {
    float vf[8];
    unpack8(*(const u32x4*)(Vh + (size_t)(ks + kk) * HD + sub * 8), vf);
#pragma unroll
    for (int g = 0; g < DA_MAXG; ++g) {
      if (g < G) {
        const float pw = sc[g][kk];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] += pw * vf[e];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < DA_MAXG; ++g) {
    if (g < G) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = acc[g][e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kq == 0) red[wave][g][sub * 8 + e] = v;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < G * HD; i += 256) {
    const int g = i / HD, d = i - g * HD;
    const float v = red[0][g][d] + red[1][g][d] + red[2][g][d] + red[3][g][d];
    const int hq = hkv * G + g;
    p.part_o[((size_t)hq * p.nsplit + split) * HD + d] = v;
    if (d == 0) {
      p.part_ml[((size_t)hq * p.nsplit + split) * 2 + 0] = (nk > 0) ? ml[g][0] : -1.0e30f;
      p.part_ml[((size_t)hq * p.nsplit + split) * 2 + 1] = (nk > 0) ? ml[g][1] : 0.f;
    }
  }
}

__global__ __launch_bounds__(128) void decode_attn_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml,
                                                                  bf16_t* __restrict__ out, int nsplit) {
  constexpr int HD = 128;
  const int hq = blockIdx.x, d = threadIdx.x;
  float M = -1.0e30f;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[((size_t)hq * nsplit + s) * 2]);
  float o = 0.f, l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = exp2f(part_ml[((size_t)hq * nsplit + s) * 2] - M);
    l += w * part_ml[((size_t)hq * nsplit + s) * 2 + 1];
    o += w * part_o[((size_t)hq * nsplit + s) * HD + d];
  }
  out[(size_t)hq * HD + d] = f2bf(l > 0.f ? o / l : 0.f);
}

extern "C" int vis_decode_attn(const void* q, const void* k_cache, const void* v_cache, const void* step_ptr,
                               void* part_o, void* part_ml, void* out, int Hq, int Hkv, int HD, int cache_tokens,
                               int slot_base, int nsplit, float scale, hipStream_t stream) {
  if (!q || !k_cache || !v_cache || !step_ptr || !part_o || !part_ml || !out) return VIS_ERR_ARG;
  if (HD != 128 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || Hq / Hkv > DA_MAXG) return VIS_ERR_ARG;
  if (nsplit <= 0 || nsplit > 64) return VIS_ERR_ARG;
  if ((long long)nsplit * DA_MAXKEYS < cache_tokens) return VIS_ERR_ARG;  // every key must be covered
  DecAttnArgs p;
  p.q = (const bf16_t*)q; p.k_cache = (const bf16_t*)k_cache; p.v_cache = (const bf16_t*)v_cache;
  p.step_ptr = (const int*)step_ptr; p.part_o = (float*)part_o; p.part_ml = (float*)part_ml;
  p.Hq = Hq; p.Hkv = Hkv; p.cache_tokens = cache_tokens; p.slot_base = slot_base; p.nsplit = nsplit;
  p.scale_log2 = scale * 1.4426950408889634f;
  vis_clear_error();
  hipLaunchKernelGGL(decode_attn_kernel, dim3(Hkv, nsplit), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(decode_attn_combine_kernel, dim3(Hq), dim3(128), 0, stream, (const float*)part_o,
                     (const float*)part_ml, (bf16_t*)out, nsplit);
  return vis_check_launch();
}

// ---------------------------------------------------------------------------
// Greedy pick.  Stage 1: per-block (max, first index); stage 2: one block merges,
// writes tokens[*step] = argmax, cur_token = argmax and then *step += 1.
// temperature sampling = Gumbel-max: argmax(logit/T + g_i), g_i = -log(-log(u_i)), u_i from a counter hash of
// (seed, step, i).  Exact categorical sampling, no softmax pass, no host round trip, graph-replayable.
__device__ __forceinline__ float gumbel_noise(unsigned seed, unsigned step, unsigned i) {
  unsigned long long z = ((unsigned long long)seed << 32) ^ ((unsigned long long)step * 0x9E3779B97F4A7C15ull) ^ i;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = ((float)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  return -__logf(-__logf(u));
}

__global__ __launch_bounds__(256) void argmax_stage1_kernel(const float* __restrict__ logits, int V,
                                                            float* __restrict__ bval, int* __restrict__ bidx,
                                                            float inv_temp, unsigned seed,
                                                            const int* __restrict__ step_ptr) {
  const int tid = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const unsigned step = (unsigned)*step_ptr;
  for (int i = blockIdx.x * 256 + tid; i < V; i += gridDim.x * 256) {
    float v = logits[i];
    if (inv_temp > 0.f) v = v * inv_temp + gumbel_noise(seed, step, (unsigned)i);
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  __shared__ float sv[4];
  __shared__ int si[4];
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    bval[blockIdx.x] = best;
    bidx[blockIdx.x] = bi;
  }
}

__global__ __launch_bounds__(64) void argmax_stage2_kernel(const float* __restrict__ bval,
                                                           const int* __restrict__ bidx, int nb,
                                                           int* __restrict__ tokens, int max_tokens,
                                                           int* __restrict__ cur_token, int* __restrict__ step_ptr) {
  const int lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < nb; i += 64) {
    const float v = bval[i];
    const int ix = bidx[i];
    if (v > best || (v == best && ix < bi)) { best = v; bi = ix; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) {
    const int st = *step_ptr;
    if (st < max_tokens) tokens[st] = bi;
    *cur_token = bi;
    *step_ptr = st + 1;
  }
}

extern "C" int vis_argmax_f32(const void* logits, int V, void* ws_val, void* ws_idx, void* tokens, int max_tokens,
                              void* cur_token, void* step_ptr, float inv_temp, unsigned seed, hipStream_t stream) {
  if (!logits || V <= 0 || !ws_val || !ws_idx || !tokens || !cur_token || !step_ptr) return VIS_ERR_ARG;
  if (!(inv_temp >= 0.f)) return VIS_ERR_ARG;
  const int nb = min(256, (V + 255) / 256);
  vis_clear_error();
  hipLaunchKernelGGL(argmax_stage1_kernel, dim3(nb), dim3(256), 0, stream, (const float*)logits, V, (float*)ws_val,
                     (int*)ws_idx, inv_temp, seed, (const int*)step_ptr);
  hipLaunchKernelGGL(argmax_stage2_kernel, dim3(1), dim3(64), 0, stream, (const float*)ws_val, (const int*)ws_idx,
                     nb, (int*)tokens, max_tokens, (int*)cur_token, (int*)step_ptr);
  return vis_check_launch();
}
