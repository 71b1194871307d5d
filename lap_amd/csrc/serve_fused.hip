// Fused consumers of split-K GEMM partials for the batch-1 denoise step (lap.py:634-667 through
// gemma.py:336-387 with only the action-expert stream active).  At M = 50 rows every projection is a
// weight-streaming GEMM split over K to fill 256 CUs; instead of a generic reduce kernel followed by separate
// RoPE / GeGLU / gated-residual / adaRMS kernels, the reduction happens inside the consumer:
//   qkv partials      -> sum -> RoPE + q-scale + head split                       (gemma.py:188-218)
//   gate|up partials  -> sum -> GeGLU                                             (gemma.py:303-312)
//   out / down partials -> sum -> gated residual -> (next) adaptive RMSNorm       (gemma.py:577-583,113-131)
// Rounding points are identical to the unfused kernels (the summed projection is rounded to bf16 first), so the
// serving path matches the training-path numerics bit for bit.
#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

__device__ __forceinline__ void st8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void ld8(const bf16* p, float (&v)[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
// sum of ks f32 partial slabs (slab stride = `slab` floats) at 8 consecutive columns, rounded to bf16 like a GEMM output
__device__ __forceinline__ void sum8(const float* part, long long off, int ks, long long slab, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  // loads of 8 (then 4) slabs in flight at a time: these kernels are pure latency chains at batch 1; the additions keep
  // the slab order (bit-identical to splitk_reduce_kernel)
  int s = 0;
  for (; s + 8 <= ks; s += 8) {
    f32x4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off);
      b[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off + 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[u][e]; v[4 + e] += b[u][e]; }
  }
  for (; s + 4 <= ks; s += 4) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off);
      b[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += a[u][e]; v[4 + e] += b[u][e]; }
  }
  for (; s < ks; ++s) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(part + s * slab + off);
    const f32x4 b = *reinterpret_cast<const f32x4*>(part + s * slab + off + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = round_bf16((v[e]));
}

// ---------------------------------------------------------------- reduce + RoPE + split
// sin / cos of every (row, frequency) of a segment, f32 [rows][HD / 2][2]: the suffix positions of a denoise loop do not
// change between its 10 x 18 projections, so the powf / sincosf work is done once (same rope_sincos => same bits)
__global__ __launch_bounds__(256) void rope_table_kernel(const int32_t* __restrict__ pos, float* __restrict__ tab, int rows, int T_seg,
                                                         int T_total, int seg_off, int HD) {
  const int half = HD / 2;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= rows * half) return;
  const int row = gid / half, i = gid % half;
  float sn, cs;
  rope_sincos((float)pos[(long long)(row / T_seg) * T_total + seg_off + row % T_seg], i, HD, sn, cs);
  tab[2 * gid] = sn;
  tab[2 * gid + 1] = cs;
}

__global__ __launch_bounds__(256) void reduce_rope_kernel(const float* __restrict__ part, int ks, const int32_t* __restrict__ pos,
                                                          const float* __restrict__ tab,
                                                          bf16* __restrict__ q, bf16* __restrict__ k, bf16* __restrict__ v,
                                                          int rows, int T_seg, int T_total, int seg_off, int NH, int HD,
                                                          float q_scale) {
  const int cph = HD / 16, tpr = (NH + 2) * cph;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * tpr) return;
  const int row = (int)(gid / tpr), rem = (int)(gid % tpr);
  const int h = rem / cph, c = rem % cph;
  const int b = row / T_seg, t = row % T_seg;
  const int W = (NH + 2) * HD, half = HD / 2;
  const long long slab = (long long)rows * W;
  float x1[8], x2[8], y1[8], y2[8];
  sum8(part, (long long)row * W + h * HD + c * 8, ks, slab, x1);
  sum8(part, (long long)row * W + h * HD + half + c * 8, ks, slab, x2);
  if (h <= NH) {
    const float p = tab ? 0.f : (float)pos[(long long)b * T_total + seg_off + t];
    float sc[16];
    if (tab) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<f32x4*>(sc + 4 * u) = *reinterpret_cast<const f32x4*>(tab + ((long long)row * half + c * 8) * 2 + 4 * u);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float sn, cs, r1, r2;
      if (tab) { sn = sc[2 * e]; cs = sc[2 * e + 1]; }
      else rope_sincos(p, c * 8 + e, HD, sn, cs);
      rope_rotate(x1[e], x2[e], sn, cs, r1, r2);
      r1 = round_bf16((r1)); r2 = round_bf16((r2));
      if (h < NH) { r1 *= q_scale; r2 *= q_scale; }
      y1[e] = r1; y2[e] = r2;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
  }
  bf16* dst = h < NH ? q + (long long)row * NH * HD + h * HD : (h == NH ? k + (long long)row * HD : v + (long long)row * HD);
  st8(dst + c * 8, y1);
  st8(dst + half + c * 8, y2);
}

// ---------------------------------------------------------------- reduce + GeGLU
__global__ __launch_bounds__(256) void reduce_geglu_kernel(const float* __restrict__ part, int ks, bf16* __restrict__ act,
                                                           int rows, int H8) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * H8) return;
  const long long row = gid / H8;
  const int c = (int)(gid % H8) * 8;
  const long long H = (long long)H8 * 8, slab = (long long)rows * 2 * H;
  float g[8], u[8], o[8];
  sum8(part, row * 2 * H + c, ks, slab, g);
  sum8(part, row * 2 * H + H + c, ks, slab, u);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = round_bf16((gelu_tanh_f(g[e]))) * u[e];
  st8(act + row * H + c, o);
}

// ---------------------------------------------------------------- reduce + gated residual + adaptive RMSNorm
// One wave per row (D <= 2048).  xn = x + bf16(y * gate);  h = adaRMS(xn; mod) when mod != null.
template <int NCH>
__global__ __launch_bounds__(256) void reduce_residual_norm_kernel(const float* __restrict__ part, int ks,
                                                                   const bf16* __restrict__ x, const bf16* __restrict__ gate,
                                                                   int ldg, const bf16* __restrict__ mod, int mod_ld,
                                                                   bf16* __restrict__ xn, bf16* __restrict__ hout, int rows,
                                                                   int D, int rps, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const int b = row / rps;
  const long long slab = (long long)rows * D;
  float v[NCH][8];
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      float y[8], xv[8], gv[8];
      sum8(part, (long long)row * D + c, ks, slab, y);
      ld8(x + (long long)row * D + c, xv);
      if (gate) {
        ld8(gate + (long long)b * ldg + c, gv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[p][e] = round_bf16((xv[e] + round_bf16((y[e] * gv[e]))));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[p][e] = round_bf16((xv[e] + y[e]));
      }
      st8(xn + (long long)row * D + c, v[p]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[p][e] * v[p][e];
    }
  }
  if (!mod) return;
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)D + eps);
  const bf16* mrow = mod + (long long)b * mod_ld;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      float sc[8], sh[8], o[8];
      ld8(mrow + c, sc);
      ld8(mrow + D + c, sh);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[p][e] * r * round_bf16((1.0f + sc[e])) + sh[e];
      st8(hout + (long long)row * D + c, o);
    }
  }
}

// ---------------------------------------------------------------- reduce (+ f32 bias) (+ residual) -> bf16, then a row norm
// The serving prefill's consumers (SigLIP fc2 -> next LayerNorm; Gemma out / down projection -> next RMSNorm): the split-K
// reduce pass, the projection's epilogue and the normalisation that follows it in ONE pass over the rows.  One wave per row
// (D <= 2048), lane / chunk assignment and operation order of splitk_reduce_kernel (gemm.hip) followed by rmsnorm_fwd_kernel /
// layernorm_fwd_kernel (norm.hip): the same bits as the three launches it replaces.
//   xn = bf16(sum_s part[s] (+ bias) (+ residual));  NORM 1: h = xn * rstd * (1 + scale)  (gemma.py:113-131, plain form);
//   NORM 2: h = (xn - mean) * (rstd * gamma) + beta  (Flax LayerNorm, use_fast_variance);  NORM 0: no h.
template <int NCH, int NORM>
__global__ __launch_bounds__(NCH * 64) void reduce_norm_kernel(const float* __restrict__ part, int ks, const float* __restrict__ bias,
                                                               const bf16* __restrict__ resid, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16* __restrict__ xn, bf16* __restrict__ hout,
                                                               int rows, int D, float eps) {
  // One BLOCK per row, one wave per 512-column chunk (round 5; before: one wave per row walking its NCH chunks one after the other,
  // 2 ks dependent load batches per wave: 10 - 11.6 us per launch at 560 x 2048, a pure latency chain).  Every wave has all the
  // slab loads of its chunk in flight at once (sum8: 8 slabs x 2 loads), 4 x the blocks.  The row statistics keep the old
  // kernel's order — lane l's running sum over its 8 values of chunk 0, then chunk 1, ... and only then the wave reduction — by
  // passing the rounded values through LDS: every wave redoes lane l's running sum over all chunks, so each holds the same bits.
  __shared__ float sv[NCH][64][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x;
  const long long slab = (long long)rows * D;
  const int c = (lane + 64 * w) * 8;
  const long long off = (long long)row * D + c;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (c < D) {
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = 0.f;
    int s = 0;
    for (; s + 8 <= ks; s += 8) {      // eight slabs in flight; additions in slab order (= splitk_reduce_kernel)
      f32x4 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off);
        b[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off + 4);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] += a[u][e]; y[4 + e] += b[u][e]; }
    }
    for (; s + 4 <= ks; s += 4) {
      f32x4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off);
        b[u] = *reinterpret_cast<const f32x4*>(part + (s + u) * slab + off + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) { y[e] += a[u][e]; y[4 + e] += b[u][e]; }
    }
    for (; s < ks; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(part + s * slab + off), b = *reinterpret_cast<const f32x4*>(part + s * slab + off + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] += a[e]; y[4 + e] += b[e]; }
    }
    if (bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] += bias[c + e];
    }
    if (resid) {
      float r[8];
      ld8(resid + off, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] += r[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = round_bf16(y[e]);
    st8(xn + off, v);
  }
  if (NORM == 0) return;
#pragma unroll
  for (int e = 0; e < 8; ++e) sv[w][lane][e] = v[e];
  __syncthreads();
  float s1 = 0.f, ss = 0.f;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    if ((lane + 64 * p) * 8 < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = sv[p][lane][e]; s1 += t; ss += t * t; }
    }
  }
  ss = wave_sum(ss);
  float mean = 0.f, r;
  if (NORM == 2) {
    s1 = wave_sum(s1);
    mean = s1 / (float)D;
    r = 1.0f / sqrtf(fmaxf(ss / (float)D - mean * mean, 0.f) + eps);
  } else {
    r = 1.0f / sqrtf(ss / (float)D + eps);
  }
  if (c < D) {
    float o[8];
    if (NORM == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[e] - mean) * (r * gamma[c + e]) + beta[c + e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[e] * r * (1.0f + gamma[c + e]);
    }
    st8(hout + (long long)row * D + c, o);
  }
}

// ---------------------------------------------------------------- the sampler's token info words / positions in one launch
// lap.py:624-654 (prefix: make_attn_mask(prefix_mask, prefix_ar), positions cumsum(mask) - 1; suffix queries see every valid
// prefix token and all suffix tokens, positions continue after the prefix) as the per-token words of csrc/attention.hip.
// ~45 tiny torch launches (cat / cumsum / shifts) of 2-10 us each inside the captured sampler before this kernel existed.
struct ServeInfoP {
  const unsigned char* img_mask[4];     // bool [B] per image key
  const unsigned char* prompt_mask;     // bool [B][Lt]
  const unsigned char* langact;         // bool [B][Lt] or null (ar mask of the prompt tokens)
  int32_t *qinfo_p, *kinfo_p, *ppos, *qinfo_s, *kinfo_all, *pos_all;
  int n_img, T, Lt, S, suffix_idx;
};
__global__ __launch_bounds__(64) void serve_infos_kernel(ServeInfoP p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int Pn = p.n_img * p.T + p.Lt, Tk = Pn + p.S;
  int run_m = 0, run_a = 0;
  for (int t0 = 0; t0 < Pn; t0 += 64) {
    const int t = t0 + lane;
    int m = 0, a = 0;
    if (t < Pn) {
      if (t < p.n_img * p.T) m = p.img_mask[t / p.T][b] != 0;
      else {
        m = p.prompt_mask[(long long)b * p.Lt + t - p.n_img * p.T] != 0;
        a = p.langact ? (p.langact[(long long)b * p.Lt + t - p.n_img * p.T] != 0) : 0;
      }
    }
    int cm = m, ca = a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int um = __shfl_up(cm, o, 64), ua = __shfl_up(ca, o, 64);
      if (lane >= o) { cm += um; ca += ua; }
    }
    cm += run_m; ca += run_a;
    if (t < Pn) {
      const int kw = ((m | (m << 1)) << 24) | ca, qw = (m << 24) | ca;
      p.qinfo_p[(long long)b * Pn + t] = qw;
      p.kinfo_p[(long long)b * Pn + t] = kw;
      p.ppos[(long long)b * Pn + t] = cm - 1;
      p.kinfo_all[(long long)b * Tk + t] = kw;
      p.pos_all[(long long)b * Tk + t] = cm - 1;
    }
    run_m = __shfl(cm, 63, 64); run_a = __shfl(ca, 63, 64);
  }
  for (int s = lane; s < p.S; s += 64) {
    p.qinfo_s[(long long)b * p.S + s] = (6 << 24) | p.suffix_idx;
    p.kinfo_all[(long long)b * Tk + Pn + s] = (4 << 24) | p.suffix_idx;
    p.pos_all[(long long)b * Tk + Pn + s] = run_m + s;
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int lap_rope_table(const int32_t* pos, float* table, int B, int T_seg, int T_total, int seg_off, int HD, void* stream) {
  if (!pos || !table || B <= 0 || T_seg <= 0 || (HD & 15)) return LAP_ERR_ARG;
  const int n = B * T_seg * (HD / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, S_, pos, table, B * T_seg, T_seg, T_total, seg_off, HD);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_fused_reduce_rope_split(const float* partials, int ksplit, const int32_t* pos, const float* table, void* q,
                                           void* k, void* v, int B, int T_seg, int T_total, int seg_off, int NH, int HD,
                                           float q_scale, void* stream) {
  if (!partials || ksplit < 1 || B <= 0 || T_seg <= 0 || (HD & 15) || NH <= 0 || (!pos && !table)) return LAP_ERR_ARG;
  const long long n = (long long)B * T_seg * (NH + 2) * (HD / 16);
  hipLaunchKernelGGL(reduce_rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_, partials, ksplit, pos, table, (bf16*)q,
                     (bf16*)k, (bf16*)v, B * T_seg, T_seg, T_total, seg_off, NH, HD, q_scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_fused_reduce_geglu(const float* partials, int ksplit, void* act, int rows, int H, void* stream) {
  if (!partials || ksplit < 1 || rows <= 0 || H <= 0 || (H & 7)) return LAP_ERR_ARG;
  const long long n = (long long)rows * (H / 8);
  hipLaunchKernelGGL(reduce_geglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_, partials, ksplit, (bf16*)act, rows,
                     H / 8);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_fused_reduce_residual_norm(const float* partials, int ksplit, const void* x, const void* gate, int ldg,
                                              const void* mod, int mod_ld, void* xn, void* h, int rows, int D,
                                              int rows_per_sample, float eps, void* stream) {
  if (!partials || ksplit < 1 || rows <= 0 || D <= 0 || (D & 7) || rows_per_sample <= 0 || (ldg & 7) || (mod_ld & 7)) return LAP_ERR_ARG;
  if (mod && !h) return LAP_ERR_ARG;
  const int nch = (D / 8 + 63) / 64;
  dim3 grid((rows + 3) / 4);
#define GO(N) hipLaunchKernelGGL(reduce_residual_norm_kernel<N>, grid, dim3(256), 0, S_, partials, ksplit, (const bf16*)x, \
                                 (const bf16*)gate, ldg, (const bf16*)mod, mod_ld, (bf16*)xn, (bf16*)h, rows, D, rows_per_sample, eps)
  switch (nch) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    default: return LAP_ERR_ARG;
  }
#undef GO
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_fused_reduce_norm(const float* partials, int ksplit, const float* bias, const void* residual, int norm,
                                     const float* gamma, const float* beta, void* xn, void* h, int rows, int D, float eps,
                                     void* stream) {
  if (!partials || ksplit < 1 || !xn || rows <= 0 || D <= 0 || (D & 7) || norm < 0 || norm > 2) return LAP_ERR_ARG;
  if (norm && (!gamma || !h)) return LAP_ERR_ARG;
  if (norm == 2 && !beta) return LAP_ERR_ARG;
  const int nch = (D / 8 + 63) / 64;
  dim3 grid(rows);
#define GO2(N, NM) hipLaunchKernelGGL((reduce_norm_kernel<N, NM>), grid, dim3(N * 64), 0, S_, partials, ksplit, bias, (const bf16*)residual, \
                                      gamma, beta, (bf16*)xn, (bf16*)h, rows, D, eps)
#define GO(N) do { if (norm == 0) GO2(N, 0); else if (norm == 1) GO2(N, 1); else GO2(N, 2); } while (0)
  switch (nch) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    default: return LAP_ERR_ARG;
  }
#undef GO
#undef GO2
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_serve_infos(const void* const* img_masks, int n_img, int T_img, const void* prompt_mask, const void* langact_mask,
                               int B, int Lt, int S, int suffix_idx, int32_t* qinfo_p, int32_t* kinfo_p, int32_t* ppos,
                               int32_t* qinfo_s, int32_t* kinfo_all, int32_t* pos_all, void* stream) {
  if (!img_masks || n_img < 0 || n_img > 4 || T_img <= 0 || !prompt_mask || B <= 0 || Lt <= 0 || S <= 0 || !qinfo_p || !kinfo_p || !ppos ||
      !qinfo_s || !kinfo_all || !pos_all)
    return LAP_ERR_ARG;
  ServeInfoP p = {};
  for (int i = 0; i < n_img; ++i) {
    if (!img_masks[i]) return LAP_ERR_ARG;
    p.img_mask[i] = (const unsigned char*)img_masks[i];
  }
  p.prompt_mask = (const unsigned char*)prompt_mask; p.langact = (const unsigned char*)langact_mask;
  p.qinfo_p = qinfo_p; p.kinfo_p = kinfo_p; p.ppos = ppos; p.qinfo_s = qinfo_s; p.kinfo_all = kinfo_all; p.pos_all = pos_all;
  p.n_img = n_img; p.T = T_img; p.Lt = Lt; p.S = S; p.suffix_idx = suffix_idx;
  hipLaunchKernelGGL(serve_infos_kernel, dim3(B), dim3(64), 0, S_, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
