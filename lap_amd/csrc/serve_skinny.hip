// Skinny-M fused projections for the batch-1 denoise step (lap.py:634-667 -> gemma.py:336-387 with only the
// action-expert stream active): one action chunk is 50 tokens = four 16-token MFMA column tiles, every projection is a
// pure weight stream.  Round 1 ran each projection as a split-K GEMM leaving f32 partial slabs plus a consumer kernel
// (10 launches of 4-12 us per layer); here ONE launch does the whole projection and its neighbours:
//
//   prologue (qkv, gate|up)  adaptive RMSNorm of the block input, recomputed by every block from the full rows it
//                            needs anyway as its operand (gemma.py:113-131): var over the row in f32, x * rsqrt(var+eps)
//                            * bf16(1 + scale) + shift, rounded to bf16;
//   main                     out^T[16 FT features, 16 tokens] = W[16 FT, K] . x^T[K, 16]: the 8 waves of a block split K
//                            (no cross-block reduction, so no partial slabs, no fences, deterministic), operands straight
//                            from global memory to MFMA fragments (weights are read exactly once per launch; x is
//                            L2-resident), partial tiles reduced across the waves through LDS in wave order;
//   epilogue                 qkv: RoPE + q-scale + head split (gemma.py:188-218,548-564) — a block owns 8 frequencies of one
//                            head and BOTH halves (d, d + HD/2) of each rotation pair;
//                            gate|up: GeGLU (gemma.py:303-312) — a block owns 8 gate columns and the same 8 up columns;
//                            out / down: gated residual x + bf16(y * gate) (gemma.py:577-583).
//
// Rounding points are the reference's (bf16 GEMM outputs, bf16 norm outputs, f32 statistics); the K summation order differs
// from the generic GEMM path (8 in-block slices instead of split-K slabs), so the two serving paths agree to bf16
// rounding noise, not bit for bit (tests/test_model_parity_gpu.py states the bound).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

#include "serve_skinny_body.hpp"     // SkinnyP, skinny_load_w, skinny_rest: shared with the persistent chain (serve_chain.hpp)

// grid.x = feature groups, grid.y = token groups
template <int EPI, bool NORM, int KS, int FT, int TT, bool SHM = false, bool NT = false>
__global__ __launch_bounds__(SK_WAVES * 64) void skinny_kernel(SkinnyP p) {
  __shared__ __attribute__((aligned(16))) float part[skinny_part_floats(FT, TT)];
  __shared__ float red[skinny_red_floats(TT)];
  bf16x8 wf[FT][KS];
  skinny_load_w<EPI, KS, FT, NT>(p, blockIdx.x, wf);
  skinny_rest<EPI, NORM, KS, FT, TT, SHM, false>(p, blockIdx.x, blockIdx.y, wf, part, red);
}

// LAP_SKINNY_NT=1: nontemporal weight loads (A/B switch, tools/bench_serve_split.py)
static bool skinny_nt() {
  static const bool nt = getenv("LAP_SKINNY_NT") && atoi(getenv("LAP_SKINNY_NT")) != 0;
  return nt;
}

template <int EPI, bool NORM, int FT, int TT>
int launch_skinny(const SkinnyP& p, int n_sub, hipStream_t s) {
  if (n_sub % FT) return LAP_ERR_ARG;
  const dim3 grid(n_sub / FT, (p.M + SK_TOK * TT - 1) / (SK_TOK * TT)), block(SK_WAVES * 64);
  const bool nt = skinny_nt();
#define GO_(KS_, SHM_) do { if (nt) hipLaunchKernelGGL((skinny_kernel<EPI, NORM, KS_, FT, TT, SHM_, true>), grid, block, 0, s, p); \
                           else hipLaunchKernelGGL((skinny_kernel<EPI, NORM, KS_, FT, TT, SHM_, false>), grid, block, 0, s, p); } while (0)
  if constexpr (NORM) {   // the prologue keeps the whole K slice of a wave in registers: K = 1024 only
    if (p.K != 32 * SK_WAVES * 4) return LAP_ERR_ARG;
    if (p.mod_ld == 0) GO_(4, true); else GO_(4, false);
  } else {
    switch (p.K / (32 * SK_WAVES)) {
      case 4: GO_(4, false); break;
      case 8: GO_(8, false); break;
      case 16: GO_(16, false); break;
      default: return LAP_ERR_ARG;
    }
  }
#undef GO_
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

bool k_ok(int K) { return K == 1024 || K == 2048 || K == 4096; }

// ---------------------------------------------------------------- per-step head / tail of the denoise loop
// tokens = bf16(x_t @ W_in^T + b_in)   (action_in_proj, nnx.Linear f32, lap.py:52; cast to bf16 at gemma.py:494)
__global__ __launch_bounds__(256) void embed_actions_kernel(const float* __restrict__ xt, const float* __restrict__ w, const float* __restrict__ b,
                                                            bf16* __restrict__ out, int rows, int ad, int D) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D) return;
  const int row = (int)(gid / D), c = (int)(gid % D);
  float a = 0.f;
  for (int k = 0; k < ad; ++k) a += xt[row * ad + k] * w[c * ad + k];
  out[gid] = f2bf(a + b[c]);
}

// final adaptive RMSNorm (final_norm_1, gemma.py:525-527) -> action_out_proj in f32 (lap.py:298-299, 661-667) -> Euler
// update x_t += dt * v_t (lap.py:669-672).  One wave per row; W_out f32 [ad][D].
template <int NCH>
__global__ __launch_bounds__(256) void final_norm_out_euler_kernel(const bf16* __restrict__ x, const bf16* __restrict__ mod, int mod_ld, int rps,
                                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                                   float* __restrict__ xt, float* __restrict__ vout, int rows, int D, int ad,
                                                                   float dt, float eps) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wv;
  if (row >= rows) return;
  float v[NCH][8];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = (lane + 64 * q) * 8;
    if (c < D) {
      const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (long long)row * D + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[q][e] = (float)t[e]; ss += v[q][e] * v[q][e]; }
    }
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)D + eps);
  const bf16* mrow = mod + (long long)(row / rps) * mod_ld;
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = (lane + 64 * q) * 8;
    if (c < D) {
      const bf16x8 sc = *reinterpret_cast<const bf16x8*>(mrow + c), sh = *reinterpret_cast<const bf16x8*>(mrow + D + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] = round_bf16(v[q][e] * r * round_bf16(1.0f + (float)sc[e]) + (float)sh[e]);
    }
  }
  for (int a = 0; a < ad; ++a) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = (lane + 64 * q) * 8;
      if (c < D) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + (long long)a * D + c), w1 = *reinterpret_cast<const f32x4*>(w + (long long)a * D + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc += v[q][e] * w0[e] + v[q][4 + e] * w1[e];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const float vt = acc + b[a];
      if (vout) vout[row * ad + a] = vt;
      xt[row * ad + a] += dt * vt;
    }
  }
}

// The same with the NEXT step's action_in_proj behind it (lap_serve_embed_actions: tokens = bf16(x_t W_in^T + b_in)), action_dim 7 / 8, D = 1024:
// the wave that updates a row of x_t holds the whole new row, so it embeds it — one launch per Euler step instead of two, and the 32 dot
// products of the out projection in batches of 8 with their reductions interleaved (the loop above walks them one by one: load ->
// 16 FMAs -> six ds_bpermute round trips, 32 times: 12 us for 50 rows).  wave_sum through permlane swaps / DPP (bit-identical: serve_panel.hip).
template <int CTRL>
__device__ __forceinline__ float dpp_add_(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp_(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  v = dpp_add_<0x128>(v);
  v = dpp_add_<0x124>(v);
  v = dpp_add_<0x4E>(v);
  v = dpp_add_<0xB1>(v);
  return v;
}

template <int NCH, int AD>
__global__ __launch_bounds__(64) void final_euler_embed_kernel(const bf16* __restrict__ x, const bf16* __restrict__ mod, int mod_ld, int rps,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               float* __restrict__ xt, float* __restrict__ vout, int rows, int D,
                                                               float dt, float eps, const float* __restrict__ w_in,
                                                               const float* __restrict__ b_in, bf16* __restrict__ tokens) {
  // One wave per block: 50 rows on 50 CUs.  The launch is a chain of memory round trips (the row the denoise chain just wrote
  // device-scope, the modulation row, both projections' weights, x_t): everything that does not depend on the row is requested
  // FIRST, so the round trips overlap instead of queueing behind the reductions (separately: 12.2 + 4.9 us per Euler step).
  constexpr int AB = AD < 8 ? AD : 8, NC = 16;      // NC: embedding columns per lane (D / 64 <= 32: two passes at D = 2048)
  const int lane = threadIdx.x;
  const int row = blockIdx.x;
  if (row >= rows) return;
  const bf16* mrow = mod + (long long)(row / rps) * mod_ld;
  bf16x8 xr[NCH], sc[NCH], sh[NCH];
  f32x4 wo[AD][NCH][2];
  float bo[AD], xo[AD];
#pragma unroll
  for (int q = 0; q < NCH; ++q) {
    const int c = (lane + 64 * q) * 8;      // (D = 512 NCH: every chunk is inside the row)
    xr[q] = *reinterpret_cast<const bf16x8*>(x + (long long)row * D + c);
    sc[q] = *reinterpret_cast<const bf16x8*>(mrow + c);
    sh[q] = *reinterpret_cast<const bf16x8*>(mrow + D + c);
#pragma unroll
    for (int a = 0; a < AD; ++a) {
      wo[a][q][0] = *reinterpret_cast<const f32x4*>(w + (long long)a * D + c);
      wo[a][q][1] = *reinterpret_cast<const f32x4*>(w + (long long)a * D + c + 4);
    }
  }
#pragma unroll
  for (int a = 0; a < AD; ++a) { bo[a] = b[a]; xo[a] = xt[row * AD + a]; }
  __builtin_amdgcn_sched_barrier(0);

  float v[NCH][8];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < NCH; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[q][e] = (float)xr[q][e]; ss += v[q][e] * v[q][e]; }
  ss = wave_sum_dpp_(ss);
  const float r = 1.0f / sqrtf(ss / (float)D + eps);
#pragma unroll
  for (int q = 0; q < NCH; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[q][e] = round_bf16(v[q][e] * r * round_bf16(1.0f + (float)sc[q][e]) + (float)sh[q][e]);
  float xn[AD];
#pragma unroll
  for (int a0 = 0; a0 < AD; a0 += AB) {
    float acc[AB];
#pragma unroll
    for (int k = 0; k < AB; ++k) {
      acc[k] = 0.f;
      if (a0 + k >= AD) continue;
#pragma unroll
      for (int q = 0; q < NCH; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k] += v[q][e] * wo[a0 + k][q][0][e] + v[q][4 + e] * wo[a0 + k][q][1][e];
    }
#pragma unroll
    for (int k = 0; k < AB; ++k) acc[k] = wave_sum_dpp_(acc[k]);
#pragma unroll
    for (int k = 0; k < AB; ++k) {
      const int a = a0 + k;
      if (a >= AD) continue;
      const float vt = acc[k] + bo[a];
      float xv = xo[a];
      xv += dt * vt;
      xn[a] = xv;
      if (lane == 0) {
        if (vout) vout[row * AD + a] = vt;
        xt[row * AD + a] = xv;
      }
    }
  }
  if (!tokens) return;
  for (int c0 = lane; c0 < D; c0 += 64 * NC) {
    float wi[NC][AD], bi[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = min(c0 + 64 * j, D - 1);
      bi[j] = b_in[c];
      if constexpr (AD % 4 == 0) {
#pragma unroll
        for (int k4 = 0; k4 < AD / 4; ++k4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(w_in + c * AD + 4 * k4);
#pragma unroll
          for (int e = 0; e < 4; ++e) wi[j][4 * k4 + e] = t[e];
        }
      } else {
#pragma unroll
        for (int k = 0; k < AD; ++k) wi[j][k] = w_in[c * AD + k];
      }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = c0 + 64 * j;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < AD; ++k) a += xn[k] * wi[j][k];
      if (c < D) tokens[(long long)row * D + c] = f2bf(a + bi[j]);
    }
  }
}

// one thread per 16-byte chunk of the packed image (serve_skinny_body.hpp PK; the row maps are skinny_load_w's)
__global__ __launch_bounds__(256) void pack_weight_kernel(const bf16* __restrict__ W, bf16* __restrict__ out, int N, int K, int kind, int HD) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * K / 8) return;
  const int lane = (int)(idx & 63), i = lane & 15, g = lane >> 4;
  const long long tile = idx >> 6;
  const int ks = (int)(tile % (K >> 5)), sb = (int)(tile / (K >> 5));
  int wrow;
  if (kind == EPI_ROPE) {
    const int bph = HD / 16, h = sb / bph, j = sb % bph;
    wrow = h * HD + (i < 8 ? j * 8 + i : HD / 2 + j * 8 + (i - 8));
  } else if (kind == EPI_GEGLU) {
    wrow = i < 8 ? sb * 8 + i : N / 2 + sb * 8 + (i - 8);
  } else {
    wrow = sb * 16 + i;
  }
  *reinterpret_cast<bf16x8*>(out + idx * 8) = *reinterpret_cast<const bf16x8*>(W + (long long)wrow * K + ks * 32 + g * 8);
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int lap_serve_pack_weight(const void* w, void* out, int N, int K, int kind, int HD, void* stream) {
  if (!w || !out || w == out || N <= 0 || K <= 0 || (N & 15) || (K & 31) || kind < EPI_ROPE || kind > EPI_RESID) return LAP_ERR_ARG;
  if (kind == EPI_ROPE && (HD < 16 || (HD & 15) || N % HD)) return LAP_ERR_ARG;
  if (kind == EPI_GEGLU && (N & 31)) return LAP_ERR_ARG;
  const long long n = (long long)N * K / 8;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_, (const bf16*)w, (bf16*)out, N, K, kind, HD);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_serve_qkv_rope(const void* x, const void* mod, int mod_ld, int rows_per_sample, const void* wqkv,
                                  const float* rope_table, void* q, void* k, void* v, int M, int D, int NH, int HD,
                                  float q_scale, float eps, void* stream) {
  if (!x || !mod || !wqkv || !rope_table || !q || !k || !v || M <= 0 || rows_per_sample <= 0 || D != 1024 || HD != 256 || NH <= 0 ||
      (mod_ld & 7))
    return LAP_ERR_ARG;
  SkinnyP p = {};
  p.x = (const bf16*)x; p.W = (const bf16*)wqkv; p.M = M; p.N = (NH + 2) * HD; p.K = D; p.ldx = D;
  p.mod = (const bf16*)mod; p.mod_ld = mod_ld; p.rps = rows_per_sample; p.eps = eps;
  p.o0 = (bf16*)q; p.o1 = (bf16*)k; p.o2 = (bf16*)v; p.rope = rope_table; p.NH = NH; p.HD = HD; p.q_scale = q_scale;
  return launch_skinny<EPI_ROPE, true, 2, 2>(p, p.N / 16, S_);
}

extern "C" int lap_serve_gate_up(const void* x, const void* mod, int mod_ld, int rows_per_sample, const void* wgu, void* act,
                                 int M, int D, int H, float eps, void* stream) {
  if (!x || !mod || !wgu || !act || M <= 0 || rows_per_sample <= 0 || D != 1024 || H <= 0 || (H & 31) || (mod_ld & 7)) return LAP_ERR_ARG;
  SkinnyP p = {};
  p.x = (const bf16*)x; p.W = (const bf16*)wgu; p.M = M; p.N = 2 * H; p.K = D; p.ldx = D;
  p.mod = (const bf16*)mod; p.mod_ld = mod_ld; p.rps = rows_per_sample; p.eps = eps;
  p.o0 = (bf16*)act;
  return launch_skinny<EPI_GEGLU, true, 4, 2>(p, H / 8, S_);
}

static int g_resid_ft = 1;
extern "C" int lap_serve_set_variant(int ft) { g_resid_ft = ft; return LAP_OK; }   // tuning knob (tools/bench_skinny.py)

extern "C" int lap_serve_proj_residual(const void* a, const void* w, const void* x, const void* gate, int gate_ld,
                                       int rows_per_sample, void* out, int M, int N, int K, void* stream) {
  if (!a || !w || !x || !out || M <= 0 || rows_per_sample <= 0 || N <= 0 || (N & 15) || !k_ok(K) || (gate_ld & 3)) return LAP_ERR_ARG;
  SkinnyP p = {};
  p.x = (const bf16*)a; p.W = (const bf16*)w; p.M = M; p.N = N; p.K = K; p.ldx = K;
  p.rps = rows_per_sample; p.o0 = (bf16*)out; p.resid = (const bf16*)x; p.gate = (const bf16*)gate; p.gate_ld = gate_ld;
  if (g_resid_ft == 2) return launch_skinny<EPI_RESID, false, 2, 1>(p, N / 16, S_);
  return launch_skinny<EPI_RESID, false, 1, 1>(p, N / 16, S_);
}

extern "C" int lap_serve_embed_actions(const float* x_t, const float* w_in, const float* b_in, void* tokens, int rows, int action_dim,
                                       int D, void* stream) {
  if (!x_t || !w_in || !b_in || !tokens || rows <= 0 || action_dim <= 0 || D <= 0) return LAP_ERR_ARG;
  const long long n = (long long)rows * D;
  hipLaunchKernelGGL(embed_actions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S_, x_t, w_in, b_in, (bf16*)tokens, rows, action_dim, D);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_serve_final_euler_embed(const void* x, const void* mod, int mod_ld, int rows_per_sample, const float* w_out,
                                           const float* b_out, float* x_t, float* v_t, int rows, int D, int action_dim, float dt,
                                           float eps, const float* w_in, const float* b_in, void* tokens, void* stream) {
  if (!x || !mod || !w_out || !b_out || !x_t || rows <= 0 || rows_per_sample <= 0 || D <= 0 || (D & 7) || (mod_ld & 7)) return LAP_ERR_ARG;
  if (action_dim != 7 && action_dim != 8) return LAP_ERR_ARG;      // (the projection weights of a row live in registers: 8 x D / 64 x 2 f32x4)
  if (tokens && (!w_in || !b_in)) return LAP_ERR_ARG;
  if (D != 1024) return LAP_ERR_ARG;                               // LAP-3B's action expert
  const dim3 grid(rows);
#define GO(A) hipLaunchKernelGGL((final_euler_embed_kernel<2, A>), grid, dim3(64), 0, S_, (const bf16*)x, (const bf16*)mod, mod_ld, \
                                 rows_per_sample, w_out, b_out, x_t, v_t, rows, D, dt, eps, w_in, b_in, (bf16*)tokens)
  if (action_dim == 7) GO(7); else GO(8);
#undef GO
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_serve_final_euler(const void* x, const void* mod, int mod_ld, int rows_per_sample, const float* w_out,
                                     const float* b_out, float* x_t, float* v_t, int rows, int D, int action_dim, float dt,
                                     float eps, void* stream) {
  if (!x || !mod || !w_out || !b_out || !x_t || rows <= 0 || rows_per_sample <= 0 || D <= 0 || (D & 7) || action_dim <= 0 || (mod_ld & 7))
    return LAP_ERR_ARG;
  const int nch = (D / 8 + 63) / 64;
  const dim3 grid((rows + 3) / 4);
#define GO(N) hipLaunchKernelGGL(final_norm_out_euler_kernel<N>, grid, dim3(256), 0, S_, (const bf16*)x, (const bf16*)mod, mod_ld, \
                                 rows_per_sample, w_out, b_out, x_t, v_t, rows, D, action_dim, dt, eps)
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
