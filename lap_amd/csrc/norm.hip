// RMSNorm / adaptive RMSNorm (gemma.py:113-131) and LayerNorm (Flax nn.LayerNorm as
// used by SigLIP, siglip_gemma3.py:93,104,167), forward and backward.
// HBM-bound row kernels: one wave per row, 16-byte (8 x bf16) accesses, f32
// statistics.  Per-column parameter gradients are accumulated in registers over
// a group of rows, combined across the 4 waves through LDS and added to HBM with
// one f32 atomic per column per block.
#include <cstdlib>

#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

constexpr int NWAVE = 4;

__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  *reinterpret_cast<bf16x8*>(p) = t;
}

// ------------------------------------------------------------------ RMSNorm fwd
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                          const bf16* __restrict__ mod, bf16* __restrict__ y,
                                                          float* __restrict__ rstd_out, int rows, int D,
                                                          int rows_per_sample, int mod_ld, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * NWAVE + w;
  if (row >= rows) return;
  const bf16* xr = x + (long long)row * D;
  float xv[NCH][8];
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      load8(xr + c, xv[p]);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += xv[p][e] * xv[p][e];
    }
  }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(ss / (float)D + eps);
  if (rstd_out && lane == 0) rstd_out[row] = r;
  const bf16* mrow = mod ? mod + (long long)(row / rows_per_sample) * mod_ld : nullptr;
  bf16* yr = y + (long long)row * D;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      float o[8];
      if (mrow) {
        float sc[8], sh[8];
        load8(mrow + c, sc);
        load8(mrow + D + c, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xv[p][e] * r * round_bf16((1.0f + sc[e])) + sh[e];  // (1 + scale) is a bf16 op in the reference
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xv[p][e] * r * (1.0f + scale[c + e]);
      }
      store8(yr + c, o);
    }
  }
}

// ------------------------------------------------------------------ RMSNorm bwd
// Group g covers rows [g*G, (g+1)*G).  Adaptive: G == rows_per_sample.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ x, const float* __restrict__ scale,
                                                          const bf16* __restrict__ mod, const float* __restrict__ rstd,
                                                          const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                          float* __restrict__ dscale, float* __restrict__ dmod,
                                                          int rows, int D, int G, int mod_ld, int dmod_ld,
                                                          int accum_dx) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NWAVE][2*D] (adaptive) / [NWAVE][D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.x;
  const int r0 = g * G, r1 = min(rows, r0 + G);
  const bool ada = mod != nullptr;
  const bf16* mrow = ada ? mod + (long long)g * mod_ld : nullptr;

  float wgt[NCH][8];
  float ds[NCH][8], dh[NCH][8];
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ds[p][e] = 0.f; dh[p][e] = 0.f; wgt[p][e] = 0.f; }
    if (c < D) {
      if (ada) {
        float sc[8];
        load8(mrow + c, sc);
#pragma unroll
        for (int e = 0; e < 8; ++e) wgt[p][e] = round_bf16((1.0f + sc[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) wgt[p][e] = 1.0f + scale[c + e];
      }
    }
  }
  // one row ahead: the loads of the wave's next row are issued before the reductions / stores of the current one.  Round 6: the row of dx an
  // accumulating call adds onto is requested at the top of its iteration, not behind the wave-wide reduction where nothing covered its latency
  // (17920 x 2048, accumulate: 62 us = 4.7 TB/s before; a row ahead like x / dy it cost the D = 2048 variant its second wave per SIMD)
  bf16x8 nx[NCH], ndy[NCH];
  float nr = 0.f;
  auto fetch = [&](int row) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        nx[p] = *reinterpret_cast<const bf16x8*>(x + (long long)row * D + c);
        ndy[p] = *reinterpret_cast<const bf16x8*>(dy + (long long)row * D + c);
      }
    }
    nr = rstd[row];
  };
  if (r0 + w < r1) fetch(r0 + w);
  for (int row = r0 + w; row < r1; row += NWAVE) {
    bf16x8 cx[NCH], cdy[NCH], cold[NCH];
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      cx[p] = nx[p]; cdy[p] = ndy[p];
      const int c = (lane + 64 * p) * 8;
      if (NCH <= 3 && accum_dx && c < D) cold[p] = *reinterpret_cast<const bf16x8*>(dx + (long long)row * D + c);     // requested HERE: in flight under the row's arithmetic and reduction
      // (NCH = 4, D = 2048: 16 more live registers cost the kernel its second wave per SIMD — it keeps the late request below)
    }
    const float r = nr;
    if (row + NWAVE < r1) fetch(row + NWAVE);
    float xv[NCH][8], gv[NCH][8];
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float dyv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[p][e] = (float)cx[p][e]; dyv[e] = (float)cdy[p][e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          gv[p][e] = dyv[e] * wgt[p][e];
          dot += gv[p][e] * xv[p][e];
          ds[p][e] += dyv[e] * xv[p][e] * r;
          dh[p][e] += dyv[e];
        }
      }
    }
    dot = wave_sum(dot);
    const float cc = dot * r * r * r / (float)D;
    bf16* dxr = dx + (long long)row * D;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = r * gv[p][e] - xv[p][e] * cc;
        if (accum_dx) {
          if constexpr (NCH > 3) cold[p] = *reinterpret_cast<const bf16x8*>(dxr + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += (float)cold[p][e];
        }
        store8(dxr + c, o);
      }
    }
  }
  // combine the per-wave column partials
  const int stride = ada ? 2 * D : D;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[w * stride + c + e] = ds[p][e];
        if (ada) red[w * stride + D + c + e] = dh[p][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < stride; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) t += red[i * stride + c];
    if (ada) {
      // thirds: [0,D) scale, [D,2D) shift
      atomicAdd(dmod + (long long)g * dmod_ld + c, t);
    } else {
      atomicAdd(dscale + c, t);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm fwd
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int rows, int D, float eps) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * NWAVE + w;
  if (row >= rows) return;
  const bf16* xr = x + (long long)row * D;
  float xv[NCH][8];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      load8(xr + c, xv[p]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += xv[p][e]; ss += xv[p][e] * xv[p][e]; }
    }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  const float mean = s / (float)D;
  // Flax default use_fast_variance=True: var = E[x^2] - E[x]^2, clamped at 0.
  const float var = fmaxf(ss / (float)D - mean * mean, 0.f);
  const float r = 1.0f / sqrtf(var + eps);
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = r;
  }
  bf16* yr = y + (long long)row * D;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (xv[p][e] - mean) * (r * gamma[c + e]) + beta[c + e];
      store8(yr + c, o);
    }
  }
}

// ---------------------------------------------------------------- LayerNorm bwd
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int D, int G, int accum_dx, float* __restrict__ dxsum) {
  // dxsum (optional): column sums of the dx this kernel stores (as stored: bf16-rounded) — the bias gradient of the linear
  // layer that produced the normalised tensor's input, which would otherwise cost a pass over dx of its own
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NWAVE][2*D] (+ [NWAVE][D] with dxsum)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * G, r1 = min(rows, r0 + G);
  float gm[NCH][8], dg[NCH][8], db[NCH][8], dxs[NCH][8];
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[p][e] = 0.f; db[p][e] = 0.f; dxs[p][e] = 0.f; gm[p][e] = (c < D) ? gamma[c + e] : 0.f; }
  }
  bf16x8 nx[NCH], ndy[NCH];   // one row ahead, as in rmsnorm_bwd_kernel
  float nmu = 0.f, nr = 0.f;
  auto fetch = [&](int row) {
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        nx[p] = *reinterpret_cast<const bf16x8*>(x + (long long)row * D + c);
        ndy[p] = *reinterpret_cast<const bf16x8*>(dy + (long long)row * D + c);
      }
    }
    nmu = mean[row]; nr = rstd[row];
  };
  if (r0 + w < r1) fetch(r0 + w);
  for (int row = r0 + w; row < r1; row += NWAVE) {
    bf16x8 cx[NCH], cdy[NCH], cold[NCH];
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      cx[p] = nx[p]; cdy[p] = ndy[p];
      const int c = (lane + 64 * p) * 8;
      if (NCH <= 3 && accum_dx && c < D) cold[p] = *reinterpret_cast<const bf16x8*>(dx + (long long)row * D + c);     // (as in rmsnorm_bwd_kernel: in flight under the row's arithmetic)
    }
    const float mu = nmu, r = nr;
    if (row + NWAVE < r1) fetch(row + NWAVE);
    float xh[NCH][8], gv[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float xv[8], dyv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[e] = (float)cx[p][e]; dyv[e] = (float)cdy[p][e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[p][e] = (xv[e] - mu) * r;
          gv[p][e] = dyv[e] * gm[p][e];
          s1 += gv[p][e];
          s2 += gv[p][e] * xh[p][e];
          dg[p][e] += dyv[e] * xh[p][e];
          db[p][e] += dyv[e];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    bf16* dxr = dx + (long long)row * D;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = r * (gv[p][e] - s1 - xh[p][e] * s2);
        if (accum_dx) {
          if constexpr (NCH > 3) cold[p] = *reinterpret_cast<const bf16x8*>(dxr + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += (float)cold[p][e];
        }
        store8(dxr + c, o);
        if (dxsum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) dxs[p][e] += (float)f2bf(o[e]);
        }
      }
    }
  }
  const int NS = dxsum ? 3 : 2;
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[w * NS * D + c + e] = dg[p][e];
        red[w * NS * D + D + c + e] = db[p][e];
        if (dxsum) red[w * NS * D + 2 * D + c + e] = dxs[p][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < NS * D; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) t += red[i * NS * D + c];
    if (c < D) atomicAdd(dgamma + c, t);
    else if (c < 2 * D) atomicAdd(dbeta + (c - D), t);
    else atomicAdd(dxsum + (c - 2 * D), t);
  }
}

inline int nch_for(int D) { return (D / 8 + 63) / 64; }

}  // namespace

#define DISPATCH_NCH(D, CALL)                          \
  switch (nch_for(D)) {                                \
    case 1: { constexpr int NCH = 1; CALL; } break;    \
    case 2: { constexpr int NCH = 2; CALL; } break;    \
    case 3: { constexpr int NCH = 3; CALL; } break;    \
    case 4: { constexpr int NCH = 4; CALL; } break;    \
    default: return LAP_ERR_ARG;                       \
  }

extern "C" int lap_rmsnorm_fwd(const void* x, const float* scale, const void* mod, void* y, float* rstd,
                               int rows, int D, int rows_per_sample, int mod_ld, float eps, void* stream) {
  if (rows <= 0 || D <= 0 || (D & 7) || (!scale && !mod) || (mod && (rows_per_sample <= 0 || (mod_ld & 7) || (mod_ld != 0 && mod_ld < 3 * D))))  // mod_ld == 0: one modulation row shared by all samples
    return LAP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((rows + NWAVE - 1) / NWAVE);
  DISPATCH_NCH(D, hipLaunchKernelGGL(rmsnorm_fwd_kernel<NCH>, grid, dim3(256), 0, s, (const bf16*)x, scale,
                                     (const bf16*)mod, (bf16*)y, rstd, rows, D, rows_per_sample, mod_ld, eps));
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_rmsnorm_bwd(const void* x, const float* scale, const void* mod, const float* rstd, const void* dy,
                               void* dx, float* dscale, float* dmod, int rows, int D, int rows_per_sample,
                               int mod_ld, int dmod_ld, int accum_dx, void* stream) {
  if (rows <= 0 || D <= 0 || (D & 7) || !rstd) return LAP_ERR_ARG;
  if (mod ? (!dmod || rows_per_sample <= 0 || rows % rows_per_sample) : (!scale || !dscale)) return LAP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // rows per block (one f32 atomic per column and block): 17,920 x 2048 takes 89.6 / 75.0 / 69.8 / 59.7 / 89.9 us at 8 / 16 / 32 / 64 / 128
  // (tools/probes/bench_norm_bwd.py; LAP_NORM_BWD_ROWS sweeps it)
  static const int g_env = getenv("LAP_NORM_BWD_ROWS") ? atoi(getenv("LAP_NORM_BWD_ROWS")) : 64;
  const int G = mod ? rows_per_sample : g_env;
  dim3 grid((rows + G - 1) / G);
  const size_t shm = (size_t)NWAVE * (mod ? 2 : 1) * D * sizeof(float);
  DISPATCH_NCH(D, hipLaunchKernelGGL(rmsnorm_bwd_kernel<NCH>, grid, dim3(256), shm, s, (const bf16*)x, scale,
                                     (const bf16*)mod, rstd, (const bf16*)dy, (bf16*)dx, dscale, dmod, rows, D, G,
                                     mod_ld, dmod_ld, accum_dx));
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int rows, int D, float eps, void* stream) {
  if (rows <= 0 || D <= 0 || (D & 7) || !gamma || !beta) return LAP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((rows + NWAVE - 1) / NWAVE);
  DISPATCH_NCH(D, hipLaunchKernelGGL(layernorm_fwd_kernel<NCH>, grid, dim3(256), 0, s, (const bf16*)x, gamma, beta,
                                     (bf16*)y, mean, rstd, rows, D, eps));
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_layernorm_bwd_sum(const void* x, const float* gamma, const float* mean, const float* rstd,
                                     const void* dy, void* dx, float* dgamma, float* dbeta, float* dxsum, int rows, int D,
                                     int accum_dx, void* stream) {
  if (rows <= 0 || D <= 0 || (D & 7) || !gamma || !mean || !rstd || !dgamma || !dbeta) return LAP_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  // rows per block: 16,384 x 1152 takes 75.9 / 54.8 / 43.2 / 49.0 / 76.8 us at 8 / 16 / 32 / 64 / 128 (three atomics per column and block)
  static const int g_env = getenv("LAP_NORM_BWD_ROWS") ? atoi(getenv("LAP_NORM_BWD_ROWS")) : 32;
  const int G = g_env;
  dim3 grid((rows + G - 1) / G);
  const size_t shm = (size_t)NWAVE * (dxsum ? 3 : 2) * D * sizeof(float);
  if (shm > 64 * 1024) return LAP_ERR_ARG;
  DISPATCH_NCH(D, hipLaunchKernelGGL(layernorm_bwd_kernel<NCH>, grid, dim3(256), shm, s, (const bf16*)x, gamma, mean,
                                     rstd, (const bf16*)dy, (bf16*)dx, dgamma, dbeta, rows, D, G, accum_dx, dxsum));
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_layernorm_bwd(const void* x, const float* gamma, const float* mean, const float* rstd,
                                 const void* dy, void* dx, float* dgamma, float* dbeta, int rows, int D, int accum_dx,
                                 void* stream) {
  return lap_layernorm_bwd_sum(x, gamma, mean, rstd, dy, dx, dgamma, dbeta, nullptr, rows, D, accum_dx, stream);
}
