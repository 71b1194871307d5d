// HBM-bound elementwise / gather kernels of the LAP hot path.  Every kernel uses
// 16-byte (8 x bf16 or 4 x f32) accesses and a flat grid; citations name the
// reference lines each one restates.
#include <cstdlib>

#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

__device__ __forceinline__ void ld8(const bf16* p, float (&v)[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void st8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  *reinterpret_cast<bf16x8*>(p) = t;
}

// streaming variants (nontemporal: read once / written once, gigabytes per launch — keep them out of the L2's way)
template <bool NT>
__device__ __forceinline__ void ld8s(const bf16* p, float (&v)[8]) {
  bf16x8 t = NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)) : *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
template <bool NT>
__device__ __forceinline__ void st8s(bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  if (NT) __builtin_nontemporal_store(t, reinterpret_cast<bf16x8*>(p));
  else *reinterpret_cast<bf16x8*>(p) = t;
}

// LAP_STREAM_NT=0: plain loads / stores in the GeGLU / GELU kernels (A/B switch; nontemporal is +3 % on the kernels and -1.4 ms per
// train step, part of it through the L2 space the neighbouring GEMMs keep)
inline bool stream_nt() {
  static const bool nt = getenv("LAP_STREAM_NT") ? atoi(getenv("LAP_STREAM_NT")) != 0 : true;
  return nt;
}

inline dim3 flat_grid(long long n, int per_block = 256) { return dim3((unsigned)((n + per_block - 1) / per_block)); }

// ------------------------------------------------------------------------ RoPE
// gemma.py:548-564: radians = pos / 10000^(2i/HD); [x1 c - x2 s, x2 c + x1 s] in f32 -> bf16;
// gemma.py:216: q *= HD^-0.5 as a bf16 multiply.
// One thread per (row, 8-wide frequency chunk): sin / cos are evaluated once and reused for the NH query heads and the
// key head of the row (they only depend on the position and the frequency index), then the value head is copied.
// HSPLIT (few rows: serving prefill, 560 rows = 35 blocks walking ten heads one after the other, latency bound): one thread per
// (row, head, chunk) instead — the same arithmetic, sin / cos evaluated per head.
template <bool BWD, bool HSPLIT = false>
__global__ __launch_bounds__(256) void rope_split_kernel(const bf16* __restrict__ a0, const bf16* __restrict__ a1,
                                                         const bf16* __restrict__ a2, const int32_t* __restrict__ pos,
                                                         bf16* __restrict__ o0, bf16* __restrict__ o1,
                                                         bf16* __restrict__ o2, int rows, int T_seg, int T_total,
                                                         int seg_off, int NH, int HD, float q_scale) {
  // FWD: a0 = qkv, outputs o0 = q, o1 = k, o2 = v.   BWD: a0 = dq, a1 = dk, a2 = dv, output o0 = dqkv.
  const int cph = HD / 16;               // 8-wide frequency chunks per head = threads per row
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  int h_lo = 0, h_hi = NH + 2;
  if (HSPLIT) {
    if (gid >= (long long)rows * cph * (NH + 2)) return;
    h_lo = (int)(gid % (NH + 2)); h_hi = h_lo + 1;
    gid /= (NH + 2);
  } else if (gid >= (long long)rows * cph) {
    return;
  }
  const int row = (int)(gid / cph);
  const int c = (int)(gid % cph);
  const int b = row / T_seg, t = row % T_seg;
  const int W = (NH + 2) * HD;
  const int half = HD / 2;
  const float p = (float)pos[(long long)b * T_total + seg_off + t];
  float sn[8], cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) rope_sincos(p, c * 8 + e, HD, sn[e], cs[e]);
#pragma unroll 2
  for (int h = h_lo; h < h_hi; ++h) {
    const bf16* src;
    if (!BWD) src = a0 + (long long)row * W + h * HD;
    else if (h < NH) src = a0 + (long long)row * NH * HD + h * HD;
    else if (h == NH) src = a1 + (long long)row * HD;
    else src = a2 + (long long)row * HD;
    float x1[8], x2[8], y1[8], y2[8];
    ld8(src + c * 8, x1);
    ld8(src + half + c * 8, x2);
    if (h <= NH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!BWD) {
          float r1, r2;
          rope_rotate(x1[e], x2[e], sn[e], cs[e], r1, r2);
          r1 = round_bf16((r1)); r2 = round_bf16((r2));
          if (h < NH) { r1 *= q_scale; r2 *= q_scale; }
          y1[e] = r1; y2[e] = r2;
        } else {   // transpose of the rotation
          float d1 = x1[e], d2 = x2[e];
          if (h < NH) { d1 *= q_scale; d2 *= q_scale; }
          rope_rotate(d1, d2, -sn[e], cs[e], y1[e], y2[e]);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
    }
    bf16* dst;
    if (BWD) dst = o0 + (long long)row * W + h * HD;
    else if (h < NH) dst = o0 + (long long)row * NH * HD + h * HD;
    else if (h == NH) dst = o1 + (long long)row * HD;
    else dst = o2 + (long long)row * HD;
    st8(dst + c * 8, y1);
    st8(dst + half + c * 8, y2);
  }
}

// ----------------------------------------------------------------------- GeGLU
// (row strides in elements: ld_gu for gu, ld_act for act / dact, ld_dgu for dgu — the consumers of act and dgu are GEMMs that
// read them through transposing LDS reads, for which a row stride of 32 / 64 KiB is a bad one: lap_geglu_*_ld)
template <bool NT>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16* __restrict__ gu, bf16* __restrict__ act,
                                                        long long nchunk, int H8, long long ld_gu, long long ld_act) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nchunk) return;
  const long long row = gid / H8;
  const int c = (int)(gid % H8) * 8;
  const long long H = (long long)H8 * 8;
  float g[8], u[8], o[8];
  ld8s<NT>(gu + row * ld_gu + c, g);
  ld8s<NT>(gu + row * ld_gu + H + c, u);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = round_bf16((gelu_tanh_f(g[e]))) * u[e];  // gelu output is a bf16 tensor upstream
  st8s<NT>(act + row * ld_act + c, o);
}
template <bool NT>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16* __restrict__ gu, const bf16* __restrict__ dact,
                                                        bf16* __restrict__ dgu, long long nchunk, int H8, long long ld_gu,
                                                        long long ld_act, long long ld_dgu) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nchunk) return;
  const long long row = gid / H8;
  const int c = (int)(gid % H8) * 8;
  const long long H = (long long)H8 * 8;
  float g[8], u[8], d[8], dg[8], du[8];
  ld8s<NT>(gu + row * ld_gu + c, g);
  ld8s<NT>(gu + row * ld_gu + H + c, u);
  ld8s<NT>(dact + row * ld_act + c, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dg[e] = d[e] * u[e] * gelu_tanh_grad_f(g[e]);
    du[e] = d[e] * round_bf16((gelu_tanh_f(g[e])));
  }
  st8s<NT>(dgu + row * ld_dgu + c, dg);
  st8s<NT>(dgu + row * ld_dgu + H + c, du);
}

template <bool NT>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long n8) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n8) return;
  float v[8];
  ld8s<NT>(x + gid * 8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
  st8s<NT>(y + gid * 8, v);
}
template <bool NT>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                       bf16* __restrict__ dx, long long n8) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n8) return;
  float v[8], d[8];
  ld8s<NT>(x + gid * 8, v);
  ld8s<NT>(dy + gid * 8, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) d[e] *= gelu_tanh_grad_f(v[e]);
  st8s<NT>(dx + gid * 8, d);
}

// ------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ tok,
                                                           bf16* __restrict__ out, int rows, int T, int D8,
                                                           int dst_rps, int dst_off, float scale, int row_lo, int row_hi) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D8) return;
  const int r = (int)(gid / D8), c = (int)(gid % D8) * 8;
  const long long D = (long long)D8 * 8;
  const int t = tok[r];
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (t >= row_lo && t < row_hi) {   // rows owned by another FSDP shard contribute zeros (summed by reduce-scatter)
    const float* src = table + (long long)(t - row_lo) * D + c;
    f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = a[e] * scale; o[4 + e] = b[e] * scale; }
  }
  const long long drow = (long long)(r / T) * dst_rps + dst_off + r % T;
  st8(out + drow * D + c, o);
}
__global__ __launch_bounds__(256) void embed_scatter_kernel(float* __restrict__ dtable, const int32_t* __restrict__ tok,
                                                            const bf16* __restrict__ dout, int rows, int T, int D8,
                                                            int src_rps, int src_off, float scale) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D8) return;
  const int r = (int)(gid / D8), c = (int)(gid % D8) * 8;
  const long long D = (long long)D8 * 8;
  const long long srow = (long long)(r / T) * src_rps + src_off + r % T;
  float d[8];
  ld8(dout + srow * D + c, d);
  float* dst = dtable + (long long)tok[r] * D + c;
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(dst + e, d[e] * scale);
}

// -------------------------------------------------------------- gated residual
__global__ __launch_bounds__(256) void gated_res_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ u,
                                                            const bf16* __restrict__ gate, bf16* __restrict__ y,
                                                            int rows, int D8, int rps, int ldg) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D8) return;
  const int r = (int)(gid / D8), c = (int)(gid % D8) * 8;
  const long long D = (long long)D8 * 8;
  float xv[8], uv[8], o[8];
  ld8(x + r * D + c, xv);
  ld8(u + r * D + c, uv);
  if (gate) {
    float gv[8];
    ld8(gate + (long long)(r / rps) * ldg + c, gv);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = xv[e] + round_bf16((uv[e] * gv[e]));  // y*gate is a bf16 product upstream
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = xv[e] + uv[e];
  }
  st8(y + r * D + c, o);
}
// du = dy * gate;  dgate[b] = sum over the sample's rows of dy * u.  One block per (sample, 32 chunks of 8 columns): 32 column
// lanes x 8 ROW lanes — row lane j takes rows j, j + 8, ... with all its loads in flight, then the 8 partial sums meet in LDS and
// are added in row-lane order (deterministic).  (Round 5; before: one thread per chunk walking the sample's 50 rows one after the
// other, 32 blocks: 29 - 81 us per launch on the action expert's stream.)
__global__ __launch_bounds__(256) void gated_res_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ u,
                                                            const bf16* __restrict__ gate, bf16* __restrict__ du,
                                                            float* __restrict__ dgate, int D8, int rps, int ldg,
                                                            int ldg_out) {
  __shared__ float part[8][32][8];
  const int b = blockIdx.y;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c8 = blockIdx.x * 32 + cl;
  const bool live = c8 < D8;
  const int c = c8 * 8;
  const long long D = (long long)D8 * 8;
  float gv[8], acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[e] = 0.f; gv[e] = 0.f; }
  if (live) {
    ld8(gate + (long long)b * ldg + c, gv);
    constexpr int U = 4;       // rows in flight per thread
    for (int t0 = rl; t0 < rps; t0 += 8 * U) {
      bf16x8 dv[U], uv[U];
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int t = t0 + 8 * i;
        if (t < rps) {
          const long long r = (long long)b * rps + t;
          dv[i] = *reinterpret_cast<const bf16x8*>(dy + r * D + c);
          uv[i] = *reinterpret_cast<const bf16x8*>(u + r * D + c);
        }
      }
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int t = t0 + 8 * i;
        if (t < rps) {
          const long long r = (long long)b * rps + t;
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = (float)dv[i][e]; o[e] = d * gv[e]; acc[e] += d * (float)uv[i][e]; }
          st8(du + r * D + c, o);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rl][cl][e] = acc[e];
  __syncthreads();
  if (rl == 0 && live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += part[j][cl][e];
      dgate[(long long)b * ldg_out + c + e] = t;
    }
  }
}

// ----------------------------------------------------------------- casts/copies
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(x + i + 4);
    float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    st8(y + i, o);
  } else {
    for (long long j = i; j < n; ++j) y[j] = f2bf(x[j]);
  }
}
// x f32 [rows][cols] (row stride ld) -> hi = bf16(x), lo = bf16(x - hi), both [rows][ld_out] with zero padding in the
// columns [cols, ld_out).  hi + lo carries 16 mantissa bits of x: products of such pairs on the bf16 MFMA path, f32
// accumulated, reproduce an f32 GEMM to ~2^-17 relative (used for the f32 SigLIP stem, siglip_gemma3.py:398-408).
__global__ __launch_bounds__(256) void split_hilo_kernel(const float* __restrict__ x, int rows, int cols, int ld,
                                                         bf16* __restrict__ hi, bf16* __restrict__ lo, int ld_out) {
  const int cpr = ld_out / 8;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * cpr) return;
  const int r = (int)(gid / cpr), c0 = (int)(gid % cpr) * 8;
  float h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = (c0 + e < cols) ? x[(long long)r * ld + c0 + e] : 0.f;
    h[e] = round_bf16(v);
    l[e] = v - h[e];
  }
  st8(hi + (long long)r * ld_out + c0, h);
  st8(lo + (long long)r * ld_out + c0, l);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float v[8];
    ld8(x + i, v);
    *reinterpret_cast<f32x4*>(y + i) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(y + i + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    for (long long j = i; j < n; ++j) y[j] = bf2f(x[j]);
  }
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                       bf16* __restrict__ y, long long n) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float u[8], v[8];
    ld8(a + i, u); ld8(b + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] += v[e];
    st8(y + i, u);
  } else {
    for (long long j = i; j < n; ++j) y[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
  }
}
__global__ __launch_bounds__(256) void copy2d_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int rows,
                                                     int cols8, int lds_, int ldd) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * cols8) return;
  const int r = (int)(gid / cols8), c = (int)(gid % cols8) * 8;
  *reinterpret_cast<bf16x8*>(dst + (long long)r * ldd + c) = *reinterpret_cast<const bf16x8*>(src + (long long)r * lds_ + c);
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int rows,
                                                        int T, int D8, int src_rps, int src_off, int dst_rps,
                                                        int dst_off, int accumulate) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D8) return;
  const int r = (int)(gid / D8), c = (int)(gid % D8) * 8;
  const long long D = (long long)D8 * 8;
  const long long srow = (long long)(r / T) * src_rps + src_off + r % T;
  const long long drow = (long long)(r / T) * dst_rps + dst_off + r % T;
  if (accumulate) {
    float a[8], b[8];
    ld8(src + srow * D + c, a);
    ld8(dst + drow * D + c, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    st8(dst + drow * D + c, a);
  } else {
    *reinterpret_cast<bf16x8*>(dst + drow * D + c) = *reinterpret_cast<const bf16x8*>(src + srow * D + c);
  }
}

// ------------------------------------------------------------------ column sums
// Block = 4 waves over a slab of CS_ROWS rows x 512 columns: lane owns 8 adjacent columns (16-byte loads), waves
// stride over rows, partials meet in LDS, one f32 atomic per column per block.
constexpr int CS_ROWS = 256;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int rows, int cols,
                                                     int ld) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool vec = (c0 + 8 <= cols) && ((ld * (int)sizeof(T)) % 16 == 0) && ((((uintptr_t)x) & 15) == 0) && (c0 * (int)sizeof(T)) % 16 == 0;
  if (vec) {
    // 8 row loads in flight per wave (rows r, r+4, .. r+28), then the adds: the plain loop had one load in flight
    for (int r = r0 + w; r < r1; r += 32) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r + 4 * u;
          v[u] = rr < r1 ? *reinterpret_cast<const bf16x8*>(x + (long long)rr * ld + c0) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
      } else {
        f32x4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = r + 4 * u;
          const bool ok = rr < r1;
          a[u] = ok ? *reinterpret_cast<const f32x4*>(x + (long long)rr * ld + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
          b[u] = ok ? *reinterpret_cast<const f32x4*>(x + (long long)rr * ld + c0 + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[e] += a[u][e]; acc[4 + e] += b[u][e]; }
      }
    }
  } else {
    for (int r = r0 + w; r < r1; r += 4) {
      const T* row = x + (long long)r * ld;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < cols) acc[e] += (float)row[c0 + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[w][lane * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int col = blockIdx.x * 512 + c;
    if (col < cols) atomicAdd(out + col, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
  }
}

// ------------------------------------------------------------------ SigLIP stem
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H,
                                                     int W, int C, int P) {
  const int GH = H / P, GW = W / P;
  const int PC = P * C;            // one patch row: contiguous in the NHWC image
  const long long total = (long long)B * GH * GW * P * PC;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int x = (int)(gid % PC);
  long long t = gid / PC;
  const int ph = (int)(t % P); t /= P;
  const int gw = (int)(t % GW); t /= GW;
  const int gh = (int)(t % GH);
  const int b = (int)(t / GH);
  out[gid] = img[(((long long)b * H + gh * P + ph) * W + gw * P) * C + x];
}
__global__ __launch_bounds__(256) void add_posemb_cast_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                              bf16* __restrict__ y, int rows, int T, int D8) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * D8) return;
  const int r = (int)(gid / D8), c = (int)(gid % D8) * 8;
  const long long D = (long long)D8 * 8;
  const float* xs = x + r * D + c;
  const float* ps = pos + (long long)(r % T) * D + c;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = xs[e] + ps[e];
  st8(y + r * D + c, o);
}
// grid (D8 chunks / 256, T): thread owns one (t, 8 columns), loops over the batch.
__global__ __launch_bounds__(256) void add_posemb_cast_bwd_kernel(const bf16* __restrict__ dy, float* __restrict__ dx,
                                                                  float* __restrict__ dpos, int nb, int T, int D8) {
  const int t = blockIdx.y;
  const int c8 = blockIdx.x * 256 + threadIdx.x;
  if (c8 >= D8) return;
  const int c = c8 * 8;
  const long long D = (long long)D8 * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int b = 0; b < nb; ++b) {
    const long long r = (long long)b * T + t;
    float d[8];
    ld8(dy + r * D + c, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc[e] += d[e]; dx[r * D + c + e] = d[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) dpos[(long long)t * D + c + e] += acc[e];
}

// ---------------------------------------------------------------- flow matching
__global__ __launch_bounds__(256) void fm_mix_kernel(const float* __restrict__ noise, const float* __restrict__ act,
                                                     const float* __restrict__ t, float* __restrict__ x_t,
                                                     float* __restrict__ u_t, int B, int n_per) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * n_per) return;
  const float tt = t[gid / n_per];
  const float e = noise[gid], a = act[gid];
  x_t[gid] = tt * e + (1.0f - tt) * a;
  u_t[gid] = e - a;
}
// openpi pi0.posemb_sincos [UPSTREAM-RECALL]: fraction = linspace(0,1,D/2); period = min*(max/min)^fraction;
// out = concat[sin(t * 2pi / period), cos(...)].
__global__ __launch_bounds__(256) void posemb_sincos_kernel(const float* __restrict__ t, float* __restrict__ out, int B,
                                                            int D, float min_p, float max_p) {
  const int half = D / 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * half) return;
  const int b = (int)(gid / half), i = (int)(gid % half);
  const float frac = (half > 1) ? (float)i / (float)(half - 1) : 0.f;
  const float period = min_p * powf(max_p / min_p, frac);
  const float ang = t[b] * (6.283185307179586f / period);
  float sn, cs;
  sincosf(ang, &sn, &cs);
  out[(long long)b * D + i] = sn;
  out[(long long)b * D + half + i] = cs;
}
__global__ __launch_bounds__(256) void swish_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = v / (1.0f + expf(-v));
}
__global__ __launch_bounds__(256) void swish_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const float s = 1.0f / (1.0f + expf(-v));
  dx[i] = dy[i] * (s + v * s * (1.0f - s));
}
// one block per sample
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ v, const float* __restrict__ u,
                                                  const float* __restrict__ coef, float* __restrict__ per_sample,
                                                  float* __restrict__ dv, int n_per) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float cf = coef ? coef[b] : 0.f;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_per; i += 256) {
    const float d = v[(long long)b * n_per + i] - u[(long long)b * n_per + i];
    acc += d * d;
    if (dv) dv[(long long)b * n_per + i] = cf * 2.0f * d / (float)n_per;
  }
  acc = block_sum<4>(acc, red);
  if (threadIdx.x == 0) per_sample[b] = acc / (float)n_per;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ x, const float* __restrict__ v, float dt, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += dt * v[i];
}

// ---------------------------------------------------------------- train-time image augmentation
// models/model_adapter.py:118-151: augmax.Chain(RandomCrop(95 %), Resize(full), Rotate(+-5 deg), ColorJitter(0.2, 0.2, 0.2))
// per sample, on [0, 1] images.  augmax composes the geometric transforms into ONE coordinate map and samples the source
// once with bilinear interpolation; so does this kernel: output pixel -> rotation about the image centre -> scale of the
// crop window -> crop offset -> 4-tap gather (outside the source: 0).  Colour on the sampled pixel: brightness, contrast
// (piecewise-linear tone curve around 0.5), saturation (S channel of HSV).  [UPSTREAM-RECALL]: augmax is not in the image;
// the random parameters come from the caller.  par[b] = {ox, oy, cw, ch, cos, sin, brightness, contrast, saturation, skip, -, -}.
__device__ __forceinline__ float aug_brightness(float v, float b) { return b < 0.f ? v * (1.f + b) : v * (1.f - b) + b; }
__device__ __forceinline__ float aug_contrast(float v, float c) {
  const float slant = tanf((c + 1.f) * 0.78539816339744831f);
  if (fabsf(slant - 1.f) < 1e-6f) return v;
  const float p1 = (slant - slant * slant) / (2.f * (1.f - slant * slant)), p2 = 1.f - p1;
  if (v < p1) return v / slant;
  if (v > p2) return v / slant + 1.f - 1.f / slant;
  return slant * (v - 0.5f) + 0.5f;
}
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                      const float* __restrict__ par, int B, int H, int W) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * H * W) return;
  const int x = (int)(gid % W), y = (int)((gid / W) % H), b = (int)(gid / ((long long)W * H));
  const float* q = par + b * 12;
  const float* src = img + (long long)b * H * W * 3;
  float* dst = out + gid * 3;
  if (q[9] != 0.f) {   // sample excluded from augmentation (VQA samples): copy
    const float* s3 = src + ((long long)y * W + x) * 3;
    dst[0] = s3[0]; dst[1] = s3[1]; dst[2] = s3[2];
    return;
  }
  const float xc = (float)x + 0.5f - 0.5f * (float)W, yc = (float)y + 0.5f - 0.5f * (float)H;
  const float xr = q[4] * xc - q[5] * yc, yr = q[5] * xc + q[4] * yc;
  const float xs = q[0] + 0.5f * q[2] + xr * (q[2] / (float)W) - 0.5f;
  const float ys = q[1] + 0.5f * q[3] + yr * (q[3] / (float)H) - 0.5f;
  const float xf = floorf(xs), yf = floorf(ys);
  const int x0 = (int)xf, y0 = (int)yf;
  const float ax = xs - xf, ay = ys - yf;
  float px[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = x0 + dx, yy = y0 + dy;
      if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
      const float wgt = (dx ? ax : 1.f - ax) * (dy ? ay : 1.f - ay);
      const float* s3 = src + ((long long)yy * W + xx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] += wgt * (s3[c] * 0.5f + 0.5f);   // [-1, 1] -> [0, 1]
    }
  float r = px[0], g = px[1], bl = px[2];
  r = aug_brightness(r, q[6]); g = aug_brightness(g, q[6]); bl = aug_brightness(bl, q[6]);
  r = aug_contrast(r, q[7]); g = aug_contrast(g, q[7]); bl = aug_contrast(bl, q[7]);
  {   // saturation: scale S of HSV, i.e. move every channel towards / away from the maximum channel
    const float mx = fmaxf(r, fmaxf(g, bl)), mn = fminf(r, fminf(g, bl));
    const float sat = mx > 0.f ? (mx - mn) / mx : 0.f;
    const float s2 = fminf(fmaxf(aug_brightness(sat, q[8]), 0.f), 1.f);
    const float k = sat > 0.f ? s2 / sat : 0.f;
    r = mx - (mx - r) * k; g = mx - (mx - g) * k; bl = mx - (mx - bl) * k;
  }
  dst[0] = fminf(fmaxf(r, 0.f), 1.f) * 2.f - 1.f;
  dst[1] = fminf(fmaxf(g, 0.f), 1.f) * 2.f - 1.f;
  dst[2] = fminf(fmaxf(bl, 0.f), 1.f) * 2.f - 1.f;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int lap_abi_version(void) { return LAP_ABI_VERSION; }

extern "C" int lap_rope_split_fwd(const void* qkv, const int32_t* pos, void* q, void* k, void* v, int B, int T_seg,
                                  int T_total, int seg_off, int NH, int HD, float q_scale, void* stream) {
  if (B <= 0 || T_seg <= 0 || (HD & 15) || NH <= 0) return LAP_ERR_ARG;
  const long long n = (long long)B * T_seg * (HD / 16);
  if (n < 65536)   // fewer than 256 blocks: one thread per head as well
    hipLaunchKernelGGL((rope_split_kernel<false, true>), flat_grid(n * (NH + 2)), dim3(256), 0, S_, (const bf16*)qkv, nullptr, nullptr, pos,
                       (bf16*)q, (bf16*)k, (bf16*)v, B * T_seg, T_seg, T_total, seg_off, NH, HD, q_scale);
  else
    hipLaunchKernelGGL(rope_split_kernel<false>, flat_grid(n), dim3(256), 0, S_, (const bf16*)qkv, nullptr, nullptr, pos,
                       (bf16*)q, (bf16*)k, (bf16*)v, B * T_seg, T_seg, T_total, seg_off, NH, HD, q_scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_rope_split_bwd(const void* dq, const void* dk, const void* dv, const int32_t* pos, void* dqkv, int B,
                                  int T_seg, int T_total, int seg_off, int NH, int HD, float q_scale, void* stream) {
  if (B <= 0 || T_seg <= 0 || (HD & 15) || NH <= 0) return LAP_ERR_ARG;
  const long long n = (long long)B * T_seg * (HD / 16);
  hipLaunchKernelGGL(rope_split_kernel<true>, flat_grid(n), dim3(256), 0, S_, (const bf16*)dq, (const bf16*)dk,
                     (const bf16*)dv, pos, (bf16*)dqkv, nullptr, nullptr, B * T_seg, T_seg, T_total, seg_off, NH, HD,
                     q_scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_geglu_fwd_ld(const void* gu, void* act, int rows, int H, int ld_gu, int ld_act, void* stream) {
  if (rows <= 0 || H <= 0 || (H & 7) || ld_gu < 2 * H || ld_act < H || (ld_gu & 7) || (ld_act & 7)) return LAP_ERR_ARG;
  const long long n = (long long)rows * (H / 8);
  const bool nt = stream_nt();
  if (nt) hipLaunchKernelGGL(geglu_fwd_kernel<true>, flat_grid(n), dim3(256), 0, S_, (const bf16*)gu, (bf16*)act, n, H / 8, (long long)ld_gu, (long long)ld_act);
  else hipLaunchKernelGGL(geglu_fwd_kernel<false>, flat_grid(n), dim3(256), 0, S_, (const bf16*)gu, (bf16*)act, n, H / 8, (long long)ld_gu, (long long)ld_act);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_geglu_fwd(const void* gu, void* act, int rows, int H, void* stream) {
  return lap_geglu_fwd_ld(gu, act, rows, H, 2 * H, H, stream);
}
extern "C" int lap_geglu_bwd_ld(const void* gu, const void* dact, void* dgu, int rows, int H, int ld_gu, int ld_dact, int ld_dgu, void* stream) {
  if (rows <= 0 || H <= 0 || (H & 7) || ld_gu < 2 * H || ld_dact < H || ld_dgu < 2 * H || ((ld_gu | ld_dact | ld_dgu) & 7)) return LAP_ERR_ARG;
  const long long n = (long long)rows * (H / 8);
  const bool nt = stream_nt();
  if (nt) hipLaunchKernelGGL(geglu_bwd_kernel<true>, flat_grid(n), dim3(256), 0, S_, (const bf16*)gu, (const bf16*)dact, (bf16*)dgu, n,
                             H / 8, (long long)ld_gu, (long long)ld_dact, (long long)ld_dgu);
  else hipLaunchKernelGGL(geglu_bwd_kernel<false>, flat_grid(n), dim3(256), 0, S_, (const bf16*)gu, (const bf16*)dact, (bf16*)dgu, n,
                          H / 8, (long long)ld_gu, (long long)ld_dact, (long long)ld_dgu);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_geglu_bwd(const void* gu, const void* dact, void* dgu, int rows, int H, void* stream) {
  return lap_geglu_bwd_ld(gu, dact, dgu, rows, H, 2 * H, H, 2 * H, stream);
}
extern "C" int lap_gelu_fwd(const void* x, void* y, long long n, void* stream) {
  if (n <= 0 || (n & 7)) return LAP_ERR_ARG;
  if (stream_nt()) hipLaunchKernelGGL(gelu_fwd_kernel<true>, flat_grid(n / 8), dim3(256), 0, S_, (const bf16*)x, (bf16*)y, n / 8);
  else hipLaunchKernelGGL(gelu_fwd_kernel<false>, flat_grid(n / 8), dim3(256), 0, S_, (const bf16*)x, (bf16*)y, n / 8);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_gelu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream) {
  if (n <= 0 || (n & 7)) return LAP_ERR_ARG;
  if (stream_nt()) hipLaunchKernelGGL(gelu_bwd_kernel<true>, flat_grid(n / 8), dim3(256), 0, S_, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n / 8);
  else hipLaunchKernelGGL(gelu_bwd_kernel<false>, flat_grid(n / 8), dim3(256), 0, S_, (const bf16*)x, (const bf16*)dy, (bf16*)dx, n / 8);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_embed_gather(const float* table, const int32_t* tok, void* out, int rows, int T, int D,
                                int dst_rows_per_sample, int dst_off, float scale, int row_lo, int row_hi, void* stream) {
  if (rows <= 0 || T <= 0 || (D & 7) || row_hi < row_lo) return LAP_ERR_ARG;
  hipLaunchKernelGGL(embed_gather_kernel, flat_grid((long long)rows * (D / 8)), dim3(256), 0, S_, table, tok, (bf16*)out,
                     rows, T, D / 8, dst_rows_per_sample, dst_off, scale, row_lo, row_hi);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_embed_scatter_add(float* dtable, const int32_t* tok, const void* dout, int rows, int T, int D,
                                     int src_rows_per_sample, int src_off, float scale, void* stream) {
  if (rows <= 0 || T <= 0 || (D & 7)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(embed_scatter_kernel, flat_grid((long long)rows * (D / 8)), dim3(256), 0, S_, dtable, tok,
                     (const bf16*)dout, rows, T, D / 8, src_rows_per_sample, src_off, scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_gated_residual_fwd(const void* x, const void* u, const void* gate, void* y, int rows, int D,
                                      int rows_per_sample, int ldg, void* stream) {
  if (rows <= 0 || (D & 7) || (gate && (rows_per_sample <= 0 || (ldg & 7)))) return LAP_ERR_ARG;
  hipLaunchKernelGGL(gated_res_fwd_kernel, flat_grid((long long)rows * (D / 8)), dim3(256), 0, S_, (const bf16*)x,
                     (const bf16*)u, (const bf16*)gate, (bf16*)y, rows, D / 8, rows_per_sample > 0 ? rows_per_sample : 1,
                     ldg);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_gated_residual_bwd(const void* dy, const void* u, const void* gate, void* du, float* dgate, int rows,
                                      int D, int rows_per_sample, int ldg, int ldg_out, void* stream) {
  if (rows <= 0 || (D & 7) || !gate || rows_per_sample <= 0 || rows % rows_per_sample || (ldg & 7)) return LAP_ERR_ARG;
  dim3 grid((D / 8 + 31) / 32, rows / rows_per_sample);
  hipLaunchKernelGGL(gated_res_bwd_kernel, grid, dim3(256), 0, S_, (const bf16*)dy, (const bf16*)u, (const bf16*)gate,
                     (bf16*)du, dgate, D / 8, rows_per_sample, ldg, ldg_out);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_colsum_bf16(const void* x, float* out, int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(colsum_kernel<bf16>, dim3((cols + 511) / 512, (rows + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, S_, (const bf16*)x,
                     out, rows, cols, ld);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_colsum_f32(const float* x, float* out, int rows, int cols, int ld, void* stream) {
  if (rows <= 0 || cols <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(colsum_kernel<float>, dim3((cols + 511) / 512, (rows + CS_ROWS - 1) / CS_ROWS), dim3(256), 0, S_, x, out, rows,
                     cols, ld);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_cast_f32_to_bf16(const float* x, void* y, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, flat_grid((n + 7) / 8), dim3(256), 0, S_, x, (bf16*)y, n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_split_f32_hilo(const float* x, int rows, int cols, int ld, void* hi, void* lo, int ld_out, void* stream) {
  if (rows <= 0 || cols <= 0 || ld < cols || ld_out < cols || (ld_out & 7) || !x || !hi || !lo) return LAP_ERR_ARG;
  hipLaunchKernelGGL(split_hilo_kernel, flat_grid((long long)rows * (ld_out / 8)), dim3(256), 0, S_, x, rows, cols, ld, (bf16*)hi,
                     (bf16*)lo, ld_out);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_cast_bf16_to_f32(const void* x, float* y, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, flat_grid((n + 7) / 8), dim3(256), 0, S_, (const bf16*)x, y, n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_add_bf16(const void* a, const void* b, void* y, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(add_bf16_kernel, flat_grid((n + 7) / 8), dim3(256), 0, S_, (const bf16*)a, (const bf16*)b, (bf16*)y,
                     n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_copy2d_bf16(const void* src, void* dst, int rows, int cols, int lds, int ldd, void* stream) {
  if (rows <= 0 || cols <= 0 || (cols & 7) || (lds & 7) || (ldd & 7)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(copy2d_kernel, flat_grid((long long)rows * (cols / 8)), dim3(256), 0, S_, (const bf16*)src,
                     (bf16*)dst, rows, cols / 8, lds, ldd);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_copy_rows_bf16(const void* src, void* dst, int rows, int T, int D, int src_rps, int src_off,
                                  int dst_rps, int dst_off, int accumulate, void* stream) {
  if (rows <= 0 || T <= 0 || (D & 7)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(copy_rows_kernel, flat_grid((long long)rows * (D / 8)), dim3(256), 0, S_, (const bf16*)src,
                     (bf16*)dst, rows, T, D / 8, src_rps, src_off, dst_rps, dst_off, accumulate);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_im2col_patch(const float* img, float* out, int B, int H, int W, int C, int P, void* stream) {
  if (B <= 0 || P <= 0 || H % P || W % P) return LAP_ERR_ARG;
  const long long n = (long long)B * H * W * C;
  hipLaunchKernelGGL(im2col_kernel, flat_grid(n), dim3(256), 0, S_, img, out, B, H, W, C, P);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_augment_images(const float* img, float* out, const float* params, int B, int H, int W, void* stream) {
  if (!img || !out || !params || img == out || B <= 0 || H <= 0 || W <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(augment_kernel, flat_grid((long long)B * H * W), dim3(256), 0, S_, img, out, params, B, H, W);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_add_posemb_cast(const float* x, const float* pos, void* y, int rows, int T, int D, void* stream) {
  if (rows <= 0 || T <= 0 || (D & 7)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(add_posemb_cast_kernel, flat_grid((long long)rows * (D / 8)), dim3(256), 0, S_, x, pos, (bf16*)y,
                     rows, T, D / 8);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_add_posemb_cast_bwd(const void* dy, float* dx, float* dpos, int rows, int T, int D, void* stream) {
  if (rows <= 0 || T <= 0 || rows % T || (D & 7)) return LAP_ERR_ARG;
  dim3 grid((D / 8 + 255) / 256, T);
  hipLaunchKernelGGL(add_posemb_cast_bwd_kernel, grid, dim3(256), 0, S_, (const bf16*)dy, dx, dpos, rows / T, T, D / 8);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_fm_mix(const float* noise, const float* actions, const float* t, float* x_t, float* u_t, int B,
                          int n_per, void* stream) {
  if (B <= 0 || n_per <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(fm_mix_kernel, flat_grid((long long)B * n_per), dim3(256), 0, S_, noise, actions, t, x_t, u_t, B,
                     n_per);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_posemb_sincos(const float* t, float* out, int B, int D, float min_period, float max_period,
                                 void* stream) {
  if (B <= 0 || D <= 0 || (D & 1)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(posemb_sincos_kernel, flat_grid((long long)B * (D / 2)), dim3(256), 0, S_, t, out, B, D, min_period,
                     max_period);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_swish_fwd(const float* x, float* y, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(swish_fwd_kernel, flat_grid(n), dim3(256), 0, S_, x, y, n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_swish_bwd(const float* x, const float* dy, float* dx, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(swish_bwd_kernel, flat_grid(n), dim3(256), 0, S_, x, dy, dx, n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_mse_fwd_bwd(const float* v, const float* u, const float* coef, float* per_sample, float* dv, int B,
                               int n_per, void* stream) {
  if (B <= 0 || n_per <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(mse_kernel, dim3(B), dim3(256), 0, S_, v, u, coef, per_sample, dv, n_per);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_axpy_f32(float* x, const float* v, float dt, long long n, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(axpy_kernel, flat_grid(n), dim3(256), 0, S_, x, v, dt, n);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
