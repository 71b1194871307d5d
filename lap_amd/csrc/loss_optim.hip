// Cross-entropy over the 257,152-entry vocabulary without one-hot targets
// (lap.py:221-260 materialises them), the fused clip + AdamW + EMA + bf16
// refresh step (scripts/train.py:363-396 with optax.chain(clip_by_global_norm,
// adamw)), and the global-norm reduction (optax.global_norm).  All HBM-bound.
#include <cstdlib>
#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

// ------------------------------------------------------------------- CE forward
// One block per row.  Online (max, sum-exp) over this vocab chunk merged into the
// running state, plus the target logit if the target falls inside the chunk.
__global__ __launch_bounds__(256) void ce_update_kernel(const float* __restrict__ logits, int ldl,
                                                        const int32_t* __restrict__ target, float* __restrict__ m,
                                                        float* __restrict__ l, float* __restrict__ tl, int v0, int vc) {
  __shared__ float red_m[4], red_l[4];
  const int r = blockIdx.x;
  const float* x = logits + (long long)r * ldl;
  float mm = -3.0e38f, ll = 0.f;
  for (int v = threadIdx.x * 4; v < vc; v += 1024) {
    float vals[4];
    int n = min(4, vc - v);
    if (n == 4 && ((((uintptr_t)(x + v)) & 15) == 0)) {
      f32x4 t = *reinterpret_cast<const f32x4*>(x + v);
      vals[0] = t[0]; vals[1] = t[1]; vals[2] = t[2]; vals[3] = t[3];
    } else {
      for (int e = 0; e < n; ++e) vals[e] = x[v + e];
    }
    for (int e = 0; e < n; ++e) {
      const float xv = vals[e];
      if (xv > mm) { ll = ll * __expf(mm - xv) + 1.0f; mm = xv; }
      else ll += __expf(xv - mm);
    }
  }
  // wave combine
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mm, o, 64), ol = __shfl_xor(ll, o, 64);
    const float nm = fmaxf(mm, om);
    ll = ll * __expf(mm - nm) + ol * __expf(om - nm);
    mm = nm;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red_m[w] = mm; red_l[w] = ll; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float gm = m[r], gl = l[r];
    for (int i = 0; i < 4; ++i) {
      const float nm = fmaxf(gm, red_m[i]);
      gl = gl * __expf(gm - nm) + red_l[i] * __expf(red_m[i] - nm);
      gm = nm;
    }
    m[r] = gm; l[r] = gl;
    const int t = target[r];
    if (t >= v0 && t < v0 + vc) tl[r] = x[t - v0];
  }
}

__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, int ldl,
                                                      const int32_t* __restrict__ target, const float* __restrict__ m,
                                                      const float* __restrict__ l, const float* __restrict__ wgt,
                                                      bf16* __restrict__ dlogits, bf16* __restrict__ dlo, int ldd, int v0, int vc) {
  const int r = blockIdx.y;
  const int v = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (v >= vc) return;
  const float lse = m[r] + __logf(l[r]);
  const float wr = wgt[r];
  const int t = target[r] - v0;
  const float* x = logits + (long long)r * ldl + v;
  bf16* d = dlogits + (long long)r * ldd + v;
  const int n = min(4, vc - v);
  for (int e = 0; e < n; ++e) {
    float p = (wr != 0.f) ? __expf(x[e] - lse) : 0.f;
    if (v + e == t) p -= 1.0f;
    const float dv = wr * p;
    d[e] = f2bf(dv);
    // the f32 cotangent as two bf16 planes (hi + lo carries 16 mantissa bits): gemma.py:153-154 keeps the logits — and with them
    // their gradient — in f32, the GEMMs that consume it here are bf16 MFMA products with f32 accumulation
    if (dlo) dlo[(long long)r * ldd + v + e] = f2bf(dv - round_bf16(dv));
  }
}

// Greedy token choice: index of the row maximum, lowest index among ties (jnp.argmax, lap.py:723).  One block per row.
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int n, int* __restrict__ out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const float* row = x + (long long)blockIdx.x * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = threadIdx.x; c < n; c += 256) {
    const float v = row[c];
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sv[w] = best; si[w] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
    out[blockIdx.x] = bi;
  }
}

// -------------------------------------------------------------------- optimizer
// Four independent 16-byte loads in flight per thread: a block sweeps contiguous 16 KiB chunks (4 x 4 KiB, thread t takes
// bytes [16 t, 16 t + 16) of each KiB-quad), so the stream stays page-local; the first version had one load in flight and
// ran at 1.5 TB/s beside the backward GEMMs it shares the chip with (profiles/r01_bench_n1_kernel_stats.md).
template <bool NT>
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const long long chunk = 4096;                                  // elements per block and iteration
  long long base = (long long)blockIdx.x * chunk;
  const long long step = (long long)gridDim.x * chunk;
  const int t4 = threadIdx.x * 4;
  for (; base + chunk <= n; base += step) {
    const float* p = x + base + t4;
    auto ld = [](const float* q) { return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)) : *reinterpret_cast<const f32x4*>(q); };
    const f32x4 t0 = ld(p), t1 = ld(p + 1024);
    const f32x4 t2 = ld(p + 2048), t3 = ld(p + 3072);
    a0 += t0[0] * t0[0] + t0[1] * t0[1] + t0[2] * t0[2] + t0[3] * t0[3];
    a1 += t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2] + t1[3] * t1[3];
    a2 += t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2] + t2[3] * t2[3];
    a3 += t3[0] * t3[0] + t3[1] * t3[1] + t3[2] * t3[2] + t3[3] * t3[3];
  }
  if (base < n) {                                                // ragged tail of the last sweep (at most one block has one)
    for (long long j = base + threadIdx.x; j < n && j < base + chunk; j += 256) a0 += x[j] * x[j];
  }
  const float acc = block_sum<4>((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

// The same sum over a BF16 gradient buffer (round 5: the GEMM-produced weight gradients are stored as bf16, what the reference's
// cast boundary hands its f32 master): 8 values per 16-byte load, squares accumulated in f32.
__global__ __launch_bounds__(256) void sumsq_bf16_kernel(const bf16* __restrict__ x, long long n, float* __restrict__ out) {
  __shared__ float red[4];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const long long chunk = 8192;                                  // elements per block and iteration (16 KiB)
  long long base = (long long)blockIdx.x * chunk;
  const long long step = (long long)gridDim.x * chunk;
  const int t8 = threadIdx.x * 8;
  auto sq8 = [](bf16x8 t) { float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = (float)t[e]; s += f * f; }
    return s; };
  for (; base + chunk <= n; base += step) {
    const bf16* p = x + base + t8;
    const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(p), t1 = *reinterpret_cast<const bf16x8*>(p + 2048);
    const bf16x8 t2 = *reinterpret_cast<const bf16x8*>(p + 4096), t3 = *reinterpret_cast<const bf16x8*>(p + 6144);
    a0 += sq8(t0); a1 += sq8(t1); a2 += sq8(t2); a3 += sq8(t3);
  }
  if (base < n) {
    for (long long j = base + threadIdx.x; j < n && j < base + chunk; j += 256) { const float f = (float)x[j]; a0 += f * f; }
  }
  const float acc = block_sum<4>((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

// Two parameters per thread and iteration, 32 VGPRs: an optimizer wave fits next to the 256x256 GEMM's 4 waves x 120
// VGPRs per SIMD, so the side-stream optimizer never evicts GEMM blocks of the next forward.  (Measured: the step time
// does not change versus a 54-VGPR version - the ~26 ms the overlapped optimizer still costs per step is HBM contention
// of its 38 B / parameter with the GEMMs' operand traffic, not CU occupancy.)
template <bool NT, bool G16>      // G16: the gradient buffer holds bf16 (read as 2 x bf16 = 4 bytes per pair of parameters)
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float* __restrict__ ema,
                      const void* __restrict__ g, bf16* __restrict__ p16, bf16* __restrict__ p16lo, long long n, const float* __restrict__ sc,
                      float b1, float b2, float eps, float wd, float max_norm) {
  const float gnorm = sqrtf(sc[0]);
  // optax.clip_by_global_norm: g if norm < max_norm else g / norm * max_norm
  const float clip = (max_norm <= 0.f || gnorm < max_norm) ? 1.0f : max_norm / gnorm;
  const float lr = sc[1], rbc1 = 1.0f / sc[2], rbc2 = 1.0f / sc[3], ed = sc[4];
  const bool ema_on = ema != nullptr && sc[5] != 0.f;
  // 32-bit BYTE offsets from the (scalar) array bases: one VGPR of addressing for all six arrays (the host wrapper
  // launches at most 2^29 elements at a time); v_sqrt / v_rcp (1 ulp) instead of the IEEE fix-up sequences
  const unsigned nb = (unsigned)n * 4u;
  const unsigned stride = gridDim.x * blockDim.x * 8u;
  auto at = [](auto* base, unsigned byte_off) { return reinterpret_cast<f32x2*>(reinterpret_cast<char*>(base) + byte_off); };
  auto cat = [](const float* base, unsigned byte_off) { return reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(base) + byte_off); };
  for (unsigned o = (blockIdx.x * blockDim.x + threadIdx.x) * 8u; o < nb; o += stride) {
    {   // n is even (unit buffers are padded to multiples of 64 elements; checked by the host wrapper)
      // NT: the master values, both moments, the EMA and the gradient are touched once per step — nontemporal, so that the
      // 36 B / parameter streaming past do not push the GEMMs' operand panels out of the L2 (the bf16 copy is stored normally:
      // the next forward reads it soon)
      auto ldf = [](const f32x2* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
      auto stf = [](f32x2* q, f32x2 val) { if (NT) __builtin_nontemporal_store(val, q); else *q = val; };
      f32x2 pv = ldf(cat(p, o)), mv = ldf(cat(m, o)), vv = ldf(cat(v, o));
      f32x2 gv;
      if constexpr (G16) {
        const bf16x2 gq = *reinterpret_cast<const bf16x2*>(reinterpret_cast<const char*>(g) + (o >> 1));
        gv[0] = (float)gq[0]; gv[1] = (float)gq[1];
      } else {
        gv = ldf(cat(reinterpret_cast<const float*>(g), o));
      }
      f32x2 ev = {0.f, 0.f};
      if (ema_on) ev = ldf(cat(ema, o));
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float gg = gv[e] * clip;
        asm("" : "+v"(gg));   // keeps the two lanes of work scalar: packed f32 math would need the constants in VGPR pairs
        mv[e] = b1 * mv[e] + (1.0f - b1) * gg;
        vv[e] = b2 * vv[e] + (1.0f - b2) * gg * gg;
        const float upd = (mv[e] * rbc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[e] * rbc2) + eps) + wd * pv[e];
        pv[e] = pv[e] - lr * upd;
        if (ema_on) ev[e] = ed * ev[e] + (1.0f - ed) * pv[e];
      }
      stf(at(p, o), pv);
      stf(at(m, o), mv);
      stf(at(v, o), vv);
      if (ema_on) stf(at(ema, o), ev);
      if (p16) { bf16x2 q; q[0] = f2bf(pv[0]); q[1] = f2bf(pv[1]); *reinterpret_cast<bf16x2*>(reinterpret_cast<char*>(p16) + (o >> 1)) = q; }
      if (p16lo) {   // the residual plane of the embedding table (hi + lo = 16 mantissa bits of the f32 value: the LM head's f32 operand)
        bf16x2 q; q[0] = f2bf(pv[0] - round_bf16(pv[0])); q[1] = f2bf(pv[1] - round_bf16(pv[1]));
        *reinterpret_cast<bf16x2*>(reinterpret_cast<char*>(p16lo) + (o >> 1)) = q;
      }
    }
  }
}

// ------------------------------------------------------------------ f32 GEMM
// 64x64 tile, BK = 16, 256 threads x (4x4) outputs, k-ordered fmaf chain (exact f32).
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, const float* __restrict__ bias, int M,
                                                       int N, int K, int lda, int ldb, int ldc, float alpha, int accum) {
  __shared__ float As[16][68], Bs[16][68];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = tid & 15, ty = tid >> 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    // 64x16 elements per operand = 1024 -> 4 per thread
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + 256 * j;
      int mm, kk;
      if (A_KC) { mm = idx >> 4; kk = idx & 15; } else { kk = idx >> 6; mm = idx & 63; }
      float va = 0.f;
      if (m0 + mm < M && k0 + kk < K)
        va = A_KC ? A[(long long)(m0 + mm) * lda + k0 + kk] : A[(long long)(k0 + kk) * lda + m0 + mm];
      As[kk][mm] = va;
      int nn, kb;
      if (B_KC) { nn = idx >> 4; kb = idx & 15; } else { kb = idx >> 6; nn = idx & 63; }
      float vb = 0.f;
      if (n0 + nn < N && k0 + kb < K)
        vb = B_KC ? B[(long long)(n0 + nn) * ldb + k0 + kb] : B[(long long)(k0 + kb) * ldb + n0 + nn];
      Bs[kb][nn] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = As[kk][ty * 4 + e]; b[e] = Bs[kk][tx * 4 + e]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mrow = m0 + ty * 4 + i;
    if (mrow >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ncol = n0 + tx * 4 + j;
      if (ncol >= N) continue;
      float vv = acc[i][j] * alpha;
      if (bias) vv += bias[ncol];
      float* c = C + (long long)mrow * ldc + ncol;
      if (accum) vv += *c;
      *c = vv;
    }
  }
}

// Skinny f32 GEMM (M <= 64 rows, both operands k-contiguous): one wave per output column n and group of 8 rows;
// the weight row streams once with coalesced 16-byte loads, the k-sum order is fixed (lane-strided, then a
// butterfly), so results are deterministic.  Serves the batch-1 time-MLP / action head (lap.py:52-62,298).
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                       float* __restrict__ C, const float* __restrict__ bias, int M, int N,
                                                       int K, int lda, int ldb, int ldc, float alpha, int accum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + w;
  const int m0 = blockIdx.y * 8;
  if (n >= N) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* brow = B + (long long)n * ldb;
  for (int k = lane; k < K; k += 64) {
    const float bv = brow[k];
#pragma unroll
    for (int r = 0; r < 8; ++r)
      if (m0 + r < M) acc[r] = fmaf(A[(long long)(m0 + r) * lda + k], bv, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = wave_sum(acc[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (m0 + r >= M) break;
      float v = acc[r] * alpha + (bias ? bias[n] : 0.f);
      float* c = C + (long long)(m0 + r) * ldc + n;
      if (accum) v += *c;
      *c = v;
    }
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int lap_ce_chunk_update(const float* logits, int ldl, const int32_t* target, float* m, float* l, float* tl,
                                   int rows, int v0, int vc, void* stream) {
  if (rows <= 0 || vc <= 0) return LAP_ERR_ARG;
  hipLaunchKernelGGL(ce_update_kernel, dim3(rows), dim3(256), 0, S_, logits, ldl, target, m, l, tl, v0, vc);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_ce_chunk_grad(const float* logits, int ldl, const int32_t* target, const float* m, const float* l,
                                 const float* w, void* dlogits, int ldd, int rows, int v0, int vc, void* stream) {
  if (rows <= 0 || vc <= 0) return LAP_ERR_ARG;
  dim3 grid((vc + 1023) / 1024, rows);
  hipLaunchKernelGGL(ce_grad_kernel, grid, dim3(256), 0, S_, logits, ldl, target, m, l, w, (bf16*)dlogits, (bf16*)nullptr, ldd, v0, vc);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_ce_chunk_grad_hilo(const float* logits, int ldl, const int32_t* target, const float* m, const float* l,
                                      const float* w, void* dlogits_hi, void* dlogits_lo, int ldd, int rows, int v0, int vc, void* stream) {
  if (rows <= 0 || vc <= 0 || !dlogits_hi || !dlogits_lo) return LAP_ERR_ARG;
  dim3 grid((vc + 1023) / 1024, rows);
  hipLaunchKernelGGL(ce_grad_kernel, grid, dim3(256), 0, S_, logits, ldl, target, m, l, w, (bf16*)dlogits_hi, (bf16*)dlogits_lo, ldd, v0, vc);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_argmax_rows_f32(const float* x, int rows, int n, int ld, int* out, void* stream) {
  if (rows <= 0 || n <= 0 || ld < n || !x || !out) return LAP_ERR_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, S_, x, ld, n, out);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_sumsq_f32(const float* x, long long n, float* sumsq, void* stream) {
  if (n <= 0) return LAP_ERR_ARG;
  const long long blocks = (n + 4095) / 4096;
  static const bool nt = getenv("LAP_SUMSQ_NT") ? atoi(getenv("LAP_SUMSQ_NT")) != 0 : false;
  static const long long cap = getenv("LAP_SUMSQ_BLOCKS") ? atoll(getenv("LAP_SUMSQ_BLOCKS")) : 2048;     // tuning knob (see adamw_launch)
  if (nt) hipLaunchKernelGGL(sumsq_kernel<true>, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, S_, x, n, sumsq);
  else hipLaunchKernelGGL(sumsq_kernel<false>, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, S_, x, n, sumsq);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
static int adamw_launch(float* p, float* m, float* v, float* ema, const void* g, void* p16, void* p16lo, long long n,
                        const float* scalars, float b1, float b2, float eps, float wd, float max_norm, void* stream, bool g16 = false) {
  if (n <= 0 || (n & 1) || !scalars) return LAP_ERR_ARG;
  // At most ONE optimizer block per CU (round 5, profiles/r05_optimizer_throttle.txt, r05_optimizer_coresidency_probe.txt): one 256-thread
  // block = one 32-register wave per SIMD fits beside the assembly GEMM's 480-register waves, a second one does not (the pass then waits
  // for GEMM launches to end and delays the next one's blocks), and thousands of short blocks get in the way of the forward's short launches.
  // Step, same box, interleaved: 4096 blocks 266.7 | 384 268.4 | 320 263.5 | 288 262.1 | 256 259.3 | 224 259.2 | 192 260.2 | 160 262.2 |
  // 128 270.6 | 64 291.7 ms.  Default: 15/16 of the device's CUs (240 on MI355X; a partitioned device gets its own count).
  // LAP_ADAMW_BLOCKS / LAP_ADAMW_THREADS: tuning knobs.
  // (the cap is kept per device: a process that drives several devices must not hand the first one's CU count to the others; the knobs are
  //  clamped to what the kernel's __launch_bounds__(256) and a non-empty grid allow)
  static long long cap_of[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  long long cap = __atomic_load_n(&cap_of[dev], __ATOMIC_RELAXED);
  if (cap == 0) {
    if (getenv("LAP_ADAMW_BLOCKS")) cap = atoll(getenv("LAP_ADAMW_BLOCKS"));
    else {
      int cus = 256;
      if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
      cap = cus * 15 / 16;
    }
    if (cap < 1) cap = 1;
    __atomic_store_n(&cap_of[dev], cap, __ATOMIC_RELAXED);
  }
  static const int threads = [] {
    int t = getenv("LAP_ADAMW_THREADS") ? atoi(getenv("LAP_ADAMW_THREADS")) : 256;
    t = t < 64 ? 64 : t > 256 ? 256 : t;
    return t & ~63;
  }();
  const long long CH = 1LL << 29;
  for (long long o = 0; o < n; o += CH) {
    const long long cnt = n - o < CH ? n - o : CH;
    const long long blocks = (cnt + 2 * threads - 1) / (2 * threads);
    static const bool nt = getenv("LAP_ADAMW_NT") ? atoi(getenv("LAP_ADAMW_NT")) != 0 : true;   // (-1.8 ms per train step: tools/ab3.sh)
    const void* go = g16 ? (const void*)((const bf16*)g + o) : (const void*)((const float*)g + o);
#define ADAMW_GO(NT_, G16_) hipLaunchKernelGGL((adamw_ema_kernel<NT_, G16_>), dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(threads), 0, S_, p + o, m + o, \
                         v + o, ema ? ema + o : nullptr, go, p16 ? (bf16*)p16 + o : nullptr, p16lo ? (bf16*)p16lo + o : nullptr, cnt, scalars, b1, b2,     \
                         eps, wd, max_norm)
    if (nt) { if (g16) ADAMW_GO(true, true); else ADAMW_GO(true, false); }
    else { if (g16) ADAMW_GO(false, true); else ADAMW_GO(false, false); }
#undef ADAMW_GO
    LAP_CHECK_LAUNCH();
  }
  return LAP_OK;
}
extern "C" int lap_adamw_ema(float* p, float* m, float* v, float* ema, const float* g, void* p16, long long n,
                             const float* scalars, float b1, float b2, float eps, float wd, float max_norm,
                             void* stream) {
  return adamw_launch(p, m, v, ema, g, p16, nullptr, n, scalars, b1, b2, eps, wd, max_norm, stream);
}
extern "C" int lap_adamw_ema_g16(float* p, float* m, float* v, float* ema, const void* g16, void* p16, long long n,
                                 const float* scalars, float b1, float b2, float eps, float wd, float max_norm,
                                 void* stream) {
  if (!g16 || ((uintptr_t)g16 & 3)) return LAP_ERR_ARG;
  return adamw_launch(p, m, v, ema, g16, p16, nullptr, n, scalars, b1, b2, eps, wd, max_norm, stream, true);
}
extern "C" int lap_sumsq_bf16(const void* x, long long n, float* sumsq, void* stream) {
  if (n <= 0 || !x || !sumsq || ((uintptr_t)x & 15)) return LAP_ERR_ARG;
  const long long blocks = (n + 8191) / 8192;
  static const long long cap = getenv("LAP_SUMSQ_BLOCKS") ? atoll(getenv("LAP_SUMSQ_BLOCKS")) : 2048;
  hipLaunchKernelGGL(sumsq_bf16_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, S_, (const bf16*)x, n, sumsq);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
extern "C" int lap_adamw_ema_hilo(float* p, float* m, float* v, float* ema, const float* g, void* p16, void* p16lo, long long n,
                                  const float* scalars, float b1, float b2, float eps, float wd, float max_norm,
                                  void* stream) {
  if (!p16 || !p16lo) return LAP_ERR_ARG;
  return adamw_launch(p, m, v, ema, g, p16, p16lo, n, scalars, b1, b2, eps, wd, max_norm, stream);
}
extern "C" int lap_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                            int ldb, int ldc, float alpha, int a_kc, int b_kc, int accum, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return LAP_ERR_ARG;
  if (a_kc && b_kc && M <= 64) {
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4, (M + 7) / 8), dim3(256), 0, S_, A, B, C, bias, M, N, K, lda, ldb, ldc,
                       alpha, accum);
    LAP_CHECK_LAUNCH();
    return LAP_OK;
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
#define GO(AK, BK) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK>), grid, dim3(256), 0, S_, A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, accum)
  if (a_kc && b_kc) GO(true, true);
  else if (a_kc && !b_kc) GO(true, false);
  else if (!a_kc && !b_kc) GO(false, false);
  else GO(false, true);
#undef GO
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
