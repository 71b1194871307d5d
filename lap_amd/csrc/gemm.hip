// bf16 MFMA GEMM for gfx950 with fused epilogues — the dominant kernel of the
// LAP-3B train step (K8/K11/K12/K13 of SURVEY.md §2.2: QKV / out-proj / GeGLU
// FFN / LM-head projections, their dgrad and wgrad).
//
//   C[M,N] = epilogue( alpha * sum_k opA(m,k) * opB(k,n) )
//
// Operand layouts (chosen per call, no data is ever transposed in HBM):
//   a_kc = 1 : A stored [M][K]  (k contiguous),  element (m,k) = A[m*lda + k]
//   a_kc = 0 : A stored [K][M]  (m contiguous),  element (m,k) = A[k*lda + m]
//   b_kc = 1 : B stored [N][K]  (k contiguous),  element (k,n) = B[n*ldb + k]
//   b_kc = 0 : B stored [K][N]  (n contiguous),  element (k,n) = B[k*ldb + n]
// With weights kept as Wt[out][in]:
//   forward  y = x . Wt^T        -> a_kc=1, b_kc=1
//   dgrad    dx = dy . Wt        -> a_kc=1, b_kc=0
//   wgrad    dWt = dy^T . x      -> a_kc=0, b_kc=0
//
// Structure: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave a
// 64x64 sub-tile = 4x4 v_mfma_f32_16x16x32_bf16 fragments (64 fp32 acc VGPRs).
// HBM -> LDS staging is buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round
// trip), double buffered: tile t+1 streams in while tile t is multiplied, one
// barrier per k-tile.  The LDS image is lane-linear per wave instruction, so the
// bank-conflict swizzle is applied to the per-lane *source* address and again on
// the fragment read (common.hpp).  Out-of-range rows / k-tails use an
// out-of-bounds buffer offset, for which the hardware writes zeros.
#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr unsigned OOB = 0x80000000u;

struct GemmParams {
  const bf16* A; const bf16* B;
  void* C;
  const void* bias;      // [N] or null
  const bf16* R;         // residual [M][ldr] bf16 or null
  int M, N, K;
  int lda, ldb, ldc, ldr;
  float alpha;
  int tiles_m, tiles_n;
};

// Epilogue flags (template): OUT_F32, HAS_BIAS(0 none,1 bf16,2 f32), ACT_GELU, HAS_RES, ACCUM
template <bool A_KC, bool B_KC, bool OUT_F32, int BIAS, bool GELU, bool RES, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  // buffer b: A tile at smem + 2b*TILE_BYTES, B tile right behind it.

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;

  // Block -> tile mapping: XCD-aware remap, then groups of 8 m-tiles sweep n.
  const int nblk = p.tiles_m * p.tiles_n;
  int t = xcd_remap(blockIdx.x, nblk);
  constexpr int GM = 8;
  const int group_sz = GM * p.tiles_n;
  const int gidx = t / group_sz;
  const int first_m = gidx * GM;
  const int gm = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (t % group_sz) % gm;
  const int tn = (t % group_sz) / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0,
      (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0,
      (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);

  // Per-thread staging descriptors: 4 LDS-DMA pieces per operand per k-tile.
  // K-contiguous operand: piece q = w*4+j covers tile rows 8q..8q+7 (1 KiB),
  //   lane -> (row = 8q + lane/8, physical chunk = lane%8).
  // M-contiguous operand: piece q covers k rows 4q..4q+3 (256 B each),
  //   lane -> (krow = 4q + lane/16, physical chunk = lane%16).
  unsigned offA[4], offB[4];     // byte offset at k0 = 0 (OOB if the row is out of range)
  int kidxA[4], kidxB[4];        // k index (element) this lane's chunk starts at, relative to k0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = w * 4 + j;
    if (A_KC) {
      const int row = 8 * q + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((row >> 1) & 7);
      kidxA[j] = c * 8;
      offA[j] = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + c * 8) * 2) : OOB;
    } else {
      const int kr = 4 * q + (lane >> 4), pc = lane & 15;
      const int c = pc ^ (mc_swz(kr) << 1);
      kidxA[j] = kr;
      offA[j] = (m0 + c * 8 < p.M) ? (unsigned)(((long long)kr * p.lda + m0 + c * 8) * 2) : OOB;
    }
    if (B_KC) {
      const int row = 8 * q + (lane >> 3), pc = lane & 7;
      const int c = pc ^ ((row >> 1) & 7);
      kidxB[j] = c * 8;
      offB[j] = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb + c * 8) * 2) : OOB;
    } else {
      const int kr = 4 * q + (lane >> 4), pc = lane & 15;
      const int c = pc ^ (mc_swz(kr) << 1);
      kidxB[j] = kr;
      offB[j] = (n0 + c * 8 < p.N) ? (unsigned)(((long long)kr * p.ldb + n0 + c * 8) * 2) : OOB;
    }
  }
  const unsigned stepA = A_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.lda * 2);
  const unsigned stepB = B_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.ldb * 2);

  auto stage = [&](int buf, int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = w * 4 + j;
      unsigned va = (offA[j] != OOB && k0 + kidxA[j] < p.K) ? offA[j] + (unsigned)kt * stepA : OOB;
      unsigned vb = (offB[j] != OOB && k0 + kidxB[j] < p.K) ? offB[j] + (unsigned)kt * stepB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(smem + buf * (2 * TILE_BYTES) + q * 1024), 16, va, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(smem + buf * (2 * TILE_BYTES) + TILE_BYTES + q * 1024), 16, vb, 0, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + BK - 1) / BK;
  stage(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    __syncthreads();  // tile kt landed (vmcnt(0) + barrier); buffer cur^1 free again
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    const char* tA = smem + cur * (2 * TILE_BYTES);
    const char* tB = tA + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = A_KC ? kc_frag(tA, wm * 64 + i * 16, kk, lane) : mc_frag<BM>(tA, wm * 64 + i * 16, kk, lane);
        fb[i] = B_KC ? kc_frag(tB, wn * 64 + i * 16, kk, lane) : mc_frag<BN>(tB, wn * 64 + i * 16, kk, lane);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // Operands swapped on purpose: D[row = n][col = m], so each lane ends up
          // with 4 consecutive n of one output row m -> one 8/16-byte store.
          acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    }
  }

  // Epilogue. Lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + 4*(lane>>4).
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= p.N) continue;  // N % 4 == 0 is required by the host wrapper
      f32x4 v = acc[i][j] * p.alpha;
      if (BIAS == 1) {
        bf16x4 b = *reinterpret_cast<const bf16x4*>((const bf16*)p.bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
      } else if (BIAS == 2) {
        f32x4 b = *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
        v += b;
      }
      if (GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
      }
      if (RES) {
        bf16x4 r = *reinterpret_cast<const bf16x4*>(p.R + (long long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      }
      if (OUT_F32) {
        float* c = (float*)p.C + (long long)m * p.ldc + n;
        if (ACCUM) v += *reinterpret_cast<const f32x4*>(c);
        *reinterpret_cast<f32x4*>(c) = v;
      } else {
        bf16* c = (bf16*)p.C + (long long)m * p.ldc + n;
        if (ACCUM) {
          bf16x4 o = *reinterpret_cast<const bf16x4*>(c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)o[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x4*>(c) = o;
      }
    }
  }
}

template <bool A_KC, bool B_KC, bool OUT_F32, int BIAS, bool GELU, bool RES, bool ACCUM>
int launch(const GemmParams& p, hipStream_t s) {
  dim3 grid(p.tiles_m * p.tiles_n);
  hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, OUT_F32, BIAS, GELU, RES, ACCUM>), grid, dim3(256), 0, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <bool A_KC, bool B_KC>
int dispatch_epi(const GemmParams& p, int flags, hipStream_t s) {
  const bool f32 = flags & LAP_GEMM_OUT_F32;
  const bool accum = flags & LAP_GEMM_ACCUM;
  const bool gelu = flags & LAP_GEMM_GELU;
  const bool res = p.R != nullptr;
  const int bias = p.bias ? ((flags & LAP_GEMM_BIAS_F32) ? 2 : 1) : 0;
  // Instantiate only the combinations the engine uses.
  if (f32) {
    if (gelu || res) return LAP_ERR_ARG;
    if (bias == 0) return accum ? launch<A_KC, B_KC, true, 0, false, false, true>(p, s)
                                : launch<A_KC, B_KC, true, 0, false, false, false>(p, s);
    if (bias == 2 && !accum) return launch<A_KC, B_KC, true, 2, false, false, false>(p, s);
    return LAP_ERR_ARG;
  }
  if (accum) return LAP_ERR_ARG;
  if (bias == 0) {
    if (gelu) return LAP_ERR_ARG;
    return res ? launch<A_KC, B_KC, false, 0, false, true, false>(p, s)
               : launch<A_KC, B_KC, false, 0, false, false, false>(p, s);
  }
  if (gelu) {
    if (res) return LAP_ERR_ARG;
    return bias == 1 ? launch<A_KC, B_KC, false, 1, true, false, false>(p, s)
                     : launch<A_KC, B_KC, false, 2, true, false, false>(p, s);
  }
  if (bias == 1) return res ? launch<A_KC, B_KC, false, 1, false, true, false>(p, s)
                            : launch<A_KC, B_KC, false, 1, false, false, false>(p, s);
  return res ? launch<A_KC, B_KC, false, 2, false, true, false>(p, s)
             : launch<A_KC, B_KC, false, 2, false, false, false>(p, s);
}

}  // namespace

extern "C" int lap_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual,
                             int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                             int a_kc, int b_kc, int flags, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return LAP_ERR_ARG;
  // 16-byte chunk granularity along each contiguous axis; 4-wide epilogue stores.
  if ((N & 3) || (ldc & 3) || (lda & 7) || (ldb & 7)) return LAP_ERR_ARG;
  if (a_kc ? (K & 7) : (M & 7)) return LAP_ERR_ARG;
  if (b_kc ? (K & 7) : (N & 7)) return LAP_ERR_ARG;
  if (residual && (ldr & 3)) return LAP_ERR_ARG;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return LAP_ERR_ARG;
  // 31-bit byte offsets inside one buffer descriptor.
  if ((long long)(a_kc ? M : K) * lda * 2 >= 0x7fffffffLL) return LAP_ERR_ARG;
  if ((long long)(b_kc ? N : K) * ldb * 2 >= 0x7fffffffLL) return LAP_ERR_ARG;
  GemmParams p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.bias = bias; p.R = (const bf16*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.alpha = alpha;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  hipStream_t s = (hipStream_t)stream;
  if (a_kc && b_kc) return dispatch_epi<true, true>(p, flags, s);
  if (a_kc && !b_kc) return dispatch_epi<true, false>(p, flags, s);
  if (!a_kc && !b_kc) return dispatch_epi<false, false>(p, flags, s);
  return dispatch_epi<false, true>(p, flags, s);
}
