// Row-panel GEMM for the serving prefill (lap.py:605-632 -> siglip_gemma3.py:86-110, gemma.py:294-321 at M = 512 .. 640 rows):
//   C[M][N] = epi( norm(A)[M][K] . W[N][K]^T )
// The prefill's projections are a few GFLOP each over a dependent chain of ~340 launches; on the LDS-tiled kernels (gemm.hip) a
// 64-deep k-step of a small tile costs ~900 cycles around 128 cycles of matrix pipe whatever the tile (docs/EXPERIMENTS.md I:
// barrier -> fragment reads -> wait -> MFMAs -> DMA issue is ONE dependent chain per wave, and grids of 72 - 272 blocks give a
// SIMD nothing else to run).  This kernel has no barrier and no LDS write inside its k-loop:
//   * a block owns BM = 64 (K <= 1216) or 32 (K <= 2048) complete rows of A: the panel [BM][K] goes into LDS ONCE (by LDS-DMA,
//     as K/64 swizzled k-tiles: fragment reads cover the 64 banks once), optionally through the LayerNorm / RMSNorm the
//     reference applies in front of the projection (the block has the whole row; bit-identical to the norm kernels, but every
//     column block of a panel normalises the same rows again: +10 us per launch against 3.4 us for the separate norm launch, so
//     the model does not use it: docs/EXPERIMENTS.md K);
//   * each of the 4 waves owns NT feature tiles of 16 weight rows and streams them straight from the fragment-packed image
//     (lap_serve_pack_weight kind 3: 1 KiB per wave instruction, the 130 GB/s-per-CU pattern of tools/probes/cu_pull.hip) into
//     a register ring PF k-steps deep;
//   * per 32-deep k-step a wave issues BM/16 ds_read_b128 + NT buffer loads + BM/16 x NT MFMAs: at BM = 64, NT = 2 the LDS
//     port (128 B/clk), the vector-memory path (64 B/clk) and the matrix pipe are all asked for 128 cycles per k-step and CU.
// Blocks that share weight columns (the M / BM row panels of one column job) are placed on ONE XCD (block id % 8), so HBM sees
// each weight byte once and the other panels read it from that XCD's L2.
// Same MFMA, same k slots, same k order as gemm.hip's unsplit tiles and the same epilogue arithmetic: bitwise equal to
// lap_gemm_bf16_ex for ksplit = 1 (tests/test_serve_panel_gpu.py); the LayerNorm prologue is lap_layernorm_fwd's arithmetic lane
// for lane (one wave per row, the same chunk-to-lane map, the same reduction order).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

constexpr int PN_WAVES = 4;

struct PanelP {
  const bf16* A; int lda;
  const bf16* Wp;          // packed [N/16][Kfull/32][64 lanes][8]
  int M, N, K, Kfull;      // K: contraction length of one split (Kfull = ksplit * K)
  const float* gamma;      // norm 1: RMSNorm scale (h = x rstd (1 + gamma)); norm 2: LayerNorm gamma
  const float* beta;
  float eps;
  int norm;
  const float* bias;       // f32 [N] or null
  const bf16* R; int ldr;  // residual or null
  bf16* C; int ldc;
  float* part;             // ksplit > 1: f32 [ksplit][M][N]
  int gelu;                // 0 / 1 / 2 (2: pre-activation rounded to bf16 first), as GemmParams; 3: 2 through gelu_exp2_f
  int dbg;                 // LAP_PANEL_DBG timing ablations (results wrong): 1 no panel DMA, 2 no k-loop, 4 no norm pass
  const char* pf_ptr;      // the NEXT launch's weights (or null): read and discarded by a fifth wave per block, see prefetch_wave
  long long pf_bytes;
  int ncol, nks, npanel;   // column jobs (blocks of 4 NT feature tiles), K splits, row panels
};

__device__ __forceinline__ void load8f(const bf16x8 t, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}

// wave_sum (common.hpp: v += shfl_xor(v, 32), 16, 8, 4, 2, 1) without its six ds_bpermute round trips: gfx950's permlane swaps for
// the 32 / 16 exchanges, DPP row rotates and quad permutes below.  Bit-identical: a + b is commutative, and once the sums are
// symmetric under xor 8 a rotate by 4 inside the row of 16 hands every lane the value its xor-4 partner holds.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
  v = dpp_add<0x128>(v);     // row_ror:8
  v = dpp_add<0x124>(v);     // row_ror:4
  v = dpp_add<0x4E>(v);      // quad_perm [2, 3, 0, 1]
  v = dpp_add<0xB1>(v);      // quad_perm [1, 0, 3, 2]
  return v;
}

// (a __device__ function: called straight from the __global__ template, the builtin makes the HOST pass drop the kernel's stub silently)
__device__ __forceinline__ void dma_piece(const __amdgpu_buffer_rsrc_t rs, char* lds, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))lds, 16, voff, soff, 0, 0);
}

// The prefill walks ~5 GB of weights once per chunk through ~340 dependent launches, and every launch meets its weights HBM-cold
// with a few tens of KB in flight per CU: the same launches run 15 - 40 % faster on weights that wait in the memory-side Infinity
// Cache (tools/probes/panel_bench.py, 64 MB of rotating weights, against the in-situ profile).  A second stream that reads one layer
// ahead costs more than it brings inside a replayed graph (10 - 60 us per cross-queue edge: profiles/r06_serve_prefetch_stream.txt),
// so the launch itself does it: a FIFTH wave per block reads this block's share of the NEXT launch's weights and throws it away
// (1 KiB per instruction, nothing kept, the wave ends with its loads in flight — it never meets the block's barriers).
__device__ __forceinline__ void prefetch_wave(const char* base, long long bytes, int lane, char* lds_dummy) {
  // LDS-DMA into a 1 KiB scratch slot behind the panel: no destination registers, so nothing the compiler could reuse while a
  // load is still in flight
  const long long per = ((bytes + gridDim.x - 1) / gridDim.x + 1023) & ~1023LL;
  const long long lo = (long long)blockIdx.x * per;
  if (lo >= bytes) return;
  const unsigned span = (unsigned)min(bytes - lo, per);
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + lo), 0, span, 0x00020000);
  for (unsigned o = 0; o < span; o += 1024) dma_piece(rs, lds_dummy, o + lane * 16, 0);   // (past the end: out of range, nothing is read)
}

// BM rows per block, NT feature tiles per wave, NCH 16-byte chunks per lane and row (ceil(K / 512)), NORM 0 / 1 (RMS) / 2 (LN).
// LDS image of the panel: K/64 k-tiles of [BM rows][128 B], gemm.hip's K-contiguous tile (16-byte chunk c of row r at chunk
// c ^ ((r >> 1) & 7): fragment reads cover the 64 banks once), tile pitch BM * 128 + 128 bytes so that the norm prologue's row
// sweeps (lane l <-> chunk l of the row: 8 chunks per tile) are conflict free as well.
template <int BM, int NT, int NCH, int NORM>
__global__ __launch_bounds__((PN_WAVES + 1) * 64) void panel_gemm_kernel(PanelP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int FM = BM / 16, RPW = BM / PN_WAVES, PF = 8, RG = BM / 8, TILEP = BM * 128 + 128;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform for the compiler: LDS-DMA bases)
  const int li = lane & 15, lg = lane >> 4;

  if (w == PN_WAVES) {       // the prefetch wave (launched only when there is something to prefetch)
    prefetch_wave(p.pf_ptr, p.pf_bytes, lane, smem + (p.K >> 6) * TILEP);
    return;
  }
  // block id -> (column job, split, row panel): the panels of one (column job, split) share an XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int job = (slot / p.npanel) * 8 + xcd, pm = slot % p.npanel;
  if (job >= p.ncol * p.nks) return;
  const int cn = job / p.nks, ks = job % p.nks;
  const int m0 = pm * BM;
  const int KS = p.K >> 5, KT = p.K >> 6;          // 32-deep k-steps, 64-deep k-tiles of this block

  // ---- the weight stream of this wave: NT feature tiles, k-steps [ks KS, ks KS + KS)
  const int t0 = (cn * PN_WAVES + w) * NT;         // first feature tile of the wave
  const int ntile = p.N >> 4;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)((long long)p.N * p.Kfull * 2), 0x00020000);
  unsigned woff[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = min(t0 + j, ntile - 1);          // (a tile past the edge streams the last one: never stored)
    woff[j] = (unsigned)((((long long)t * (p.Kfull >> 5) + (long long)ks * KS) * 64 + lane) * 16);
  }
  bf16x8 wb[PF][NT];
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      wb[s][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff[j] + (unsigned)s * 1024u, 0, 0));

  // ---- the A panel by LDS-DMA: piece (kt, rg) = rows 8 rg .. 8 rg + 7 of k-tile kt = 1 KiB; lane L brings row 8 rg + (L >> 3),
  // the chunk that belongs in slot L & 7.  The pieces of the block are dealt round-robin to the waves.
  {
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (unsigned)(((long long)(p.M - 1) * p.lda + p.Kfull) * 2), 0x00020000);
    unsigned aoff[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int row = rg * 8 + (lane >> 3), m = m0 + row;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      aoff[rg] = m < p.M ? (unsigned)(((long long)m * p.lda + (long long)ks * p.K) * 2 + chunk * 16) : 0x80000000u;
    }
    if (!(p.dbg & 1))
    for (int kt = w; kt < KT; kt += PN_WAVES) {
      char* tile = smem + kt * TILEP;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        dma_piece(rsA, tile + rg * 1024, aoff[rg], (unsigned)kt * 128u);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (NORM != 0 && !(p.dbg & 4)) {
    // Rows w, w + 4, ...: one wave per row, lane l holds the chunks l, l + 64, ... (lap_layernorm_fwd's map and summation order).
    // Four rows at a time and no branch around the loads, so that their LDS reads and reductions overlap (one wave per SIMD:
    // nothing else hides them); lanes past the end of the row read the row's last chunk and count as zeros.
    constexpr int RB = 4;
    const int nchunk = p.K >> 3;
    float gm[NCH][8], bt[NCH][8];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = min(lane + 64 * q, nchunk - 1) * 8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { gm[q][e] = g0[e]; gm[q][4 + e] = g1[e]; }
      if constexpr (NORM == 2) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c), b1 = *reinterpret_cast<const f32x4*>(p.beta + c + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bt[q][e] = b0[e]; bt[q][4 + e] = b1[e]; }
      }
    }
    bool valid[NCH];
    int coff[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int c = lane + 64 * q;
      valid[q] = c < nchunk;
      const int cc = min(c, nchunk - 1);
      coff[q] = (cc >> 3) * TILEP + ((cc & 7) << 4);     // (the swizzle is applied per row below)
    }
    for (int r0 = 0; r0 < RPW; r0 += RB) {
      bf16x8 raw[RB][NCH];
      char* rowp[RB];
      int sw[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int rl = (r0 + k) * PN_WAVES + w;
        rowp[k] = smem + rl * 128;
        sw[k] = ((rl >> 1) & 7) << 4;
#pragma unroll
        for (int q = 0; q < NCH; ++q) raw[k][q] = *reinterpret_cast<const bf16x8*>(rowp[k] + (coff[q] ^ sw[k]));
      }
      float sm[RB], ss[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        sm[k] = 0.f; ss[k] = 0.f;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = valid[q] ? (float)raw[k][q][e] : 0.f;
            sm[k] += x; ss[k] += x * x;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        ss[k] = wave_sum_dpp(ss[k]);
        if constexpr (NORM == 2) sm[k] = wave_sum_dpp(sm[k]);
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        float mean = 0.f, rr;
        if constexpr (NORM == 2) {
          mean = sm[k] / (float)p.K;
          const float var = fmaxf(ss[k] / (float)p.K - mean * mean, 0.f);   // Flax use_fast_variance (norm.hip layernorm_fwd_kernel)
          rr = 1.0f / sqrtf(var + p.eps);
        } else {
          rr = 1.0f / sqrtf(ss[k] / (float)p.K + p.eps);                    // gemma.py:113-131 (norm.hip rmsnorm_fwd_kernel, plain form)
        }
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (float)raw[k][q][e];
            if constexpr (NORM == 2) o[e] = f2bf((x - mean) * (rr * gm[q][e]) + bt[q][e]);
            else o[e] = f2bf(x * rr * (1.0f + gm[q][e]));
          }
          if (valid[q]) *reinterpret_cast<bf16x8*>(rowp[k] + (coff[q] ^ sw[k])) = o;
        }
      }
    }
    __syncthreads();
  }

  // ---- main loop: no barrier, no LDS write
  f32x4 acc[FM][NT];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int swl = (li >> 1) & 7;                   // (rows 16 i + li: the swizzle does not depend on i)
  const char* abase0 = smem + li * 128 + ((lg ^ swl) << 4);          // k-step 0 of a tile
  const char* abase1 = smem + li * 128 + (((4 + lg) ^ swl) << 4);    // k-step 1
  if (!(p.dbg & 2)) {
    // Groups of PF k-steps without a branch inside (the compiler then counts the outstanding loads exactly: the wait in front of
    // step s leaves the (PF - 1) NT newer weight loads in flight; one conditional load in the body and every step waits for
    // vmcnt(0), i.e. for the load issued one step earlier).  Refills past the end of the slice fetch bytes nobody uses (another
    // tile's, or zeros past the end of the buffer); the fragments of step s + 1 are read while step s multiplies.
    bf16x8 fa[2][FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(abase0 + i * 2048);
    const int nfull = KS / PF, rem = KS - nfull * PF;
    for (int g = 0; g < nfull; ++g) {
      const int s0 = g * PF;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int s = s0 + u;
        const char* ab = ((u & 1) ? abase0 : abase1) + min((s + 1) >> 1, KT - 1) * TILEP;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[(u + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048);
        __builtin_amdgcn_sched_barrier(0);   // (or the reads of step s + 1 are sunk behind the MFMAs of step s into the same registers)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(wb[u][j], fa[u & 1][i], acc[i][j]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          wb[u][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, woff[j] + (unsigned)(s + PF) * 1024u, 0, 0));
        __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler sinks all PF refills to the end of the group: the ring runs empty once per group)
      }
    }
    const int s0 = nfull * PF;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
      if (u < rem) {
        const int s = s0 + u;
        const char* ab = ((u & 1) ? abase0 : abase1) + min((s + 1) >> 1, KT - 1) * TILEP;
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[(u + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(ab + i * 2048);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(wb[u][j], fa[u & 1][i], acc[i][j]);
      }
    }
  }

  // ---- epilogue (store_tile4's arithmetic, gemm_common.hpp): lane (li, lg) holds C[m = 16 i + li][n = 16 t + 4 lg .. + 3]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = t0 + j;
    if (t >= ntile) continue;
    const int n = t * 16 + 4 * lg;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && !p.part) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int m = m0 + i * 16 + li;
      if (m >= p.M) continue;
      if (p.part) {
        *reinterpret_cast<f32x4*>(p.part + ((long long)ks * p.M + m) * p.N + n) = acc[i][j];
        continue;
      }
      f32x4 v = acc[i][j] * 1.0f;
      if (p.bias) v += bv;
      if (p.gelu == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_exp2_f(round_bf16(v[e]));
      } else if (p.gelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(p.gelu == 2 ? round_bf16(v[e]) : v[e]);
      }
      if (p.R) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(p.R + (long long)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
      }
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
      *reinterpret_cast<bf16x4*>(p.C + (long long)m * p.ldc + n) = o;
    }
  }
}

template <int BM, int NT, int NCH, int NORM>
int launch_panel(PanelP& p, hipStream_t s) {
  auto kern = panel_gemm_kernel<BM, NT, NCH, NORM>;
  const int lds = (p.K >> 6) * (BM * 128 + 128) + 1024;   // (+ the prefetch wave's scratch slot)
  static bool once = false;     // per instantiation
  if (!once) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    once = true;
  }
  p.npanel = (p.M + BM - 1) / BM;
  p.ncol = ((p.N >> 4) + PN_WAVES * NT - 1) / (PN_WAVES * NT);
  const int jobs = p.ncol * p.nks;
  const int grid = ((jobs + 7) / 8) * 8 * p.npanel;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((PN_WAVES + (p.pf_ptr ? 1 : 0)) * 64), lds, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <int BM, int NCH>
int launch_panel_nt(PanelP& p, int nt, hipStream_t s) {
#define GO_(NT_) (p.norm == 2 ? launch_panel<BM, NT_, NCH, 2>(p, s) : p.norm == 1 ? launch_panel<BM, NT_, NCH, 1>(p, s) : launch_panel<BM, NT_, NCH, 0>(p, s))
  switch (nt) {
    case 1: return GO_(1);
    case 2: return GO_(2);
    case 3: return GO_(3);
    case 4: return GO_(4);
    default: return LAP_ERR_ARG;
  }
#undef GO_
}

}  // namespace

extern "C" int lap_panel_gemm_ok(int M, int N, int K, int ksplit) {
  if (M <= 0 || N <= 0 || K <= 0 || ksplit < 1 || (N & 15) || K % (32 * ksplit)) return 0;
  const int k = K / ksplit;
  return k <= 2048 && (k & 63) == 0;     // the panel [64][k <= 1216] or [32][k <= 2048] bf16 (+ 128 B per k-tile) fits the CU's 160 KiB
}

// nt = feature tiles per wave (0: chosen here: the largest of 2 .. 4 column widths whose grid still fills one round of 256 CUs best)
extern "C" int lap_panel_gemm_pf(const void* A, int lda, const void* Wp, void* C, int ldc, const float* bias, const void* residual, int ldr,
                                 int norm, const float* gamma, const float* beta, float eps, int M, int N, int K, int flags, int ksplit,
                                 float* partials, int nt, const void* next_weights, long long next_bytes, void* stream) {
  if (!A || !Wp || !lap_panel_gemm_ok(M, N, K, ksplit) || (lda & 7) || norm < 0 || norm > 2) return LAP_ERR_ARG;
  if (ksplit > 1 && !partials) return LAP_ERR_ARG;
  if (partials ? (bias || residual || (flags & LAP_GEMM_GELU) || (ksplit > 1 && norm)) : (!C || (ldc & 3))) return LAP_ERR_ARG;
  // buffer descriptors: 32-bit byte ranges, and the A side marks rows past M with offset 0x80000000
  if (((long long)(M - 1) * lda + K) * 2 >= (1LL << 31) || (long long)N * K * 2 >= (1LL << 32)) return LAP_ERR_ARG;
  if (norm && !gamma) return LAP_ERR_ARG;
  if (norm == 2 && !beta) return LAP_ERR_ARG;
  if (residual && (ldr & 3)) return LAP_ERR_ARG;
  if (flags & ~(LAP_GEMM_GELU | LAP_GEMM_GELU_BF16 | LAP_GEMM_GELU_EXP2)) return LAP_ERR_ARG;
  if ((flags & LAP_GEMM_GELU_EXP2) && (flags & (LAP_GEMM_GELU | LAP_GEMM_GELU_BF16)) != (LAP_GEMM_GELU | LAP_GEMM_GELU_BF16)) return LAP_ERR_ARG;
  PanelP p = {};
  p.A = (const bf16*)A; p.lda = lda; p.Wp = (const bf16*)Wp; p.M = M; p.N = N; p.K = K / ksplit; p.Kfull = K;
  p.gamma = gamma; p.beta = beta; p.eps = eps; p.norm = norm; p.bias = bias; p.R = (const bf16*)residual; p.ldr = ldr;
  p.C = (bf16*)C; p.ldc = ldc; p.part = partials;
  p.gelu = (flags & LAP_GEMM_GELU) ? ((flags & LAP_GEMM_GELU_EXP2) ? 3 : (flags & LAP_GEMM_GELU_BF16) ? 2 : 1) : 0;
  p.nks = ksplit;
  if (next_weights && next_bytes >= 16) { p.pf_ptr = (const char*)next_weights; p.pf_bytes = next_bytes & ~15LL; }
  static const int dbg = getenv("LAP_PANEL_DBG") ? atoi(getenv("LAP_PANEL_DBG")) : 0;
  p.dbg = dbg;
  const int bm = p.K <= 1216 ? 64 : 32;
  if (nt <= 0) {   // one round of <= 256 blocks with the narrowest columns (the shortest weight stream per CU)
    const int npanel = (M + bm - 1) / bm;
    for (nt = 1; nt < 4; ++nt) {
      const int ncol = ((N >> 4) + PN_WAVES * nt - 1) / (PN_WAVES * nt);
      if (((ncol * ksplit + 7) / 8) * 8 * npanel <= 256) break;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  if (bm == 64) {
    if (p.K <= 512) return launch_panel_nt<64, 1>(p, nt, s);
    if (p.K <= 1024) return launch_panel_nt<64, 2>(p, nt, s);
    return launch_panel_nt<64, 3>(p, nt, s);
  }
  return launch_panel_nt<32, 4>(p, nt, s);
}

extern "C" int lap_panel_gemm(const void* A, int lda, const void* Wp, void* C, int ldc, const float* bias, const void* residual, int ldr,
                              int norm, const float* gamma, const float* beta, float eps, int M, int N, int K, int flags, int ksplit,
                              float* partials, int nt, void* stream) {
  return lap_panel_gemm_pf(A, lda, Wp, C, ldc, bias, residual, ldr, norm, gamma, beta, eps, M, N, K, flags, ksplit, partials, nt, nullptr, 0, stream);
}
