// Shared pieces of the bf16 (gemm.hip) and fp8 (gemm_fp8.hip) MFMA GEMM kernels: launch parameters, the logical tile
// order, and the epilogues (direct 4-wide stores, LDS-staged full-row stores of the 256x256 kernels).
#pragma once
#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

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
  int bias_kind;         // 0 none, 1 bf16, 2 f32
  int gelu, accum;
  int ksplit, ktiles_per_split;
  float* part;           // split-K scratch: f32 [ksplit][M][N] partial products (null: atomic accumulate into C)
  int tile_base;         // this launch covers the logical tiles [tile_base, tile_base + gridDim.x)
  int tile_count;        // host side only: tiles of this launch (0 = all from tile_base)
  int part_compact;      // partial slabs are [ksplit][gridDim.x][256][256] (tail split of the 256x256 kernel)
  int sub256;            // 128x128 launch that covers tiles [tile_base, ..) of the 256x256 grid, 4 blocks (quadrants) per tile
  int geglu;             // tile 15 only: N = 2H rows of B are [gate | up]; a block computes 128 gate + the matching 128 up columns and
                         // stores act = bf16(bf16(gelu(gate)) * up) [M][ldc] (gemma.py:303-312): the serving prefill's gate|up + GeGLU
  int epi_lds;           // output of the 256x256 kernel goes out through LDS in full rows (set by the host); 2: with nontemporal stores
  int dbg;               // LAP_GEMM_EXPERIMENTAL builds only: ablation bits of gemm_sp_kernel (1: no in-loop LDS-DMA, 2: no MFMA)
  const float* qscale_a; // fp8 kernels: device scalars s_a, s_b the operands were multiplied by before rounding to e4m3;
  const float* qscale_b; //   the product is divided by s_a * s_b (alpha applies on top)
};

// logical tile -> (m-tile, n-tile): groups of GM m-tiles sweep n so that neighbouring tiles share operand panels
template <int GM>
__device__ __forceinline__ void tile_coords(const GemmParams& p, int t, int& tm, int& tn) {
  const int group_sz = GM * p.tiles_n;
  const int first_m = (t / group_sz) * GM;
  const int gm = min(p.tiles_m - first_m, GM);
  tm = first_m + (t % group_sz) % gm;
  tn = (t % group_sz) / gm;
}

// 64-byte-row K-contiguous tile (BK = 32): 16-byte chunk c of row r is stored at chunk c ^ ((-(r >> 2)) & 3), which
// makes every 16-lane service group of ds_read_b128 cover all 64 banks once.
__device__ __forceinline__ unsigned kc32_tile_off(int row, int chunk16) {
  return (unsigned)(row * 64 + ((chunk16 ^ ((-(row >> 2)) & 3)) << 4));
}
__device__ __forceinline__ bf16x8 kc32_frag(const char* tile, int row0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(tile + kc32_tile_off(row0 + i, g));
}

// Epilogue for 4 consecutive outputs C[m][n..n+3] held by one lane (shared by every kernel shape).
template <bool OUT_F32>
__device__ __forceinline__ void store_tile4(const GemmParams& p, int m, int n, f32x4 a) {
  if (p.part) {  // split-K partial: raw product, epilogue happens in splitk_reduce_kernel
    if (p.part_compact) {
      const int slot = xcd_remap(blockIdx.x, gridDim.x);
      *reinterpret_cast<f32x4*>(p.part + (((long long)blockIdx.y * gridDim.x + slot) << 16) + ((m & 255) << 8) + (n & 255)) = a;
    } else {
      *reinterpret_cast<f32x4*>(p.part + ((long long)blockIdx.y * p.M + m) * p.N + n) = a;
    }
    return;
  }
  f32x4 v = a * p.alpha;
  if (p.bias_kind == 1) {
    bf16x4 b = *reinterpret_cast<const bf16x4*>((const bf16*)p.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
  } else if (p.bias_kind == 2) {
    v += *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
  }
  if (p.gelu) {   // 2: the pre-activation is rounded to bf16 first (= a bf16 Dense output followed by a GELU kernel)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(p.gelu == 2 ? round_bf16(v[e]) : v[e]);
  }
  if (p.R) {
    bf16x4 r = *reinterpret_cast<const bf16x4*>(p.R + (long long)m * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
  }
  if (OUT_F32) {
    float* c = (float*)p.C + (long long)m * p.ldc + n;
    if (p.ksplit > 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(c + e, v[e]);
    } else {
      if (p.accum) v += *reinterpret_cast<const f32x4*>(c);
      *reinterpret_cast<f32x4*>(c) = v;
    }
  } else {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x4*>((bf16*)p.C + (long long)m * p.ldc + n) = o;
  }
}

// Epilogue of the 256x256 kernels through LDS (set by the host: p.epi_lds).  bf16: straight from the accumulators a lane
// owns 8-byte pieces of 16 different rows (32-byte runs per row: 64 narrow stores per wave instruction, store-issue bound,
// and with one block per CU nothing overlaps them - ~7 us per tile, 11-18 % of a K <= 2048 tile).  The operand tiles are
// dead after the k-loop, so the finished bf16 tile is written to LDS (row pitch 528 B, conflict-free for the
// ds_write_b64 pattern) and leaves as 16-byte stores, two full 512-byte rows per wave instruction.
// alpha / bias / GELU / residual are applied in registers exactly as in the direct path: same bits.  f32 (weight gradients,
// logits): the same in two halves of 128 rows (a 256 x 256 f32 tile does not fit; 1040-byte pitch), four full 1 KiB rows
// per wave instruction; with beta = 1 the old values come in the same coalesced way.
template <int NW, int WTM, int WTN, bool OUT_F32>
__device__ __forceinline__ void staged_epilogue(const GemmParams& p, char* smem, f32x4 (&acc)[WTM / 16][WTN / 16], int wm, int wn,
                                                int m0, int n0, int tid, int lane) {
  constexpr int FM = WTM / 16, FN = WTN / 16, BM = 256, BN = 256;
  const int li = lane & 15, lg = lane >> 4;
  if constexpr (!OUT_F32) {
    constexpr int CP = BN * 2 + 16;   // 528 B: the 32 lanes of a ds_write_b64 half-wave (16 rows x 2 column groups) hit 64 distinct banks
    __syncthreads();   // every wave is done reading the operand tiles (no LDS-DMA in flight after the last k-tile)
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int ml = wm * WTM + i * 16 + li;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int nl = wn * WTN + j * 16 + 4 * lg;
        const int m = m0 + ml, n = n0 + nl;
        f32x4 v = acc[i][j] * p.alpha;
        if (m < p.M && n < p.N) {
          if (p.bias_kind == 1) {
            bf16x4 b = *reinterpret_cast<const bf16x4*>((const bf16*)p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
          } else if (p.bias_kind == 2) {
            v += *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
          }
          if (p.gelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(p.gelu == 2 ? round_bf16(v[e]) : v[e]);
          }
          if (p.R) {
            bf16x4 r = *reinterpret_cast<const bf16x4*>(p.R + (long long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
          }
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x4*>(smem + ml * CP + nl * 2) = o;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < BM * BN / 8 / (NW * 64); ++it) {
      const int id = tid + NW * 64 * it;
      const int r = id >> 5, c = id & 31;
      const int m = m0 + r, n = n0 + c * 8;
      if (m < p.M && n < p.N)   // N % 8 == 0 on this path: a 16-byte piece is inside or outside as a whole
      {
        const bf16x8 val = *reinterpret_cast<const bf16x8*>(smem + r * CP + c * 16);
        bf16x8* dst = reinterpret_cast<bf16x8*>((bf16*)p.C + (long long)m * p.ldc + n);
        if (p.epi_lds == 2) __builtin_nontemporal_store(val, dst);   // full rows, written once: keep them out of the L2's way
        else *dst = val;
      }
    }
  } else {
    constexpr int CP = BN * 4 + 16;   // 1040 B: the 16 lanes of a ds_write_b128 group (16 rows) hit 64 distinct banks
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();   // operand tiles dead (first pass) / previous half moved out (second pass)
      if ((wm * WTM) / 128 == half) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int ml = (wm * WTM) % 128 + i * 16 + li;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int nl = wn * WTN + j * 16 + 4 * lg;
            const int n = n0 + nl;
            f32x4 v = acc[i][j] * p.alpha;
            if (n < p.N) {
              if (p.bias_kind == 1) {
                bf16x4 b = *reinterpret_cast<const bf16x4*>((const bf16*)p.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)b[e];
              } else if (p.bias_kind == 2) {
                v += *reinterpret_cast<const f32x4*>((const float*)p.bias + n);
              }
            }
            *reinterpret_cast<f32x4*>(smem + ml * CP + nl * 4) = v;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 128 * BN / 4 / (NW * 64); ++it) {
        const int id = tid + NW * 64 * it;
        const int r = id >> 6, c = id & 63;
        const int m = m0 + half * 128 + r, n = n0 + c * 4;
        if (m < p.M && n < p.N) {
          f32x4 v = *reinterpret_cast<const f32x4*>(smem + r * CP + c * 16);
          float* dst = (float*)p.C + (long long)m * p.ldc + n;
          if (p.accum) v += *reinterpret_cast<const f32x4*>(dst);
          if (p.epi_lds == 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
          else *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace
