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
// Production kernels: the software-pipelined 8-wave 256x256 kernel (gemm_sp_kernel, "tile 10") for K % 64 == 0, the
// 16-wave lockstep kernel below ("tile 5") otherwise, the 128x128 kernel ("tile 6") for small shapes and short-K tails.
// Structure of the generic kernel: BM x BN x BK block tile, WGM x WGN waves, each wave a (BM/WGM) x
// (BN/WGN) sub-tile of v_mfma_f32_16x16x32_bf16 fragments.  Production shapes:
//   256x256x64 / 16 waves (64x64 per wave): 128 FLOP per L2 byte, one block per CU
//            (128 KiB LDS), 4 waves per SIMD hide the ds_read -> MFMA latency and the
//            per-k-tile barrier — the large GEMMs (1.0-1.3 PF measured);
//   128x128x64 /  8 waves (64x32 per wave): 64 FLOP/B, two blocks per CU — small or
//            awkward shapes (a 128x128 tile at the 2.5 PF MFMA peak would need
//            39 TB/s from L2, more than the ~35 TB/s the XCD L2s deliver).
// Other instantiations (8-wave 256x256, BK=32 4-stage rings, 256x128, the two ping-pong
// kernels = tiles 8 / 9) are kept selectable through lap_gemm_bf16_ex for A/B measurements.
// HBM -> LDS staging is buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round trip),
// double buffered: tile t+1 streams in while tile t is multiplied, one barrier
// per k-tile.  The LDS image is lane-linear per wave instruction, so the
// bank-conflict swizzle is applied to the per-lane *source* address and again on
// the fragment read (common.hpp).  Out-of-range rows / k-tails use an
// out-of-bounds buffer offset, for which the hardware writes zeros.
// Split-K (f32 output, atomic accumulate) covers weight gradients whose output
// has too few tiles to fill 256 CUs.
#include "common.hpp"
#include <stdlib.h>
#include "gemm_common.hpp"

// The staged epilogue's full-row stores go out nontemporal (C is written once and read from HBM by its consumer anyway; the
// operand panels keep the L2): -0.3 .. -1.0 ms per train step in three A/B pairs, serving unchanged.  LAP_GEMM_NT_STORE=0: plain.
static int epi_lds_mode() {
  static const int mode = (getenv("LAP_GEMM_NT_STORE") && atoi(getenv("LAP_GEMM_NT_STORE")) == 0) ? 1 : 2;
  return mode;
}
namespace {


// BK: k-depth of one LDS stage (32 or 64); NS: LDS stages.  Loads of tile t+NS-1 are issued while tile t is
// multiplied; the wait before the (single, raw) barrier is a COUNTED vmcnt that leaves NS-2 tiles in flight.
template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool A_KC, bool B_KC, bool OUT_F32>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_kernel(GemmParams p) {
  constexpr int NW = WGM * WGN;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile
  constexpr int FM = WTM / 16, FN = WTN / 16;     // fragments per wave
  constexpr int KSUB = BK / 32;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = A_BYTES / 1024 / NW, PB = B_BYTES / 1024 / NW;  // 1 KiB LDS-DMA pieces per wave
  static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tile / wave mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NS stages: [A tile | B tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;

  // Block -> tile mapping: XCD-aware remap, then groups of GM m-tiles sweep n.
  constexpr int GM = (BM == 256) ? 4 : 8;
  int tm, tn;
  if (BM == 128 && BN == 128 && p.sub256) {
    // last round of a 256x256 grid re-tiled: 4 consecutive blocks (same XCD) are the quadrants of one 256x256 tile
    const int b = xcd_remap(blockIdx.x, gridDim.x);
    tile_coords<4>(p, p.tile_base + (b >> 2), tm, tn);   // (p.tiles_m / tiles_n describe the 256x256 grid here)
    tm = 2 * tm + ((b >> 1) & 1);
    tn = 2 * tn + (b & 1);
    if (tm * BM >= p.M || tn * BN >= p.N) return;   // quadrant outside the matrix (uniform per block)
  } else {
    tile_coords<GM>(p, p.tile_base + xcd_remap(blockIdx.x, gridDim.x), tm, tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0, (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);

  // Staging descriptors.  Piece q covers linear 16-byte chunks [64q, 64q+64) of the tile image;
  // K-contiguous tile: 8 chunks per row (row = m/n index); M-contiguous tile: BM/8 chunks per k-row.
  unsigned offA[PA], offB[PB];
  int kidxA[PA], kidxB[PB];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int ci = (w * PA + j) * 64 + lane;
    if (A_KC) {
      const int row = BK == 64 ? ci >> 3 : ci >> 2, pc = BK == 64 ? ci & 7 : ci & 3;
      const int c = BK == 64 ? pc ^ ((row >> 1) & 7) : pc ^ ((-(row >> 2)) & 3);
      kidxA[j] = c * 8;
      offA[j] = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + c * 8) * 2) : OOB;
    } else {
      constexpr int CPR = BM / 8;
      const int kr = ci / CPR, pc = ci % CPR;
      const int c = pc ^ (mc_swz(kr) << 1);
      kidxA[j] = kr;
      offA[j] = (m0 + c * 8 < p.M) ? (unsigned)(((long long)kr * p.lda + m0 + c * 8) * 2) : OOB;
    }
  }
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int ci = (w * PB + j) * 64 + lane;
    if (B_KC) {
      const int row = BK == 64 ? ci >> 3 : ci >> 2, pc = BK == 64 ? ci & 7 : ci & 3;
      const int c = BK == 64 ? pc ^ ((row >> 1) & 7) : pc ^ ((-(row >> 2)) & 3);
      kidxB[j] = c * 8;
      int brow = n0 + row;
      if constexpr (BM == 320 && BN == 256 && !OUT_F32) {
        // GeGLU pairing: wave column group wn (64 tile columns) = 32 gate columns + the 32 up columns of the same features
        if (p.geglu) brow = ((row & 63) < 32 ? 0 : p.N / 2) + tn * 128 + (row >> 6) * 32 + (row & 31);
      }
      offB[j] = (brow < p.N) ? (unsigned)(((long long)brow * p.ldb + c * 8) * 2) : OOB;
    } else {
      constexpr int CPR = BN / 8;
      const int kr = ci / CPR, pc = ci % CPR;
      const int c = pc ^ (mc_swz(kr) << 1);
      kidxB[j] = kr;
      offB[j] = (n0 + c * 8 < p.N) ? (unsigned)(((long long)kr * p.ldb + n0 + c * 8) * 2) : OOB;
    }
  }
  const int kt0_probe = blockIdx.y * p.ktiles_per_split + NS - 1;
  unsigned stepA = A_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.lda * 2);
  unsigned stepB = B_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.ldb * 2);
#ifdef LAP_GEMM_EXPERIMENTAL
  // timing probe (results wrong): operand tiles fetched as if the operand were stored tile by tile, 1 KiB contiguous per LDS-DMA
  // piece (dbg bit 8: B, bit 16: A) instead of 8 rows x 128 bytes — what a fragment-packed operand image would cost the loop
  {
    const int nkt = (p.K + BK - 1) / BK;
    if (p.dbg & 8) {
#pragma unroll
      for (int j = 0; j < PB; ++j) { offB[j] = (unsigned)(((long long)tn * nkt) * B_BYTES + (w * PB + j) * 1024 + lane * 16); kidxB[j] = 0; }
      stepB = B_BYTES;
    }
    if (p.dbg & 16) {
#pragma unroll
      for (int j = 0; j < PA; ++j) { offA[j] = (unsigned)(((long long)tm * nkt) * A_BYTES + (w * PA + j) * 1024 + lane * 16); kidxA[j] = 0; }
      stepA = A_BYTES;
    }
  }
#endif

  // K a multiple of the k-tile (every production shape): no piece ever reaches past K, so a piece's per-lane offset is the
  // loop-invariant offA / offB (OOB for rows outside the matrix: 0x80000000 is out of range whatever is added) and the k-tile's
  // byte offset travels in the instruction's SCALAR offset — no per-piece compare / select / add in the loop (round 5: the
  // small-tile loops issue ~100 instructions per k-step around 8 - 16 MFMAs; the selects and their exec-mask branches were 40).
  const bool k_even = (p.K % BK) == 0;
#ifdef LAP_GEMM_EXPERIMENTAL
  const bool probe_no_dma = p.dbg & 32, probe_no_mma = p.dbg & 64;   // timing ablations of the k-loop (results wrong)
#else
  constexpr bool probe_no_dma = false, probe_no_mma = false;
#endif
  auto stage = [&](int buf, int kt) {
    const int k0 = kt * BK;
    char* base = smem + buf * STAGE;
    if (probe_no_dma && kt >= kt0_probe) return;
    if (k_even) {
      const unsigned sa = (unsigned)kt * stepA, sb = (unsigned)kt * stepB;
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        const unsigned va = offA[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base + (w * PA + j) * 1024), 16, va, sa, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const unsigned vb = offB[j];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + A_BYTES + (w * PB + j) * 1024), 16, vb, sb, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        unsigned va = (offA[j] != OOB && k0 + kidxA[j] < p.K) ? offA[j] + (unsigned)kt * stepA : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base + (w * PA + j) * 1024), 16, va, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        unsigned vb = (offB[j] != OOB && k0 + kidxB[j] < p.K) ? offB[j] + (unsigned)kt * stepB : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + A_BYTES + (w * PB + j) * 1024), 16, vb, 0, 0, 0);
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * p.ktiles_per_split;
  const int kt1 = min(nkt_all, kt0 + p.ktiles_per_split);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (kt0 + s < kt1) stage(s, kt0 + s);
  int cur = 0;
  for (int kt = kt0; kt < kt1; ++kt) {
    // my loads of tile kt have landed (NS-2 younger tiles may still be in flight) ...
    if (NS > 2 && kt + NS - 2 < kt1) wait_vmcnt<(NS - 2) * (PA + PB)>(); else wait_vmcnt<0>();
    // ... and so have everybody else's; every wave is also done reading the buffer refilled below.
    __builtin_amdgcn_s_barrier();
    if (kt + NS - 1 < kt1) stage(cur == 0 ? NS - 1 : cur - 1, kt + NS - 1);
    const char* tA = smem + cur * STAGE;
    const char* tB = tA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
      bf16x8 fa[FM], fb[FN];
      // M-contiguous operands: transposing reads issued raw (the builtin makes the compiler drain the LDS-DMA
      // prefetch of the next tile first, see common.hpp), fenced once per k-step
      bf16x4 ta[A_KC ? 1 : FM][2], tb[B_KC ? 1 : FN][2];
      if (!A_KC) {
#pragma unroll
        for (int i = 0; i < FM; ++i) mc_frag_raw<BM>(tA, wm * WTM + i * 16, kk, lane, ta[i]);
      }
      if (!B_KC) {
#pragma unroll
        for (int j = 0; j < FN; ++j) mc_frag_raw<BN>(tB, wn * WTN + j * 16, kk, lane, tb[j]);
      }
      constexpr bool RAW_KC = BK == 64;   // K-contiguous operands through raw b128 reads as well (+1-3 %)
      if (A_KC) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
          fa[i] = RAW_KC ? kc_frag_raw(tA, wm * WTM + i * 16, kk, lane) : kc32_frag(tA, wm * WTM + i * 16, lane);
      }
      if (B_KC) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
          fb[j] = RAW_KC ? kc_frag_raw(tB, wn * WTN + j * 16, kk, lane) : kc32_frag(tB, wn * WTN + j * 16, lane);
      }
      if (RAW_KC || !A_KC || !B_KC) {
        lds_wait_all();
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          if (!A_KC) { lds_tie(ta[i][0]); lds_tie(ta[i][1]); fa[i] = join8(ta[i][0], ta[i][1]); }
          else if (RAW_KC) lds_tie(fa[i]);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (!B_KC) { lds_tie(tb[j][0]); lds_tie(tb[j][1]); fb[j] = join8(tb[j][0], tb[j][1]); }
          else if (RAW_KC) lds_tie(fb[j]);
        }
      }
      if (!probe_no_mma) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          // Operands swapped on purpose: D[row = n][col = m], so each lane ends up
          // with 4 consecutive n of one output row m -> one 8/16-byte store.
          acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
      }
    }
    cur = (cur + 1 == NS) ? 0 : cur + 1;
  }

  // Epilogue. Lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + 4*(lane>>4).
  const int li = lane & 15, lg = lane >> 4;
  if constexpr (BM == 256 && BN == 256 && NW == 16) {
    if (p.epi_lds) { staged_epilogue<NW, WTM, WTN, OUT_F32>(p, smem, acc, wm, wn, m0, n0, tid, lane); return; }
  }
  if constexpr (BM == 320 && BN == 256 && !OUT_F32) {
    if (p.geglu) {   // fragments j = 0, 1: gate columns; j = 2, 3: the up columns of the same features (csrc/elementwise.hip geglu_fwd's bits)
      static_assert(FN == 4, "GeGLU pairing assumes 64-column wave tiles");
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int gc = tn * 128 + wn * 32 + j * 16 + 4 * lg;
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gt = round_bf16(acc[i][j][e]);
            // geglu 2: the GELU of lap_gemm_asm_geglu_fwd (the training step's kernel: v_exp / v_rcp), 80 of them per lane are 10 us of tanhf otherwise
            o[e] = f2bf(round_bf16(p.geglu == 2 ? gelu_exp2_f(gt) : gelu_tanh_f(gt)) * round_bf16(acc[i][j + 2][e]));
          }
          *reinterpret_cast<bf16x4*>((bf16*)p.C + (long long)m * p.ldc + gc) = o;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * lg;
      if (n >= p.N) continue;  // N % 4 == 0 is required by the host wrapper
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

#ifdef LAP_GEMM_EXPERIMENTAL   // ping-pong probes (tiles 8 / 9): measured, documented in DESIGN.md §4, not production

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong 256x256x64 kernel: 8 waves (2 x 4), 128x64 per wave, one block per CU.  Each k-tile is cut into four
// phases (one 64x32 quadrant of the wave's output x K = 64 = 16 MFMAs); a phase is
//     [ds_read the fragments this quadrant needs | issue ONE half-tile of LDS-DMA | counted vmcnt]  barrier
//     [16 MFMAs]  barrier
// and the second wave row runs one barrier behind the first, so on every SIMD one wave is in its MFMA segment while
// its partner is in its LDS segment: the matrix pipe and the LDS port work concurrently by construction instead of
// by luck of the wave scheduler.  Quadrant order (0,0) (0,1) (1,1) (1,0) reuses the A half for two phases and keeps
// both B halves in registers: 12 / 4 / 8 / 0 ds_read_b128 per phase.
// Staging granularity is the HALF tile (the 128 rows of A, or 128 columns of B, that one quadrant row / column
// reads: a 16 KiB image, 2 LDS-DMA pieces per wave), 8 slots = 2 k-tiles.  A half is dead as soon as its quadrant
// has been read, so it is refilled with the k-tile after next: phase 1 issues B1(kt+1), phase 2 A1(kt+1), phase 3
// A0(kt+2), phase 4 B0(kt+2) - every half is requested five phases (1.25 k-tiles) before its first reader and four
// halves (64 KiB per block) are in flight at every wait, which is always the same `s_waitcnt vmcnt(8)` (never 0
// inside the loop; past the end of K the pieces are out-of-range buffer loads, i.e. zero fills without traffic).
// Ordering: a wait in the LDS segment of phase p makes the half visible to readers from phase p + 1 on (one barrier
// more than usual because the two groups are a barrier apart); a half is refilled at the earliest two phases after
// its last read.  All fragment reads are raw (common.hpp), fenced by the lgkmcnt(0) that opens the MFMA segment.
template <bool A_KC, bool B_KC, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, HALF = 128 * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // slot (buf, h): h = 0 A0, 1 A1, 2 B0, 3 B1
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;   // waves 0-3: row 0 (leading group), waves 4-7: row 1 (one barrier behind)

  const int t = p.tile_base + xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords<4>(p, t, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0, (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);
  // Half images.  A half mh holds the rows {wm' * 128 + mh * 64 + r} at local index wm' * 64 + r; B half nh holds the
  // columns {wn' * 64 + nh * 32 + c} at local index wn' * 32 + c.  K-contiguous: [128 rows][64 k]; M-contiguous:
  // [64 k][128 m].  Piece q = 2 w + j covers linear 16-byte chunks [64 q, 64 q + 64) of the image.
  unsigned off[4][2];
  int kidxA[2], kidxB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = (w * 2 + j) * 64 + lane;
    int la, ca, lb, cb;   // local row / column index, logical 16-byte chunk
    if (A_KC) { la = ci >> 3; ca = (ci & 7) ^ ((la >> 1) & 7); kidxA[j] = ca * 8; }
    else { const int kr = ci >> 4; ca = (ci & 15) ^ (mc_swz(kr) << 1); la = ca * 8; kidxA[j] = kr; }
    if (B_KC) { lb = ci >> 3; cb = (ci & 7) ^ ((lb >> 1) & 7); kidxB[j] = cb * 8; }
    else { const int kr = ci >> 4; cb = (ci & 15) ^ (mc_swz(kr) << 1); lb = cb * 8; kidxB[j] = kr; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gm = m0 + (la >> 6) * 128 + h * 64 + (la & 63);
      const int gn = n0 + (lb >> 5) * 64 + h * 32 + (lb & 31);
      off[h][j] = gm < p.M ? (unsigned)((A_KC ? (long long)gm * p.lda + ca * 8 : (long long)kidxA[j] * p.lda + gm) * 2) : OOB;
      off[2 + h][j] = gn < p.N ? (unsigned)((B_KC ? (long long)gn * p.ldb + cb * 8 : (long long)kidxB[j] * p.ldb + gn) * 2) : OOB;
    }
  }
  const unsigned stepA = A_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.lda * 2);
  const unsigned stepB = B_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.ldb * 2);

  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * p.ktiles_per_split;
  const int kt1 = min(nkt_all, kt0 + p.ktiles_per_split);
  const int kend = min(p.K, kt1 * BK);
  auto issue = [&](int buf, int h, int kt) {
    char* base = smem + (buf * 4 + h) * HALF + w * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kidx = h < 2 ? kidxA[j] : kidxB[j];
      const unsigned v = (off[h][j] != OOB && kt * BK + kidx < kend) ? off[h][j] + (unsigned)kt * (h < 2 ? stepA : stepB) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(h < 2 ? rsA : rsB, (LDS_PTR(void))(base + j * 1024), 16, v, 0, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 fa[4][2], fb[2][2][2];                               // A half: [m-frag][kk]; B: [n-half][n-frag][kk]
  bf16x4 ra[A_KC ? 1 : 4][2][2], rb[B_KC ? 1 : 2][2][2][2];   // transposing reads arrive as two 64-bit halves
  auto read_a = [&](const char* img) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (A_KC) fa[i][kk] = kc_frag_raw(img, wm * 64 + i * 16, kk, lane);
        else mc_frag_raw<128>(img, wm * 64 + i * 16, kk, lane, ra[A_KC ? 0 : i][kk]);
      }
  };
  auto read_b = [&](const char* img, int nh) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (B_KC) fb[nh][j][kk] = kc_frag_raw(img, wn * 32 + j * 16, kk, lane);
        else mc_frag_raw<128>(img, wn * 32 + j * 16, kk, lane, rb[B_KC ? 0 : nh][j][kk]);
      }
  };
  auto tie_a = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (A_KC) lds_tie(fa[i][kk]);
        else { lds_tie(ra[A_KC ? 0 : i][kk][0]); lds_tie(ra[A_KC ? 0 : i][kk][1]); fa[i][kk] = join8(ra[A_KC ? 0 : i][kk][0], ra[A_KC ? 0 : i][kk][1]); }
      }
  };
  auto tie_b = [&](int nh) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (B_KC) lds_tie(fb[nh][j][kk]);
        else {
          lds_tie(rb[B_KC ? 0 : nh][j][kk][0]); lds_tie(rb[B_KC ? 0 : nh][j][kk][1]);
          fb[nh][j][kk] = join8(rb[B_KC ? 0 : nh][j][kk][0], rb[B_KC ? 0 : nh][j][kk][1]);
        }
      }
  };
#define PP_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
#define PP_MMA(MH, NH)                                                                         \
  {                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
          acc[MH * 4 + i][NH * 2 + j] = mfma16(fb[NH][j][kk], fa[i][kk], acc[MH * 4 + i][NH * 2 + j]); \
    __builtin_amdgcn_s_setprio(0);                                                             \
  }

  // prologue: the six halves the steady state would already have requested, in its order
  issue(0, 0, kt0); issue(0, 2, kt0); issue(0, 3, kt0); issue(0, 1, kt0); issue(1, 0, kt0 + 1); issue(1, 2, kt0 + 1);
  wait_vmcnt<8>();                            // A0, B0 of the first k-tile
  PP_BAR()
  if (wm == 1) __builtin_amdgcn_s_barrier();  // second wave row drops one barrier behind
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    const char* img = smem + buf * 4 * HALF;
    // phase 1: quadrant (0,0)
    read_a(img);
    read_b(img + 2 * HALF, 0);
    issue(buf ^ 1, 3, kt + 1);
    wait_vmcnt<8>();                          // B1(kt) for phase 2
    PP_BAR()
    lds_wait_all();
    tie_a(); tie_b(0);
    PP_MMA(0, 0)
    PP_BAR()
    // phase 2: quadrant (0,1)
    read_b(img + 3 * HALF, 1);
    issue(buf ^ 1, 1, kt + 1);
    wait_vmcnt<8>();                          // A1(kt) for phase 3
    PP_BAR()
    lds_wait_all();
    tie_b(1);
    PP_MMA(0, 1)
    PP_BAR()
    // phase 3: quadrant (1,1)
    read_a(img + HALF);
    issue(buf, 0, kt + 2);
    PP_BAR()
    lds_wait_all();
    tie_a();
    PP_MMA(1, 1)
    PP_BAR()
    // phase 4: quadrant (1,0): nothing to read
    issue(buf, 2, kt + 2);
    wait_vmcnt<8>();                          // A0, B0 of the next k-tile
    PP_BAR()
    PP_MMA(1, 0)
    PP_BAR()
  }
  wait_vmcnt<0>();
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the stagger
#undef PP_MMA
#undef PP_BAR

  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-wave ping-pong (tile 9): the 256x256 / 64x64-per-wave geometry of the production kernel, but the k-loop advances in
// 32-deep SUB-TILES (one MFMA k-step: 8 fragment reads, 16 MFMAs per wave) and the wave rows alternate between two
// groups one barrier apart (rows 0, 2 lead; rows 1, 3 trail), so that each SIMD always has two waves in their MFMA
// segment (512 pipe cycles) while its other two are in their LDS segment (reads of the next sub-tile, 2 LDS-DMA pieces,
// counted wait).  The lockstep kernel pays a matrix-pipe bubble after every barrier (all 16 waves issue DMA + reads and
// wait for LDS at the same time); here that segment of one group hides under the other group's MFMAs.
// Ring of 4 sub-tile slots (32 KiB each: A [256][32] | B [256][32], the BK = 32 images of common.hpp / kc32_*): phase st
// reads slot st, issues sub-tile st + 2 into the slot last read two phases ago, and `s_waitcnt vmcnt(2)` makes sub-tile
// st + 1 visible for the next phase (wait in phase p -> read in phase p + 1, as for tile 8).
__device__ __forceinline__ bf16x8 kc32_frag_raw(const char* tile, int row0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_addr_of(tile) + kc32_tile_off(row0 + i, g)));
  return r;
}

template <bool A_KC, bool B_KC, bool OUT_F32>
__global__ __launch_bounds__(1024) void gemm_pp16_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, SK = 32, OPB = 256 * SK * 2, SLOT = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 4 slots: [A sub-tile | B sub-tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int trailing = wm & 1;

  const int t = p.tile_base + xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords<4>(p, t, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0, (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);
  // one 1 KiB piece per wave and operand: linear 16-byte chunks [64 w, 64 w + 64) of the sub-tile image
  unsigned offA, offB;
  int kidxA, kidxB;
  {
    const int ci = w * 64 + lane;
    if (A_KC) {
      const int row = ci >> 2, c = (ci & 3) ^ ((-(row >> 2)) & 3);
      kidxA = c * 8;
      offA = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      kidxA = kr;
      offA = (m0 + c * 8 < p.M) ? (unsigned)(((long long)kr * p.lda + m0 + c * 8) * 2) : OOB;
    }
    if (B_KC) {
      const int row = ci >> 2, c = (ci & 3) ^ ((-(row >> 2)) & 3);
      kidxB = c * 8;
      offB = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      kidxB = kr;
      offB = (n0 + c * 8 < p.N) ? (unsigned)(((long long)kr * p.ldb + n0 + c * 8) * 2) : OOB;
    }
  }
  const unsigned stepA = A_KC ? (unsigned)(SK * 2) : (unsigned)((long long)SK * p.lda * 2);
  const unsigned stepB = B_KC ? (unsigned)(SK * 2) : (unsigned)((long long)SK * p.ldb * 2);
  const int st0 = blockIdx.y * p.ktiles_per_split * 2;
  const int st1 = min((p.K + SK - 1) / SK, st0 + p.ktiles_per_split * 2);
  const int kend = min(p.K, st1 * SK);
  auto issue = [&](int st) {   // past the end: out-of-range pieces (zero fill, no traffic) keep the wait counts uniform
    char* base = smem + (st & 3) * SLOT + w * 1024;
    const unsigned va = (offA != OOB && st * SK + kidxA < kend) ? offA + (unsigned)st * stepA : OOB;
    const unsigned vb = (offB != OOB && st * SK + kidxB < kend) ? offB + (unsigned)st * stepB : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base), 16, va, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + OPB), 16, vb, 0, 0, 0);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4], fb[4];
  bf16x4 ra[A_KC ? 1 : 4][2], rb[B_KC ? 1 : 4][2];

#define P16_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  issue(st0); issue(st0 + 1);
  wait_vmcnt<2>();                                // sub-tile st0 landed
  P16_BAR()
  if (trailing) __builtin_amdgcn_s_barrier();     // rows 1, 3 drop one barrier behind
  __builtin_amdgcn_sched_barrier(0);
  for (int st = st0; st < st1; ++st) {
    const char* tA = smem + (st & 3) * SLOT;
    const char* tB = tA + OPB;
    // ---- LDS segment
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (A_KC) fa[i] = kc32_frag_raw(tA, wm * 64 + i * 16, lane);
      else mc_frag_raw<BM>(tA, wm * 64 + i * 16, 0, lane, ra[A_KC ? 0 : i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (B_KC) fb[j] = kc32_frag_raw(tB, wn * 64 + j * 16, lane);
      else mc_frag_raw<BN>(tB, wn * 64 + j * 16, 0, lane, rb[B_KC ? 0 : j]);
    }
    issue(st + 2);
    wait_vmcnt<2>();                              // sub-tile st + 1 (read in the next phase)
    P16_BAR()
    // ---- MFMA segment
    lds_wait_all();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (A_KC) lds_tie(fa[i]);
      else { lds_tie(ra[A_KC ? 0 : i][0]); lds_tie(ra[A_KC ? 0 : i][1]); fa[i] = join8(ra[A_KC ? 0 : i][0], ra[A_KC ? 0 : i][1]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (B_KC) lds_tie(fb[j]);
      else { lds_tie(rb[B_KC ? 0 : j][0]); lds_tie(rb[B_KC ? 0 : j][1]); fb[j] = join8(rb[B_KC ? 0 : j][0], rb[B_KC ? 0 : j][1]); }
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    P16_BAR()
  }
  wait_vmcnt<0>();
  if (!trailing) __builtin_amdgcn_s_barrier();    // balance the stagger
#undef P16_BAR

  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

#endif  // LAP_GEMM_EXPERIMENTAL

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined 8-wave kernel (tile 10): 256x256x64 tile, 128x64 per wave, TWO fragment register sets.  Per k-tile
//   A1  set 0 = (kt, k-half 0) is back: 8 MFMAs, issue the raw reads of (kt, k-half 1) into set 1, 8 more MFMAs
//   A2  set 1 landed, my LDS-DMA of tile kt+1 landed, barrier          (every read of tile kt has completed)
//   A3  last 16 MFMAs on set 0, with the 8 LDS-DMA pieces of tile kt+2 threaded between them (into kt's buffer)
//   C   8 MFMAs on set 1 = (kt, k-half 1), issue the raw reads of (kt+1, k-half 0) into set 0, the other 24 MFMAs
// (reads are always issued AFTER a first group of MFMAs has been queued: +1 ms per train step over issuing them first)
// so that after the barrier every wave already holds 48 MFMAs of work whose operands are in registers: the next reads'
// latency, the DMA issue cost and the barrier skew hide under them instead of idling the matrix pipe (the lockstep
// kernel's bubble).  One barrier per k-tile, two LDS buffers, prefetch distance one k-tile.  K % 64 == 0.
template <int OFF>
__device__ __forceinline__ bf16x8 ds_read_b128_raw(unsigned addr) {   // valid after lds_wait_*() + lds_tie(), like kc_frag_raw
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <int WGM, int WGN, bool A_KC, bool B_KC, bool OUT_F32, bool TWOB = false>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_sp_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, A_BYTES = BM * BK * 2, STAGE = 2 * A_BYTES;
  constexpr int NW = WGM * WGN, WTM = BM / WGM, WTN = BN / WGN, FM = WTM / 16, FN = WTN / 16, PC = 32 / NW;   // PC pieces / operand / wave
  static_assert(FM == 8 && (FN == 4 || FN == 8), "wave tile 128 x 64 or 128 x 128");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  int tm, tn;
  tile_coords<4>(p, p.tile_base + xcd_remap(blockIdx.x, gridDim.x), tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0, (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);
  unsigned offA[PC], offB[PC];
  int kcA[PC], kcB[PC];     // K-contiguous operands: first k of the lane's chunk inside a k-tile (K % 64 != 0: chunks past K read as zeros)
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int ci = (w * PC + j) * 64 + lane;
    kcA[j] = 0; kcB[j] = 0;
    if (A_KC) {
      const int row = ci >> 3, c = (ci & 7) ^ ((row >> 1) & 7);
      kcA[j] = c * 8;
      offA[j] = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      offA[j] = (m0 + c * 8 < p.M) ? (unsigned)(((long long)kr * p.lda + m0 + c * 8) * 2) : OOB;
    }
    if (B_KC) {
      const int row = ci >> 3, c = (ci & 7) ^ ((row >> 1) & 7);
      kcB[j] = c * 8;
      offB[j] = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      offB[j] = (n0 + c * 8 < p.N) ? (unsigned)(((long long)kr * p.ldb + n0 + c * 8) * 2) : OOB;
    }
  }
  const unsigned stepA = A_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.lda * 2);
  const unsigned stepB = B_KC ? (unsigned)(BK * 2) : (unsigned)((long long)BK * p.ldb * 2);
  const int nkt_all = (p.K + BK - 1) / BK;   // (rows of an M- / N-contiguous operand past K lie outside its descriptor: zeros)
  const int kt0 = blockIdx.y * p.ktiles_per_split;
  const int kt1 = min(nkt_all, kt0 + p.ktiles_per_split);
  auto piece = [&](int buf, int kt, int j) {   // j < 4: operand A, else B; past the end: zero fill, keeps the counts uniform
    char* base = smem + buf * STAGE;
    if (j < PC) {
      const unsigned v = (offA[j] != OOB && kt < kt1 && (!A_KC || kt * BK + kcA[j] < p.K)) ? offA[j] + (unsigned)kt * stepA : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base + (w * PC + j) * 1024), 16, v, 0, 0, 0);
    } else {
      const unsigned v = (offB[j - PC] != OOB && kt < kt1 && (!B_KC || kt * BK + kcB[j - PC] < p.K)) ? offB[j - PC] + (unsigned)kt * stepB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + A_BYTES + (w * PC + j - PC) * 1024), 16, v, 0, 0, 0);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][FM], fb[2][FN];
  bf16x4 ra[A_KC ? 1 : 2][A_KC ? 1 : FM][2], rb[B_KC ? 1 : 2][B_KC ? 1 : FN][2];
  // K-contiguous fragments of one operand differ by 16 rows = 2048 bytes and share the swizzle ((row >> 1) & 7 does not
  // see multiples of 16): one address register + immediate offsets instead of one register per read
  auto reads = [&](int set, const char* tA, int kk) {
    const char* tB = tA + A_BYTES;
    if (B_KC) {
      const unsigned b0 = lds_addr_of(tB) + kc_tile_off(wn * WTN + (lane & 15), kk * 4 + (lane >> 4));
      fb[set][0] = ds_read_b128_raw<0>(b0); fb[set][1] = ds_read_b128_raw<2048>(b0);
      fb[set][2] = ds_read_b128_raw<4096>(b0); fb[set][3] = ds_read_b128_raw<6144>(b0);
      if constexpr (FN == 8) {
        fb[set][4] = ds_read_b128_raw<8192>(b0); fb[set][5] = ds_read_b128_raw<10240>(b0);
        fb[set][6] = ds_read_b128_raw<12288>(b0); fb[set][7] = ds_read_b128_raw<14336>(b0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j) mc_frag_raw<BN>(tB, wn * WTN + j * 16, kk, lane, rb[B_KC ? 0 : set][B_KC ? 0 : j]);
    }
    if (A_KC) {
      const unsigned a0 = lds_addr_of(tA) + kc_tile_off(wm * WTM + (lane & 15), kk * 4 + (lane >> 4));
      fa[set][0] = ds_read_b128_raw<0>(a0); fa[set][1] = ds_read_b128_raw<2048>(a0);
      fa[set][2] = ds_read_b128_raw<4096>(a0); fa[set][3] = ds_read_b128_raw<6144>(a0);
      fa[set][4] = ds_read_b128_raw<8192>(a0); fa[set][5] = ds_read_b128_raw<10240>(a0);
      fa[set][6] = ds_read_b128_raw<12288>(a0); fa[set][7] = ds_read_b128_raw<14336>(a0);
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) mc_frag_raw<BM>(tA, wm * WTM + i * 16, kk, lane, ra[A_KC ? 0 : set][A_KC ? 0 : i]);
    }
  };
  auto tie = [&](int set) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (B_KC) lds_tie(fb[set][j]);
      else { lds_tie(rb[B_KC ? 0 : set][B_KC ? 0 : j][0]); lds_tie(rb[B_KC ? 0 : set][B_KC ? 0 : j][1]);
             fb[set][j] = join8(rb[B_KC ? 0 : set][B_KC ? 0 : j][0], rb[B_KC ? 0 : set][B_KC ? 0 : j][1]); }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (A_KC) lds_tie(fa[set][i]);
      else { lds_tie(ra[A_KC ? 0 : set][A_KC ? 0 : i][0]); lds_tie(ra[A_KC ? 0 : set][A_KC ? 0 : i][1]);
             fa[set][i] = join8(ra[A_KC ? 0 : set][A_KC ? 0 : i][0], ra[A_KC ? 0 : set][A_KC ? 0 : i][1]); }
    }
  };
#ifdef LAP_GEMM_EXPERIMENTAL
  const bool no_dma = p.dbg & 1, no_mma = p.dbg & 2;
#define SP_MMA(SET, I0, I1)                                                        \
  if (!no_mma) { _Pragma("unroll") for (int i_ = I0; i_ < I1; ++i_)                \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[i_][j] = mfma16(fb[SET][j], fa[SET][i_], acc[i_][j]); }
#else
  constexpr bool no_dma = false, no_mma = false;
#define SP_MMA(SET, I0, I1)                                                        \
  _Pragma("unroll") for (int i_ = I0; i_ < I1; ++i_)                               \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) acc[i_][j] = mfma16(fb[SET][j], fa[SET][i_], acc[i_][j]);
#endif
#define SP_FENCE() __builtin_amdgcn_sched_barrier(0)

  // prologue: both buffers requested, tile kt0 awaited, its first k-half read
#pragma unroll
  for (int j = 0; j < 2 * PC; ++j) piece(0, kt0, j);
#pragma unroll
  for (int j = 0; j < 2 * PC; ++j) piece(1, kt0 + 1, j);
  wait_vmcnt<2 * PC>();
  __builtin_amdgcn_s_barrier();
  SP_FENCE();
  reads(0, smem, 0);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    const char* tA = smem + cur * STAGE;
    // A1
    lds_wait_all();                                          // set 0 (read under the previous tile's last MFMAs) is back
    tie(0);
    SP_MMA(0, 0, 2)
    SP_FENCE();
    reads(1, tA, 1);
    SP_FENCE();
    SP_MMA(0, 2, 4)
    SP_FENCE();
    // A2
    lds_wait_all();
    tie(1);
    if constexpr (!TWOB) wait_vmcnt<0>();   // TWOB: this barrier only says "tile kt is read"; the landing of kt+1 is awaited in C
    SP_FENCE();
    __builtin_amdgcn_s_barrier();
    SP_FENCE();
    // A3: the rest of set 0 with the refill of this tile's buffer (tile kt + 2) threaded through
#pragma unroll
    for (int i = 4; i < 8; ++i) {
      SP_MMA(0, i, i + 1)
      SP_FENCE();
      if (!no_dma) {
#pragma unroll
        for (int q = 0; q < PC / 2; ++q) piece(cur, kt + 2, (PC / 2) * (i - 4) + q);
      }
      SP_FENCE();
    }
    // C
    SP_MMA(1, 0, 2)
    SP_FENCE();
    if constexpr (TWOB) {   // two tiles of LDS-DMA in flight: only the older one (kt + 1) has to have landed here
      wait_vmcnt<2 * PC>();
      SP_FENCE();
      __builtin_amdgcn_s_barrier();
      SP_FENCE();
    }
    reads(0, smem + (cur ^ 1) * STAGE, 0);   // (kt + 1, k-half 0); past the end: harmless reads of a zero-filled buffer
    SP_FENCE();
    SP_MMA(1, 2, 8)
    SP_FENCE();
  }
  wait_vmcnt<0>();
  lds_wait_all();
#undef SP_MMA
#undef SP_FENCE

  if (p.epi_lds) { staged_epilogue<NW, WTM, WTN, OUT_F32>(p, smem, acc, wm, wn, m0, n0, tid, lane); return; }
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong kernel (tile 12), designed from the round-2 ablation of tile 10: a wave's instruction stream is
// serial (32 MFMAs, then 12 fragment reads + 4 LDS-DMA pieces, per 32-deep k-half), and in tile 10 the two waves of a
// SIMD do the same thing at the same time, so the matrix pipe idles whenever they load.  Here the block's waves form two
// groups (waves 0-3 / 4-7: one wave of each per SIMD) that run every interval in OPPOSITE order:
//     group 0:  barrier | L(j): read k-half j into set j&1, issue the DMA of k-half j+3, wait  | M(j):   32 MFMAs
//     group 1:  barrier | M(j-1): 32 MFMAs on the set read in the previous interval            | L(j)
// so that one wave of each SIMD multiplies while the other loads; ONE barrier per k-half, nothing else synchronises.
// LDS: a ring of FOUR 32 KiB slots, one per k-half (k-half j in slot j & 3; [A 256 x 32 | B 256 x 32], 64-byte rows with
// the kc32 swizzle, or 32 k-rows x 512 B for M-contiguous operands).  Hand-offs: a wave waits for its OWN pieces of k-half
// j (counted vmcnt: the 8 pieces of k-halves j+1, j+2 stay in flight) before barrier j, so after it the k-half is complete;
// both groups have finished reading k-half j-1 before barrier j (group 1's L(j-1) ends interval j-1), so its slot is
// refilled with k-half j+3 during interval j — three intervals (~2 us) ahead of its use instead of tile 10's < 1 k-tile.
template <bool A_KC, bool B_KC, bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_pq_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, KH = 32, HALF = BM * KH * 2, SLOT = 2 * HALF, NSLOT = 4;
  constexpr int WGN = 4, NW = 8, WTM = 128, WTN = 64, FM = 8, FN = 4, PC = 2;   // PC: 1 KiB pieces per operand, wave and k-half
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  const bool g1 = w >= 4;
  int tm, tn;
  tile_coords<4>(p, p.tile_base + xcd_remap(blockIdx.x, gridDim.x), tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.A, 0, (int)min((long long)(A_KC ? p.M : p.K) * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.B, 0, (int)min((long long)(B_KC ? p.N : p.K) * p.ldb * 2, 0x7fffffffLL), 0x00020000);
  unsigned offA[PC], offB[PC];
  int kcA[PC], kcB[PC];     // K-contiguous operands: first k of the lane's chunk inside a k-half (chunks past K read as zeros)
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int ci = (w * PC + j) * 64 + lane;     // 16-byte chunk of one operand's k-half image (1024 chunks)
    kcA[j] = 0; kcB[j] = 0;
    if (A_KC) {
      const int row = ci >> 2, c = (ci & 3) ^ ((-(row >> 2)) & 3);
      kcA[j] = c * 8;
      offA[j] = (m0 + row < p.M) ? (unsigned)(((long long)(m0 + row) * p.lda + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      offA[j] = (m0 + c * 8 < p.M) ? (unsigned)(((long long)kr * p.lda + m0 + c * 8) * 2) : OOB;
    }
    if (B_KC) {
      const int row = ci >> 2, c = (ci & 3) ^ ((-(row >> 2)) & 3);
      kcB[j] = c * 8;
      offB[j] = (n0 + row < p.N) ? (unsigned)(((long long)(n0 + row) * p.ldb + c * 8) * 2) : OOB;
    } else {
      const int kr = ci >> 5, c = (ci & 31) ^ (mc_swz(kr) << 1);
      offB[j] = (n0 + c * 8 < p.N) ? (unsigned)(((long long)kr * p.ldb + n0 + c * 8) * 2) : OOB;
    }
  }
  const unsigned stepA = A_KC ? (unsigned)(KH * 2) : (unsigned)((long long)KH * p.lda * 2);
  const unsigned stepB = B_KC ? (unsigned)(KH * 2) : (unsigned)((long long)KH * p.ldb * 2);
  const int h0 = 2 * blockIdx.y * p.ktiles_per_split;
  const int h1 = min(2 * ((p.K + 2 * KH - 1) / (2 * KH)), h0 + 2 * p.ktiles_per_split);   // k-halves [h0, h1): an even count (past K: zeros)
  auto pieces = [&](int jh) {   // all 4 pieces of this wave for k-half jh -> slot (jh - h0) & 3; past the end: zero fill
    char* base = smem + ((jh - h0) & (NSLOT - 1)) * SLOT;
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const unsigned v = (offA[j] != OOB && jh < h1 && (!A_KC || jh * KH + kcA[j] < p.K)) ? offA[j] + (unsigned)jh * stepA : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base + (w * PC + j) * 1024), 16, v, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const unsigned v = (offB[j] != OOB && jh < h1 && (!B_KC || jh * KH + kcB[j] < p.K)) ? offB[j] + (unsigned)jh * stepB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + HALF + (w * PC + j) * 1024), 16, v, 0, 0, 0);
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ONE fragment register set: a wave never multiplies and reads at the same time (that is the other wave's job), and a
  // group-1 wave re-reads the registers its MFMAs of k-half j-1 were just issued from (an MFMA takes its A / B operands in
  // its first cycles; the LDS data returns tens of cycles later)
  bf16x8 fa[1][FM], fb[1][FN];
  bf16x4 ra[1][A_KC ? 1 : FM][2], rb[1][B_KC ? 1 : FN][2];
  const int li = lane & 15, lg = lane >> 4;
  auto reads = [&](int set, const char* tA) {
    const char* tB = tA + HALF;
    if (B_KC) {   // fragments are 16 rows = 1024 bytes apart and share the swizzle
      const unsigned b0 = lds_addr_of(tB) + kc32_tile_off(wn * WTN + li, lg);
      fb[set][0] = ds_read_b128_raw<0>(b0); fb[set][1] = ds_read_b128_raw<1024>(b0);
      fb[set][2] = ds_read_b128_raw<2048>(b0); fb[set][3] = ds_read_b128_raw<3072>(b0);
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j) mc_frag_raw<BN>(tB, wn * WTN + j * 16, 0, lane, rb[0][B_KC ? 0 : j]);
    }
    if (A_KC) {
      const unsigned a0 = lds_addr_of(tA) + kc32_tile_off(wm * WTM + li, lg);
      fa[set][0] = ds_read_b128_raw<0>(a0); fa[set][1] = ds_read_b128_raw<1024>(a0);
      fa[set][2] = ds_read_b128_raw<2048>(a0); fa[set][3] = ds_read_b128_raw<3072>(a0);
      fa[set][4] = ds_read_b128_raw<4096>(a0); fa[set][5] = ds_read_b128_raw<5120>(a0);
      fa[set][6] = ds_read_b128_raw<6144>(a0); fa[set][7] = ds_read_b128_raw<7168>(a0);
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) mc_frag_raw<BM>(tA, wm * WTM + i * 16, 0, lane, ra[0][A_KC ? 0 : i]);
    }
  };
  auto tie = [&](int set) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (B_KC) lds_tie(fb[set][j]);
      else { lds_tie(rb[0][B_KC ? 0 : j][0]); lds_tie(rb[0][B_KC ? 0 : j][1]);
             fb[set][j] = join8(rb[0][B_KC ? 0 : j][0], rb[0][B_KC ? 0 : j][1]); }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      if (A_KC) lds_tie(fa[set][i]);
      else { lds_tie(ra[0][A_KC ? 0 : i][0]); lds_tie(ra[0][A_KC ? 0 : i][1]);
             fa[set][i] = join8(ra[0][A_KC ? 0 : i][0], ra[0][A_KC ? 0 : i][1]); }
    }
  };
#ifdef LAP_GEMM_EXPERIMENTAL   // ablation / variant bits (lap_gemm_set_debug): 1 no in-loop DMA, 2 no MFMA, 4 no s_setprio
  const bool no_dma = p.dbg & 1, no_mma = p.dbg & 2, no_prio = p.dbg & 4;
#else
  constexpr bool no_dma = false, no_mma = false, no_prio = false;
#endif
#define PQ_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PQ_L(SET, JH)                                                         \
  reads(SET, smem + (((JH) - h0) & (NSLOT - 1)) * SLOT);                      \
  PQ_FENCE();                                                                 \
  if (!no_dma) pieces((JH) + 3);                                              \
  PQ_FENCE();                                                                 \
  lds_wait_all();                                                             \
  tie(SET);                                                                   \
  PQ_FENCE();
#define PQ_M(SET)                                                             \
  if (!no_prio) __builtin_amdgcn_s_setprio(1);                                \
  if (!no_mma) {                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < FM; ++i_)                           \
    _Pragma("unroll") for (int j_ = 0; j_ < FN; ++j_) acc[i_][j_] = mfma16(fb[SET][j_], fa[SET][i_], acc[i_][j_]); } \
  if (!no_prio) __builtin_amdgcn_s_setprio(0);                                \
  PQ_FENCE();
#define PQ_SYNC()                                                             \
  wait_vmcnt<8>();   /* my 4 pieces of this k-half landed; the next two k-halves may fly */ \
  PQ_FENCE();                                                                 \
  __builtin_amdgcn_s_barrier();                                               \
  PQ_FENCE();

  pieces(h0); pieces(h0 + 1); pieces(h0 + 2);
  if (!g1) {      // two separate loops (same barrier count): one loop with a group branch inside made hipcc spill 450 bytes
    for (int jh = h0; jh < h1; ++jh) {
      PQ_SYNC()
      PQ_L(0, jh)
      PQ_M(0)
    }
  } else {
    PQ_SYNC()
    PQ_L(0, h0)
    for (int jh = h0 + 1; jh < h1; ++jh) {
      PQ_SYNC()
      PQ_M(0)
      PQ_L(0, jh)
    }
    PQ_M(0)
  }
  wait_vmcnt<0>();
  lds_wait_all();
#undef PQ_SYNC
#undef PQ_M
#undef PQ_L
#undef PQ_FENCE

  if (p.epi_lds) { staged_epilogue<NW, WTM, WTN, OUT_F32>(p, smem, acc, wm, wn, m0, n0, tid, lane); return; }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

#ifdef LAP_GEMM_EXPERIMENTAL   // probe: correct (bitwise), not faster than tile 10 on the forward layout
// ---------------------------------------------------------------------------------------------------------------
// Ping-pong kernel for the forward layout (tile 13: both operands K-contiguous).  Tile 12's 32-deep k-half slots would
// cut every 128-byte operand row into two 64-byte LDS-DMA requests; here the two intervals of a 64-deep k-tile split the
// wave's A FRAGMENTS instead of K, so every piece still moves full lines:
//     interval 2t   : A_lo(t) x B(t)   (A fragments 0-3, both k-steps, 32 MFMAs)      reads: 8 B + 8 A_lo fragments
//     interval 2t+1 : A_hi(t) x B(t)   (A fragments 4-7; B stays in registers)        reads: 8 A_hi fragments
// with the two wave groups in opposite order inside every interval (group 0: read, multiply; group 1: multiply what it read
// in the previous interval, read) and one barrier per interval, as in tile 12.  LDS: rings of two for each of
// A_lo (16 KiB: tile rows 0-63 and 128-191), A_hi (16 KiB: rows 64-127, 192-255) and B (32 KiB) = 128 KiB.  A_hi(t+1) is
// requested in interval 2t (its slot held A_hi(t-1), read in interval 2t-1), A_lo / B(t+2) in interval 2t+1 — three
// intervals ahead of their first use; a wave's own pieces are awaited with vmcnt(8) in both kinds of interval.
template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_pn_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, BK = 64, HALF_A = 128 * BK * 2, B_BYTES = BN * BK * 2;   // 16 KiB, 32 KiB
  constexpr int OFF_ALO = 0, OFF_AHI = 2 * HALF_A, OFF_B = 4 * HALF_A;                         // rings of two
  constexpr int WGN = 4, NW = 8, WTM = 128, WTN = 64, FM = 8, FN = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  const bool g1 = w >= 4;
  int tm, tn;
  tile_coords<4>(p, p.tile_base + xcd_remap(blockIdx.x, gridDim.x), tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)min((long long)p.M * p.lda * 2, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)min((long long)p.N * p.ldb * 2, 0x7fffffffLL), 0x00020000);
  // per-lane source offsets: A halves 2 pieces per wave each, B 4 pieces; image row r of an A half <-> tile row
  // (r & 63) + 128 (r >> 6) (+ 64 for the hi half); 8 chunks of 16 bytes per 128-byte row, XOR-swizzled on the source side
  unsigned offAlo[2], offAhi[2], offB[4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = (w * 2 + j) * 64 + lane;
    const int r = ci >> 3, c = (ci & 7) ^ ((r >> 1) & 7);
    const int trow = (r & 63) + 128 * (r >> 6);
    offAlo[j] = (m0 + trow < p.M) ? (unsigned)(((long long)(m0 + trow) * p.lda + c * 8) * 2) : OOB;
    offAhi[j] = (m0 + trow + 64 < p.M) ? (unsigned)(((long long)(m0 + trow + 64) * p.lda + c * 8) * 2) : OOB;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ci = (w * 4 + j) * 64 + lane;
    const int r = ci >> 3, c = (ci & 7) ^ ((r >> 1) & 7);
    offB[j] = (n0 + r < p.N) ? (unsigned)(((long long)(n0 + r) * p.ldb + c * 8) * 2) : OOB;
  }
  const int kt0 = blockIdx.y * p.ktiles_per_split;
  const int kt1 = min(p.K / BK, kt0 + p.ktiles_per_split);
  auto dma_alo_b = [&](int kt) {     // A_lo(kt) + B(kt): 6 pieces
    const int sl = (kt - kt0) & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned v = (offAlo[j] != OOB && kt < kt1) ? offAlo[j] + (unsigned)kt * (BK * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(smem + OFF_ALO + sl * HALF_A + (w * 2 + j) * 1024), 16, v, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned v = (offB[j] != OOB && kt < kt1) ? offB[j] + (unsigned)kt * (BK * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(smem + OFF_B + sl * B_BYTES + (w * 4 + j) * 1024), 16, v, 0, 0, 0);
    }
  };
  auto dma_ahi = [&](int kt) {       // A_hi(kt): 2 pieces
    const int sl = (kt - kt0) & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned v = (offAhi[j] != OOB && kt < kt1) ? offAhi[j] + (unsigned)kt * (BK * 2) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(smem + OFF_AHI + sl * HALF_A + (w * 2 + j) * 1024), 16, v, 0, 0, 0);
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[2][4], fb[2][FN];    // [k-step][fragment]: ONE set (see tile 12)
  const int li = lane & 15, lg = lane >> 4;
  auto read_b = [&](int sl) {
    const char* tB = smem + OFF_B + sl * B_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned b0 = lds_addr_of(tB) + kc_tile_off(wn * WTN + li, kk * 4 + lg);
      fb[kk][0] = ds_read_b128_raw<0>(b0); fb[kk][1] = ds_read_b128_raw<2048>(b0);
      fb[kk][2] = ds_read_b128_raw<4096>(b0); fb[kk][3] = ds_read_b128_raw<6144>(b0);
    }
  };
  auto read_a = [&](const char* tA) {   // the wave's four fragments of an A half: image rows wm * 64 + 16 f + li
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned a0 = lds_addr_of(tA) + kc_tile_off(wm * 64 + li, kk * 4 + lg);
      fa[kk][0] = ds_read_b128_raw<0>(a0); fa[kk][1] = ds_read_b128_raw<2048>(a0);
      fa[kk][2] = ds_read_b128_raw<4096>(a0); fa[kk][3] = ds_read_b128_raw<6144>(a0);
    }
  };
  auto tie_a = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) lds_tie(fa[kk][i]);
  };
  auto tie_b = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < FN; ++j) lds_tie(fb[kk][j]);
  };
#define PN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PN_SYNC()                                                             \
  wait_vmcnt<8>();                                                            \
  PN_FENCE();                                                                 \
  __builtin_amdgcn_s_barrier();                                               \
  PN_FENCE();
#define PN_L_EVEN(KT)   /* B(KT) and A_lo(KT) into registers; request A_hi(KT + 1) */ \
  read_b(((KT) - kt0) & 1);                                                   \
  read_a(smem + OFF_ALO + (((KT) - kt0) & 1) * HALF_A);                       \
  PN_FENCE();                                                                 \
  dma_ahi((KT) + 1);                                                          \
  PN_FENCE();                                                                 \
  lds_wait_all();                                                             \
  tie_b(); tie_a();                                                           \
  PN_FENCE();
#define PN_L_ODD(KT)    /* A_hi(KT) into registers; request A_lo / B(KT + 2) */ \
  read_a(smem + OFF_AHI + (((KT) - kt0) & 1) * HALF_A);                       \
  PN_FENCE();                                                                 \
  dma_alo_b((KT) + 2);                                                        \
  PN_FENCE();                                                                 \
  lds_wait_all();                                                             \
  tie_a();                                                                    \
  PN_FENCE();
#define PN_M(I0)                                                              \
  __builtin_amdgcn_s_setprio(1);                                              \
  _Pragma("unroll") for (int kk_ = 0; kk_ < 2; ++kk_)                         \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                          \
      _Pragma("unroll") for (int j_ = 0; j_ < FN; ++j_)                       \
        acc[(I0) + i_][j_] = mfma16(fb[kk_][j_], fa[kk_][i_], acc[(I0) + i_][j_]); \
  __builtin_amdgcn_s_setprio(0);                                              \
  PN_FENCE();

  // prologue: the steady-state request order from its start — A_lo/B(0), A_hi(0), A_lo/B(1); interval 0 then asks for A_hi(1)
  dma_alo_b(kt0); dma_ahi(kt0); dma_alo_b(kt0 + 1);
  if (!g1) {
    for (int kt = kt0; kt < kt1; ++kt) {
      PN_SYNC()
      PN_L_EVEN(kt)
      PN_M(0)
      PN_SYNC()
      PN_L_ODD(kt)
      PN_M(4)
    }
  } else {
    PN_SYNC()
    PN_L_EVEN(kt0)
    for (int kt = kt0; kt < kt1; ++kt) {
      PN_SYNC()
      PN_M(0)
      PN_L_ODD(kt)
      if (kt + 1 < kt1) {
        PN_SYNC()
        PN_M(4)
        PN_L_EVEN(kt + 1)
      }
    }
    PN_M(4)
  }
  wait_vmcnt<0>();
  lds_wait_all();
#undef PN_M
#undef PN_L_ODD
#undef PN_L_EVEN
#undef PN_SYNC
#undef PN_FENCE

  if (p.epi_lds) { staged_epilogue<NW, WTM, WTN, OUT_F32>(p, smem, acc, wm, wn, m0, n0, tid, lane); return; }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(p, m, n, acc[i][j]);
    }
  }
}

template <bool OUT_F32>
int launch_pn(GemmParams p, hipStream_t s) {
  constexpr int LDS = OUT_F32 ? 128 * (256 * 4 + 16) : 256 * (256 * 2 + 16);
  if (p.K & 63) return LAP_ERR_ARG;
  auto kern = gemm_pn_kernel<OUT_F32>;
  p.epi_lds = (!p.part && p.ksplit == 1 && !(OUT_F32 && p.R) && !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) ? epi_lds_mode() : 0;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int nkt = p.K / 64;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base;
  hipLaunchKernelGGL(kern, dim3(count, p.ksplit), dim3(512), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

#endif  // LAP_GEMM_EXPERIMENTAL

template <bool A_KC, bool B_KC, bool OUT_F32>
int launch_pq(GemmParams p, hipStream_t s) {
  constexpr int LDS = OUT_F32 ? 128 * (256 * 4 + 16) : 256 * (256 * 2 + 16);   // >= the four 32 KiB slots
  if (p.K & 7) return LAP_ERR_ARG;      // (a ragged last k-tile is zero-filled by the kernel)
  auto kern = gemm_pq_kernel<A_KC, B_KC, OUT_F32>;
  p.epi_lds = (!p.part && p.ksplit == 1 && !(OUT_F32 && p.R) && !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) ? epi_lds_mode() : 0;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int nkt = (p.K + 63) / 64;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base;
  hipLaunchKernelGGL(kern, dim3(count, p.ksplit), dim3(512), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <int WGM, int WGN, bool A_KC, bool B_KC, bool OUT_F32, bool TWOB = false>
int launch_sp(GemmParams p, hipStream_t s) {
  constexpr int LDS = OUT_F32 ? 128 * (256 * 4 + 16) : 256 * (256 * 2 + 16);   // >= the two operand stages (128 KiB)
  if (p.K & 7) return LAP_ERR_ARG;      // (a ragged last k-tile is zero-filled by the kernel)
  auto kern = gemm_sp_kernel<WGM, WGN, A_KC, B_KC, OUT_F32, TWOB>;
  p.epi_lds = (!p.part && p.ksplit == 1 && !(OUT_F32 && p.R) && !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) ? epi_lds_mode() : 0;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int nkt = (p.K + 63) / 64;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base;
  hipLaunchKernelGGL(kern, dim3(count, p.ksplit), dim3(WGM * WGN * 64), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

#ifdef LAP_GEMM_EXPERIMENTAL
template <bool A_KC, bool B_KC, bool OUT_F32>
int launch_pp16(GemmParams p, hipStream_t s) {
  constexpr int LDS = 4 * 2 * 256 * 32 * 2;
  auto kern = gemm_pp16_kernel<A_KC, B_KC, OUT_F32>;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int nkt = (p.K + 63) / 64;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base;
  hipLaunchKernelGGL(kern, dim3(count, p.ksplit), dim3(1024), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <bool A_KC, bool B_KC, bool OUT_F32>
int launch_pp(GemmParams p, hipStream_t s) {
  constexpr int LDS = 2 * (256 + 256) * 64 * 2;
  auto kern = gemm_pp_kernel<A_KC, B_KC, OUT_F32>;
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    done = true;
  }
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int nkt = (p.K + 63) / 64;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base;
  hipLaunchKernelGGL(kern, dim3(count, p.ksplit), dim3(512), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
#endif  // LAP_GEMM_EXPERIMENTAL

// out = epilogue(alpha * sum_s part[s]) for the two-phase split-K path.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
  const long long gid = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (gid >= (long long)p.M * p.N) return;
  const int m = (int)(gid / p.N), n = (int)(gid % p.N);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < p.ksplit; ++sp) v += *reinterpret_cast<const f32x4*>(p.part + (long long)sp * p.M * p.N + gid);
  v *= p.alpha;
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
  if (OUT_F32) {
    float* c = (float*)p.C + (long long)m * p.ldc + n;
    if (p.accum) v += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = v;
  } else {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x4*>((bf16*)p.C + (long long)m * p.ldc + n) = o;
  }
}

template <int BM, int BN, int WGM, int WGN, int BK, int NS, bool A_KC, bool B_KC, bool OUT_F32>
int launch(GemmParams p, hipStream_t s) {
  // staged epilogue of the 256x256 kernel: a bf16 tile (544-byte rows) or half an f32 tile (128 rows of 1088 bytes)
  constexpr int EPI = (BM == 256 && BN == 256 && WGM * WGN == 16) ? (OUT_F32 ? 128 * (BN * 4 + 16) : BM * (BN * 2 + 16)) : 0;
  constexpr int LDS = NS * (BM + BN) * BK * 2 > EPI ? NS * (BM + BN) * BK * 2 : EPI;
  auto kern = gemm_kernel<BM, BN, WGM, WGN, BK, NS, A_KC, B_KC, OUT_F32>;
  p.epi_lds = (EPI > 0 && !p.part && p.ksplit == 1 && !(OUT_F32 && p.R) && !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) ? epi_lds_mode() : 0;
  if (LDS > 65536) {
    static bool done = false;  // benign race: the attribute is idempotent
    if (!done) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
  }
  const bool sub = BM == 128 && BN == 128 && p.sub256;
  p.tiles_m = sub ? (p.M + 255) / 256 : (p.M + BM - 1) / BM;
  p.tiles_n = sub ? (p.N + 255) / 256 : (p.N + BN - 1) / BN;
  const int nkt = (p.K + BK - 1) / BK;
  p.ktiles_per_split = (nkt + p.ksplit - 1) / p.ksplit;
  const int count = sub ? 4 * p.tile_count : (p.tile_count > 0 ? p.tile_count : p.tiles_m * p.tiles_n - p.tile_base);
  dim3 grid(count, p.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(WGM * WGN * 64), LDS, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <bool A_KC, bool B_KC, bool OUT_F32>
int dispatch_tile(const GemmParams& p, int tile, hipStream_t s) {
  switch (tile) {
    // production: 10 (software-pipelined 8-wave 256x256, K % 64 == 0), 5 (16-wave 256x256), 6 (128x128); 2 is the
    // direct-epilogue 8-wave kernel the bitwise tests compare against; 0 (default below) the 4-wave 128x128 kernel
    case 12: return launch_pq<A_KC, B_KC, OUT_F32>(p, s);
    case 10: return launch_sp<2, 4, A_KC, B_KC, OUT_F32>(p, s);
    case 6: return launch<128, 128, 2, 4, 64, 2, A_KC, B_KC, OUT_F32>(p, s);
    // serving-prefill shapes (M = 512 / 560 rows, forward layout, bf16 out): see pick_serving_tile
    case 15: if constexpr (A_KC && B_KC) return launch<320, 256, 2, 4, 64, 2, true, true, OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 16: if constexpr (A_KC && B_KC) return launch<64, 128, 2, 4, 64, 3, true, true, OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 17: if constexpr (A_KC && B_KC) return launch<64, 64, 2, 2, 64, 4, true, true, OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 18: if constexpr (A_KC && B_KC) return launch<128, 64, 4, 2, 64, 3, true, true, OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 19: if constexpr (A_KC && B_KC) return launch<320, 128, 2, 4, 64, 2, true, true, OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 5: return launch<256, 256, 4, 4, 64, 2, A_KC, B_KC, OUT_F32>(p, s);
    case 2: return launch<256, 256, 2, 4, 64, 2, A_KC, B_KC, OUT_F32>(p, s);
#ifdef LAP_GEMM_EXPERIMENTAL   // probes kept for the record: build with LAP_GEMM_EXPERIMENTAL=1 python -m lap_amd.build
    case 13: if constexpr (A_KC && B_KC) return launch_pn<OUT_F32>(p, s); else return LAP_ERR_ARG;
    case 11: return launch_sp<2, 4, A_KC, B_KC, OUT_F32, true>(p, s);   // tile 10 with two barriers per k-tile (probe)
    case 9: return launch_pp16<A_KC, B_KC, OUT_F32>(p, s);
    case 8: return launch_pp<A_KC, B_KC, OUT_F32>(p, s);
    case 7: return launch<256, 128, 4, 4, 64, 2, A_KC, B_KC, OUT_F32>(p, s);
    case 4: return launch<128, 128, 2, 2, 32, 4, A_KC, B_KC, OUT_F32>(p, s);
    case 3: return launch<256, 256, 2, 4, 32, 4, A_KC, B_KC, OUT_F32>(p, s);
    case 1: return launch<256, 128, 4, 2, 64, 3, A_KC, B_KC, OUT_F32>(p, s);
#else
    case 13: case 11: case 9: case 8: case 7: case 4: case 3: case 1: return LAP_ERR_ARG;   // not in this build
#endif
    default: return launch<128, 128, 2, 2, 64, 2, A_KC, B_KC, OUT_F32>(p, s);
  }
}

// Reduce + epilogue for the TAIL split of the 256x256 kernel: slabs [ksplit][count][256][256], slot s = logical tile
// tile_base + s.  64 blocks per tile, 4 outputs per thread.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void splitk_tail_reduce_kernel(GemmParams p, int count) {
  const int slot = blockIdx.x >> 6;
  const int within = ((blockIdx.x & 63) * 256 + threadIdx.x) * 4;
  int tm, tn;
  tile_coords<4>(p, p.tile_base + slot, tm, tn);
  const int m = tm * 256 + (within >> 8), n = tn * 256 + (within & 255);
  if (m >= p.M || n >= p.N) return;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < p.ksplit; ++sp) v += *reinterpret_cast<const f32x4*>(p.part + (((long long)sp * count + slot) << 16) + within);
  v *= p.alpha;
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
  if (OUT_F32) {
    float* c = (float*)p.C + (long long)m * p.ldc + n;
    if (p.accum) v += *reinterpret_cast<const f32x4*>(c);
    *reinterpret_cast<f32x4*>(c) = v;
  } else {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x4*>((bf16*)p.C + (long long)m * p.ldc + n) = o;
  }
}

// Tile heuristic between the two production shapes (measured on MI355X, tools/bench_kernels.py):
//   tile 5 = 256x256, 16 waves, 1 block/CU  — 1.0-1.3 PF when its rounds over the 256 CUs are well filled;
//   tile 6 = 128x128,  8 waves, 2 blocks/CU — 0.9-1.0 PF, finer quantisation, better for short K / few tiles.
// score = round-fill efficiency x relative kernel efficiency (which grows with K for the big tile).
int pick_tile(int M, int N, int K) {
  const long long t5 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  const long long t6 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const double fill5 = (double)t5 / (256.0 * ((t5 + 255) / 256));
  const double fill6 = (double)t6 / (512.0 * ((t6 + 511) / 512));
  const double eff5 = 1.15 + 0.13 * (K >= 8192 ? 1.0 : K / 8192.0);
  if (t6 <= 256) return 6;  // not even one round of small tiles: finest granularity (+ split-K) wins
  return (fill5 * eff5 > fill6) ? 5 : 6;
}

// Automatic two-phase split-K (only when the caller lends scratch): fills the last round of a poorly filled
// 256x256 grid, or spreads a GEMM with a handful of output tiles (skinny-M serving, small weights) over the chip.
int pick_ksplit(int tile, int M, int N, int K, long long scratch_bytes, bool fwd_layout) {
  const long long cap = scratch_bytes / ((long long)M * N * 4);
  if (cap < 2) return 1;
  if (tile == 5 || tile == 8) {
    const long long t5 = (long long)((M + 255) / 256) * ((N + 255) / 256);
    const double fill1 = (double)t5 / (256.0 * ((t5 + 255) / 256));
    if (fill1 >= 0.8 || K < 4096) return 1;
    int best = 1;
    double score = fill1;
    for (int sp = 2; sp <= 4 && sp <= cap; ++sp) {
      const long long w = t5 * sp;
      const double sc = (double)w / (256.0 * ((w + 255) / 256)) - 0.02 * sp;
      if (sc > score) { score = sc; best = sp; }
    }
    return best;
  }
  if (tile == 6 || tile == 0) {
    const long long t6 = (long long)((M + 127) / 128) * ((N + 127) / 128);
    // windows tuned per regime (tools/gemm_sweep.py): the serving prefill (forward layout, M <= 1024) splits up to 256
    // tiles; the train step's action-expert GEMMs (M = 1600: 104 tiles walking K = 2048 .. 8192) up to 128
    const bool serving = fwd_layout && M <= 1024;
    if (t6 > (serving ? 256 : 128) || K < 1024) return 1;
    // Few tiles (serving prefill at M ~ 512, small weights): a 128x128 block that walks all of K loads 512 * K bytes through
    // ONE CU's vector-memory path (~45 GB/s, tools/bench_skinny.py) — 22 us at K = 2048 whatever the MFMA rate.  Splitting K
    // until the chip holds two blocks per CU shortens that chain; the reduce pass costs ~5 us + the slab traffic.
    long long sp = (t6 > 96 ? 512 : 256) / t6;
    if (sp > K / 256) sp = K / 256;
    if (!serving && t6 > 96 && sp > K / 1024) sp = K / 1024;
    if (sp > 16) sp = 16;
    if (sp > cap) sp = cap;
    return sp < 2 ? 1 : (int)sp;
  }
  return 1;
}

}  // namespace

static int g_gemm_dbg = 0;

extern "C" int lap_gemm_set_debug(int bits) {   // ablation knob; has an effect in LAP_GEMM_EXPERIMENTAL builds only
#ifdef LAP_GEMM_EXPERIMENTAL
  g_gemm_dbg = bits;
  return LAP_OK;
#else
  return bits ? LAP_ERR_ARG : LAP_OK;
#endif
}

// Weight gradient dW [M][N] (f32) = A^T B over K rows (A [K][M], B [K][N]) with its sum of squares folded in where the assembly
// kernel takes the product (*folded = 1: *sumsq has received sum(dW^2)); every other shape runs as lap_gemm_bf16_ex would run it and
// leaves the norm to the caller (*folded = 0).  The routing rule is the plain-product rule of lap_gemm_bf16_ex.
extern "C" int lap_gemm_wgrad_f32(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, float* sumsq,
                                  int* folded, void* scratch, long long scratch_bytes, void* stream) {
  if (!folded) return LAP_ERR_ARG;
  *folded = 0;
  static const bool no_asm = getenv("LAP_GEMM_NO_ASM") != nullptr;
  if (sumsq && !no_asm && M > 0 && N > 0 && lap_gemm_asm_ok(0, 0, 1, M, N, K, lda, ldb, ldc)) {
    const long long t5 = (long long)(M / 256) * (N / 256);
    const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
    if (t5 >= 128 && fill >= 0.8) {
      *folded = 1;
      return lap_gemm_asm_wgrad(A, B, C, M, N, K, lda, ldb, ldc, sumsq, stream);
    }
  }
  return lap_gemm_bf16_ex(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 1.0f, 0, 0, LAP_GEMM_OUT_F32, -1, 0, scratch, scratch_bytes, stream);
}

// The same for a weight gradient stored as bf16 (dW [M][N] bf16; ParamStore.grad_dtype): the bf16 assembly kernels fold the squares of
// their f32 accumulators.
extern "C" int lap_gemm_wgrad_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, float* sumsq,
                                   int* folded, void* scratch, long long scratch_bytes, void* stream) {
  if (!folded) return LAP_ERR_ARG;
  *folded = 0;
  static const bool no_asm = getenv("LAP_GEMM_NO_ASM") != nullptr;
  if (sumsq && !no_asm && M > 0 && N > 0 && lap_gemm_asm_ok(0, 0, 0, M, N, K, lda, ldb, ldc)) {
    const long long t5 = (long long)(M / 256) * (N / 256);
    const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
    if (t5 >= 128 && fill >= 0.8) {
      *folded = 1;
      return lap_gemm_asm_wgrad_b16(A, B, C, M, N, K, lda, ldb, ldc, sumsq, stream);
    }
  }
  return lap_gemm_bf16_ex(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 1.0f, 0, 0, 0, -1, 0, scratch, scratch_bytes, stream);
}

extern "C" int lap_gemm_bf16_ex(const void* A, const void* B, void* C, const void* bias, const void* residual,
                                int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                                int a_kc, int b_kc, int flags, int tile, int ksplit, void* scratch,
                                long long scratch_bytes, void* stream) {
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
  const bool f32 = flags & LAP_GEMM_OUT_F32;
  if ((flags & LAP_GEMM_ACCUM) && !f32) return LAP_ERR_ARG;
  if ((flags & LAP_GEMM_GELU) && f32) return LAP_ERR_ARG;
  if (tile < -1 || tile > 19 || ksplit < 0) return LAP_ERR_ARG;
  if (flags & LAP_GEMM_GEGLU) {   // gate|up projection + GeGLU in one launch (serving prefill): the 320-row tile only
    if (!a_kc || !b_kc || f32 || bias || residual || (flags & ~(LAP_GEMM_GEGLU | LAP_GEMM_GELU_EXP2)) || (N & 255) || M > 640 || alpha != 1.0f || ksplit > 1 || (tile >= 0 && tile != 15))
      return LAP_ERR_ARG;
    tile = 15; ksplit = 1;
  }
  // Few output tiles but a very long contraction (LM-head dgrad: 1504 x 2048 over K = 257152; prefill down
  // projection): the big tile with enough K splits to cover the chip beats the small tile walking all of K.
  if (tile < 0 && ksplit == 0 && scratch != nullptr && K >= 16384 && !(flags & LAP_GEMM_PARTIALS)) {
    const long long t5 = (long long)((M + 255) / 256) * ((N + 255) / 256);
    if (t5 <= 128) {
      long long sp = 256 / t5;
      const long long cap = scratch_bytes / ((long long)M * N * 4);
      if (sp > 8) sp = 8;
      if (sp > (K + 63) / 64 / 16) sp = (K + 63) / 64 / 16;
      if (sp > cap) sp = cap;
      if (sp >= 2) { tile = 5; ksplit = (int)sp; }
    }
  }
  // Plain products over whole 256-tiles (no bias / residual / GELU / accumulate / split): the hand-scheduled assembly main
  // loop (csrc/gemm_asm_kernels.s: forward bf16, data-gradient bf16, weight-gradient f32 layouts; same accumulation order,
  // bitwise equal) whenever its persistent rounds are well filled, or the contraction is too short for the tail split
  // below to pay.  tile 14 forces it (tests); LAP_GEMM_NO_ASM=1 disables it (A/B runs).
  {
    static const bool no_asm = getenv("LAP_GEMM_NO_ASM") != nullptr;
    static const bool no_asm_nn = getenv("LAP_GEMM_NO_ASM_NN") != nullptr;
    const bool plain = !bias && !residual && !(flags & (LAP_GEMM_GELU | LAP_GEMM_ACCUM | LAP_GEMM_PARTIALS)) && alpha == 1.0f && ksplit <= 1 &&
                       lap_gemm_asm_ok(a_kc, b_kc, f32, M, N, K, lda, ldb, ldc);
    // forward + f32 bias per column (Flax Dense of SigLIP: qkv, fc1), N any multiple of 16
    const bool biased = a_kc && b_kc && !f32 && bias && (flags & LAP_GEMM_BIAS_F32) && !residual &&
                        !(flags & (LAP_GEMM_GELU | LAP_GEMM_ACCUM | LAP_GEMM_PARTIALS)) && alpha == 1.0f && ksplit <= 1 &&
                        lap_gemm_asm_bias_ok(M, N, K, lda, ldb, ldc) && !((uintptr_t)bias & 15);
    // forward + bf16 residual with C's leading dimension (+ optional f32 bias): out / down projections of a block
    const bool resid = a_kc && b_kc && !f32 && residual && ldr == ldc && !((uintptr_t)residual & 15) &&
                       (!bias || ((flags & LAP_GEMM_BIAS_F32) && !((uintptr_t)bias & 15))) &&
                       !(flags & (LAP_GEMM_GELU | LAP_GEMM_ACCUM | LAP_GEMM_PARTIALS)) && alpha == 1.0f && ksplit <= 1 &&
                       lap_gemm_asm_res_ok(bias != nullptr, M, N, K, lda, ldb, ldc);
    if (tile == 14 && resid) return lap_gemm_asm_res(A, B, C, (const float*)bias, residual, M, N, K, lda, ldb, ldc, stream);
    if (tile < 0 && resid && !no_asm) {
      static const bool no_res = getenv("LAP_GEMM_NO_ASM_RES") != nullptr;      // A/B switch
      const long long tm = M / 256, tn = (N + 255) / 256, t5 = tm * tn, rounds = t5 / 256;
      const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
      if (!no_res && t5 >= 128 && (fill >= 0.8 || (bias && K <= 2048))) return lap_gemm_asm_res(A, B, C, (const float*)bias, residual, M, N, K, lda, ldb, ldc, stream);
      // (the M cut of the plain products below, with the residual rows following the cut)
      static const bool no_msplit = getenv("LAP_GEMM_NO_MSPLIT") != nullptr;
      static const bool longk = getenv("LAP_GEMM_NO_MSPLIT_LONGK") == nullptr;   // (A/B switch, see the plain products below)
      if (!no_res && !no_msplit && !bias && ksplit == 0 && scratch != nullptr && (K <= 4096 || longk) && rounds >= 1 && fill < 0.8 && (rounds * 256) % tn == 0) {
        const int M0 = (int)(rounds * 256 / tn) * 256;
        if (int rc = lap_gemm_asm_res(A, B, C, nullptr, residual, M0, N, K, lda, ldb, ldc, stream)) return rc;
        return lap_gemm_bf16_ex((const char*)A + (long long)M0 * lda * 2, B, (char*)C + (long long)M0 * ldc * 2, nullptr,
                                (const char*)residual + (long long)M0 * ldr * 2, M - M0, N, K, lda, ldb, ldc, ldr, 1.0f, a_kc, b_kc, flags, -1, 0,
                                scratch, scratch_bytes, stream);
      }
    }
    if (tile == 14 && biased) return lap_gemm_asm_bias(A, B, C, (const float*)bias, M, N, K, lda, ldb, ldc, stream);
    if (tile < 0 && biased && !no_asm) {
      const long long t5 = (long long)(M / 256) * ((N + 255) / 256);
      const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
      if (t5 >= 128 && fill >= 0.8) return lap_gemm_asm_bias(A, B, C, (const float*)bias, M, N, K, lda, ldb, ldc, stream);
    }
    if (tile == 14) return plain ? lap_gemm_asm(A, B, C, M, N, K, lda, ldb, ldc, a_kc, b_kc, f32, stream) : LAP_ERR_ARG;
    // M = 17,920 rows x N = 2,048 columns are 70 x 8 = 560 tiles: 2.19 rounds of the chip.  The persistent kernel would run a
    // third round for 48 tiles; the HIP tile splits those 48 along K.  Same cut here, made along M: rows [0, 64 x 256) are
    // exactly two rounds for the assembly kernel, the last 1,536 rows a product of their own on the automatic route (its
    // K split covers the chip).  tools/bench_msplit.py (isolated, us): qkv data gradient K = 2560 195 -> 152, out data gradient
    // K = 2048 161 -> 124, plain forward K = 2048 129 -> 121.  LAP_GEMM_NO_MSPLIT=1: off (A/B).
    // Long contractions with a K-contiguous A (gate|up data gradient K = 32768, down forward + residual K = 16384; the engine pads
    // A's rows off the 16 KiB stride): isolated the cut is a wash (1726 vs 1762 us, 922 vs 949 us: the last 1536 rows cost 181 /
    // 100 us on the HIP tile's K split either way), in the train step it is worth 2.8 ms (302.0 -> 299.2 ms, interleaved on
    // one box) — the assembly kernel's two rounds leave the optimizer stream more of the chip than the HIP tile's.
    // LAP_GEMM_NO_MSPLIT_LONGK=1: off (A/B).
    static const bool msplit_longk = getenv("LAP_GEMM_NO_MSPLIT_LONGK") == nullptr;
    if (tile < 0 && plain && !no_asm && ksplit == 0 && scratch != nullptr && (K <= 4096 || (msplit_longk && a_kc))) {
      static const bool off = getenv("LAP_GEMM_NO_MSPLIT") != nullptr;
      const long long tm = M / 256, tn = N / 256, t5 = tm * tn, rounds = t5 / 256;
      const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
      if (!off && rounds >= 1 && fill < 0.8 && (rounds * 256) % tn == 0) {
        const int M0 = (int)(rounds * 256 / tn) * 256;
        const int esz = f32 ? 4 : 2;
        const void* A1 = a_kc ? (const void*)((const char*)A + (long long)M0 * lda * 2) : (const void*)((const char*)A + (long long)M0 * 2);
        void* C1 = (void*)((char*)C + (long long)M0 * ldc * esz);
        if (int rc = lap_gemm_asm(A, B, C, M0, N, K, lda, ldb, ldc, a_kc, b_kc, f32, stream)) return rc;
        return lap_gemm_bf16_ex(A1, B, C1, nullptr, nullptr, M - M0, N, K, lda, ldb, ldc, 0, 1.0f, a_kc, b_kc, flags, -1, 0, scratch, scratch_bytes, stream);
      }
    }
    // Ragged M with very many rows (the embedding table's weight gradient: 257,152 = 1004.5 x 256 rows): the whole m-tiles on the
    // assembly kernel, the last M % 256 rows as a product of their own (weight-gradient layout only: A's tail is a column offset)
    if (tile < 0 && !no_asm && !a_kc && !b_kc && f32 && (M & 255) && M >= 65536 && !bias && !residual &&
        !(flags & (LAP_GEMM_GELU | LAP_GEMM_ACCUM | LAP_GEMM_PARTIALS)) && alpha == 1.0f && ksplit == 0) {
      const int M0 = M & ~255;
      if (lap_gemm_asm_ok(0, 0, 1, M0, N, K, lda, ldb, ldc)) {
        if (int rc = lap_gemm_asm(A, B, C, M0, N, K, lda, ldb, ldc, 0, 0, 1, stream)) return rc;
        return lap_gemm_bf16_ex((const char*)A + (long long)M0 * 2, B, (char*)C + (long long)M0 * ldc * 4, nullptr, nullptr, M - M0, N, K, lda, ldb,
                                ldc, 0, 1.0f, 0, 0, flags, -1, 0, scratch, scratch_bytes, stream);
      }
    }
    if (tile < 0 && plain && !no_asm) {
      const long long t5 = (long long)(M / 256) * (N / 256);
      const double fill = (double)t5 / (256.0 * ((t5 + 255) / 256));
      // measured against the HIP tiles (tools/bench_asm_gemm.py bench): forward +10-17 % whenever the rounds are filled or the
      // contraction is short; weight gradient +10 % (tall outputs run as the wide product of the swapped operands with
      // transposed stores, see lap_gemm_asm); data gradient: 0 .. +14 % on short contractions with filled rounds (the
      // ping-pong tile's rate on a weight with 32 KiB rows varies from box to box), long ones keep the tail split
      const bool win = (a_kc && b_kc) ? (fill >= 0.8 || K <= 4096) : (!a_kc && !b_kc) ? fill >= 0.8 : (fill >= 0.8 && K <= 4096 && !no_asm_nn);
      if (t5 >= 128 && win) return lap_gemm_asm(A, B, C, M, N, K, lda, ldb, ldc, a_kc, b_kc, f32, stream);
    }
  }
  // N = 256 j + 128 with many rows (SigLIP's width 1152 at B x 512 rows: out / fc2 forward, qkv / fc1 data gradients): the 256-wide
  // tiling needs j + 1 column tiles of which the last is half empty, e.g. 64 x 4.5 -> 320 tile slots = two rounds of the chip for
  // 1.13 rounds of work.  Cut the product along N instead: columns [0, 256 j) are whole tiles (64 x 4 = exactly one round at
  // B = 32, and a plain / bias-only product of that shape is eligible for the assembly kernels), the last 128 columns a second,
  // small product on the 128 x 128 tile (with its automatic K split: those 128 columns see another f32 summation order,
  // everything else keeps its bits).
  // tools/bench_nsplit.py (isolated, us): fc2 forward K = 4304 240 -> 200, qkv data gradient K = 3456 141 -> 122, fc1 data gradient
  // K = 4304 172 -> 159; K = 1152 (out projection) loses 4 us to the second launch, so short contractions stay whole.  The tail
  // product re-reads all of A for 128 columns, which is what keeps the gain below the 1.5 / 2 rounds it removes.
  // LAP_GEMM_NO_NSPLIT=1: off (A/B).
  if (tile < 0 && ksplit == 0 && (N & 255) == 128 && N >= 640 && N <= 2048 + 128 && M >= 4096 && K >= 2048 && !(flags & LAP_GEMM_PARTIALS)) {
    static const bool off = getenv("LAP_GEMM_NO_NSPLIT") != nullptr;
    const long long tm = (M + 255) / 256, t_all = tm * (N / 256 + 1), t_whole = tm * (N / 256);
    const double fill_all = (double)t_all / (256.0 * ((t_all + 255) / 256)), fill_whole = (double)t_whole / (256.0 * ((t_whole + 255) / 256));
    if (!off && fill_all < 0.8 && fill_whole >= 0.9) {
      const int N0 = N - 128;
      const int esz = f32 ? 4 : 2;
      const char* Bp = (const char*)B;
      const void* B1 = b_kc ? (const void*)(Bp + (long long)N0 * ldb * 2) : (const void*)(Bp + (long long)N0 * 2);
      const void* bias1 = bias ? (const void*)((const char*)bias + (long long)N0 * ((flags & LAP_GEMM_BIAS_F32) ? 4 : 2)) : nullptr;
      const void* res1 = residual ? (const void*)((const char*)residual + (long long)N0 * 2) : nullptr;
      void* C1 = (void*)((char*)C + (long long)N0 * esz);
      if (int rc = lap_gemm_bf16_ex(A, B, C, bias, residual, M, N0, K, lda, ldb, ldc, ldr, alpha, a_kc, b_kc, flags, -1, 0, scratch, scratch_bytes, stream)) return rc;
      return lap_gemm_bf16_ex(A, B1, C1, bias1, res1, M, 128, K, lda, ldb, ldc, ldr, alpha, a_kc, b_kc, flags, -1, 0, scratch, scratch_bytes, stream);
    }
  }
  // Serving prefill (batch-1 action chunk: 512 SigLIP rows, 560 Gemma rows; forward layout, bf16 out).  Every block of such
  // a GEMM is bound by what it pulls through its CU's vector-memory path (~45 GB/s), so the tile is the one with the fewest
  // operand bytes per block that still covers the chip WITHOUT a split-K reduce pass behind it (tools/bench_prefill_gemm.py,
  // hipGraph-timed, automatic choice -> here): SigLIP qkv 22.8 -> 12.5 us, out 14.7 -> 10.1, fc1 24.5 -> 16.8, head 19.3 ->
  // 10.1; Gemma qkv 22.2 -> 17.1, out 18.2 -> 17.4, gate|up 90.8 -> 74.2 (a 320-row tile: 560 rows are 2 x 280, not 3 x 256).
  // Long contractions (SigLIP fc2, Gemma down: K >= 4096) keep their K split.  LAP_GEMM_NO_SERVING_TILES=1: off (A/B).
  if (tile < 0 && ksplit == 0 && a_kc && b_kc && M > 256 && M <= 768 && !(flags & LAP_GEMM_PARTIALS) && (f32 || !(flags & LAP_GEMM_ACCUM))) {
    static const bool off = getenv("LAP_GEMM_NO_SERVING_TILES") != nullptr;
    if (!off) {
      if (K <= 1536 && N >= 1024) { tile = N >= 3072 ? 16 : 17; ksplit = 1; }     // (also the f32 hi / lo stem products)
      else if (f32) {}
      else if (K <= 2560 && N >= 8192 && M > 512 && M <= 640) { tile = 15; ksplit = 1; }
      else if (K <= 2560 && N >= 2048 && N < 8192) { tile = 16; ksplit = 1; }
    }
  }
  if (tile < 0) tile = pick_tile(M, N, K);
  // Tail split (256x256 kernel, automatic split only): the full rounds of 256 tiles run unsplit; only the tiles of
  // the last, poorly filled round are split along K so that they fill the chip for 1/sp of a round.
  int tail_tiles = 0, tail_sp = 0;
  if (ksplit == 0 && scratch != nullptr && tile == 5 && !(flags & LAP_GEMM_PARTIALS)) {
    const long long t5 = (long long)((M + 255) / 256) * ((N + 255) / 256);
    const int tail = (int)(t5 % 256), nkt = (K + 63) / 64;
    if (t5 > 256 && tail > 0 && tail < 200) {
      // Cost model (microseconds, calibrated on in-situ traces): a tile costs ~6 + 1.9 per 64-deep k-tile; splitting the
      // tail sp ways shortens that to nkt / sp k-tiles but adds a reduce pass over sp f32 slabs + the output
      // (~25 us of launch + latency + bytes at ~2.5 TB/s).  Short contractions (K <= 2048) are cheaper unsplit
      // (measured: tools/bench_tail.py).
      int smax = 256 / tail;
      if (smax > 8) smax = 8;
      const long long cap = scratch_bytes / ((long long)tail * 65536 * 4);
      if (smax > cap) smax = (int)cap;
      const double whole = 6.0 + 1.9 * nkt;
      double best = whole;
      int sp = 1;
      for (int c = 2; c <= smax; ++c) {
        const double cost = 6.0 + 1.9 * ((nkt + c - 1) / c) + 25.0 + (double)tail * 65536.0 * (4.0 * c + 2.0) / 2.5e6;
        if (cost < best - 4.0) { best = cost; sp = c; }   // (a split has to pay for its extra launch clearly)
      }
      // third option, for short contractions: the tail as 4 x tail quadrants on the 128x128 kernel (one round of it when
      // tail <= 128), modelled as 0.64 of a 256x256 tile time + its launch.  Measured gain is small: 4-9 us per GEMM
      // isolated (tools/bench_tail.py), 345.0 -> 344.1 ms per train step in an interleaved A/B (within noise)
      if (tail <= 128 && 0.64 * whole + 6.0 < best - 4.0) { tail_tiles = tail; tail_sp = 1; }
      else if (sp >= 2) { tail_tiles = tail; tail_sp = sp; }
    }
  }
  if (ksplit == 0 && scratch != nullptr && !tail_tiles) ksplit = pick_ksplit(tile, M, N, K, scratch_bytes, a_kc && b_kc);
  // K % 8 == 0: the software-pipelined 8-wave kernel (tile 10) for the forward layout, the ping-pong kernel (tile 12) as
  // soon as an operand is M-contiguous (data / weight gradients: +5-11 % measured, tools/bench_kernels.py; on the forward
  // layout its 64-byte k-half rows cost more in LDS-DMA requests than the ping-pong gains); else the 16-wave kernel
  static const bool no_pq = getenv("LAP_GEMM_NO_PINGPONG") != nullptr;   // A/B switch for benchmarks
  static const bool no_ktail = getenv("LAP_GEMM_NO_KTAIL") != nullptr;      // A/B switch: ragged K back on the lockstep tile
  const int big = (!(K & 7) && (!(K & 63) || !no_ktail)) ? ((a_kc && b_kc) || no_pq ? 10 : 12) : 5;
  if (tile == 5) tile = big;
  const bool two_phase = ksplit > 1 && scratch != nullptr;
  if (two_phase && scratch_bytes < (long long)ksplit * M * N * 4) return LAP_ERR_ARG;
  if (ksplit > 1 && !two_phase && (!f32 || !(flags & LAP_GEMM_ACCUM) || bias || residual)) return LAP_ERR_ARG;
  GemmParams p = {};
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.bias = bias; p.R = (const bf16*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.alpha = alpha;
  p.bias_kind = bias ? ((flags & LAP_GEMM_BIAS_F32) ? 2 : 1) : 0;
  p.gelu = (flags & LAP_GEMM_GELU) ? ((flags & LAP_GEMM_GELU_BF16) ? 2 : 1) : 0;
  p.accum = (flags & LAP_GEMM_ACCUM) ? 1 : 0;
  p.geglu = (flags & LAP_GEMM_GEGLU) ? ((flags & LAP_GEMM_GELU_EXP2) ? 2 : 1) : 0;
  p.ksplit = ksplit > 1 ? ksplit : 1;
  p.dbg = g_gemm_dbg;
  p.part = two_phase ? (float*)scratch : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (flags & LAP_GEMM_PARTIALS) {
    if (!scratch || ksplit < 1 || scratch_bytes < (long long)(ksplit > 1 ? ksplit : 1) * M * N * 4) return LAP_ERR_ARG;
    p.part = (float*)scratch;   // also valid for ksplit == 1: one slab
  }
  if (tail_tiles) {
    const int t5 = ((M + 255) / 256) * ((N + 255) / 256);
    int rc;
    // (a) the full rounds, straight to C
    p.ksplit = 1; p.part = nullptr; p.tile_base = 0; p.tile_count = t5 - tail_tiles;
    if (f32) {
      if (a_kc && b_kc) rc = dispatch_tile<true, true, true>(p, big, s);
      else if (a_kc && !b_kc) rc = dispatch_tile<true, false, true>(p, big, s);
      else if (!a_kc && !b_kc) rc = dispatch_tile<false, false, true>(p, big, s);
      else rc = dispatch_tile<false, true, true>(p, big, s);
    } else {
      if (a_kc && b_kc) rc = dispatch_tile<true, true, false>(p, big, s);
      else if (a_kc && !b_kc) rc = dispatch_tile<true, false, false>(p, big, s);
      else if (!a_kc && !b_kc) rc = dispatch_tile<false, false, false>(p, big, s);
      else rc = dispatch_tile<false, true, false>(p, big, s);
    }
    if (rc) return rc;
    if (tail_sp == 1) {
      // (b') the tail tiles as quadrants on the 128x128 kernel, straight to C with the caller's epilogue
      p.sub256 = 1; p.tile_base = t5 - tail_tiles; p.tile_count = tail_tiles;
      if (f32) {
        if (a_kc && b_kc) return dispatch_tile<true, true, true>(p, 6, s);
        if (a_kc && !b_kc) return dispatch_tile<true, false, true>(p, 6, s);
        if (!a_kc && !b_kc) return dispatch_tile<false, false, true>(p, 6, s);
        return dispatch_tile<false, true, true>(p, 6, s);
      }
      if (a_kc && b_kc) return dispatch_tile<true, true, false>(p, 6, s);
      if (a_kc && !b_kc) return dispatch_tile<true, false, false>(p, 6, s);
      if (!a_kc && !b_kc) return dispatch_tile<false, false, false>(p, 6, s);
      return dispatch_tile<false, true, false>(p, 6, s);
    }
    // (b) the tail tiles, split along K into compact f32 slabs, then reduce + epilogue
    p.ksplit = tail_sp; p.part = (float*)scratch; p.part_compact = 1; p.tile_base = t5 - tail_tiles; p.tile_count = tail_tiles;
    if (a_kc && b_kc) rc = dispatch_tile<true, true, true>(p, big, s);
    else if (a_kc && !b_kc) rc = dispatch_tile<true, false, true>(p, big, s);
    else if (!a_kc && !b_kc) rc = dispatch_tile<false, false, true>(p, big, s);
    else rc = dispatch_tile<false, true, true>(p, big, s);
    if (rc) return rc;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    if (f32) hipLaunchKernelGGL(splitk_tail_reduce_kernel<true>, dim3(tail_tiles * 64), dim3(256), 0, s, p, tail_tiles);
    else hipLaunchKernelGGL(splitk_tail_reduce_kernel<false>, dim3(tail_tiles * 64), dim3(256), 0, s, p, tail_tiles);
    LAP_CHECK_LAUNCH();
    return LAP_OK;
  }
  if (two_phase || (flags & LAP_GEMM_PARTIALS)) {
    int rc;
    if (a_kc && b_kc) rc = dispatch_tile<true, true, true>(p, tile, s);
    else if (a_kc && !b_kc) rc = dispatch_tile<true, false, true>(p, tile, s);
    else if (!a_kc && !b_kc) rc = dispatch_tile<false, false, true>(p, tile, s);
    else rc = dispatch_tile<false, true, true>(p, tile, s);
    if (rc) return rc;
    if (flags & LAP_GEMM_PARTIALS) return LAP_OK;
    const long long n4 = (long long)M * N / 4;
    if (f32) hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
    LAP_CHECK_LAUNCH();
    return LAP_OK;
  }
  if (f32) {
    if (a_kc && b_kc) return dispatch_tile<true, true, true>(p, tile, s);
    if (a_kc && !b_kc) return dispatch_tile<true, false, true>(p, tile, s);
    if (!a_kc && !b_kc) return dispatch_tile<false, false, true>(p, tile, s);
    return dispatch_tile<false, true, true>(p, tile, s);
  }
  if (a_kc && b_kc) return dispatch_tile<true, true, false>(p, tile, s);
  if (a_kc && !b_kc) return dispatch_tile<true, false, false>(p, tile, s);
  if (!a_kc && !b_kc) return dispatch_tile<false, false, false>(p, tile, s);
  return dispatch_tile<false, true, false>(p, tile, s);
}

extern "C" int lap_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual,
                             int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                             int a_kc, int b_kc, int flags, void* stream) {
  return lap_gemm_bf16_ex(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, alpha, a_kc, b_kc, flags, -1, 0, nullptr, 0,
                          stream);
}
