// The body of the skinny-M fused projections (design notes: serve_skinny.hip), as device functions, so that the stand-alone
// launches (serve_skinny.hip) and the persistent denoise-layer chain (serve_chain.hpp) run the SAME arithmetic in the same
// order: the two paths are bitwise equal by construction.  Included inside an anonymous namespace, after common.hpp.
//
// COH (the chain only): activations written and read INSIDE one launch by blocks on different XCDs.  Their stores and loads
// carry the device-scope bit (sc1) — what gfx950 uses for relaxed agent-scope atomics: stores write through to memory, loads do
// not hit possibly stale lines of the XCD's own L2 — so the chain's grid barrier needs no cache write-back / invalidate
// (tools/probes/gridbar_sc1.hip: 2.7 us per barrier round against 8.5 us with release / acquire fences, no visibility errors).
#pragma once

constexpr int SK_WAVES = 8;
constexpr int SK_TOK = 16;      // tokens per block (one MFMA column tile)

enum { EPI_ROPE = 1, EPI_GEGLU = 2, EPI_RESID = 3, EPI_PARTIAL = 4 };     // PARTIAL: f32 partial sums of a K slice (serve_chain_tp.hpp)

struct SkinnyP {
  const bf16* x;        // [M][ldx] block input (K columns used)
  const bf16* W;        // [N][K] weights, K contiguous
  int M, N, K, ldx;
  // prologue
  const bf16* mod;      // scale | shift | gate, [.., 3K] per sample (row stride mod_ld; 0 = one row for all samples)
  int mod_ld, rps;      // rows per sample
  float eps;
  // epilogues
  bf16 *o0, *o1, *o2;   // ROPE: q [M][NH*HD], k [M][HD], v [M][HD];  GEGLU: act [M][H];  RESID: out [M][N]
  const bf16* resid;    // RESID: x [M][N]
  const bf16* gate;     // RESID: gate [.., N] per sample (row stride gate_ld; NULL = plain add)
  int gate_ld;
  const float* rope;    // ROPE: sin / cos f32 [M][HD/2][2] (lap_rope_table)
  int NH, HD;
  float q_scale;
  // PK only (see pk_off): which of the block input / residual input / output are still row-major (the chain's first and last stage)
  int x_rm, resid_rm, o_rm;
  // PK only, the tensor-parallel chain (serve_chain_tp.hpp): the block contracts over the K slice that starts at k-step ks0 (a wave's
  // steps are ks0 + w KS + s; K stays the full row length of the packed operands); sbv != NULL: the feature tiles of the block are
  // sbv[0 .. FT) instead of bx FT + f; PARTIAL: f32 sums [M][N] at `pout`
  int ks0;
  const int* sbv;
  float* pout;
  int pld;              // PARTIAL: row stride of pout in floats
};

// ---- PK: fragment-packed layouts (round 4).  tools/probes/cu_pull.hip: a lane that loads its MFMA fragment straight from a
// row-major matrix makes a wave instruction touch 16 rows x 64 B, and the CU's vector-memory path then delivers 37 GB/s whatever
// the data's temperature (one tag lookup per 16 B); 1 KiB contiguous per wave instruction gets 130 GB/s out of the L2.  So the
// chain keeps every operand in the order the MFMA wants it: a [16 rows x 32 k] tile is 64 lanes x 16 B = 1 KiB, lane = 16 g + i
// holds row i, k = 8 g .. 8 g + 7; tiles of one row group are consecutive in k.  Weights are packed once per parameter version
// (lap_serve_pack_weight, with the epilogue's row pairing folded in), activations are WRITTEN packed by the producing epilogue.
// Same values, same summation order: bitwise equal to the row-major path.
// (pk_off: common.hpp)

__device__ __forceinline__ float sum4groups(float v) {   // over lanes {i, i+16, i+32, i+48}
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// the weight stream: every byte is used by one launch once (nontemporal: does not displace the L2-resident activations)
template <bool NT>
__device__ __forceinline__ bf16x8 ldw8(const __amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, NT ? 2 : 0);
  return __builtin_bit_cast(bf16x8, v);
}
// activations: device-scope (sc1) accesses when COH
template <bool COH>
__device__ __forceinline__ bf16x8 ldx8(const __amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, COH ? 16 : 0);
  return __builtin_bit_cast(bf16x8, v);
}
template <bool COH>
__device__ __forceinline__ bf16x4 ldx4(const bf16* base, long long elem, unsigned total_bytes) {   // base: wave uniform
  if constexpr (COH) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, total_bytes, 0x00020000);
    u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)(elem * 2), 0, 16);
    return __builtin_bit_cast(bf16x4, v);
  } else {
    return *reinterpret_cast<const bf16x4*>(base + elem);
  }
}
template <bool COH>
__device__ __forceinline__ void stx4(bf16* p, bf16x4 v) {
  if constexpr (COH) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(bits) : "memory");
  } else {
    *reinterpret_cast<bf16x4*>(p) = v;
  }
}

// The thread id behind an opaque copy: inside the chain's layer loop the compiler would otherwise hoist every lane-derived address
// of every stage out of the loop and keep them all alive (spilling the prefetched weights instead).
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

constexpr int skinny_part_floats(int FT, int TT) { return SK_WAVES * FT * TT * 64 * 4; }   // per-wave partial output tiles
constexpr int skinny_red_floats(int TT) { return SK_WAVES * TT * SK_TOK; }

// ---- the weight stream of block column bx: which weight row feeds operand row i of feature tile f (sub-block sb = bx * FT + f).
// Independent of every activation: the chain issues it BEFORE the barrier that guards the block input.
template <int EPI, int KS, int FT, bool NT, bool PK = false>
__device__ __forceinline__ void skinny_load_w(const SkinnyP& p, int bx, bf16x8 (&wf)[FT][KS]) {
  const int tid = opaque_tid();
  const int lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int k0 = w * (KS * 32) + g * 8;
  const auto rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((long long)p.N * p.K * 2), 0x00020000);
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int sb = PK && p.sbv ? p.sbv[f] : bx * FT + f;
    if constexpr (PK) {      // tile (sb, k-step ks0 + w KS + s): 1 KiB, this lane's 16 bytes at lane * 16
      const unsigned woff = (unsigned)((((long long)sb * (p.K >> 5) + p.ks0 + w * KS) * 64 + lane) * 16);
#pragma unroll
      for (int s = 0; s < KS; ++s) wf[f][s] = ldw8<NT>(rsW, woff + s * 1024);
      continue;
    }
    int wrow;
    if (EPI == EPI_ROPE) {
      const int bph = p.HD / 16, h = sb / bph, j = sb % bph;
      wrow = h * p.HD + (i < 8 ? j * 8 + i : p.HD / 2 + j * 8 + (i - 8));
    } else if (EPI == EPI_GEGLU) {
      wrow = i < 8 ? sb * 8 + i : p.N / 2 + sb * 8 + (i - 8);
    } else {
      wrow = sb * 16 + i;
    }
    const unsigned woff = (unsigned)(((long long)wrow * p.K + k0) * 2);
#pragma unroll
    for (int s = 0; s < KS; ++s) wf[f][s] = ldw8<NT>(rsW, woff + s * 64);
  }
}

// ---- everything behind the weight loads: block input (+ adaptive RMSNorm), MFMAs, cross-wave reduction, epilogue.
// Block = (FT x 16 output features, TT x 16 tokens) at (bx, by).  Measured on MI355X (tools/bench_skinny.py): a block's time is
// ~2.4 us + (bytes it loads) / ~35 GB/s — the per-CU vector-memory path, not HBM latency (L2-warm weights are only 0.6 us
// faster) — so the shape of a block is chosen to minimise (TT + FT) * 16 * K * 2 bytes per block at one round of <= 256
// blocks: qkv 32 x 32, gate|up 32 tokens x 64 features, out / down 16 x 16.  The token groups of a chunk re-read the same weight
// rows; with the number of block columns a multiple of 8 they share an XCD (block id % 8), so the repeats are L2 hits and HBM
// sees every weight byte once.
// SHM: every token shares ONE modulation row (mod_ld == 0: the denoise step, where the condition is the step's time) — the
// prologue then loads scale / shift once per k-slice instead of once per token tile (qkv: 256 -> 132 KB per block).
// part: skinny_part_floats(FT, TT) floats of LDS, red: skinny_red_floats(TT).
// LST (the tensor-parallel chain): outputs go to consumers of the SAME XCD — plain stores (they stay in that XCD's L2) where COH alone
// would write through with the device-scope bit; loads keep it (L1 bypass, served by the L2).
template <int EPI, bool NORM, int KS, int FT, int TT, bool SHM, bool COH, bool PK = false, bool LST = false>   // KS: 32-deep k-steps per wave (K slice = KS * 32 * SK_WAVES)
__device__ __forceinline__ void skinny_rest(const SkinnyP& p, int bx, int by, bf16x8 (&wf)[FT][KS], float* part, float* red) {
  const int tid = opaque_tid();
  const int lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int k0 = w * (KS * 32) + g * 8;
  const bool xpk = PK && !p.x_rm;       // (wave uniform) packed block input: [row tiles of 16][K / 32 k-steps][1 KiB]
  const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xpk ? (unsigned)(((p.M + 15) >> 4) * p.K * 32) : (unsigned)(((long long)(p.M - 1) * p.ldx + p.K) * 2), 0x00020000);
  const unsigned xstep = xpk ? 1024u : 64u;
  bf16x8 xf[TT][KS];
  constexpr int MT = NORM ? (SHM ? 1 : TT) : 1;
  bf16x8 sc[MT][NORM ? KS : 1], sh[MT][NORM ? KS : 1];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int r = (by * TT + t) * SK_TOK + i;     // this lane's token row of tile t (MFMA column i)
    // rows past M read as zeros (row-major: out of range offset; packed: tiles past the last one are out of range, the pad rows of
    // the last tile are never written and belong to columns of the product nobody stores)
    const unsigned xoff = xpk ? (unsigned)((((long long)(by * TT + t) * (p.K >> 5) + p.ks0 + w * KS) * 64 + lane) * 16)
                              : (r < p.M ? (unsigned)(((long long)r * p.ldx + k0) * 2) : 0x80000000u);
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[t][s] = ldx8<COH>(rsX, xoff + s * xstep);
    if (NORM && (!SHM || t == 0)) {
      const bf16* mrow = p.mod + (SHM ? 0 : (long long)((r < p.M ? r : 0) / p.rps) * p.mod_ld) + k0;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sc[SHM ? 0 : t][s] = *reinterpret_cast<const bf16x8*>(mrow + s * 32);
        sh[SHM ? 0 : t][s] = *reinterpret_cast<const bf16x8*>(mrow + p.K + s * 32);
      }
    }
  }
  // every load of the block is in flight before anything waits: left alone, hipcc sinks each load next to its use and
  // turns the weight stream into a chain of dependent round trips
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[FT][TT];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (NORM) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float ss = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = (float)xf[t][s][e]; ss += v * v; }
      ss = sum4groups(ss);
      if (g == 0) red[(w * TT + t) * SK_TOK + i] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float tot = 0.f;
#pragma unroll
      for (int ww = 0; ww < SK_WAVES; ++ww) tot += red[(ww * TT + t) * SK_TOK + i];
      const float rstd = 1.0f / sqrtf(tot / (float)p.K + p.eps);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          h[e] = f2bf((float)xf[t][s][e] * rstd * round_bf16(1.0f + (float)sc[SHM ? 0 : t][s][e]) + (float)sh[SHM ? 0 : t][s][e]);
        xf[t][s] = h;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[f][t] = mfma16(wf[f][s], xf[t][s], acc[f][t]);
  // ---- cross-wave reduction (wave order): lane (i, g) of tile (f, t) holds features 4g .. 4g+3 of token i
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int t = 0; t < TT; ++t) *reinterpret_cast<f32x4*>(part + ((w * (FT * TT) + f * TT + t) * 64 + lane) * 4) = acc[f][t];
  __syncthreads();
  if (w >= FT * TT) return;          // wave (f, t) finishes output tile (f, t)
  const int f = w / TT, t = w % TT, sb = PK && p.sbv ? p.sbv[f] : bx * FT + f;
  const int r = (by * TT + t) * SK_TOK + i;
  f32x4 y = *reinterpret_cast<const f32x4*>(part + (w * 64 + lane) * 4);
#pragma unroll
  for (int ww = 1; ww < SK_WAVES; ++ww) y += *reinterpret_cast<const f32x4*>(part + ((ww * (FT * TT) + w) * 64 + lane) * 4);
  if constexpr (EPI == EPI_PARTIAL) {     // f32 partial sums of this block's K slice: features 4 g .. 4 g + 3 of token row r
    if (r < p.M) {
      float* dst = p.pout + (long long)r * p.pld + sb * 16 + 4 * g;
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(y) : "memory");     // (read by every XCD behind the grid barrier)
    }
    return;
  }
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = round_bf16(y[e]);      // the projection's bf16 output
  // partner lane (g ^ 2) holds the other half of each pair (rotation partner d + HD/2, or the up column of a gate column)
  float pv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) pv[e] = __shfl_xor(v[e], 32, 64);
  if (r >= p.M) return;
  if (EPI == EPI_RESID) {
    const int c = sb * 16 + 4 * g;
    const long long ro = PK && !p.resid_rm ? pk_off(r, c, p.N) : (long long)r * p.N + c;
    const bf16x4 xr = ldx4<COH>(p.resid, ro, (unsigned)((long long)(PK && !p.resid_rm ? ((p.M + 15) & ~15) : p.M) * p.N * 2));
    bf16x4 o;
    if (p.gate) {
      const bf16x4 gt = *reinterpret_cast<const bf16x4*>(p.gate + (long long)(r / p.rps) * p.gate_ld + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf((float)xr[e] + round_bf16(v[e] * (float)gt[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf((float)xr[e] + v[e]);
    }
    stx4<COH>(p.o0 + (PK && !p.o_rm ? pk_off(r, c, p.N) : (long long)r * p.N + c), o);
  } else if (EPI == EPI_GEGLU) {
    if (g >= 2) return;        // lanes g = 0, 1 hold gate columns 4g .. 4g+3; their partners the matching up columns
    const int H = p.N / 2, c = sb * 8 + 4 * g;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(round_bf16(gelu_tanh_f(v[e])) * pv[e]);
    stx4<COH && !LST>(p.o0 + (PK ? pk_off(r, c, H) : (long long)r * H + c), o);
  } else {   // EPI_ROPE
    const int HD = p.HD, half = HD / 2, bph = HD / 16, h = sb / bph, j = sb % bph;
    const int f0 = j * 8 + 4 * (g & 1);             // first of this lane's 4 frequencies
    bf16x4 o;
    if (h <= p.NH) {                                 // q heads and the k head rotate
      const float* tb = p.rope + ((long long)r * half + f0) * 2;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tb), t1 = *reinterpret_cast<const f32x4*>(tb + 4);
      const float sn[4] = {t0[0], t0[2], t1[0], t1[2]}, cs[4] = {t0[1], t0[3], t1[1], t1[3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x1 = g < 2 ? v[e] : pv[e], x2 = g < 2 ? pv[e] : v[e];
        float r1, r2;
        rope_rotate(x1, x2, sn[e], cs[e], r1, r2);
        float rr = round_bf16(g < 2 ? r1 : r2);
        if (h < p.NH) rr *= p.q_scale;
        o[e] = f2bf(rr);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
    }
    const int d = (g < 2 ? 0 : half) + f0;
    // (PK: the queries packed like every other MFMA operand of the chain; k / v stay row-major: they go to LDS as whole rows)
    bf16* dst = h < p.NH ? (PK ? p.o0 + pk_off(r, h * HD + d, p.NH * HD) - d : p.o0 + (long long)r * p.NH * HD + h * HD)
                         : (h == p.NH ? p.o1 + (long long)r * HD : p.o2 + (long long)r * HD);
    stx4<COH && !LST>(dst + d, o);
  }
}
