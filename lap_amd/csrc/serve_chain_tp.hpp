// The denoise step's expert layers as ONE persistent launch, TENSOR PARALLEL over the chip's 8 XCDs (round 4; lap.py:634-667 ->
// gemma.py:336-387 with only the suffix stream active and a KV cache).  serve_chain.hpp cuts every projection along N over all 256
// CUs, so every one of its five stages per layer ends at a chip-wide seam: a grid barrier (2.4 - 3.4 us each: 14 of the layer's 34 us)
// and activations that cross XCDs through device-scope stores and the memory side.  Measured on MI355X (tools/probes/cu_pull.hip):
// a barrier among the 32 blocks of ONE XCD with a hand-off through that XCD's L2 (plain stores, L1-bypassing loads) costs 1.07 us
// per round, no visibility errors.  So the layer is mapped the way a tensor-parallel group of eight devices would run it:
//
//   XCD h owns   query head h (its 256 q columns; the one k | v head is computed by every XCD for itself: 1 MB of weights re-read
//                out of the Infinity Cache instead of an all-to-all), attention of head h, the out projection's K slice h
//                (o_h [64 x 256] . Wo[256 h .. 256 h + 255][1024]), hidden columns 512 h .. 512 h + 511 of the MLP (gate | up, GeGLU)
//                and the down projection's K slice over them;
//   chip-wide    TWO seams per layer instead of five: the f32 partial sums of the out / down projections (8 slabs [64 x 1024],
//                written through) meet behind a grid barrier, and every XCD reduces them for itself — sum over the slabs in
//                fixed order, bf16 rounding of the projection, gated residual, adaRMS of the next block — into its own copy of
//                the residual stream and of the normed operand;
//   inside       five XCD-local barriers per layer: [reduce + norm] | [q_h k v + RoPE] | [attention of head h, one (key run, query
//                tile) per block + combine] | [out K slice] || [reduce + norm] | [gate | up slice + GeGLU] | [down K slice].
//
// Rounding points are the flat chain's (bf16 projection outputs, gated residual, bf16 norm outputs, f32 statistics); the K sums
// of the out / down projections and the norm's sum of squares are taken in another order, so the two chains agree to bf16 rounding
// noise, not bit for bit (tests/test_kernels_gpu.py states the bound).
// Placement: HIP promises nothing about block -> XCD placement, so nothing here assumes it: a block reads its XCC id from the
// hardware register and draws its rank inside the XCD from a ticket counter; every local hand-off is between blocks that ARE on one
// XCD.  What the kernel needs is 32 blocks on each of 8 XCDs (one per CU of an MI355X); a block that draws a rank >= 32 raises the
// error flag, every wait then gives up at once, the output is poisoned (serve_chain.hpp) and the caller falls back to the flat chain.
//
// Included by attention.hip inside its anonymous namespace, after serve_chain.hpp.
#pragma once

constexpr int TP_X = 8, TP_CU = 32;

struct TpP {
  ChainP c;                        // shapes, packed weights, caches, modulation slots, RoPE table, counters; q / o / act packed, global
  float* slab_o; float* slab_d;    // [64][8][D] f32: the XCDs' partial sums of the out / down projection, row major over (row, XCD)
  bf16* xs8; bf16* xn8;            // [8][64 x D] packed: each XCD's copy of the residual stream / of the adaRMS-normed operand
  bf16* k8; bf16* v8;              // [8][64][HD] row-major: each XCD's copy of the fresh keys / values
};

__device__ __forceinline__ int tp_xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }     // HW_REG_XCC_ID[3:0]

// barrier among the TP_CU blocks of one XCD: this block's stores are in the XCD's L2 (vmcnt(0)), arrival on the XCD's counter
__device__ __forceinline__ void tp_arrive(unsigned* ctrs, int xcd, unsigned& lround) {
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ++lround;
  if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(ctrs + CH_CTR_STRIDE * (CH_CTR_LOCAL + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void tp_wait(unsigned* ctrs, int xcd, unsigned lround) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctrs + CH_CTR_STRIDE * (CH_CTR_LOCAL + xcd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lround * TP_CU) {
      __builtin_amdgcn_s_sleep(1);
      if (chain_give_up(ctrs, ++spins)) break;
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The chip-wide seam, consumer side: rows 2 cu, 2 cu + 1 of the residual stream for THIS XCD (every XCD does all 64 rows for itself).
//   MODE 0  first layer: x = x_in (row-major, written by the launch before this one); no partial sums
//   MODE 1  y = sum of the 8 slabs (fixed order) -> bf16 -> x = x_prev + bf16(y * gate)      (gemma.py:577-583; skinny EPI_RESID's arithmetic)
//   MODE 2  as 1, the result goes to the caller's row-major x_out (XCD 0 writes it) and nothing else happens
// then (MODE 0 / 1) xs <- x and xn <- adaRMS(x; scale | shift of `modn`) (gemma.py:113-131; the skinny NORM prologue's arithmetic).
// 256 threads per row, 4 columns each (D = 1024).
template <int MODE>
__device__ __forceinline__ void tp_reduce(const ChainP& c, const float* slab, const bf16* gate, const bf16* modn, bf16* xs, bf16* xn, int cu, int xcd,
                                          float* red) {
  const int tid = opaque_tid();
  const int D = c.D;
  const int half = tid >> 8, col = (tid & 255) * 4;
  const int r = 2 * cu + half;
  const bool valid = r < c.M;
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
    if (MODE == 0) {
      const bf16x4 xi = *reinterpret_cast<const bf16x4*>(c.x_in + (long long)r * D + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (float)xi[e];
    } else {
      const auto rsS = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (unsigned)(TP_X * 64 * D * 4), 0x00020000);
      f32x4 part[TP_X];
#pragma unroll
      for (int j = 0; j < TP_X; ++j)
        part[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, (unsigned)((((r * TP_X + j) * D) + col) * 4), 0, 16));
      const bf16x4 xo = ldx4<true>(xs, pk_off(r, col, D), (unsigned)(64 * D * 2));
      const bf16x4 gt = *reinterpret_cast<const bf16x4*>(gate + col);
      f32x4 y = part[0];
#pragma unroll
      for (int j = 1; j < TP_X; ++j) y += part[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = (float)f2bf((float)xo[e] + round_bf16(round_bf16(y[e]) * (float)gt[e]));
    }
  }
  if (MODE == 2) {
    if (valid && xcd == 0) {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = f2bf(x[e]);
      *reinterpret_cast<bf16x4*>(c.x_out + (long long)r * D + col) = o;
    }
    return;
  }
  if (valid) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(x[e]);
    *reinterpret_cast<bf16x4*>(xs + pk_off(r, col, D)) = o;
  }
  float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
  ss = wave_sum(ss);
  const int w = tid >> 6;
  if ((tid & 63) == 0) red[w] = ss;
  __syncthreads();
  const float tot = red[half * 4] + red[half * 4 + 1] + red[half * 4 + 2] + red[half * 4 + 3];
  const float rstd = 1.0f / sqrtf(tot / (float)D + c.eps);
  if (valid) {
    const bf16x4 sc = *reinterpret_cast<const bf16x4*>(modn + col), sh = *reinterpret_cast<const bf16x4*>(modn + D + col);
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = f2bf(x[e] * rstd * round_bf16(1.0f + (float)sc[e]) + (float)sh[e]);
    *reinterpret_cast<bf16x4*>(xn + pk_off(r, col, D)) = h;
  }
}

__global__ __launch_bounds__(512) void serve_chain_tp_kernel(TpP t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // SV_LDS bytes: the attention stage's images; the projections use the front
  const ChainP& c = t.c;
  float* part = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + 65536);
  const int nb = gridDim.x;
  unsigned round = 0, lround = 0;
  int nstamp = 0;
  const int vb = blockIdx.x;
#define TP_STAMP() do { if (c.clk && threadIdx.x == 0 && (vb == 0 || vb == nb - 1)) c.clk[(vb ? 4096 : 0) + nstamp++] = wall_clock64(); } while (0)
  // ---- who am I inside my XCD
  __shared__ int s_rank;
  const int xcd = tp_xcc_id() & (TP_X - 1);
  if (threadIdx.x == 0) s_rank = (int)__hip_atomic_fetch_add(c.ctrs + CH_CTR_STRIDE * (CH_CTR_RANK + xcd), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int cu = s_rank;
  if (cu >= TP_CU) {     // not 32 blocks per XCD: this launch cannot work — flag it (every wait gives up, the output is poisoned)
    if (threadIdx.x == 0) __hip_atomic_fetch_or(c.ctrs + CH_CTR_STRIDE * CH_CTR_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cu &= TP_CU - 1;
  }
  const int D = c.D, H = c.H, NH = c.NH, HD = c.HD, QKV = NH * HD;
  bf16* xs = t.xs8 + (long long)xcd * 64 * D;
  bf16* xn = t.xn8 + (long long)xcd * 64 * D;
  bf16* kx = t.k8 + (long long)xcd * 64 * HD;
  bf16* vx = t.v8 + (long long)xcd * 64 * HD;
  // slabs are [row][XCD][D]: a row's eight partial sums are one contiguous 32 KiB.  ([XCD][row][D] put the eight loads of a thread
  // 256 KiB apart — the same memory channel — and the reduction took 4.0 us per seam instead of ...: tools/probes/chain_clock.py)
  float* so = t.slab_o + (long long)xcd * D;
  float* sd = t.slab_d + (long long)xcd * D;
  const int ns = c.sr.nruns, QT = (c.rps + 15) / 16, nA = ns * QT;          // attention blocks of one head: (run, query tile)

  int sbq[3];                                                               // this block's three feature tiles of [q_h | k | v]
#pragma unroll
  for (int f = 0; f < 3; ++f) {
    const int s = 3 * (cu & 15) + f;
    sbq[f] = s < 16 ? 16 * xcd + s : NH * 16 + (s - 16);
  }
  SkinnyP pq = {}, po = {}, pg = {}, pd = {};
  pq.x = xn; pq.M = c.M; pq.N = (NH + 2) * HD; pq.K = D; pq.ldx = D; pq.rps = c.rps; pq.eps = c.eps; pq.sbv = sbq;
  pq.o0 = c.q; pq.o1 = kx; pq.o2 = vx; pq.rope = c.rope; pq.NH = NH; pq.HD = HD; pq.q_scale = c.q_scale;
  po.x = c.o; po.M = c.M; po.N = D; po.K = QKV; po.ldx = QKV; po.rps = c.rps; po.ks0 = (QKV / 32 / TP_X) * xcd; po.pout = so; po.pld = TP_X * D;
  pg.x = xn; pg.M = c.M; pg.N = 2 * H; pg.K = D; pg.ldx = D; pg.rps = c.rps; pg.eps = c.eps; pg.o0 = c.act;
  pd.x = c.act; pd.M = c.M; pd.N = D; pd.K = H; pd.ldx = H; pd.rps = c.rps; pd.ks0 = (H / 32 / TP_X) * xcd; pd.pout = sd; pd.pld = TP_X * D;
  AttnP ap = c.attn;
  ap.q[1] = c.q; ap.k[1] = kx; ap.v[1] = vx; ap.o[1] = c.o;
  const int bxg = 16 * xcd + (cu & 15), byg = cu >> 4;                      // gate | up: 4 paired tiles x 32 tokens per block

  bf16x8 wq[3][4];
  pq.W = c.wqkv[0];
  skinny_load_w<EPI_ROPE, 4, 3, false, true>(pq, 0, wq);
  for (int l = 0; l < c.depth; ++l) {
    bf16x8 wo[2][1], wg[4][4], wd[2][2];
    const bf16* slot_a = c.mod + (long long)(2 * l) * c.slot_ld;
    const bf16* slot_f = slot_a + c.slot_ld;
    // ---- chip-wide seam (behind the grid barrier of the previous layer's down projection): residual + adaRMS of the attention block
    if (l == 0) tp_reduce<0>(c, nullptr, nullptr, slot_a, xs, xn, cu, xcd, red);
    else tp_reduce<1>(c, t.slab_d, slot_a - c.slot_ld + 2 * D, slot_a, xs, xn, cu, xcd, red);     // (the gate of the previous layer's MLP block)
    TP_STAMP();
    tp_arrive(c.ctrs, xcd, lround);
    tp_wait(c.ctrs, xcd, lround);
    TP_STAMP();
    // ---- q_h | k | v + RoPE / split
    skinny_rest<EPI_ROPE, false, 4, 3, 2, false, true, true, true>(pq, 0, cu >> 4, wq, part, red);
    TP_STAMP();
    tp_arrive(c.ctrs, xcd, lround);
    ap.k[0] = c.ck[l]; ap.v[0] = c.cv[l];     // the cached keys / values of the block's run: on their way into LDS during the barrier
    if (cu < nA) attn_run_body<true, 1, true, true>(ap, c.sr, cu % ns, cu / ns, xcd, 0, smem);
    tp_wait(c.ctrs, xcd, lround);
    TP_STAMP();
    // ---- attention of head `xcd`: one (key run, query tile) per block, then the combine of the runs of a tile
    if (cu < nA) {
      const int sI = cu % ns, qI = cu / ns;
      attn_run_body<true, 2, true, true>(ap, c.sr, sI, qI, xcd, 0, smem);
      __builtin_amdgcn_s_waitcnt(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (threadIdx.x == 0) {
        unsigned* hc = c.ctrs + CH_CTR_STRIDE * (CH_CTR_HEAD + xcd * QT + qI);
        (void)__hip_atomic_fetch_add(hc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(l + 1) * (unsigned)ns;
        unsigned spins = 0;
        while (__hip_atomic_load(hc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (chain_give_up(c.ctrs, ++spins)) break;
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const int per_tile = 16 * (HD / 4);
      for (int it = sI * 512 + (int)threadIdx.x; it < per_tile; it += ns * 512) {
        const int tq = qI * 16 + it / (HD / 4), j = it % (HD / 4);
        if (tq < c.rps) attn_serve_combine_body<8, true, true, true>(ap, ((long long)tq * NH + xcd) * (HD / 4) + j);
      }
    }
    TP_STAMP();
    tp_arrive(c.ctrs, xcd, lround);
    po.W = c.wo[l];
    skinny_load_w<EPI_PARTIAL, 1, 2, false, true>(po, cu, wo);
    __builtin_amdgcn_sched_barrier(0);
    tp_wait(c.ctrs, xcd, lround);
    TP_STAMP();
    // ---- out projection, K slice of head `xcd`: f32 partial sums for every XCD
    skinny_rest<EPI_PARTIAL, false, 1, 2, 4, false, true, true, true>(po, cu, 0, wo, part, red);
    TP_STAMP();
    chain_arrive(c.ctrs, round, nb);
    pg.W = c.wgu[l];
    skinny_load_w<EPI_GEGLU, 4, 4, false, true>(pg, bxg, wg);     // (two stages ahead: in flight through the seam and the reduction)
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    TP_STAMP();
    // ---- chip-wide seam: residual of the attention block + adaRMS of the MLP block
    tp_reduce<1>(c, t.slab_o, slot_a + 2 * D, slot_f, xs, xn, cu, xcd, red);
    TP_STAMP();
    tp_arrive(c.ctrs, xcd, lround);
    tp_wait(c.ctrs, xcd, lround);
    TP_STAMP();
    // ---- gate | up of hidden columns 512 xcd .. + GeGLU
    skinny_rest<EPI_GEGLU, false, 4, 4, 2, false, true, true, true>(pg, bxg, byg, wg, part, red);
    TP_STAMP();
    tp_arrive(c.ctrs, xcd, lround);
    pd.W = c.wd[l];
    skinny_load_w<EPI_PARTIAL, 2, 2, false, true>(pd, cu, wd);
    __builtin_amdgcn_sched_barrier(0);
    tp_wait(c.ctrs, xcd, lround);
    TP_STAMP();
    // ---- down projection, K slice over those columns
    skinny_rest<EPI_PARTIAL, false, 2, 2, 4, false, true, true, true>(pd, cu, 0, wd, part, red);
    TP_STAMP();
    chain_arrive(c.ctrs, round, nb);
    pq.W = c.wqkv[l + 1 < c.depth ? l + 1 : l];     // (the last layer re-reads its own: nobody uses them)
    skinny_load_w<EPI_ROPE, 4, 3, false, true>(pq, 0, wq);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    TP_STAMP();
  }
  // ---- the last layer's MLP residual: the caller's row-major x_out
  tp_reduce<2>(c, t.slab_d, c.mod + (long long)(2 * c.depth - 1) * c.slot_ld + 2 * D, nullptr, xs, xn, cu, xcd, red);
  if (__hip_atomic_load(c.ctrs + CH_CTR_STRIDE * CH_CTR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    __syncthreads();
    unsigned short* xo = reinterpret_cast<unsigned short*>(c.x_out);
    for (int j = blockIdx.x * 512 + (int)threadIdx.x; j < c.M * D; j += nb * 512) xo[j] = 0x7fc0;     // bf16 NaN (serve_chain.hpp)
  }
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(c.ctrs + CH_CTR_STRIDE * CH_CTR_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)nb - 1) {
      for (int j = 0; j <= CH_GROUPS; ++j) __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = 0; j < NH * QT; ++j) __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * (CH_CTR_HEAD + j), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = 0; j < 2 * TP_X; ++j) __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * (CH_CTR_LOCAL + j), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * CH_CTR_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// shapes: the flat chain's, one sample, one query head per XCD, and a head's attention blocks within one XCD's 32 CUs
inline bool chain_tp_ok(int B, int S, int D, int H, int NH, int HD, int NKV, int prefix_len) {
  if (!chain_ok(B, S, D, H, NH, HD, NKV, prefix_len) || B != 1 || NH != TP_X) return false;
  const ServeRuns sr = serve_runs(prefix_len, S, serve_run_cap(B, NH, S));
  return sr.nruns * ((S + 15) / 16) <= TP_CU;
}

int launch_chain_tp(const TpP& t, hipStream_t s) {
  if (!chain_device_ok()) return LAP_ERR_ARG;
  static bool attr = false;
  if (!attr) {
    if (int e = set_lds(serve_chain_tp_kernel, SV_LDS)) return e;
    attr = true;
  }
  if (t.c.sr.nruns < 1 || t.c.sr.nruns > 8) return LAP_ERR_ARG;
  hipLaunchKernelGGL(serve_chain_tp_kernel, dim3(CH_BLOCKS), dim3(512), SV_LDS, s, t);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
