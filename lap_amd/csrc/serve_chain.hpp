// One launch for the 18 action-expert layers of a denoise step (lap.py:634-667 -> gemma.py:336-387 with only the suffix stream
// active and a KV cache): the six launches per layer of the stand-alone path (serve_skinny.hip, attention_serve.hpp) become five
// STAGES of one persistent kernel — 256 blocks, one per CU, all resident — separated by a software grid barrier:
//
//   [adaRMS + qkv + RoPE/split] | attention over [cache | fresh keys], one (key run, query tile, head) per block, + combine of the
//   runs (the runs of one (head, query tile) meet at a counter of their own) | [out projection + gated residual] | [adaRMS + gate|up + GeGLU] |
//   [down projection + gated residual]
//
// Why it pays now (it did not in round 2, docs/EXPERIMENTS.md): the barrier used to cost 6.5-11 us because agent-scope release /
// acquire fences write back and invalidate the XCD's L2.  Here every activation that crosses a stage is stored and loaded with
// the device-scope bit (sc1, serve_skinny_body.hpp), so the barrier is "s_waitcnt vmcnt(0)" + a two-level arrival counter
// (per-XCD group, the group's last arriver bumps the master) + a poll: 2.7 us per round on MI355X, no visibility errors
// (tools/probes/gridbar_sc1.hip) — against ~4.5 us of fixed cost per dependent launch inside a hipGraph.  And a block knows its
// NEXT stage's weight rows (and the attention stage its run of the CACHED keys / values) before the barrier: it issues those
// loads first and polls while they fly, so the per-CU load path (the real bound of these projections, ~35-48 GB/s per CU)
// streams them during what used to be launch latency.
// The stage bodies are the stand-alone kernels' device functions: the chain is bitwise equal to the six-launch path.
// Measured (tools/probes/chain_clock.py, 18 layers, 50 action tokens, 816 cached keys): 48.0 us per layer = 29 us of stage work on
// the slowest block (qkv 5.4, attention + combine 11.0, out 2.7, gate|up 5.6, down 4.4) + 18 us in the five barriers (3.1-4.4 us
// each: the 2.7 us round plus the skew between blocks), against 50.5 us for the six launches inside a replayed graph; one chunk
// (prefill + 10 steps) 15.6 -> 15.4-15.5 ms replayed from a hipGraph, 20.0 -> 15.7 ms launched eagerly (1469 -> 399 kernels).
//
// Included by attention.hip inside its anonymous namespace, after attention_serve.hpp.
#pragma once

#include "serve_skinny_body.hpp"

constexpr int CH_MAX_DEPTH = 32;
constexpr int CH_GROUPS = 8;             // arrival groups = XCDs (block id % 8)
constexpr int CH_CTR_STRIDE = 64;        // counters 256 B apart
constexpr int CH_CTR_EXIT = 9, CH_CTR_ERR = 10, CH_CTR_HEAD = 11, CH_MAX_HEADS = 64;    // HEAD: one counter per (sample, head, query tile), see the attention stage
constexpr int CH_CTR_LOCAL = CH_CTR_HEAD + CH_MAX_HEADS, CH_CTR_RANK = CH_CTR_LOCAL + 8;     // the tensor-parallel chain (serve_chain_tp.hpp): per-XCD barrier counters, rank tickets
constexpr int CH_CTR_WORDS = (CH_CTR_RANK + 8) * CH_CTR_STRIDE;
constexpr int CH_BLOCKS = 256;

struct ChainP {
  int depth, M, rps;                     // M = B * S action tokens, S rows per sample
  int D, H, NH, HD;
  const bf16* x_in; bf16* x_out;         // [M][D]
  const bf16* mod; int slot_ld;          // modulation slots (scale | shift | gate, 3 D each), slot j at mod + j * slot_ld; one row for all tokens
  const bf16* wqkv[CH_MAX_DEPTH]; const bf16* wo[CH_MAX_DEPTH]; const bf16* wgu[CH_MAX_DEPTH]; const bf16* wd[CH_MAX_DEPTH];
  const bf16* ck[CH_MAX_DEPTH]; const bf16* cv[CH_MAX_DEPTH];
  AttnP attn;                            // everything but the per-layer cache pointers
  ServeRuns sr;
  const float* rope;
  float q_scale, eps;
  bf16 *q, *k, *v, *o, *xa, *act;        // activations between the stages
  bf16* xs;                              // PK: the residual stream between the layers, packed (x_in / x_out stay row-major for the caller)
  unsigned* ctrs;                        // CH_CTR_WORDS words, zero before the first launch (the kernel leaves them zero)
  unsigned long long* clk;               // tuning aid (tools/probes/chain_clock.py): block 0 / block 255 stamp the 100 MHz clock at every stage edge; NULL normally
};

#define CH_STAMP() do { if (c.clk && threadIdx.x == 0 && (vb == 0 || vb == nb - 1)) c.clk[(vb ? 4096 : 0) + nstamp++] = wall_clock64(); } while (0)

// End of a stage: this wave's stores are performed (and nothing of the stage is in flight), every wave of the block is through
// with the stage's LDS, and the block has ARRIVED at the grid barrier.  The arrival is posted BEFORE the next stage's weight
// prefetch is issued: vector-memory operations return in order, so an arrival counter read behind 64 KB of weight loads would
// hold the group's last arriver — and with it every block of the chip — until those loads have landed.
// Two levels: a counter per XCD group (block id % 8), the group's last arriver bumps the master everybody polls.  (Measured
// alternatives: one flat counter 4.2 us per round instead of 2.7; no master, every block polling the eight group counters: the
// polls get in the way of the arrivals on the same lines, +4 us per layer; group = the real XCD with the group counter as an
// atomic of that XCD's own L2: 49.4 vs 49.1 us per layer, nothing.)
__device__ __forceinline__ void chain_arrive(unsigned* ctrs, unsigned& round, int nb) {
  __builtin_amdgcn_s_waitcnt(0);       // vmcnt(0) expcnt(0) lgkmcnt(0) (a real instruction: the compiler's counter model sees it)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (orders the inline-assembly sc1 stores in front of it as well)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  ++round;
  if (threadIdx.x == 0) {
    const int group = blockIdx.x % CH_GROUPS, gsize = nb / CH_GROUPS;
    const unsigned old = __hip_atomic_fetch_add(ctrs + CH_CTR_STRIDE * (1 + group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old + 1) % gsize == 0) __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("" ::: "memory");
}
// Never hang the GPU: a wait is bounded (2^21 polls, about a second); the block that gives up raises the error flag
// (lap_serve_chain_status) and goes on, and from then on EVERY wait of the launch gives up after 1024 polls (~900
// barriers must not time out one second at a time).  The launch's output is poisoned at its end (chain_poison).
__device__ __forceinline__ bool chain_give_up(unsigned* ctrs, unsigned spins) {
  if (spins & 1023u) return false;
  if (spins <= (1u << 21) && __hip_atomic_load(ctrs + CH_CTR_STRIDE * CH_CTR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return false;
  __hip_atomic_fetch_or(ctrs + CH_CTR_STRIDE * CH_CTR_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
// wait until all blocks have arrived
__device__ __forceinline__ void chain_wait(unsigned* ctrs, unsigned round) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * CH_GROUPS) {
      __builtin_amdgcn_s_sleep(1);
      if (chain_give_up(ctrs, ++spins)) break;
    }
  }
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// PK: fragment-packed weights and activations (serve_skinny_body.hpp): every operand load of every stage is 1 KiB contiguous per
// wave instruction.  Same arithmetic in the same order as the row-major form.
template <int NS, bool PK>
__global__ __launch_bounds__(512) void serve_chain_kernel(ChainP c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];     // SV_LDS bytes: the attention stage's images; the projections use the front
  float* part = reinterpret_cast<float*>(smem);
  float* red = reinterpret_cast<float*>(smem + 65536);
  const int vb = blockIdx.x, nb = gridDim.x;
  unsigned round = 0;
  int nstamp = 0;
  const int tg32 = (c.M + 31) / 32, tg16 = (c.M + 15) / 16;
  const int fgQ = (c.NH + 2) * c.HD / 32, fgO = c.D / 16, fgG = c.H / 32;
  const int nQ = fgQ * tg32, nO = fgO * tg16, nG = fgG * tg32;
  const int ns = c.sr.nruns, QT = (c.rps + 15) / 16, nA = ns * QT * c.NH * c.attn.B;     // attention blocks: (run, query tile, head, sample)
  const int QKV = c.NH * c.HD;

  SkinnyP pq = {}, po = {}, pg = {}, pd = {};
  pq.M = c.M; pq.N = (c.NH + 2) * c.HD; pq.K = c.D; pq.ldx = c.D; pq.mod_ld = 0; pq.rps = c.rps; pq.eps = c.eps;
  pq.o0 = c.q; pq.o1 = c.k; pq.o2 = c.v; pq.rope = c.rope; pq.NH = c.NH; pq.HD = c.HD; pq.q_scale = c.q_scale;
  po.x = c.o; po.M = c.M; po.N = c.D; po.K = QKV; po.ldx = QKV; po.rps = c.rps; po.o0 = c.xa; po.gate_ld = 0;
  pg.x = c.xa; pg.M = c.M; pg.N = 2 * c.H; pg.K = c.D; pg.ldx = c.D; pg.mod_ld = 0; pg.rps = c.rps; pg.eps = c.eps; pg.o0 = c.act;
  pd.x = c.act; pd.M = c.M; pd.N = c.D; pd.K = c.H; pd.ldx = c.H; pd.rps = c.rps; pd.o0 = c.x_out; pd.resid = c.xa; pd.gate_ld = 0;
  AttnP ap = c.attn;
  ap.q[1] = c.q; ap.k[1] = c.k; ap.v[1] = c.v; ap.o[1] = c.o;

  // (the weight registers of a stage live from their prefetch to the stage's MFMAs only: declared per iteration and loaded
  // unconditionally — a value carried around the loop, or defined on one side of a branch only, stays allocated / gets spilled)
  bf16x8 wq[2][4];
  pq.W = c.wqkv[0];
  skinny_load_w<EPI_ROPE, 4, 2, false, PK>(pq, vb % fgQ, wq);
  bf16* const xs = PK ? c.xs : c.x_out;      // the stream between the layers
  for (int l = 0; l < c.depth; ++l) {
    bf16x8 wo[1][8], wg[4][4], wd[1][16];
    const bf16* slot_a = c.mod + (long long)(2 * l) * c.slot_ld;
    const bf16* slot_f = slot_a + c.slot_ld;
    // ---- adaRMS + qkv + RoPE / split
    pq.x = l == 0 ? c.x_in : xs; pq.mod = slot_a; pq.x_rm = l == 0;
    if (vb < nQ) skinny_rest<EPI_ROPE, true, 4, 2, 2, true, true, PK>(pq, vb % fgQ, vb / fgQ, wq, part, red);
    CH_STAMP();
    chain_arrive(c.ctrs, round, nb);
    ap.k[0] = c.ck[l]; ap.v[0] = c.cv[l];     // the cached keys / values of the block's run: on their way into LDS during the barrier
    if (vb < nA) attn_run_body<true, 1, PK>(ap, c.sr, vb % ns, (vb / ns) % QT, (vb / (ns * QT)) % c.NH, vb / (ns * QT * c.NH), smem);
    chain_wait(c.ctrs, round);
    CH_STAMP();
    // ---- attention of the action queries over [cached prefix | fresh keys], one key run per block
    if (vb < nA) {
      const int sI = vb % ns, qI = (vb / ns) % QT, hI = (vb / (ns * QT)) % c.NH, bI = vb / (ns * QT * c.NH);
      unsigned long long* tclk = c.clk && (vb == 0 || vb == nb - 1) ? c.clk + (vb ? 6144 : 2048) + 8 * l : nullptr;
      if (tclk && threadIdx.x == 0) tclk[7] = wall_clock64();
      attn_run_body<true, 2, PK>(ap, c.sr, sI, qI, hI, bI, smem, tclk);
      // ---- combine of the key runs: only the `ns` blocks of one (sample, head, query tile) depend on each other, so they meet at
      // a counter of their own instead of a grid barrier (7 arrivals instead of 256), and each merges its share of the tile's rows
      __builtin_amdgcn_s_waitcnt(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tclk && threadIdx.x == 0) tclk[3] = wall_clock64();
      if (threadIdx.x == 0) {
        unsigned* hc = c.ctrs + CH_CTR_STRIDE * (CH_CTR_HEAD + (bI * c.NH + hI) * QT + qI);
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
      if (tclk && threadIdx.x == 0) tclk[4] = wall_clock64();
      const int per_tile = 16 * (c.HD / 4);                          // 4-column items of this (sample, head, tile): row t, columns 4 j ..
      for (int it = sI * 512 + (int)threadIdx.x; it < per_tile; it += ns * 512) {
        const int t = qI * 16 + it / (c.HD / 4), j = it % (c.HD / 4);
        if (t < c.rps) attn_serve_combine_body<NS, true, PK>(ap, (((long long)bI * c.rps + t) * c.NH + hI) * (c.HD / 4) + j);
      }
    }
    CH_STAMP();
    chain_arrive(c.ctrs, round, nb);
    po.W = c.wo[l];     // the out projection's weights: in flight during the barrier
    skinny_load_w<EPI_RESID, 8, 1, false, PK>(po, vb % fgO, wo);      // (unconditional: a block without work in the stage loads rows it drops)
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    CH_STAMP();
    // ---- out projection + gated residual
    po.resid = l == 0 ? c.x_in : xs; po.gate = slot_a + 2 * c.D; po.resid_rm = l == 0;
    if (vb < nO) skinny_rest<EPI_RESID, false, 8, 1, 1, false, true, PK>(po, vb % fgO, vb / fgO, wo, part, red);
    CH_STAMP();
    chain_arrive(c.ctrs, round, nb);
    pg.W = c.wgu[l];
    skinny_load_w<EPI_GEGLU, 4, 4, false, PK>(pg, vb % fgG, wg);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    CH_STAMP();
    // ---- adaRMS + gate|up + GeGLU
    pg.mod = slot_f;
    if (vb < nG) skinny_rest<EPI_GEGLU, true, 4, 4, 2, true, true, PK>(pg, vb % fgG, vb / fgG, wg, part, red);
    CH_STAMP();
    chain_arrive(c.ctrs, round, nb);
    pd.W = c.wd[l];
    skinny_load_w<EPI_RESID, 16, 1, false, PK>(pd, vb % fgO, wd);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    CH_STAMP();
    // ---- down projection + gated residual
    pd.gate = slot_f + 2 * c.D;
    pd.o_rm = l + 1 == c.depth; pd.o0 = pd.o_rm ? c.x_out : xs;       // the caller's x_out is row-major
    if (vb < nO) skinny_rest<EPI_RESID, false, 16, 1, 1, false, true, PK>(pd, vb % fgO, vb / fgO, wd, part, red);
    CH_STAMP();
    chain_arrive(c.ctrs, round, nb);
    pq.W = c.wqkv[l + 1 < c.depth ? l + 1 : l];     // (the last layer re-reads its own: nobody uses them)
    skinny_load_w<EPI_ROPE, 4, 2, false, PK>(pq, vb % fgQ, wq);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(c.ctrs, round);
    CH_STAMP();
  }
  // a barrier gave up somewhere in this launch: its results are unsynchronised garbage — make them NaN so that no caller can
  // mistake them for actions (the flag stays up for lap_serve_chain_status; Policy.infer falls back to the separate launches)
  if (__hip_atomic_load(c.ctrs + CH_CTR_STRIDE * CH_CTR_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    unsigned short* xo = reinterpret_cast<unsigned short*>(c.x_out);
    for (int j = vb * 512 + (int)threadIdx.x; j < c.M * c.D; j += nb * 512) xo[j] = 0x7fc0;     // bf16 NaN
  }
  // the last block out leaves the counters at zero for the next launch (nobody polls any more: every block is past its last wait)
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(c.ctrs + CH_CTR_STRIDE * CH_CTR_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)nb - 1) {
      for (int j = 0; j <= CH_GROUPS; ++j) __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int j = 0; j < c.attn.B * c.NH * QT; ++j)
        __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * (CH_CTR_HEAD + j), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(c.ctrs + CH_CTR_STRIDE * CH_CTR_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// shapes the chain is built for: the LAP-3B action expert at one chunk of <= 64 action tokens (every stage in one round of 256 blocks)
inline bool chain_ok(int B, int S, int D, int H, int NH, int HD, int NKV, int prefix_len) {
  if (B < 1 || S < 1 || D != 1024 || HD != 256 || NH * HD != 2048 || H != 4096 || NKV != 1 || prefix_len < 0) return false;
  const int M = B * S;
  if (S > 64 || M > 64) return false;
  const int QT = (S + 15) / 16;
  const ServeRuns sr = serve_runs(prefix_len, S, serve_run_cap(B, NH, S));
  return sr.nruns >= 1 && sr.nruns <= 8 && sr.nruns * QT * NH * B <= CH_BLOCKS && B * NH * QT <= CH_MAX_HEADS;
}

// The CURRENT device can hold the chain's 256 blocks at once (one per CU; the kernel's LDS footprint admits one block per CU).
// Queried per device and cached per device id (a process-wide static remembered the first device's answer).
inline bool chain_device_ok() {
  static int cus_of[64];       // 0 = not asked yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (cus_of[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    cus_of[dev] = n > 0 ? n : -1;
  }
  return cus_of[dev] >= CH_BLOCKS;
}

int launch_chain(const ChainP& c, bool packed, hipStream_t s) {
  if (!chain_device_ok()) return LAP_ERR_ARG;       // all 256 blocks must be resident at once: one per CU of an MI355X
  static bool attr = false;
  if (!attr) {
    if (int e = set_lds(serve_chain_kernel<8, false>, SV_LDS)) return e;
    if (int e = set_lds(serve_chain_kernel<8, true>, SV_LDS)) return e;
    attr = true;
  }
  if (c.sr.nruns < 1 || c.sr.nruns > 8) return LAP_ERR_ARG;
  if (packed) hipLaunchKernelGGL((serve_chain_kernel<8, true>), dim3(CH_BLOCKS), dim3(512), SV_LDS, s, c);
  else hipLaunchKernelGGL((serve_chain_kernel<8, false>), dim3(CH_BLOCKS), dim3(512), SV_LDS, s, c);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
