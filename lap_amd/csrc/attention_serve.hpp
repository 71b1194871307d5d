// Few-query attention of the batch-1 denoise step (lap.py:634-667 -> gemma.py:223-272 with a KV cache): <= 64 queries of
// the suffix segment per sample against [cached prefix keys | fresh suffix keys], head size 256.  Included by attention.hip
// after attention_dma.hpp (shares AttnP, DmaCfg<256>, dma_frag_bases, the lane-group reductions).
//
// Built the way the skinny projections are (serve_skinny.hip): with 50 queries there is nothing to pipeline against, so EVERY
// load of a block is issued before anything waits, and the work is cut so that the whole chip takes part (round 4; round 3 gave a
// block one head x ALL queries x 128 keys: 40 blocks of 153 KB, 11.9 us per layer inside the persistent chain):
//   block = (key run, query tile of 16, head, sample), 512 threads; a run is <= 128 consecutive keys of ONE key segment — the
//           cached prefix is cut into runs of equal length (a multiple of 16), the fresh keys get runs of their own (serve_runs);
//           the LAP-3B denoise step (560 cached + 50 fresh keys, 50 queries, 8 heads) is 8 runs x 4 tiles x 8 heads = 256 blocks of
//           <= 80 keys: 80 KB of K / V per block, 8 KB of queries;
//   loads   K and V images (rows x 512 B each, the ring's swizzle) arrive by LDS-DMA pieces issued back to back, the query
//           fragments by direct loads, the keys' info words by one load per thread; ONE wait, ONE barrier;
//   wave    S^T = K Q^T for ONE 16-key tile of the run (8 MFMAs), the run's softmax statistics and the bf16 probabilities cross the
//           waves through LDS (log2 domain), then O^T = V^T P^T for 32 of the head's 256 output columns through transposing reads.
// A run's result (normalised O in f32 + its log-sum-exp) goes to the scratch layout of the generic kernel and the runs of a
// (sample, head, query tile) are merged by attn_serve_combine_body.
constexpr int SV_KEYS = 128;
constexpr int SV_STAT = 2 * SV_KEYS * 512 + SV_KEYS * 4;       // per-wave softmax statistics (max, sum: 2 x [8][16] f32)
constexpr int SV_PROB = SV_STAT + 2 * 8 * 16 * 4;            // probabilities [8 key tiles][64 lanes] x 4 bf16
constexpr int SV_LDS = SV_PROB + 8 * 64 * 8;
constexpr int SV_MAX_RUNS = 16;

__device__ __forceinline__ bf16x4 ds_read_tr_at(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))lds_addr);
}

// Key runs: nr0 runs of rl0 rows over the cached prefix, then nr1 runs of rl1 rows over the fresh keys (the last run of a segment
// may be shorter).  Run lengths are multiples of 16 (an MFMA key tile; a DMA piece is 2 rows) and <= SV_KEYS.
struct ServeRuns { int rl0, nr0, rl1, nr1, nruns; };
// `cap`: the most runs the caller can take.  The fresh keys get as few runs as they need, the prefix — the long side, re-read from
// the cache by every query tile — the rest.  nruns = -1: does not fit.
inline ServeRuns serve_runs(int klen0, int klen1, int cap) {
  ServeRuns s = {16, 0, 16, 0, -1};
  const int min0 = (klen0 + SV_KEYS - 1) / SV_KEYS, min1 = (klen1 + SV_KEYS - 1) / SV_KEYS;
  if (klen0 < 0 || klen1 < 0 || klen0 + klen1 == 0 || min0 + min1 > cap) return s;
  if (klen1 > 0) { s.rl1 = ((klen1 + min1 - 1) / min1 + 15) & ~15; s.nr1 = (klen1 + s.rl1 - 1) / s.rl1; }
  if (klen0 > 0) { const int n0 = cap - min1; s.rl0 = ((klen0 + n0 - 1) / n0 + 15) & ~15; s.nr0 = (klen0 + s.rl0 - 1) / s.rl0; }
  s.nruns = s.nr0 + s.nr1;
  return s;
}
// The cap both callers use (so that the stand-alone launches and the persistent chain cut the keys the same way and stay
// bitwise equal): every block of the attention stage in one round of 256, at most 8 runs.
inline int serve_run_cap(int B, int NH, int qlen1) {
  const int groups = B * NH * ((qlen1 + 15) / 16);
  const int cap = groups > 0 ? 256 / groups : 0;
  return cap > 8 ? 8 : cap;
}

// device-scope (sc1) f32 accesses for results that blocks of other XCDs read inside the same launch (the persistent chain,
// serve_chain.hpp; see serve_skinny_body.hpp on COH)
template <bool COH>
__device__ __forceinline__ void st_f32x4(float* p, const f32x4& v) {
  // (s_nop: a VALU write to the data registers of a > 8-byte store needs 2 wait states behind it; the compiler's hazard
  //  recogniser does not look inside inline assembly — without it a few values per tile were overwritten in flight)
  if constexpr (COH) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool COH>
__device__ __forceinline__ void st_f32(float* p, float v) {
  if constexpr (COH) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}

// One block of the split attention: key run `run` x query tile `qt` of head h of sample b.  smem: SV_LDS bytes [K image | V image |
// info words].
// COH: q / the fresh keys and values were written, and the partial results will be read, by other blocks of the SAME launch.
// PK (the chain, round 4): q and the combined output o are fragment-packed [row tiles of 16][NH HD / 32][1 KiB] (pk_off, common.hpp) —
// the query fragments arrive as whole KiB per wave instruction instead of 16 rows x 64 B.
// PHASE 0: the whole block.  The chain splits it at its grid barrier: PHASE 1 issues the DMA pieces of a run of the CACHED prefix
// (they do not depend on the launch's own results) and returns; PHASE 2, behind the barrier, issues the rest and goes on.
// LST (the tensor-parallel chain, serve_chain_tp.hpp): the run's partial results are read by blocks of the SAME XCD: plain stores.
template <bool COH, int PHASE = 0, bool PK = false, bool LST = false>
__device__ __forceinline__ void attn_run_body(const AttnP& p, const ServeRuns& sr, const int run, const int qt, const int h, const int b, char* smem,
                                              unsigned long long* tclk = nullptr) {      // tclk: tuning aid (thread 0 stamps the 100 MHz clock)
  using C = DmaCfg<256>;
  constexpr int HD = 256, PITCH = C::PITCH, KS = C::KS;
  constexpr int KOFF = 0, VOFF = SV_KEYS * PITCH;
  int* sKw = reinterpret_cast<int*>(smem + 2 * SV_KEYS * PITCH);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));      // (opaque: keeps the chain's layer loop from hoisting this stage's lane arithmetic, serve_skinny_body.hpp)
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int hk = h / (p.NH / p.NKV);
  const int Sq = p.qlen[1], Pn = p.klen[0], Sk = p.klen[1], Tk = Pn + Sk;
  const bool fresh = run >= sr.nr0;                                          // a run of the fresh keys (segment 1)
  const int base = fresh ? (run - sr.nr0) * sr.rl1 : run * sr.rl0;           // its first row inside the segment
  const int nrows = min(fresh ? sr.rl1 : sr.rl0, (fresh ? Sk : Pn) - base);
  const int nt = (nrows + 15) >> 4, nt2 = (nt + 1) & ~1;                      // 16-key tiles with data; tiles the P.V product touches

  // ---- every DMA piece of the block, back to back (piece = 2 rows of 512 B; up to 8 K + 8 V pieces per wave; rows past the run
  // inside the tiles P.V touches are fetched out of range, i.e. as zeros: stale LDS could hold NaN patterns)
  if (PHASE == 1 && fresh) return;
  const int seg = fresh ? 1 : 0;
  const long long kvoff = (long long)b * p.klen[seg] * p.kv_rs[seg] + hk * HD;
  const auto rsK = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[seg] + kvoff), 0, seg_records(p.klen[seg], p.kv_rs[seg], HD), 0x00020000);
  const auto rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[seg] + kvoff), 0, seg_records(p.klen[seg], p.kv_rs[seg], HD), 0x00020000);
  const int rb = p.kv_rs[seg] * 2;
  auto issue = [&](int kv) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pc = w * 8 + j;                       // wave uniform
      if (2 * pc >= nt2 * 16) continue;
      const int row = 2 * pc + (lane >> 5);
      const int col = ((lane & 31) ^ C::swz(row)) << 4;
      char* dst = smem + pc * 1024 + (kv ? VOFF : KOFF);
      const unsigned off = row < nrows ? (unsigned)((base + row) * rb + col) : DMA_OOB;
      // (fresh rows inside the chain: written by other XCDs a moment ago)
      if (COH && fresh) __builtin_amdgcn_raw_ptr_buffer_load_lds(kv ? rsV : rsK, (LDS_PTR(void))dst, 16, off, 0, 0, 16);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(kv ? rsV : rsK, (LDS_PTR(void))dst, 16, off, 0, 0, 0);
    }
  };
  if (PHASE == 1) {
    issue(0);
    issue(1);
    return;
  }
  // Inside the chain the fresh keys / values were written by other XCDs a moment ago, and the DMA path does not honour the
  // device-scope bit the way register loads do (stale lines of this XCD's L2 were observed).  Round 3 dropped them with an
  // agent-scope acquire fence in front of the DMA; with one block per (run, query tile, head) that is 32 fences on one XCD and
  // the fresh runs' loads took 15.5 us (tools/probes/chain_clock.py).  So these rows (<= 64 of them: chain_ok) come through
  // REGISTERS — device-scope loads like every other activation of the chain — and are written to the DMA's LDS positions.
  const bool via_regs = COH && fresh;
  u32x4 kr[4], vr[4];
  if (via_regs) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = w * 4 + j, row = 2 * pc + (lane >> 5);
      const int col = ((lane & 31) ^ C::swz(row)) << 4;
      const unsigned off = (2 * pc < nt2 * 16 && row < nrows) ? (unsigned)((base + row) * rb + col) : DMA_OOB;
      kr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsK, off, 0, 16);
      vr[j] = __builtin_amdgcn_raw_buffer_load_b128(rsV, off, 0, 16);
    }
  } else if (PHASE == 0 || fresh) {
    issue(0);
    issue(1);
  }
  // ---- my query row, its info word, and the keys' info words (in flight together with the DMA)
  const int myq = qt * 16 + i;
  const bool vq = myq < Sq;
  bf16x8 qf[KS];
  if constexpr (COH) {
    if constexpr (PK) {
      const int row = b * Sq + myq, QK = p.q_rs[1];
      const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.q[1], 0, (unsigned)(((p.B * Sq + 15) >> 4) * QK * 32), 0x00020000);
      const unsigned qoff = vq ? (unsigned)((((long long)(row >> 4) * (QK >> 5) + h * (HD >> 5)) * 64 + g * 16 + (row & 15)) * 16) : DMA_OOB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + kk * 1024, 0, 16));
    } else {
      const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.q[1], 0, (unsigned)((long long)p.B * Sq * p.q_rs[1] * 2), 0x00020000);
      const unsigned qoff = vq ? (unsigned)((((long long)b * Sq + myq) * p.q_rs[1] + h * HD + g * 8) * 2) : DMA_OOB;   // invalid rows read as zeros
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + kk * 64, 0, 16));
    }
  } else {
    load_row_frags<HD>(p.q[1] + (b * (long long)Sq + myq) * p.q_rs[1] + h * HD, vq, lane, qf);
  }
  const int qi = !vq ? 0 : (p.qinfo ? p.qinfo[(long long)b * Sq + myq] : 0x7fffffff);
  const int qcls = qi >> 24, qidx = qi & 0xffffff;
  if (tid < SV_KEYS) {
    const int joint = tid < nrows ? (fresh ? Pn : 0) + base + tid : -1;       // index into the sample's [prefix | fresh] key list
    sKw[tid] = joint < 0 ? 0 : (p.kinfo ? p.kinfo[(long long)b * Tk + joint] : 0x7f000000);
  }
  if (via_regs) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = w * 4 + j;
      if (2 * pc < nt2 * 16) {
        *reinterpret_cast<u32x4*>(smem + KOFF + pc * 1024 + lane * 16) = kr[j];
        *reinterpret_cast<u32x4*>(smem + VOFF + pc * 1024 + lane * 16) = vr[j];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // a bare barrier behind it: images, queries, info words are here
  if (tclk && tid == 0) tclk[0] = wall_clock64();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (tclk && tid == 0) tclk[1] = wall_clock64();

  // ---- S^T = K Q^T: wave t owns key tile t of the run (16 keys x 16 queries, 8 MFMAs); lane owns query column i and keys
  // 16 t + 4 g + r.  The run's softmax statistics and the probabilities cross the waves through LDS (two block barriers) — round
  // 4: every wave used to compute the scores of ALL keys for its own columns of P.V (8 x the MFMAs and 8 x the K image through the
  // LDS port: 3.1 us of the stage, tools/probes/chain_clock.py).
  float* sMax = reinterpret_cast<float*>(smem + SV_STAT);          // [8 waves][16 queries]
  float* sSum = sMax + 8 * 16;
  bf16x4* sP = reinterpret_cast<bf16x4*>(smem + SV_PROB);          // [8 key tiles][64 lanes]: the C/D registers of S^T as the B operand of P.V
  const char* kp[C::KREGS];
  unsigned va_unused[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va_unused);
  const float c2 = p.scale * LOG2E;
  f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
  bool ok[4] = {false, false, false, false};
  float mx = NEG_BIG;
  if (w < nt) {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      sc = mfma16(*reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + w * 16 * PITCH + kimm<HD>(kk)), qf[kk], sc);
    const i32x4 kw = *reinterpret_cast<const i32x4*>(sKw + w * 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ok[r] = (qcls & (kw[r] >> 24)) != 0 && (kw[r] & 0xffffff) <= qidx;
      if (ok[r]) mx = fmaxf(mx, sc[r]);
    }
  }
  mx = max_over_groups(mx);
  if (g == 0) sMax[w * 16 + i] = mx;
  __syncthreads();
  float m = NEG_BIG;
#pragma unroll
  for (int t = 0; t < 8; ++t) m = fmaxf(m, sMax[t * 16 + i]);
  m *= c2;
  float lw = 0.f;
  bf16x4 pr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float e = ok[r] ? __builtin_amdgcn_exp2f(sc[r] * c2 - m) : 0.f;
    lw += e;
    pr[r] = f2bf(e);
  }
  lw = sum_over_groups(lw);
  if (g == 0) sSum[w * 16 + i] = lw;
  sP[w * 64 + lane] = pr;
  __syncthreads();
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) l += sSum[t * 16 + i];
  // ---- O^T = V^T P^T for this wave's two column fragments (columns 32 w .. 32 w + 31)
  const int tr_row = 4 * g + (i >> 2);
  const unsigned lane_swz = (unsigned)(C::swz(tr_row) ^ ((i & 3) >> 1));
  const unsigned lane_off = lds_addr_of(smem) + (unsigned)(VOFF + tr_row * PITCH + ((i & 1) << 3));
  f32x4 acc[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (2 * c < nt) {
      const bf16x8 pb = join8(sP[(2 * c) * 64 + lane], sP[(2 * c + 1) * 64 + lane]);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const unsigned a0 = lane_off + (((unsigned)(2 * (2 * w + d)) ^ lane_swz) << 4) + (unsigned)(c * 32 * PITCH);     // (LDS byte addresses)
        acc[d] = mfma16(join8(ds_read_tr_at(a0), ds_read_tr_at(a0 + 16 * PITCH)), pb, acc[d]);
      }
    }
  }
  if (tclk && tid == 0) tclk[2] = wall_clock64();
  if (!vq) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const long long row = ((long long)run * p.B + b) * Sq + myq;
  float* op = p.part + (row * p.NH + h) * HD + w * 32;
#pragma unroll
  for (int d = 0; d < 2; ++d) st_f32x4<COH && !LST>(op + d * 16 + 4 * g, acc[d] * inv);
  if (w == 0 && g == 0) st_f32<COH && !LST>(p.lpart + (((long long)run * p.B + b) * p.NH + h) * Sq + myq, l > 0.f ? (m + __builtin_amdgcn_logf(l)) * LN2 : NEG_BIG);
}

// grid = (runs, query tiles, heads x samples)
__global__ __launch_bounds__(512) void attn_serve_kernel(AttnP p, ServeRuns sr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_run_body<false>(p, sr, blockIdx.x, blockIdx.y, blockIdx.z % p.NH, blockIdx.z / p.NH, smem);
}

// O = sum_i exp(lse_i - lse) O_i over the NS key splits (the generic combine's arithmetic, every load issued up front).
// One thread per (b, q, h, 4 d).
template <int NS, bool COH, bool PK = false, bool LST = false>
__device__ __forceinline__ void attn_serve_combine_body(const AttnP& p, const long long gid) {
  constexpr int HD = 256;
  const int S = p.qlen[1];
  const long long n4 = (long long)p.B * S * p.NH * HD / 4;
  if (gid >= n4) return;
  const int d0 = (int)(gid % (HD / 4)) * 4;
  long long r = gid / (HD / 4);
  const int h = (int)(r % p.NH); r /= p.NH;
  const int t = (int)(r % S);
  const int b = (int)(r / S);
  float li[NS];
  f32x4 oi[NS];
  if constexpr (COH) {
    const unsigned nl = (unsigned)((long long)p.nsplit * p.B * p.NH * S * 4), np4 = (unsigned)((long long)p.nsplit * p.B * S * p.NH * HD * 4);
    const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)p.lpart, 0, nl, 0x00020000);
    const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, np4, 0x00020000);
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {     // splits past nsplit: out of range, read as zeros and get NEG_BIG below
      const unsigned lo = sp < p.nsplit ? (unsigned)(((((long long)sp * p.B + b) * p.NH + h) * S + t) * 4) : DMA_OOB;
      const unsigned po = sp < p.nsplit ? (unsigned)((((((long long)sp * p.B + b) * S + t) * p.NH + h) * HD + d0) * 4) : DMA_OOB;
      li[sp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, lo, 0, 16));
      oi[sp] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, po, 0, 16));
    }
#pragma unroll
    for (int sp = 0; sp < NS; ++sp)
      if (sp >= p.nsplit) li[sp] = NEG_BIG;
  } else {
#pragma unroll
    for (int sp = 0; sp < NS; ++sp) {
      li[sp] = sp < p.nsplit ? p.lpart[(((long long)sp * p.B + b) * p.NH + h) * S + t] : NEG_BIG;
      oi[sp] = sp < p.nsplit ? *reinterpret_cast<const f32x4*>(p.part + ((((long long)sp * p.B + b) * S + t) * p.NH + h) * HD + d0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float mx = NEG_BIG;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) mx = fmaxf(mx, li[sp]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    if (li[sp] <= NEG_BIG * 0.5f) continue;
    const float wgt = __expf(li[sp] - mx);
    den += wgt;
    acc += oi[sp] * wgt;
  }
  bf16* dst = PK ? p.o[1] + pk_off((int)(b * (long long)S + t), h * HD + d0, p.o_rs[1]) : p.o[1] + (b * (long long)S + t) * p.o_rs[1] + h * HD + d0;
  if constexpr (COH && !LST) {
    const float sc = den > 0.f ? 1.0f / den : 0.f;
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[e] * sc);
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, o);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(bits) : "memory");
  } else {
    store4(dst, acc, den > 0.f ? 1.0f / den : 0.f);
  }
}

template <int NS>
__global__ __launch_bounds__(256) void attn_serve_combine_kernel(AttnP p) {
  attn_serve_combine_body<NS, false>(p, (long long)blockIdx.x * 256 + threadIdx.x);
}

int launch_serve(const AttnP& p, hipStream_t s) {
  const ServeRuns sr = serve_runs(p.klen[0], p.klen[1], p.nsplit);
  if (sr.nruns < 1 || sr.nruns > SV_MAX_RUNS || p.klen[1] < 0) return LAP_ERR_ARG;
  AttnP q = p;
  q.nsplit = sr.nruns;          // (p.nsplit was the cap: what the scratch was sized for)
  static bool attr = false;
  if (!attr) {
    if (int e = set_lds(attn_serve_kernel, SV_LDS)) return e;
    attr = true;
  }
  hipLaunchKernelGGL(attn_serve_kernel, dim3(sr.nruns, (p.qlen[1] + 15) / 16, p.NH * p.B), dim3(512), SV_LDS, s, q, sr);
  LAP_CHECK_LAUNCH();
  const long long n4 = (long long)p.B * p.qlen[1] * p.NH * 256 / 4;
  const dim3 grid((unsigned)((n4 + 255) / 256));
  if (sr.nruns <= 4) hipLaunchKernelGGL(attn_serve_combine_kernel<4>, grid, dim3(256), 0, s, q);
  else if (sr.nruns <= 8) hipLaunchKernelGGL(attn_serve_combine_kernel<8>, grid, dim3(256), 0, s, q);
  else hipLaunchKernelGGL(attn_serve_combine_kernel<16>, grid, dim3(256), 0, s, q);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
