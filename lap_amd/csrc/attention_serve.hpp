// Few-query attention of the batch-1 denoise step (lap.py:634-667 -> gemma.py:223-272 with a KV cache): <= 64 queries of
// the suffix segment per sample against [cached prefix keys | fresh suffix keys], head size 256.  Included by attention.hip
// after attention_dma.hpp (shares AttnP, DmaCfg<256>, dma_frag_bases, the lane-group reductions).
//
// The generic LDS-DMA kernel walks its share of the keys as a chain of 32-row tiles (one barrier and one DMA round trip
// per tile) behind a prologue that stages info words and tile summaries: 12.8 us per launch for 125 KB of K / V per block
// in the denoise step.  With 50 queries there is nothing to pipeline against, so this kernel is built the way the skinny
// projections are (serve_skinny.hip): EVERY load of the block is issued before anything waits.
//   grid  = (key splits, query heads, samples), 512 threads;
//   block = 128 keys: K and V images (128 x 512 B each, the ring's swizzle) arrive by 128 LDS-DMA pieces issued back to
//           back, the query fragments by direct loads, the keys' info words by one load per thread; ONE wait, ONE barrier;
//   wave  = (query tile of 16, half of the head's 256 output columns): S^T = K Q^T for all 128 keys (64 MFMAs, done by
//           both halves), full-row softmax over the block's keys in the log2 domain, then O^T = V^T P^T for its 128 columns
//           (128 MFMAs through transposing reads).
// A split's result (normalised O in f32 + its log-sum-exp) goes to the scratch layout of the generic kernel and is merged
// by attn_serve_combine_kernel.  Key splits never straddle the two key segments' buffers inside one DMA piece: the prefix
// is cut into runs of 128 rows, and the fresh keys start at an even row of the last, partly filled prefix split when they
// fit there (560 + 50 keys: 4 full splits + one of 48 | 50 rows), otherwise they get a split of their own.
constexpr int SV_KEYS = 128;
constexpr int SV_LDS = 2 * SV_KEYS * 512 + SV_KEYS * 4;

__device__ __forceinline__ bf16x4 ds_read_tr_at(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))lds_addr);
}

struct ServeSplits { int nfull, rem, nsplit, sfx_split, sfx_row0; };
inline ServeSplits serve_splits(int klen0, int klen1) {
  ServeSplits s;
  s.nfull = klen0 / SV_KEYS;
  s.rem = klen0 % SV_KEYS;
  const int rem_pad = (s.rem + 1) & ~1;
  const bool joint = s.rem > 0 && klen1 > 0 && rem_pad + klen1 <= SV_KEYS;
  s.sfx_split = klen1 > 0 ? (joint ? s.nfull : s.nfull + (s.rem > 0 ? 1 : 0)) : -1;
  s.sfx_row0 = joint ? rem_pad : 0;
  s.nsplit = s.nfull + (s.rem > 0 ? 1 : 0) + ((klen1 > 0 && !joint) ? 1 : 0);
  return s;
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

// One block of the split attention: key split s of head h of sample b.  smem: SV_LDS bytes [K image | V image | info words].
// COH: q / the fresh keys and values were written, and the partial results will be read, by other blocks of the SAME launch.
// PHASE 0: the whole block.  The chain splits it at its grid barrier: PHASE 1 issues the DMA pieces that read the CACHED prefix
// only (they do not depend on the launch's own results) and returns; PHASE 2, behind the barrier, issues the fresh rows and goes on.
// PK (the chain, round 4): q and the combined output o are fragment-packed [row tiles of 16][NH HD / 32][1 KiB] (pk_off,
// serve_skinny_body.hpp) — the query fragments arrive as whole KiB per wave instruction instead of 16 rows x 64 B.
template <bool COH, int PHASE = 0, bool PK = false>
__device__ __forceinline__ void attn_serve_body(const AttnP& p, const ServeSplits& sp, const int s, const int h, const int b, char* smem) {
  using C = DmaCfg<256>;
  constexpr int HD = 256, PITCH = C::PITCH, KS = C::KS;
  constexpr int KOFF = 0, VOFF = SV_KEYS * PITCH;
  int* sKw = reinterpret_cast<int*>(smem + 2 * SV_KEYS * PITCH);
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));      // (opaque: keeps the chain's layer loop from hoisting this stage's lane arithmetic, serve_skinny_body.hpp)
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int hk = h / (p.NH / p.NKV);
  const int S = p.qlen[1], Pn = p.klen[0], Tk = p.klen[0] + p.klen[1];
  const int np = s < sp.nfull ? SV_KEYS : (s == sp.nfull ? sp.rem : 0);     // prefix rows of this split: [0, np)
  const int pbase = s * SV_KEYS;
  const bool has_sfx = s == sp.sfx_split;                                    // suffix rows: [sfx_row0, sfx_row0 + S)
  const int srow0 = sp.sfx_row0;

  // ---- every DMA piece of the block, back to back (piece = 2 rows of 512 B; 8 K + 8 V pieces per wave)
    const long long kvoff0 = (long long)b * p.klen[0] * p.kv_rs[0] + hk * HD, kvoff1 = (long long)b * p.klen[1] * p.kv_rs[1] + hk * HD;
    const auto rsK0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
    const auto rsV0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
    const auto rsK1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
    const auto rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
    const int rb0 = p.kv_rs[0] * 2, rb1 = p.kv_rs[1] * 2;
    // K pieces now; the V pieces are issued behind the query / info loads (below), so that `vmcnt(8)` — everything but the 8 youngest
    // operations — means "K image, queries and info words are here" and the scores + softmax run while the V image is still landing
    auto issue = [&](int kv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pc = w * 8 + j;                       // wave uniform
        const int row = 2 * pc + (lane >> 5);
        const int col = ((lane & 31) ^ C::swz(row)) << 4;
        char* dst = smem + pc * 1024 + (kv ? VOFF : KOFF);
        if (has_sfx && 2 * pc >= srow0) {               // rows past the segment end read as zeros (num_records)
          if (PHASE == 1) continue;
          const unsigned off = (unsigned)((row - srow0) * rb1 + col);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(kv ? rsV1 : rsK1, (LDS_PTR(void))dst, 16, off, 0, 0, COH ? 16 : 0);
        } else {
          if (PHASE == 2) continue;
          const unsigned off = row < np ? (unsigned)((pbase + row) * rb0 + col) : DMA_OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(kv ? rsV0 : rsK0, (LDS_PTR(void))dst, 16, off, 0, 0, 0);
        }
      }
    };
    if (PHASE == 1) {
      issue(0);
      issue(1);
      return;
    }
    // the fresh keys / values were written by other XCDs a moment ago and the DMA path does not honour the device-scope bit the
    // way register loads do (stale lines of this XCD's L2 were observed): drop them first
    if (COH && has_sfx) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    issue(0);
  // ---- my query row, its info word, and the keys' info words (in flight together with the DMA)
  const int qt = w & 3, dh = w >> 2;
  const int myq = qt * 16 + i;
  const bool vq = myq < S;
  bf16x8 qf[KS];
  if constexpr (COH) {
    if constexpr (PK) {
      const int row = b * S + myq, QK = p.q_rs[1];
      const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.q[1], 0, (unsigned)(((p.B * S + 15) >> 4) * QK * 32), 0x00020000);
      const unsigned qoff = vq ? (unsigned)((((long long)(row >> 4) * (QK >> 5) + h * (HD >> 5)) * 64 + g * 16 + (row & 15)) * 16) : DMA_OOB;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + kk * 1024, 0, 16));
    } else {
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)p.q[1], 0, (unsigned)((long long)p.B * S * p.q_rs[1] * 2), 0x00020000);
    const unsigned qoff = vq ? (unsigned)((((long long)b * S + myq) * p.q_rs[1] + h * HD + g * 8) * 2) : DMA_OOB;   // invalid rows read as zeros
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) qf[kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsQ, qoff + kk * 64, 0, 16));
    }
  } else {
    load_row_frags<HD>(p.q[1] + (b * (long long)S + myq) * p.q_rs[1] + h * HD, vq, lane, qf);
  }
  const int qi = !vq ? 0 : (p.qinfo ? p.qinfo[(long long)b * S + myq] : 0x7fffffff);
  const int qcls = qi >> 24, qidx = qi & 0xffffff;
  int kword = 0;
  if (tid < SV_KEYS) {
    const int r = tid;
    int joint = -1;                                   // index into the sample's [prefix | suffix] key list
    if (has_sfx && r >= srow0) { if (r - srow0 < S) joint = Pn + r - srow0; }
    else if (r < np) joint = pbase + r;
    kword = joint < 0 ? 0 : (p.kinfo ? p.kinfo[(long long)b * Tk + joint] : 0x7f000000);
  }
  issue(1);
  if (tid < SV_KEYS) sKw[tid] = kword;
  // all but the 8 V pieces just issued (vector memory operations return in order) + my info word in LDS; a bare barrier: the
  // fence of __syncthreads() would drain the V pieces as well
  if (PHASE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (the prefix images have been landing since before the barrier)
  else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- S^T = K Q^T over the block's 128 keys; lane owns query column i and keys 16 t + 4 g + r
  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);
  f32x4 sc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
      sc[t] = mfma16(*reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + t * 16 * PITCH + kimm<HD>(kk)), qf[kk], sc[t]);
  }
  const float c2 = p.scale * LOG2E;
  bool ok[8][4];
  float mx = NEG_BIG;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const i32x4 kw = *reinterpret_cast<const i32x4*>(sKw + t * 16 + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ok[t][r] = (qcls & (kw[r] >> 24)) != 0 && (kw[r] & 0xffffff) <= qidx;
      if (ok[t][r]) mx = fmaxf(mx, sc[t][r]);
    }
  }
  const float m = max_over_groups(mx) * c2;
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[t][r] = ok[t][r] ? __builtin_amdgcn_exp2f(sc[t][r] * c2 - m) : 0.f;
      l += sc[t][r];
    }
  l = sum_over_groups(l);
  wait_vm0();          // the V image
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // ---- O^T = V^T P^T for this wave's 8 column fragments (columns 128 dh .. 128 dh + 127)
  f32x4 acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const bf16x8 pb = pack8(sc[2 * c], sc[2 * c + 1]);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const unsigned a0 = va[d] + (unsigned)(VOFF + dh * 256 + c * 32 * PITCH);     // (va holds LDS byte addresses)
      acc[d] = mfma16(join8(ds_read_tr_at(a0), ds_read_tr_at(a0 + 16 * PITCH)), pb, acc[d]);
    }
  }
  if (!vq) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const long long row = ((long long)s * p.B + b) * S + myq;
  float* op = p.part + (row * p.NH + h) * HD + dh * 128;
#pragma unroll
  for (int d = 0; d < 8; ++d) st_f32x4<COH>(op + d * 16 + 4 * g, acc[d] * inv);
  if (dh == 0 && g == 0) st_f32<COH>(p.lpart + (((long long)s * p.B + b) * p.NH + h) * S + myq, l > 0.f ? (m + __builtin_amdgcn_logf(l)) * LN2 : NEG_BIG);
}

__global__ __launch_bounds__(512) void attn_serve_kernel(AttnP p, ServeSplits sp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_serve_body<false>(p, sp, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// O = sum_i exp(lse_i - lse) O_i over the NS key splits (the generic combine's arithmetic, every load issued up front).
// One thread per (b, q, h, 4 d).
template <int NS, bool COH, bool PK = false>
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
  if constexpr (COH) {
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
  const ServeSplits sp = serve_splits(p.klen[0], p.klen[1]);
  if (sp.nsplit != p.nsplit || sp.nsplit > 16) return LAP_ERR_ARG;
  static bool attr = false;
  if (!attr) {
    if (int e = set_lds(attn_serve_kernel, SV_LDS)) return e;
    attr = true;
  }
  hipLaunchKernelGGL(attn_serve_kernel, dim3(sp.nsplit, p.NH, p.B), dim3(512), SV_LDS, s, p, sp);
  LAP_CHECK_LAUNCH();
  const long long n4 = (long long)p.B * p.qlen[1] * p.NH * 256 / 4;
  const dim3 grid((unsigned)((n4 + 255) / 256));
  if (sp.nsplit <= 4) hipLaunchKernelGGL(attn_serve_combine_kernel<4>, grid, dim3(256), 0, s, p);
  else if (sp.nsplit <= 8) hipLaunchKernelGGL(attn_serve_combine_kernel<8>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(attn_serve_combine_kernel<16>, grid, dim3(256), 0, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
