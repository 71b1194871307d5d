// ROUND-6 EXPERIMENT, NOT PART OF THE LIBRARY (measured slower than the streaming kernels: profiles/r06_siglip_attention_head_resident_ab.txt, docs/EXPERIMENTS.md J).
// To try it: copy next to csrc/attention_dma.hpp, #include it behind that header inside attention.hip and route launch_fwd / launch_bwd<72> through
// launch_fwd_res / launch_bwd_res where res_path_ok holds.

// Head-resident attention kernels for short, unmasked sequences with a small head (SigLIP So400m/14: 256 tokens per image, 16 heads of
// 72: Flax MultiHeadDotProductAttention of siglip_gemma3.py:59-114).  Included by attention.hip behind attention_dma.hpp, whose LDS tile
// image, DMA pieces, fragment addresses and per-tile arithmetic these kernels reuse.
//
// The streaming kernels (attention_dma.hpp) cut such a head into 4 query tiles x 8 key tiles: a block walks 8 tiles of 32 rows with one
// barrier + one DMA wait per tile and only 11 MFMAs per wave in between — barrier latency, not work (forward 0.25 PF, rocprofv3 round 6:
// 76 us forward, 117 + 97 us backward per SigLIP block at B = 32).  Here ONE block owns one (image, head): all of K and V (forward, dQ) or
// Q and dO (dK / dV) are fetched into LDS once (8 x [tile | tile] = 128 KiB, one barrier), and every wave walks the 8 resident tiles with
// FOUR 16-row sub-tiles of the other side in registers — 44 / 68 / 88 MFMAs per tile and wave, no barrier inside the loop.
//
// Bitwise contract: wave w's sub-tile u holds rows 64 u + 16 w + i — exactly the 16 rows wave w of the streaming kernel's block u holds —
// and does to them what that wave does (same k-steps, same lazy running maximum with its wave-wide vote, same order of key / query tiles),
// so outputs, log-sum-exp, delta and all three gradients equal the streaming kernels' bit for bit (tests/test_kernels_gpu.py).
//
// Eligibility (res_path_ok): one segment, no info words (every query sees every key), Tq % 64 == 0, Tk % 64 == 0, both <= 256, one kv head
// per query head, no key split.

constexpr int RES_NQ = 4;          // 16-row sub-tiles per wave (64 rows per wave, 256 per block)
constexpr int RES_MAX_ROWS = 256;

bool res_path_ok(const AttnP& p, int hd) {
  return attn_variant() != 3 && attn_variant() != 0 && hd == 72 && !p.qinfo && !p.kinfo && p.qlen[1] == 0 && p.klen[1] == 0 && p.qlen[0] > 0 && p.klen[0] > 0 &&
         (p.qlen[0] & 63) == 0 && (p.klen[0] & 63) == 0 && p.qlen[0] <= RES_MAX_ROWS && p.klen[0] <= RES_MAX_ROWS && p.NH == p.NKV && p.nsplit == 1 &&
         p.hsplit <= 1 && p.scale > 0.f && !p.stop;
}

// all `nt` 32-row tiles of two row-major operands (row strides in elements) into LDS: tile t at smem + t * 2 * TILE (A) and + TILE (B)
template <int HD>
__device__ __forceinline__ void res_fetch_pair(char* smem, const bf16* a, int stride_a, const bf16* b2, int stride_b, int rows, int nt, int w, int lane) {
  using C = DmaCfg<HD>;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, seg_records(rows, stride_a, HD), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)b2, 0, seg_records(rows, stride_b, HD), 0x00020000);
  const int rba = stride_a * 2, rbb = stride_b * 2;
  int dma_row[C::PIECES], dma_col[C::PIECES];
#pragma unroll
  for (int j = 0; j < C::PIECES; ++j) dma_lane<HD>(w, j, lane, dma_row[j], dma_col[j]);
  for (int t = 0; t < nt; ++t) {
    char* base = smem + t * 2 * C::TILE;
#pragma unroll
    for (int j = 0; j < C::PIECES; ++j) {
      const unsigned offa = dma_col[j] < 0 ? DMA_OOB : (unsigned)((t * 32 + dma_row[j]) * rba + dma_col[j]);
      const unsigned offb = dma_col[j] < 0 ? DMA_OOB : (unsigned)((t * 32 + dma_row[j]) * rbb + dma_col[j]);
      char* dst = base + (w * C::PIECES + j) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))dst, 16, offa, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(dst + C::TILE), 16, offb, 0, 0, 0);
    }
  }
}

// the DF transposing fragment pairs of the tile at LDS byte offset `off` (runtime) in one burst + fence
template <int HD>
__device__ __forceinline__ void res_tr_burst(const unsigned (&va)[DmaCfg<HD>::VREGS], unsigned off, bf16x4 (&r)[2 * DmaCfg<HD>::DF]) {
#pragma unroll
  for (int d = 0; d < DmaCfg<HD>::DF; ++d) {
    r[2 * d] = ds_read_tr_raw<0>(va[d] + off);
    r[2 * d + 1] = ds_read_tr_raw<16 * DmaCfg<HD>::PITCH>(va[d] + off);
  }
  lds_wait_all();
#pragma unroll
  for (int d = 0; d < 2 * DmaCfg<HD>::DF; ++d) lds_tie(r[d]);
}

// ======================================================================== forward (MODE 0) and backward dQ (MODE 1)
template <int HD, int MODE>
__global__ __launch_bounds__(256, 1) void attn_res_q_kernel(AttnP p) {
  using C = DmaCfg<HD>;
  constexpr int KS = C::KS, DF = C::DF, TILE = C::TILE, PITCH = C::PITCH, NQ = RES_NQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // nt x [K tile | V tile]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Tq = p.qlen[0], Tk = p.klen[0], nt = Tk >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int h = bid % p.NH, b = bid / p.NH;
  const int nq = Tq >> 6;                                        // live sub-tiles (wave uniform)

  const long long kvoff = (long long)b * Tk * p.kv_rs[0] + h * HD;
  res_fetch_pair<HD>(smem, p.k[0] + kvoff, p.kv_rs[0], p.v[0] + kvoff, p.kv_rs[0], Tk, nt, w, lane);

  // ---- my query rows: sub-tile u = rows 64 u + 16 w + i
  bf16x8 qf[NQ][KS], dof[NQ][KS];
  float lse2[NQ], dl_q[NQ];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    const bool vq = u < nq;
    const long long row = (long long)b * Tq + u * 64 + w * 16 + i;
    load_row_frags<HD>(p.q[0] + row * p.q_rs[0] + h * HD, vq, lane, qf[u]);
    lse2[u] = 0.f; dl_q[u] = 0.f;
    if constexpr (MODE == 1) {
      load_row_frags<HD>(p.d_o[0] + row * p.o_rs[0] + h * HD, vq, lane, dof[u]);
      const long long srow = ((long long)b * p.NH + h) * Tq + u * 64 + w * 16 + i;
      lse2[u] = (vq ? p.lse[srow] : LSE_EMPTY) * LOG2E;
      if (p.fuse_delta) {
        bf16x8 of[KS];
        load_row_frags<HD>(p.o[0] + row * p.o_rs[0] + h * HD, vq, lane, of);
        float a = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
          for (int e = 0; e < 8; ++e) a += (float)dof[u][kk][e] * (float)of[kk][e];
        dl_q[u] = sum_over_groups(a);
        if (vq && g == 0) p.delta[srow] = dl_q[u];
      } else {
        dl_q[u] = vq ? p.delta[srow] : 0.f;
      }
    }
  }

  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);

  const float c2 = p.scale * LOG2E;
  float m[NQ], l[NQ];
  f32x4 acc_o[NQ][DF];
#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    m[u] = NEG_BIG; l[u] = 0.f;
#pragma unroll
    for (int d = 0; d < DF; ++d) acc_o[u][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  wait_vm0();          // my pieces of every tile (and my rows) have landed ...
  __syncthreads();     // ... and everybody's

  for (int t = 0; t < nt; ++t) {
    const unsigned koff = (unsigned)(t * 2 * TILE), voff = koff + TILE;
    bf16x8 k0[KS], k1[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      k0[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + koff + kimm<HD>(kk));
      k1[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + koff + 16 * PITCH + kimm<HD>(kk));
    }
    bf16x8 pb[NQ];
    if constexpr (MODE == 1) {
      bf16x8 v0[KS], v1[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        v0[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + voff + kimm<HD>(kk));
        v1[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + voff + 16 * PITCH + kimm<HD>(kk));
      }
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        if (u >= nq) continue;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          s0 = mfma16(k0[kk], qf[u][kk], s0);      // S^T[key][q]
          s1 = mfma16(k1[kk], qf[u][kk], s1);
          d0 = mfma16(v0[kk], dof[u][kk], d0);     // dP^T[key][q]
          d1 = mfma16(v1[kk], dof[u][kk], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - lse2[u]) * (d0[r] - dl_q[u]) * p.scale;
          d1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - lse2[u]) * (d1[r] - dl_q[u]) * p.scale;
        }
        pb[u] = pack8(d0, d1);
      }
      // dQ^T += K^T dS^T
      bf16x4 vr[2 * DF];
      res_tr_burst<HD>(va, koff, vr);
#pragma unroll
      for (int u = 0; u < NQ; ++u) {
        if (u >= nq) continue;
#pragma unroll
        for (int d = 0; d < DF; ++d) acc_o[u][d] = mfma16(join8(vr[2 * d], vr[2 * d + 1]), pb[u], acc_o[u][d]);
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      if (u >= nq) continue;
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        s0 = mfma16(k0[kk], qf[u][kk], s0);
        s1 = mfma16(k1[kk], qf[u][kk], s1);
      }
      // online softmax in the log2 domain with the streaming kernel's lazy running maximum (attention_dma.hpp), unmasked branch
      const float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      const float m_new = fmaxf(m[u], max_over_groups(mx) * c2);
      if (__any(m_new > m[u] + 8.0f)) {
        const float alpha = __builtin_amdgcn_exp2f(m[u] - m_new);
        l[u] *= alpha;
#pragma unroll
        for (int d = 0; d < DF; ++d) acc_o[u][d] *= alpha;
        m[u] = m_new;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - m[u]);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - m[u]);
      }
      l[u] += ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
      pb[u] = pack8(s0, s1);
    }
    // O^T += V^T P^T
    bf16x4 vr[2 * DF];
    res_tr_burst<HD>(va, voff, vr);
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
      if (u >= nq) continue;
#pragma unroll
      for (int d = 0; d < DF; ++d) acc_o[u][d] = mfma16(join8(vr[2 * d], vr[2 * d + 1]), pb[u], acc_o[u][d]);
    }
  }

#pragma unroll
  for (int u = 0; u < NQ; ++u) {
    if (u >= nq) continue;
    const long long row = (long long)b * Tq + u * 64 + w * 16 + i;
    if constexpr (MODE == 1) {
      bf16* dqrow = p.dq[0] + row * p.q_rs[0] + h * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d)
        if (d * 16 + 4 * g < HD) store4(dqrow + d * 16 + 4 * g, acc_o[u][d], 1.0f);
    } else {
      const float lt = sum_over_groups(l[u]);
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
      const float lse = (m[u] + __builtin_amdgcn_logf(lt)) * LN2;   // natural log-sum-exp of scale * q.k
      bf16* orow = p.o[0] + row * p.o_rs[0] + h * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d)
        if (d * 16 + 4 * g < HD) store4(orow + d * 16 + 4 * g, acc_o[u][d], inv);
      if (p.lse && g == 0) p.lse[((long long)b * p.NH + h) * Tq + u * 64 + w * 16 + i] = lt > 0.f ? lse : LSE_EMPTY;
    }
  }
}

// ======================================================================== backward: dK, dV
// Block = (image, head): all Q and dO tiles of the head resident, log-sum-exp and delta of its rows next to them; wave w's sub-tile u owns
// keys 64 u + 16 w + i with their K and V rows in registers.  Per query tile:  S = Q K^T, dP = dO V^T,  P = exp2(S c2 - lse),
// dS = P o (dP - delta) scale,  dV^T += dO^T P,  dK^T += Q^T dS — attn_dma_kv_kernel's step for every sub-tile.
template <int HD>
__global__ __launch_bounds__(256, 1) void attn_res_kv_kernel(AttnP p) {
  using C = DmaCfg<HD>;
  constexpr int KS = C::KS, DF = C::DF, TILE = C::TILE, PITCH = C::PITCH, NQ = RES_NQ;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // nt x [Q tile | dO tile], then lse[Tq], delta[Tq]
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Tq = p.qlen[0], Tk = p.klen[0], nt = Tq >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int h = bid % p.NH, b = bid / p.NH;
  const int nk = Tk >> 6;                                        // live key sub-tiles (wave uniform)
  float* sL = reinterpret_cast<float*>(smem + nt * 2 * TILE);
  float* sX = sL + RES_MAX_ROWS;

  res_fetch_pair<HD>(smem, p.q[0] + (long long)b * Tq * p.q_rs[0] + h * HD, p.q_rs[0], p.d_o[0] + (long long)b * Tq * p.o_rs[0] + h * HD, p.o_rs[0], Tq, nt, w, lane);
  {
    const long long srow = ((long long)b * p.NH + h) * Tq;
    if ((int)threadIdx.x < Tq) { sL[threadIdx.x] = p.lse[srow + threadIdx.x]; sX[threadIdx.x] = p.delta[srow + threadIdx.x]; }
  }

  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);
  const float c2 = p.scale * LOG2E;

  wait_vm0();
  __syncthreads();

  // Two key sub-tiles at a time (four would need 2 x 160 accumulator registers + 96 of K / V rows: the compiler spilled 59 registers to
  // scratch); each pair walks the resident query tiles once more — LDS reads, not HBM.
  constexpr int NK = 2;
  for (int u0 = 0; u0 < nk; u0 += NK) {
    bf16x8 kf[NK][KS], vf[NK][KS];
    f32x4 acc_dk[NK][DF], acc_dv[NK][DF];
#pragma unroll
    for (int uu = 0; uu < NK; ++uu) {
      const bool vk = u0 + uu < nk;
      const long long koff = ((long long)b * Tk + (u0 + uu) * 64 + w * 16 + i) * p.kv_rs[0] + h * HD;
      load_row_frags<HD>(p.k[0] + koff, vk, lane, kf[uu]);
      load_row_frags<HD>(p.v[0] + koff, vk, lane, vf[uu]);
#pragma unroll
      for (int d = 0; d < DF; ++d) { acc_dk[uu][d] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dv[uu][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    for (int t = 0; t < nt; ++t) {
      const unsigned qoff = (unsigned)(t * 2 * TILE), doff = qoff + TILE;
      // lane: key column i, query rows 16 f + 4 g + r of the tile
      const f32x4 l0 = *reinterpret_cast<const f32x4*>(sL + t * 32 + 4 * g) * LOG2E, l1 = *reinterpret_cast<const f32x4*>(sL + t * 32 + 16 + 4 * g) * LOG2E;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(sX + t * 32 + 4 * g), x1 = *reinterpret_cast<const f32x4*>(sX + t * 32 + 16 + 4 * g);
      bf16x8 q0[KS], q1[KS], o0[KS], o1[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        q0[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + qoff + kimm<HD>(kk));
        q1[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + qoff + 16 * PITCH + kimm<HD>(kk));
        o0[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + doff + kimm<HD>(kk));
        o1[kk] = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + doff + 16 * PITCH + kimm<HD>(kk));
      }
      bf16x8 pb[NK], db[NK];
#pragma unroll
      for (int uu = 0; uu < NK; ++uu) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          s0 = mfma16(q0[kk], kf[uu][kk], s0);      // S[q][key]
          s1 = mfma16(q1[kk], kf[uu][kk], s1);
          d0 = mfma16(o0[kk], vf[uu][kk], d0);      // dP[q][key]
          d1 = mfma16(o1[kk], vf[uu][kk], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - l0[r]);
          s1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - l1[r]);
          d0[r] = s0[r] * (d0[r] - x0[r]) * p.scale;
          d1[r] = s1[r] * (d1[r] - x1[r]) * p.scale;
        }
        pb[uu] = pack8(s0, s1); db[uu] = pack8(d0, d1);
      }
      bf16x4 rd[2 * DF], rq[2 * DF];
      res_tr_burst<HD>(va, doff, rd);
      res_tr_burst<HD>(va, qoff, rq);
#pragma unroll
      for (int uu = 0; uu < NK; ++uu) {
#pragma unroll
        for (int d = 0; d < DF; ++d) {
          acc_dv[uu][d] = mfma16(join8(rd[2 * d], rd[2 * d + 1]), pb[uu], acc_dv[uu][d]);
          acc_dk[uu][d] = mfma16(join8(rq[2 * d], rq[2 * d + 1]), db[uu], acc_dk[uu][d]);
        }
      }
    }
#pragma unroll
    for (int uu = 0; uu < NK; ++uu) {
      if (u0 + uu >= nk) continue;
      const long long koff = ((long long)b * Tk + (u0 + uu) * 64 + w * 16 + i) * p.kv_rs[0] + h * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        if (d * 16 + 4 * g >= HD) continue;
        store4(p.dk[0] + koff + d * 16 + 4 * g, acc_dk[uu][d], 1.0f);
        store4(p.dv[0] + koff + d * 16 + 4 * g, acc_dv[uu][d], 1.0f);
      }
    }
  }
}

template <int HD>
int launch_fwd_res(const AttnP& p, hipStream_t s) {
  const int lds = (p.klen[0] >> 5) * 2 * DmaCfg<HD>::TILE;
  auto kern = attn_res_q_kernel<HD, 0>;
  if (int e = set_lds(kern, lds)) return e;
  hipLaunchKernelGGL(kern, dim3(p.B * p.NH), dim3(256), lds, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <int HD>
int launch_bwd_res(const AttnP& p, hipStream_t s) {
  AttnP q = p;
  q.fuse_delta = 1;      // dQ first: it publishes delta = rowsum(dO o O) for the dK / dV launch behind it
  const int lds_q = (p.klen[0] >> 5) * 2 * DmaCfg<HD>::TILE;
  auto kq = attn_res_q_kernel<HD, 1>;
  if (int e = set_lds(kq, lds_q)) return e;
  hipLaunchKernelGGL(kq, dim3(p.B * p.NH), dim3(256), lds_q, s, q);
  LAP_CHECK_LAUNCH();
  const int lds_kv = (p.qlen[0] >> 5) * 2 * DmaCfg<HD>::TILE + 2 * RES_MAX_ROWS * 4;
  auto kkv = attn_res_kv_kernel<HD>;
  if (int e = set_lds(kkv, lds_kv)) return e;
  hipLaunchKernelGGL(kkv, dim3(p.B * p.NH), dim3(256), lds_kv, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
