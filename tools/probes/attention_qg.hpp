// PARKED (round 6, second session): measured slower than the kernel it was to replace — see profiles/r06_attention_two_heads_per_wave_ab.txt.
// To build it again: paste this block into lap_amd/csrc/attention_dma.hpp in front of the "backward: dK, dV" section and route
// launch_fwd_dma<256> to attn_dma_qg_kernel<2> with grid B * NH / 2 * ceil(Tq / 64) when nsplit == 1 and (NH / NKV) % 2 == 0
// (tools/probes/attn_qg_ab.py compares the two through lap_attention_set_variant 3 / 4).
// ======================================================================== forward, G query heads of one kv head per wave (MQA / GQA)
// The forward above as it is, except that a wave carries its 16 query ROWS for G heads that share one kv head (gemma.py:234-235: the
// einsum BTKGH,BSKH groups them the same way): every K fragment read from LDS feeds G score products and every transposed V
// fragment G output products, the mask test is per (row, key) and shared, one barrier per tile serves G heads.  LDS bytes, barriers
// and mask work per flop / G; the price is registers (G x (32 for Q + 64 for O)): one block per CU at G = 2.  Per-row arithmetic is
// the single-head kernel's in the same order: bitwise equal outputs.  nsplit = 1 only (training forward).
template <int G>
__global__ __launch_bounds__(256, 1) void attn_dma_qg_kernel(AttnP p) {
  constexpr int HD = 256;
  using C = DmaCfg<HD>;
  constexpr int KS = C::KS, DF = C::DF, BQ = 64, TILE = C::TILE, PITCH = C::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int ntq = (Tq + BQ - 1) / BQ;
  const int NHG = p.NH / G;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qtile = bid % ntq, hg = (bid / ntq) % NHG, b = bid / (ntq * NHG);
  const int h0 = hg * G;
  const int hk = h0 / (p.NH / p.NKV);
  const int nt0 = (p.klen[0] + 31) >> 5, nt1 = (p.klen[1] + 31) >> 5, ntk = nt0 + nt1;
  int* sWords = reinterpret_cast<int*>(smem + 4 * TILE);
  int* sSum = sWords + ntk * 32;
  const int gt0 = 0, gt1 = ntk;

  const int myq = qtile * BQ + w * 16 + i;
  const bool vq = myq < Tq;
  const int qsg = myq >= p.qlen[0], qloc = myq - (qsg ? p.qlen[0] : 0);
  bf16x8 qf[G][KS];
#pragma unroll
  for (int j = 0; j < G; ++j)
    load_row_frags<HD>(p.q[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.q_rs[qsg] + (h0 + j) * HD, vq, lane, qf[j]);
  const int qi = !vq ? 0 : (p.qinfo ? p.qinfo[(long long)b * Tq + myq] : 0x7fffffff);
  const int qcls = qi >> 24, qidx = qi & 0xffffff;

  auto stage_infos = [&]() {
    for (int base = gt0 + w; base < gt1; base += 32) {
      int v[8], valid[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + 4 * u;
        const int sg = t >= nt0, tl = sg ? t - nt0 : t;
        valid[u] = t < gt1 ? min(32, p.klen[sg] - tl * 32) : 0;
        v[u] = 0;
        if (lane < valid[u]) v[u] = p.kinfo ? p.kinfo[(long long)b * Tk + (sg ? p.klen[0] : 0) + tl * 32 + lane] : 0x7f000000;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = base + 4 * u;
        if (t < gt1) put_infos32(sWords + t * 32, sSum + t * 4, v[u], valid[u], lane, false);
      }
    }
  };

  StreamCursor cur;
  const int rowbytes0 = p.kv_rs[0] * 2, rowbytes1 = p.kv_rs[1] * 2;
  cur.seg = gt0 >= nt0; cur.tile = cur.seg ? gt0 - nt0 : gt0;
  const long long kvoff0 = (long long)b * p.klen[0] * p.kv_rs[0] + hk * HD, kvoff1 = (long long)b * p.klen[1] * p.kv_rs[1] + hk * HD;
  const auto rsK0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
  const auto rsV0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
  const auto rsK1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
  const auto rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
  int dma_row[C::PIECES], dma_col[C::PIECES];
#pragma unroll
  for (int j = 0; j < C::PIECES; ++j) dma_lane<HD>(w, j, lane, dma_row[j], dma_col[j]);
  auto issue = [&](int stage) {
    const int rb = cur.seg ? rowbytes1 : rowbytes0;
    char* base = smem + stage * 2 * TILE;
#pragma unroll
    for (int j = 0; j < C::PIECES; ++j) {
      const unsigned off = dma_col[j] < 0 ? DMA_OOB : (unsigned)((cur.tile * 32 + dma_row[j]) * rb + dma_col[j]);
      char* dst = base + (w * C::PIECES + j) * 1024;
      if (cur.seg == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK0, (LDS_PTR(void))dst, 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV0, (LDS_PTR(void))(dst + TILE), 16, off, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK1, (LDS_PTR(void))dst, 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV1, (LDS_PTR(void))(dst + TILE), 16, off, 0, 0, 0);
      }
    }
    if (++cur.tile == (cur.seg ? nt1 : nt0)) { cur.seg = 1; cur.tile = 0; }
  };

  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);

  const float c2 = p.scale * LOG2E;
  float m[G], l[G];
  f32x4 acc_o[G][DF];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    m[j] = NEG_BIG; l[j] = 0.f;
#pragma unroll
    for (int d = 0; d < DF; ++d) acc_o[j][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  unsigned long long skipmask = 0, fastmask = 0;
  auto step = [&](auto STC, int gt) {
    constexpr int ST = decltype(STC)::value;
    constexpr int KOFF = ST * 2 * TILE, VOFF = KOFF + TILE;
    wait_vm0();
    __syncthreads();
    if (gt + 1 < gt1) issue(ST ^ 1);
    if ((skipmask >> (gt - gt0)) & 1) return;
    const bool fast = (fastmask >> (gt - gt0)) & 1;

    f32x4 s0[G], s1[G];
#pragma unroll
    for (int j = 0; j < G; ++j) { s0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + kimm<HD>(kk));
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + 16 * PITCH + kimm<HD>(kk));
#pragma unroll
      for (int j = 0; j < G; ++j) {
        s0[j] = mfma16(k0, qf[j][kk], s0[j]);
        s1[j] = mfma16(k1, qf[j][kk], s1[j]);
      }
    }
    bool ok0[4], ok1[4];
    if (!fast) {
      const i32x4 kw0 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 4 * g);
      const i32x4 kw1 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok0[r] = (qcls & (kw0[r] >> 24)) != 0 && (kw0[r] & 0xffffff) <= qidx;
        ok1[r] = (qcls & (kw1[r] >> 24)) != 0 && (kw1[r] & 0xffffff) <= qidx;
      }
    }
    bf16x8 pb[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      float mx;
      if (fast) {
        mx = fmaxf(fmaxf(fmaxf(s0[j][0], s0[j][1]), fmaxf(s0[j][2], s0[j][3])), fmaxf(fmaxf(s1[j][0], s1[j][1]), fmaxf(s1[j][2], s1[j][3])));
      } else {
        mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (ok0[r]) mx = fmaxf(mx, s0[j][r]);
          if (ok1[r]) mx = fmaxf(mx, s1[j][r]);
        }
      }
      const float m_new = fmaxf(m[j], max_over_groups(mx) * c2);
      if (__any(m_new > m[j] + 8.0f)) {
        const float alpha = __builtin_amdgcn_exp2f(m[j] - m_new);
        l[j] *= alpha;
#pragma unroll
        for (int d = 0; d < DF; ++d) acc_o[j][d] *= alpha;
        m[j] = m_new;
      }
      if (fast) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0[j][r] = __builtin_amdgcn_exp2f(s0[j][r] * c2 - m[j]);
          s1[j][r] = __builtin_amdgcn_exp2f(s1[j][r] * c2 - m[j]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s0[j][r] = ok0[r] ? __builtin_amdgcn_exp2f(s0[j][r] * c2 - m[j]) : 0.f;
          s1[j][r] = ok1[r] ? __builtin_amdgcn_exp2f(s1[j][r] * c2 - m[j]) : 0.f;
        }
      }
      l[j] += ((s0[j][0] + s0[j][1]) + (s0[j][2] + s0[j][3])) + ((s1[j][0] + s1[j][1]) + (s1[j][2] + s1[j][3]));
      pb[j] = pack8(s0[j], s1[j]);
    }

    // O^T += V^T P^T: every transposed V fragment read once for the G heads
    bf16x4 vr[2][8];
#define LAP_ISSUE_V(GRP, R)                                                               \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                      \
    R[2 * jj] = ds_read_tr_raw<VOFF + ((GRP) >> 1) * 256>(va[((GRP) & 1) * 4 + jj]);      \
    R[2 * jj + 1] = ds_read_tr_raw<VOFF + ((GRP) >> 1) * 256 + 16 * 512>(va[((GRP) & 1) * 4 + jj]); \
  }
#define LAP_PV(GRP, R)                                                                    \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                      \
    const bf16x8 vfrag = join8(R[2 * jj], R[2 * jj + 1]);                                 \
    _Pragma("unroll") for (int j = 0; j < G; ++j)                                         \
      acc_o[j][(GRP) * 4 + jj] = mfma16(vfrag, pb[j], acc_o[j][(GRP) * 4 + jj]);          \
  }
    LAP_ISSUE_V(0, vr[0])
    LAP_ISSUE_V(1, vr[1]) lds_wait8<8>(vr[0]); LAP_PV(0, vr[0])
    LAP_ISSUE_V(2, vr[0]) lds_wait8<8>(vr[1]); LAP_PV(1, vr[1])
    LAP_ISSUE_V(3, vr[1]) lds_wait8<8>(vr[0]); LAP_PV(2, vr[0])
    lds_wait8<0>(vr[1]); LAP_PV(3, vr[1])
#undef LAP_ISSUE_V
#undef LAP_PV
  };

  if (gt0 < gt1) issue(0);
  stage_infos();
  __syncthreads();
  {
    const int t = gt0 + lane;
    i32x4 sm = {0, 0, 0, 0};
    if (t < gt1) sm = *reinterpret_cast<const i32x4*>(sSum + t * 4);
    bool f = sm[3] == 32;
    int qor = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = __builtin_amdgcn_readlane(qcls, r), x = __builtin_amdgcn_readlane(qidx, r);
      qor |= c;
      f = f && (c & sm[0]) != 0 && sm[2] <= x;
    }
    skipmask = __ballot((qor & sm[1]) == 0);
    fastmask = __ballot(f);
  }
  for (int gt = gt0; gt < gt1; gt += 2) {
    step(IC<0>{}, gt);
    if (gt + 1 < gt1) step(IC<1>{}, gt + 1);
  }

#pragma unroll
  for (int j = 0; j < G; ++j) {
    const float lt = sum_over_groups(l[j]);
    if (!vq) continue;
    const int h = h0 + j;
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    const float lse = (m[j] + __builtin_amdgcn_logf(lt)) * LN2;
    bf16* orow = p.o[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.o_rs[qsg] + h * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d)
      if (d * 16 + 4 * g < HD) store4(orow + d * 16 + 4 * g, acc_o[j][d], inv);
    if (p.lse && g == 0) p.lse[((long long)b * p.NH + h) * Tq + myq] = lt > 0.f ? lse : LSE_EMPTY;
  }
}

