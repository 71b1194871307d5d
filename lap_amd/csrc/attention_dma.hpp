// HD = 256 attention kernels (the Gemma-2B / 300M head size of the LAP hot path): the streamed side of each kernel
// arrives in 32-row tiles through LDS-DMA (`buffer_load ... lds`, no VGPR round trip) into a 2-stage ring, so the
// loads of tile t+1 fly while tile t is multiplied and one barrier per tile is all the synchronisation there is.
//
// Included by attention.hip inside its anonymous namespace (shares AttnP, mask_ok, pack8, store4 ...).
//
// LDS image of a tile: 32 rows x 512 B (256 bf16 of d).  The DMA writes 1 KiB = 2 rows per wave instruction,
// lane-linear, so the conflict-avoidance must be a permutation of the SOURCE addresses rather than padding: the
// 16-byte chunk c of row r lives at chunk  c ^ ((r & 7) << 1).
//   * ds_read_b128 fragments (16 rows x 4 chunks): the two half-groups of a 16-lane service group land on even / odd
//     chunks, 8 distinct positions each -> all 64 banks once;
//   * ds_read_b64_tr_b16 fragments (8 rows x 32 B per 32 lanes): the 32-byte block index becomes d ^ (r & 7) -> the 8
//     rows hit 8 distinct blocks of the 256-byte bank window.
// Rows past the end of a segment are fetched with an out-of-range buffer offset, which the hardware turns into zeros.

constexpr int T32_TILE = 32 * 512;         // bytes of one 32 x 256 bf16 tile (head size 256)

// Per head size: k-steps of the contraction over d (KS), 16-wide output fragments along d (DF), LDS row pitch, real
// 16-byte chunks per row, and the source-address swizzle.  Head size 72 (SigLIP) uses a 256-byte pitch: one bank
// window per row, chunk c of row r at  c ^ (((r & 7) << 1) | ((r >> 3) & 1)) - the 16 rows of a ds_read_b128 service
// group hit 16 distinct chunks, the 8 rows of a transposing read 8 distinct 32-byte blocks; chunks 9..15 of a row
// (d >= 72) are fetched with an out-of-range offset, i.e. as zeros.
template <int HD> struct DmaCfg;
template <> struct DmaCfg<256> {
  static constexpr int KS = 8, DF = 16, PITCH = 512, REAL_CHUNKS = 32, TILE = 32 * 512, PIECES = 4, KREGS = 4, VREGS = 8, BLOCKS = 2, KV_BLOCKS = 1;
  __device__ static __forceinline__ int swz(int row) { return (row & 7) << 1; }
};
template <> struct DmaCfg<72> {
  static constexpr int KS = 3, DF = 5, PITCH = 256, REAL_CHUNKS = 9, TILE = 32 * 256, PIECES = 2, KREGS = 3, VREGS = 5, BLOCKS = 4, KV_BLOCKS = 3;
  __device__ static __forceinline__ int swz(int row) { return ((row & 7) << 1) | ((row >> 3) & 1); }
};
// DMA piece j of wave w (1 KiB = 1024 / PITCH rows): tile row and byte column of the lane's 16-byte chunk, or col < 0
// for a padding chunk that must read as zeros.
template <int HD>
__device__ __forceinline__ void dma_lane(int w, int j, int lane, int& row, int& col) {
  using C = DmaCfg<HD>;
  constexpr int CPR = C::PITCH / 16;          // chunks per row
  const int piece = w * C::PIECES + j;
  row = piece * (1024 / C::PITCH) + lane / CPR;
  const int c = (lane % CPR) ^ C::swz(row);
  col = c < C::REAL_CHUNKS ? c * 16 : -1;
}
// Loop invariant per-lane LDS offsets.  K-contiguous fragment (ds_read_b128) of tile row nf*16 + i, k-step kk:
// kbase[kreg(kk)] + kimm(kk) + nf * 16 * PITCH;  transposing read of d-fragment d: vbase[vreg(d)] + vimm(d) (+ 16 rows).
template <int HD>
__device__ __forceinline__ void dma_frag_bases(const char* smem, int lane, const char* (&kbase)[DmaCfg<HD>::KREGS], unsigned (&vbase)[DmaCfg<HD>::VREGS]) {
  using C = DmaCfg<HD>;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int c = 0; c < C::KREGS; ++c) kbase[c] = smem + i * C::PITCH + (((4 * c + g) ^ C::swz(i)) << 4);
  const int tr_row = 4 * g + (i >> 2);
  const unsigned lane_swz = (unsigned)(C::swz(tr_row) ^ ((i & 3) >> 1));
  const unsigned lane_off = lds_addr_of(smem) + (unsigned)(tr_row * C::PITCH + ((i & 1) << 3));
#pragma unroll
  for (int c = 0; c < C::VREGS; ++c) vbase[c] = lane_off + (((unsigned)(2 * c) ^ lane_swz) << 4);
}
template <int HD> __device__ __forceinline__ constexpr int kreg(int kk) { return HD == 256 ? (kk & 3) : kk; }
template <int HD> __device__ __forceinline__ constexpr int kimm(int kk) { return HD == 256 ? (kk >> 2) * 256 : 0; }
constexpr int T32_INFO_INTS = 48;          // 32 info words + and / or / index summaries (padded)
constexpr unsigned DMA_OOB = 0x80000000u;

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Inclusive scan step inside a row of 16 lanes (DPP row_shr, no LDS round trip); lanes shifted in get `ident`.
template <int SHR>
__device__ __forceinline__ int dpp_shr(int v, int ident) {
  return __builtin_amdgcn_update_dpp(ident, v, 0x110 + SHR, 0xf, 0xf, false);
}
// One wave publishes a 32-row tile of info words (lane < 32 holds row `lane`; rows past the end of the segment and
// lanes >= 32 hold 0 = class 0 = matches nothing) and its summary {AND of class bits, OR of class bits, max (keys) /
// min (queries) of the index field, number of valid rows}.
__device__ __forceinline__ void put_infos32(int* words, int* sum, int v, int valid, int lane, bool want_min) {
  const bool in = lane < valid;
  int c_and = in ? (v >> 24) : (lane < 32 ? 0 : -1);
  int c_or = v >> 24;
  const int id_idx = want_min ? 0xffffff : 0;
  int idx = in ? (v & 0xffffff) : (lane < 32 ? 0xffffff - id_idx : id_idx);   // an invalid row defeats "all ok"
#define LAP_SCAN_STEP(SHR)                                                              \
  c_and &= dpp_shr<SHR>(c_and, -1);                                                     \
  c_or |= dpp_shr<SHR>(c_or, 0);                                                        \
  { const int o = dpp_shr<SHR>(idx, id_idx); idx = want_min ? min(idx, o) : max(idx, o); }
  LAP_SCAN_STEP(1) LAP_SCAN_STEP(2) LAP_SCAN_STEP(4) LAP_SCAN_STEP(8)
#undef LAP_SCAN_STEP
  // lanes 15 and 31 hold the totals of rows 0-15 / 16-31
  const int a = __builtin_amdgcn_readlane(c_and, 15) & __builtin_amdgcn_readlane(c_and, 31);
  const int o = __builtin_amdgcn_readlane(c_or, 15) | __builtin_amdgcn_readlane(c_or, 31);
  const int x0 = __builtin_amdgcn_readlane(idx, 15), x1 = __builtin_amdgcn_readlane(idx, 31);
  const int x = want_min ? min(x0, x1) : max(x0, x1);
  if (lane < 32) words[lane] = v;
  if (lane == 0) *reinterpret_cast<i32x4*>(sum) = i32x4{a, o, x, valid};
}

// max over the four lanes {i, i+16, i+32, i+48} (gfx950 v_permlane{32,16}_swap: pure VALU, no LDS round trip)
__device__ __forceinline__ float max_over_groups(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float sum_over_groups(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

template <int V> struct IC { static constexpr int value = V; };
// K-contiguous fragment through a raw ds_read_b128 (see ds_read_tr_raw: valid only behind lds_wait4x / lds_wait_all + lds_tie)
template <int OFFSET>
__device__ __forceinline__ bf16x8 ds_read_b128_raw(unsigned lds_addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFFSET));
  return r;
}
template <int LEFT>
__device__ __forceinline__ void lds_wait4x(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(LEFT));
}
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// Streamed-side cursor: the 32-row tiles of the two segments form one list.  `rs*` are buffer descriptors whose
// num_records end right after the last valid row, so rows past the end of a segment read as zeros with no per-lane
// range test.  K and V share the row stride, hence the per-lane offsets.
struct StreamCursor {
  int seg, tile;        // next tile to fetch
};
__device__ __forceinline__ int seg_records(int len, int row_stride, int hd) { return len > 0 ? ((len - 1) * row_stride + hd) * 2 : 0; }

// All DF transposing fragment reads of one tile (LDS byte offset OFF) in one burst + fence; for small head sizes.
template <int HD, int OFF>
__device__ __forceinline__ void tr_burst(const unsigned (&va)[DmaCfg<HD>::VREGS], bf16x4 (&r)[2 * DmaCfg<HD>::DF]) {
  static_assert(DmaCfg<HD>::VREGS == DmaCfg<HD>::DF, "one address register per d-fragment");
#pragma unroll
  for (int d = 0; d < DmaCfg<HD>::DF; ++d) {
    r[2 * d] = ds_read_tr_raw<OFF>(va[d]);
    r[2 * d + 1] = ds_read_tr_raw<OFF + 16 * DmaCfg<HD>::PITCH>(va[d]);
  }
  lds_wait_all();
#pragma unroll
  for (int d = 0; d < 2 * DmaCfg<HD>::DF; ++d) lds_tie(r[d]);
}

// =============================================================================== forward
// Block = 4 waves x 16 queries of the JOINT query sequence (the two segments are tiled as one list, a lane resolves
// the segment of its own row); two blocks per CU.  The loop body is written for a low instruction count -- the
// kernel is issue bound (each wave issues at most one instruction every 4 cycles; 32 MFMAs per 32-key tile leave room
// for ~250 others): LDS addresses are loop invariant registers + immediates (the two stages are unrolled), the
// info words / tile summaries are staged once per block, the running-max exchange uses v_permlane swaps, the
// accumulator rescale is skipped while no lane's maximum moves, and exp runs in the log2 domain (one FMA + v_exp).
//
// MODE 1 = backward dQ with the same skeleton: per key tile S^T = K Q^T and dP^T = V dO^T (V read K-contiguous like
// K), dS^T = P^T o (dP^T - delta) * scale with P recomputed from the saved log-sum-exp, and dQ^T += K^T dS^T through
// the transposing reads of the K tile.
template <int HD, int MODE>
__global__ __launch_bounds__(256, DmaCfg<HD>::BLOCKS) void attn_dma_q_kernel(AttnP p) {
  using C = DmaCfg<HD>;
  constexpr int KS = C::KS, DF = C::DF, BQ = 64, TILE = C::TILE, PITCH = C::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x [K tile | V tile], info words, summaries
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int ntq = (Tq + BQ - 1) / BQ;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qtile = bid % ntq, h = (bid / ntq) % p.NH, b = bid / (ntq * p.NH);
  const int hk = h / (p.NH / p.NKV);
  const int nt0 = (p.klen[0] + 31) >> 5, nt1 = (p.klen[1] + 31) >> 5, ntk = nt0 + nt1;
  int* sWords = reinterpret_cast<int*>(smem + 4 * TILE);
  int* sSum = sWords + ntk * 32;
  const int per = (ntk + p.nsplit - 1) / p.nsplit;
  const int gt0 = blockIdx.y * per, gt1 = min(ntk, gt0 + per);

  // ---- my query row
  const int myq = qtile * BQ + w * 16 + i;   // joint row
  const bool vq = myq < Tq;
  const int qsg = myq >= p.qlen[0], qloc = myq - (qsg ? p.qlen[0] : 0);
  bf16x8 qf[KS];
  load_row_frags<HD>(p.q[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.q_rs[qsg] + h * HD, vq, lane, qf);
  const int qi = !vq ? 0 : (p.qinfo ? p.qinfo[(long long)b * Tq + myq] : 0x7fffffff);   // invalid row: class 0
  const int qcls = qi >> 24, qidx = qi & 0xffffff;
  bf16x8 dof[KS];   // MODE 1 only (dead otherwise)
  float lse2 = 0.f, dl_q = 0.f;     // MODE 1: log2-domain log-sum-exp and delta = rowsum(dO o O) of my row
  if constexpr (MODE == 1) {
    load_row_frags<HD>(p.d_o[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.o_rs[qsg] + h * HD, vq, lane, dof);
    lse2 = (vq ? p.lse[((long long)b * p.NH + h) * Tq + myq] : LSE_EMPTY) * LOG2E;
    if (p.fuse_delta) {
      // delta = rowsum(dO o O) of my row, computed here (the dO fragments are in registers anyway) and published for the dK / dV
      // launch BEHIND this one: no separate pass over O and dO.  Lane (i, g) holds 8 of every 32 d; the four g sum up.
      bf16x8 of[KS];
      load_row_frags<HD>(p.o[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.o_rs[qsg] + h * HD, vq, lane, of);
      float a = 0.f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)dof[kk][e] * (float)of[kk][e];
      dl_q = sum_over_groups(a);
      if (vq && g == 0) p.delta[((long long)b * p.NH + h) * Tq + myq] = dl_q;
    } else {
      dl_q = vq ? p.delta[((long long)b * p.NH + h) * Tq + myq] : 0.f;
    }
  }

  // ---- stage the info words + tile summaries of my share of the key tiles (once per block); all loads of a
  // batch of 8 tiles per wave are in flight together
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

  // ---- streamed side
  StreamCursor cur;
  const int rowbytes0 = p.kv_rs[0] * 2, rowbytes1 = p.kv_rs[1] * 2;
  cur.seg = gt0 >= nt0; cur.tile = cur.seg ? gt0 - nt0 : gt0;
  const long long kvoff0 = (long long)b * p.klen[0] * p.kv_rs[0] + hk * HD, kvoff1 = (long long)b * p.klen[1] * p.kv_rs[1] + hk * HD;
  const auto rsK0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
  const auto rsV0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[0] + kvoff0), 0, seg_records(p.klen[0], p.kv_rs[0], HD), 0x00020000);
  const auto rsK1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.k[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
  const auto rsV1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.v[1] + kvoff1), 0, seg_records(p.klen[1], p.kv_rs[1], HD), 0x00020000);
  // DMA piece j of this wave: tile row 8w + 2j + lane/32, physical chunk lane%32 holds logical chunk ^ swz(row)
  int dma_row[C::PIECES], dma_col[C::PIECES];
#pragma unroll
  for (int j = 0; j < C::PIECES; ++j) dma_lane<HD>(w, j, lane, dma_row[j], dma_col[j]);
  auto issue = [&](int stage) {   // fetch tile `cur` into `stage`, advance the cursor
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

  // ---- loop invariant LDS addresses (dma_frag_bases)
  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);

  const float c2 = p.scale * LOG2E;     // logits in the log2 domain: s * c2
  float m = NEG_BIG, l = 0.f;
  f32x4 acc_o[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned long long skipmask = 0, fastmask = 0;   // per-wave tile decisions, filled in below
  auto step = [&](auto STC, int gt) {
    constexpr int ST = decltype(STC)::value;
    constexpr int KOFF = ST * 2 * TILE, VOFF = KOFF + TILE;
    wait_vm0();          // my pieces of tile gt have landed ...
    __syncthreads();     // ... and everybody's; every wave is done with the stage refilled below
    if (gt + 1 < gt1) issue(ST ^ 1);

    // tile-level mask decision (wave uniform, precomputed): skip / no masking needed / per-element
    if ((skipmask >> (gt - gt0)) & 1) return;       // nothing in this tile is visible to this wave's queries
    const bool fast = (fastmask >> (gt - gt0)) & 1;

    if constexpr (MODE == 1) {
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + kimm<HD>(kk));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + 16 * PITCH + kimm<HD>(kk));
        const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + VOFF + kimm<HD>(kk));
        const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + VOFF + 16 * PITCH + kimm<HD>(kk));
        s0 = mfma16(k0, qf[kk], s0);      // S^T[key][q]
        s1 = mfma16(k1, qf[kk], s1);
        d0 = mfma16(v0, dof[kk], d0);     // dP^T[key][q]
        d1 = mfma16(v1, dof[kk], d1);
      }
      if (fast) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - lse2) * (d0[r] - dl_q) * p.scale;
          d1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - lse2) * (d1[r] - dl_q) * p.scale;
        }
      } else {
        const i32x4 kw0 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 4 * g);
        const i32x4 kw1 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool a0 = (qcls & (kw0[r] >> 24)) != 0 && (kw0[r] & 0xffffff) <= qidx;
          const bool a1 = (qcls & (kw1[r] >> 24)) != 0 && (kw1[r] & 0xffffff) <= qidx;
          d0[r] = a0 ? __builtin_amdgcn_exp2f(s0[r] * c2 - lse2) * (d0[r] - dl_q) * p.scale : 0.f;
          d1[r] = a1 ? __builtin_amdgcn_exp2f(s1[r] * c2 - lse2) * (d1[r] - dl_q) * p.scale : 0.f;
        }
      }
      const bf16x8 pb = pack8(d0, d1);
      if constexpr (HD == 256) {
      // dQ^T += K^T dS^T: K fragments through raw transposing reads, software pipelined in groups of 4 d-fragments
      bf16x4 vr[2][8];
#define LAP_ISSUE_K(GRP, R)                                                               \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
    R[2 * j] = ds_read_tr_raw<KOFF + ((GRP) >> 1) * 256>(va[((GRP) & 1) * 4 + j]);        \
    R[2 * j + 1] = ds_read_tr_raw<KOFF + ((GRP) >> 1) * 256 + 16 * 512>(va[((GRP) & 1) * 4 + j]); \
  }
#define LAP_DQ(GRP, R)                                                                    \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
    acc_o[(GRP) * 4 + j] = mfma16(join8(R[2 * j], R[2 * j + 1]), pb, acc_o[(GRP) * 4 + j]);
      LAP_ISSUE_K(0, vr[0])
      LAP_ISSUE_K(1, vr[1]) lds_wait8<8>(vr[0]); LAP_DQ(0, vr[0])
      LAP_ISSUE_K(2, vr[0]) lds_wait8<8>(vr[1]); LAP_DQ(1, vr[1])
      LAP_ISSUE_K(3, vr[1]) lds_wait8<8>(vr[0]); LAP_DQ(2, vr[0])
      lds_wait8<0>(vr[1]); LAP_DQ(3, vr[1])
#undef LAP_ISSUE_K
#undef LAP_DQ
      } else {
        // few d-fragments (head size 72: 5): all transposing reads in one burst, one fence, then the MFMAs
        bf16x4 vr[2 * DF];
        tr_burst<HD, KOFF>(va, vr);
#pragma unroll
        for (int d = 0; d < DF; ++d) acc_o[d] = mfma16(join8(vr[2 * d], vr[2 * d + 1]), pb, acc_o[d]);
      }
      return;
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + kimm<HD>(kk));
      const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + KOFF + 16 * PITCH + kimm<HD>(kk));
      s0 = mfma16(k0, qf[kk], s0);
      s1 = mfma16(k1, qf[kk], s1);
    }
    // online softmax in the log2 domain; lane owns query column i and keys 16 nf + 4 g + r.
    // The running maximum is LAZY: it only moves (and the accumulators are only rescaled) when some row's new
    // maximum exceeds the one in use by more than 2^8; until then p = exp2(s - m) is allowed to reach 256 -- the
    // bf16 rounding of P is scale invariant and l / O carry the same factor, so the result is unchanged.
    i32x4 kw0 = {0, 0, 0, 0}, kw1 = {0, 0, 0, 0};
    bool ok0[4], ok1[4];
    float mx;
    if (fast) {
      mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
    } else {
      kw0 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 4 * g);
      kw1 = *reinterpret_cast<const i32x4*>(sWords + gt * 32 + 16 + 4 * g);
      mx = NEG_BIG;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok0[r] = (qcls & (kw0[r] >> 24)) != 0 && (kw0[r] & 0xffffff) <= qidx;
        ok1[r] = (qcls & (kw1[r] >> 24)) != 0 && (kw1[r] & 0xffffff) <= qidx;
        if (ok0[r]) mx = fmaxf(mx, s0[r]);
        if (ok1[r]) mx = fmaxf(mx, s1[r]);
      }
    }
    const float m_new = fmaxf(m, max_over_groups(mx) * c2);
    if (__any(m_new > m + 8.0f)) {
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int d = 0; d < DF; ++d) acc_o[d] *= alpha;
      m = m_new;
    }
    if (fast) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - m);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - m);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = ok0[r] ? __builtin_amdgcn_exp2f(s0[r] * c2 - m) : 0.f;
        s1[r] = ok1[r] ? __builtin_amdgcn_exp2f(s1[r] * c2 - m) : 0.f;
      }
    }
    l += ((s0[0] + s0[1]) + (s0[2] + s0[3])) + ((s1[0] + s1[1]) + (s1[2] + s1[3]));
    const bf16x8 pb = pack8(s0, s1);

    if constexpr (HD == 256) {
    // O^T += V^T P^T: V fragments through raw transposing reads, software pipelined in groups of 4 d-fragments
    // (measured, round 4: all 16 K fragments of the S product in flight ahead of their MFMAs and the first V groups issued before
    // the softmax change nothing here, 166.9 -> 164.3 us; neither does dropping the DMA wait, 152 us, or the barrier, 138 us —
    // two waves per SIMD already cover each other's LDS latency; docs/EXPERIMENTS.md H)
    bf16x4 vr[2][8];
#define LAP_ISSUE_V(GRP, R)                                                               \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                         \
    R[2 * j] = ds_read_tr_raw<VOFF + ((GRP) >> 1) * 256>(va[((GRP) & 1) * 4 + j]);        \
    R[2 * j + 1] = ds_read_tr_raw<VOFF + ((GRP) >> 1) * 256 + 16 * 512>(va[((GRP) & 1) * 4 + j]); \
  }
#define LAP_PV(GRP, R)                                                                    \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
    acc_o[(GRP) * 4 + j] = mfma16(join8(R[2 * j], R[2 * j + 1]), pb, acc_o[(GRP) * 4 + j]);
    LAP_ISSUE_V(0, vr[0])
    LAP_ISSUE_V(1, vr[1]) lds_wait8<8>(vr[0]); LAP_PV(0, vr[0])
    LAP_ISSUE_V(2, vr[0]) lds_wait8<8>(vr[1]); LAP_PV(1, vr[1])
    LAP_ISSUE_V(3, vr[1]) lds_wait8<8>(vr[0]); LAP_PV(2, vr[0])
    lds_wait8<0>(vr[1]); LAP_PV(3, vr[1])
#undef LAP_ISSUE_V
#undef LAP_PV
    } else {
      bf16x4 vr[2 * DF];
      tr_burst<HD, VOFF>(va, vr);
#pragma unroll
      for (int d = 0; d < DF; ++d) acc_o[d] = mfma16(join8(vr[2 * d], vr[2 * d + 1]), pb, acc_o[d]);
    }
  };

  if (gt0 < gt1) issue(0);
  stage_infos();      // overlaps the first tile's DMA
  __syncthreads();
  // Tile-level mask decisions of this wave for its whole share of key tiles, as two scalar bit masks (lane t decides
  // tile gt0 + t): skip = no query of the wave sees any key of the tile; fast = every query sees every key.
  {
    const int t = gt0 + lane;
    i32x4 sm = {0, 0, 0, 0};
    if (t < gt1) sm = *reinterpret_cast<const i32x4*>(sSum + t * 4);
    bool f = sm[3] == 32;
    int qor = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // lanes 0..15 hold the wave's 16 query rows
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

  if (MODE == 1) {
    if (!vq) return;
    bf16* dqrow = p.dq[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.q_rs[qsg] + h * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d)
      if (d * 16 + 4 * g < HD) store4(dqrow + d * 16 + 4 * g, acc_o[d], 1.0f);
    return;
  }
  l = sum_over_groups(l);
  if (!vq) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  const float lse = (m + __builtin_amdgcn_logf(l)) * LN2;   // natural log-sum-exp of scale * q.k
  if (p.nsplit > 1) {
    // partial result of this key share: normalised O (f32) + its log-sum-exp; combined by attn_fwd_combine_kernel
    const long long row = ((long long)blockIdx.y * p.B + b) * Tq + myq;
    float* op = p.part + (row * p.NH + h) * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d)
      if (d * 16 + 4 * g < HD) *reinterpret_cast<f32x4*>(op + d * 16 + 4 * g) = acc_o[d] * inv;
    if (g == 0) p.lpart[(((long long)blockIdx.y * p.B + b) * p.NH + h) * Tq + myq] = l > 0.f ? lse : NEG_BIG;
    return;
  }
  bf16* orow = p.o[qsg] + (b * (long long)p.qlen[qsg] + qloc) * p.o_rs[qsg] + h * HD;
#pragma unroll
  for (int d = 0; d < DF; ++d)
    if (d * 16 + 4 * g < HD) store4(orow + d * 16 + 4 * g, acc_o[d], inv);
  if (p.lse && g == 0) p.lse[((long long)b * p.NH + h) * Tq + myq] = l > 0.f ? lse : LSE_EMPTY;
}

// ======================================================================== backward: dK, dV
// Block = (b, kv head, head group, 64-key tile of one key segment); wave owns 16 keys (lane column i) whose K and V
// rows live in registers.  The streamed side is the list of (query head of the group) x (32-row query tile): the Q
// and dO tiles arrive through the 2-stage LDS-DMA ring, the 32 log-sum-exp / delta values of the tile through two
// dword DMAs of wave 0.  Per tile:  S = Q K^T, dP = dO V^T (Q / dO read K-contiguous),  P = exp2(S c2 - lse),
// dS = P o (dP - delta) scale,  dV^T += dO^T P,  dK^T += Q^T dS (Q / dO read through raw transposing reads).
template <int HD>
__global__ __launch_bounds__(256, DmaCfg<HD>::KV_BLOCKS) void attn_dma_kv_kernel(AttnP p) {
  using C = DmaCfg<HD>;
  constexpr int KS = C::KS, DF = C::DF, TILE = C::TILE, PITCH = C::PITCH;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x [Q tile | dO tile], 2 x [lse | delta], q infos
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const BlockId id = decode_block(p.klen[0], p.klen[1], p.NKV * p.hsplit);
  const int b = id.b, hk = id.h / p.hsplit, hg = id.h % p.hsplit, kseg = id.seg;
  const int klen = p.klen[kseg];
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int mykey = id.tile * 64 + w * 16 + i;
  const bool vk = mykey < klen;
  const long long koff = (b * (long long)klen + mykey) * p.kv_rs[kseg] + hk * HD;
  const int nq0 = (p.qlen[0] + 31) >> 5, nq1 = (p.stop && kseg == 0) ? 0 : (p.qlen[1] + 31) >> 5, ntq = nq0 + nq1;
  float* sLD = reinterpret_cast<float*>(smem + 4 * TILE);          // per stage: 64 floats lse(+pad), 64 floats delta(+pad)
  int* sWords = reinterpret_cast<int*>(smem + 4 * TILE + 1024);
  int* sSum = sWords + ntq * 32;

  bf16x8 kf[KS], vf[KS];
  load_row_frags<HD>(p.k[kseg] + koff, vk, lane, kf);
  load_row_frags<HD>(p.v[kseg] + koff, vk, lane, vf);
  const int ki = !vk ? 0 : (p.kinfo ? p.kinfo[(long long)b * Tk + (kseg ? p.klen[0] : 0) + mykey] : 0x7f000000);
  const int kcls = ki >> 24, kidx = ki & 0xffffff;

  // ---- query info words + tile summaries (once per block; the same for every head)
  for (int base = w; base < ntq; base += 32) {
    int v[8], valid[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = base + 4 * u;
      const int sg = t >= nq0, tl = sg ? t - nq0 : t;
      valid[u] = t < ntq ? min(32, p.qlen[sg] - tl * 32) : 0;
      v[u] = 0;
      if (lane < valid[u]) v[u] = p.qinfo ? p.qinfo[(long long)b * Tq + (sg ? p.qlen[0] : 0) + tl * 32 + lane] : 0x7fffffff;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = base + 4 * u;
      if (t < ntq) put_infos32(sWords + t * 32, sSum + t * 4, v[u], valid[u], lane, true);
    }
  }

  // ---- streamed side: items = (head of my group) x (query tile)
  const int hpk = p.NH / p.NKV, hpg = hpk / p.hsplit;
  const int h_first = hk * hpk + hg * hpg;
  const int items = hpg * ntq;
  int cur_h = h_first, cur_t = 0;   // next item to fetch
  int dma_row[C::PIECES], dma_col[C::PIECES];
#pragma unroll
  for (int j = 0; j < C::PIECES; ++j) dma_lane<HD>(w, j, lane, dma_row[j], dma_col[j]);
  auto issue = [&](int stage) {
    const int sg = cur_t >= nq0, tl = sg ? cur_t - nq0 : cur_t;
    const int qlen = p.qlen[sg], qrs = p.q_rs[sg], ors = p.o_rs[sg];
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)(p.q[sg] + (long long)b * qlen * qrs + cur_h * HD), 0, seg_records(qlen, qrs, HD), 0x00020000);
    const auto rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(p.d_o[sg] + (long long)b * qlen * ors + cur_h * HD), 0, seg_records(qlen, ors, HD), 0x00020000);
    char* base = smem + stage * 2 * TILE;
#pragma unroll
    for (int j = 0; j < C::PIECES; ++j) {
      char* dst = base + (w * C::PIECES + j) * 1024;
      const unsigned oq = dma_col[j] < 0 ? DMA_OOB : (unsigned)((tl * 32 + dma_row[j]) * qrs * 2 + dma_col[j]);
      const unsigned od = dma_col[j] < 0 ? DMA_OOB : (unsigned)((tl * 32 + dma_row[j]) * ors * 2 + dma_col[j]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (LDS_PTR(void))dst, 16, oq, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (LDS_PTR(void))(dst + TILE), 16, od, 0, 0, 0);
    }
    if (w == 0) {   // lse / delta of the tile's rows: lanes 0..31 fetch, lanes 32..63 (and rows past the end) write zeros
      const long long roff = ((long long)b * p.NH + cur_h) * Tq + (sg ? p.qlen[0] : 0);
      const auto rsL = __builtin_amdgcn_make_buffer_rsrc((void*)(p.lse + roff), 0, qlen * 4, 0x00020000);
      const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.delta + roff), 0, qlen * 4, 0x00020000);
      const unsigned off = lane < 32 ? (unsigned)((tl * 32 + lane) * 4) : DMA_OOB;
      float* dst = sLD + stage * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsL, (LDS_PTR(void))dst, 4, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (LDS_PTR(void))(dst + 64), 4, off, 0, 0, 0);
    }
    if (++cur_t == ntq) { cur_t = 0; ++cur_h; }
  };

  // ---- loop invariant LDS addresses (same tile image as the forward kernel)
  const char* kp[C::KREGS];
  unsigned va[C::VREGS];
  dma_frag_bases<HD>(smem, lane, kp, va);
  unsigned ka[C::KREGS];
#pragma unroll
  for (int c = 0; c < C::KREGS; ++c) ka[c] = lds_addr_of(kp[c]);

  const float c2 = p.scale * LOG2E;
  f32x4 acc_dk[DF], acc_dv[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) { acc_dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  unsigned long long skipmask = 0, fastmask = 0;   // per query tile, the same for every head
  auto step = [&](auto STC, int qt) {
    constexpr int ST = decltype(STC)::value;
    constexpr int QOFF = ST * 2 * TILE, DOFF = QOFF + TILE;
    wait_vm0();
    __syncthreads();
    if (cur_h < h_first + hpg) issue(ST ^ 1);
    if ((skipmask >> qt) & 1) return;      // no query of this tile sees this wave's keys
    const bool fast = (fastmask >> qt) & 1;

    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    // lane: key column i, query rows 16 f + 4 g + r
    const float* ld = sLD + ST * 128;
    f32x4 l0 = *reinterpret_cast<const f32x4*>(ld + 4 * g) * LOG2E, l1 = *reinterpret_cast<const f32x4*>(ld + 16 + 4 * g) * LOG2E;
    f32x4 x0 = *reinterpret_cast<const f32x4*>(ld + 64 + 4 * g), x1 = *reinterpret_cast<const f32x4*>(ld + 80 + 4 * g);
    if constexpr (HD == 256) { lds_tie(l0); lds_tie(l1); lds_tie(x0); lds_tie(x1); }     // the compiler's own lgkmcnt(0) for these goes HERE, not behind the raw reads below
    if constexpr (HD == 256) {
      // One block per CU = one wave per SIMD: K, V (64 registers), the dK / dV accumulators (128) and the fragment buffers do not
      // fit 256 registers (two blocks per CU spilled 66 dwords to scratch, reloaded per tile: 458 us; one block without
      // spills: 352 us).  With nobody else on the SIMD to hide the LDS latency, the K-contiguous fragment reads run in a sliding
      // window of 12 ahead of their MFMAs (lgkmcnt is a 4-bit counter) instead of read, wait, MFMA, read, ...: 300 us.  Same
      // operands, same order of accumulation: bitwise equal.
      bf16x8 fr[32];
#define LAP_RQ(KK) \
      fr[4 * KK] = ds_read_b128_raw<QOFF + kimm<HD>(KK)>(ka[kreg<HD>(KK)]); \
      fr[4 * KK + 1] = ds_read_b128_raw<QOFF + 16 * PITCH + kimm<HD>(KK)>(ka[kreg<HD>(KK)]); \
      fr[4 * KK + 2] = ds_read_b128_raw<DOFF + kimm<HD>(KK)>(ka[kreg<HD>(KK)]); \
      fr[4 * KK + 3] = ds_read_b128_raw<DOFF + 16 * PITCH + kimm<HD>(KK)>(ka[kreg<HD>(KK)]);
#define LAP_SQ(KK, LEFT) \
      lds_wait4x<LEFT>(fr[4 * KK], fr[4 * KK + 1], fr[4 * KK + 2], fr[4 * KK + 3]); \
      s0 = mfma16(fr[4 * KK], kf[KK], s0); s1 = mfma16(fr[4 * KK + 1], kf[KK], s1); \
      d0 = mfma16(fr[4 * KK + 2], vf[KK], d0); d1 = mfma16(fr[4 * KK + 3], vf[KK], d1);
      LAP_RQ(0) LAP_RQ(1) LAP_RQ(2)
      LAP_SQ(0, 8) LAP_RQ(3)
      LAP_SQ(1, 8) LAP_RQ(4)
      LAP_SQ(2, 8) LAP_RQ(5)
      LAP_SQ(3, 8) LAP_RQ(6)
      LAP_SQ(4, 8) LAP_RQ(7)
      LAP_SQ(5, 8) LAP_SQ(6, 4) LAP_SQ(7, 0)
#undef LAP_RQ
#undef LAP_SQ
    } else {
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const bf16x8 q0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + QOFF + kimm<HD>(kk));
      const bf16x8 q1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + QOFF + 16 * PITCH + kimm<HD>(kk));
      const bf16x8 o0 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + DOFF + kimm<HD>(kk));
      const bf16x8 o1 = *reinterpret_cast<const bf16x8*>(kp[kreg<HD>(kk)] + DOFF + 16 * PITCH + kimm<HD>(kk));
      s0 = mfma16(q0, kf[kk], s0);      // S[q][key]
      s1 = mfma16(q1, kf[kk], s1);
      d0 = mfma16(o0, vf[kk], d0);      // dP[q][key]
      d1 = mfma16(o1, vf[kk], d1);
    }
    }
    if (fast) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(s0[r] * c2 - l0[r]);
        s1[r] = __builtin_amdgcn_exp2f(s1[r] * c2 - l1[r]);
        d0[r] = s0[r] * (d0[r] - x0[r]) * p.scale;
        d1[r] = s1[r] * (d1[r] - x1[r]) * p.scale;
      }
    } else {
      const i32x4 qw0 = *reinterpret_cast<const i32x4*>(sWords + qt * 32 + 4 * g);
      const i32x4 qw1 = *reinterpret_cast<const i32x4*>(sWords + qt * 32 + 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool a0 = ((qw0[r] >> 24) & kcls) != 0 && kidx <= (qw0[r] & 0xffffff);
        const bool a1 = ((qw1[r] >> 24) & kcls) != 0 && kidx <= (qw1[r] & 0xffffff);
        s0[r] = a0 ? __builtin_amdgcn_exp2f(s0[r] * c2 - l0[r]) : 0.f;
        s1[r] = a1 ? __builtin_amdgcn_exp2f(s1[r] * c2 - l1[r]) : 0.f;
        d0[r] = s0[r] * (d0[r] - x0[r]) * p.scale;
        d1[r] = s1[r] * (d1[r] - x1[r]) * p.scale;
      }
    }
    const bf16x8 pb = pack8(s0, s1), db = pack8(d0, d1);
    if constexpr (HD == 256) {
    // dV^T += dO^T P, dK^T += Q^T dS: raw transposing reads, double buffered in groups of 2 d-fragments x 2 tiles
    bf16x4 tr[2][8];
#define LAP_ISSUE_T(GRP, R)                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                         \
    R[4 * j] = ds_read_tr_raw<DOFF + ((GRP) >> 2) * 256>(va[((GRP) & 3) * 2 + j]);        \
    R[4 * j + 1] = ds_read_tr_raw<DOFF + ((GRP) >> 2) * 256 + 16 * 512>(va[((GRP) & 3) * 2 + j]); \
    R[4 * j + 2] = ds_read_tr_raw<QOFF + ((GRP) >> 2) * 256>(va[((GRP) & 3) * 2 + j]);    \
    R[4 * j + 3] = ds_read_tr_raw<QOFF + ((GRP) >> 2) * 256 + 16 * 512>(va[((GRP) & 3) * 2 + j]); \
  }
#define LAP_DKDV(GRP, R)                                                                  \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                         \
    acc_dv[(GRP) * 2 + j] = mfma16(join8(R[4 * j], R[4 * j + 1]), pb, acc_dv[(GRP) * 2 + j]);      \
    acc_dk[(GRP) * 2 + j] = mfma16(join8(R[4 * j + 2], R[4 * j + 3]), db, acc_dk[(GRP) * 2 + j]);  \
  }
    LAP_ISSUE_T(0, tr[0])
    LAP_ISSUE_T(1, tr[1]) lds_wait8<8>(tr[0]); LAP_DKDV(0, tr[0])
    LAP_ISSUE_T(2, tr[0]) lds_wait8<8>(tr[1]); LAP_DKDV(1, tr[1])
    LAP_ISSUE_T(3, tr[1]) lds_wait8<8>(tr[0]); LAP_DKDV(2, tr[0])
    LAP_ISSUE_T(4, tr[0]) lds_wait8<8>(tr[1]); LAP_DKDV(3, tr[1])
    LAP_ISSUE_T(5, tr[1]) lds_wait8<8>(tr[0]); LAP_DKDV(4, tr[0])
    LAP_ISSUE_T(6, tr[0]) lds_wait8<8>(tr[1]); LAP_DKDV(5, tr[1])
    LAP_ISSUE_T(7, tr[1]) lds_wait8<8>(tr[0]); LAP_DKDV(6, tr[0])
    lds_wait8<0>(tr[1]); LAP_DKDV(7, tr[1])
#undef LAP_ISSUE_T
#undef LAP_DKDV
    } else {
      bf16x4 rd[2 * DF], rq[2 * DF];
      tr_burst<HD, DOFF>(va, rd);
      tr_burst<HD, QOFF>(va, rq);    // (the second burst's fence also covers the first)
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        acc_dv[d] = mfma16(join8(rd[2 * d], rd[2 * d + 1]), pb, acc_dv[d]);
        acc_dk[d] = mfma16(join8(rq[2 * d], rq[2 * d + 1]), db, acc_dk[d]);
      }
    }
  };

  if (items > 0) issue(0);
  __syncthreads();   // query info table complete
  {
    i32x4 sm = {0, 0, 0, 0};
    if (lane < ntq) sm = *reinterpret_cast<const i32x4*>(sSum + lane * 4);
    bool f = sm[3] == 32;
    int kor = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // lanes 0..15 hold the wave's 16 keys
      const int c = __builtin_amdgcn_readlane(kcls, r), x = __builtin_amdgcn_readlane(kidx, r);
      kor |= c;
      f = f && (c & sm[0]) != 0 && x <= sm[2];
    }
    skipmask = __ballot((kor & sm[1]) == 0);
    fastmask = __ballot(f);
  }
  for (int it = 0; it < items; it += 2) {
    // the parity of the stage follows the item index; the query tile index is item % ntq
    step(IC<0>{}, it % ntq);
    if (it + 1 < items) step(IC<1>{}, (it + 1) % ntq);
  }

  if (!vk) return;
  if (p.hsplit > 1) {
    // f32 partial of this head group: packed [b][joint key][kv head][HD]
    const long long n_all = (long long)p.B * Tk * p.NKV * HD;
    const long long row = ((long long)b * Tk + (kseg ? p.klen[0] : 0) + mykey) * p.NKV + hk;
    float* pk = p.part + (long long)hg * n_all + row * HD;
    float* pv = pk + (long long)p.hsplit * n_all;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      if (d * 16 + 4 * g >= HD) continue;
      *reinterpret_cast<f32x4*>(pk + d * 16 + 4 * g) = acc_dk[d];
      *reinterpret_cast<f32x4*>(pv + d * 16 + 4 * g) = acc_dv[d];
    }
    return;
  }
#pragma unroll
  for (int d = 0; d < DF; ++d) {
    if (d * 16 + 4 * g >= HD) continue;
    store4(p.dk[kseg] + koff + d * 16 + 4 * g, acc_dk[d], 1.0f);
    store4(p.dv[kseg] + koff + d * 16 + 4 * g, acc_dv[d], 1.0f);
  }
}
