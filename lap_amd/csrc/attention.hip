// Flash-style joint attention for the LAP hot path on gfx950: forward, and a
// two-kernel backward (dK/dV per key tile, dQ per query tile).
//
// Reference semantics (src/lap/models/backbones/gemma.py:234-272):
//   logits = einsum(q, k) in f32, masked with -2.38e38, softmax in f32, probs -> bf16,
//   encoded = einsum(probs, v);  q already carries head_dim^-0.5 (gemma.py:216).
// The [B,T,T] boolean mask of lap.py:303-364 / make_attn_mask is never built:
// each token carries one int32 (class bits << 24 | cumulative ar index) and
//   allowed(q, k) = (class(q) & class(k)) != 0  &&  idx(k) <= idx(q).
// Queries and keys of a sample are the concatenation of up to two segments (the
// two expert streams in training; KV-cache prefix + fresh suffix in serving), so
// neither the token streams nor the cache are ever concatenated in HBM.
//
// Tiling: 256 threads = 4 waves; a block owns 64 rows (queries for fwd / dQ, keys
// for dK/dV), each wave 16 of them; the other side streams through LDS in tiles
// of 64 rows, zero padded to DP = roundup(HD, 32) columns.  All products are
// v_mfma_f32_16x16x32_bf16.  S is computed transposed (S^T = K Q^T) so that a
// lane owns one query column: softmax statistics are per-lane scalars and the C/D
// registers of S^T are already the k-permuted B operand of O^T = V^T P^T.
#include "common.hpp"
#include "../../include/lap_hip.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LSE_EMPTY = 1.0e30f;   // lse of a fully masked row: exp(s - lse) == 0

struct AttnP {
  const bf16* q[2]; bf16* o[2]; const bf16* k[2]; const bf16* v[2];
  const bf16* d_o[2]; bf16* dq[2]; bf16* dk[2]; bf16* dv[2];
  int qlen[2], klen[2];
  int q_rs[2], kv_rs[2], o_rs[2];   // row strides (elements) of q/dq, k/v/dk/dv, o/dO
  const int32_t* qinfo; const int32_t* kinfo;
  float* lse; float* delta;
  int fuse_delta;   // backward, LDS-DMA kernels: the dQ launch computes delta itself and runs FIRST (no attn_delta_kernel)
  float scale;                      // logits = scale * q.k
  int B, NH, NKV, stop;
  int hsplit;                       // dK/dV: query heads of one kv head are split over hsplit blocks ...
  float* part;                      // ... which write f32 partials [2][hsplit][B*Tk*NKV*HD] reduced by attn_dkdv_reduce_kernel
  int nsplit;                       // fwd: key tiles split over nsplit blocks (grid.y); partial O in `part`, partial lse in `lpart`
  float* lpart;
};

template <int HD> struct Cfg {
  static constexpr int DP = (HD + 31) / 32 * 32;   // padded depth
  static constexpr int KS = DP / 32;               // 32-deep k-steps along d
  static constexpr int DF = DP / 16;               // 16-wide fragments along d
  static constexpr int STRIDE = DP * 2 + 32;       // LDS row stride in bytes: b128 and tr reads both conflict free
  static constexpr int TILE = 64 * STRIDE;
};

__device__ __forceinline__ bool mask_ok(int qi, int ki) {
  return (((qi >> 24) & (ki >> 24)) != 0) && ((ki & 0xffffff) <= (qi & 0xffffff));
}

// 64 x DP tile: global rows (row stride `rs` elements) -> LDS, zero filled outside [valid_rows) x [HD).
template <int HD>
__device__ __forceinline__ void load_tile(char* lds, const bf16* base, long long rs, int valid_rows) {
  constexpr int CH = Cfg<HD>::DP / 8;
  for (int idx = threadIdx.x; idx < 64 * CH; idx += 256) {
    const int row = idx / CH, c = idx % CH;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < valid_rows && c * 8 < HD) v = *reinterpret_cast<const bf16x8*>(base + row * rs + c * 8);
    *reinterpret_cast<bf16x8*>(lds + row * Cfg<HD>::STRIDE + c * 16) = v;
  }
}
// Operand with row/col index = tile row (lane i), contiguous-k along d: one ds_read_b128.
template <int HD>
__device__ __forceinline__ bf16x8 frag_kc(const char* t, int row0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(t + (row0 + i) * Cfg<HD>::STRIDE + (kk * 4 + g) * 16);
}
// Operand with row/col index = tile column (col0 + lane i), k = tile rows krow0 + {4g..4g+3, 16+4g..}: two tr reads.
template <int HD>
__device__ __forceinline__ bf16x8 frag_tr(const char* t, int krow0, int col0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const char* p = t + (krow0 + 4 * g + (i >> 2)) * Cfg<HD>::STRIDE + (col0 + (i & 3) * 4) * 2;
  bf16x4 lo = ds_read_tr(p);
  bf16x4 hi = ds_read_tr(p + 16 * Cfg<HD>::STRIDE);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// Register operand straight from a global row: lane group g takes d = 32kk + 8g .. +7 (zero outside [0,HD) / invalid row).
template <int HD>
__device__ __forceinline__ void load_row_frags(const bf16* row, bool valid, int lane, bf16x8 (&f)[Cfg<HD>::KS]) {
  const int g = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < Cfg<HD>::KS; ++kk) {
    const int d = kk * 32 + g * 8;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid && d < HD) v = *reinterpret_cast<const bf16x8*>(row + d);
    f[kk] = v;
  }
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = f2bf(a[e]); r[4 + e] = f2bf(b[e]); }
  return r;
}
// lane writes 4 consecutive d (8 bytes) of one row
__device__ __forceinline__ void store4(bf16* p, const f32x4& v, float s) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e] * s);
  *reinterpret_cast<bf16x4*>(p) = o;
}

// Stage the info words of one 64-row tile into LDS (ints [0,64)) with three summaries: [64] AND of the class bits,
// [65] OR of the class bits, [66] max (keys) / min (queries) of the index field.  Rows past `valid` get class 0.
// Executed by wave 0 between the two tile-load barriers.  With these a wave decides per tile, without touching the
// individual words, whether every (query, key) pair is allowed (no masking work at all), none is (tile skipped),
// or the tile is mixed (per-element path).  `want_min` selects min instead of max for the index summary.
__device__ __forceinline__ void stage_infos(int* sInfo, const int32_t* info, int valid, bool want_min) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int v = (lane < valid) ? info[lane] : 0;
    sInfo[lane] = v;
    int c_and = (lane < valid) ? (v >> 24) : 0, c_or = v >> 24;
    int idx = (lane < valid) ? (v & 0xffffff) : (want_min ? 0 : 0xffffff);
    if (valid < 64 && lane >= valid) { c_and = 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      c_and &= __shfl_xor(c_and, o, 64);
      c_or |= __shfl_xor(c_or, o, 64);
      const int other = __shfl_xor(idx, o, 64);
      idx = want_min ? min(idx, other) : max(idx, other);
    }
    if (lane == 0) { sInfo[64] = c_and; sInfo[65] = c_or; sInfo[66] = idx; }
  }
}

struct BlockId { int b, h, seg, tile; };
__device__ __forceinline__ BlockId decode_block(int len0, int len1, int nheads) {
  const int nt0 = (len0 + 63) >> 6, nt1 = (len1 + 63) >> 6, nt = nt0 + nt1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  BlockId r;
  const int t = bid % nt;
  r.h = (bid / nt) % nheads;
  r.b = bid / (nt * nheads);
  r.seg = t >= nt0;
  r.tile = r.seg ? t - nt0 : t;
  return r;
}

// =============================================================================== forward
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + C::TILE;
  int* sInfo = reinterpret_cast<int*>(smem + 2 * C::TILE);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const BlockId id = decode_block(p.qlen[0], p.qlen[1], p.NH);
  const int b = id.b, h = id.h;
  const int qlen = p.qlen[id.seg];
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int qinfo_off = id.seg ? p.qlen[0] : 0;
  const int myq = id.tile * 64 + w * 16 + i;
  const bool vq = myq < qlen;
  const int hk = h / (p.NH / p.NKV);

  bf16x8 qf[C::KS];
  load_row_frags<HD>(p.q[id.seg] + (b * (long long)qlen + myq) * p.q_rs[id.seg] + h * HD, vq, lane, qf);
  const int qi = (p.qinfo && vq) ? p.qinfo[(long long)b * Tq + qinfo_off + myq] : 0x7fffffff;

  float m = NEG_BIG, l = 0.f;
  f32x4 acc_o[C::DF];
#pragma unroll
  for (int d = 0; d < C::DF; ++d) acc_o[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // key tiles of both segments form one list; blockIdx.y takes a contiguous share of it (KV split for launches with
  // few query tiles, e.g. the batch-1 denoise step: 8 blocks would otherwise walk 10 tiles each, serially)
  const int nt0 = (p.klen[0] + 63) >> 6, ntk = nt0 + ((p.klen[1] + 63) >> 6);
  const int per = (ntk + p.nsplit - 1) / p.nsplit;
  const int gt0 = blockIdx.y * per, gt1 = min(ntk, gt0 + per);
  for (int gt = gt0; gt < gt1; ++gt) {
    const int ks = gt >= nt0, kt = ks ? gt - nt0 : gt;
    const int klen = p.klen[ks];
    const int kinfo_off = ks ? p.klen[0] : 0;
    const long long krs = p.kv_rs[ks];
    const bf16* kb = p.k[ks] + (long long)b * klen * krs + hk * HD;
    const bf16* vb = p.v[ks] + (long long)b * klen * krs + hk * HD;
    {
      __syncthreads();
      const int vr = min(64, klen - kt * 64);
      load_tile<HD>(sK, kb + (long long)kt * 64 * krs, krs, vr);
      load_tile<HD>(sV, vb + (long long)kt * 64 * krs, krs, vr);
      if (p.kinfo) stage_infos(sInfo, p.kinfo + (long long)b * Tk + kinfo_off + kt * 64, vr, false);
      __syncthreads();

      // tile-level mask decision (wave uniform): skip / no masking needed / per-element
      bool all_ok = vr == 64, none = false;
      if (p.kinfo) {
        const int kand = sInfo[64], kor = sInfo[65], kmax = sInfo[66];
        all_ok = all_ok && (((qi >> 24) & kand) != 0) && (kmax <= (qi & 0xffffff));
        none = vq && (((qi >> 24) & kor) == 0);
      }
      if (__all(none || !vq)) continue;       // nothing in this tile is visible to this wave's queries
      const bool fast = __all(all_ok || !vq);

      f32x4 s[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) s[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) s[nf] = mfma16(frag_kc<HD>(sK, nf * 16, kk, lane), qf[kk], s[nf]);

      // mask + online softmax; lane owns query column i and keys 16nf + 4g + r
      float smax = NEG_BIG;
      bool ok[4][4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        i32x4 kw = {0, 0, 0, 0};
        if (!fast && p.kinfo) kw = *reinterpret_cast<const i32x4*>(sInfo + nf * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool a = true;
          if (!fast) {
            const int key = kt * 64 + nf * 16 + 4 * g + r;
            a = key < klen;
            if (a && p.kinfo) a = mask_ok(qi, kw[r]);
          }
          ok[nf][r] = a;
          s[nf][r] *= p.scale;
          if (a) smax = fmaxf(smax, s[nf][r]);
        }
      }
      smax = fmaxf(smax, __shfl_xor(smax, 16, 64));
      smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
      const float m_new = fmaxf(m, smax);
      const float alpha = __expf(m - m_new);
      float lsum = 0.f;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = ok[nf][r] ? __expf(s[nf][r] - m_new) : 0.f;
          s[nf][r] = pv;
          lsum += pv;
        }
      l = l * alpha + lsum;
      m = m_new;
#pragma unroll
      for (int d = 0; d < C::DF; ++d) acc_o[d] *= alpha;
      const bf16x8 p0 = pack8(s[0], s[1]), p1 = pack8(s[2], s[3]);
#pragma unroll
      for (int d = 0; d < C::DF; ++d) {
        acc_o[d] = mfma16(frag_tr<HD>(sV, 0, d * 16, lane), p0, acc_o[d]);
        acc_o[d] = mfma16(frag_tr<HD>(sV, 32, d * 16, lane), p1, acc_o[d]);
      }
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!vq) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (p.nsplit > 1) {
    // partial result of this key share: normalised O (f32) + its log-sum-exp; combined by attn_fwd_combine_kernel
    const long long row = ((long long)blockIdx.y * p.B + b) * Tq + qinfo_off + myq;
    float* op = p.part + (row * p.NH + h) * HD;
#pragma unroll
    for (int d = 0; d < C::DF; ++d) {
      const int d0 = d * 16 + 4 * g;
      if (d0 < HD) *reinterpret_cast<f32x4*>(op + d0) = acc_o[d] * inv;
    }
    if (g == 0) p.lpart[(((long long)blockIdx.y * p.B + b) * p.NH + h) * Tq + qinfo_off + myq] = l > 0.f ? m + __logf(l) : NEG_BIG;
    return;
  }
  bf16* orow = p.o[id.seg] + (b * (long long)qlen + myq) * p.o_rs[id.seg] + h * HD;
#pragma unroll
  for (int d = 0; d < C::DF; ++d) {
    const int d0 = d * 16 + 4 * g;
    if (d0 < HD) store4(orow + d0, acc_o[d], inv);
  }
  if (p.lse && g == 0) p.lse[((long long)b * p.NH + h) * Tq + qinfo_off + myq] = l > 0.f ? m + __logf(l) : LSE_EMPTY;
}

// Combine the nsplit partial results: O = sum_i exp(lse_i - lse) O_i.  One thread per (b, q, h, 4 d).
template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(AttnP p) {
  const int Tq = p.qlen[0] + p.qlen[1];
  const long long n4 = (long long)p.B * Tq * p.NH * HD / 4;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n4) return;
  const int d0 = (int)(gid % (HD / 4)) * 4;
  long long r = gid / (HD / 4);
  const int h = (int)(r % p.NH); r /= p.NH;
  const int t = (int)(r % Tq);
  const int b = (int)(r / Tq);
  float mx = NEG_BIG;
  for (int sp = 0; sp < p.nsplit; ++sp) mx = fmaxf(mx, p.lpart[(((long long)sp * p.B + b) * p.NH + h) * Tq + t]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float den = 0.f;
  for (int sp = 0; sp < p.nsplit; ++sp) {
    const float li = p.lpart[(((long long)sp * p.B + b) * p.NH + h) * Tq + t];
    if (li <= NEG_BIG * 0.5f) continue;
    const float wgt = __expf(li - mx);
    den += wgt;
    acc += *reinterpret_cast<const f32x4*>(p.part + ((((long long)sp * p.B + b) * Tq + t) * p.NH + h) * HD + d0) * wgt;
  }
  const int seg = t >= p.qlen[0];
  const int tt = seg ? t - p.qlen[0] : t;
  store4(p.o[seg] + (b * (long long)p.qlen[seg] + tt) * p.o_rs[seg] + h * HD + d0, acc, den > 0.f ? 1.0f / den : 0.f);
  if (p.lse && d0 == 0) p.lse[((long long)b * p.NH + h) * Tq + t] = den > 0.f ? mx + __logf(den) : LSE_EMPTY;
}

// ====================================================================== delta = rowsum(dO * O)
// 16 lanes per (b, t, h) row (a wave covers 4 rows: head size 72 keeps 9 of 16 lanes busy instead of 9 of 64; head size
// 256 has two 16-byte loads per tensor and lane in flight); the rows of one token are contiguous, so a wave reads
// 4 * HD * 2 contiguous bytes per tensor.
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnP p) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l16 = lane & 15;
  const int Tq = p.qlen[0] + p.qlen[1];
  const long long item = ((long long)blockIdx.x * 4 + w) * 4 + (lane >> 4);     // (b, t, h)
  const bool live = item < (long long)p.B * Tq * p.NH;
  const long long it = live ? item : 0;
  const int h = (int)(it % p.NH);
  const int t = (int)((it / p.NH) % Tq);
  const int b = (int)(it / ((long long)p.NH * Tq));
  const int seg = t >= p.qlen[0];
  const int tt = seg ? t - p.qlen[0] : t;
  const long long off = (b * (long long)p.qlen[seg] + tt) * p.o_rs[seg] + h * HD;
  float acc = 0.f;
#pragma unroll
  for (int c0 = 0; c0 < HD; c0 += 128) {
    const int c = c0 + l16 * 8;
    if (c < HD) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(p.o[seg] + off + c);
      bf16x8 d = *reinterpret_cast<const bf16x8*>(p.d_o[seg] + off + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)d[e];
    }
  }
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if (l16 == 0 && live) p.delta[((long long)b * p.NH + h) * Tq + t] = acc;
}

// ======================================================================== backward: dK, dV
// Block = (b, kv head, key tile); wave owns 16 keys (lane column i); loops over the
// query heads sharing this kv head and over all query tiles.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnP p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;
  char* sD = smem + C::TILE;
  int* sInfo = reinterpret_cast<int*>(smem + 2 * C::TILE);        // 64 query info words + 3 summaries
  float* sLse = reinterpret_cast<float*>(smem + 2 * C::TILE + 320);  // lse and delta of the 64 query rows
  float* sDl = sLse + 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const BlockId id = decode_block(p.klen[0], p.klen[1], p.NKV * p.hsplit);
  const int b = id.b, hk = id.h / p.hsplit, hg = id.h % p.hsplit, kseg = id.seg;
  const int klen = p.klen[kseg];
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int mykey = id.tile * 64 + w * 16 + i;
  const bool vk = mykey < klen;
  const long long koff = (b * (long long)klen + mykey) * p.kv_rs[kseg] + hk * HD;

  bf16x8 kf[C::KS], vf[C::KS];
  load_row_frags<HD>(p.k[kseg] + koff, vk, lane, kf);
  load_row_frags<HD>(p.v[kseg] + koff, vk, lane, vf);
  const int ki = (p.kinfo && vk) ? p.kinfo[(long long)b * Tk + (kseg ? p.klen[0] : 0) + mykey] : 0;

  f32x4 acc_dk[C::DF], acc_dv[C::DF];
#pragma unroll
  for (int d = 0; d < C::DF; ++d) { acc_dk[d] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_dv[d] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int hpk = p.NH / p.NKV, hpg = hpk / p.hsplit;
  for (int hh = hg * hpg; hh < (hg + 1) * hpg; ++hh) {
    const int h = hk * hpk + hh;
    for (int qs = 0; qs < 2; ++qs) {
      const int qlen = p.qlen[qs];
      if (qlen == 0) continue;
      if (p.stop && qs == 1 && kseg == 0) continue;   // stop_action_to_vlm_grad
      const int qinfo_off = qs ? p.qlen[0] : 0;
      const long long qrs = p.q_rs[qs], ors = p.o_rs[qs];
      const bf16* qb = p.q[qs] + (long long)b * qlen * qrs + h * HD;
      const bf16* dob = p.d_o[qs] + (long long)b * qlen * ors + h * HD;
      const float* lse = p.lse + ((long long)b * p.NH + h) * Tq + qinfo_off;
      const float* dl = p.delta + ((long long)b * p.NH + h) * Tq + qinfo_off;
      for (int qt = 0; qt * 64 < qlen; ++qt) {
        __syncthreads();
        const int vr = min(64, qlen - qt * 64);
        load_tile<HD>(sQ, qb + (long long)qt * 64 * qrs, qrs, vr);
        load_tile<HD>(sD, dob + (long long)qt * 64 * ors, ors, vr);
        if (p.qinfo) stage_infos(sInfo, p.qinfo + (long long)b * Tq + qinfo_off + qt * 64, vr, true);
        if (threadIdx.x >= 64 && threadIdx.x < 128) {
          const int r = threadIdx.x - 64;
          sLse[r] = r < vr ? lse[qt * 64 + r] : LSE_EMPTY;
          sDl[r] = r < vr ? dl[qt * 64 + r] : 0.f;
        }
        __syncthreads();
        bool all_ok = vk && vr == 64, none = false;
        if (p.qinfo) {
          const int qand = sInfo[64], qor = sInfo[65], qmin = sInfo[66];
          all_ok = all_ok && (((ki >> 24) & qand) != 0) && ((ki & 0xffffff) <= qmin);
          none = ((ki >> 24) & qor) == 0;
        }
        if (__all(none || !vk)) continue;      // no query of this tile sees this wave's keys
        const bool fast = __all(all_ok);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) { s[qf] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[qf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
          for (int qf = 0; qf < 4; ++qf) {
            s[qf] = mfma16(frag_kc<HD>(sQ, qf * 16, kk, lane), kf[kk], s[qf]);     // S[q][key]
            dp[qf] = mfma16(frag_kc<HD>(sD, qf * 16, kk, lane), vf[kk], dp[qf]);   // dP[q][key]
          }
        // lane: key column i, query rows 16qf + 4g + r
#pragma unroll
        for (int qf = 0; qf < 4; ++qf) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + qf * 16 + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDl + qf * 16 + 4 * g);
          i32x4 qw = {0, 0, 0, 0};
          if (!fast && p.qinfo) qw = *reinterpret_cast<const i32x4*>(sInfo + qf * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bool a = true;
            if (!fast) {
              a = vk && (qt * 64 + qf * 16 + 4 * g + r) < qlen;
              if (a && p.qinfo) a = mask_ok(qw[r], ki);
            }
            float pv = 0.f, ds = 0.f;
            if (a) {
              pv = __expf(s[qf][r] * p.scale - l4[r]);
              ds = pv * (dp[qf][r] - d4[r]) * p.scale;
            }
            s[qf][r] = pv;
            dp[qf][r] = ds;
          }
        }
        const bf16x8 p0 = pack8(s[0], s[1]), p1 = pack8(s[2], s[3]);
        const bf16x8 d0 = pack8(dp[0], dp[1]), d1 = pack8(dp[2], dp[3]);
#pragma unroll
        for (int d = 0; d < C::DF; ++d) {
          acc_dv[d] = mfma16(frag_tr<HD>(sD, 0, d * 16, lane), p0, acc_dv[d]);    // dV^T[d][key] += dO^T P
          acc_dv[d] = mfma16(frag_tr<HD>(sD, 32, d * 16, lane), p1, acc_dv[d]);
          acc_dk[d] = mfma16(frag_tr<HD>(sQ, 0, d * 16, lane), d0, acc_dk[d]);    // dK^T[d][key] += Q^T dS
          acc_dk[d] = mfma16(frag_tr<HD>(sQ, 32, d * 16, lane), d1, acc_dk[d]);
        }
      }
    }
  }
  if (!vk) return;
  if (p.hsplit > 1) {
    // f32 partial of this head group: packed [b][joint key][kv head][HD]
    const long long n_all = (long long)p.B * Tk * p.NKV * HD;
    const long long row = ((long long)b * Tk + (kseg ? p.klen[0] : 0) + mykey) * p.NKV + hk;
    float* pk = p.part + (long long)hg * n_all + row * HD;
    float* pv = pk + (long long)p.hsplit * n_all;
#pragma unroll
    for (int d = 0; d < C::DF; ++d) {
      const int d0 = d * 16 + 4 * g;
      if (d0 < HD) {
        *reinterpret_cast<f32x4*>(pk + d0) = acc_dk[d];
        *reinterpret_cast<f32x4*>(pv + d0) = acc_dv[d];
      }
    }
    return;
  }
#pragma unroll
  for (int d = 0; d < C::DF; ++d) {
    const int d0 = d * 16 + 4 * g;
    if (d0 < HD) {
      store4(p.dk[kseg] + koff + d0, acc_dk[d], 1.0f);
      store4(p.dv[kseg] + koff + d0, acc_dv[d], 1.0f);
    }
  }
}

// Sum the hsplit f32 partials and write bf16 dK / dV with the caller's row stride.
template <int HD>
__global__ __launch_bounds__(256) void attn_dkdv_reduce_kernel(AttnP p) {
  const int Tk = p.klen[0] + p.klen[1];
  const long long n_all = (long long)p.B * Tk * p.NKV * HD;
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n_all) return;
  f32x4 sk = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
  for (int hgp = 0; hgp < p.hsplit; ++hgp) {
    sk += *reinterpret_cast<const f32x4*>(p.part + (long long)hgp * n_all + i4);
    sv += *reinterpret_cast<const f32x4*>(p.part + ((long long)p.hsplit + hgp) * n_all + i4);
  }
  const int d0 = (int)(i4 % HD);
  long long r = i4 / HD;
  const int hk = (int)(r % p.NKV); r /= p.NKV;
  const int t = (int)(r % Tk);
  const int b = (int)(r / Tk);
  const int seg = t >= p.klen[0];
  const int tt = seg ? t - p.klen[0] : t;
  const long long off = (b * (long long)p.klen[seg] + tt) * p.kv_rs[seg] + hk * HD + d0;
  store4(p.dk[seg] + off, sk, 1.0f);
  store4(p.dv[seg] + off, sv, 1.0f);
}

// ============================================================================ backward: dQ
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnP p) {
  using C = Cfg<HD>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + C::TILE;
  int* sInfo = reinterpret_cast<int*>(smem + 2 * C::TILE);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const BlockId id = decode_block(p.qlen[0], p.qlen[1], p.NH);
  const int b = id.b, h = id.h;
  const int qlen = p.qlen[id.seg];
  const int Tq = p.qlen[0] + p.qlen[1], Tk = p.klen[0] + p.klen[1];
  const int qinfo_off = id.seg ? p.qlen[0] : 0;
  const int myq = id.tile * 64 + w * 16 + i;
  const bool vq = myq < qlen;
  const int hk = h / (p.NH / p.NKV);
  const long long qoff = (b * (long long)qlen + myq) * p.q_rs[id.seg] + h * HD;
  const long long ooff = (b * (long long)qlen + myq) * p.o_rs[id.seg] + h * HD;

  bf16x8 qf[C::KS], dof[C::KS];
  load_row_frags<HD>(p.q[id.seg] + qoff, vq, lane, qf);
  load_row_frags<HD>(p.d_o[id.seg] + ooff, vq, lane, dof);
  const int qi = (p.qinfo && vq) ? p.qinfo[(long long)b * Tq + qinfo_off + myq] : 0x7fffffff;
  const float lse_q = vq ? p.lse[((long long)b * p.NH + h) * Tq + qinfo_off + myq] : LSE_EMPTY;
  const float dl_q = vq ? p.delta[((long long)b * p.NH + h) * Tq + qinfo_off + myq] : 0.f;

  f32x4 acc[C::DF];
#pragma unroll
  for (int d = 0; d < C::DF; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ks = 0; ks < 2; ++ks) {
    const int klen = p.klen[ks];
    if (klen == 0) continue;
    const int kinfo_off = ks ? p.klen[0] : 0;
    const long long krs = p.kv_rs[ks];
    const bf16* kb = p.k[ks] + (long long)b * klen * krs + hk * HD;
    const bf16* vb = p.v[ks] + (long long)b * klen * krs + hk * HD;
    for (int kt = 0; kt * 64 < klen; ++kt) {
      __syncthreads();
      const int vr = min(64, klen - kt * 64);
      load_tile<HD>(sK, kb + (long long)kt * 64 * krs, krs, vr);
      load_tile<HD>(sV, vb + (long long)kt * 64 * krs, krs, vr);
      if (p.kinfo) stage_infos(sInfo, p.kinfo + (long long)b * Tk + kinfo_off + kt * 64, vr, false);
      __syncthreads();
      bool all_ok = vr == 64, none = false;
      if (p.kinfo) {
        const int kand = sInfo[64], kor = sInfo[65], kmax = sInfo[66];
        all_ok = all_ok && (((qi >> 24) & kand) != 0) && (kmax <= (qi & 0xffffff));
        none = vq && (((qi >> 24) & kor) == 0);
      }
      if (__all(none || !vq)) continue;
      const bool fast = __all(all_ok || !vq);
      f32x4 s[4], dp[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) { s[nf] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[nf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int kk = 0; kk < C::KS; ++kk)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
          s[nf] = mfma16(frag_kc<HD>(sK, nf * 16, kk, lane), qf[kk], s[nf]);      // S^T[key][q]
          dp[nf] = mfma16(frag_kc<HD>(sV, nf * 16, kk, lane), dof[kk], dp[nf]);   // dP^T[key][q]
        }
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        i32x4 kw = {0, 0, 0, 0};
        if (!fast && p.kinfo) kw = *reinterpret_cast<const i32x4*>(sInfo + nf * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool a = vq;
          if (!fast) {
            const int key = kt * 64 + nf * 16 + 4 * g + r;
            a = vq && key < klen;
            if (a && p.kinfo) a = mask_ok(qi, kw[r]);
          }
          dp[nf][r] = a ? __expf(s[nf][r] * p.scale - lse_q) * (dp[nf][r] - dl_q) * p.scale : 0.f;
        }
      }
      const bf16x8 d0 = pack8(dp[0], dp[1]), d1 = pack8(dp[2], dp[3]);
#pragma unroll
      for (int d = 0; d < C::DF; ++d) {
        acc[d] = mfma16(frag_tr<HD>(sK, 0, d * 16, lane), d0, acc[d]);    // dQ^T[d][q] += K^T dS^T
        acc[d] = mfma16(frag_tr<HD>(sK, 32, d * 16, lane), d1, acc[d]);
      }
    }
  }
  if (!vq) return;
#pragma unroll
  for (int d = 0; d < C::DF; ++d) {
    const int d0 = d * 16 + 4 * g;
    if (d0 < HD) store4(p.dq[id.seg] + qoff + d0, acc[d], 1.0f);
  }
}

#include "attention_dma.hpp"

template <typename K>
int set_lds(K kernel, int bytes) {
  if (bytes > 65536) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

#include "attention_serve.hpp"
#include "serve_chain.hpp"
#include "serve_chain_tp.hpp"

// Tuning / test knob (lap_attention_set_variant): -1 = automatic; 0 = generic kernels also for HD = 256;
// 1 = the HD = 256 LDS-DMA kernels (what automatic picks whenever their LDS info table fits); 2 = as 1 with the backward's
// delta as a separate pass in front (the arrangement before round 4; A/B measurements).
int g_attn_variant = -1;
int attn_variant() { return g_attn_variant; }

// HD = 256 LDS-DMA kernels keep the info words of every key (query) tile in LDS: 144 bytes per 32-row tile
constexpr int DMA_MAX_INFO_BYTES = 16384;
bool dma_path_ok(const AttnP& p) {
  const int ntk = (p.klen[0] + 31) / 32 + (p.klen[1] + 31) / 32, ntq = (p.qlen[0] + 31) / 32 + (p.qlen[1] + 31) / 32;
  // (and the per-wave tile decisions in 64-bit masks: at most 64 streamed tiles per block)
  const int per = (ntk + p.nsplit - 1) / p.nsplit;
  return attn_variant() != 0 && max(ntk, ntq) * 144 <= DMA_MAX_INFO_BYTES && per <= 64 && ntq <= 64 && p.scale > 0.f;
}

template <int HD>
int launch_fwd_dma(const AttnP& p, hipStream_t s) {
  const int nt = (p.qlen[0] + p.qlen[1] + 63) / 64;
  const int ntk = (p.klen[0] + 31) / 32 + (p.klen[1] + 31) / 32;
  const int lds = 4 * DmaCfg<HD>::TILE + ntk * 144;
  auto kern = attn_dma_q_kernel<HD, 0>;
  if (int e = set_lds(kern, lds)) return e;
  hipLaunchKernelGGL(kern, dim3(p.B * p.NH * nt, p.nsplit), dim3(256), lds, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

template <int HD>
int launch_fwd(const AttnP& p, hipStream_t s) {
  if constexpr (HD == 256 || HD == 72) {
    if (dma_path_ok(p)) {
      if (int e = launch_fwd_dma<HD>(p, s)) return e;
      if (p.nsplit > 1) {
        const long long n4 = (long long)p.B * (p.qlen[0] + p.qlen[1]) * p.NH * HD / 4;
        hipLaunchKernelGGL(attn_fwd_combine_kernel<HD>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
        LAP_CHECK_LAUNCH();
      }
      return LAP_OK;
    }
  }
  const int nt = (p.qlen[0] + 63) / 64 + (p.qlen[1] + 63) / 64;
  const int lds = 2 * Cfg<HD>::TILE + 1024;   // + staged info words / lse / delta
  if (int e = set_lds(attn_fwd_kernel<HD>, lds)) return e;
  hipLaunchKernelGGL(attn_fwd_kernel<HD>, dim3(p.B * p.NH * nt, p.nsplit), dim3(256), lds, s, p);
  LAP_CHECK_LAUNCH();
  if (p.nsplit > 1) {
    const long long n4 = (long long)p.B * (p.qlen[0] + p.qlen[1]) * p.NH * HD / 4;
    hipLaunchKernelGGL(attn_fwd_combine_kernel<HD>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
    LAP_CHECK_LAUNCH();
  }
  return LAP_OK;
}
template <int HD>
int launch_bwd(const AttnP& p, hipStream_t s) {
  const int Tq = p.qlen[0] + p.qlen[1];
  const int ntk = (p.klen[0] + 63) / 64 + (p.klen[1] + 63) / 64;
  auto reduce = [&]() -> int {
    if (p.hsplit > 1) {
      const long long n4 = (long long)p.B * (p.klen[0] + p.klen[1]) * p.NKV * HD / 4;
      hipLaunchKernelGGL(attn_dkdv_reduce_kernel<HD>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, p);
      LAP_CHECK_LAUNCH();
    }
    return LAP_OK;
  };
  auto delta = [&]() -> int {
    const long long items = (long long)p.B * Tq * p.NH;
    hipLaunchKernelGGL(attn_delta_kernel<HD>, dim3((unsigned)((items + 15) / 16)), dim3(256), 0, s, p);
    LAP_CHECK_LAUNCH();
    return LAP_OK;
  };
  if constexpr (HD == 256 || HD == 72) {
    if (dma_path_ok(p)) {
      // dQ first: it computes delta = rowsum(dO o O) for its own rows and publishes it for the dK / dV launch behind it
      // (variant 2: the separate delta pass in front, for A/B measurements)
      AttnP q = p;
      q.fuse_delta = attn_variant() != 2;
      if (!q.fuse_delta) { if (int e = delta()) return e; }
      const int nt = (Tq + 63) / 64;
      const int ntk32 = (p.klen[0] + 31) / 32 + (p.klen[1] + 31) / 32;
      const int lds_q = 4 * DmaCfg<HD>::TILE + ntk32 * 144;
      auto kq = attn_dma_q_kernel<HD, 1>;
      if (int e = set_lds(kq, lds_q)) return e;
      hipLaunchKernelGGL(kq, dim3(p.B * p.NH * nt, 1), dim3(256), lds_q, s, q);
      LAP_CHECK_LAUNCH();
      const int ntq32 = (p.qlen[0] + 31) / 32 + (p.qlen[1] + 31) / 32;
      const int lds_kv = 4 * DmaCfg<HD>::TILE + 1024 + ntq32 * 144;
      auto kkv = attn_dma_kv_kernel<HD>;
      if (int e = set_lds(kkv, lds_kv)) return e;
      hipLaunchKernelGGL(kkv, dim3(p.B * p.NKV * p.hsplit * ntk), dim3(256), lds_kv, s, p);
      LAP_CHECK_LAUNCH();
      return reduce();
    }
  }
  const int lds = 2 * Cfg<HD>::TILE + 1024;   // + staged info words / lse / delta
  if (int e = delta()) return e;
  if (int e = set_lds(attn_bwd_dkdv_kernel<HD>, lds)) return e;
  hipLaunchKernelGGL(attn_bwd_dkdv_kernel<HD>, dim3(p.B * p.NKV * p.hsplit * ntk), dim3(256), lds, s, p);
  LAP_CHECK_LAUNCH();
  if (int e = reduce()) return e;
  const int ntq = (p.qlen[0] + 63) / 64 + (p.qlen[1] + 63) / 64;
  if (int e = set_lds(attn_bwd_dq_kernel<HD>, lds)) return e;
  hipLaunchKernelGGL(attn_bwd_dq_kernel<HD>, dim3(p.B * p.NH * ntq), dim3(256), lds, s, p);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

bool check_common(int B, int NH, int NKV, int HD, const int* qlen, const int* klen) {
  if (B <= 0 || NH <= 0 || NKV <= 0 || NH % NKV) return false;
  if (HD != 16 && HD != 72 && HD != 256) return false;
  if (qlen[0] < 0 || qlen[1] < 0 || klen[0] < 0 || klen[1] < 0) return false;
  if (qlen[0] + qlen[1] == 0 || klen[0] + klen[1] == 0) return false;
  return true;
}

}  // namespace

extern "C" int lap_attention_set_variant(int variant) {
  if (variant < -1 || variant > 2) return LAP_ERR_ARG;
  g_attn_variant = variant;
  return LAP_OK;
}

extern "C" int lap_attention_fwd(const lap_attn_fwd_args* a, void* stream) {
  if (!a || !check_common(a->B, a->NH, a->NKV, a->HD, a->q_len, a->k_len)) return LAP_ERR_ARG;
  AttnP p = {};
  for (int s = 0; s < 2; ++s) {
    p.q[s] = (const bf16*)a->q[s]; p.o[s] = (bf16*)a->o[s];
    p.k[s] = (const bf16*)a->k[s]; p.v[s] = (const bf16*)a->v[s];
    p.qlen[s] = a->q_len[s]; p.klen[s] = a->k_len[s];
    p.q_rs[s] = a->q_rs[s] ? a->q_rs[s] : a->NH * a->HD;
    p.o_rs[s] = a->o_rs[s] ? a->o_rs[s] : a->NH * a->HD;
    p.kv_rs[s] = a->kv_rs[s] ? a->kv_rs[s] : a->NKV * a->HD;
    if ((p.q_rs[s] | p.o_rs[s] | p.kv_rs[s]) & 7) return LAP_ERR_ARG;
    if (p.qlen[s] && (!p.q[s] || !p.o[s])) return LAP_ERR_ARG;
    if (p.klen[s] && (!p.k[s] || !p.v[s])) return LAP_ERR_ARG;
  }
  if ((a->qinfo == nullptr) != (a->kinfo == nullptr)) return LAP_ERR_ARG;
  p.qinfo = a->qinfo; p.kinfo = a->kinfo; p.lse = a->lse;
  p.scale = a->scale;
  p.nsplit = 1; p.hsplit = 1;
  if (a->nsplit > 1) {
    const long long Tq = a->q_len[0] + a->q_len[1];
    const long long need = (long long)a->nsplit * a->B * Tq * a->NH * (a->HD + 1);
    if (!a->scratch || a->scratch_floats < need || (a->HD & 3)) return LAP_ERR_ARG;
    p.nsplit = a->nsplit;
    p.part = a->scratch;
    p.lpart = a->scratch + (long long)a->nsplit * a->B * Tq * a->NH * a->HD;
  }
  p.B = a->B; p.NH = a->NH; p.NKV = a->NKV;
  hipStream_t s = (hipStream_t)stream;
  switch (a->HD) {
    case 16: return launch_fwd<16>(p, s);
    case 72: return launch_fwd<72>(p, s);
    case 256: return launch_fwd<256>(p, s);
  }
  return LAP_ERR_ARG;
}

extern "C" int lap_attention_serve_splits(int k_len0, int k_len1, int B, int NH, int q_len1) {
  if (k_len0 < 0 || k_len1 < 0 || k_len0 + k_len1 == 0 || B <= 0 || NH <= 0 || q_len1 <= 0) return 0;
  int cap = serve_run_cap(B, NH, q_len1);      // the persistent chain's rule: same key runs, same bits
  if (serve_runs(k_len0, k_len1, cap).nruns < 1) cap = (k_len0 + SV_KEYS - 1) / SV_KEYS + (k_len1 + SV_KEYS - 1) / SV_KEYS;
  return cap <= SV_MAX_RUNS ? cap : 0;
}

extern "C" int lap_attention_serve(const lap_attn_fwd_args* a, void* stream) {
  if (!a || !check_common(a->B, a->NH, a->NKV, a->HD, a->q_len, a->k_len)) return LAP_ERR_ARG;
  if (a->HD != 256 || a->q_len[0] != 0 || a->q_len[1] > 64 || a->lse || !a->q[1] || !a->o[1]) return LAP_ERR_ARG;
  AttnP p = {};
  for (int s = 0; s < 2; ++s) {
    p.q[s] = (const bf16*)a->q[s]; p.o[s] = (bf16*)a->o[s];
    p.k[s] = (const bf16*)a->k[s]; p.v[s] = (const bf16*)a->v[s];
    p.qlen[s] = a->q_len[s]; p.klen[s] = a->k_len[s];
    p.q_rs[s] = a->q_rs[s] ? a->q_rs[s] : a->NH * a->HD;
    p.o_rs[s] = a->o_rs[s] ? a->o_rs[s] : a->NH * a->HD;
    p.kv_rs[s] = a->kv_rs[s] ? a->kv_rs[s] : a->NKV * a->HD;
    if ((p.q_rs[s] | p.o_rs[s] | p.kv_rs[s]) & 7) return LAP_ERR_ARG;
    if (p.klen[s] && (!p.k[s] || !p.v[s])) return LAP_ERR_ARG;
    if ((long long)p.klen[s] * p.kv_rs[s] * 2 >= 0x7fffffffLL) return LAP_ERR_ARG;
  }
  if ((a->qinfo == nullptr) != (a->kinfo == nullptr)) return LAP_ERR_ARG;
  p.qinfo = a->qinfo; p.kinfo = a->kinfo; p.scale = a->scale;
  if (!(a->scale > 0.f)) return LAP_ERR_ARG;
  const long long need = (long long)a->nsplit * a->B * a->q_len[1] * a->NH * (a->HD + 1);
  if (a->nsplit < 1 || !a->scratch || a->scratch_floats < need) return LAP_ERR_ARG;
  p.nsplit = a->nsplit; p.hsplit = 1;
  p.part = a->scratch;
  p.lpart = a->scratch + (long long)a->nsplit * a->B * a->q_len[1] * a->NH * a->HD;
  p.B = a->B; p.NH = a->NH; p.NKV = a->NKV;
  return launch_serve(p, (hipStream_t)stream);
}

// ---- the denoise step's 18 layers in one persistent launch (serve_chain.hpp)
extern "C" int lap_serve_chain_ok(int B, int S, int D, int H, int NH, int HD, int NKV, int prefix_len) {
  return chain_ok(B, S, D, H, NH, HD, NKV, prefix_len) && chain_device_ok() ? 1 : 0;
}

extern "C" int lap_serve_chain_tp_ok(int B, int S, int D, int H, int NH, int HD, int NKV, int prefix_len) {
  return chain_tp_ok(B, S, D, H, NH, HD, NKV, prefix_len) && chain_device_ok() ? 1 : 0;
}

extern "C" int lap_serve_chain_counter_words(void) { return CH_CTR_WORDS; }

// 0: every barrier of every launch so far completed; 1: a block gave up waiting (the results of that launch are invalid)
extern "C" int lap_serve_chain_status(const unsigned* counters, int* status) {
  if (!counters || !status) return LAP_ERR_ARG;
  unsigned v = 0;
  if (hipError_t e = hipMemcpy(&v, counters + CH_CTR_STRIDE * CH_CTR_ERR, sizeof(v), hipMemcpyDeviceToHost); e != hipSuccess) return (int)e;
  *status = v != 0;
  return LAP_OK;
}

extern "C" int lap_serve_chain(const lap_serve_chain_args* a, void* stream) {
  if (!a || a->depth < 1 || a->depth > CH_MAX_DEPTH || a->depth > LAP_CHAIN_MAX_DEPTH) return LAP_ERR_ARG;
  if (!chain_ok(a->B, a->S, a->D, a->H, a->NH, a->HD, 1, a->prefix_len)) return LAP_ERR_ARG;
  if (!a->x_in || !a->x_out || !a->mod || !a->rope_table || !a->q || !a->k || !a->v || !a->o || !a->xa || !a->act || !a->attn_scratch ||
      !a->counters || (a->mod_slot_stride & 7) || a->mod_slot_stride < 3 * a->D || (a->packed && !a->xs))
    return LAP_ERR_ARG;
  if ((a->qinfo == nullptr) != (a->kinfo == nullptr)) return LAP_ERR_ARG;
  const int M = a->B * a->S, kv_rs = a->kv_rs ? a->kv_rs : a->HD;
  if ((kv_rs & 7) || (long long)a->prefix_len * kv_rs * 2 >= 0x7fffffffLL) return LAP_ERR_ARG;
  ChainP c = {};
  c.depth = a->depth; c.M = M; c.rps = a->S; c.D = a->D; c.H = a->H; c.NH = a->NH; c.HD = a->HD;
  c.x_in = (const bf16*)a->x_in; c.x_out = (bf16*)a->x_out; c.mod = (const bf16*)a->mod; c.slot_ld = a->mod_slot_stride;
  for (int l = 0; l < a->depth; ++l) {
    if (!a->wqkv[l] || !a->wo[l] || !a->wgu[l] || !a->wd[l] || (a->prefix_len > 0 && (!a->cache_k[l] || !a->cache_v[l]))) return LAP_ERR_ARG;
    if (((uintptr_t)a->wqkv[l] | (uintptr_t)a->wo[l] | (uintptr_t)a->wgu[l] | (uintptr_t)a->wd[l] | (uintptr_t)a->cache_k[l] | (uintptr_t)a->cache_v[l]) & 15)
      return LAP_ERR_ARG;
    c.wqkv[l] = (const bf16*)a->wqkv[l]; c.wo[l] = (const bf16*)a->wo[l]; c.wgu[l] = (const bf16*)a->wgu[l]; c.wd[l] = (const bf16*)a->wd[l];
    c.ck[l] = (const bf16*)a->cache_k[l]; c.cv[l] = (const bf16*)a->cache_v[l];
  }
  c.sr = serve_runs(a->prefix_len, a->S, serve_run_cap(a->B, a->NH, a->S));
  const long long need = (long long)c.sr.nruns * M * a->NH * (a->HD + 1);
  if (a->attn_scratch_floats < need) return LAP_ERR_ARG;
  AttnP& p = c.attn;
  p.qlen[0] = 0; p.qlen[1] = a->S; p.klen[0] = a->prefix_len; p.klen[1] = a->S;
  p.q_rs[0] = p.q_rs[1] = a->NH * a->HD; p.o_rs[0] = p.o_rs[1] = a->NH * a->HD; p.kv_rs[0] = kv_rs; p.kv_rs[1] = a->HD;
  p.qinfo = a->qinfo; p.kinfo = a->kinfo; p.scale = 1.0f;       // (q carries head_dim^-0.5 from the qkv stage, gemma.py:216)
  p.B = a->B; p.NH = a->NH; p.NKV = 1; p.nsplit = c.sr.nruns; p.hsplit = 1;
  p.part = a->attn_scratch; p.lpart = a->attn_scratch + (long long)c.sr.nruns * M * a->NH * a->HD;
  c.rope = a->rope_table; c.q_scale = a->q_scale; c.eps = a->eps;
  c.q = (bf16*)a->q; c.k = (bf16*)a->k; c.v = (bf16*)a->v; c.o = (bf16*)a->o; c.xa = (bf16*)a->xa; c.act = (bf16*)a->act;
  c.xs = (bf16*)a->xs;
  c.ctrs = a->counters;
  c.clk = (unsigned long long*)a->debug_clock;
  if (a->packed == 2) {      // tensor parallel over the XCDs (serve_chain_tp.hpp)
    if (!chain_tp_ok(a->B, a->S, a->D, a->H, a->NH, a->HD, 1, a->prefix_len) || !a->tp_slabs || !a->tp_xs || !a->tp_xn || !a->tp_k || !a->tp_v)
      return LAP_ERR_ARG;
    TpP t;
    t.c = c;
    t.slab_o = a->tp_slabs; t.slab_d = a->tp_slabs + (long long)TP_X * 64 * a->D;
    t.xs8 = (bf16*)a->tp_xs; t.xn8 = (bf16*)a->tp_xn; t.k8 = (bf16*)a->tp_k; t.v8 = (bf16*)a->tp_v;
    return launch_chain_tp(t, (hipStream_t)stream);
  }
  return launch_chain(c, a->packed != 0, (hipStream_t)stream);
}

extern "C" int lap_attention_bwd(const lap_attn_bwd_args* a, void* stream) {
  if (!a || !check_common(a->B, a->NH, a->NKV, a->HD, a->q_len, a->k_len)) return LAP_ERR_ARG;
  if (!a->lse || !a->delta) return LAP_ERR_ARG;
  AttnP p = {};
  for (int s = 0; s < 2; ++s) {
    p.q[s] = (const bf16*)a->q[s]; p.o[s] = (bf16*)a->o[s]; p.d_o[s] = (const bf16*)a->d_o[s];
    p.k[s] = (const bf16*)a->k[s]; p.v[s] = (const bf16*)a->v[s];
    p.dq[s] = (bf16*)a->dq[s]; p.dk[s] = (bf16*)a->dk[s]; p.dv[s] = (bf16*)a->dv[s];
    p.qlen[s] = a->q_len[s]; p.klen[s] = a->k_len[s];
    p.q_rs[s] = a->q_rs[s] ? a->q_rs[s] : a->NH * a->HD;
    p.o_rs[s] = a->o_rs[s] ? a->o_rs[s] : a->NH * a->HD;
    p.kv_rs[s] = a->kv_rs[s] ? a->kv_rs[s] : a->NKV * a->HD;
    if ((p.q_rs[s] | p.o_rs[s] | p.kv_rs[s]) & 7) return LAP_ERR_ARG;
    if (p.qlen[s] && (!p.q[s] || !p.o[s] || !p.d_o[s] || !p.dq[s])) return LAP_ERR_ARG;
    if (p.klen[s] && (!p.k[s] || !p.v[s] || !p.dk[s] || !p.dv[s])) return LAP_ERR_ARG;
  }
  if ((a->qinfo == nullptr) != (a->kinfo == nullptr)) return LAP_ERR_ARG;
  p.qinfo = a->qinfo; p.kinfo = a->kinfo; p.lse = (float*)a->lse; p.delta = a->delta;
  p.scale = a->scale;
  p.B = a->B; p.NH = a->NH; p.NKV = a->NKV; p.stop = a->stop_q1_to_k0;
  p.nsplit = 1;
  // grouped-query models: spread the query heads of a kv head over several blocks when scratch is provided
  p.hsplit = 1; p.part = nullptr;
  {
    const int hpk = a->NH / a->NKV;
    const long long need = 2LL * a->B * (a->k_len[0] + a->k_len[1]) * a->NKV * a->HD;
    int hs = a->hsplit > 0 ? a->hsplit : 1;
    if (hs > 1) {
      if (hpk % hs || !a->scratch || a->scratch_floats < need * hs) return LAP_ERR_ARG;
      p.hsplit = hs; p.part = a->scratch;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  switch (a->HD) {
    case 16: return launch_bwd<16>(p, s);
    case 72: return launch_bwd<72>(p, s);
    case 256: return launch_bwd<256>(p, s);
  }
  return LAP_ERR_ARG;
}
