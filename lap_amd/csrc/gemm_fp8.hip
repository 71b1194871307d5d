// fp8 (OCP e4m3) MFMA GEMM for gfx950 — BASELINE.json config 5 ("LAP-3B fp8 weights/activations, CDNA4 fp8 MFMA").
//
//   C[M,N] = epilogue( alpha / (s_a * s_b) * sum_k A8[m,k] * B8[n,k] )        A8 [M][K], B8 [N][K]: fp8 bytes, K contiguous
//
// Only the NT form exists: the forward uses (activations [M][in], weights [out][in]); the data gradient uses
// (output gradients [M][out], the TRANSPOSED fp8 copy of the weights [in][out]) — the quantisation pass writes both weight
// layouts, so no transposing LDS reads are needed for 8-bit data.  Weight gradients stay bf16 (gemm.hip).
//
// Kernel: the 256 x 256 block tile of the bf16 kernels with a 128-BYTE k-tile — the same LDS image (rows of 128 bytes,
// 16-byte chunks XOR-swizzled on the source address), the same LDS-DMA pieces, the same logical tile order and epilogues
// (gemm_common.hpp).  What changes is the arithmetic: one v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair and k-tile
// (K = 128 per instruction, twice the FLOPs of the two bf16 16x16x32 MFMAs it replaces, at the same matrix-pipe time), with
// the block scales fixed to 2^0 — scaling is per TENSOR and applied once in the epilogue.  Lane (i, g) of an operand holds
// the 32 bytes [32 g, 32 g + 32) of row i: which k inside the instruction a byte is matched with does not matter, because
// both operands are loaded the same way and a dot product is order independent.
// 8 waves (2 x 4), 128 x 64 per wave, two LDS stages, one barrier per k-tile, fragment reads of A's second half re-issued
// under the MFMAs of the first (a second full register set does not fit: 96 operand + 128 accumulator VGPRs).
//
// Quantisation (per-tensor "current scaling"): amax over the tensor, s = 448 / amax, q = e4m3(clamp(x * s)).
#include "gemm_common.hpp"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ f32x4 mfma_f8(i32x8 a, i32x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int OFF>
__device__ __forceinline__ i32x4 ds_read_b128_i(unsigned addr) {   // raw: valid after lds_wait_all() + lds_tie()
  i32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

template <bool OUT_F32>
__global__ __launch_bounds__(512) void gemm_f8_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, BKB = 128, A_BYTES = BM * BKB, STAGE = 2 * A_BYTES;   // BKB: bytes = fp8 elements per k-tile row
  constexpr int WGM = 2, WGN = 4, NW = 8, WTM = 128, WTN = 64, FM = 8, FN = 4, PC = 4;   // PC: 1 KiB DMA pieces per operand per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  int tm, tn;
  tile_coords<4>(p, p.tile_base + xcd_remap(blockIdx.x, gridDim.x), tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const char* A8 = reinterpret_cast<const char*>(p.A);
  const char* B8 = reinterpret_cast<const char*>(p.B);
  auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A8, 0, (int)min((long long)p.M * p.lda, 0x7fffffffLL), 0x00020000);
  auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B8, 0, (int)min((long long)p.N * p.ldb, 0x7fffffffLL), 0x00020000);
  unsigned offA[PC], offB[PC];
#pragma unroll
  for (int j = 0; j < PC; ++j) {
    const int ci = (w * PC + j) * 64 + lane;                   // 16-byte chunk of the tile image: 8 chunks per 128-byte row
    const int row = ci >> 3, c = (ci & 7) ^ ((row >> 1) & 7);
    offA[j] = (m0 + row < p.M) ? (unsigned)((long long)(m0 + row) * p.lda + c * 16) : OOB;
    offB[j] = (n0 + row < p.N) ? (unsigned)((long long)(n0 + row) * p.ldb + c * 16) : OOB;
  }
  const int nkt = p.K / BKB;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const unsigned va = (offA[j] != OOB && kt < nkt) ? offA[j] + (unsigned)kt * BKB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_PTR(void))(base + (w * PC + j) * 1024), 16, va, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PC; ++j) {
      const unsigned vb = (offB[j] != OOB && kt < nkt) ? offB[j] + (unsigned)kt * BKB : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_PTR(void))(base + A_BYTES + (w * PC + j) * 1024), 16, vb, 0, 0, 0);
    }
  };
  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // lane (i, g): chunks 2g and 2g + 1 of row i; fragments of one operand are 16 rows = 2048 bytes apart and share the swizzle
  const int li = lane & 15, lg = lane >> 4;
  auto frag_addr = [&](const char* tile, int row0, int half) {
    return lds_addr_of(tile) + kc_tile_off(row0 + li, 2 * lg + half);
  };
  i32x4 ra[FM][2], rb[FN][2];
  auto read_b = [&](const char* tB) {
    const unsigned b0 = frag_addr(tB, wn * WTN, 0), b1 = frag_addr(tB, wn * WTN, 1);
    rb[0][0] = ds_read_b128_i<0>(b0); rb[0][1] = ds_read_b128_i<0>(b1);
    rb[1][0] = ds_read_b128_i<2048>(b0); rb[1][1] = ds_read_b128_i<2048>(b1);
    rb[2][0] = ds_read_b128_i<4096>(b0); rb[2][1] = ds_read_b128_i<4096>(b1);
    rb[3][0] = ds_read_b128_i<6144>(b0); rb[3][1] = ds_read_b128_i<6144>(b1);
  };
  auto read_a = [&](const char* tA, int hi) {   // fragments 4 hi .. 4 hi + 3
    const unsigned a0 = frag_addr(tA, wm * WTM + hi * 64, 0), a1 = frag_addr(tA, wm * WTM + hi * 64, 1);
    ra[4 * hi + 0][0] = ds_read_b128_i<0>(a0); ra[4 * hi + 0][1] = ds_read_b128_i<0>(a1);
    ra[4 * hi + 1][0] = ds_read_b128_i<2048>(a0); ra[4 * hi + 1][1] = ds_read_b128_i<2048>(a1);
    ra[4 * hi + 2][0] = ds_read_b128_i<4096>(a0); ra[4 * hi + 2][1] = ds_read_b128_i<4096>(a1);
    ra[4 * hi + 3][0] = ds_read_b128_i<6144>(a0); ra[4 * hi + 3][1] = ds_read_b128_i<6144>(a1);
  };
  auto join = [](const i32x4 (&r)[2]) { return __builtin_shufflevector(r[0], r[1], 0, 1, 2, 3, 4, 5, 6, 7); };
#define F8_FENCE() __builtin_amdgcn_sched_barrier(0)

  stage(0, 0);
  stage(1, 1);
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const char* tA = smem + cur * STAGE;
    const char* tB = tA + A_BYTES;
    wait_vmcnt<2 * PC>();                     // my pieces of tile kt landed (tile kt + 1 may still fly)
    F8_FENCE();
    __builtin_amdgcn_s_barrier();             // ... and everybody else's; all reads of tile kt - 1 are complete
    F8_FENCE();
    read_b(tB);
    read_a(tA, 0);
    read_a(tA, 1);
    lds_wait_all();
#pragma unroll
    for (int j = 0; j < FN; ++j) { lds_tie(rb[j][0]); lds_tie(rb[j][1]); }
#pragma unroll
    for (int i = 0; i < FM; ++i) { lds_tie(ra[i][0]); lds_tie(ra[i][1]); }
    F8_FENCE();
    __builtin_amdgcn_s_barrier();             // every wave holds tile kt in registers: its buffer may be refilled
    F8_FENCE();
    stage(cur, kt + 2);
    F8_FENCE();
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = mfma_f8(join(rb[j]), join(ra[i]), acc[i][j]);   // D[row = n][col = m], as in gemm.hip
    F8_FENCE();
  }
  wait_vmcnt<0>();
#undef F8_FENCE
  // per-tensor scales: the operands were multiplied by s_a / s_b before rounding
  GemmParams q = p;
  if (p.qscale_a) q.alpha = p.alpha / (p.qscale_a[0] * p.qscale_b[0]);
  if (q.epi_lds) { staged_epilogue<NW, WTM, WTN, OUT_F32>(q, smem, acc, wm, wn, m0, n0, tid, lane); return; }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * WTM + i * 16 + li;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * WTN + j * 16 + 4 * lg;
      if (n >= p.N) continue;
      store_tile4<OUT_F32>(q, m, n, acc[i][j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ quantisation
__global__ __launch_bounds__(256) void amax_bf16_kernel(const bf16* __restrict__ x, long long rows, int cols, long long ld, float* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  const int c8 = cols / 8;
  const long long n8 = rows * c8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const long long r = i / c8;
    const int c = (int)(i % c8) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ld + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
  }
  m = wave_max(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) red[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned*>(amax), __float_as_uint(m));     // non-negative floats order like their bit patterns
  }
}

__device__ __forceinline__ float f8_scale(float amax) { return 448.0f / fmaxf(amax, 1e-12f); }   // e4m3 max normal = 448

__device__ __forceinline__ unsigned pack4_f8(float a, float b, float c, float d) {
  const float L = 448.0f;
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -L), L), fminf(fmaxf(b, -L), L), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -L), L), fminf(fmaxf(d, -L), L), w, true);
  return (unsigned)w;
}

// out8[r][c] = e4m3(x[r][c] * s), s = 448 / amax (written to *scale by block 0).  8 elements per thread.
__global__ __launch_bounds__(256) void quant_f8_kernel(const bf16* __restrict__ x, long long rows, int cols, long long ld,
                                                       const float* __restrict__ amax, unsigned char* __restrict__ out, long long ldo,
                                                       float* __restrict__ scale) {
  const float s = f8_scale(amax[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = s;
  const int c8 = cols / 8;
  const long long n8 = rows * c8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const long long r = i / c8;
    const int c = (int)(i % c8) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * ld + c);
    u32x2 o;
    o[0] = pack4_f8((float)v[0] * s, (float)v[1] * s, (float)v[2] * s, (float)v[3] * s);
    o[1] = pack4_f8((float)v[4] * s, (float)v[5] * s, (float)v[6] * s, (float)v[7] * s);
    *reinterpret_cast<u32x2*>(out + r * ldo + c) = o;
  }
}

// Weights: out8[r][c] as above AND the transposed copy out8t[c][r] (64 x 64 tiles through LDS).
__global__ __launch_bounds__(256) void quant_f8_t_kernel(const bf16* __restrict__ x, int rows, int cols, const float* __restrict__ amax,
                                                         unsigned char* __restrict__ out, unsigned char* __restrict__ outT,
                                                         float* __restrict__ scale) {
  __shared__ unsigned char tile[64][68];
  const float s = f8_scale(amax[0]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale[0] = s;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  {
    const int r = threadIdx.x >> 2, cq = (threadIdx.x & 3) * 16;     // 16 elements per thread
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    if (r0 + r < rows) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (c0 + cq + 8 * h < cols) {
          const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + (long long)(r0 + r) * cols + c0 + cq + 8 * h);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[8 * h + e] = (float)t[e] * s;
        }
      }
    }
    unsigned wv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[q] = pack4_f8(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    if (r0 + r < rows) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (c0 + cq + 8 * h < cols)
          *reinterpret_cast<u32x2*>(out + (long long)(r0 + r) * cols + c0 + cq + 8 * h) = u32x2{wv[2 * h], wv[2 * h + 1]};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<unsigned*>(&tile[r][cq + 4 * q]) = wv[q];
  }
  __syncthreads();
  {
    const int c = threadIdx.x >> 2, rq = (threadIdx.x & 3) * 16;     // transposed: row c of outT, 16 consecutive r
    if (c0 + c < cols) {
      unsigned wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned t = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) t |= (unsigned)tile[rq + 4 * q + e][c] << (8 * e);
        wv[q] = t;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (r0 + rq + 8 * h < rows)     // rows % 8 == 0
          *reinterpret_cast<u32x2*>(outT + (long long)(c0 + c) * rows + r0 + rq + 8 * h) = u32x2{wv[2 * h], wv[2 * h + 1]};
    }
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int lap_gemm_fp8(const void* A8, const void* B8, void* C, const void* residual, const float* scale_a,
                            const float* scale_b, int M, int N, int K, int lda, int ldb, int ldc, int ldr, float alpha,
                            int flags, void* stream) {
  if (!A8 || !B8 || !C || !scale_a || !scale_b || M <= 0 || N <= 0 || K <= 0) return LAP_ERR_ARG;
  if ((K & 127) || (N & 3) || (ldc & 3) || (lda & 15) || (ldb & 15) || (residual && (ldr & 3))) return LAP_ERR_ARG;
  if (((uintptr_t)A8 | (uintptr_t)B8 | (uintptr_t)C) & 15) return LAP_ERR_ARG;
  if ((long long)M * lda >= 0x7fffffffLL || (long long)N * ldb >= 0x7fffffffLL) return LAP_ERR_ARG;
  const bool f32 = flags & LAP_GEMM_OUT_F32;
  if (flags & ~(LAP_GEMM_OUT_F32 | LAP_GEMM_ACCUM)) return LAP_ERR_ARG;
  if ((flags & LAP_GEMM_ACCUM) && !f32) return LAP_ERR_ARG;
  GemmParams p = {};
  p.A = (const bf16*)A8; p.B = (const bf16*)B8; p.C = C; p.R = (const bf16*)residual;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.alpha = alpha;
  p.accum = (flags & LAP_GEMM_ACCUM) ? 1 : 0;
  p.ksplit = 1;
  p.qscale_a = scale_a; p.qscale_b = scale_b;
  p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
  p.epi_lds = (!(f32 && p.R) && !(N & 7) && !(ldc & 7)) ? 1 : 0;
  const int LDS = f32 ? 128 * (256 * 4 + 16) : 256 * (256 * 2 + 16);   // >= the two operand stages (128 KiB)
  auto launch = [&](auto kern) -> int {
    static bool done_f32 = false, done_b16 = false;
    bool& done = f32 ? done_f32 : done_b16;
    if (!done) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return (int)e;
      done = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), LDS, S_, p);
    LAP_CHECK_LAUNCH();
    return LAP_OK;
  };
  return f32 ? launch(gemm_f8_kernel<true>) : launch(gemm_f8_kernel<false>);
}

extern "C" int lap_amax_bf16(const void* x, long long rows, int cols, long long ld, float* amax, void* stream) {
  if (!x || !amax || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7)) return LAP_ERR_ARG;
  const long long n8 = rows * (cols / 8);
  const long long blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(amax_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, S_, (const bf16*)x, rows, cols, ld, amax);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_quantize_fp8(const void* x, long long rows, int cols, long long ld, const float* amax, void* out8, long long ldo,
                                float* scale, void* stream) {
  if (!x || !amax || !out8 || !scale || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || (ldo & 7)) return LAP_ERR_ARG;
  const long long n8 = rows * (cols / 8);
  const long long blocks = (n8 + 255) / 256;
  hipLaunchKernelGGL(quant_f8_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, S_, (const bf16*)x, rows, cols, ld, amax,
                     (unsigned char*)out8, ldo, scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}

extern "C" int lap_quantize_fp8_weight(const void* w, int rows, int cols, const float* amax, void* out8, void* out8_t, float* scale,
                                       void* stream) {
  if (!w || !amax || !out8 || !out8_t || !scale || rows <= 0 || cols <= 0 || (cols & 7) || (rows & 7)) return LAP_ERR_ARG;
  hipLaunchKernelGGL(quant_f8_t_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, S_, (const bf16*)w, rows, cols, amax,
                     (unsigned char*)out8, (unsigned char*)out8_t, scale);
  LAP_CHECK_LAUNCH();
  return LAP_OK;
}
