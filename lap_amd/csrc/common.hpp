// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of lap_amd.
//
// Conventions used by every kernel in this directory:
//   * wave = 64 lanes; blocks are multiples of 64 threads.
//   * bf16 storage is raw `__bf16`; all arithmetic is fp32 and rounds to bf16
//     with round-to-nearest-even exactly once where the reference casts.
//   * MFMA = v_mfma_f32_16x16x32_bf16.  A dot product is order independent, so
//     any assignment of the 32 k indices of a k-step to (lane group, element)
//     slots is valid as long as both operands use the same one.  The GEMM uses
//     the natural one (group g holds k = 8g..8g+7); the attention kernels use
//     {4g..4g+3} u {16+4g..16+4g+3}, which makes the C/D layout of S^T = K.Q^T
//     directly reusable as the P operand of P.V (no LDS round trip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T*

#define LAP_OK 0
#define LAP_ERR_ARG 1001   // bad argument (shape / alignment / unsupported combination)

#define LAP_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t _e = hipGetLastError();                       \
    if (_e != hipSuccess) return (int)_e;                    \
  } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // RNE
// A bf16 rounding that the reference performs between two f32 operations (bf16 products, bf16 adds ...).  The value
// is passed through an empty asm so that the compiler cannot fold the fptrunc/fpext pair into the surrounding
// arithmetic: left to itself it contracted `x + bf16(y * g)` into fma(y, g, x) in one kernel and not in another,
// which breaks bit-identity between the fused serving kernels and the generic layer path.
__device__ __forceinline__ float round_bf16(float x) {
  float r = (float)(bf16)x;
  asm("" : "+v"(r));
  return r;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
  // jax.nn.gelu(approximate=True): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
// tanh-GELU in its sigmoid form, x / (1 + exp2(x (c1 x^2 + c0))), c0 = -2 sqrt(2/pi) log2(e), c1 = 0.044715 c0, through v_exp_f32 /
// v_rcp_f32: the arithmetic of the training step's lap_gemm_asm_nt_bias_gelu epilogue (tools/gen_gemm_asm.py gelu_packed), instruction
// for instruction.  jax.nn.gelu(approximate=True) to ~2 ulp of f32 before the bf16 rounding; ~9 VALU operations instead of tanhf's ~40
// (a 64 x 192 tile of one wave per SIMD spends 3 us in tanhf).  Used by the serving prefill's epilogues (serve_panel.hip, gemm.hip's GeGLU tile).
__device__ __forceinline__ float gelu_exp2_f(float x) {
  float p = __builtin_fmaf(x * x, __builtin_bit_cast(float, 0xbdd2d3e8u), __builtin_bit_cast(float, 0xc0135761u));
  p = x * p;
  asm("" : "+v"(p));     // (keeps x * p a product of its own: no re-association into the fma)
  const float d = 1.0f + __builtin_amdgcn_exp2f(p);
  return x * __builtin_amdgcn_rcpf(d);
}

__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  float t = tanhf(u);
  float du = k0 * (1.0f + 3.0f * k1 * x * x);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// RoPE (gemma.py:548-564) building blocks shared by every kernel that rotates, written with explicitly rounded
// operations so that all of them produce the same bits regardless of how the compiler would contract a*b - c*d.
__device__ __forceinline__ void rope_sincos(float pos, int i, int HD, float& sn, float& cs) {
  const float ts = powf(10000.0f, (2.0f / (float)HD) * (float)i);   // timescale of frequency index i
  sincosf(__fdiv_rn(pos, ts), &sn, &cs);
}
__device__ __forceinline__ void rope_rotate(float x1, float x2, float sn, float cs, float& r1, float& r2) {
  r1 = __fsub_rn(__fmul_rn(x1, cs), __fmul_rn(x2, sn));
  r2 = __fadd_rn(__fmul_rn(x2, cs), __fmul_rn(x1, sn));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of NW waves; `red` is an LDS array of >= NW floats.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ---- LDS fragment reads (GEMM family: contiguous k assignment) ---------------
// MFMA 16x16x32 operand: lane (i = lane & 15, g = lane >> 4) holds k = 8g .. 8g+7
// of the 32-deep k-step for operand row i.
//
// K-contiguous tile: 128 bytes per row (64 bf16 of k), row = m or n index.
// 16-byte chunks of a row are XOR-swizzled with ((row >> 1) & 7): every 16-lane
// service group of ds_read_b128 then covers all 64 banks exactly once.
__device__ __forceinline__ unsigned kc_tile_off(int row, int chunk16) {
  return (unsigned)(row * 128 + ((chunk16 ^ ((row >> 1) & 7)) << 4));
}
// Fragment for rows [row0, row0+16) and k-step kk (0/1) of a 64-deep tile.
__device__ __forceinline__ bf16x8 kc_frag(const char* tile, int row0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  return *reinterpret_cast<const bf16x8*>(tile + kc_tile_off(row0 + i, kk * 4 + g));
}

// M-contiguous tile: row = k index (64 rows), MW bf16 of m per row (MW*2 bytes).
// 32-byte blocks of a row are XOR-swizzled with mc_swz(k) so that the 8 k-rows
// {8g'+r, r<4, g' in {0,1}} read together by ds_read_b64_tr_b16 are conflict free.
__device__ __forceinline__ int mc_swz(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
template <int MW>
__device__ __forceinline__ unsigned mc_tile_off(int krow, int chunk16) {
  return (unsigned)(krow * (MW * 2) + ((chunk16 ^ (mc_swz(krow) << 1)) << 4));
}
__device__ __forceinline__ bf16x4 ds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_PTR(bf16x4))(p));
}
// The same read issued from inline asm.  The compiler does not model it as an LDS access, which is the point: for
// the builtin (an instruction without memory operand) it conservatively drains EVERY outstanding LDS-DMA prefetch
// (`s_waitcnt vmcnt(0)`) before the first transposing read of a loop body, serialising prefetch and compute.  The
// price: no automatic lgkmcnt wait either -- every result MUST pass through one of the lds_wait*() fences below
// (which tie the registers to the wait) before it is used.  LDS operations retire in order, so a counted wait that
// leaves the N youngest reads in flight is safe regardless of what else the compiler has outstanding.
template <int OFFSET>
__device__ __forceinline__ bf16x4 ds_read_tr_raw(unsigned lds_addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFFSET));
  return r;
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)(LDS_PTR(const char))(p); }
template <int LEFT>
__device__ __forceinline__ void lds_wait4(bf16x4 (&r)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(LEFT));
}
template <int LEFT>
__device__ __forceinline__ void lds_wait8(bf16x4 (&r)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
               : "n"(LEFT));
}
// Zero-instruction fence: pins `r` behind every earlier asm volatile (the wait) for the compiler's scheduler.
template <typename T> __device__ __forceinline__ void lds_tie(T& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// Fragment for m columns [m0, m0+16) and k-step kk of a 64-deep tile.  A
// 16-lane group reads a [4 k][16 m] block (lane i supplies row i/4, 8 bytes at
// column 4*(i%4)) and receives column i, 4 consecutive k.
template <int MW>
__device__ __forceinline__ bf16x8 mc_frag(const char* tile, int m0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int kr = kk * 32 + 8 * g + (i >> 2);
  const int mcol = m0 + (i & 3) * 4;
  const int c16 = mcol >> 3, half = (mcol >> 2) & 1;
  bf16x4 lo = ds_read_tr(tile + mc_tile_off<MW>(kr, c16) + (half << 3));
  bf16x4 hi = ds_read_tr(tile + mc_tile_off<MW>(kr + 4, c16) + (half << 3));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// kc_frag through a raw ds_read_b128 (valid after lds_wait_all() + lds_tie()).
__device__ __forceinline__ bf16x8 kc_frag_raw(const char* tile, int row0, int kk, int lane) {
  const int i = lane & 15, g = lane >> 4;
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_addr_of(tile) + kc_tile_off(row0 + i, kk * 4 + g)));
  return r;
}
// mc_frag through raw reads (see ds_read_tr_raw): results are valid only after lds_wait_all() + lds_tie().
template <int MW>
__device__ __forceinline__ void mc_frag_raw(const char* tile, int m0, int kk, int lane, bf16x4 (&r)[2]) {
  const int i = lane & 15, g = lane >> 4;
  const int kr = kk * 32 + 8 * g + (i >> 2);
  const int mcol = m0 + (i & 3) * 4;
  const int c16 = mcol >> 3, half = (mcol >> 2) & 1;
  const unsigned a = lds_addr_of(tile) + mc_tile_off<MW>(kr, c16) + (half << 3);
  r[0] = ds_read_tr_raw<0>(a);
  r[1] = ds_read_tr_raw<4 * MW * 2>(a);   // k-row + 4: same swizzle
}

// Fragment-packed matrix layout of the serving chain (serve_skinny_body.hpp "PK"): a [16 rows x 32 k] tile is 1 KiB, lane
// 16 g + i of a wave holds row i, k = 8 g .. 8 g + 7; the tiles of one row group are consecutive in k.
__device__ __forceinline__ long long pk_off(int r, int c, int K) {     // element offset of (row r, column c) of a packed [.., K] matrix
  return ((long long)(r >> 4) * (K >> 5) + (c >> 5)) * 512 + ((((c & 31) >> 3) << 4) + (r & 15)) * 8 + (c & 7);
}

// XCD-aware block id remap (bijective for any grid size): consecutive logical
// ids land on the same XCD (blocks are dispatched round-robin over 8 XCDs), so
// neighbouring tiles share operand panels in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
