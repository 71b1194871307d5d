// Canary kernels for tools/probes/concurrency_canary.py: do a co-resident kernel's waves keep their registers / their LDS?
#include <hip/hip_runtime.h>
// Every lane keeps 64 VGPRs with known values alive while it spins; mismatches are counted into err[0].
__global__ __launch_bounds__(256) void vgpr_canary(long long ticks, unsigned* err) {
  unsigned v[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) { v[i] = threadIdx.x * 64u + i; asm volatile("" : "+v"(v[i])); }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { asm volatile("s_sleep 1"); }
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) { asm volatile("" : "+v"(v[i])); bad += (v[i] != threadIdx.x * 64u + i); }
  if (bad) atomicAdd(err, bad);
}
// A block fills `bytes` of LDS with a pattern, spins, and checks it; mismatches into err[1], first bad offset into err[2].
__global__ __launch_bounds__(256) void lds_canary(long long ticks, int words, unsigned* err) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0xC0DE0000u + i;
  __syncthreads();
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { asm volatile("s_sleep 1"); }
  __syncthreads();
  unsigned bad = 0; int first = -1;
  for (int i = threadIdx.x; i < words; i += 256) if (lds[i] != 0xC0DE0000u + i) { ++bad; if (first < 0) first = i; }
  if (bad) { atomicAdd(err + 1, bad); atomicMin((int*)err + 2, first); }
}
// Every wave sums lane ids with the ds_bpermute butterfly (what wave_sum / __shfl_xor compile to) over and over; wrong sums into err[3].
__global__ __launch_bounds__(256) void shfl_canary(long long ticks, unsigned* err) {
  const int lane = threadIdx.x & 63;
  long long t0 = wall_clock64();
  unsigned bad = 0, it = 0;
  while (wall_clock64() - t0 < ticks) {
    float v = (float)(lane + (it & 1023));
    float w = (float)(2 * lane + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); w += __shfl_xor(w, o, 64); }
    bad += (v != 2016.0f + 64.0f * (it & 1023)) + (w != 4096.0f);
    ++it;
  }
  if (bad) atomicAdd(err + 3, bad);
}
// The row statistics of layernorm_bwd in isolation: s1 = sum(dy * g), s2 = sum(dy * g * x) per row, one wave per row, D = 1152.
// mode 0: butterfly via __shfl_xor; mode 1: per-lane partials written out (no cross-lane step at all).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__global__ __launch_bounds__(256) void rowstat_kernel(const __bf16* x, const __bf16* dy, const float* gamma, float* out, int rows, int mode) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, D = 1152;
  float gm[3][8];
  for (int p = 0; p < 3; ++p) { const int c = (lane + 64 * p) * 8; for (int e = 0; e < 8; ++e) gm[p][e] = c < D ? gamma[c + e] : 0.f; }
  for (int row = blockIdx.x * 16 + w; row < min(rows, blockIdx.x * 16 + 16); row += 4) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(x + (long long)row * D + c);
        bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(dy + (long long)row * D + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float gv = (float)b[e] * gm[p][e]; s1 += gv; s2 += gv * (float)a[e]; }
      }
    }
    if (mode == 1) { out[((long long)row * 64 + lane) * 2] = s1; out[((long long)row * 64 + lane) * 2 + 1] = s2; continue; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) { out[row * 2] = s1; out[row * 2 + 1] = s2; }
  }
}
extern "C" int rowstat_launch(const void* x, const void* dy, const void* gamma, void* out, int rows, int mode, int lds, void* stream) {
  hipLaunchKernelGGL(rowstat_kernel, dim3((rows + 15) / 16), dim3(256), lds, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)dy, (const float*)gamma, (float*)out, rows, mode);
  return (int)hipGetLastError();
}
extern "C" int canary_launch(int kind, int blocks, int us, int lds_bytes, void* err, void* stream) {
  if (kind == 2) hipLaunchKernelGGL(shfl_canary, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)us * 100, (unsigned*)err);
  else if (kind == 0) hipLaunchKernelGGL(vgpr_canary, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)us * 100, (unsigned*)err);
  else {
    (void)hipFuncSetAttribute((const void*)lds_canary, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_canary, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)us * 100, lds_bytes / 4, (unsigned*)err);
  }
  return (int)hipGetLastError();
}
