#include <hip/hip_runtime.h>

// ---- a copy of csrc/norm.hip's layernorm_bwd_kernel that also writes, per row and lane, the partial sums before the butterfly
// and the reduced values after it (tools/probes/concurrency_lnb_debug.py)
#include "../../lap_amd/csrc/common.hpp"
namespace {
constexpr int NWAVE = 4;
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
  for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
  bf16x8 t;
  for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
  *reinterpret_cast<bf16x8*>(p) = t;
}
template <int NCH>
__global__ __launch_bounds__(256) void lnb_debug_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const bf16* __restrict__ dy, bf16* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int rows, int D, int G, int accum_dx, float* dbg) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [NWAVE][2*D]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * G, r1 = min(rows, r0 + G);
  float gm[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[p][e] = 0.f; db[p][e] = 0.f; gm[p][e] = (c < D) ? gamma[c + e] : 0.f; }
  }
  // (No hand-rolled prefetch of the next row here: with it, the kernel's row statistics came out different in a few rows per
  // launch whenever blocks of the 128 x 128 GEMM shared its CUs — tools/probes/concurrency_stress6.py; inputs, registers and
  // LDS of co-resident kernels were checked intact, tools/probes/concurrency_canary.py.  Occupancy hides the latency.)
  for (int row = r0 + w; row < r1; row += NWAVE) {
    bf16x8 cx[NCH], cdy[NCH];
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        cx[p] = *reinterpret_cast<const bf16x8*>(x + (long long)row * D + c);
        cdy[p] = *reinterpret_cast<const bf16x8*>(dy + (long long)row * D + c);
      }
    }
    const float mu = mean[row], r = rstd[row];
    float xh[NCH][8], gv[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float xv[8], dyv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { xv[e] = (float)cx[p][e]; dyv[e] = (float)cdy[p][e]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[p][e] = (xv[e] - mu) * r;
          gv[p][e] = dyv[e] * gm[p][e];
          s1 += gv[p][e];
          s2 += gv[p][e] * xh[p][e];
          dg[p][e] += dyv[e] * xh[p][e];
          db[p][e] += dyv[e];
        }
      }
    }
    dbg[((long long)row * 64 + lane) * 4 + 0] = s1; dbg[((long long)row * 64 + lane) * 4 + 1] = s2;
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    dbg[((long long)row * 64 + lane) * 4 + 2] = s1; dbg[((long long)row * 64 + lane) * 4 + 3] = s2;
    bf16* dxr = dx + (long long)row * D;
#pragma unroll
    for (int p = 0; p < NCH; ++p) {
      const int c = (lane + 64 * p) * 8;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = r * (gv[p][e] - s1 - xh[p][e] * s2);
        if (accum_dx) {
          float old[8];
          load8(dxr + c, old);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += old[e];
        }
        store8(dxr + c, o);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NCH; ++p) {
    const int c = (lane + 64 * p) * 8;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[w * 2 * D + c + e] = dg[p][e];
        red[w * 2 * D + D + c + e] = db[p][e];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) t += red[i * 2 * D + c];
    if (c < D) atomicAdd(dgamma + c, t);
    else atomicAdd(dbeta + (c - D), t);
  }
}

}  // namespace
extern "C" int lnb_debug_launch(const void* x, const float* gamma, const float* mean, const float* rstd, const void* dy, void* dx,
                                float* dgamma, float* dbeta, int rows, int D, float* dbg, void* stream) {
  const int G = 16;
  const size_t shm = (size_t)NWAVE * 2 * D * sizeof(float);
  hipLaunchKernelGGL(lnb_debug_kernel<3>, dim3((rows + G - 1) / G), dim3(256), shm, (hipStream_t)stream, (const bf16*)x, gamma, mean, rstd,
                     (const bf16*)dy, (bf16*)dx, dgamma, dbeta, rows, D, G, 0, dbg);
  return (int)hipGetLastError();
}
