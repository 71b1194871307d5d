// Probe: latency of a software grid barrier on MI355X (256 co-resident blocks x 512 threads), with a visibility check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 24)) break;   // never hang the box
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default for __atomic_thread_fence? use the hip builtin below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// two-level: G group counters (256 B apart), the last arriver of a group bumps the master; everybody polls the master
template <int SCOPE_GROUP>
__device__ __forceinline__ void grid_barrier2(unsigned* ctrs, int group, int gsize, int G, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    unsigned old;
    if (SCOPE_GROUP == 0) old = __hip_atomic_fetch_add(ctrs + 64 * (1 + group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else old = __hip_atomic_fetch_add(ctrs + 64 * (1 + group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((old + 1) % gsize == 0) __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * G) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 24)) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void probe2(unsigned* ctrs, int* slots, int* errs, int iters, int payload, int G) {
  const int bid = blockIdx.x, nb = gridDim.x;
  int group = bid % G, gsize = nb / G;
  if (MODE == 2) {   // group = the XCD the block really runs on
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    group = x & 15;
  }
  int bad = 0;
  for (int r = 1; r <= iters; ++r) {
    int* s = slots + (r & 1) * nb * 64;
    if (threadIdx.x < payload) s[bid * 64 + threadIdx.x] = r;
    grid_barrier2<MODE == 2>(ctrs, group, gsize, G, (unsigned)r);
    const int other = (bid + 37) % nb;
    if (threadIdx.x < payload && s[other * 64 + threadIdx.x] != r) ++bad;
  }
  if (bad) atomicAdd(errs, bad);
}

__global__ void xcc_probe(int* out) {
  unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)x;
}

__global__ __launch_bounds__(512) void probe(unsigned* ctr, int* slots, int* errs, int iters, int payload) {
  const int bid = blockIdx.x, nb = gridDim.x;
  int bad = 0;
  for (int r = 1; r <= iters; ++r) {
    int* s = slots + (r & 1) * nb * 64;
    if (threadIdx.x < payload) s[bid * 64 + threadIdx.x] = r;
    grid_barrier(ctr, (unsigned)r * nb);
    const int other = (bid + 37) % nb;
    if (threadIdx.x < payload && s[other * 64 + threadIdx.x] != r) ++bad;
  }
  if (bad) atomicAdd(errs, bad);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned* ctr; int *slots, *errs;
  hipMalloc(&ctr, 4); hipMalloc(&slots, 2 * nb * 64 * 4); hipMalloc(&errs, 4);
  for (int payload : {1, 64}) {
    hipMemset(ctr, 0, 4); hipMemset(slots, 0, 2 * nb * 64 * 4); hipMemset(errs, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, ctr, slots, errs, 10, payload);   // warm
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 0, 0, ctr, slots, errs, iters, payload);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int e; hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
    printf("blocks %d payload %d: %.3f us per barrier round, visibility errors %d\n", nb, payload, ms * 1e3 / iters, e);
  }
  {
    int* xo; hipMalloc(&xo, nb * 4); hipLaunchKernelGGL(xcc_probe, dim3(nb), dim3(512), 0, 0, xo);
    int h[1024]; hipMemcpy(h, xo, nb * 4, hipMemcpyDeviceToHost);
    int mism = 0; for (int i = 0; i < nb; ++i) mism += ((h[i] & 15) != i % 8);
    printf("xcc ids of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %x", h[i]); printf("  (bid%%8 mismatches: %d)\n", mism);
  }
  unsigned* ctrs; hipMalloc(&ctrs, 64 * 4 * 65);
  for (int mode : {0, 2}) for (int G : {8, 16, 32}) {
    if (mode == 2 && G != 8) continue;
    hipMemset(ctrs, 0, 64 * 4 * 65); hipMemset(slots, 0, 2 * nb * 64 * 4); hipMemset(errs, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    if (mode == 0) hipLaunchKernelGGL(probe2<0>, dim3(nb), dim3(512), 0, 0, ctrs, slots, errs, iters, 64, G);
    else hipLaunchKernelGGL(probe2<2>, dim3(nb), dim3(512), 0, 0, ctrs, slots, errs, iters, 64, G);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int e; hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
    printf("two-level mode %d G %d blocks %d: %.3f us per barrier round, visibility errors %d\n", mode, G, nb, ms * 1e3 / iters, e);
  }
  return 0;
}
