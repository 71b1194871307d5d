// Probe: a software grid barrier WITHOUT cache maintenance on MI355X.  The payload moves through device-scope (sc1) buffer stores
// and loads — what the gfx942/gfx950 memory model uses for relaxed agent-scope atomics — so release is "s_waitcnt vmcnt(0)" and
// acquire is nothing: no buffer_wbl2 / buffer_inv.  Two-level arrival (per-XCD counter, last arriver bumps the master).  Reports us
// per barrier round and visibility errors (a block reads what a block of ANOTHER XCD wrote in front of the barrier).
// MODE 0: fences + plain accesses (reference point), MODE 1: sc1 accesses, no fences, MODE 2: sc1 accesses, one flat counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void barrier(unsigned* ctrs, int group, int gsize, int G, unsigned round, int nb) {
  if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    long spins = 0;
    if (MODE == 2) {
      __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * nb) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 22)) break;
      }
    } else {
      const unsigned old = __hip_atomic_fetch_add(ctrs + 64 * (1 + group), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((old + 1) % gsize == 0) __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * G) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 22)) break;
      }
    }
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// payload: every thread below `payload` writes 16 bytes (its block's slot), then reads the slot of a block on another XCD
template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned* ctrs, unsigned* slots, int* errs, int iters, int payload, int G) {
  const int bid = blockIdx.x, nb = gridDim.x;
  const int group = bid % G, gsize = nb / G;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, (unsigned)(2u * nb * 512 * 16), 0x00020000);
  int bad = 0;
  for (int r = 1; r <= iters; ++r) {
    const unsigned base = (unsigned)((r & 1) * nb * 512 * 16);
    if ((int)threadIdx.x < payload) {
      const u32x4 v = {(unsigned)r, (unsigned)bid, threadIdx.x, (unsigned)r};
      const unsigned off = base + (unsigned)((bid * 512 + threadIdx.x) * 16);
      if (MODE == 0) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(slots) + off) = v;
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
    }
    barrier<MODE>(ctrs, group, gsize, G, (unsigned)r, nb);
    if ((int)threadIdx.x < payload) {
      const int other = (bid + 37) % nb;
      const unsigned off = base + (unsigned)((other * 512 + threadIdx.x) * 16);
      u32x4 v;
      if (MODE == 0) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(slots) + off);
      else v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
      if (v[0] != (unsigned)r || v[1] != (unsigned)other || v[3] != (unsigned)r) ++bad;
    }
  }
  if (bad) atomicAdd(errs, bad);
}

// one-directional hand-off: two kernels on two streams; the consumer is launched together with the producer, spins on the producer's
// arrival counter, then reads the producer's payload.  us per (producer, consumer) pair against the same pair in stream order.
template <bool FLAG>
__global__ __launch_bounds__(512) void producer(unsigned* ctrs, unsigned* slots, unsigned round, int G) {
  const int bid = blockIdx.x, nb = gridDim.x;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, (unsigned)(nb * 512u * 16), 0x00020000);
  const u32x4 v = {round, (unsigned)bid, threadIdx.x, round};
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((bid * 512 + threadIdx.x) * 16), 0, FLAG ? 16 : 0);
  if (FLAG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(ctrs + 64 * (1 + bid % G), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((old + 1) % (nb / G) == 0) __hip_atomic_fetch_add(ctrs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <bool FLAG>
__global__ __launch_bounds__(512) void consumer(unsigned* ctrs, unsigned* slots, int* errs, unsigned round, int G) {
  const int bid = blockIdx.x, nb = gridDim.x;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, (unsigned)(nb * 512u * 16), 0x00020000);
  if (FLAG) {
    if (threadIdx.x == 0) {
      long spins = 0;
      while (__hip_atomic_load(ctrs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * G) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 22)) break;
      }
    }
    __syncthreads();
  }
  const int other = (bid + 37) % nb;
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((other * 512 + threadIdx.x) * 16), 0, FLAG ? 16 : 0);
  if (v[0] != round || v[1] != (unsigned)other) atomicAdd(errs, 1);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *ctrs, *slots; int* errs;
  hipMalloc(&ctrs, 64 * 4 * 65); hipMalloc(&slots, 2ull * nb * 512 * 16); hipMalloc(&errs, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode : {0, 1, 2}) for (int payload : {64, 512}) {
    hipMemset(ctrs, 0, 64 * 4 * 65); hipMemset(slots, 0, 2ull * nb * 512 * 16); hipMemset(errs, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(nb), dim3(512), 0, 0, ctrs, slots, errs, iters, payload, 8);
    else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(nb), dim3(512), 0, 0, ctrs, slots, errs, iters, payload, 8);
    else hipLaunchKernelGGL(probe<2>, dim3(nb), dim3(512), 0, 0, ctrs, slots, errs, iters, payload, 8);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int e; hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
    printf("mode %d (%s) payload %3d x 16 B per block: %.3f us per barrier round, visibility errors %d\n", mode,
           mode == 0 ? "fences, plain accesses" : mode == 1 ? "sc1 accesses, two-level" : "sc1 accesses, flat counter", payload, ms * 1e3 / iters, e);
  }
  // hand-off between two kernels
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int pairs = 500;
  for (int flag : {0, 1, 0, 1}) {
    hipMemset(ctrs, 0, 64 * 4 * 65); hipMemset(errs, 0, 4); hipDeviceSynchronize();
    hipEvent_t evs[2]; hipEventCreateWithFlags(&evs[0], hipEventDisableTiming); hipEventCreateWithFlags(&evs[1], hipEventDisableTiming);
    hipEventRecord(a, s0);
    for (int r = 1; r <= pairs; ++r) {
      if (!flag) {
        hipLaunchKernelGGL(producer<false>, dim3(nb), dim3(512), 0, s0, ctrs, slots, (unsigned)r, 8);
        hipLaunchKernelGGL(consumer<false>, dim3(nb), dim3(512), 0, s0, ctrs, slots, errs, (unsigned)r, 8);
      } else {
        // producer r on s0 behind consumer r-1 (s1): the slots are rewritten only after they were read
        hipEventRecord(evs[1], s1); hipStreamWaitEvent(s0, evs[1], 0);
        hipLaunchKernelGGL(producer<true>, dim3(nb), dim3(512), 0, s0, ctrs, slots, (unsigned)r, 8);
        // consumer r on s1: launched as soon as consumer r-1 is through, i.e. together with producer r
        hipLaunchKernelGGL(consumer<true>, dim3(nb), dim3(512), 0, s1, ctrs, slots, errs, (unsigned)r, 8);
      }
    }
    if (flag) { hipEventRecord(evs[0], s1); hipStreamWaitEvent(s0, evs[0], 0); }
    hipEventRecord(b, s0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int e; hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
    printf("hand-off %s: %.3f us per producer + consumer pair, errors %d\n", flag ? "by flag, two streams" : "in stream order      ", ms * 1e3 / pairs, e);
  }
  return 0;
}
