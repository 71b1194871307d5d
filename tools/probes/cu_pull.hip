// Probe (round 4): what bounds a batch-1 serving stage — the bytes ONE CU can pull per microsecond, by access pattern and by where
// the bytes live; and what a hand-off / barrier costs when it stays inside one XCD.
//
//   part 1  pull rate: 256 blocks x 512 threads (one per CU), every thread issues NLD 16-byte loads, waits, repeats.
//           patterns: FRAG   = the MFMA-fragment-direct pattern of serve_skinny_body.hpp (lane (i, g): row i, 64-byte pieces of a row
//                              per wave instruction: 16 rows x 64 B)
//                     LINEAR = lane l reads 16 B at l * 16: 1 KiB contiguous per wave instruction
//                     DMA    = LINEAR through buffer_load ... lds (no register round trip)
//           sources:  hot    = every block re-reads the same 128 KiB (L2 hits in every XCD)
//                     own    = block b re-reads its own 64 KiB region (L2 hits, distinct lines per CU)
//                     sc1    = `own` with device-scope loads (what the persistent chain's activations use)
//                     hbm    = every pass a fresh region of a 4 GiB buffer
//   part 2  XCD-local barrier (32 blocks of one XCD, group = HW_REG_XCC_ID) with a same-XCD hand-off: producer plain stores,
//           consumer sc1 (L1-bypassing) loads; counters as agent-scope atomics vs workgroup-scope (L2-local) atomics.
//           Reports us per round and visibility errors.
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_pull cu_pull.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T) __attribute__((address_space(3))) T*

enum { FRAG = 0, LINEAR = 1, DMA = 2 };

template <int PAT, int AUX, int NLD>
__global__ __launch_bounds__(512) void pull(const char* base, long long block_stride, long long pass_stride, int wrap, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    const char* p = base + (long long)blockIdx.x * block_stride + (long long)(it % wrap) * pass_stride;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (unsigned)(NLD * 8192), 0x00020000);
    if (PAT == DMA) {
#pragma unroll
      for (int s = 0; s < NLD; ++s)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_PTR(void))(smem + (w * NLD + s) * 1024), 16, (unsigned)(((w * NLD + s) * 64 + lane) * 16), 0, 0, AUX);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc ^= *reinterpret_cast<unsigned*>(smem + ((tid * 36 + it) & 8191) * 4);
    } else {
      u32x4 v[NLD];
#pragma unroll
      for (int s = 0; s < NLD; ++s) {
        // FRAG: 16 rows of NLD * 512 bytes; wave w owns a run of NLD * 64 bytes of every row, lane (i, g) 16 B of row i per step
        const unsigned off = PAT == FRAG ? (unsigned)(i * (NLD * 512) + w * (NLD * 64) + s * 64 + g * 16) : (unsigned)(((w * NLD + s) * 64 + lane) * 16);
        v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
      }
#pragma unroll
      for (int s = 0; s < NLD; ++s) acc ^= v[s][0] ^ v[s][1] ^ v[s][2] ^ v[s][3];
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// ---- part 2
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

__global__ void census(int* xcc_of, int* count) {
  if (threadIdx.x == 0) {
    const int x = xcc_id();
    xcc_of[blockIdx.x] = x;
    atomicAdd(count + x, 1);
  }
}

// MODE 0: agent-scope counters (as serve_chain.hpp does, but per XCD only: no master);  MODE 1: workgroup-scope (L2-local) atomic
// add + sc1 poll;  MODE 2: as 1 with the poll as a workgroup-scope atomic load... (L1 may serve it: expected to hang -> bounded)
template <int MODE>
__device__ __forceinline__ void xcd_barrier(unsigned* ctr, unsigned target, int* timeouts) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    long spins = 0;
    for (;;) {
      if (__hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) break;     // somebody gave up: no more waiting
      unsigned v;
      if (MODE == 0) v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)ctr, 0, 4u, 0x00020000);
        v = __builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 16);     // sc1: bypass L1, L2-served
      }
      if (v >= target) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 18)) { atomicAdd(timeouts, 1); break; }
    }
  }
  __syncthreads();
}

// every block of an XCD writes `payload` x 16 B (plain stores, or sc1 when SC1ST), barrier inside the XCD, reads the slot of the
// XCD's next block with sc1 loads
template <int MODE, bool SC1ST>
__global__ __launch_bounds__(512) void local_probe(unsigned* ctrs, unsigned* slots, const int* rank_in_xcc, int* errs, int* timeouts, int iters, int payload, int per_xcc) {
  const int bid = blockIdx.x, nb = gridDim.x;
  const int x = xcc_id();
  __shared__ int s_rank;
  if (threadIdx.x == 0) s_rank = (int)atomicAdd(ctrs + 64 * (16 + x), 1u);     // arrival order inside the XCD = local rank
  __syncthreads();
  const int rank = s_rank;
  (void)rank_in_xcc;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, (unsigned)(2u * nb * 512 * 16), 0x00020000);
  int bad = 0;
  // all blocks of the XCD have their rank (first barrier doubles as that)
  xcd_barrier<MODE>(ctrs + 64 * x, (unsigned)per_xcc, timeouts);
  for (int r = 1; r <= iters; ++r) {
    const unsigned base = (unsigned)((r & 1) * nb * 512 * 16);
    if ((int)threadIdx.x < payload) {
      const u32x4 v = {(unsigned)r, (unsigned)(x * 64 + rank), threadIdx.x, (unsigned)r};
      const unsigned off = base + (unsigned)(((x * per_xcc + rank) * 512 + threadIdx.x) * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, SC1ST ? 16 : 0);
    }
    xcd_barrier<MODE>(ctrs + 64 * x, (unsigned)(r + 1) * per_xcc, timeouts);
    if ((int)threadIdx.x < payload) {
      const int other = (rank + 7) % per_xcc;
      const unsigned off = base + (unsigned)(((x * per_xcc + other) * 512 + threadIdx.x) * 16);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
      if (v[0] != (unsigned)r || v[1] != (unsigned)(x * 64 + other) || v[3] != (unsigned)r) ++bad;
    }
  }
  if (bad) atomicAdd(errs, bad);
  (void)bid;
}

template <int PAT, int AUX, int NLD>
static void run_pull(const char* name, const char* src, const char* buf, long long bs, long long ps, int wrap, unsigned* sink, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int lds = PAT == DMA ? NLD * 8192 : 0;
  if (lds > 65536) hipFuncSetAttribute((const void*)pull<PAT, AUX, NLD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((pull<PAT, AUX, NLD>), dim3(256), dim3(512), lds, 0, buf, bs, ps, wrap, 4, sink);     // warm
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((pull<PAT, AUX, NLD>), dim3(256), dim3(512), lds, 0, buf, bs, ps, wrap, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)NLD * 8192 * iters;
  printf("%-7s %-4s %3d KiB per pass: %7.3f us per pass, %6.1f GB/s per CU, %5.2f TB/s chip\n", name, src, NLD * 8, ms * 1e3 / iters, bytes / (ms * 1e-3) / 1e9,
         bytes * 256 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  char* buf; unsigned* sink;
  const long long BIG = 4ll << 30;
  hipMalloc(&buf, BIG); hipMalloc(&sink, 4);
  hipMemset(buf, 1, BIG);
  hipDeviceSynchronize();
#define ROW(PAT, NAME, NLD)                                                                        \
  run_pull<PAT, 0, NLD>(NAME, "hot", buf, 0, 0, 1, sink, iters);                                   \
  run_pull<PAT, 0, NLD>(NAME, "own", buf, (long long)NLD * 8192, 0, 1, sink, iters);               \
  run_pull<PAT, 16, NLD>(NAME, "sc1", buf, (long long)NLD * 8192, 0, 1, sink, iters);              \
  run_pull<PAT, 0, NLD>(NAME, "hbm", buf, (long long)NLD * 8192, 256ll * NLD * 8192, (int)(BIG / (256ll * NLD * 8192)), sink, iters);
  ROW(FRAG, "frag", 8)
  ROW(LINEAR, "linear", 8)
  ROW(DMA, "dma", 8)
  ROW(FRAG, "frag", 16)
  ROW(LINEAR, "linear", 16)
  ROW(DMA, "dma", 16)
  ROW(LINEAR, "linear", 32)
  run_pull<LINEAR, 2, 16>("lin-nt", "hbm", buf, 16ll * 8192, 256ll * 16 * 8192, (int)(BIG / (256ll * 16 * 8192)), sink, iters);
  run_pull<DMA, 2, 16>("dma-nt", "hbm", buf, 16ll * 8192, 256ll * 16 * 8192, (int)(BIG / (256ll * 16 * 8192)), sink, iters);

  // ---- part 2
  const int nb = 256;
  int *xcc_of, *count, *errs, *timeouts; unsigned *ctrs, *slots;
  hipMalloc(&xcc_of, nb * 4); hipMalloc(&count, 64); hipMalloc(&errs, 4); hipMalloc(&timeouts, 4);
  hipMalloc(&ctrs, 64 * 4 * 64); hipMalloc(&slots, 2ull * nb * 512 * 16);
  hipMemset(count, 0, 64);
  hipLaunchKernelGGL(census, dim3(nb), dim3(512), 0, 0, xcc_of, count);
  std::vector<int> hx(nb), hc(16);
  hipMemcpy(hx.data(), xcc_of, nb * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), count, 64, hipMemcpyDeviceToHost);
  printf("blocks per XCC (256-block grid of 512 threads):");
  for (int j = 0; j < 8; ++j) printf(" %d", hc[j]);
  int same = 0;
  for (int b = 0; b < nb; ++b) same += hx[b] == b % 8;
  printf("   | blocks with xcc == id %% 8: %d of %d\n", same, nb);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int rounds = 2000;
  for (int mode = 0; mode < 2; ++mode) for (int sc1st = 0; sc1st < 2; ++sc1st) for (int payload : {64, 512}) {
    hipMemset(ctrs, 0, 64 * 4 * 64); hipMemset(slots, 0, 2ull * nb * 512 * 16); hipMemset(errs, 0, 4); hipMemset(timeouts, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    if (mode == 0 && !sc1st) hipLaunchKernelGGL((local_probe<0, false>), dim3(nb), dim3(512), 0, 0, ctrs, slots, xcc_of, errs, timeouts, rounds, payload, 32);
    if (mode == 0 && sc1st) hipLaunchKernelGGL((local_probe<0, true>), dim3(nb), dim3(512), 0, 0, ctrs, slots, xcc_of, errs, timeouts, rounds, payload, 32);
    if (mode == 1 && !sc1st) hipLaunchKernelGGL((local_probe<1, false>), dim3(nb), dim3(512), 0, 0, ctrs, slots, xcc_of, errs, timeouts, rounds, payload, 32);
    if (mode == 1 && sc1st) hipLaunchKernelGGL((local_probe<1, true>), dim3(nb), dim3(512), 0, 0, ctrs, slots, xcc_of, errs, timeouts, rounds, payload, 32);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int e, t; hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost); hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost);
    printf("XCD-local barrier, %s counters, %s stores + sc1 loads, payload %3d x 16 B: %.3f us per round, visibility errors %d, timeouts %d\n",
           mode == 0 ? "agent-scope    " : "workgroup-scope", sc1st ? "sc1  " : "plain", payload, ms * 1e3 / rounds, e, t);
  }
  return 0;
}
