// Does vmcnt count loads and stores IN ORDER on gfx950?  A slow load (cold line, HBM), then a fast store (hot line), then
// s_waitcnt vmcnt(1): if a store could retire ahead of the older load, the wait would pass with the load still in flight and the
// destination register would still hold the sentinel.  Counts such observations over many waves and rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const unsigned* __restrict__ cold, unsigned* __restrict__ hot, unsigned* __restrict__ bad, size_t stride_words, int rounds) {
  const size_t lane_line = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  unsigned n_bad = 0;
  for (int r = 0; r < rounds; ++r) {
    const unsigned* src = cold + (lane_line + (size_t)r * gridDim.x * blockDim.x) * stride_words;   // a line nobody touched before
    unsigned* dst = hot + threadIdx.x;                                                                 // the same hot line every time
    unsigned v = 0xdeadbeefu, seen;
    asm volatile(
        "global_load_dword %0, %2, off\n\t"
        "global_store_dword %3, %4, off\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_mov_b32 %1, %0\n\t"
        "s_waitcnt vmcnt(0)"
        : "+v"(v), "=v"(seen) : "v"(src), "v"(dst), "v"(r) : "memory");
    if (seen == 0xdeadbeefu) ++n_bad;
    if (v != 0x12345678u) n_bad += 1000000;   // (sanity: the load itself must deliver the pattern)
  }
  if (n_bad) atomicAdd(bad, n_bad);
}
int main() {
  const int blocks = 2048, threads = 256, rounds = 64;
  const size_t stride_words = 64;   // 256 B apart: every lane its own line
  const size_t n = (size_t)blocks * threads * rounds * stride_words;
  unsigned *cold, *hot, *bad;
  hipMalloc(&cold, n * 4); hipMalloc(&hot, 4096); hipMalloc(&bad, 4);
  std::vector<unsigned> pat(1 << 20, 0x12345678u);
  for (size_t off = 0; off < n; off += pat.size()) hipMemcpy(cold + off, pat.data(), std::min(pat.size(), n - off) * 4, hipMemcpyHostToDevice);
  hipMemset(bad, 0, 4); hipMemset(hot, 0, 4096);
  // evict: stream a big buffer through the caches
  unsigned* junk; hipMalloc(&junk, 1ull << 30); hipMemset(junk, 1, 1ull << 30); hipDeviceSynchronize();
  probe<<<blocks, threads>>>(cold, hot, bad, stride_words, rounds);
  hipDeviceSynchronize();
  unsigned h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("loads observed incomplete after vmcnt(1) behind a younger store: %u of %zu\n", h, (size_t)blocks * threads * rounds);
  return 0;
}
