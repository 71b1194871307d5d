// A co-resident load for GEMM scheduling experiments (tools/probes/coresident.py): `blocks` workgroups that each hold
// `lds` bytes of LDS and spin for `us` microseconds.  With lds > 29 KiB a spinning block keeps a 128 KiB GEMM block off
// its CU (the collective-kernel-owns-the-CU case); with lds = 0 it shares the CU and competes for issue slots.
#include <hip/hip_runtime.h>
__global__ void spin_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  long long t0 = wall_clock64();
  int x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) { x = x * 1664525 + 1013904223; }
  if (x == 0x7fffffff) sink[0] = x + lds[0];
}
extern "C" int spin_launch(int blocks, int threads, int lds, int us, void* sink, void* stream) {
  hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (long long)us * 100, (int*)sink);
  return (int)hipGetLastError();
}
