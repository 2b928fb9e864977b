// How expensive are the per-workgroup reduction atomics?  N workgroups x 256 threads each add K floats
// (coalesced: thread t -> address rep*K + i*256 + t) into 8-way replicated accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(float* acc, int K) {
  const int rep = blockIdx.x % 8;
  for (int i = threadIdx.x; i < K; i += blockDim.x)
    __hip_atomic_fetch_add(&acc[rep * K + i], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
int main() {
  float* acc; hipMalloc(&acc, 8 * 4096 * 4); hipMemset(acc, 0, 8 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int K : {512, 1536}) for (int N : {256, 600, 1200, 4800}) {
    k<<<N, 256>>>(acc, K); hipDeviceSynchronize();
    hipEventRecord(e0); for (int it = 0; it < 10; ++it) k<<<N, 256>>>(acc, K); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("K=%4d floats x N=%4d WGs: %7.2f us/launch  (%.1f M atomics, %.2f G atomics/s)\n", K, N, ms * 100, K * (double)N / 1e6, K * (double)N / (ms * 1e-4) / 1e9);
  }
  return 0;
}
