// What a weight gradient FUSED into the data-gradient pass would pay to get its accumulators out (round 4).
//
// A pointwise weight gradient d W[256][256] accumulated inside the per-layer data-gradient launch lives in the registers of
// all 256 workgroups at once (each sees 1/256 .. 1/128 of the rows), so EVERY launch ends by flushing
// 256 workgroups x 128 KB (channel-split pairs) = 32 MB of fp32 partial sums — against 157 MB of tensor traffic of the whole
// launch — where the batched launch at the end of backward (one layer segment per workgroup) flushes ~1 MB per layer.
// This probe times the three ways to get 32 MB of register-resident partials into one 256 x 256 result:
//   A  plain stores of per-workgroup slabs + a reduction kernel            (write 32 MB + read 32 MB)
//   B  device-scope f32 atomics into the result                             (8.4 M atomics)
//   C  workgroup-scope f32 atomics into one slab per XCD (blockIdx % 8) + an 8-slab reduction; checked for exactness
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/flush_probe.hip -o tools/flush_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)

constexpr int NWG = 256, NT = 512, PER = 64;          // 64 floats per lane = 128 KB per workgroup (half of d W)
constexpr int HALF = 256 * 128;                        // floats per workgroup

__device__ __forceinline__ float val(int wg, int i) { return (float)((wg * 7 + i) % 13) - 6.f; }

template <int MODE>
__global__ __launch_bounds__(NT) void flush_kernel(float* __restrict__ out) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int pair = (wg >> 4) * 8 + (wg & 7), h = (wg >> 3) & 1;      // workgroups b and b + 8 share an XCD (b % 8)
  float acc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) acc[j] = val(wg, j * NT + tid);
  if (MODE == 0) {
    float* slab = out + (size_t)wg * HALF;
#pragma unroll
    for (int j = 0; j < PER; ++j) slab[j * NT + tid] = acc[j];
  } else if (MODE == 1) {
    float* dst = out + (size_t)h * HALF;
#pragma unroll
    for (int j = 0; j < PER; ++j) __hip_atomic_fetch_add(&dst[j * NT + tid], acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    float* dst = out + ((size_t)(wg & 7) * 2 + h) * HALF;
#pragma unroll
    for (int j = 0; j < PER; ++j) __hip_atomic_fetch_add(&dst[j * NT + tid], acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  (void)pair;
}
// out[h][i] = sum over the slabs of half h
__global__ void reduce_kernel(const float* __restrict__ slabs, int nslab_per_half, int mode, float* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * HALF; i += gridDim.x * blockDim.x) {
    const int h = i / HALF, e = i % HALF;
    float s = 0.f;
    if (mode == 0) {
      for (int wg = 0; wg < NWG; ++wg)
        if (((wg >> 3) & 1) == h) s += slabs[(size_t)wg * HALF + e];
    } else {
      for (int x = 0; x < 8; ++x) s += slabs[((size_t)x * 2 + h) * HALF + e];
    }
    out[i] = s;
  }
}

int main() {
  float *slabs, *res;
  CK(hipMalloc(&slabs, (size_t)NWG * HALF * 4));
  CK(hipMalloc(&res, (size_t)2 * HALF * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> want(2 * HALF, 0.f), got(2 * HALF);
  for (int wg = 0; wg < NWG; ++wg)
    for (int i = 0; i < HALF; ++i) want[(size_t)((wg >> 3) & 1) * HALF + i] += (float)((wg * 7 + i) % 13) - 6.f;
  const char* names[3] = {"A plain slabs + reduce", "B device-scope atomics", "C workgroup-scope atomics per XCD + reduce"};
  for (int mode = 0; mode < 3; ++mode) {
    float ms_sum = 0.f;
    const int iters = 20;
    for (int it = 0; it < iters + 2; ++it) {
      if (mode != 0) CK(hipMemsetAsync(mode == 1 ? res : slabs, 0, (size_t)(mode == 1 ? 2 : 16) * HALF * 4, 0));
      CK(hipDeviceSynchronize());
      hipEventRecord(e0, 0);
      if (mode == 0) { hipLaunchKernelGGL(flush_kernel<0>, dim3(NWG), dim3(NT), 0, 0, slabs); hipLaunchKernelGGL(reduce_kernel, dim3(256), dim3(256), 0, 0, slabs, 128, 0, res); }
      if (mode == 1) hipLaunchKernelGGL(flush_kernel<1>, dim3(NWG), dim3(NT), 0, 0, res);
      if (mode == 2) { hipLaunchKernelGGL(flush_kernel<2>, dim3(NWG), dim3(NT), 0, 0, slabs); hipLaunchKernelGGL(reduce_kernel, dim3(256), dim3(256), 0, 0, slabs, 8, 1, res); }
      hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it >= 2) ms_sum += ms;
    }
    CK(hipMemcpy(got.data(), res, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    printf("%-46s %7.2f us per layer (x 51 sub-block layers = %5.2f ms per step), %zu of %zu sums wrong\n", names[mode],
           ms_sum / iters * 1e3f, ms_sum / iters * 51, bad, got.size());
  }
  return 0;
}
