// What a pointwise weight gradient FUSED into the per-layer data-gradient launch would pay to get its accumulators out.
// Round 5 rewrite: the round-4 probe timed a latency-chain reduction kernel and called it the flush.
// This one measures the two things the design would really do:
//
//   (i)  in-launch cost: every workgroup of the layer's launch ends by storing its register-resident partial d W as ONE
//        coalesced slab (16-byte stores), measured as the MARGINAL time of a streaming launch shaped like dgrad_dw_v6
//        (256 persistent workgroups, 3 tensor reads + 2 tensor writes of 76800 x 256 bf16, rotating over many buffers so the
//        data is cold) with and without the flush, and stand-alone;
//   (ii) ONE deferred reduction per gradient bucket: all 51 layers' slabs summed by a single long launch written like the
//        round-4 slab_reduce_kernel (16-byte loads, 8 partials in flight per thread, >= 4 waves per SIMD).
//
// Variants of the slab: f32 256 KB per workgroup (the whole 256 x 256 accumulator: the 1-wave-per-SIMD AGPR design),
// f32 128 KB (channel-split pairs), bf16 128 KB (the whole accumulator packed to bf16 before the store).
// build: hipcc --offload-arch=gfx950 -O3 tools/flush_probe.hip -o tools/flush_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s\n", (int)e, #x); return 1; } } while (0)

constexpr int NWG = 256, NT = 512, ROWS = 76800, C = 256, LAYERS = 51;
constexpr size_t TENSOR = (size_t)ROWS * C;            // bf16 elements per tensor (39.3 MB)

__device__ __forceinline__ uint32_t pk(float a, float b) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// MODE 0: stream only.  1: + f32 slab of PER floats per lane.  2: + bf16-packed slab of PER floats per lane.
// STREAM = false: the flush alone.
template <int MODE, int PER, bool STREAM>
__global__ __launch_bounds__(NT, 2) void layer_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ c,
                                                      uint4* __restrict__ o1, uint4* __restrict__ o2, float* __restrict__ slabs, float seedv) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  float acc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) acc[j] = seedv * (float)(j + 1);
  if (STREAM) {
    // 3 reads + 2 writes, 16 bytes per lane and stream, one 32-row tile (16 KB per stream) per step, the next tile prefetched
    const size_t nvec = TENSOR / 8;
    const size_t per_wg = nvec / NWG;
    const size_t base = (size_t)wg * per_wg;
    uint4 pa[2], pb[2], pc[2];
    auto ld = [&](size_t i) {
#pragma unroll
      for (int q = 0; q < 2; ++q) { pa[q] = a[base + i + q * NT]; pb[q] = b[base + i + q * NT]; pc[q] = c[base + i + q * NT]; }
    };
    ld(tid);
    for (size_t i = tid; i < per_wg; i += 2 * NT) {
      uint4 xa[2], xb[2], xc[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) { xa[q] = pa[q]; xb[q] = pb[q]; xc[q] = pc[q]; }
      if (i + 2 * NT < per_wg) ld(i + 2 * NT);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 r1, r2;
        r1.x = xa[q].x ^ xb[q].x; r1.y = xa[q].y ^ xc[q].y; r1.z = xa[q].z + xb[q].z; r1.w = xa[q].w + xc[q].w;
        r2.x = xb[q].x + xc[q].x; r2.y = xb[q].y ^ xa[q].y; r2.z = xc[q].z ^ xa[q].z; r2.w = xb[q].w + xc[q].w;
        o1[base + i + q * NT] = r1;
        o2[base + i + q * NT] = r2;
        // keep the accumulators live and data-dependent (as MFMA results would be)
        acc[(q * 7) % PER] += __uint_as_float((r1.x & 0x007fffffu) | 0x3f800000u) - 1.f;
      }
    }
  }
  if (MODE == 1) {
    float4* slab = reinterpret_cast<float4*>(slabs + (size_t)wg * PER * NT);
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) slab[j * NT + tid] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
  } else if (MODE == 2) {
    uint4* slab = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(slabs) + (size_t)wg * PER * NT);
#pragma unroll
    for (int j = 0; j < PER / 8; ++j) {
      uint4 v;
      v.x = pk(acc[8 * j], acc[8 * j + 1]); v.y = pk(acc[8 * j + 2], acc[8 * j + 3]);
      v.z = pk(acc[8 * j + 4], acc[8 * j + 5]); v.w = pk(acc[8 * j + 6], acc[8 * j + 7]);
      slab[j * NT + tid] = v;
    }
  } else if (!STREAM || acc[0] == 12345.678f) {
    slabs[tid] = acc[0];
  }
}

// deferred reduction, ONE launch for all layers: out[layer][e] = sum over the layer's `parts` slabs.  A thread owns 4 (f32)
// or 8 (bf16) consecutive elements (one 16-byte load per partial) and keeps 8 partial loads in flight.
template <bool BF16>
__global__ __launch_bounds__(256) void reduce_all_kernel(const void* __restrict__ slabs, int parts, size_t slab_elems, size_t layer_stride_elems,
                                                          float* __restrict__ out, int out_elems) {
  constexpr int EPT = BF16 ? 8 : 4;
  const int layer = blockIdx.y;
  const int e0 = (blockIdx.x * 256 + threadIdx.x) * EPT;
  if (e0 >= out_elems) return;
  float s[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) s[i] = 0.f;
  const char* base = reinterpret_cast<const char*>(slabs) + ((size_t)layer * layer_stride_elems + e0) * (BF16 ? 2 : 4);
  const size_t pstride = slab_elems * (BF16 ? 2 : 4);
  for (int p0 = 0; p0 < parts; p0 += 8) {
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (p0 + u < parts) ? *reinterpret_cast<const uint4*>(base + (size_t)(p0 + u) * pstride) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (BF16) {
        s[0] += __uint_as_float(v[u].x << 16); s[1] += __uint_as_float(v[u].x & 0xffff0000u);
        s[2] += __uint_as_float(v[u].y << 16); s[3] += __uint_as_float(v[u].y & 0xffff0000u);
        s[4] += __uint_as_float(v[u].z << 16); s[5] += __uint_as_float(v[u].z & 0xffff0000u);
        s[6] += __uint_as_float(v[u].w << 16); s[7] += __uint_as_float(v[u].w & 0xffff0000u);
      } else {
        s[0] += __uint_as_float(v[u].x); s[1] += __uint_as_float(v[u].y); s[2] += __uint_as_float(v[u].z); s[3] += __uint_as_float(v[u].w);
      }
    }
  }
  float* o = out + (size_t)layer * out_elems + e0;
#pragma unroll
  for (int i = 0; i < EPT; i += 4) *reinterpret_cast<float4*>(o + i) = make_float4(s[i], s[i + 1], s[i + 2], s[i + 3]);
}

template <int MODE, int PER, bool STREAM>
static float time_layers(uint16_t** bufs, int nbuf, float* slabs, size_t slab_floats_per_layer, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = layer_kernel<MODE, PER, STREAM>;
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int l = 0; l < reps; ++l) {
      // rotate over the buffers like the step rotates over its workspace (cold data), one slab region per layer
      const uint16_t* a = bufs[(5 * l) % nbuf]; const uint16_t* b = bufs[(5 * l + 1) % nbuf]; const uint16_t* c = bufs[(5 * l + 2) % nbuf];
      uint16_t* o1 = bufs[(5 * l + 3) % nbuf]; uint16_t* o2 = bufs[(5 * l + 4) % nbuf];
      hipLaunchKernelGGL(kern, dim3(NWG), dim3(NT), 0, 0, (const uint4*)a, (const uint4*)b, (const uint4*)c, (uint4*)o1, (uint4*)o2,
                         slabs + (size_t)(l % LAYERS) * slab_floats_per_layer, 1.f + l);
    }
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f / reps;      // us per launch
}

int main() {
  const int nbuf = 60;                                       // 60 x 39.3 MB = 2.4 GB of rotating tensors
  std::vector<uint16_t*> bufs(nbuf);
  for (auto& p : bufs) { CK(hipMalloc(&p, TENSOR * 2)); CK(hipMemset(p, 0x3c, TENSOR * 2)); }
  const size_t slab_floats_per_layer = (size_t)NWG * 128 * NT;    // 64 MB per layer (the f32 256 KB variant)
  float* slabs; CK(hipMalloc(&slabs, slab_floats_per_layer * 4 * LAYERS));   // 3.3 GB
  CK(hipMemset(slabs, 0, slab_floats_per_layer * 4 * LAYERS));
  float* out; CK(hipMalloc(&out, (size_t)LAYERS * 65536 * 4));
  const int reps = LAYERS;
  const float t_stream = time_layers<0, 128, true>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps);
  printf("streaming launch alone (3 reads + 2 writes of 76800 x 256 bf16, 256 persistent workgroups, cold): %.2f us  (%.2f TB/s)\n",
         t_stream, 5.0 * TENSOR * 2 / t_stream * 1e-6);
  struct V { const char* name; float with, alone; double mb; };
  V v[3] = {
      {"f32, 256 KB per workgroup (64 MB per layer)", time_layers<1, 128, true>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps),
       time_layers<1, 128, false>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps), 64.0 * 1.048576},
      {"f32, 128 KB per workgroup (32 MB: channel-split pairs)", time_layers<1, 64, true>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps),
       time_layers<1, 64, false>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps), 32.0 * 1.048576},
      {"bf16, 128 KB per workgroup (32 MB: packed accumulator)", time_layers<2, 128, true>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps),
       time_layers<2, 128, false>(bufs.data(), nbuf, slabs, slab_floats_per_layer, reps), 32.0 * 1.048576},
  };
  // (ii) ONE reduction over all 51 layers
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float red[3];
  for (int k = 0; k < 3; ++k) {
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
      // make the slabs cold again: stream over the rotating tensors
      time_layers<0, 128, true>(bufs.data(), nbuf, slabs, slab_floats_per_layer, 12);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      if (k == 0)        // 256 slabs of 65536 f32 per layer
        hipLaunchKernelGGL(reduce_all_kernel<false>, dim3(65536 / 4 / 256, LAYERS), dim3(256), 0, 0, slabs, 256, (size_t)65536, slab_floats_per_layer, out, 65536);
      else if (k == 1)   // 256 slabs of 32768 f32 per layer: two halves of 128 slabs each
        hipLaunchKernelGGL(reduce_all_kernel<false>, dim3(32768 / 4 / 256, 2 * LAYERS), dim3(256), 0, 0, slabs, 128, (size_t)32768, slab_floats_per_layer / 2, out, 32768);
      else               // 256 slabs of 65536 bf16 per layer
        hipLaunchKernelGGL(reduce_all_kernel<true>, dim3(65536 / 8 / 256, LAYERS), dim3(256), 0, 0, slabs, 256, (size_t)65536, slab_floats_per_layer * 2, out, 65536);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    red[k] = best;
  }
  CK(hipGetLastError());
  printf("%-58s %10s %10s %12s %14s %12s\n", "slab variant", "in-launch", "alone", "x51 in-launch", "1 reduction", "flush/step");
  for (int k = 0; k < 3; ++k) {
    const float marg = v[k].with - t_stream;
    printf("%-58s %7.2f us %7.2f us %9.3f ms %8.3f ms (%.2f TB/s) %8.3f ms\n", v[k].name, marg, v[k].alone, marg * LAYERS * 1e-3f, red[k],
           v[k].mb * 1e6 * LAYERS / (red[k] * 1e-3) * 1e-12, marg * LAYERS * 1e-3f + red[k]);
  }
  printf("(in-launch = time of the streaming launch with the flush minus without; alone = a launch that only stores the slabs;\n"
         " 1 reduction = ONE launch summing the 256 slabs of all %d layers, slabs cold)\n", LAYERS);
  return 0;
}
