// Machine calibration for the TitaNet hot path (tuning tool, not part of the product):
//   * streaming kernels over 39.3 MB tensors (the rows x 256 bf16 activation size at B = 256, T = 300): read-only, copy,
//     2 reads + 1 write, 3 reads + 1 write; "hot" (same buffers every launch: Infinity-Cache resident) vs "cold" (rotating over
//     2.5 GB of buffers);
//   * the same reads through LDS-DMA (global_load_lds_dwordx4) into an LDS ring;
//   * the cost of an in-kernel grid barrier (one workgroup per CU).
// build: hipcc --offload-arch=gfx950 -O3 -o stream_bench tools/stream_bench.hip ; run: ./stream_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", (int)e, hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gbl_void;

template <int U>
__global__ __launch_bounds__(256) void read_k(const uint4* __restrict__ a, size_t n, uint32_t* sink) {
  uint32_t s = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * stride < n) ? a[i + u * stride] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (s == 0x12345678u) *sink = s;
}
template <int U>
__global__ __launch_bounds__(256) void write_k(uint4* __restrict__ c, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) c[i + u * stride] = make_uint4((uint32_t)i, u, 3, 4);
  }
}
template <int U, int NIN>
__global__ __launch_bounds__(256) void rw_k(const uint4* __restrict__ a, const uint4* __restrict__ b, const uint4* __restrict__ d,
                                            uint4* __restrict__ c, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    uint4 va[U], vb[U], vd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      const bool ok = j < n;
      va[u] = ok ? a[j] : make_uint4(0, 0, 0, 0);
      if (NIN > 1) vb[u] = ok ? b[j] : make_uint4(0, 0, 0, 0);
      if (NIN > 2) vd[u] = ok ? d[j] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      uint4 r = va[u];
      if (NIN > 1) { r.x += vb[u].x; r.y ^= vb[u].y; r.z += vb[u].z; r.w ^= vb[u].w; }
      if (NIN > 2) { r.x ^= vd[u].x; r.y += vd[u].y; r.z ^= vd[u].z; r.w += vd[u].w; }
      if (j < n) c[j] = r;
    }
  }
}


// each workgroup (256 threads) walks `blocks` of blk16 uint4: contiguous range per workgroup (mode 0) or interleaved (mode 1)
__global__ __launch_bounds__(256) void read_blocks_k(const uint4* __restrict__ a, size_t n, int blk16, int mode, uint32_t* sink) {
  const size_t nblk = n / blk16;
  const size_t per = (nblk + gridDim.x - 1) / gridDim.x;
  uint32_t s = 0;
  for (size_t k = 0; k < per; ++k) {
    const size_t b = mode ? k * gridDim.x + blockIdx.x : blockIdx.x * per + k;
    if (b >= nblk) break;
    const uint4* src = a + b * blk16;
    for (int i = threadIdx.x; i < blk16; i += 1024) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i + u * 256 < blk16) ? src[i + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
  }
  if (s == 0x12345678u) *sink = s;
}

// ---- LDS-DMA streaming read: persistent workgroups of 256 threads, ring of NS slots x 16 KB; every wave issues its 4
// 1-KB pieces of slot s + DEPTH, waits (counted vmcnt) for slot s, workgroup barrier, folds the slot from LDS.
template <int NS, int DEPTH>
__global__ __launch_bounds__(256) void dma_read_k(const uint4* __restrict__ a, size_t n_slots, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t s = 0;
  // slots of this workgroup: blockIdx.x, + gridDim.x, ...
  const size_t first = blockIdx.x, stride = gridDim.x;
  const size_t mine = first < n_slots ? (n_slots - first + stride - 1) / stride : 0;
  auto issue = [&](size_t k) {          // k-th slot of this workgroup -> ring slot k % NS
    const uint4* src = a + (first + k * stride) * 1024 + wave * 256 + lane;
    // LDS byte address of the wave's 4 KB part of the slot (dynamic LDS starts at 0: no static __shared__ in this kernel)
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((k % NS) * 16384 + wave * 4096));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned keep;
      // hidden from hipcc's waitcnt bookkeeping (cdna_hip_programming.md 5.7): M0 written in the statement that reads it
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + p * 64), "s"(dst + p * 1024) : "memory");
    }
  };
  for (int k = 0; k < DEPTH && (size_t)k < mine; ++k) issue(k);
  for (size_t k = 0; k < mine; ++k) {
    if (k + DEPTH < mine) {
      issue(k + DEPTH);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * DEPTH) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const uint4* slot = reinterpret_cast<const uint4*>(ring + (k % NS) * 16384);
#pragma unroll
    for (int p = 0; p < 4; ++p) { const uint4 v = slot[p * 256 + tid]; s += v.x ^ v.y ^ v.z ^ v.w; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (s == 0x12345678u) *sink = s;
}

// ---- grid barrier (one 512-thread workgroup per CU): monotonic counter, relaxed polling, bounded spin
__global__ __launch_bounds__(512) void barrier_k(unsigned* counter, int iters, unsigned* fail, float* data) {
  const unsigned nwg = gridDim.x;
  for (int it = 1; it <= iters; ++it) {
    if (data) atomicAdd(&data[(blockIdx.x % 8) * 512 + threadIdx.x], 1.0f);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = nwg * (unsigned)it;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 2000000) { *fail = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

static float time_launches(int n, const std::vector<hipEvent_t>& ev, void (*fn)(int, void*), void* ctx) {
  for (int i = 0; i < 3; ++i) fn(i, ctx);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(ev[0], 0));
  for (int i = 0; i < n; ++i) fn(i + 3, ctx);
  CK(hipEventRecord(ev[1], 0));
  CK(hipEventSynchronize(ev[1]));
  float ms;
  CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
  return ms * 1e3f / n;   // us per launch
}

struct Ctx { std::vector<uint4*> bufs; size_t n; int nset; int grid; int kind; uint32_t* sink; };

template <int U>
static void run_kind(int it, void* p) {
  Ctx& c = *(Ctx*)p;
  const int s = (it % c.nset) * 4;
  const uint4 *a = c.bufs[s], *b = c.bufs[s + 1], *d = c.bufs[s + 2];
  uint4* o = c.bufs[s + 3];
  switch (c.kind) {
    case 0: hipLaunchKernelGGL(read_k<U>, dim3(c.grid), dim3(256), 0, 0, a, c.n, c.sink); break;
    case 1: hipLaunchKernelGGL(write_k<U>, dim3(c.grid), dim3(256), 0, 0, o, c.n); break;
    case 2: hipLaunchKernelGGL((rw_k<U, 1>), dim3(c.grid), dim3(256), 0, 0, a, b, d, o, c.n); break;
    case 3: hipLaunchKernelGGL((rw_k<U, 2>), dim3(c.grid), dim3(256), 0, 0, a, b, d, o, c.n); break;
    case 4: hipLaunchKernelGGL((rw_k<U, 3>), dim3(c.grid), dim3(256), 0, 0, a, b, d, o, c.n); break;
  }
}
template <int NS, int DEPTH>
static void run_dma(int it, void* p) {
  Ctx& c = *(Ctx*)p;
  const int s = (it % c.nset) * 4;
  hipLaunchKernelGGL((dma_read_k<NS, DEPTH>), dim3(c.grid), dim3(256), NS * 16384, 0, c.bufs[s], c.n / 1024, c.sink);
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)256 * 300 * 256 * 2;     // 39.3 MB
  const size_t n = bytes / 16;
  const int NSET = 16;                                    // 16 x 4 x 39.3 MB = 2.5 GB
  std::vector<uint4*> bufs(NSET * 4);
  for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  uint32_t* sink; CK(hipMalloc(&sink, 64));
  std::vector<hipEvent_t> ev(2);
  CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  const char* names[] = {"read 1t", "write 1t", "copy 2t", "2 reads + 1 write 3t", "3 reads + 1 write 4t"};
  const int passes[] = {1, 1, 2, 3, 4};
  for (int kind = 0; kind < 5; ++kind)
    for (int hot = 0; hot < 2; ++hot)
      for (int grid : {256, 1024, 2048, 4096}) {
        Ctx c{bufs, n, hot ? 1 : NSET, grid, kind, sink};
        const float u4 = time_launches(32, ev, run_kind<4>, &c);
        const float u8 = time_launches(32, ev, run_kind<8>, &c);
        printf("%-22s %-4s grid %5d x256thr: U=4 %7.2f us (%5.2f TB/s)   U=8 %7.2f us (%5.2f TB/s)\n", names[kind], hot ? "hot" : "cold", grid,
               u4, passes[kind] * bytes / u4 / 1e6, u8, passes[kind] * bytes / u8 / 1e6);
      }
  if (argc > 1 && atoi(argv[1]) == 1) {
    // long launches: ONE kernel over 32 x 39.3 MB = 1.26 GB per stream (what a 1-2 ms kernel of the step can reach, without
    // the ramp-up / drain of a 39 MB launch)
    const size_t big = bytes * 32, nb = big / 16;
    uint4 *A, *B2, *D2, *O;
    CK(hipMalloc(&A, big)); CK(hipMalloc(&B2, big)); CK(hipMalloc(&D2, big)); CK(hipMalloc(&O, big));
    CK(hipMemset(A, 1, big)); CK(hipMemset(B2, 1, big)); CK(hipMemset(D2, 1, big)); CK(hipMemset(O, 1, big));
    for (int grid : {256, 512, 1024, 4096, 16384}) {
      float t[5];
      for (int kind = 0; kind < 5; ++kind) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(ev[0], 0));
          switch (kind) {
            case 0: hipLaunchKernelGGL(read_k<8>, dim3(grid), dim3(256), 0, 0, A, nb, sink); break;
            case 1: hipLaunchKernelGGL(write_k<8>, dim3(grid), dim3(256), 0, 0, O, nb); break;
            case 2: hipLaunchKernelGGL((rw_k<4, 1>), dim3(grid), dim3(256), 0, 0, A, B2, D2, O, nb); break;
            case 3: hipLaunchKernelGGL((rw_k<4, 2>), dim3(grid), dim3(256), 0, 0, A, B2, D2, O, nb); break;
            case 4: hipLaunchKernelGGL((rw_k<4, 3>), dim3(grid), dim3(256), 0, 0, A, B2, D2, O, nb); break;
          }
          CK(hipEventRecord(ev[1], 0));
          CK(hipEventSynchronize(ev[1]));
          float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
          if (rep > 0 && ms < best) best = ms;
        }
        t[kind] = passes[kind] * (float)big / best / 1e9f;
      }
      printf("long launch (1.26 GB/stream) grid %5d: read %5.2f  write %5.2f  copy %5.2f  2r1w %5.2f  3r1w %5.2f TB/s\n", grid, t[0], t[1], t[2], t[3], t[4]);
    }
    for (int blk16 : {1024, 4096}) for (int mode = 0; mode < 2; ++mode) for (int grid : {256, 512}) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(ev[0], 0));
        hipLaunchKernelGGL(read_blocks_k, dim3(grid), dim3(256), 0, 0, A, nb, blk16, mode, sink);
        CK(hipEventRecord(ev[1], 0));
        CK(hipEventSynchronize(ev[1]));
        float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("long read, %3d KB blocks, %s, grid %d: %5.2f TB/s\n", blk16 / 64, mode ? "interleaved over workgroups" : "one contiguous range per workgroup", grid, (float)big / best / 1e9f);
    }
    return 0;
  }
  for (int hot = 0; hot < 2; ++hot)
    for (int grid : {256, 512, 1024}) {
      Ctx c{bufs, n, hot ? 1 : NSET, grid, 0, sink};
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_read_k<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_read_k<8, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_read_k<4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384));
      const float a = time_launches(32, ev, run_dma<4, 2>, &c);
      const float b = time_launches(32, ev, run_dma<4, 3>, &c);
      const float d = time_launches(32, ev, run_dma<8, 6>, &c);
      printf("LDS-DMA read 1t        %-4s grid %5d x256thr: ring4/depth2 %7.2f us (%5.2f TB/s)  ring4/depth3 %7.2f (%5.2f)  ring8/depth6 %7.2f (%5.2f)\n",
             hot ? "hot" : "cold", grid, a, bytes / a / 1e6, b, bytes / b / 1e6, d, bytes / d / 1e6);
    }
  {
    unsigned* counter; unsigned* fail; float* data;
    CK(hipMalloc(&counter, 64)); CK(hipMalloc(&fail, 64)); CK(hipMalloc(&data, 8 * 512 * 4));
    const int grid = prop.multiProcessorCount;
    for (int with_data = 0; with_data < 2; ++with_data)
      for (int iters : {1, 101}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipMemset(counter, 0, 64)); CK(hipMemset(fail, 0, 64)); CK(hipMemset(data, 0, 8 * 512 * 4));
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(ev[0], 0));
          hipLaunchKernelGGL(barrier_k, dim3(grid), dim3(512), 0, 0, counter, iters, fail, with_data ? data : nullptr);
          CK(hipEventRecord(ev[1], 0));
          CK(hipEventSynchronize(ev[1]));
          float ms; CK(hipEventElapsedTime(&ms, ev[0], ev[1]));
          if (ms * 1e3f < best) best = ms * 1e3f;
        }
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("grid barrier (%d WGs x 512 thr, %s): %d barriers in %.2f us (fail=%u)\n", grid, with_data ? "+512 atomics/WG" : "bare", iters, best, f);
      }
  }
  return 0;
}
