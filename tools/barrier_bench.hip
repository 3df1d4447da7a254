// micro-benchmark: cost of a grid-wide barrier inside one persistent kernel against the kernel boundary it would replace
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#ifndef SLEEP
#define SLEEP 0
#endif
struct Bar { unsigned count; unsigned gen; unsigned fail; };

__device__ __forceinline__ bool grid_barrier(Bar *b, unsigned nblocks, unsigned &local_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned target = local_gen + 1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == nblocks) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        if (++spins > 20000000u) { ok = false; atomicExch(&b->fail, 1u); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  local_gen += 1;
  __syncthreads();
  return ok;
}

// work: y[i] = a * x[i] + stage over n doubles, grid-stride; then barrier; repeated `stages` times
__global__ __launch_bounds__(256) void k_persist_xcd(Bar *b, const double *x, double *y, int n, int stages, int do_work) {
  // only the workgroups of ONE XCD take part (workgroups go round robin over the 8 XCDs): the barrier's atomics and the data
  // stay in that XCD's L2
  if (blockIdx.x & 7) return;
  unsigned gen = 0;
  const unsigned nb = gridDim.x >> 3, bid = blockIdx.x >> 3;
  for (int s = 0; s < stages; ++s) {
    if (do_work) {
      const double *in = (s & 1) ? y : x;
      double *out = (s & 1) ? (double *)x : y;
      for (int i = bid * 256 + threadIdx.x; i < n; i += nb * 256) out[i] = in[i] * 1.0000001 + s;
    }
    if (!grid_barrier(b, nb, gen)) return;
  }
}
__global__ __launch_bounds__(256) void k_persist(Bar *b, const double *x, double *y, int n, int stages, int do_work) {
  unsigned gen = 0;
  const unsigned nb = gridDim.x;
  for (int s = 0; s < stages; ++s) {
    if (do_work) {
      const double *in = (s & 1) ? y : x;
      double *out = (s & 1) ? (double *)x : y;
      for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) out[i] = in[i] * 1.0000001 + s;
    }
    if (!grid_barrier(b, nb, gen)) return;
  }
}
__global__ __launch_bounds__(256) void k_stage(const double *in, double *out, int n, int s, int do_work) {
  if (do_work) for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = in[i] * 1.0000001 + s;
}

int main() {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, dev));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persist, 256, 0));
  printf("CUs %d, resident blocks of 256 per CU %d, cooperative launch %d\n", p.multiProcessorCount, per_cu, p.cooperativeLaunch);
  const int n = 400000 * 6;  // ~19 MB in, 19 MB out per stage: one estimate pass
  double *x, *y; Bar *b;
  CK(hipMalloc(&x, sizeof(double) * n)); CK(hipMalloc(&y, sizeof(double) * n)); CK(hipMalloc(&b, sizeof(Bar)));
  CK(hipMemset(x, 0, sizeof(double) * n)); CK(hipMemset(y, 0, sizeof(double) * n));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int stages = 200;
  for (int bpc = 1; bpc <= std::min(per_cu, 4); bpc *= 2) {
    for (int work = 0; work < 2; ++work) {
      const int grid = p.multiProcessorCount * bpc;
      CK(hipMemsetAsync(b, 0, sizeof(Bar), st));
      int nn = n, stg = stages, w = work;
      void *args[] = { &b, &x, &y, &nn, &stg, &w };
      // warm-up + timed
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(b, 0, sizeof(Bar), st));
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((const void *)k_persist, dim3(grid), dim3(256), args, 0, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
      }
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      Bar hb; CK(hipMemcpy(&hb, b, sizeof hb, hipMemcpyDeviceToHost));
      printf("persistent grid %5d work %d: %.2f us per stage+barrier (fail %u)\n", grid, work, 1000.0 * ms / stages, hb.fail);
    }
  }
  for (int per_xcd = 8; per_xcd <= 64; per_xcd *= 2) {
    for (int work = 0; work < 2; ++work) {
      const int grid = 8 * per_xcd;
      int nn = work ? 16384 * 6 : n, stg = stages, w = work;  // a coarse level: 16k pairs
      void *args[] = { &b, &x, &y, &nn, &stg, &w };
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(b, 0, sizeof(Bar), st));
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((const void *)k_persist_xcd, dim3(grid), dim3(256), args, 0, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
      }
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      Bar hb; CK(hipMemcpy(&hb, b, sizeof hb, hipMemcpyDeviceToHost));
      printf("one XCD, %3d workgroups, work %d (0.8 MB per stage): %.2f us per stage+barrier (fail %u)\n", per_xcd, work, 1000.0 * ms / stages, hb.fail);
    }
  }
  for (int work = 0; work < 2; ++work) {
    const int grid = 1568;  // what the product kernels launch for 400k entries
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(k_stage, dim3(grid), dim3(256), 0, st, (s & 1) ? y : x, (s & 1) ? x : y, n, s, work);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launch chain  grid %5d work %d: %.2f us per kernel\n", grid, work, 1000.0 * ms / stages);
  }
  return 0;
}
