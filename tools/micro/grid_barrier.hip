// developer tool: what does ONE grid-wide barrier cost on gfx950, bare and with freshly written data to publish, against the
// dependent kernel boundary it would replace?  (round-3 review: "measure the bare cost of a grid barrier ... per-XCD arrive in
// L2, then one cross-XCD word"; the question behind it: should the 30 external substeps of a small tile be ONE launch?)
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
// Three ways to separate NPH phases of the same tiny job (every workgroup rewrites `bytes` bytes of its slab, then reads its
// neighbour workgroup's slab -- a cross-workgroup, mostly cross-XCD dependence like a stencil's halo):
//   launches   NPH kernel launches on one stream
//   flat       one launch, one monotonic counter (release fence, add, relaxed poll, acquire fence): k_ext_loop's barrier
//   xcd        one launch, hierarchical: a counter per XCD (workgroup -> XCD = blockIdx.x & 7), the XCD's last arriver releases
//              (its L2's write-back), adds to the top counter, polls it, acquires and bumps the XCD's generation word, which the
//              other workgroups of the XCD poll; each of them acquires (its CU's L1).
// Every spin is bounded (a workgroup that waits ~0.5 s raises an abort word and everybody leaves): the grid always drains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define NPH 64
#define LINE 32   // unsigned per 128-byte line

__device__ __forceinline__ unsigned ld_relaxed(unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void phase_work(double *slab, size_t per, int nwg, int ph, double *sink) {
  // write my slab, read the slab of the workgroup "across" (another XCD for most)
  const int me = blockIdx.x, other = (me + nwg / 2 + 1) % nwg;
  double acc = 0.;
  for (size_t n = threadIdx.x; n < per; n += blockDim.x) acc += slab[(size_t)other * per + n];
  for (size_t n = threadIdx.x; n < per; n += blockDim.x) slab[(size_t)me * per + n] = acc * 1e-30 + (double)(ph + 1);
  if (acc == -1.) *sink = acc;
}

__global__ void k_phase(double *slab, size_t per, int nwg, int ph, double *sink) { phase_work(slab, per, nwg, ph, sink); }

// bar[0]: arrivals, bar[LINE]: abort
__device__ __forceinline__ bool barrier_flat(unsigned *bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1;
    for (unsigned spin = 0;; spin++) {
      if ((int)(ld_relaxed(&bar[0]) - target) >= 0) break;
      if (ld_relaxed(&bar[LINE]) != 0u || spin > (1u << 21)) { __hip_atomic_store(&bar[LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
// bar[x * LINE] x = 0..7: arrivals of XCD x; bar[(8 + x) * LINE]: generation of XCD x; bar[16 * LINE]: top; bar[17 * LINE]: abort
__device__ __forceinline__ bool barrier_xcd(unsigned *bar, unsigned gen, unsigned per_xcd) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const int x = blockIdx.x & 7;
    int good = 1;
    const unsigned old = __hip_atomic_fetch_add(&bar[x * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == gen * per_xcd) {                            // the XCD's last arriver: publish for the whole XCD
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // this XCD's L2 writes its dirty lines back
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&bar[16 * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (unsigned spin = 0;; spin++) {
        if ((int)(ld_relaxed(&bar[16 * LINE]) - gen * 8u) >= 0) break;
        if (ld_relaxed(&bar[17 * LINE]) != 0u || spin > (1u << 21)) { __hip_atomic_store(&bar[17 * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __hip_atomic_store(&bar[(8 + x) * LINE], good ? gen : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (unsigned spin = 0;; spin++) {
        const unsigned g = ld_relaxed(&bar[(8 + x) * LINE]);
        if (g == 0xFFFFFFFFu) { good = 0; break; }
        if ((int)(g - gen) >= 0) break;
        if (ld_relaxed(&bar[17 * LINE]) != 0u || spin > (1u << 21)) { __hip_atomic_store(&bar[17 * LINE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // this CU's L1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
template <int KIND> __global__ void __launch_bounds__(256) k_loop(double *slab, size_t per, int nwg, unsigned *bar, unsigned base, double *sink, int *err) {
  for (int ph = 0; ph < NPH; ph++) {
    phase_work(slab, per, nwg, ph, sink);
    const bool ok = KIND == 0 ? barrier_flat(bar, base + (unsigned)(ph + 1) * (unsigned)nwg) : barrier_xcd(bar, base + (unsigned)(ph + 1), (unsigned)(nwg / 8));
    if (!ok) { if (threadIdx.x == 0) *err = 1; return; }
  }
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  unsigned *bar; double *slab, *sink; int *err;
  hipMalloc(&bar, 32 * LINE * sizeof(unsigned)); hipMalloc(&sink, 8); hipMalloc(&err, 4);
  const size_t maxbytes = 64 << 10;
  hipMalloc(&slab, (size_t)4 * ncu * maxbytes);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("%d CUs; %d phases per measurement; us per phase boundary = (time - time of the same phases' work alone) is not separable: rows give TOTAL us per phase\n", ncu, NPH);
  printf("%-10s %-8s %12s %12s %12s\n", "wg/CU", "KB/wg", "launches", "flat", "xcd");
  for (int wpc = 1; wpc <= 2; wpc++)
    for (size_t bytes : {(size_t)0, (size_t)2048, (size_t)20480, (size_t)65536}) {
      const int nwg = ncu * wpc;                                // a multiple of 8 on this chip (256 CUs)
      const size_t per = bytes / 8;
      double us[3] = {0, 0, 0};
      for (int kind = 0; kind < 3; kind++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
          hipMemsetAsync(bar, 0, 32 * LINE * sizeof(unsigned), st); hipMemsetAsync(err, 0, 4, st);
          hipMemsetAsync(slab, 0, (size_t)nwg * bytes + 8, st);
          hipStreamSynchronize(st);
          hipEventRecord(a, st);
          if (kind == 0) for (int ph = 0; ph < NPH; ph++) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(256), 0, st, slab, per, nwg, ph, sink);
          if (kind == 1) hipLaunchKernelGGL(k_loop<0>, dim3(nwg), dim3(256), 0, st, slab, per, nwg, bar, 0u, sink, err);
          if (kind == 2) hipLaunchKernelGGL(k_loop<1>, dim3(nwg), dim3(256), 0, st, slab, per, nwg, bar, 0u, sink, err);
          hipEventRecord(b, st);
          hipStreamSynchronize(st);
          float ms; hipEventElapsedTime(&ms, a, b);
          int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
          if (e) { printf("barrier gave up (kind %d)\n", kind); ms = -1.f; }
          // check: every slab holds NPH
          if (per) { double v; hipMemcpy(&v, slab + (size_t)(nwg - 1) * per, 8, hipMemcpyDeviceToHost); if (v != (double)NPH) printf("wrong data (kind %d): %g\n", kind, v); }
          if (ms < best) best = ms;
        }
        us[kind] = best * 1e3 / NPH;
      }
      printf("%-10d %-8.1f %12.2f %12.2f %12.2f\n", wpc, bytes / 1024., us[0], us[1], us[2]);
    }
  return 0;
}
