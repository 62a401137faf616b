// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A stand-in for <hip/hip_runtime.h> that lets g++ compile the UNMODIFIED kernel sources of
// extpom_amd/csrc for the host, running each launch as a serial loop over the grid.  It exists
// so that kernel logic (index ranges, fusion, boundary handling) can be checked bit-for-bit
// against the CPU oracle in a container that has no GPU.  The result is tests/_emu/libpomgpu_emu.so,
// loaded only by tests/test_kernels_emulated.py.  It is NOT a fallback: the package loader
// (extpom_amd/lib.py) knows nothing about it and fails loudly without a real HIP device.
// Kernels that use cross-lane or LDS operations (k_reduce.hip) are not emulated.
#pragma once
#define POMGPU_EMU 1
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct double2 { double x, y; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
extern thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

typedef int hipError_t;
typedef struct emu_stream_ *hipStream_t;
typedef struct emu_event_ *hipEvent_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipStreamNonBlocking = 1 };

static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { static char tag[64]; static int n = 0; *s = (hipStream_t)&tag[n++ & 63]; return hipSuccess; }   // distinct, non-null, never dereferenced
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, int) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, int) { return hipSuccess; }   // launches run in program order
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// Lanes along x are run as workgroups of width 1 (blockDim.x = 1, gridDim.x scaled): global indices
// blockIdx.x*blockDim.x+threadIdx.x are unchanged, and every lane is both the first and the last
// lane of its "wavefront", so the neighbour-lane helpers (lane_w / lane_e) always take their
// edge-lane path and the value of the shuffle below is never used.
static inline double __shfl_up(double x, unsigned, int) { return x; }
static inline double __shfl_down(double x, unsigned, int) { return x; }
template <typename K, typename... A>
static inline void emu_launch(K kern, dim3 g, dim3 b, const A &...a) {
  gridDim = dim3(g.x * b.x, g.y, g.z);
  blockDim = dim3(1, b.y, b.z);
  for (unsigned bz = 0; bz < g.z; bz++)
    for (unsigned by = 0; by < g.y; by++)
      for (unsigned bx = 0; bx < g.x * b.x; bx++) {
        blockIdx = dim3(bx, by, bz);
        for (unsigned tz = 0; tz < b.z; tz++)
          for (unsigned ty = 0; ty < b.y; ty++) {
            threadIdx = dim3(0, ty, tz);
            kern(a...);
          }
      }
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu_launch(kern, dim3(grid), dim3(block), __VA_ARGS__)
static inline void __syncthreads() {}
#define __shared__ static thread_local
