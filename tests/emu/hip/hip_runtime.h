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
#include <functional>
#include <vector>

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
// Streams.  Work on a stream runs at once, in program order (the EARLIEST moment the real device could run it) -- unless the stream
// has been marked `deferred` (emu_stream_defer: the library's side stream, when POMGPU_EMU_DEFER_SIDE is set in the environment):
// then its launches wait in a queue and run at the LATEST moment the real device could run them -- when another stream waits
// for an event recorded behind them, or when the host synchronises.  A consumer on the main stream that forgot to wait for its
// side-stream producer, or a side-stream kernel whose operands the main stream overwrites before the join, computes the same
// bits in the eager mode and different ones in the deferred mode: tests/test_kernels_emulated_tiles.py runs both.
struct emu_stream_ {
  std::vector<std::function<void()>> q;   // deferred work, oldest first
  size_t done = 0;                        // items of q already run (q is cleared when it has been run to its end)
  size_t base = 0;                        // items dropped from q so far: positions in events count from the stream's creation
  bool deferred = false;
};
struct emu_event_ { emu_stream_ *s = nullptr; size_t upto = 0; };   // recorded on deferred stream s behind its first `upto` items
typedef emu_stream_ *hipStream_t;
typedef emu_event_ *hipEvent_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipStreamNonBlocking = 1 };
static inline void emu_stream_run(emu_stream_ *s, size_t upto) {   // upto: position since the stream's creation; (size_t)-1 = everything
  if (!s) return;
  upto = upto <= s->base ? 0 : upto - s->base;
  if (upto > s->q.size()) upto = s->q.size();
  while (s->done < upto) { auto f = std::move(s->q[s->done]); s->done++; f(); }
  if (s->done == s->q.size()) { s->base += s->q.size(); s->q.clear(); s->done = 0; }
}
static inline bool emu_deferred(hipStream_t s) { return s && s->deferred; }
static inline void emu_stream_defer(hipStream_t s) { if (s && getenv("POMGPU_EMU_DEFER_SIDE")) s->deferred = true; }

static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { if (emu_deferred(s)) emu_stream_run(s, (size_t)-1); return hipSuccess; }
// copies and fills on a deferred stream: behind everything that stream still holds (the source may be a host temporary)
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t st) { hipStreamSynchronize(st); memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { hipStreamSynchronize(st); memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, int) { *s = new emu_stream_(); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { hipStreamSynchronize(s); delete s; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event_(); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, int) { *e = new emu_event_(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (e) { e->s = emu_deferred(s) ? s : nullptr; e->upto = emu_deferred(s) ? s->base + s->q.size() : 0; }
  return hipSuccess;
}
// an eager stream has run everything it was given: waiting for it is a no-op; waiting for a deferred stream runs that stream up to the event
static inline hipError_t hipStreamWaitEvent(hipStream_t waiter, hipEvent_t e, int) {
  if (e && e->s && e->s != waiter) emu_stream_run(e->s, e->upto);
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t e) { if (e && e->s) emu_stream_run(e->s, e->upto); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// Lanes along x are run as workgroups of width 1 (blockDim.x = 1, gridDim.x scaled): global indices
// blockIdx.x*blockDim.x+threadIdx.x are unchanged, and every lane is both the first and the last
// lane of its "wavefront", so the neighbour-lane helpers (lane_w / lane_e) always take their
// edge-lane path and the value of the shuffle below is never used.
static inline double __shfl_up(double x, unsigned, int) { return x; }
static inline double __shfl_down(double x, unsigned, int) { return x; }
template <typename K, typename... A>
static inline void emu_run(K kern, dim3 g, dim3 b, const A &...a) {
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
template <typename K, typename... A>
static inline void emu_launch(hipStream_t st, K kern, dim3 g, dim3 b, const A &...a) {
  if (emu_deferred(st)) st->q.push_back([=]() { emu_run(kern, g, b, a...); });   // arguments by value, as a real launch takes them
  else emu_run(kern, g, b, a...);
}
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) emu_launch(stream, kern, dim3(grid), dim3(block), __VA_ARGS__)
static inline void __syncthreads() {}
#define __shared__ static thread_local
