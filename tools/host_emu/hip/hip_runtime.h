// Host stand-in for <hip/hip_runtime.h>: lets the solve kernel be compiled as plain C++ and run one
// lane at a time under AddressSanitizer / gdb.  Development aid only; meaningful for lane-group width 1
// (a lane never talks to another lane there), every cross-lane primitive degenerates to the identity.
#pragma once
#define _GNU_SOURCE 1
#include <math.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__
// dynamic LDS of the emulated wave (block-scope `extern __shared__` declarations inside namespace alilqr bind here)
namespace alilqr { extern double lds_pool[]; }

struct emu_dim3 { unsigned x = 1, y = 1, z = 1; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;
inline thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

// ---- lane groups as threads: the G lanes of a group run as G threads that meet at every cross-lane
//      primitive; groups are independent and run one after the other.
#include <pthread.h>
#include <thread>
#include <vector>
struct emu_group {
  int G = 1;
  pthread_barrier_t bar;
  unsigned long long slot[64];
};
inline thread_local emu_group* emu_grp = nullptr;
inline thread_local int emu_gl = 0;
template <class T> inline T __shfl(T v, int src, int width) {
  static_assert(sizeof(T) <= 8, "");
  emu_group* g = emu_grp;
  if (!g || g->G == 1) return v;
  unsigned long long w = 0; std::memcpy(&w, &v, sizeof(T));
  g->slot[emu_gl] = w;
  pthread_barrier_wait(&g->bar);
  unsigned long long r = g->slot[src & (width - 1)];
  pthread_barrier_wait(&g->bar);
  T o; std::memcpy(&o, &r, sizeof(T));
  return o;
}
// wave-wide form (source = a lane of the wave): only the LEGACY line search uses it, to read the state of another group
// of the wave; groups run one after the other here, so only sources inside the own group can be served
template <class T> inline T __shfl(T v, int src) {
  emu_group* g = emu_grp;
  const int G = g ? g->G : 1, base = (threadIdx.x & 63) & ~(G - 1);
  if (src < base || src >= base + G) { std::fprintf(stderr, "host emulation: cross-group __shfl (LEGACY line search) is not emulated\n"); std::abort(); }
  return __shfl(v, src - base, G);
}
inline long long __double_as_longlong(double d) { long long r; std::memcpy(&r, &d, sizeof r); return r; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// ballot over the group only, placed at the group's bit positions inside the wave
inline unsigned long long __ballot(bool p) {
  emu_group* g = emu_grp;
  if (!g || g->G == 1) return p ? (1ull << (threadIdx.x & 63)) : 0ull;
  g->slot[emu_gl] = p ? 1 : 0;
  pthread_barrier_wait(&g->bar);
  unsigned long long m = 0;
  const int base = (threadIdx.x & 63) & ~(g->G - 1);
  for (int i = 0; i < g->G; ++i) if (g->slot[i]) m |= 1ull << (base + i);
  pthread_barrier_wait(&g->bar);
  return m;
}
// __any is only used to skip work whose result is unused when no lane needs it, from code that not every
// lane of a group reaches: the lane's own predicate is an exact stand-in and needs no rendezvous
inline bool __any(bool p) { return p; }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
// lanes of a wave run in lockstep on the GPU, so a fence orders one lane's stores before the next phase's
// loads of the other lanes; threads need a rendezvous for the same effect
inline void __threadfence() {}
inline void __threadfence_block() { emu_group* g = emu_grp; if (g && g->G > 1) pthread_barrier_wait(&g->bar); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }

typedef int hipError_t;
typedef void* hipStream_t;
constexpr hipError_t hipSuccess = 0;
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
constexpr int hipMemcpyDeviceToHost = 2;
constexpr int hipMemcpyDeviceToDevice = 3;
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
typedef void* hipEvent_t;
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define __align__(n) alignas(n)
#define HIP_SYMBOL(x) (&(x))
template <class T> inline hipError_t hipGetSymbolAddress(void** p, T* sym) { *p = (void*)sym; return hipSuccess; }

// every block, every group, one after the other; the lanes of a group as threads
inline int emu_group_width() { const char* e = std::getenv("ALILQR_LANES"); int g = e ? std::atoi(e) : 1; return g < 1 ? 1 : g; }
#define hipLaunchKernelGGL(kernel, grid, block, shm, stream, ...)                   \
  do {                                                                              \
    const int G_ = emu_group_width();                                               \
    const dim3 grid_ = dim3(grid), block_ = dim3(block);                            \
    for (unsigned b_ = 0; b_ < grid_.x; ++b_)                                       \
      for (unsigned l0_ = 0; l0_ < block_.x; l0_ += G_) {                           \
        emu_group grp_; grp_.G = G_;                                                \
        pthread_barrier_init(&grp_.bar, nullptr, G_);                               \
        std::vector<std::thread> th_;                                               \
        for (int g_ = 0; g_ < G_; ++g_)                                             \
          th_.emplace_back([&, g_] {                                                \
            blockDim = block_; gridDim = grid_; blockIdx.x = b_; threadIdx.x = l0_ + g_;             \
            emu_grp = &grp_; emu_gl = g_;                                           \
            kernel(__VA_ARGS__);                                                    \
          });                                                                       \
        for (auto& t_ : th_) t_.join();                                             \
        pthread_barrier_destroy(&grp_.bar);                                         \
      }                                                                             \
  } while (0)
