// fetch_calib.hip — what rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access widths of the solve kernel.
// The solve kernel moves doubles: 8 bytes per lane per instruction, contiguous over the lanes of a roll-out or the
// same address for all lanes of a group.  The counters are documented for 16-byte-per-lane streaming only, so this
// program reads / writes buffers of a known size (1 GiB, four times the Infinity Cache) in those patterns:
//   read8    every lane loads consecutive doubles            (global_load_dwordx2, 512 contiguous bytes per wave)
//   read16   every lane loads consecutive double2            (global_load_dwordx4, the documented case)
//   bcast8   the 32 lanes of a half wave load the SAME double, consecutive half waves consecutive doubles
//   write8   every lane stores consecutive doubles
// Run each under its own counter pass:  rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib   (then WRITE_SIZE)
// and compare the counter of each kernel with the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read8(const double* __restrict__ p, size_t n, double* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (; i < n; i += stride) s += p[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  double s = 0.0;
  for (; i < n; i += stride) { double2 v = p[i]; s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
__global__ void bcast8(const double* __restrict__ p, size_t n, double* out) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 32, stride = ((size_t)gridDim.x * blockDim.x) / 32;
  double s = 0.0;
  for (; i < n; i += stride) s += p[i];
  if (s == 12345.678) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (double)i;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / sizeof(double);
  double *buf = nullptr, *out = nullptr;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipDeviceSynchronize());
  const dim3 grid(256 * 16), block(256);
  hipLaunchKernelGGL(read8, grid, block, 0, 0, buf, n, out);
  hipLaunchKernelGGL(read16, grid, block, 0, 0, (const double2*)buf, n / 2, out);
  hipLaunchKernelGGL(bcast8, grid, block, 0, 0, buf, n / 32, out);      // 1/32 of the buffer: 32 MiB of distinct bytes
  hipLaunchKernelGGL(write8, grid, block, 0, 0, buf, n);
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  printf("read8 bytes %zu\nread16 bytes %zu\nbcast8 distinct bytes %zu (requested by lanes: %zu)\nwrite8 bytes %zu\n",
         bytes, bytes, bytes / 32, bytes, bytes);
  CHECK(hipFree(buf));
  CHECK(hipFree(out));
  return 0;
}
