// fetch_calib: what rocprofv3's FETCH_SIZE / WRITE_SIZE report for a KNOWN number of streamed bytes at 4 / 8 / 12 / 16 bytes per lane
// on gfx950 (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads `bytes` once
// (grid-stride, every lane one element of the width per trip) and writes one word per block.  Run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetch_calib
// and divide Counter_Value * 1024 by the byte count printed here (scripts/fetch_calib.sh does both).
// Build: hipcc -O2 --offload-arch=gfx950 tools/fetch_calib.cpp -o tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct u32x3 { unsigned a, b, c; };

template <typename V> __device__ unsigned fold(V v);
template <> __device__ unsigned fold<unsigned>(unsigned v) { return v; }
template <> __device__ unsigned fold<u32x2>(u32x2 v) { return v[0] ^ v[1]; }
template <> __device__ unsigned fold<u32x3>(u32x3 v) { return v.a ^ v.b ^ v.c; }
template <> __device__ unsigned fold<u32x4>(u32x4 v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }

template <typename V> __global__ void read_kernel(const V* __restrict__ p, long n, unsigned* out) {
  unsigned acc = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc ^= fold<V>(p[i]);
  if (acc == 0x12345678u) out[blockIdx.x] = acc;  // (never true for the fill pattern; keeps the loads)
}
// the 12-byte packed RGB taps of the photometric / smoothness kernels: three dwords at a 12-byte stride
extern "C" __global__ void read12_kernel(const unsigned* __restrict__ p, long n3, unsigned* out) {
  unsigned acc = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n3; i += (long)gridDim.x * blockDim.x) {
    const u32x3 v = *(const u32x3*)(p + 3 * i);
    acc ^= v.a ^ v.b ^ v.c;
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
extern "C" __global__ void write16_kernel(u32x4* __restrict__ p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = u32x4{1u, 2u, 3u, (unsigned)i};
}

int main() {
  const long bytes = 768L << 20;  // 768 MiB: three times the Infinity Cache, so the reads come from HBM
  void* buf; unsigned* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(buf, 0x5a, bytes));
  CK(hipDeviceSynchronize());
  const int grid = 4096, block = 256;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(read_kernel<unsigned>, dim3(grid), dim3(block), 0, 0, (const unsigned*)buf, bytes / 4, out);
    hipLaunchKernelGGL(read_kernel<u32x2>, dim3(grid), dim3(block), 0, 0, (const u32x2*)buf, bytes / 8, out);
    hipLaunchKernelGGL(read12_kernel, dim3(grid), dim3(block), 0, 0, (const unsigned*)buf, bytes / 12, out);
    hipLaunchKernelGGL(read_kernel<u32x4>, dim3(grid), dim3(block), 0, 0, (const u32x4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(write16_kernel, dim3(grid), dim3(block), 0, 0, (u32x4*)buf, bytes / 16);
    CK(hipDeviceSynchronize());
  }
  printf("bytes_per_kernel %ld (read 4 / 8 / 12 / 16 bytes per lane, write 16 bytes per lane)\n", bytes);
  return 0;
}
