// HBM rates of an MI355X as a kernel sees them: write-only, read-only and copy streams of float4
// (sizes past the 256 MB Infinity Cache), plus the write pattern of the conv epilogue (256-byte
// runs, 4 runs per wave-instruction).  The 1x1 convs of the bench step are output-dominated:
// their roof is the WRITE rate.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/mem_rate.hip -o tools/micro/mem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void wr(f32x4* p, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void wr_nt(f32x4* p, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, &p[i]);
}
__global__ void rd(const f32x4* p, size_t n, float* out) {
  f32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i];
  if (a.x == 123.f) out[0] = a.y + a.z + a.w;
}
__global__ void cp(const f32x4* s, f32x4* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// epilogue-like: a wave-instruction writes 4 runs of 256 B, 512 B apart (64 of 128 channels of 4 pixels)
__global__ void wr_epi(f32x4* p, size_t npix) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t q = (size_t)blockIdx.x * 4 + wave; q < npix / 4; q += (size_t)gridDim.x * 4)
    for (int half = 0; half < 2; ++half)
      p[(q * 4 + (lane >> 4)) * 32 + half * 16 + (lane & 15)] = v;
}
template <class F> static float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}
int main() {
  const size_t bytes = (size_t)1 << 30;                 // 1 GiB per stream
  const size_t n = bytes / 16;
  f32x4 *a, *b; float* o;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&o, 64);
  (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
  for (int blocks : {1024, 2048, 8192}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, a, n); });
    printf("blocks %5d  write      %.2f TB/s\n", blocks, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(wr_nt, dim3(blocks), dim3(256), 0, 0, a, n); });
    printf("blocks %5d  write nt   %.2f TB/s\n", blocks, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(wr_epi, dim3(blocks), dim3(256), 0, 0, a, n / 32); });
    printf("blocks %5d  write epi  %.2f TB/s\n", blocks, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, a, n, o); });
    printf("blocks %5d  read       %.2f TB/s\n", blocks, bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL(cp, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    printf("blocks %5d  copy       %.2f TB/s (read + write)\n", blocks, 2.0 * bytes / t / 1e9);
  }
  return 0;
}
