// Does the bf16 MFMA SHAPE matter on this device at equal work?  (MI355X_MICROARCH.md, DVFS
// give-back item 7: the chip may hold a higher clock on 16x16x32 than on 32x32x16.)
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_shape_bf16.hip -o /tmp/mfma_shape && /tmp/mfma_shape
// Each wave computes a [128 x 64] output block over K (random bf16 operands re-read from LDS by
// ds_read_b128 every step, like the convolution kernels): 8 accumulators of 32x32 or 32 of
// 16x16, 12 fragment reads per 32-deep K step either way.  512-thread workgroups (two waves per
// SIMD), one per CU.  Prints TFLOP/s and the in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void loop(const unsigned* __restrict__ in, float* out,
                                               unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16 * 1024];   // 64 KB of operands
  for (int i = threadIdx.x; i < 16 * 1024; i += 512) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const u32x4* frag = reinterpret_cast<const u32x4*>(lds) + lane;    // conflict-free: lane-linear
  float sink = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const u32x4* f = frag + ((it & 15) << 6) * 12 % 3072;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 a[4], b[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = f[(ks * 6 + j) * 64];
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = f[(ks * 6 + 4 + n) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[2 * j + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[n]), acc[2 * j + n], 0, 0, 0);
      }
    }
    for (int j = 0; j < 8; ++j)
      for (int q = 0; q < 16; ++q) sink += acc[j][q];
  } else {
    f32x4 acc[32];
    for (int j = 0; j < 32; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      const u32x4* f = frag + ((it & 15) << 6) * 12 % 3072;
      u32x4 a[8], b[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = f[j * 64];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = f[(8 + n) * 64];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[4 * j + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[n]), acc[4 * j + n], 0, 0, 0);
    }
    for (int j = 0; j < 32; ++j)
      for (int q = 0; q < 4; ++q) sink += acc[j][q];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (sink == 123456.789f) out[0] = sink;
}

template <int SHAPE>
void run(int iters, const unsigned* in, float* out, unsigned long long* clk) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(loop<SHAPE>, dim3(grid), dim3(512), 0, 0, in, out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(loop<SHAPE>, dim3(grid), dim3(512), 0, 0, in, out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  unsigned long long h[512];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double flops = (double)grid * 8 * iters * 2.0 * 128 * 64 * 32;
  printf("shape %2d  iters %5d  %.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz  cycles/iter %.1f\n",
         SHAPE, iters, ms, flops / (ms * 1e-3) / 1e12, ghz, (double)h[0] / iters);
}

int main() {
  unsigned* in;
  float* out;
  unsigned long long* clk;
  hipMalloc(&in, 16 * 1024 * 4);
  hipMalloc(&out, 4);
  hipMalloc(&clk, 512 * 8);
  static unsigned h[16 * 1024];
  srand(1);
  for (int i = 0; i < 16 * 1024; ++i) {
    // two random bf16 values in [-1, 1) per word
    auto bf = [] { float v = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &v, 4); return u >> 16; };
    h[i] = bf() | (bf() << 16);
  }
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep)
    for (int iters : {2000, 8000}) {
      run<32>(iters, in, out, clk);
      run<16>(iters, in, out, clk);
    }
  return 0;
}
