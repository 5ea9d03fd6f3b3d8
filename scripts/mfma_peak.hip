// Calibration: what does a bare f32-input MFMA loop reach on this device, all CUs busy?
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints TFLOP/s for v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 at 1, 2 and 4 waves
// per SIMD (operands in registers, 4 independent accumulators, random data).  The NetVLAD /
// Gram / top-n kernels are priced against the 157.3 TF datasheet figure; this is the number
// the silicon actually sustains, measured in the same session.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ in, float* out,
                                                 int iters) {
  const float a0 = in[threadIdx.x], b0 = in[256 + threadIdx.x];
  float a = a0, b = b0;
  float sink = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      a = -a;
    }
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 16; ++q) sink += acc[j][q];
  } else {
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      a = -a;
    }
    for (int j = 0; j < 4; ++j)
      for (int q = 0; q < 4; ++q) sink += acc[j][q];
  }
  if (sink == 123456.789f) out[0] = sink;   // keep the chain alive
}

template <int SHAPE>
double run(int blocks_per_cu, int iters, const float* in, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(grid), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
  const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * flop_per_mfma;
  return flops / (ms * 1e-3) / 1e12;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 512 * sizeof(float));
  hipMalloc(&out, sizeof(float));
  float h[512];
  srand(1);
  for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  // 16 / 32 iterations = 512 / 1024 MFMAs per wave: the per-wave work of one NetVLAD tile
  for (int iters : {16, 32, 64, 200, 2000}) {
    for (int bpc : {1, 2, 4}) {
      printf("iters %5d  waves/SIMD %d  32x32x2: %7.1f TF   16x16x4: %7.1f TF\n", iters, bpc,
             run<32>(bpc, iters, in, out), run<16>(bpc, iters, in, out));
    }
  }
  return 0;
}
