// Device calibration: what does THIS device sustain on a bare bf16 MFMA loop under its power
// cap?  The convolution kernels are priced against the 2.5 PFLOP/s datasheet figure
// (bench.py `roofline.peak`, MI355X_MICROARCH.md); a chip held at 1.3 kW runs matrix-heavy code
// at 2.0-2.2 GHz, not 2.4, and boxes of one pool differ by a few per cent.  bench.py times this
// loop in the process of the run and reports it beside the datasheet figure
// (`roofline.sustained`), so that a reader can tell a slow box from a slow kernel.
//
// The loop is the convolution kernels' inner structure with everything else removed: a
// 512-thread workgroup per CU (two waves per SIMD), each wave a [128 x 64] accumulator block,
// 12 operand fragments per 32-deep k-step read from LDS by ds_read_b128 (conflict-free,
// lane-linear), no global traffic, no barriers.  Operands are the caller's 64 KB of bf16 values
// (random in [-1, 1): all-zero operands draw less power and over-state the ceiling).
#include "scl_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void mfma_bf16_loop_kernel(const unsigned* __restrict__ in,
                                                                float* __restrict__ out,
                                                                int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[16 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += 512) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const u32x4* frag = reinterpret_cast<const u32x4*>(lds) + lane;
  float sink = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
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
                __builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[n]),
                acc[2 * j + n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) sink += acc[j][q];
  } else {
    f32x4 acc[32];
#pragma unroll
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
              __builtin_bit_cast(bf16x8, a[j]), __builtin_bit_cast(bf16x8, b[n]), acc[4 * j + n],
              0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) sink += acc[j][q];
  }
  // one float per workgroup: keeps the chain alive and lets a test check the arithmetic
  if (threadIdx.x == 0) out[blockIdx.x] = sink;
}

}  // namespace

extern "C" double scl_calibrate_mfma_bf16_flops(int workgroups, int iters) {
  if (workgroups < 1 || iters < 1) return 0.0;
  return (double)workgroups * 8.0 * (double)iters * 2.0 * 128.0 * 64.0 * 32.0;
}

extern "C" int scl_calibrate_mfma_bf16(int shape, int workgroups, int iters, const void* operands,
                                       float* sink, void* stream) {
  if (!operands || !sink) return SCL_E_NULL;
  if ((shape != 16 && shape != 32) || iters < 1 || iters > (1 << 24)) return SCL_E_KIND;
  if (workgroups < 1 || workgroups > 65536 || ((uintptr_t)operands % 16)) return SCL_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (shape == 32)
    SCL_LAUNCH("mfma_bf16_loop_kernel<32>", mfma_bf16_loop_kernel<32>, dim3(workgroups), dim3(512),
               0, st, (const unsigned*)operands, sink, iters);
  else
    SCL_LAUNCH("mfma_bf16_loop_kernel<16>", mfma_bf16_loop_kernel<16>, dim3(workgroups), dim3(512),
               0, st, (const unsigned*)operands, sink, iters);
  return scl_launch_status();
}
