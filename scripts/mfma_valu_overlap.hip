// Do matrix instructions and vector-ALU work of two DIFFERENT waves on one SIMD overlap?
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a chain of matrix
// instructions, waves 4-7 (the second wave of every SIMD) run independent v_fma_f32 chains.
// Times: matrix team alone, vector team alone, both.  "both ~ max" means the two pipes run
// side by side; "both ~ sum" means they share the datapath.  Done for the f32-input MFMA the
// NetVLAD / Gram / top-n kernels use and, for contrast, for a bf16 MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// PRIO: 0 none, 1 s_setprio 1 on the vector team, 2 s_setprio 1 on the matrix team.
// SWAP: the matrix team is the second-dispatched (younger) half of the workgroup.
template <int KIND, int PRIO = 0, int SWAP = 0>   // KIND 0: 32x32x2 f32, 1: 32x32x16 bf16
__global__ __launch_bounds__(512) void overlap(const float* __restrict__ in, float* out,
                                               int n_mfma, int n_valu) {
  const int wid = threadIdx.x >> 6;
  float sink = 0.f;
  const bool matrix_team = SWAP ? wid >= 4 : wid < 4;
  if (PRIO == 1 && !matrix_team) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && matrix_team) __builtin_amdgcn_s_setprio(1);
  if (matrix_team) {
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    if (KIND == 0) {
      const float a = in[threadIdx.x], b = in[512 + threadIdx.x];
      for (int i = 0; i < n_mfma; i += 2) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      }
    } else {
      bf16x8 a, b;
      for (int k = 0; k < 8; ++k) {
        a[k] = (__bf16)in[(threadIdx.x + k) & 1023];
        b[k] = (__bf16)in[(threadIdx.x + 8 + k) & 1023];
      }
      for (int i = 0; i < n_mfma; i += 2) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[1], 0, 0, 0);
      }
    }
    for (int j = 0; j < 2; ++j)
      for (int q = 0; q < 16; ++q) sink += acc[j][q];
  } else {
    float x[8];
    const float m = in[threadIdx.x & 255], c = in[256 + (threadIdx.x & 255)];
    for (int j = 0; j < 8; ++j) x[j] = in[(threadIdx.x + j) & 1023];
    for (int i = 0; i < n_valu; i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], m, c);
    }
    for (int j = 0; j < 8; ++j) sink += x[j];
  }
  if (sink == 123456.789f) out[0] = sink;
}

template <int KIND, int PRIO = 0, int SWAP = 0>
float run(const float* in, float* out, int n_mfma, int n_valu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((overlap<KIND, PRIO, SWAP>), dim3(256), dim3(512), 0, 0, in, out, n_mfma, n_valu);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((overlap<KIND, PRIO, SWAP>), dim3(256), dim3(512), 0, 0, in, out, n_mfma, n_valu);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float *in, *out;
  hipMalloc(&in, 1024 * sizeof(float));
  hipMalloc(&out, sizeof(float));
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.0001f * (float)i;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int n_mfma = 20000;       // 32x32x2 f32: 64 cycles each -> 1.28 M cycles
  const int n_valu_f32 = 320000;  // wave64 v_fma_f32: 4 cycles each -> 1.28 M cycles
  printf("one wave of each team per SIMD; times in us\n");
  {
    const float a = run<0>(in, out, n_mfma, 0), b = run<0>(in, out, 0, n_valu_f32),
                c = run<0>(in, out, n_mfma, n_valu_f32);
    printf("v_mfma_f32_32x32x2_f32   x %d : %8.1f   v_fma_f32 x %d : %8.1f   both : %8.1f   "
           "(sum %.1f, max %.1f)\n", n_mfma, a, n_valu_f32, b, c, a + b, a > b ? a : b);
  }
  {
    const int n_bf = 40000;       // 32x32x16 bf16: 32 cycles each -> 1.28 M cycles
    const float a = run<1>(in, out, n_bf, 0), b = run<1>(in, out, 0, n_valu_f32),
                c = run<1>(in, out, n_bf, n_valu_f32);
    printf("v_mfma_f32_32x32x16_bf16 x %d : %8.1f   v_fma_f32 x %d : %8.1f   both : %8.1f   "
           "(sum %.1f, max %.1f)\n", n_bf, a, n_valu_f32, b, c, a + b, a > b ? a : b);
  }
  // who yields?  priorities and dispatch order
  {
    const int n_bf = 40000;
    printf("bf16 both, vector team prio 1 : %8.1f\n", run<1, 1, 0>(in, out, n_bf, n_valu_f32));
    printf("bf16 both, matrix team prio 1 : %8.1f\n", run<1, 2, 0>(in, out, n_bf, n_valu_f32));
    printf("bf16 both, matrix team younger: %8.1f\n", run<1, 0, 1>(in, out, n_bf, n_valu_f32));
    printf("bf16 both, matrix younger+prio: %8.1f\n", run<1, 2, 1>(in, out, n_bf, n_valu_f32));
    printf("f32  both, vector team prio 1 : %8.1f\n", run<0, 1, 0>(in, out, n_mfma, n_valu_f32));
    printf("f32  both, matrix team younger: %8.1f\n", run<0, 0, 1>(in, out, n_mfma, n_valu_f32));
    // half the vector work: does it hide completely under the matrix chain?
    printf("bf16 both, half the vector work: %8.1f (matrix alone %.1f)\n",
           run<1, 0, 0>(in, out, n_bf, n_valu_f32 / 2), run<1, 0, 0>(in, out, n_bf, 0));
    printf("f32  both, half the vector work: %8.1f (matrix alone %.1f)\n",
           run<0, 0, 0>(in, out, n_mfma, n_valu_f32 / 2), run<0, 0, 0>(in, out, n_mfma, 0));
  }
  return 0;
}
