// Does the LDS-DMA stream (global_load_lds_dwordx4) care whether one wave-instruction reads
// whole 128-byte lines?  Every convolution kernel of this repo is bound by that stream
// (DESIGN.md §7), and their staging reads HALF lines per instruction: 16 pixels x 64 bytes (one
// 32-channel chunk / plane of a channels-last pixel), the other half following in another
// instruction.
//   hipcc --offload-arch=gfx950 -O3 scripts/dma_line_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// One 512-thread workgroup per CU streams its slice of a 1 GB buffer of 128-byte "pixels" into a
// ring of 16 KB LDS stages (2 / 4 / 8 stages: 16 / 48 / 112 KB in flight per CU), in three lane -> address maps:
//   0  whole lines:  instruction j, lane l -> pixel 8 j + l / 8, piece l % 8
//   1  half lines:   instruction j, lane l -> pixel 16 (j / 2) + l / 4, piece 4 (j % 2) + l % 4
//                    (both halves of a line in consecutive instructions of the same wave)
//   2  half lines, halves far apart: first all first halves of the stage, then all second halves
// The buffer is larger than the Infinity Cache; a second pass over a 96 MB buffer shows the
// cache-resident case.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const char* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_byte)
      : "memory");
}

constexpr int STAGE = 16 * 1024;      // bytes per stage = 128 pixels

template <int MAP, int NST>
__global__ __launch_bounds__(512, 1) void stream(const char* __restrict__ buf, long long bytes_per_wg,
                                                 int passes, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = buf + (long long)blockIdx.x * bytes_per_wg;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int per_pass = (int)(bytes_per_wg / STAGE);
  const int nstage = per_pass * passes;
  // a stage = 16 instructions of 1 KB, two per wave
  auto issue = [&](int st) {
    const char* sb = base + (long long)(st % per_pass) * STAGE;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int j = wid + 8 * n;                       // instruction 0 .. 15 of the stage
      int pix, piece;
      if (MAP == 0) {
        pix = 8 * j + (lane >> 3);
        piece = lane & 7;
      } else if (MAP == 1) {
        pix = 16 * (j >> 1) + (lane >> 2);
        piece = 4 * (j & 1) + (lane & 3);
      } else {
        pix = 16 * (j & 7) + (lane >> 2);
        piece = 4 * (j >> 3) + (lane & 3);
      }
      glds16(sb + pix * 128 + piece * 16, lds0 + (st % NST) * STAGE + j * 1024);
    }
  };
  unsigned acc = 0;
  for (int st = 0; st < NST - 1 && st < nstage; ++st) issue(st);
  for (int st = 0; st < nstage; ++st) {
    if (st + NST - 1 < nstage) {
      issue(st + NST - 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NST - 1)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    acc += reinterpret_cast<const unsigned*>(lds + (st % NST) * STAGE)[threadIdx.x * 8];
    __syncthreads();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MAP, int NST>
static double run(const char* buf, size_t total, int cus, unsigned* sink) {
  const long long per = (long long)(total / cus / STAGE) * STAGE;
  const int passes = (int)(((size_t)1 << 30) / total);    // about 1 GB of traffic per launch
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<MAP, NST>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream<MAP, NST>), dim3(cus), dim3(512), NST * STAGE, 0, buf, per, passes, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MAP, NST>), dim3(cus), dim3(512), NST * STAGE, 0, buf, per, passes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)per * cus * passes * reps / (ms * 1e-3) / 1e12;       // TB/s
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned* sink;
  hipMalloc(&sink, 256);
  // 1 GB: HBM; 96 MB: Infinity Cache; 16 MB: the L2s (each workgroup re-reads its 64 KB slice)
  for (size_t total : {(size_t)1 << 30, (size_t)96 << 20, (size_t)16 << 20}) {
    char* buf;
    hipMalloc(&buf, total);
    hipMemset(buf, 1, total);
    const double a = run<0, 4>(buf, total, cus, sink), b = run<1, 4>(buf, total, cus, sink),
                 c = run<2, 4>(buf, total, cus, sink), d = run<0, 2>(buf, total, cus, sink),
                 e = run<0, 8>(buf, total, cus, sink);
    printf("%4zu MB, 48 KB in flight per CU: whole lines %.2f TB/s (%.1f B/clk/CU at 2.1 GHz), half lines with the "
           "halves adjacent %.2f, apart %.2f;  whole lines with 16 KB in flight %.2f, with 112 KB %.2f\n",
           total >> 20, a, a * 1e12 / cus / 2.1e9, b, c, d, e);
    hipFree(buf);
  }
  return 0;
}
