// How fast can 24 x 480 x 640 pixels of 128 bytes (944 MiB, conv1_1's output map) be WRITTEN, by pattern?
//   0  linear: workgroup-contiguous 16-byte stores (what a fill does)
//   1  conv_first_kernel's pattern: tiles of 8 rows x 32 pixels handed out round-robin to 8 workgroups per CU,
//      wave w writes tile rows 2 w, 2 w + 1, a store instruction = 8 pixels x 128 bytes (1 KB contiguous)
//   2  the same tiles, but 64 pixels wide and 4 rows high (a store instruction still 1 KB, rows 8 KB contiguous)
//   3  pattern 1 with the tiles of a workgroup CONSECUTIVE (tile = blockIdx.x * per + k) instead of strided
// hipcc --offload-arch=gfx950 -O3 scripts/store_pattern_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(u32x4* __restrict__ y, int B, int H, int W) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const u32x4 v = {(unsigned)threadIdx.x, blockIdx.x, 3u, 4u};
  if (MODE == 0) {
    const long n = (long)B * H * W * 8;                       // 16-byte pieces
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (long i = lo + threadIdx.x; i < hi; i += 256) y[i] = v;
    return;
  }
  const int TW = MODE == 2 ? 64 : 32, TH = MODE == 2 ? 4 : 8;
  const int tiles_x = W / TW, tiles_y = H / TH, per_img = tiles_x * tiles_y, ntiles = B * per_img;
  const int per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
  for (int k = 0; k < per_wg; ++k) {
    const int tile = MODE == 3 ? blockIdx.x * per_wg + k : blockIdx.x + k * gridDim.x;
    if (tile >= ntiles) break;
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;
    const int rows_per_wave = TH / 4;
    for (int mt = 0; mt < rows_per_wave; ++mt) {
      const int oy = ty + rows_per_wave * wid + mt;
      for (int kk = 0; kk < TW / 8; ++kk) {
        const int ox = tx + 8 * kk + (lane >> 3);
        y[(((long)b * H + oy) * W + ox) * 8 + (lane & 7)] = v;
      }
    }
  }
}

template <int MODE>
float run(u32x4* y, int grid) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, y, 24, 480, 640);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  u32x4* y;
  const size_t bytes = (size_t)24 * 480 * 640 * 128;
  hipMalloc(&y, bytes);
  for (int grid : {2048, 8192}) {
    printf("grid %4d: linear %.1f us | conv_first pattern %.1f us | 4 x 64 tiles %.1f us | consecutive tiles per workgroup %.1f us   (%.0f MB)\n",
           grid, run<0>(y, grid), run<1>(y, grid), run<2>(y, grid), run<3>(y, grid), bytes / 1e6);
  }
  return 0;
}
