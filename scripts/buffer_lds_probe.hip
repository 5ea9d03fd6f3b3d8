// What `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) does on gfx950,
// measured before wrw64_kernel relies on it (round 5):
//   1. rows outside the resource (voffset >= num_records, or "negative" = wrapped) arrive as zeros;
//   2. the LDS destination (M0) may lie above 64 KB (the kernel's two staging buffers span 156 KB);
//   3. is an SGPR offset (soffset) part of the bounds check?  (GCN documents it as excluded.)
// hipcc --offload-arch=gfx950 -O3 scripts/buffer_lds_probe.hip -o /tmp/blp && /tmp/blp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void blds16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte)
      : "memory");
}

// mode 0: voffset = base_off + 16 * lane; mode 1: the same through soffset (voffset = 16 * lane)
__global__ void probe(const unsigned* src, unsigned num_bytes, int base_off, int mode, unsigned lds_dst,
                      unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  for (unsigned i = threadIdx.x; i < 40 * 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned long long a = (unsigned long long)src;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r.z = __builtin_amdgcn_readfirstlane(num_bytes);
  r.w = 0x00020000u;
  const unsigned ldsb = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned*)lds + lds_dst);
  if (mode == 0)
    blds16(r, (unsigned)(base_off + 16 * (int)threadIdx.x), 0u, ldsb);
  else
    blds16(r, 16u * threadIdx.x, (unsigned)__builtin_amdgcn_readfirstlane(base_off), ldsb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int k = 0; k < 4; ++k) out[4 * threadIdx.x + k] = lds[lds_dst / 4 + 4 * threadIdx.x + k];
}

int main() {
  const int n = 4096;                                   // dwords in the resource: 16 KB
  std::vector<unsigned> h(2 * n);
  for (int i = 0; i < 2 * n; ++i) h[i] = 0x1000000u + i;
  unsigned *d, *o;
  hipMalloc(&d, 2 * n * 4);
  hipMalloc(&o, 256 * 4);
  hipMemcpy(d, h.data(), 2 * n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  std::vector<unsigned> r(256);
  struct Case { const char* what; int off; int mode; unsigned lds; };
  // the resource covers the SECOND half of the allocation: "negative" offsets are mapped memory
  const Case cases[] = {
      {"inside, LDS 0", 256, 0, 0},
      {"inside, LDS at 100 KB", 256, 0, 100 * 1024},
      {"inside, LDS at 150 KB", 256, 0, 150 * 1024},
      {"straddles the end (last 32 lanes outside)", n * 4 - 512, 0, 0},
      {"entirely past the end", n * 4 + 1024, 0, 0},
      {"negative offset (wraps): first 16 lanes before the start", -256, 0, 0},
      {"soffset: straddles the end", n * 4 - 512, 1, 0},
      {"soffset: entirely past the end", n * 4 + 1024, 1, 0},
  };
  int bad = 0;
  for (const Case& c : cases) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d + n, (unsigned)(n * 4), c.off, c.mode, c.lds, o);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_out = 0, other = 0, n_in = 0, n_out = 0;
    for (int l = 0; l < 64; ++l)
      for (int k = 0; k < 4; ++k) {
        const long long byte = (long long)c.off + 16 * l + 4 * k;
        const bool inside = byte >= 0 && byte + 4 <= (long long)n * 4;
        const unsigned v = r[4 * l + k];
        if (inside) { ++n_in; if (v == 0x1000000u + n + (unsigned)(byte / 4)) ++ok_in; else ++other; }
        else { ++n_out; if (v == 0u) ++zero_out; else ++other; }
      }
    printf("%-58s %s  inside %d/%d right, outside %d/%d zero, other %d  (first words %08x %08x, last %08x)\n",
           c.what, hipGetErrorString(e), ok_in, n_in, zero_out, n_out, other, r[0], r[1], r[255]);
    if (c.mode == 0 && other) ++bad;
  }
  printf(bad ? "PROBE: buffer-LDS path NOT usable as assumed\n" : "PROBE: voffset bounds check + high LDS destinations OK\n");
  return bad ? 1 : 0;
}
