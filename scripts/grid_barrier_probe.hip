// What a grid-wide barrier among 256 co-resident workgroups costs on gfx950, by form (round 6, before
// the one-launch pairwise-loss forward relies on one):
//   0  one counter: agent-scope atomic add by one lane per workgroup, dword sc1 poll of it
//   1  one FLAG PER WORKGROUP: an sc1 dword store of the phase number, wave 0 polls all flags
//      (lane l reads flags 4 l .. 4 l + 3 with one 16-byte sc1 load: 1 KB per poll)
//   2  eight counters on eight 128-byte lines (shard = blockIdx & 7), 32 adds each; poll = 8 loads
//   3  no barrier at all (the loop overhead)
// Each kernel runs NB barriers back to back; time per barrier = (kernel time - form 3) / NB.
// hipcc --offload-arch=gfx950 -O3 scripts/grid_barrier_probe.hip -o /tmp/gbp && /tmp/gbp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) unsigned* u32_gptr;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld_sc1_x4(const unsigned* base, unsigned bytes, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(base), 0, bytes, 0x00020000);
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16));
}

template <int FORM>
__global__ __launch_bounds__(256) void probe(unsigned* sync, int nb, int sleep, unsigned* out) {
  __shared__ int flag;
  const unsigned G = gridDim.x;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int k = 1; k <= nb; ++k) {
    if (FORM == 3) {
      __syncthreads();
      continue;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (FORM == 0) {
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load((u32_gptr)sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)k * G)
          __builtin_amdgcn_s_sleep(1);
      }
    } else if (FORM == 1) {
      if (threadIdx.x < 64) {
        if (threadIdx.x == 0)
          __hip_atomic_store((u32_gptr)(sync + blockIdx.x), (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned lane = threadIdx.x;
        for (;;) {
          const unsigned idx = 4 * lane < G ? 4 * lane : 0;
          const u32x4 v = ld_sc1_x4(sync, G * 4, idx * 4);
          bool ok = true;
          for (int c = 0; c < 4; ++c)
            if (4 * lane + c < G) ok = ok && v[c] >= (unsigned)k;
          if (__all(ok)) break;
          if (sleep) __builtin_amdgcn_s_sleep(1);
        }
      }
    } else if (FORM == 2) {
      if (threadIdx.x < 64) {
        if (threadIdx.x == 0)
          __hip_atomic_fetch_add(sync + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned lane = threadIdx.x;
        for (;;) {
          unsigned v = (unsigned)k * G;
          if (lane < 8) v = __hip_atomic_load((u32_gptr)(sync + 32 * lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // shard s receives ceil / floor of G / 8 arrivals per barrier
          const unsigned want = lane < 8 ? (unsigned)k * ((G + 7 - lane) / 8) : 0;
          if (__all(v >= want)) break;
          if (sleep) __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(__builtin_readcyclecounter() - t0);
  (void)flag;
}

template <int FORM>
float run(unsigned* sync, int grid, int nb, int sleep, unsigned* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(sync, 0, 4096);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<FORM>, dim3(grid), dim3(256), 0, 0, sync, nb, sleep, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  unsigned *sync, *out;
  hipMalloc(&sync, 4096);
  hipMalloc(&out, 64);
  const int nb = 50;
  for (int grid : {64, 256}) {
    const float base = run<3>(sync, grid, nb, 1, out);
    const float f0 = run<0>(sync, grid, nb, 1, out);
    const float f1 = run<1>(sync, grid, nb, 1, out);
    const float f1n = run<1>(sync, grid, nb, 0, out);
    const float f2 = run<2>(sync, grid, nb, 1, out);
    const float f2n = run<2>(sync, grid, nb, 0, out);
    printf("grid %3d: loop only %.2f us | per barrier: one counter %.2f us, flag per workgroup %.2f (no sleep %.2f), "
           "8 sharded counters %.2f (no sleep %.2f)\n",
           grid, base, (f0 - base) / nb, (f1 - base) / nb, (f1n - base) / nb, (f2 - base) / nb, (f2n - base) / nb);
  }
  return 0;
}
