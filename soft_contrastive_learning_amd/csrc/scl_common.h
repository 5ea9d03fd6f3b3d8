// Shared device helpers for the gfx950 kernels.  Wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "scl_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

#define SCL_WAVE 64

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 (a k-ordered
// fmaf chain).  Lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].
// Accumulator register r of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int half) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ float bf16_to_f32(unsigned short v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// round-to-nearest-even f32 -> bf16 (a plain cast keeps NaN a NaN on gfx950)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// two values at once: one v_cvt_pk_bf16_f32 instead of two conversions + a shift + an or
// (a in the low half); the same rounding
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// max of both bf16 halves of a word with `floor16` AS 16-BIT INTEGERS (one v_pk_max_i16): with
// floor16 = 0 that is ReLU on the rounded pair — a bf16 is negative exactly when it is as an
// integer, rounding keeps the sign, so max(round(v), 0) = round(max(v, 0)), -0 -> +0 (a NaN with
// the sign bit clear stays) — and with the least int16 the identity
__device__ __forceinline__ unsigned max2_i16(unsigned w, short floor16) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  const s16x2_t a = __builtin_bit_cast(s16x2_t, w), z = {floor16, floor16};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, z));
}

// Convolution weights come as bf16 or as the float32 master copy (SCL_W_F32 in the flags
// argument: rounded to bf16 on the way into the packed image, so no separate cast pass);
// weight gradients go out in the same type.
__device__ __forceinline__ unsigned short weight_bf16(const void* w, int64_t i, int f32) {
  return f32 ? f32_to_bf16(static_cast<const float*>(w)[i])
             : static_cast<const unsigned short*>(w)[i];
}
__device__ __forceinline__ void store_weight_grad(void* gw, int64_t i, float v, int f32) {
  if (f32)
    static_cast<float*>(gw)[i] = v;
  else
    static_cast<unsigned short*>(gw)[i] = f32_to_bf16(v);
}

// ReLU' on packed bf16: each 16-bit half of g survives where the matching half of y is > 0
// (sign clear and not zero) and becomes +0 elsewhere — three packed 16-bit integer ops per
// word (v_pk_max_i16, v_pk_min_i16, v_pk_mul_lo_u16).
__device__ __forceinline__ unsigned relu_mask_word(unsigned g, unsigned y) {
  unsigned keep, out;
  asm("v_pk_max_i16 %0, %1, 0" : "=v"(keep) : "v"(y));            // negative halves -> 0
  asm("v_pk_min_i16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(keep) : "v"(keep));   // positive -> 1
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(out) : "v"(g), "v"(keep));
  return out;
}
template <typename V4>   // four packed words (the translation units' own u32x4)
__device__ __forceinline__ V4 relu_mask(V4 g, V4 y) {
  return V4{relu_mask_word(g.x, y.x), relu_mask_word(g.y, y.y), relu_mask_word(g.z, y.z),
               relu_mask_word(g.w, y.w)};
}

// element loads for the two feature-map dtypes
template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ f32x4 ld4(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
  }
  static __device__ __forceinline__ f32x2 ld2(const float* p) {
    return *reinterpret_cast<const f32x2*>(p);
  }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ f32x4 ld4(const unsigned short* p) {
    u16x4 v = *reinterpret_cast<const u16x4*>(p);
    f32x4 r;
    r[0] = bf16_to_f32(v[0]);
    r[1] = bf16_to_f32(v[1]);
    r[2] = bf16_to_f32(v[2]);
    r[3] = bf16_to_f32(v[3]);
    return r;
  }
  static __device__ __forceinline__ f32x2 ld2(const unsigned short* p) {
    const unsigned v = *reinterpret_cast<const unsigned*>(p);
    f32x2 r;
    r[0] = __uint_as_float(v << 16);
    r[1] = __uint_as_float(v & 0xffff0000u);
    return r;
  }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

// Un-pooling of a 2x2 max-pooling gradient in registers.  g = 8 bf16 channels of a POOLED
// gradient pixel (four words), idx = their 8 window-position bytes (2 dy + dx, values 0..3).
// UnpoolFlags puts, for each position, a flag into the top bit of every index byte that equals
// it; unpool8(g, f, pos) is the 16 bytes of the full-size gradient at window position pos: the
// channels routed there, zero elsewhere.  v_perm_b32's selectors 8..11 replicate bit 15 / 31 of
// either source word over a byte — (flags << 8, flags) therefore expands to a 16-bit mask per
// channel with one instruction per output word.
struct UnpoolFlags {
  unsigned e[2][4];        // [index word][position]
};
__device__ __forceinline__ UnpoolFlags unpool_flags(unsigned i0, unsigned i1) {
  UnpoolFlags f;
#pragma unroll
  for (int hw = 0; hw < 2; ++hw) {
    const unsigned iw = hw ? i1 : i0;
    const unsigned b0 = (iw << 7) & 0x80808080u, b1 = (iw << 6) & 0x80808080u;
    f.e[hw][3] = b0 & b1;
    f.e[hw][2] = b1 ^ f.e[hw][3];
    f.e[hw][1] = b0 ^ f.e[hw][3];
    f.e[hw][0] = (b0 | b1) ^ 0x80808080u;
  }
  return f;
}
template <typename V4>
__device__ __forceinline__ V4 unpool8(V4 g, const UnpoolFlags& f, int pos) {
  V4 o;
#pragma unroll
  for (int hw = 0; hw < 2; ++hw) {
    const unsigned e = f.e[hw][pos], eh = e << 8;
    o[2 * hw] = g[2 * hw] & __builtin_amdgcn_perm(eh, e, 0x08080a0au);       // channels 0, 1
    o[2 * hw + 1] = g[2 * hw + 1] & __builtin_amdgcn_perm(eh, e, 0x09090b0bu);   // channels 2, 3
  }
  return o;
}

// Raw (unconverted) vector loads: a prefetch must leave its destination registers untouched
// until the data is consumed — converting bf16 at load time makes the compiler wait for the
// load right after issuing it.  ldN returns the memory image, cvtN widens it to f32.
template <typename T>
struct Raw;
template <>
struct Raw<float> {
  typedef f32x4 v4;
  typedef f32x2 v2;
  static __device__ __forceinline__ v4 ld4(const float* p) { return *reinterpret_cast<const v4*>(p); }
  static __device__ __forceinline__ v2 ld2(const float* p) { return *reinterpret_cast<const v2*>(p); }
  static __device__ __forceinline__ f32x4 cvt4(v4 r) { return r; }
  static __device__ __forceinline__ f32x2 cvt2(v2 r) { return r; }
  static __device__ __forceinline__ v4 ones4() { return v4{1.f, 1.f, 1.f, 1.f}; }
};
template <>
struct Raw<unsigned short> {
  typedef u16x4 v4;
  typedef unsigned v2;
  static __device__ __forceinline__ v4 ld4(const unsigned short* p) {
    return *reinterpret_cast<const v4*>(p);
  }
  static __device__ __forceinline__ v2 ld2(const unsigned short* p) {
    return *reinterpret_cast<const v2*>(p);
  }
  static __device__ __forceinline__ f32x4 cvt4(v4 r) {
    return f32x4{bf16_to_f32(r[0]), bf16_to_f32(r[1]), bf16_to_f32(r[2]), bf16_to_f32(r[3])};
  }
  static __device__ __forceinline__ f32x2 cvt2(v2 r) {
    return f32x2{__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u)};
  }
  static __device__ __forceinline__ v4 ones4() { return v4{0x3f80, 0x3f80, 0x3f80, 0x3f80}; }
};

// butterfly reductions inside one 32-lane half of the wave (xor 1..16)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int m = 1; m < 32; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}

// block-wide reductions through a caller-provided LDS scratch of >= 32 floats
template <int OP>  // 0 sum, 1 max, 2 min
__device__ __forceinline__ float block_reduce(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = OP == 0 ? wave_sum(v) : (OP == 1 ? wave_max(v) : wave_min(v));
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) {
    float o = scratch[i];
    r = OP == 0 ? r + o : (OP == 1 ? fmaxf(r, o) : fminf(r, o));
  }
  return r;
}

// ---- optional per-kernel timing (bench.py's live roofline measurement) ----------------
// A process-wide sink set by scl_prof_begin(); when present every launch is bracketed by
// HIP events on the launch stream.  No sink (the normal case) = plain launches.
struct SclProfSink {
  int capacity;
  int count;              // claimed slots (atomic)
  hipEvent_t* ev;         // 2 * capacity
  const char** name;      // capacity
};
// Process-wide (PyTorch runs backward on its own autograd thread, so a thread-local sink
// would miss every backward kernel); slots are claimed atomically.
extern SclProfSink* volatile scl_prof_sink;

#define SCL_LAUNCH(kname, kernel, grid, block, lds, st, ...)                     \
  do {                                                                           \
    SclProfSink* ps_ = scl_prof_sink;                                            \
    int slot_ = -1;                                                              \
    if (ps_) {                                                                   \
      slot_ = __atomic_fetch_add(&ps_->count, 1, __ATOMIC_RELAXED);              \
      if (slot_ >= ps_->capacity) slot_ = -1;                                    \
    }                                                                            \
    if (slot_ >= 0) {                                                            \
      ps_->name[slot_] = (kname);                                                \
      (void)hipEventRecord(ps_->ev[2 * slot_], (st));                            \
    }                                                                            \
    hipLaunchKernelGGL(kernel, grid, block, lds, (st), __VA_ARGS__);             \
    if (slot_ >= 0) (void)hipEventRecord(ps_->ev[2 * slot_ + 1], (st));          \
  } while (0)

// Diagnostic kernel variant selector.  It exists in the DIAGNOSTIC build only (-DSCL_DIAG ->
// libscl_hip_diag.so: the A/B, ablation and clock-stamp variants that scripts/ and the equality
// tests select with scl_debug_set_variant).  The product library (libscl_hip.so) is compiled
// without it: scl_variant() is the constant 0 there, every variant branch of the dispatch code
// and every `dbg` test inside a kernel folds away, and scl_debug_set_variant rejects anything
// but 0 — the shipped library has no process-wide switch that changes a result.
#ifdef SCL_DIAG
extern volatile int scl_debug_variant;
static inline int scl_variant() { return scl_debug_variant; }
#define SCL_DIAG_ONLY(x) (x)
#else
static constexpr int scl_variant() { return 0; }
#define SCL_DIAG_ONLY(x) 0
#endif

// 3x3 convolution with LDS-resident weights on v_mfma_f32_16x16x32_bf16 (convh.hip); arguments
// already validated by convg_dispatch (convg.hip)
#define SCL_CONVH_DEFAULT true
int scl_convh_dispatch(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                       int64_t w_stride_h, int64_t w_stride_w, int flags, int B, int H, int W,
                       int cin, int kout, void* out, const float* bias, int relu, const void* mask,
                       void* pidx, void* workspace, int dv, void* stream);

// CUs the persistent convolution grids leave free (SCL_RESERVE_CUS, default 0; or
// scl_set_reserve_cus).  With more than one rank the RCCL kernels need somewhere to run while
// those grids hold every CU; the knob exists so that this can be measured
// on a multi-GPU node without a rebuild — bench.py times a few steps each way in-process.
extern volatile int scl_reserve_cus;   // -1 until first read from the environment
static inline int scl_usable_cus(int cus) {
  int reserve = scl_reserve_cus;
  if (reserve < 0) {
    const char* e = getenv("SCL_RESERVE_CUS");
    reserve = e ? atoi(e) : 0;
    if (reserve < 0) reserve = 0;
    scl_reserve_cus = reserve;
  }
  const int left = cus - reserve;
  return left >= 8 ? left : (cus < 8 ? cus : 8);
}

// CU count of the calling thread's CURRENT device, cached per device ordinal (round 4 cached the
// first device's count in one unsynchronised static per file: wrong for two devices in one
// process).  Relaxed atomics: the value is idempotent, a race stores the same number twice.
static inline int scl_device_cus() {
  static std::atomic<int> table[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = table[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    int c = 0;
    n = (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0) ? c : 256;
    table[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// One-time function attributes (dynamic LDS limits) are set once PER DEVICE: a function object
// belongs to a device, and one process may drive several.
struct SclDeviceOnce { std::once_flag f[64]; };
template <class F> static inline void scl_call_once(SclDeviceOnce& o, F&& fn) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0;
  std::call_once(o.f[d], fn);
}

static inline int scl_launch_status() { return (int)hipGetLastError(); }
static inline bool scl_aligned256(const void* p) { return (((uintptr_t)p) & 255u) == 0; }
static inline size_t scl_round256(size_t n) { return (n + 255u) & ~(size_t)255u; }
