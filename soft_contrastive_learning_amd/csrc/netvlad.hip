// NetVLAD head on gfx950: channel L2 norm + soft-assignment + residual aggregation +
// intra / global normalisation, forward and backward.
//
// Reference semantics: model/nets.py:66-67 = tf.nn.l2_normalize(x, axis=-1) followed
// by netvlad_tf.layers.netVLAD(x, 64) (external, restated in oracle/netvlad_np.py).
// The TF graph materialises the 5-D tensor [B,H',W',D,K] (157 MB per 640x480 image)
// in forward and again in autodiff; here nothing larger than [B,N,64] is ever written.
//
// Every contraction runs on the f32-input matrix cores (v_mfma_f32_16x16x4_f32 for the
// location-tile kernels, v_mfma_f32_32x32x2_f32 for the aggregation): exact f32 (bitwise a
// k-ordered fmaf chain), so the governing roofline is the 157.3 TF f32 MFMA peak, not HBM
// (SURVEY.md H1).  Linearity is used to keep x raw:
//   s[n,k] = rn[n] * sum_d x[n,d] W[d,k],      rn[n] = rsqrt(max(sum_d x^2, 1e-12))
//   V[d,k] = sum_n (a[n,k] rn[n]) x[n,d] + C[d,k] * sum_n a[n,k]
//
// Forward kernels
//   transpose_w_kernel         W[512,64] -> Wt[64,512] (source of the LDS operand image)
//   rowtile16_kernel<ASSIGN>   per 16-location tile: x.W against a double-buffered LDS image
//                              of Wt, row norms from the same x loads, softmax over K by
//                              16-lane butterflies -> a, rn (+ logits for training)
//   aggregate_kernel           per (image, 64-channel tile, location split): x^T.(a rn),
//                              4 waves along n, LDS tree reduce -> partial slabs
//   finish_sum/norm_kernel     slabs + C*asum, intra-norm over D, global norm -> out
// Backward kernels
//   bwd_dots/bwd_du_kernel     grad through both norms (closed form from four column dots)
//                              -> dU (both layouts, + bf16 planes), c.dU
//   rowtile16_kernel<DASSIGN>  x.dU[b] -> d a -> softmax backward -> ds, <dxhat,xhat>
//   aggregate_kernel           x^T.(ds rn) -> per-image dW slabs
//   dx16_kernel                [a | ds].[dU | W]^T and the l2-norm Jacobian -> grad_x
//   wgrad_finish_kernel        sums over the batch -> grad_w, grad_c
#include <mutex>
#include <type_traits>

#include "scl_common.h"
#include "vlad_planes.h"

namespace {

constexpr int D = SCL_VLAD_D;   // 512
constexpr int K = SCL_VLAD_K;   // 64
constexpr int NSPLIT = 4;       // location splits with their own slab in aggregate_kernel
constexpr int VROWS = SCL_VLAD_SAVE_ROWS;   // rows of save_vlad per image: 512 of U, asum, sync words

// ------------------------------------------------------------------ small kernels
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ w,
                                                          float* __restrict__ wt) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over D*K, k fastest
  if (idx < D * K) wt[(idx % K) * D + idx / K] = w[idx];
}

enum RowMode { ASSIGN = 0, DASSIGN = 1 };

struct RowTileArgs {
  const void* x;       // [B,N,512]
  const float* bt;     // ASSIGN: Wt [64][512];  DASSIGN: dUt [B][64][512]
  int64_t bt_stride;   // floats between images (0 for ASSIGN)
  int B, N, pre_l2;
  // ASSIGN outputs
  float* assign;       // [B,N,64]
  float* logit;        // [B,N,64] or NULL
  float* rnorm;        // [B,N]
  // DASSIGN inputs / outputs
  const float* a_in;     // [B,N,64]
  const float* logit_in; // [B,N,64]
  const float* rn_in;    // [B,N]
  const float* cdu;      // [B,64]
  float* ds;             // [B,N,64]
  float* rowdot;         // [B,N]
};

// ---------------------------------------------------------------------------------------
// rowtile16_kernel: [16 locations] x [512 channels] x [64 clusters] per wave on
// v_mfma_f32_16x16x4_f32.
//   * 1800 tiles at 24 x 1200 locations -> two resident workgroups per CU, so one wave's
//     staging / softmax epilogue can overlap another's MFMAs (a 32-location tile leaves one
//     900-tile wave per SIMD and exposes both);
//   * the [64][512] operand is staged in four 128-channel chunks, double-buffered: chunk
//     c+1 is loaded to registers before chunk c is contracted and written to LDS after it,
//     so only the first 32 KB fill is exposed (a full 128 KB image costs >= 5 us per CU);
//   * LDS chunk image = 4 planes (one per k-group g of the MFMA) x 64 rows x 36 floats:
//     lane (i, g) reads plane g, row 16 kt + i with ds_read_b128; the plane stride is a
//     multiple of 64 floats and the row stride 36 floats = 9 slots, so the 16 lanes of every
//     ds_read_b128 service group hit 16 different 16-byte slots (conflict-free).
// Accumulator register j of lane l is S[row = 4 (l >> 4) + j][cluster = 16 kt + (l & 15)].
constexpr int RT_CH = 128;                      // channels per staged chunk
constexpr int RT_LD = 36;                       // floats per plane row
constexpr int RT_PLANE = K * RT_LD;             // 2304 floats (= 36 x 64)
constexpr int RT_CHUNK = 4 * RT_PLANE;          // 9216 floats per buffer
constexpr size_t kRowTile16Lds = 2 * (size_t)RT_CHUNK * sizeof(float);   // 73,728 B

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float q16_sum(float v) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float q16_max(float v) {
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// Shared epilogue of the row-tile kernels: acc[kt][j] = raw contraction of row 4 g + j with
// cluster 16 kt + i; ss = this lane's partial sum of squares of row i (ASSIGN only).
template <int MODE>
__device__ __forceinline__ void rowtile_epilogue(const RowTileArgs& p, f32x4 (&acc)[4], float ss,
                                                 float* bt_lds, int b, int n0) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int n = n0 + i;
  const bool row_ok = n < p.N;
  // Epilogue I/O goes through a per-wave [16][68] LDS scratch (the operand buffers are free:
  // every wave passed the last barrier) so each lane moves 16 bytes and 4 lanes cover one
  // location's 64 clusters contiguously; accumulator-layout accesses would be 4 bytes per
  // lane in 64-byte pieces (measured: 4.6 us of a 30 us kernel).
  //   accumulator layout: value (row 4g+j, cluster 16kt+i);  row layout: lane -> row lane>>2,
  //   cluster groups 4(4m + (lane&3)) .. +3 for m = 0..3
  float* scr = bt_lds + wid * (16 * 68);
  const int row_e = lane >> 2, seg = lane & 3;
  const bool ok_e = n0 + row_e < p.N;
  const int64_t o_e = ((int64_t)b * p.N + (ok_e ? n0 + row_e : 0)) * K + 4 * seg;
  auto put_acc_layout = [&](const float (&v)[4][4]) {      // v[j][kt]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) scr[(4 * g + j) * 68 + 16 * kt + i] = v[j][kt];
  };
  auto get_acc_layout = [&](float (&v)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) v[j][kt] = scr[(4 * g + j) * 68 + 16 * kt + i];
  };
  auto store_rows = [&](float* dst) {                      // scratch -> global, 16 B per lane
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&scr[row_e * 68 + 16 * m + 4 * seg]);
      if (ok_e) *reinterpret_cast<f32x4*>(dst + o_e + 16 * m) = v;
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto load_rows = [&](const float* src_rows) {            // global -> scratch, 16 B per lane
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src_rows + o_e + 16 * m);
      *reinterpret_cast<f32x4*>(&scr[row_e * 68 + 16 * m + 4 * seg]) =
          ok_e ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_wave_barrier();
  };

  if (MODE == ASSIGN) {
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    const float rn = p.pre_l2 ? 1.0f / sqrtf(fmaxf(ss, 1e-12f)) : 1.0f;
    if (g == 0 && row_ok) p.rnorm[(int64_t)b * p.N + n] = rn;
    float av[4][4], sv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float rnr = __shfl(rn, 4 * g + j, 64);
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        sv[j][kt] = acc[kt][j] * rnr;
        m = fmaxf(m, sv[j][kt]);
      }
      m = q16_max(m);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        av[j][kt] = expf(sv[j][kt] - m);
        sum += av[j][kt];
      }
      const float inv = 1.0f / q16_sum(sum);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) av[j][kt] *= inv;
    }
    put_acc_layout(av);
    store_rows(p.assign);
    if (p.logit) {
      put_acc_layout(sv);
      store_rows(p.logit);
    }
  } else {
    float cd[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) cd[kt] = p.cdu[b * K + 16 * kt + i];
    float a[4][4], lg[4][4], ds[4][4];
    load_rows(p.a_in);
    get_acc_layout(a);
    load_rows(p.logit_in);
    get_acc_layout(lg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 4 * g + j;
      const bool ok = n0 + row < p.N;
      const int64_t gr = (int64_t)b * p.N + (ok ? n0 + row : 0);
      const float rnr = p.rn_in[gr];
      float t[4];
      float dot = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        t[kt] = acc[kt][j] * rnr;                         // xhat · dU
        dot += a[j][kt] * (t[kt] + cd[kt]);               // + c · dU
      }
      dot = q16_sum(dot);
      float rd = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        ds[j][kt] = a[j][kt] * ((t[kt] + cd[kt]) - dot);
        // <d xhat[n,:], xhat[n,:]> = sum_k a (xhat·dU) + ds (xhat·W)
        rd += a[j][kt] * t[kt] + ds[j][kt] * lg[j][kt];
      }
      rd = q16_sum(rd);
      if (ok && i == 0) p.rowdot[gr] = rd;
    }
    put_acc_layout(ds);
    store_rows(p.ds);
  }
}

// grid (ceil(ceil(N/16) / 4), B); block 256: wave w owns 16-location tile 4 * blockIdx.x + w.
// VAR (diagnostic builds only, see scl_debug_set_variant): bit 0 = no epilogue stores,
// bit 1 = no x loads (constant operand), bit 2 = no operand staging and no barriers.
template <typename T, int MODE, int VAR = 0>
__global__ __launch_bounds__(256, 2) void rowtile16_kernel(RowTileArgs p) {
  extern __shared__ __attribute__((aligned(16))) float bt_lds[];  // [2][4][64][RT_LD]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x * 4 + wid) * 16;
  const bool active = n0 < p.N;          // wave-uniform; idle waves still stage and sync
  const int n = n0 + i;
  const bool row_ok = n < p.N;
  const float* src = p.bt + (int64_t)b * p.bt_stride;
  const T* xrow = reinterpret_cast<const T*>(p.x) + ((int64_t)b * p.N + (row_ok ? n : 0)) * D + 4 * g;

  f32x4 st[8];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int idx = v * 256 + threadIdx.x;
      // 8 consecutive lanes take the 8 groups of one plane (conflict-free ds_write_b128);
      // a 32-lane run still covers one whole 512-byte row segment
      const int k = idx >> 5, c4 = 4 * (idx & 7) + ((idx >> 3) & 3);
      st[v] = *reinterpret_cast<const f32x4*>(src + k * D + chunk * RT_CH + c4 * 4);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int idx = v * 256 + threadIdx.x;
      // 8 consecutive lanes take the 8 groups of one plane (conflict-free ds_write_b128);
      // a 32-lane run still covers one whole 512-byte row segment
      const int k = idx >> 5, c4 = 4 * (idx & 7) + ((idx >> 3) & 3);
      // logical 4-channel group c4 = 4 t + g  ->  plane g, row k, column 4 t
      *reinterpret_cast<f32x4*>(&bt_lds[buf * RT_CHUNK + (c4 & 3) * RT_PLANE + k * RT_LD +
                                        4 * (c4 >> 2)]) = st[v];
    }
  };
  typename Raw<T>::v4 xc[8], xn[8];   // raw prefetch registers: converted when consumed
  auto x_load = [&](int chunk, typename Raw<T>::v4* dst) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (VAR & 2)
        dst[t] = Raw<T>::ones4();
      else
        dst[t] = Raw<T>::ld4(xrow + chunk * RT_CH + 16 * t);
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ss = 0.f;

  if (!(VAR & 4)) stage_load(0);
  if (active) x_load(0, xc);
  if (!(VAR & 4)) {
    stage_store(0);
    __syncthreads();
  }
#pragma unroll 1
  for (int c = 0; c < D / RT_CH; ++c) {
    const bool more = c + 1 < D / RT_CH;
    if (more) {
      if (!(VAR & 4)) stage_load(c + 1);
      if (active) x_load(c + 1, xn);
    }
    if (active) {
      const float* wb = &bt_lds[(c & 1) * RT_CHUNK + g * RT_PLANE + i * RT_LD];
      // B fragments are double-buffered in registers: the ds_read_b128 of step t+1 are in
      // flight under the 16 MFMAs of step t (with one register set hipcc issues them only
      // after the last MFMA and exposes the LDS latency 32 times per tile)
      f32x4 wv[2][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        wv[0][kt] = *reinterpret_cast<const f32x4*>(wb + kt * 16 * RT_LD);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f32x4 xa = Raw<T>::cvt4(xc[t]);
        if (!row_ok) xa = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t + 1 < 8) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            wv[(t + 1) & 1][kt] =
                *reinterpret_cast<const f32x4*>(wb + kt * 16 * RT_LD + 4 * (t + 1));
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this step's MFMAs
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) acc[kt] = mfma16(xa[cc], wv[t & 1][kt][cc], acc[kt]);
          ss = fmaf(xa[cc], xa[cc], ss);
        }
      }
    }
    if (!(VAR & 4)) {
      if (more) stage_store((c + 1) & 1);
      __syncthreads();
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < 8; ++t) xc[t] = xn[t];
    }
  }
  if (!active) return;
  if (VAR & 1) {   // keep the accumulators alive without the epilogue
    float s = ss;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) s += acc[kt][0] + acc[kt][1] + acc[kt][2] + acc[kt][3];
    if (s == 1.2345e-7f) p.rnorm[0] = s;
    return;
  }

  rowtile_epilogue<MODE>(p, acc, ss, bt_lds, b, n0);
}

// ---------------------------------------------------------------------------------------
// bf16 feature maps run on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).  x is bf16
// already; the float32 operand (W^T, or dU^T of the image) arrives split into three bf16 planes
// o = o1 + o2 + o3 (24 mantissa bits), so every product x * o_p is exact in float32 and
// x.o = x.o1 + x.o2 + x.o3 differs from the float32 contraction only by accumulation order —
// at 3 x 16 cycles per 16x16x32 step instead of 8 x 32 cycles of 16x16x4 float32 steps, and
// with no bf16 -> f32 conversion of x at all.  Row norms by float32 FMAs on the same registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16b(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// LDS-DMA: 64 lanes x 16 bytes from per-lane global addresses into 1 KB of consecutive LDS at
// the wave-uniform byte address lds_byte.  Inline asm, so hipcc does not order it against LDS
// reads of the other buffer; the kernel waits for it itself (s_waitcnt vmcnt).
__device__ __forceinline__ void nv_glds16(const unsigned short* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_byte)
      : "memory");
}
// The same with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: the
// eight-wave kernels have no registers to spare for per-lane 64-bit row addresses (hipcc hoisted
// four of them out of the step loop, spilled two, and put the reload — an s_waitcnt vmcnt(0) —
// between two DMA instructions: every stage then waited out a full memory latency twice).
// The lane's offset 16 (lane ^ cx) is formed inside the asm block (two vector instructions): kept
// as a value it is one more candidate for that spill.
__device__ __forceinline__ void nv_glds16s(const unsigned short* sbase, unsigned lane, unsigned cx, unsigned lds_byte) {
  unsigned keep, voff;
  asm volatile(
      "v_xor_b32 %1, %3, %2\n\tv_lshlrev_b32 %1, 4, %1\n\t"
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep), "=&v"(voff)
      : "v"(lane), "s"(cx), "s"(sbase), "s"(lds_byte)
      : "memory");
}
__device__ __forceinline__ unsigned nv_lds_byte_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// V_part[b, half, d, k] = sum_{n in half} x[b,n,d] * (coefn[b,n,k] * rn[b,n])
// grid (8 channel blocks of 64, NSPLIT, B); block 256; wave w contracts n-chunk split*4+w.
template <typename T>
__global__ __launch_bounds__(256) void aggregate_kernel(const void* __restrict__ xv,
                                                        const float* __restrict__ coefn,
                                                        const float* __restrict__ rn, int N,
                                                        float* __restrict__ part,
                                                        float* __restrict__ colsum_part) {
  // Each wave owns one n-chunk and a [64 channels x 64 clusters] block = 2 x 2 accumulator
  // tiles.  Channel tile t holds channels d0 + 2*i + t (i = tile row), so one 4-byte (bf16)
  // or 8-byte (f32) load per lane covers both tiles and a half-wave reads a full 128/256-B
  // line of the location.  Operands run one 8-step batch ahead of the MFMAs.
  __shared__ float red[2][4][16][64];   // 32 KB: tree reduction over the 4 waves
  __shared__ float csum[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int d0 = blockIdx.x * 64, split = blockIdx.y, b = blockIdx.z;
  int per = (N + NSPLIT * 4 - 1) / (NSPLIT * 4);
  per = (per + 1) & ~1;
  const int chunk = split * 4 + wid;
  const int n_begin = chunk * per;
  int n_end = n_begin + per;
  if (n_end > N) n_end = N;
  const int steps = n_end > n_begin ? (n_end - n_begin + 1) / 2 : 0;
  const int n_safe = n_begin < N ? n_begin : 0;
  const T* x = reinterpret_cast<const T*>(xv) + (int64_t)b * N * D + d0 + 2 * r;
  const float* cf = coefn + (int64_t)b * N * K + r;
  const float* rnb = rn + (int64_t)b * N;
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) acc[t][kt] = zero16();
  float s0 = 0.f, s1 = 0.f;

  // The prefetch registers are written by loads ONLY; every use of a loaded value (scaling
  // by rn, masking of padding steps, the column sums) happens when its batch is consumed.
  // Touching them at load time makes hipcc wait vmcnt(0) right after issuing the prefetch.
  constexpr int U = 8;
  typename Raw<T>::v2 xc[U], xn[U];
  float a0c[U], a1c[U], wc[U], a0n[U], a1n[U], wn[U];
  auto load_batch = [&](int sb, typename Raw<T>::v2* xb, float* a0b, float* a1b, float* wb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = n_begin + 2 * (sb + u) + h;
      const int ns = n < n_end ? n : n_safe;
      xb[u] = Raw<T>::ld2(x + (int64_t)ns * D);
      wb[u] = rnb[ns];
      a0b[u] = cf[(int64_t)ns * K];
      a1b[u] = cf[(int64_t)ns * K + 32];
    }
  };
  if (steps > 0) load_batch(0, xc, a0c, a1c, wc);
#pragma unroll 1
  for (int sb = 0; sb < steps; sb += U) {
    if (sb + U < steps) load_batch(sb + U, xn, a0n, a1n, wn);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // padding steps (beyond n_end) contribute nothing: zero operands
      const bool ok = n_begin + 2 * (sb + u) + h < n_end;
      const float b0 = ok ? a0c[u] * wc[u] : 0.f;
      const float b1 = ok ? a1c[u] * wc[u] : 0.f;
      const f32x2 xf = Raw<T>::cvt2(xc[u]);
      const float x0 = ok ? xf[0] : 0.f, x1 = ok ? xf[1] : 0.f;
      s0 += ok ? a0c[u] : 0.f;
      s1 += ok ? a1c[u] : 0.f;
      acc[0][0] = mfma32(x0, b0, acc[0][0]);
      acc[0][1] = mfma32(x0, b1, acc[0][1]);
      acc[1][0] = mfma32(x1, b0, acc[1][0]);
      acc[1][1] = mfma32(x1, b1, acc[1][1]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xc[u] = xn[u];
      a0c[u] = a0n[u];
      a1c[u] = a1n[u];
      wc[u] = wn[u];
    }
  }
  s0 += __shfl_xor(s0, 32, 64);
  s1 += __shfl_xor(s1, 32, 64);
  if (h == 0) {
    csum[wid][r] = s0;
    csum[wid][32 + r] = s1;
  }
  // fixed-order tree: (w0 + w2) + (w1 + w3)
  if (wid >= 2) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[wid - 2][t * 2 + kt][q][lane] = acc[t][kt][q];
  }
  __syncthreads();
  if (wid < 2) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][kt][q] += red[wid][t * 2 + kt][q][lane];
  }
  __syncthreads();
  if (wid == 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[0][t * 2 + kt][q][lane] = acc[t][kt][q];
  }
  __syncthreads();
  if (wid == 0) {
    float* out = part + (((int64_t)b * NSPLIT + split) * D + d0) * K;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ch = 2 * acc_row(q, h) + t;
          out[ch * K + kt * 32 + r] = acc[t][kt][q] + red[0][t * 2 + kt][q][lane];
        }
  }
  if (colsum_part && blockIdx.x == 0 && threadIdx.x < 64) {
    const int k = threadIdx.x;
    colsum_part[((int64_t)b * NSPLIT + split) * K + k] =
        (csum[0][k] + csum[2][k]) + (csum[1][k] + csum[3][k]);
  }
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// =======================================================================================
// Fused kernels for bf16 feature maps (round 3): ONE pass over x per direction.
//
// vlad_fwd_kernel: soft-assignment AND aggregation of one (image, location slice) per
// workgroup.  What bounded the two-kernel form (rowtile_ring + aggregate16b) was bytes per CU,
// not matrix or LDS time: every 128 locations re-streamed the 196 KB operand image of W
// through the CU's L2 -> LDS path (12-13 B per cycle and CU), and x and the assignments made a
// second trip for the aggregation.  Here
//   * W^T lives in REGISTERS for the whole kernel: wave w owns clusters 16 w .. 16 w + 15 as
//     three bf16 planes (16 k-steps x 3 planes x 4 registers = 192 per lane);
//   * the partial VLAD of the slice lives in REGISTERS too: V[512 channels][the wave's 16
//     clusters] = 32 accumulator tiles = 128 per lane (one wave per SIMD, 512-register budget);
//   * only x moves: 32-location steps through a 3-stage LDS ring filled by LDS-DMA (32 KB per
//     stage, two stages in flight), each x tile consumed twice while it is in LDS — row-wise
//     (ds_read_b128) as the B operand of the logits, column-wise (ds_read_b64_tr_b16) as the A
//     operand of the aggregation.  Unit u(loc, c) = 64 loc + (c ^ (loc & 15)) (c = 16-byte
//     chunk of the row) makes both kinds of read conflict-free (scripts/lds_conflicts.py); the
//     swizzle sits on the DMA's source address.
// The logits run with the operands SWAPPED (S^T = W^T x^T), so a lane holds four consecutive
// clusters of ONE location: the softmax needs two cross-lane steps instead of four per value, and
// a / logits leave as 16-byte stores.  Row norms come from the matrix cores as well: x_frag is
// both operands of one more MFMA per k-step, whose diagonal is sum_d x^2 (bf16 products are
// exact in float32).  The softmax over the 64 clusters spans the four waves: each wave
// publishes (max, sum exp) of its 16 clusters per location, ONE barrier, then every wave
// finishes a = e^(s - m_w) e^(m_w - M) / total.  The coefficients a * rn then go through a
// per-wave [32 loc][16 cl] x 3-plane LDS image (8-byte writes, transposed reads) to become
// the B operand of the aggregation — no round trip through HBM, no second read of x.
// Contraction index k = 8 g + e of the aggregation is location pi(k) of the step (see vf_pi):
// any permutation works as long as both operands use it, and this one keeps the transposed
// reads of the x tile conflict-free.
// Output: a, logits, rn (training), the slice's slab in accumulator order and its column sums
// of a.  grid (S slices, B images), block 256, one workgroup per CU.
constexpr int VF_STEP = 32;                       // locations per step
constexpr int VF_STAGE = VF_STEP * D * 2;         // bytes per x stage (32,768)
constexpr int VF_NST = 3;
constexpr int VF_AHEAD = 6;                      // A fragments in flight in the aggregation (<= 7)
constexpr int VF_CFLD = 48;                       // bytes per coefficient row (16 clusters + pad)
constexpr int VF_NPL = 2;                         // bf16 planes of the float32 operands (W^T, dU^T,
                                                  // a rn, ds rn): o = o1 + o2 + O(2^-17 |o|); 3 = exact
constexpr int VF_CFPL = VF_STEP * VF_CFLD;        // bytes per plane (1,536)
constexpr int VF_CF = VF_NPL * VF_CFPL;           // bytes per wave
constexpr int VF_EXCH = 4 * VF_STEP * 16;         // [wave][location][up to 4 floats]
constexpr size_t kVladFusedLds = (size_t)VF_NST * VF_STAGE + 4 * VF_CF + VF_EXCH;   // 118,784 B

// location (within the step) of contraction index k = 8 g + e
__host__ __device__ constexpr int vf_pi(int g, int e) {
  return 16 * (g >> 1) + 2 * (4 * (g & 1) + (e & 3)) + (e >> 2);
}

// W [512][64] float32 -> register images of the fused kernels: 16-byte unit
// ((w * 16 + s) * VF_NPL + plane) * 64 + lane  holds  W_plane[ch 32 s + 8 g + e][cluster 16 w + i],
// e = 0..7, for lane = 16 g + i: wave w loads its 48 fragments with coalesced 16-byte loads.
// VF_WREP identical copies, workgroups take copy (slice index) % VF_WREP.  (Tried with 4: every
// workgroup reads this image in the same order at the same time, but the prologue — 8.3 k cycles
// for 192 KB + 64 KB of x — did not move: it runs at the CU's L2 -> register rate, 30 B per cycle,
// not into a hot L2 channel.  Left at 1.)
// grid (64, VF_WREP), block 64 (block = (w, s)).
constexpr int VF_WREP = 1;
constexpr int VF_WIMG = VF_NPL * D * K;           // bf16 elements per copy
#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(64) void vlad_split_w_kernel(const float* __restrict__ w,
                                                          unsigned short* __restrict__ img) {
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  const int wv = blockIdx.x >> 4, s = blockIdx.x & 15;
  img += (int64_t)blockIdx.y * VF_WIMG;
  unsigned short h[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e)
    split3_bf16(w[(32 * s + 8 * g + e) * K + 16 * wv + i], h[0][e], h[1][e], h[2][e]);
#pragma unroll
  for (int pl = 0; pl < VF_NPL; ++pl) {
    uint4 v;
    v.x = (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16);
    v.y = (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16);
    v.z = (unsigned)h[pl][4] | ((unsigned)h[pl][5] << 16);
    v.w = (unsigned)h[pl][6] | ((unsigned)h[pl][7] << 16);
    reinterpret_cast<uint4*>(img)[(((wv * 16 + s) * VF_NPL + pl) * 64) + lane] = v;
  }
}
#endif   // SCL_DIAG

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// v + (v of the lane 16 / 32 away): one register swap between 16-lane rows / 32-lane halves
// (v_permlane16_swap_b32 / v_permlane32_swap_b32) and one add — no trip through the LDS crossbar
// like ds_bpermute (what __shfl_xor compiles to for these distances).
__device__ __forceinline__ float vf_pair16(float v, bool want_max) {
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
  return want_max ? fmaxf(a, b) : a + b;
}
__device__ __forceinline__ float vf_pair32(float v, bool want_max) {
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
  return want_max ? fmaxf(a, b) : a + b;
}
// sum / max over the four lanes (i, g = 0..3) that share a location
__device__ __forceinline__ float vf_gsum(float v) { return vf_pair32(vf_pair16(v, false), false); }
__device__ __forceinline__ float vf_gmax(float v) { return vf_pair32(vf_pair16(v, true), true); }

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
__device__ __forceinline__ u32x4 vf_ldsr128(unsigned byte) { return *(const lds_u32x4*)(size_t)byte; }
__device__ __forceinline__ uint2 vf_ldsr_tr(unsigned byte) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(size_t)byte);
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ void vf_ldsw64(unsigned byte, unsigned lo, unsigned hi) {
  *(lds_u32x2*)(size_t)byte = u32x2{lo, hi};
}
// counted wait on the wave's vector-memory queue (immediate operand: a small switch)
__device__ __forceinline__ void vf_wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

struct VladFwdArgs {
  const unsigned short* x;      // [B][N][512] bf16
  const unsigned short* wimg;   // vlad_split_w_kernel's image
  int N, pre_l2, steps_per_slice;
  float* assign;                // [B][N][64] or NULL (inference)
  float* logit;                 // [B][N][64] or NULL
  float* rnorm;                 // [B][N] or NULL
  float* slab;                  // [S][B][4 waves][32 tiles][64 lanes][4]
  float* colsum;                // [S][B][64]
  int dbg;                      // scl_debug_set_variant(916): in-kernel clock stamps
                                // (scripts/vlad_stamps.py); 0 in production
  unsigned long long* stamps;   // [workgroup][32] shader-clock stamps of wave 0 (dbg bit 4)
  float* trash;                 // 64 x 16 bytes: where the stores of rows past the end go (the
                                // kernel counts its own vector-memory queue, so every step must
                                // issue the same number of stores)
  unsigned long long* gran;     // [B][8] granules of vlad_finish_kernel's exchange: zeroed here
  // Round 6, inference with ONE workgroup per image (S == 1, nothing saved): the finish runs in the
  // tail of vlad_fwd8_kernel<false, false, true> — no slab, no second launch.  Else NULL.
  const float* fin_centers;     // [512][64]
  float* fin_out;               // [B][32768]
};

#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void vlad_fwd_kernel(VladFwdArgs p) {
  // 1 KB alignment: the transposed-read addresses are formed by XOR on (stage base + offset)
  extern __shared__ __attribute__((aligned(1024))) unsigned char vf_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  // image on blockIdx.x: workgroup b + B * sl, so with B a multiple of 8 all slices of an image
  // run on one XCD (round-robin placement: speed only) and their slabs meet in one L2
  const int b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
  const int nsteps_img = (p.N + VF_STEP - 1) / VF_STEP;
  const int st_lo = sl * p.steps_per_slice;
  const int st_hi = st_lo + p.steps_per_slice < nsteps_img ? st_lo + p.steps_per_slice : nsteps_img;
  const int nst = st_hi - st_lo;                           // >= 1 by the host's choice of S
  const unsigned lds0 = nv_lds_byte_of(vf_lds);
  const unsigned cf0 = lds0 + VF_NST * VF_STAGE + wid * VF_CF;
  float* exch = reinterpret_cast<float*>(vf_lds + VF_NST * VF_STAGE + 4 * VF_CF);
  const unsigned short* xb = p.x + (int64_t)b * p.N * D;
  unsigned long long* stp =
      (SCL_DIAG_ONLY(p.dbg) & 16) && threadIdx.x == 0 ? p.stamps + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;
#define VF_STAMP(k)                                         \
  do {                                                      \
    if (SCL_DIAG_ONLY(p.dbg) & 16) {                                       \
      __builtin_amdgcn_sched_barrier(0);                    \
      if (stp) stp[k] = __builtin_amdgcn_s_memtime();       \
      __builtin_amdgcn_sched_barrier(0);                    \
    }                                                       \
  } while (0)
  VF_STAMP(0);
  // (the finish kernel behind this one polls these words: they must be zero when it starts)
  if (sl == 0 && threadIdx.x < 8) p.gran[b * 8 + threadIdx.x] = 0ull;

  // ---- x stage by LDS-DMA: wave w brings rows w, w + 4, .. of the step; lane l of row r
  // fetches chunk l ^ (r & 15) into unit 64 r + l (rows past the end re-read the last row)
  auto stage = [&](int step) {
    const unsigned base = lds0 + (unsigned)((step - st_lo) % VF_NST) * VF_STAGE;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int r = wid + 4 * v;
      int n = VF_STEP * step + r;
      n = n < p.N ? n : p.N - 1;
      nv_glds16(xb + (int64_t)n * D + ((lane ^ (r & 15)) << 3), base + r * 1024);
    }
  };
  stage(st_lo);
  if (nst > 1) stage(st_lo + 1);

  // ---- the wave's slice of W^T: 16 k-steps x 3 planes
  u32x4 wf[16][VF_NPL];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.wimg + (int64_t)(sl % VF_WREP) * VF_WIMG) +
                       (int64_t)wid * 16 * VF_NPL * 64 + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) wf[s][pl] = src[(s * VF_NPL + pl) * 64];
  }

  if (SCL_DIAG_ONLY(p.dbg) & 16) {
    VF_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VF_STAMP(2);
  }
  f32x4 accv[32];
#pragma unroll
  for (int ct = 0; ct < 32; ++ct) accv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};

  // per-lane address parts.  Row fragment of (t, s): unit (16 t + i, 4 s + g):
  //   byte = 1024 (16 t + i) + 64 (s ^ (i >> 2)) + 16 (g ^ (i & 3));  s ^ ih = (s & ~3) | ((s & 3) ^ ih)
  unsigned rowoff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rowoff[k] = 1024u * i + 64u * (k ^ (i >> 2)) + 16u * (g ^ (i & 3));
  // Transposed fragment (half h): lane 4 q + p of group g reads 8 bytes at row pi(8 g + 4 h + q),
  // channels 16 ct + 4 p ..: byte = 1024 row + ((32 Rh + 16 (pb ^ R0) + 8 (p & 1)) ^ 32 ct)
  float dsel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dsel[j] = (g == (i >> 2) && (i & 3) == j) ? 1.0f : 0.0f;
  const int q = (lane >> 2) & 3, pp = lane & 3;
  unsigned troff[2], cfoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 16 * (g >> 1) + 2 * (4 * (g & 1) + q) + h;     // vf_pi(g, 4 h + q)
    const int R = row & 15;
    troff[h] = 1024u * row + 32u * (R >> 1) + 16u * ((pp >> 1) ^ (R & 1)) + 8u * (pp & 1);
    cfoff[h] = (unsigned)row * VF_CFLD + 8u * pp;
  }

#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int step = st_lo + st;
    // This stage's DMA must have landed; younger than it in the wave's queue, and allowed to
    // stay in flight: the next stage's DMA (8) and the previous step's stores (5 when saving).
    vf_wait_vm((st + 1 < nst ? 8 : 0) + (st >= 1 && SAVE ? 5 : 0));
    __builtin_amdgcn_s_barrier();         // landed for every wave; the stage read in step - 1 is free
    if (st < 4) VF_STAMP(4 + 6 * st);
    const unsigned sb = lds0 + (unsigned)(st % VF_NST) * VF_STAGE;

    // ---- logits (swapped: lane = location, registers = 4 consecutive clusters) + row norms
    f32x4 accl[2], accn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      accl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      accn[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 xf[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        xf[s][t] = vf_ldsr128(sb + rowoff[s & 3] + 256u * (s >> 2) + 16384u * t);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 2 < 16) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[(s + 2) & 3][t] =
              vf_ldsr128(sb + rowoff[(s + 2) & 3] + 256u * ((s + 2) >> 2) + 16384u * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int pl = 0; pl < VF_NPL; ++pl) accl[t] = mfma16b(wf[s][pl], xf[s & 3][t], accl[t]);
        accn[t] = mfma16b(xf[s & 3][t], xf[s & 3][t], accn[t]);
      }
    }

    if (st < 4) VF_STAMP(5 + 6 * st);
    // ---- softmax over the 64 clusters (this wave: 16 of them), coefficients, outputs
    float av[2][4], ev[2][4], mloc[2], rnv[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = VF_STEP * step + 16 * t + i;
      ok[t] = n < p.N;
      // diagonal of the tile's Gram: row 4 g + j == column i (dsel: 1 on that register, else 0)
      float d = accn[t][0] * dsel[0] + accn[t][1] * dsel[1] + accn[t][2] * dsel[2] + accn[t][3] * dsel[3];
      d = vf_gsum(d);
      rnv[t] = p.pre_l2 ? rsqrtf(fmaxf(d, 1e-12f)) : 1.0f;
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ev[t][j] = accl[t][j] * rnv[t];                    // the logit
        m = fmaxf(m, ev[t][j]);
      }
      m = vf_gmax(m);
      mloc[t] = m;
      if (SAVE)
        *reinterpret_cast<f32x4*>(ok[t] ? p.logit + ((int64_t)b * p.N + n) * K + 16 * wid + 4 * g
                                        : p.trash + 4 * lane) =
            f32x4{ev[t][0], ev[t][1], ev[t][2], ev[t][3]};
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ev[t][j] = __expf(ev[t][j] - m);
        sum += ev[t][j];
      }
      sum = vf_gsum(sum);
      if (g == 0) *reinterpret_cast<f32x2*>(exch + (wid * VF_STEP + 16 * t + i) * 2) = f32x2{m, sum};
    }
    __builtin_amdgcn_s_barrier();
    if (st < 4) VF_STAMP(6 + 6 * st);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = VF_STEP * step + 16 * t + i;
      f32x2 ms[4];
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2)
        ms[w2] = *reinterpret_cast<const f32x2*>(exch + (w2 * VF_STEP + 16 * t + i) * 2);
      const float M = fmaxf(fmaxf(ms[0][0], ms[1][0]), fmaxf(ms[2][0], ms[3][0]));
      float tot = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) tot += ms[w2][1] * __expf(ms[w2][0] - M);
      const float sc = __fdividef(__expf(mloc[t] - M), tot);
      unsigned short h[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[t][j] = ev[t][j] * sc;
        const float a_ok = ok[t] ? av[t][j] : 0.f;
        cs[j] += a_ok;
        split3_bf16(a_ok * rnv[t], h[0][j], h[1][j], h[2][j]);
      }
      if (SAVE)
        *reinterpret_cast<f32x4*>(ok[t] ? p.assign + ((int64_t)b * p.N + n) * K + 16 * wid + 4 * g
                                        : p.trash + 4 * lane) =
            f32x4{av[t][0], av[t][1], av[t][2], av[t][3]};
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) {
        vf_ldsw64(cf0 + pl * VF_CFPL + (16 * t + i) * VF_CFLD + 8 * g,
                  (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16),
                  (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16));
      }
    }
    if (SAVE) {   // rn: wave w writes the step's locations 8 w .. 8 w + 7 (one store per wave)
      const int t = wid >> 1;
      const int n = VF_STEP * step + 16 * t + i;
      const float r = t == 0 ? rnv[0] : rnv[1];
      const bool mine = g == 0 && (i >> 3) == (wid & 1) && n < p.N;
      *(mine ? p.rnorm + (int64_t)b * p.N + n : p.trash + 4 * lane) = r;
    }

    if (st < 4) VF_STAMP(7 + 6 * st);
    if (st + 2 < nst) stage(step + 2);     // into the stage of step - 1

    // ---- aggregation: V[ch][cl] += sum_loc x[loc][ch] * (a rn)[loc][cl]
    u32x4 bfr[VF_NPL];
#pragma unroll
    for (int pl = 0; pl < VF_NPL; ++pl) {
      const uint2 lo = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[0]);
      const uint2 hi = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[1]);
      bfr[pl] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    // address of the transposed fragment of channel tile ct: (stage + troff[h]) ^ 32 ct touches
    // address bits 5..9 only, and (Rh ^ ct) = (ct & 24) | ((ct & 7) ^ Rh): eight per-lane bases
    // (ct & 7) per half, the rest is an immediate offset — no address arithmetic in the loop
    unsigned ta[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c7 = 0; c7 < 8; ++c7) ta[h][c7] = sb + (troff[h] ^ (32u * c7));
    // A fragments VF_AHEAD channel tiles ahead of the matrix work (an LDS round trip is several
    // times the MFMAs of a tile)
    u32x4 af[8];
#pragma unroll
    for (int ct = 0; ct < VF_AHEAD; ++ct) {
      const uint2 lo = vf_ldsr_tr(ta[0][ct & 7] + 32u * (ct & 24)), hi = vf_ldsr_tr(ta[1][ct & 7] + 32u * (ct & 24));
      af[ct] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) {
      if (ct + VF_AHEAD < 32) {
        const int cn = ct + VF_AHEAD;
        const uint2 lo = vf_ldsr_tr(ta[0][cn & 7] + 32u * (cn & 24)),
                    hi = vf_ldsr_tr(ta[1][cn & 7] + 32u * (cn & 24));
        af[cn & 7] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) accv[ct] = mfma16b(af[ct & 7], bfr[pl], accv[ct]);
    }
    if (st < 4) VF_STAMP(8 + 6 * st);
  }
  VF_STAMP(28);

  // ---- the slice's slab, in accumulator order, and the column sums of a
  f32x4* slab = reinterpret_cast<f32x4*>(p.slab) + ((((int64_t)sl * B + b) * 4 + wid) * 32) * 64 + lane;
#pragma unroll
  for (int ct = 0; ct < 32; ++ct) slab[ct * 64] = accv[ct];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) cs[j] += __shfl_xor(cs[j], m, 64);
  }
  if (i == 0)
    *reinterpret_cast<f32x4*>(p.colsum + ((int64_t)sl * B + b) * K + 16 * wid + 4 * g) =
        f32x4{cs[0], cs[1], cs[2], cs[3]};
  if (SCL_DIAG_ONLY(p.dbg) & 16) {
    VF_STAMP(29);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VF_STAMP(30);
  }
#undef VF_STAMP
}
#endif   // SCL_DIAG

// U = sum of the slices' slabs + C * asum for one 16-channel tile -> vlad[b] (natural [513][64]
// layout, the saved pre-norm VLAD) and the tile's column sums of squares.  Slabs are in
// accumulator order: 16-byte unit ((w * 32 + ct) * 64 + lane) = U[16 ct + 4 g + 0..3][16 w + i].
// grid (B, 32 channel tiles), block 256: thread = (wave w, lane); all S loads of a thread are in
// flight together (31 MB of slabs at 24 x 1200: a reader with one load at a time took 20 us).
// Image on blockIdx.x like vlad_fwd_kernel: with B a multiple of 8 an image's slabs are read on
// the XCD whose L2 they were written through (placement is speed only).
constexpr int VF_MAXS = 16;
#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256) void vlad_finish_sum_kernel(const float* __restrict__ slab,
                                                              const float* __restrict__ colsum,
                                                              const float* __restrict__ centers,
                                                              int S, float* __restrict__ vlad,
                                                              float* __restrict__ colsq_part) {
  const int b = blockIdx.x, ct = blockIdx.y, B = gridDim.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int k = 16 * w + i;
  const f32x4* src = reinterpret_cast<const f32x4*>(slab) + (((int64_t)b * 4 + w) * 32 + ct) * 64 + lane;
  const int64_t sstride = (int64_t)B * 4 * 32 * 64;          // units between slices
  f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;
  for (int s0 = 0; s0 < S; s0 += VF_MAXS) {
    f32x4 v[VF_MAXS];
    float a[VF_MAXS];
#pragma unroll
    for (int s = 0; s < VF_MAXS; ++s) {
      const bool ok = s0 + s < S;
      v[s] = ok ? src[(s0 + s) * sstride] : f32x4{0.f, 0.f, 0.f, 0.f};
      a[s] = ok ? colsum[((int64_t)(s0 + s) * B + b) * K + k] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < VF_MAXS; ++s) {                       // fixed order: bitwise reproducible
      u += v[s];
      asum += a[s];
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int d = 16 * ct + 4 * g + j;
    const float v = u[j] + centers[d * K + k] * asum;
    vlad[((int64_t)b * VROWS + d) * K + k] = v;
    ss = fmaf(v, v, ss);
  }
  if (ct == 0 && g == 0) vlad[((int64_t)b * VROWS + D) * K + k] = asum;
  ss = vf_gsum(ss);
  if (g == 0) colsq_part[((int64_t)b * 32 + ct) * K + k] = ss;
}
#endif   // SCL_DIAG

// vlad_bwd_kernel: the backward twin of vlad_fwd_kernel — x.dU[b], the softmax backward and the
// weight-gradient aggregation x^T.(ds rn) of one (image, location slice) in one pass over x.
// The image's dU^T sits in registers (bwd_du_kernel writes it as the same register image the
// forward uses for W^T), the slice's partial dW in the accumulators.  Per location the softmax
// backward needs sums over all 64 clusters, i.e. over the four waves; every wave publishes four
// partial sums over its 16 clusters,
//   P1 = sum a da,  P2 = sum a t,  P3 = sum a da lg,  P4 = sum a lg     (t = xhat.dU, da = t + c.dU)
// and after ONE barrier   dot = sum P1,   <dxhat, xhat> = sum P2 + sum P3 - dot sum P4
// (= sum_k a t + ds lg with ds = a (da - dot)).  Outputs: ds, rowdot, the slice's dW slab.
struct VladBwdArgs {
  const unsigned short* x;      // [B][N][512] bf16
  const unsigned short* duimg;  // [B] register images of dU^T (bwd_du_kernel)
  const float* a;               // [B][N][64] saved assignments
  const float* lg;              // [B][N][64] saved logits
  const float* rn;              // [B][N]
  const float* cdu;             // [B][64]
  int N, steps_per_slice;
  float* ds;                    // [B][N][64]
  float* rowdot;                // [B][N]
  float* slab;                  // [S][B][4][32][64][4]
  float* trash;
};

#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256, 1) void vlad_bwd_kernel(VladBwdArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char vf_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
  const int nsteps_img = (p.N + VF_STEP - 1) / VF_STEP;
  const int st_lo = sl * p.steps_per_slice;
  const int st_hi = st_lo + p.steps_per_slice < nsteps_img ? st_lo + p.steps_per_slice : nsteps_img;
  const int nst = st_hi - st_lo;
  const unsigned lds0 = nv_lds_byte_of(vf_lds);
  const unsigned cf0 = lds0 + VF_NST * VF_STAGE + wid * VF_CF;
  float* exch = reinterpret_cast<float*>(vf_lds + VF_NST * VF_STAGE + 4 * VF_CF);   // [wave][loc][4]
  const unsigned short* xb = p.x + (int64_t)b * p.N * D;

  auto stage = [&](int step) {
    const unsigned base = lds0 + (unsigned)((step - st_lo) % VF_NST) * VF_STAGE;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int r = wid + 4 * v;
      int n = VF_STEP * step + r;
      n = n < p.N ? n : p.N - 1;
      nv_glds16(xb + (int64_t)n * D + ((lane ^ (r & 15)) << 3), base + r * 1024);
    }
  };
  stage(st_lo);
  if (nst > 1) stage(st_lo + 1);

  u32x4 wf[16][VF_NPL];
  {
    const u32x4* src =
        reinterpret_cast<const u32x4*>(p.duimg) + ((int64_t)b * 4 + wid) * 16 * VF_NPL * 64 + lane;
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) wf[s][pl] = src[(s * VF_NPL + pl) * 64];
  }
  const f32x4 cd = *reinterpret_cast<const f32x4*>(p.cdu + b * K + 16 * wid + 4 * g);

  f32x4 accv[32];
#pragma unroll
  for (int ct = 0; ct < 32; ++ct) accv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned rowoff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rowoff[k] = 1024u * i + 64u * (k ^ (i >> 2)) + 16u * (g ^ (i & 3));
  const int q = (lane >> 2) & 3, pp = lane & 3;
  unsigned troff[2], cfoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 16 * (g >> 1) + 2 * (4 * (g & 1) + q) + h;
    const int R = row & 15;
    troff[h] = 1024u * row + 32u * (R >> 1) + 16u * ((pp >> 1) ^ (R & 1)) + 8u * (pp & 1);
    cfoff[h] = (unsigned)row * VF_CFLD + 8u * pp;
  }

#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int step = st_lo + st;
    // younger than this stage's DMA and allowed in flight: the next stage's DMA (8) and the
    // previous step's three stores (its loads were waited for when they were used)
    vf_wait_vm((st + 1 < nst ? 8 : 0) + (st >= 1 ? 3 : 0));
    __builtin_amdgcn_s_barrier();
    const unsigned sb = lds0 + (unsigned)(st % VF_NST) * VF_STAGE;

    // the step's saved forward values, in flight under the matrix work
    f32x4 a4[2], l4[2];
    float rn2[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = VF_STEP * step + 16 * t + i;
      ok[t] = n < p.N;
      const int64_t row = (int64_t)b * p.N + (ok[t] ? n : p.N - 1);
      a4[t] = *reinterpret_cast<const f32x4*>(p.a + row * K + 16 * wid + 4 * g);
      l4[t] = *reinterpret_cast<const f32x4*>(p.lg + row * K + 16 * wid + 4 * g);
      rn2[t] = p.rn[row];
    }

    f32x4 accl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) accl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 xf[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        xf[s][t] = vf_ldsr128(sb + rowoff[s & 3] + 256u * (s >> 2) + 16384u * t);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 2 < 16) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[(s + 2) & 3][t] =
              vf_ldsr128(sb + rowoff[(s + 2) & 3] + 256u * ((s + 2) >> 2) + 16384u * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < VF_NPL; ++pl) accl[t] = mfma16b(wf[s][pl], xf[s & 3][t], accl[t]);
    }

    float tv[2][4], dav[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tv[t][j] = accl[t][j] * rn2[t];                    // xhat . dU
        dav[t][j] = tv[t][j] + cd[j];
        const float ad = a4[t][j] * dav[t][j];
        p1 += ad;
        p2 = fmaf(a4[t][j], tv[t][j], p2);
        p3 = fmaf(ad, l4[t][j], p3);
        p4 = fmaf(a4[t][j], l4[t][j], p4);
      }
      p1 = vf_gsum(p1);
      p2 = vf_gsum(p2);
      p3 = vf_gsum(p3);
      p4 = vf_gsum(p4);
      if (g == 0) *reinterpret_cast<f32x4*>(exch + (wid * VF_STEP + 16 * t + i) * 4) = f32x4{p1, p2, p3, p4};
    }
    __builtin_amdgcn_s_barrier();
    float rd2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = VF_STEP * step + 16 * t + i;
      f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2)
        ps += *reinterpret_cast<const f32x4*>(exch + (w2 * VF_STEP + 16 * t + i) * 4);
      const float dot = ps[0];
      rd2[t] = ps[1] + ps[2] - dot * ps[3];
      float dsv[4];
      unsigned short h[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dsv[j] = a4[t][j] * (dav[t][j] - dot);
        split3_bf16(ok[t] ? dsv[j] * rn2[t] : 0.f, h[0][j], h[1][j], h[2][j]);
      }
      *reinterpret_cast<f32x4*>(ok[t] ? p.ds + ((int64_t)b * p.N + n) * K + 16 * wid + 4 * g
                                      : p.trash + 4 * lane) = f32x4{dsv[0], dsv[1], dsv[2], dsv[3]};
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl)
        vf_ldsw64(cf0 + pl * VF_CFPL + (16 * t + i) * VF_CFLD + 8 * g,
                  (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16),
                  (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16));
    }
    {   // rowdot: wave w writes the step's locations 8 w .. 8 w + 7
      const int t = wid >> 1;
      const int n = VF_STEP * step + 16 * t + i;
      const float r = t == 0 ? rd2[0] : rd2[1];
      const bool mine = g == 0 && (i >> 3) == (wid & 1) && n < p.N;
      *(mine ? p.rowdot + (int64_t)b * p.N + n : p.trash + 4 * lane) = r;
    }
    if (st + 2 < nst) stage(step + 2);

    u32x4 bfr[VF_NPL];
#pragma unroll
    for (int pl = 0; pl < VF_NPL; ++pl) {
      const uint2 lo = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[0]);
      const uint2 hi = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[1]);
      bfr[pl] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    // address of the transposed fragment of channel tile ct: (stage + troff[h]) ^ 32 ct touches
    // address bits 5..9 only, and (Rh ^ ct) = (ct & 24) | ((ct & 7) ^ Rh): eight per-lane bases
    // (ct & 7) per half, the rest is an immediate offset — no address arithmetic in the loop
    unsigned ta[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c7 = 0; c7 < 8; ++c7) ta[h][c7] = sb + (troff[h] ^ (32u * c7));
    // A fragments VF_AHEAD channel tiles ahead of the matrix work (an LDS round trip is several
    // times the MFMAs of a tile)
    u32x4 af[8];
#pragma unroll
    for (int ct = 0; ct < VF_AHEAD; ++ct) {
      const uint2 lo = vf_ldsr_tr(ta[0][ct & 7] + 32u * (ct & 24)), hi = vf_ldsr_tr(ta[1][ct & 7] + 32u * (ct & 24));
      af[ct] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int ct = 0; ct < 32; ++ct) {
      if (ct + VF_AHEAD < 32) {
        const int cn = ct + VF_AHEAD;
        const uint2 lo = vf_ldsr_tr(ta[0][cn & 7] + 32u * (cn & 24)),
                    hi = vf_ldsr_tr(ta[1][cn & 7] + 32u * (cn & 24));
        af[cn & 7] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) accv[ct] = mfma16b(af[ct & 7], bfr[pl], accv[ct]);
    }
  }

  f32x4* slab = reinterpret_cast<f32x4*>(p.slab) + ((((int64_t)sl * B + b) * 4 + wid) * 32) * 64 + lane;
#pragma unroll
  for (int ct = 0; ct < 32; ++ct) slab[ct * 64] = accv[ct];
}
#endif   // SCL_DIAG

// =======================================================================================
// Round 4: the two fused kernels with EIGHT waves per workgroup (two per SIMD).
//
// What the stamps of the four-wave kernels showed (profiles/r03/vlad_stamps_fwd.txt): a 32-location
// step takes 6.4 k cycles, of which 160 MFMAs x 16 cycles = 2.6 k are matrix work; the rest is the
// softmax / coefficient arithmetic and its cross-lane and LDS traffic (2.1 k), LDS latency in front
// of the aggregation's products, and two barriers — all strictly one after the other, because a
// wave that owns a SIMD alone issues in order: a vector instruction costs it 4 cycles (2 with a
// second wave on the SIMD) and nothing runs while it waits.  Here wave (w, h) = clusters 16 w .. +15
// x CHANNEL HALF h:
//   * registers per wave halve (W^T fragments 64, VLAD accumulators 64), so two waves fit a SIMD:
//     one wave's LDS round trips, waits and vector work hide behind the other's, and vector
//     instructions issue at twice the rate;
//   * the logits are partial sums over the wave's 256 channels: the wave keeps the partial of
//     location tile t = h and hands the other tile's to its partner (w, 1 - h) through LDS — each
//     wave then runs the softmax arithmetic of 16 locations instead of 32;
//   * the row norms (one more MFMA per k-step: the diagonal of x x^T) were computed by all four
//     waves alike; now wave (w, h) takes k-steps w and w + 4 of its half and the eight partials are
//     added after the exchange: 4 instead of 32 norm MFMAs per wave and step (136 instead of 160 MFMAs
//     per SIMD and step);
//   * aggregation: the wave's 16 channel tiles x its 16 clusters, B operand = the pair's coefficient
//     image (rows 16 h .. of it written by wave (w, h)).
// Four barriers per step (stage landed | partials exchanged | (max, sum exp) exchanged |
// coefficients written) instead of two; x staging, LDS images, slab layout, saved outputs and the
// W^T register image are the four-wave kernel's (vlad_fwd_kernel), so every consumer is unchanged.
// scl_debug_set_variant(922) runs the four-wave kernels (same-box A/B).
constexpr int V8_XL = 8 * 64 * 16;               // partial logits for the partner: [wave][lane] x 16 B
constexpr int V8_PN = 2 * 16 * 8 * 4;            // norm partials [tile][location][wave]
constexpr int V8_EXCH = 2 * 4 * 16 * 16;         // [tile][cluster group][location] x up to 4 floats
constexpr int V8_CS = 4 * 16 * 4;                // column sums of a of the waves h = 1
constexpr size_t kVlad8Lds = (size_t)VF_NST * VF_STAGE + 4 * VF_CF + V8_XL + V8_PN + V8_EXCH + V8_CS;
constexpr int V8_AHEAD = 4;                      // A fragments in flight in the aggregation

__device__ __forceinline__ void v8_wait_vm(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// STAMPS: the diagnostic instance (scripts/vlad_stamps.py, scl_debug_set_variant(918)); the production
// instances carry no stamp code.
//
// Tried on top of this and NOT kept (second half of round 4, profiles/r04/vlad_stamps_fwd8_pipelined.txt):
// a software pipeline over the steps — aggregation of step i - 1 with the first half of step i's softmax
// chain hand-interleaved into it, partial logits of step i + 1 with the second half, two barriers per
// step instead of four.  Blocks measured 1.8 k + 2.6 k cycles against 1.0 k + 1.25 k for the bare matrix
// phases and 1.2 k + 0.8 k for the bare chains: vector and matrix instructions of a SIMD's two waves
// ADD (what scripts/mfma_valu_overlap.hip showed in round 1 for two waves of different workgroups
// holds inside a workgroup too), and the extra live state (256 registers per wave) cost spills; the
// kernel came out 0.9 us SLOWER than this one.  What pays on this chip is fewer vector instructions
// and fewer barrier intervals, not overlap.
template <bool SAVE, bool STAMPS = false, bool FIN = false>
__global__ __launch_bounds__(512) void vlad_fwd8_kernel(VladFwdArgs p) {
  static_assert(!FIN || !SAVE, "the fused finish is the inference path");
  extern __shared__ __attribute__((aligned(1024))) unsigned char vf_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = wid & 3, h = wid >> 2;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
  const int nsteps_img = (p.N + VF_STEP - 1) / VF_STEP;
  const int st_lo = sl * p.steps_per_slice;
  const int st_hi = st_lo + p.steps_per_slice < nsteps_img ? st_lo + p.steps_per_slice : nsteps_img;
  const int nst = st_hi - st_lo;                           // >= 1 by the host's choice of S
  const unsigned lds0 = nv_lds_byte_of(vf_lds);
  const unsigned cf0 = lds0 + VF_NST * VF_STAGE + w * VF_CF;
  unsigned char* xl = vf_lds + VF_NST * VF_STAGE + 4 * VF_CF;
  float* pn = reinterpret_cast<float*>(xl + V8_XL);
  float* exch = pn + V8_PN / 4;
  float* csx = exch + V8_EXCH / 4;
  const unsigned short* xb = p.x + (int64_t)b * p.N * D;
#define VF_STAMP(k)                                                                              \
  do {                                                                                           \
    if constexpr (STAMPS) {                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      if (threadIdx.x == 0)                                                                      \
        p.stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 + (k)] = __builtin_amdgcn_s_memtime(); \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  } while (0)
  VF_STAMP(0);
  // (the finish kernel behind this one polls these words: they must be zero when it starts)
  if (sl == 0 && threadIdx.x < 8) p.gran[b * 8 + threadIdx.x] = 0ull;

  // ---- x stage by LDS-DMA: wave wid brings rows wid, wid + 8, .. of the step; lane l of row r
  // fetches chunk l ^ (r & 15) into unit 64 r + l (rows past the end re-read the last row)
  auto stage = [&](int step) {
    const unsigned base = lds0 + (unsigned)((step - st_lo) % VF_NST) * VF_STAGE;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = wid + 8 * v;                           // (wave-uniform: the row base is scalar)
      int n = VF_STEP * step + r;
      n = n < p.N ? n : p.N - 1;
      nv_glds16s(xb + (int64_t)n * D, (unsigned)lane, (unsigned)(r & 15), base + r * 1024);
    }
  };
  stage(st_lo);
  // ---- the wave's slice of W^T: its 8 k-steps (channels 256 h + 32 s ..) x the planes
  u32x4 wf[8][VF_NPL];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.wimg) + (int64_t)(w * 16 + 8 * h) * VF_NPL * 64 + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) wf[s][pl] = src[(s * VF_NPL + pl) * 64];
  }
  if (nst > 1) stage(st_lo + 1);
  VF_STAMP(1);

  f32x4 accv[16];
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) accv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  // wave-uniform bases of the image's saved rows + 32-bit per-lane offsets
  float* const as_img = SAVE ? p.assign + (int64_t)b * p.N * K : nullptr;
  float* const rn_img = SAVE ? p.rnorm + (int64_t)b * p.N : nullptr;

  // Row fragment of (tile t, global k-step s' = 8 h + s): unit (16 t + i, 4 s' + g):
  //   byte = 1024 (16 t + i) + 256 (s' >> 2) + 64 ((s' & 3) ^ (i >> 2)) + 16 (g ^ (i & 3))
  unsigned rowoff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rowoff[k] = 1024u * i + 512u * h + 64u * (k ^ (i >> 2)) + 16u * (g ^ (i & 3));
  float dsel[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dsel[j] = (g == (i >> 2) && (i & 3) == j) ? 1.0f : 0.0f;
  const int q = (lane >> 2) & 3, pp = lane & 3;
  unsigned troff[2], cfoff[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = 16 * (g >> 1) + 2 * (4 * (g & 1) + q) + hh;     // vf_pi(g, 4 hh + q)
    const int R = row & 15;
    troff[hh] = 1024u * row + 512u * h + 32u * (R >> 1) + 16u * ((pp >> 1) ^ (R & 1)) + 8u * (pp & 1);
    cfoff[hh] = (unsigned)row * VF_CFLD + 8u * pp;
  }

#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int step = st_lo + st;
    // This stage's DMA must have landed; younger in the wave's queue, allowed to stay in flight:
    // the next stage's DMA (4) and the previous step's two stores (when saving: assignments and
    // row norms; round 6: the logits are no longer saved — the backward pass takes log a instead).
    v8_wait_vm((st + 1 < nst ? 4 : 0) + (st >= 1 && SAVE ? 2 : 0));
    __builtin_amdgcn_s_barrier();         // landed for every wave; step - 1 is finished everywhere
    if (st < 4) VF_STAMP(4 + 6 * st);
    const unsigned sb = lds0 + (unsigned)(st % VF_NST) * VF_STAGE;

    // ---- partial logits over the wave's 256 channels (swapped: lane = location, registers = 4
    // consecutive clusters) + its share of the row norms
    f32x4 accl[2], accn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      accl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      accn[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4 xf[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        xf[s][t] = vf_ldsr128(sb + rowoff[s & 3] + 256u * (s >> 2) + 16384u * t);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 2 < 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[(s + 2) & 3][t] =
              vf_ldsr128(sb + rowoff[(s + 2) & 3] + 256u * ((s + 2) >> 2) + 16384u * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int pl = 0; pl < VF_NPL; ++pl) accl[t] = mfma16b(wf[s][pl], xf[s & 3][t], accl[t]);
      }
      if ((s & 3) == w) {                                   // (wave-uniform)
#pragma unroll
        for (int t = 0; t < 2; ++t) accn[t] = mfma16b(xf[s & 3][t], xf[s & 3][t], accn[t]);
      }
    }
    if (st < 4) VF_STAMP(5 + 6 * st);

    // ---- exchange 1: the other tile's partial logits to the partner, norm partials to everybody
    *reinterpret_cast<f32x4*>(xl + (wid * 64 + lane) * 16) = h ? accl[0] : accl[1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      // diagonal of the tile's Gram: row 4 g + j == column i (dsel: 1 on that register, else 0)
      float d = accn[t][0] * dsel[0] + accn[t][1] * dsel[1] + accn[t][2] * dsel[2] + accn[t][3] * dsel[3];
      d = vf_gsum(d);
      if (g == 0) pn[(t * 16 + i) * 8 + wid] = d;
    }
    __builtin_amdgcn_s_barrier();
    if (st < 4) VF_STAMP(6 + 6 * st);

    // ---- the wave's tile (t = h): logits, softmax over the 64 clusters (this wave: 16 of them)
    // Locations past the end of the image are copies of its last one (the stage re-reads that row):
    // their saved rows are stored over it with the same values — one destination, and every step
    // issues the same stores (the kernel counts its own vector-memory queue).
    const int nraw = VF_STEP * step + 16 * h + i;
    const bool ok = nraw < p.N;
    const unsigned n = ok ? nraw : p.N - 1;
    float ev[4], av[4], mloc, rnv;
    {
      const f32x4 own = h ? accl[1] : accl[0];
      const f32x4 oth = *reinterpret_cast<const f32x4*>(xl + ((wid ^ 4) * 64 + lane) * 16);
      const f32x4 lo4 = h ? oth : own, hi4 = h ? own : oth;   // channel halves in a fixed order
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(pn + (h * 16 + i) * 8);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(pn + (h * 16 + i) * 8 + 4);
      const float d = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
      rnv = p.pre_l2 ? rsqrtf(fmaxf(d, 1e-12f)) : 1.0f;
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ev[j] = (lo4[j] + hi4[j]) * rnv;                    // the logit
        m = fmaxf(m, ev[j]);
      }
      m = vf_gmax(m);
      mloc = m;
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ev[j] = __expf(ev[j] - m);
        sum += ev[j];
      }
      sum = vf_gsum(sum);
      if (g == 0) *reinterpret_cast<f32x2*>(exch + ((h * 4 + w) * 16 + i) * 2) = f32x2{m, sum};
    }
    __builtin_amdgcn_s_barrier();
    if (st < 4) VF_STAMP(7 + 6 * st);
    {
      f32x2 ms[4];
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2)
        ms[w2] = *reinterpret_cast<const f32x2*>(exch + ((h * 4 + w2) * 16 + i) * 2);
      const float M = fmaxf(fmaxf(ms[0][0], ms[1][0]), fmaxf(ms[2][0], ms[3][0]));
      float tot = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) tot += ms[w2][1] * __expf(ms[w2][0] - M);
      const float sc = __fdividef(__expf(mloc - M), tot);
      unsigned short hh[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        av[j] = ev[j] * sc;
        const float a_ok = ok ? av[j] : 0.f;
        cs[j] += a_ok;
        split3_bf16(a_ok * rnv, hh[0][j], hh[1][j], hh[2][j]);
      }
      if (SAVE) *reinterpret_cast<f32x4*>(as_img + (n * K + 16 * w + 4 * g)) = f32x4{av[0], av[1], av[2], av[3]};
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) {
        vf_ldsw64(cf0 + pl * VF_CFPL + (16 * h + i) * VF_CFLD + 8 * g,
                  (unsigned)hh[pl][0] | ((unsigned)hh[pl][1] << 16),
                  (unsigned)hh[pl][2] | ((unsigned)hh[pl][3] << 16));
      }
      if (SAVE) {   // rn: wave (w, h) writes locations 4 w .. 4 w + 3 of its tile (one store per wave)
        if (g == 0 && (i >> 2) == w) rn_img[n] = rnv;
      }
    }
    __builtin_amdgcn_s_barrier();          // the pair's coefficient rows are both written
    if (st < 4) VF_STAMP(8 + 6 * st);
    if (st + 2 < nst) stage(step + 2);     // into the stage of step - 1

    // ---- aggregation: V[ch][cl] += sum_loc x[loc][ch] * (a rn)[loc][cl], channel tiles 16 h ..
    u32x4 bfr[VF_NPL];
#pragma unroll
    for (int pl = 0; pl < VF_NPL; ++pl) {
      const uint2 lo = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[0]);
      const uint2 hi = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[1]);
      bfr[pl] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    // address of the transposed fragment of channel tile ct = 16 h + c: (stage + troff[hh]) ^ 32 (c & 7)
    // + 32 (c & 8) (troff carries the 512 h): XOR touches address bits 5..7 only
    const unsigned tb0 = sb + troff[0], tb1 = sb + troff[1];
    u32x4 af[8];
#pragma unroll
    for (int c = 0; c < V8_AHEAD; ++c) {
      const uint2 lo = vf_ldsr_tr((tb0 ^ (32u * (c & 7))) + 32u * (c & 8)),
                  hi = vf_ldsr_tr((tb1 ^ (32u * (c & 7))) + 32u * (c & 8));
      af[c] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c + V8_AHEAD < 16) {
        const int cn = c + V8_AHEAD;
        const uint2 lo = vf_ldsr_tr((tb0 ^ (32u * (cn & 7))) + 32u * (cn & 8)),
                    hi = vf_ldsr_tr((tb1 ^ (32u * (cn & 7))) + 32u * (cn & 8));
        af[cn & 7] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) accv[c] = mfma16b(af[c & 7], bfr[pl], accv[c]);
    }
    if (st < 4) VF_STAMP(9 + 6 * st);
  }
  VF_STAMP(28);

#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) cs[j] += __shfl_xor(cs[j], m, 64);
  }
  if constexpr (FIN) {
    // ---- the finish of vlad_finish_kernel on the accumulators (this workgroup holds the whole
    // image): lane (i, g) of wave (w, h) owns cluster k = 16 w + i, channels 16 (16 h + c) + 4 g + j.
    //   U = V + C * asum;  col_k = sum_d U^2;  q_k = 1 / sqrt(col_k + 1e-12);
    //   tot = sum_k q_k^2 col_k;  out = U q_k / sqrt(tot + 1e-12)
    // (the sums run in another order than the finish kernel's: same values to rounding).
    float* fl = reinterpret_cast<float*>(vf_lds);              // the x stages are dead
    float* f_as = fl;                                          // [2][64] column sums of a by h
    float* f_sq = fl + 128;                                    // [2][64] sum U^2 by channel half
    float* f_tq = fl + 256;                                    // [4]     q^2 col by cluster group
    __builtin_amdgcn_s_barrier();                              // every wave is done with the stages
    if (i == 0) *reinterpret_cast<f32x4*>(f_as + h * 64 + 16 * w + 4 * g) = f32x4{cs[0], cs[1], cs[2], cs[3]};
    __builtin_amdgcn_s_barrier();
    const int k = 16 * w + i;
    const float asum = f_as[k] + f_as[64 + k];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = accv[c][j] + p.fin_centers[(16 * (16 * h + c) + 4 * g + j) * K + k] * asum;
        accv[c][j] = u;
        ss = fmaf(u, u, ss);
      }
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (g == 0) f_sq[h * 64 + k] = ss;
    __builtin_amdgcn_s_barrier();
    const float col = f_sq[k] + f_sq[64 + k];
    const float q = 1.0f / sqrtf(col + 1e-12f);
    float tq = q * q * col;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) tq += __shfl_xor(tq, m, 64);
    if (h == 0 && lane == 0) f_tq[w] = tq;
    __builtin_amdgcn_s_barrier();
    const float tot = (f_tq[0] + f_tq[1]) + (f_tq[2] + f_tq[3]);
    const float sc = q * (1.0f / sqrtf(tot + 1e-12f));
    float* orow = p.fin_out + (int64_t)b * D * K + k;
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) orow[(16 * (16 * h + c) + 4 * g + j) * K] = accv[c][j] * sc;
    return;
  }
  // ---- the slice's slab, in accumulator order (unit ((w * 32 + ct) * 64 + lane), ct = 16 h + c),
  // and the column sums of a (the two tile halves of a cluster group combined through LDS)
  f32x4* slab =
      reinterpret_cast<f32x4*>(p.slab) + ((((int64_t)sl * B + b) * 4 + w) * 32 + 16 * h) * 64 + lane;
#pragma unroll
  for (int c = 0; c < 16; ++c) slab[c * 64] = accv[c];
  if (h == 1 && i == 0) *reinterpret_cast<f32x4*>(csx + (w * 4 + g) * 4) = f32x4{cs[0], cs[1], cs[2], cs[3]};
  __builtin_amdgcn_s_barrier();
  if (h == 0 && i == 0) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(csx + (w * 4 + g) * 4);
    *reinterpret_cast<f32x4*>(p.colsum + ((int64_t)sl * B + b) * K + 16 * w + 4 * g) =
        f32x4{cs[0] + o[0], cs[1] + o[1], cs[2] + o[2], cs[3] + o[3]};
  }
  if constexpr (STAMPS) {
    VF_STAMP(29);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VF_STAMP(30);
  }
#undef VF_STAMP
}

// vlad_bwd8_kernel: the backward twin with eight waves (see vlad_bwd_kernel for the algebra):
// wave (w, h) contracts x with the image's dU^T over its channel half, hands the other tile's
// partial to its partner, runs the softmax backward of its 16 locations (partial sums over its 16
// clusters exchanged across the four cluster groups) and aggregates its 16 channel tiles of dW.
__global__ __launch_bounds__(512) void vlad_bwd8_kernel(VladBwdArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char vf_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int w = wid & 3, h = wid >> 2;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sl = blockIdx.y, B = gridDim.x;
  const int nsteps_img = (p.N + VF_STEP - 1) / VF_STEP;
  const int st_lo = sl * p.steps_per_slice;
  const int st_hi = st_lo + p.steps_per_slice < nsteps_img ? st_lo + p.steps_per_slice : nsteps_img;
  const int nst = st_hi - st_lo;
  const unsigned lds0 = nv_lds_byte_of(vf_lds);
  const unsigned cf0 = lds0 + VF_NST * VF_STAGE + w * VF_CF;
  unsigned char* xl = vf_lds + VF_NST * VF_STAGE + 4 * VF_CF;
  float* exch = reinterpret_cast<float*>(xl + V8_XL + V8_PN);     // [tile][cluster group][loc][4]
  const unsigned short* xb = p.x + (int64_t)b * p.N * D;

  auto stage = [&](int step) {
    const unsigned base = lds0 + (unsigned)((step - st_lo) % VF_NST) * VF_STAGE;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = wid + 8 * v;                           // (wave-uniform: the row base is scalar)
      int n = VF_STEP * step + r;
      n = n < p.N ? n : p.N - 1;
      nv_glds16s(xb + (int64_t)n * D, (unsigned)lane, (unsigned)(r & 15), base + r * 1024);
    }
  };
  stage(st_lo);
  u32x4 wf[8][VF_NPL];
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.duimg) +
                       (((int64_t)b * 4 + w) * 16 + 8 * h) * VF_NPL * 64 + lane;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) wf[s][pl] = src[(s * VF_NPL + pl) * 64];
  }
  const f32x4 cd = *reinterpret_cast<const f32x4*>(p.cdu + b * K + 16 * w + 4 * g);
  if (nst > 1) stage(st_lo + 1);

  f32x4 accv[16];
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) accv[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned rowoff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) rowoff[k] = 1024u * i + 512u * h + 64u * (k ^ (i >> 2)) + 16u * (g ^ (i & 3));
  const int q = (lane >> 2) & 3, pp = lane & 3;
  unsigned troff[2], cfoff[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = 16 * (g >> 1) + 2 * (4 * (g & 1) + q) + hh;
    const int R = row & 15;
    troff[hh] = 1024u * row + 512u * h + 32u * (R >> 1) + 16u * ((pp >> 1) ^ (R & 1)) + 8u * (pp & 1);
    cfoff[hh] = (unsigned)row * VF_CFLD + 8u * pp;
  }

#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int step = st_lo + st;
    // younger than this stage's DMA and allowed in flight: the next stage's DMA (4) and the
    // previous step's two stores (its loads were waited for when they were used)
    v8_wait_vm((st + 1 < nst ? 4 : 0) + (st >= 1 ? 2 : 0));
    __builtin_amdgcn_s_barrier();
    const unsigned sb = lds0 + (unsigned)(st % VF_NST) * VF_STAGE;

    // the saved forward values of the wave's tile, in flight under the matrix work
    const int n = VF_STEP * step + 16 * h + i;
    const bool ok = n < p.N;
    const int64_t row = (int64_t)b * p.N + (ok ? n : p.N - 1);
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.a + row * K + 16 * w + 4 * g);
    const float rn2 = p.rn[row];

    f32x4 accl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) accl[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 xf[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        xf[s][t] = vf_ldsr128(sb + rowoff[s & 3] + 256u * (s >> 2) + 16384u * t);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s + 2 < 8) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          xf[(s + 2) & 3][t] =
              vf_ldsr128(sb + rowoff[(s + 2) & 3] + 256u * ((s + 2) >> 2) + 16384u * t);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < VF_NPL; ++pl) accl[t] = mfma16b(wf[s][pl], xf[s & 3][t], accl[t]);
    }
    *reinterpret_cast<f32x4*>(xl + (wid * 64 + lane) * 16) = h ? accl[0] : accl[1];
    __builtin_amdgcn_s_barrier();

    // Round 6: the logits are not saved.  They enter the backward pass only through the row-norm
    // term sum_k ds_k l_k with sum_k ds_k = 0, so any l_k + const serves: l_k = log a_k (the logit
    // minus the location's log-sum-exp).  What the constant would have contributed is
    // lse * dot * (1 - sum_k a_k), i.e. rounding-level.  a_k == 0 (underflow) carries ds_k == 0:
    // its term is dropped instead of becoming 0 * -inf.
    f32x4 l4;
#pragma unroll
    for (int j = 0; j < 4; ++j) l4[j] = a4[j] > 1.0e-37f ? __logf(a4[j]) : 0.f;

    float tv[4], dav[4];
    {
      const f32x4 own = h ? accl[1] : accl[0];
      const f32x4 oth = *reinterpret_cast<const f32x4*>(xl + ((wid ^ 4) * 64 + lane) * 16);
      const f32x4 lo4 = h ? oth : own, hi4 = h ? own : oth;
      float p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tv[j] = (lo4[j] + hi4[j]) * rn2;                    // xhat . dU
        dav[j] = tv[j] + cd[j];
        const float ad = a4[j] * dav[j];
        p1 += ad;
        p2 = fmaf(a4[j], tv[j], p2);
        p3 = fmaf(ad, l4[j], p3);
        p4 = fmaf(a4[j], l4[j], p4);
      }
      p1 = vf_gsum(p1);
      p2 = vf_gsum(p2);
      p3 = vf_gsum(p3);
      p4 = vf_gsum(p4);
      if (g == 0) *reinterpret_cast<f32x4*>(exch + ((h * 4 + w) * 16 + i) * 4) = f32x4{p1, p2, p3, p4};
    }
    __builtin_amdgcn_s_barrier();
    {
      f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) ps += *reinterpret_cast<const f32x4*>(exch + ((h * 4 + w2) * 16 + i) * 4);
      const float dot = ps[0];
      const float rd2 = ps[1] + ps[2] - dot * ps[3];
      float dsv[4];
      unsigned short hh[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dsv[j] = a4[j] * (dav[j] - dot);
        split3_bf16(ok ? dsv[j] * rn2 : 0.f, hh[0][j], hh[1][j], hh[2][j]);
      }
      *reinterpret_cast<f32x4*>(ok ? p.ds + ((int64_t)b * p.N + n) * K + 16 * w + 4 * g
                                   : p.trash + 4 * lane) = f32x4{dsv[0], dsv[1], dsv[2], dsv[3]};
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl)
        vf_ldsw64(cf0 + pl * VF_CFPL + (16 * h + i) * VF_CFLD + 8 * g,
                  (unsigned)hh[pl][0] | ((unsigned)hh[pl][1] << 16),
                  (unsigned)hh[pl][2] | ((unsigned)hh[pl][3] << 16));
      {   // rowdot: wave (w, h) writes locations 4 w .. 4 w + 3 of its tile
        const bool mine = g == 0 && (i >> 2) == w && ok;
        *(mine ? p.rowdot + (int64_t)b * p.N + n : p.trash + 4 * lane) = rd2;
      }
    }
    __builtin_amdgcn_s_barrier();
    if (st + 2 < nst) stage(step + 2);

    u32x4 bfr[VF_NPL];
#pragma unroll
    for (int pl = 0; pl < VF_NPL; ++pl) {
      const uint2 lo = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[0]);
      const uint2 hi = vf_ldsr_tr(cf0 + pl * VF_CFPL + cfoff[1]);
      bfr[pl] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
    const unsigned tb0 = sb + troff[0], tb1 = sb + troff[1];
    u32x4 af[8];
#pragma unroll
    for (int c = 0; c < V8_AHEAD; ++c) {
      const uint2 lo = vf_ldsr_tr((tb0 ^ (32u * (c & 7))) + 32u * (c & 8)),
                  hi = vf_ldsr_tr((tb1 ^ (32u * (c & 7))) + 32u * (c & 8));
      af[c] = u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c + V8_AHEAD < 16) {
        const int cn = c + V8_AHEAD;
        const uint2 lo = vf_ldsr_tr((tb0 ^ (32u * (cn & 7))) + 32u * (cn & 8)),
                    hi = vf_ldsr_tr((tb1 ^ (32u * (cn & 7))) + 32u * (cn & 8));
        af[cn & 7] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl) accv[c] = mfma16b(af[c & 7], bfr[pl], accv[c]);
    }
  }
  f32x4* slab =
      reinterpret_cast<f32x4*>(p.slab) + ((((int64_t)sl * B + b) * 4 + w) * 32 + 16 * h) * 64 + lane;
#pragma unroll
  for (int c = 0; c < 16; ++c) slab[c * 64] = accv[c];
}

// grad_w[d,k] = sum over all (slice, image) slabs, in two levels so that every CU reads its share
// of the 31 MB: vlad_wgrad_partial_kernel, grid (32 channel tiles, VW_GROUPS), sums one group's
// slabs (VF_MAXS loads in flight) into partial[group][...]; vlad_wgrad_finish_kernel, grid 32,
// adds the groups in order and forms grad_c[d,k] = sum_b dU[b,d,k] * asum[b,k].  Fixed orders
// throughout: bitwise reproducible.  Slabs in accumulator order (see vlad_finish_sum_kernel).
constexpr int VW_GROUPS = 8;
#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256) void vlad_wgrad_partial_kernel(const float* __restrict__ slab,
                                                                 int total,
                                                                 float* __restrict__ partial) {
  const int ct = blockIdx.x, grp = blockIdx.y;
  const int per = (total + VW_GROUPS - 1) / VW_GROUPS;
  const int lo = grp * per, hi = lo + per < total ? lo + per : total;
  const int64_t unit = ((int64_t)(threadIdx.x >> 6) * 32 + ct) * 64 + (threadIdx.x & 63);
  const f32x4* src = reinterpret_cast<const f32x4*>(slab) + unit;
  const int64_t stride = (int64_t)4 * 32 * 64;                // units between (slice, image) slabs
  f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = lo; s0 < hi; s0 += VF_MAXS) {
    f32x4 v[VF_MAXS];
#pragma unroll
    for (int s = 0; s < VF_MAXS; ++s)
      v[s] = s0 + s < hi ? src[(s0 + s) * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < VF_MAXS; ++s) u += v[s];
  }
  reinterpret_cast<f32x4*>(partial)[grp * stride + unit] = u;
}
#endif   // SCL_DIAG

#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256) void vlad_wgrad_finish_kernel(const float* __restrict__ partial,
                                                                const float* __restrict__ du,
                                                                const float* __restrict__ save_vlad,
                                                                int B, float* __restrict__ grad_w,
                                                                float* __restrict__ grad_c) {
  const int ct = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int k = 16 * w + i;
  const int64_t stride = (int64_t)4 * 32 * 64;
  const f32x4* src = reinterpret_cast<const f32x4*>(partial) + ((int64_t)w * 32 + ct) * 64 + lane;
  f32x4 v[VW_GROUPS];
#pragma unroll
  for (int s = 0; s < VW_GROUPS; ++s) v[s] = src[s * stride];
  f32x4 u = v[0];
#pragma unroll
  for (int s = 1; s < VW_GROUPS; ++s) u += v[s];
  float gc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b0 = 0; b0 < B; b0 += 8) {
    float as8[8], dv[8][4];
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int b = b0 + bb < B ? b0 + bb : B - 1;
      as8[bb] = b0 + bb < B ? save_vlad[((int64_t)b * VROWS + D) * K + k] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) dv[bb][j] = du[((int64_t)b * D + 16 * ct + 4 * g + j) * K + k];
    }
#pragma unroll
    for (int bb = 0; bb < 8; ++bb)
#pragma unroll
      for (int j = 0; j < 4; ++j) gc[j] = fmaf(dv[bb][j], as8[bb], gc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int d = 16 * ct + 4 * g + j;
    grad_w[d * K + k] = u[j];
    grad_c[d * K + k] = gc[j];
  }
}
#endif   // SCL_DIAG

// ---------------------------------------------------------------------------------------
// Sibling exchange inside a launch (round 4).  The forward finish and the backward prologue each
// need, per image, ONE or TWO sums over all 64 clusters (the global norm; <grad_out, Vn>) between
// two passes over the image's [512][64] block — which used to be a kernel boundary each
// (finish_sum | finish_norm, bwd_dots | bwd_du: four launches at the 4-6 us floor of a dependent
// do-almost-nothing kernel).  Both are now ONE kernel of 8 workgroups per image, workgroup c owning
// the clusters 8 c .. 8 c + 7 of ALL 512 channels, so that every per-cluster quantity is local and
// only the workgroup's partial of the global scalars travels: an 8-byte {tag = 1, value} granule
// written by ONE agent-scope (sc1, write-through) store — the data is the flag, no fence on either
// side (MI355X_MICROARCH.md, hand-off price list, row handoff-1to1; cdna_hip_programming.md
// Guideline 16 R2).  Wave 0 of each workgroup re-reads the image's granules until all tags are set.
// The spin is BOUNDED: a workgroup whose siblings are not resident in time computes their scalars
// itself with the same routine (same instruction sequence, same bits) — correctness never depends on
// co-residency or dispatch order, only speed does.  Granules are zeroed by the kernel in front
// (forward: vlad_fwd_kernel, same call) or kept zero between calls (backward: row 513 of save_vlad,
// zeroed by the forward pass and re-zeroed by the last workgroup of the prologue to leave).
typedef unsigned long long gran_t;
typedef __attribute__((address_space(1))) gran_t* gran_gptr;
typedef __attribute__((address_space(1))) unsigned* u32_gptr;
__device__ __forceinline__ void gran_put(gran_t* g, float v) {
  __hip_atomic_store((gran_gptr)g, (1ull << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// wave-wide (call it from ONE wave): lanes 0 .. n - 1 poll granule `lane` until every tag is set;
// v = the lane's value.  false after `limit` polls.
__device__ __forceinline__ bool gran_sweep(gran_t* g, int n, int limit, float& v) {
  const int lane = threadIdx.x & 63;
  for (int spins = 0;; ++spins) {
    const gran_t x = lane < n ? __hip_atomic_load((gran_gptr)g + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                              : (1ull << 32);
    v = __uint_as_float((unsigned)x);
    if (__all((unsigned)(x >> 32) == 1u)) return true;
    if (spins >= limit) return false;
    __builtin_amdgcn_s_sleep(8);
  }
}
constexpr int kSpinLimit = 20000;          // x (poll + s_sleep 8) ~ several ms: never reached in practice
// scl_debug_set_variant(921): no patience at all — every workgroup that does not find its siblings'
// granules at the first poll takes the self-computing path (tests: same bits either way)
inline int spin_limit() { return scl_variant() == 921 ? 0 : kSpinLimit; }

// vlad_finish_kernel: everything behind the fused forward kernel in ONE launch — U = sum of the
// image's slice slabs + C * asum, the intra-normalisation per cluster, the global normalisation,
// out = U q g, and the saved pre-norm VLAD (rows 0..511 U, 512 asum, 513 zeroed sync words of the
// backward prologue).  grid (8 cluster groups, B), block 256: thread (ip = cluster in the group,
// g, cs) sums the 16-byte slab units of channel tiles ct = cs + 8 r, r = 0..3 (unit ((w * 32 + ct)
// * 64 + 16 g + i) = U[16 ct + 4 g + 0..3][16 w + i]: a workgroup reads one 128-byte half of every
// 256-byte row), keeps its 16 values of U in registers across the exchange and writes both outputs
// from them: U is neither re-read nor does any value make a round trip through memory.
struct VladFinishArgs {
  const float* slab;      // [S][B][8192 units][4]
  const float* colsum;    // [S][B][64]
  const float* centers;   // [512][64]
  int S, B;
  float* vlad;            // [B][VROWS][64]
  float* out;             // [B][32768]
  gran_t* gran;           // [B][8], zero on entry
  int spin_limit;
};

constexpr int VFIN_NB = 10;                // slices in flight per thread (x 4 units): the kernel is
                                           // one round trip long, everything it needs must be requested at once
__global__ __launch_bounds__(256) void vlad_finish_kernel(VladFinishArgs p) {
  __shared__ float fin_lds[4 * 8 + 8 + 8];   // [wave][ip] partial column sums | siblings' values | flag
  float* red = fin_lds;
  float* vals = fin_lds + 32;
  int* okf = reinterpret_cast<int*>(fin_lds + 40);
  // (Round 5 measured the transposed grid (B, 8) — the eight workgroups of an image on the XCD whose
  // workgroups wrote its slabs when B % 8 == 0: 12.7 -> 11.7 us at 24 images, and 14 MILLISECONDS at
  // 96, where an image's siblings are dispatched 96 workgroups apart and every exchange runs into
  // its spin limit.  Siblings must be neighbours in dispatch order: cluster group on x.)
  const int c = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ip = t & 7, g = (t >> 3) & 3, cs = t >> 5;

  // the cluster group's pass over the slabs: u = its 16 values of U, asum, q of cluster 8 c2 + ip;
  // returns the group's share of sum_k q_k^2 col_k (the same value in every thread)
  auto group = [&](int c2, f32x4 (&u)[4], float& asum_out, float& q_out) -> float {
    const int w = c2 >> 1, k = 8 * c2 + ip;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.slab) + (int64_t)b * 8192 + (w * 32 + cs) * 64 + 16 * g +
                       8 * (c2 & 1) + ip;
    const int64_t sstride = (int64_t)p.B * 8192;               // units between slices
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    // The kernel is ONE memory round trip long if everything a thread needs is requested before
    // anything is consumed (left to itself hipcc keeps ~10 loads in flight and waits between them:
    // four round trips for the 40 slab units): the centres first, then the slabs, then a scheduling
    // fence.
    float cen[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) cen[r][j] = p.centers[(16 * (cs + 8 * r) + 4 * g + j) * K + k];
    for (int s0 = 0; s0 < p.S; s0 += VFIN_NB) {
      f32x4 v[VFIN_NB][4];
      float a[VFIN_NB];
#pragma unroll
      for (int s = 0; s < VFIN_NB; ++s) {                      // branch-free: past the end re-reads
        const int sc = s0 + s < p.S ? s0 + s : p.S - 1;        // the last slice and drops it
#pragma unroll
        for (int r = 0; r < 4; ++r) v[s][r] = src[sc * sstride + r * 512];
        a[s] = p.colsum[((int64_t)sc * p.B + b) * K + k];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < VFIN_NB; ++s) {                      // fixed order: bitwise reproducible
        const float keep = s0 + s < p.S ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += v[s][r] * keep;
        asum += a[s] * keep;
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[r][j] = acc[r][j] + cen[r][j] * asum;
        ss = fmaf(u[r][j], u[r][j], ss);
      }
    // over the 32 threads that share ip: lane bits 3..5, then the four waves
    ss += __shfl_xor(ss, 8, 64);
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    __syncthreads();
    if (lane < 8) red[wv * 8 + ip] = ss;
    __syncthreads();
    const float col = (red[ip] + red[8 + ip]) + (red[16 + ip] + red[24 + ip]);
    // matconvnetNormalize: x / sqrt(sum x^2 + 1e-12), epsilon inside the sqrt
    const float q = 1.0f / sqrtf(col + 1e-12f);
    float tq = q * q * col;
    tq += __shfl_xor(tq, 1, 64);
    tq += __shfl_xor(tq, 2, 64);
    tq += __shfl_xor(tq, 4, 64);
    asum_out = asum;
    q_out = q;
    return tq;
  };

  f32x4 u[4];
  float asum, q;
  const float mine = group(c, u, asum, q);
  if (t == 0) gran_put(p.gran + b * 8 + c, mine);
  if (wv == 0) {
    float v;
    const bool ok = gran_sweep(p.gran + b * 8, 8, p.spin_limit, v);
    if (lane < 8) vals[lane] = v;
    if (lane == 0) *okf = ok ? 1 : 0;
  }
  __syncthreads();
  if (!*okf) {                                                 // (uniform) siblings not seen in time
    for (int c2 = 0; c2 < 8; ++c2) {
      f32x4 u2[4];
      float a2, q2;
      const float tv = group(c2, u2, a2, q2);
      if (t == 0) vals[c2] = tv;
    }
    __syncthreads();
  }
  float tot = 0.f;
#pragma unroll
  for (int c2 = 0; c2 < 8; ++c2) tot += vals[c2];
  const float gn = 1.0f / sqrtf(tot + 1e-12f);
  const int k = 8 * c + ip;
  float* vrow = p.vlad + (int64_t)b * VROWS * K;
  float* orow = p.out + (int64_t)b * D * K;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = 16 * (cs + 8 * r) + 4 * g + j;
      vrow[d * K + k] = u[r][j];
      orow[d * K + k] = u[r][j] * q * gn;
    }
  if (cs == 0 && g == 0) vrow[D * K + k] = asum;
  if (c == 0 && t < K) reinterpret_cast<unsigned*>(vrow)[(D + 1) * K + t] = 0u;   // the backward's sync words
}

// vlad_bwd_prologue_kernel: the gradient through both normalisations (bwd_dots_kernel's comment
// has the closed form) in ONE launch, 8 workgroups per image, workgroup c = clusters 8 c .. + 7 of
// all 512 channels: A_k, col_k, Bc_k, Dc_k are local; the two sums over all clusters, tot = sum_k
// q_k^2 col_k and S1 = sum_k q_k A_k, travel as granules (above).  Thread (ip, r): cluster 8 c + ip,
// 8-channel chunks r and r + 32; its 16 values of U and grad_out stay in registers for dU.  Outputs as
// bwd_du_kernel: dU float32 [d][k] (+ transposed for the float32-MFMA kernels), c.dU, and for bf16
// feature maps the two register images of the fused kernels (duimg straight from the registers —
// 8 consecutive channels of a cluster are one thread's — dximg through a [512][8] LDS tile).
struct VladProArgs {
  float* save_vlad;          // [B][VROWS][64]; row 513: 16 granules + the leave counter (word 32)
  const float* grad_out;     // [B][32768]
  const float* centers;
  float* du;                 // [B][512][64]
  float* dut;                // [B][64][512] or NULL
  unsigned short* duimg;     // or NULL
  unsigned short* dximg;
  float* cdu;                // [B][64]
  int spin_limit;
};

__global__ __launch_bounds__(256) void vlad_bwd_prologue_kernel(VladProArgs p) {
  __shared__ __attribute__((aligned(16))) float pro_lds[512 * 8 + 4 * 4 * 8 + 16 + 8];
  float* tile = pro_lds;                       // [512 ch][8 clusters]
  float* red = pro_lds + 512 * 8;              // [wave][dot][ip]
  float* vals = red + 4 * 4 * 8;               // the image's 16 granule values
  int* okf = reinterpret_cast<int*>(vals + 16);
  const int c = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ip = t & 7, r = t >> 3;
  const float* urow = p.save_vlad + (int64_t)b * VROWS * K;
  const float* grow = p.grad_out + (int64_t)b * D * K;
  gran_t* gran = reinterpret_cast<gran_t*>(p.save_vlad + ((int64_t)b * VROWS + D + 1) * K);
  unsigned* leave = reinterpret_cast<unsigned*>(p.save_vlad + ((int64_t)b * VROWS + D + 1) * K) + 32;

  // the four column dots of cluster 8 c2 + ip; (tot, S1) shares of the group (same in every thread)
  auto group = [&](int c2, float (&uu)[16], float (&gg)[16], float (&dots)[4], float& tq, float& sq) {
    const int k = 8 * c2 + ip;
    float cc[16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 8 * (r + 32 * h) + e;
        uu[8 * h + e] = urow[d * K + k];
        gg[8 * h + e] = grow[d * K + k];
        cc[8 * h + e] = p.centers[d * K + k];
      }
    __builtin_amdgcn_sched_barrier(0);                        // all 48 loads requested before the first use
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[0] = fmaf(gg[e], uu[e], s[0]);
      s[1] = fmaf(uu[e], uu[e], s[1]);
      s[2] = fmaf(gg[e], cc[e], s[2]);
      s[3] = fmaf(uu[e], cc[e], s[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                             // threads sharing ip: lane bits 3..5
      s[j] += __shfl_xor(s[j], 8, 64);
      s[j] += __shfl_xor(s[j], 16, 64);
      s[j] += __shfl_xor(s[j], 32, 64);
    }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wv * 4 + j) * 8 + ip] = s[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      dots[j] = (red[(0 * 4 + j) * 8 + ip] + red[(1 * 4 + j) * 8 + ip]) +
                (red[(2 * 4 + j) * 8 + ip] + red[(3 * 4 + j) * 8 + ip]);
    const float qq = 1.0f / sqrtf(dots[1] + 1e-12f);
    tq = qq * qq * dots[1];
    sq = qq * dots[0];
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      tq += __shfl_xor(tq, m, 64);
      sq += __shfl_xor(sq, m, 64);
    }
  };

  float uu[16], gg[16], dots[4], tq, sq;
  group(c, uu, gg, dots, tq, sq);
  if (t == 0) {
    gran_put(gran + 2 * c, tq);
    gran_put(gran + 2 * c + 1, sq);
  }
  if (wv == 0) {
    float v;
    const bool ok = gran_sweep(gran, 16, p.spin_limit, v);
    if (lane < 16) vals[lane] = v;
    if (lane == 0) *okf = ok ? 1 : 0;
  }
  __syncthreads();
  if (!*okf) {                                                 // (uniform) compute the siblings' shares here
    for (int c2 = 0; c2 < 8; ++c2) {
      float u2[16], g2[16], d2[4], t2, s2;
      group(c2, u2, g2, d2, t2, s2);
      if (t == 0) {
        vals[2 * c2] = t2;
        vals[2 * c2 + 1] = s2;
      }
    }
    __syncthreads();
  }
  float tot = 0.f, s1 = 0.f;
#pragma unroll
  for (int c2 = 0; c2 < 8; ++c2) {
    tot += vals[2 * c2];
    s1 += vals[2 * c2 + 1];
  }
  const int k = 8 * c + ip;
  const float ak = dots[0], col = dots[1], bk = dots[2], dk = dots[3];
  const float q = 1.0f / sqrtf(col + 1e-12f);
  const float gn = 1.0f / sqrtf(tot + 1e-12f);
  const float rr = gn * (q * ak - gn * gn * s1 * q * q * col);
  const float cu = q * q * (gn * gn * gn * s1 + rr);   // coefficient of U
  const float cg = q * gn;                             // coefficient of grad_out
  float vv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) vv[e] = cg * gg[e] - cu * uu[e];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) p.du[((int64_t)b * D + 8 * (r + 32 * h) + e) * K + k] = vv[8 * h + e];
  if (r == 0) p.cdu[b * K + k] = cg * bk - cu * dk;
  if (p.dut) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float* trow = p.dut + ((int64_t)b * K + k) * D + 8 * (r + 32 * h);
      *reinterpret_cast<f32x4*>(trow) = f32x4{vv[8 * h], vv[8 * h + 1], vv[8 * h + 2], vv[8 * h + 3]};
      *reinterpret_cast<f32x4*>(trow + 4) = f32x4{vv[8 * h + 4], vv[8 * h + 5], vv[8 * h + 6], vv[8 * h + 7]};
    }
  }
  if (p.duimg) {                                               // (uniform over the launch)
    // channels 8 ch8 + e of cluster k: unit ((w * 16 + s) * NPL + plane) * 64 + 16 gq + i with
    // w = k >> 4, i = k & 15, s = ch8 >> 2, gq = ch8 & 3
    uint4* img = reinterpret_cast<uint4*>(p.duimg + (int64_t)b * VF_NPL * D * K);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ch8 = r + 32 * h;
      unsigned short hh[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3_bf16(vv[8 * h + e], hh[0][e], hh[1][e], hh[2][e]);
#pragma unroll
      for (int pl = 0; pl < VF_NPL; ++pl)
        img[(((k >> 4) * 16 + (ch8 >> 2)) * VF_NPL + pl) * 64 + 16 * (ch8 & 3) + (k & 15)] =
            make_uint4((unsigned)hh[pl][0] | ((unsigned)hh[pl][1] << 16), (unsigned)hh[pl][2] | ((unsigned)hh[pl][3] << 16),
                       (unsigned)hh[pl][4] | ((unsigned)hh[pl][5] << 16), (unsigned)hh[pl][6] | ((unsigned)hh[pl][7] << 16));
    }
    // the other orientation: 8 consecutive clusters (= this workgroup's) of one channel
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[(8 * (r + 32 * h) + e) * 8 + ip] = vv[8 * h + e];
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int ch = t + 256 * rep;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(tile + ch * 8), v1 = *reinterpret_cast<const f32x4*>(tile + ch * 8 + 4);
      u32x4 hi, lo;
      split2x8(v0, v1, hi, lo);
      // unit (((wv2 * 8 + nt) * 2 + s) * 2 + plane) * 64 + 16 gq + i: channel 128 wv2 + 16 nt + i,
      // clusters 32 s + 8 gq .. + 7 = 8 c ..
      u32x4* img = reinterpret_cast<u32x4*>(p.dximg) + (int64_t)b * (4 * 8 * 2 * 2 * 64) +
                   ((((ch >> 4) * 2 + (c >> 2)) * 2) * 64) + 16 * (c & 3) + (ch & 15);
      img[0] = hi;
      img[64] = lo;
    }
  }
  // last workgroup of the image to get here: every sibling has read the granules — zero them for the
  // next backward pass over the same saved tensors
  if (t == 0) {
    const unsigned old = __hip_atomic_fetch_add((u32_gptr)leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 7u) {
#pragma unroll
      for (int j = 0; j < 16; ++j)
        __hip_atomic_store((gran_gptr)gran + j, (gran_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((u32_gptr)leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Forward finish in two small launches of 8 x B workgroups (a single workgroup per image
// left 232 CUs idle and took 3x longer):
//   finish_sum_kernel   U = slabs + C * asum for one 64-channel block -> vlad[b] (the saved
//                       pre-norm VLAD) and the block's column sums of squares
//   finish_norm_kernel  q_k = rsqrt(col_k + eps), g = rsqrt(sum_k q_k^2 col_k + eps) — the
//                       global norm needs only the column sums — then out = U q g
// grid (8, B); block 256: thread -> k = t & 63, dq = t >> 6 (4 groups of 16 channels).
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ part,
                                                         const float* __restrict__ colsum_part,
                                                         const float* __restrict__ centers,
                                                         float* __restrict__ vlad,
                                                         float* __restrict__ colsq_part) {
  __shared__ float colbuf[4 * 64];
  const int blk = blockIdx.x, b = blockIdx.y, k = threadIdx.x & 63, dq = threadIdx.x >> 6;
  float asum = 0.f;
#pragma unroll
  for (int s = 0; s < NSPLIT; ++s) asum += colsum_part[((int64_t)b * NSPLIT + s) * K + k];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = blk * 64 + dq * 16 + i;
    float p = 0.f;
#pragma unroll
    for (int s = 0; s < NSPLIT; ++s) p += part[(((int64_t)b * NSPLIT + s) * D + d) * K + k];
    const float u = p + centers[d * K + k] * asum;
    vlad[((int64_t)b * VROWS + d) * K + k] = u;
    ss = fmaf(u, u, ss);
  }
  if (blk == 0 && dq == 0) vlad[((int64_t)b * VROWS + D) * K + k] = asum;
  colbuf[dq * 64 + k] = ss;
  __syncthreads();
  if (dq == 0)
    colsq_part[((int64_t)b * 8 + blk) * K + k] =
        (colbuf[k] + colbuf[64 + k]) + (colbuf[128 + k] + colbuf[192 + k]);
}

__global__ __launch_bounds__(256) void finish_norm_kernel(float* vlad,
                                                          const float* __restrict__ colsq_part,
                                                          int nparts, float* __restrict__ out) {
  const int blk = blockIdx.x, b = blockIdx.y, k = threadIdx.x & 63, dq = threadIdx.x >> 6;
  // row 513 of the saved VLAD: sync words of the backward prologue, zero between calls
  if (blk == 0 && dq == 0) reinterpret_cast<unsigned*>(vlad)[((int64_t)b * VROWS + D + 1) * K + k] = 0u;
  float cp[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) cp[j] = j < nparts ? colsq_part[((int64_t)b * nparts + j) * K + k] : 0.f;
  float col = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) col += cp[j];
  // matconvnetNormalize: x / sqrt(sum x^2 + 1e-12), epsilon inside the sqrt
  const float q = 1.0f / sqrtf(col + 1e-12f);
  // every one of the 4 waves holds all 64 columns: a wave sum gives sum_k (q_k^2 col_k)
  const float tot = wave_sum(q * q * col);
  const float g = 1.0f / sqrtf(tot + 1e-12f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = blk * 64 + dq * 16 + i;
    out[(int64_t)b * D * K + d * K + k] = vlad[((int64_t)b * VROWS + d) * K + k] * q * g;
  }
}

// Gradient through the global and the intra normalisation, in closed form from four column
// dots so that 8 x B workgroups can work on an image (one 1024-thread workgroup per image
// took 17.7 us at B = 24).  With U the pre-norm VLAD, q_k = rsqrt(col_k + eps),
// Vn = U q, g = rsqrt(sum_k q_k^2 col_k + eps), out = Vn g and go = d loss / d out:
//   A_k = sum_d go U,  col_k = sum_d U^2,  Bc_k = sum_d go C,  Dc_k = sum_d U C
//   S1 = sum_k q_k A_k                       (= <go, Vn>)
//   r_k = g (q_k A_k - g^2 S1 q_k^2 col_k)   (= <dVn, Vn>_k)
//   dU  = q_k g go - q_k^2 (g^3 S1 + r_k) U
//   c.dU_k = q_k g Bc_k - q_k^2 (g^3 S1 + r_k) Dc_k
// bwd_dots_kernel: grid (8, B), block 256 (k = t & 63, dq = t >> 6): the four partial dots of
// one 64-channel block -> dots[b][blk][4][64].
#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256) void bwd_dots_kernel(const float* __restrict__ save_vlad,
                                                       const float* __restrict__ grad_out,
                                                       const float* __restrict__ centers,
                                                       float* __restrict__ dots) {
  __shared__ float buf[4][4 * 64];
  const int blk = blockIdx.x, b = blockIdx.y, k = threadIdx.x & 63, dq = threadIdx.x >> 6;
  float sa = 0.f, sc = 0.f, sb = 0.f, sd = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = blk * 64 + dq * 16 + i;
    const float u = save_vlad[((int64_t)b * VROWS + d) * K + k];
    const float go = grad_out[(int64_t)b * D * K + d * K + k];
    const float c = centers[d * K + k];
    sa = fmaf(go, u, sa);
    sc = fmaf(u, u, sc);
    sb = fmaf(go, c, sb);
    sd = fmaf(u, c, sd);
  }
  buf[0][dq * 64 + k] = sa;
  buf[1][dq * 64 + k] = sc;
  buf[2][dq * 64 + k] = sb;
  buf[3][dq * 64 + k] = sd;
  __syncthreads();
  // wave dq finishes dot number dq
  const float* src = buf[dq];
  dots[(((int64_t)b * 8 + blk) * 4 + dq) * K + k] =
      (src[k] + src[64 + k]) + (src[128 + k] + src[192 + k]);
}
#endif   // SCL_DIAG

// bwd_du_kernel: grid (8, B), block 256: dU of one 64-channel block as float32 [d][k] (and
// transposed, for the float32-MFMA row-tile kernel), c.dU from block 0, and for bf16 feature maps
// the two REGISTER IMAGES the fused kernels load with coalesced 16-byte reads:
//   duimg  (vlad_bwd_kernel): unit ((w * 16 + s) * VF_NPL + plane) * 64 + lane =
//          dU_plane[ch 32 s + 8 g + e][cluster 16 w + i] — eight consecutive CHANNELS of a
//          cluster: what this thread holds in registers;
//   dximg  (vlad_dx_kernel):  unit (((w * 8 + nt) * 2 + s) * 2 + plane) * 64 + lane =
//          dU_plane[ch 128 w + 16 nt + i][cluster 32 s + 8 g + e] — eight consecutive CLUSTERS
//          of a channel: the other orientation, through a [64 ch][64 k] LDS tile;
//   wdximg (image 0's workgroups): the same image of W, shared by all images.
constexpr int DU_TLD = 65;                                // floats per tile row (conflict-free columns)
#ifdef SCL_DIAG   // superseded: reachable through scl_debug_set_variant only (A/B, parity runs)
__global__ __launch_bounds__(256) void bwd_du_kernel(const float* __restrict__ save_vlad,
                                                     const float* __restrict__ grad_out,
                                                     const float* __restrict__ dots,
                                                     float* __restrict__ du,
                                                     float* __restrict__ dut,
                                                     unsigned short* __restrict__ duimg,
                                                     unsigned short* __restrict__ dximg,
                                                     const float* __restrict__ assign_w,
                                                     unsigned short* __restrict__ wdximg,
                                                     float* __restrict__ cdu) {
  __shared__ float tile[64 * DU_TLD];
  const int blk = blockIdx.x, b = blockIdx.y, k = threadIdx.x & 63, dq = threadIdx.x >> 6;
  float dot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float s = 0.f;
#pragma unroll
    for (int bl = 0; bl < 8; ++bl) s += dots[(((int64_t)b * 8 + bl) * 4 + j) * K + k];
    dot[j] = s;
  }
  const float ak = dot[0], col = dot[1], bk = dot[2], dk = dot[3];
  const float q = 1.0f / sqrtf(col + 1e-12f);
  const float tot = wave_sum(q * q * col);
  const float g = 1.0f / sqrtf(tot + 1e-12f);
  const float s1 = wave_sum(q * ak);
  const float r = g * (q * ak - g * g * s1 * q * q * col);
  const float cu = q * q * (g * g * g * s1 + r);   // coefficient of U
  const float cg = q * g;                          // coefficient of go
  float vals[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = blk * 64 + dq * 16 + i;
    const float u = save_vlad[((int64_t)b * VROWS + d) * K + k];
    const float go = grad_out[(int64_t)b * D * K + d * K + k];
    vals[i] = cg * go - cu * u;
    du[((int64_t)b * D + d) * K + k] = vals[i];
  }
  if (blk == 0 && dq == 0) cdu[b * K + k] = cg * bk - cu * dk;
  if (dut) {   // transposed float32 copy (float32-MFMA row-tile kernel only)
    float* trow = dut + ((int64_t)b * K + k) * D + blk * 64 + dq * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 4)
      *reinterpret_cast<f32x4*>(trow + i) = f32x4{vals[i], vals[i + 1], vals[i + 2], vals[i + 3]};
  }
  if (!duimg) return;                                      // (uniform over the launch)
  {
    // this thread's 16 channels are k-step s = 2 blk + (dq >> 1), lane groups g = 2 (dq & 1) and
    // + 1, of wave k >> 4
    unsigned short h[3][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) split3_bf16(vals[i], h[0][i], h[1][i], h[2][i]);
    uint4* img = reinterpret_cast<uint4*>(duimg + (int64_t)b * VF_NPL * D * K);
    const int wv = k >> 4, ii = k & 15, s = 2 * blk + (dq >> 1);
#pragma unroll
    for (int pl = 0; pl < VF_NPL; ++pl) {
      unsigned w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        w[i] = (unsigned)h[pl][2 * i] | ((unsigned)h[pl][2 * i + 1] << 16);
      const int g0 = 2 * (dq & 1);
      img[((wv * 16 + s) * VF_NPL + pl) * 64 + 16 * g0 + ii] = make_uint4(w[0], w[1], w[2], w[3]);
      img[((wv * 16 + s) * VF_NPL + pl) * 64 + 16 * (g0 + 1) + ii] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
  // the other orientation through LDS: tile[channel of the block][cluster]
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[(dq * 16 + i) * DU_TLD + k] = vals[i];
  __syncthreads();
  const int wv = blk >> 1;                                  // the block's channels: wave wv, tiles 4 (blk & 1) ..
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int u = threadIdx.x + 256 * rep;                  // 512 units: (tile nt', k-step s, lane)
    const int ntl = u >> 7, s = (u >> 6) & 1, lane = u & 63, i = lane & 15, gq = lane >> 4;
    const float* row = &tile[(16 * ntl + i) * DU_TLD + 32 * s + 8 * gq];
    const f32x4 v0{row[0], row[1], row[2], row[3]}, v1{row[4], row[5], row[6], row[7]};
    u32x4 hi, lo;
    split2x8(v0, v1, hi, lo);
    u32x4* img = reinterpret_cast<u32x4*>(dximg) + (int64_t)b * (4 * 8 * 2 * 2 * 64) +
                 (((wv * 8 + 4 * (blk & 1) + ntl) * 2 + s) * 2) * 64 + lane;
    img[0] = hi;
    img[64] = lo;
    if (b == 0) {                                           // W in the same orientation
      const float* wr = assign_w + (int64_t)(64 * blk + 16 * ntl + i) * K + 32 * s + 8 * gq;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
      split2x8(w0, w1, hi, lo);
      u32x4* wimg = reinterpret_cast<u32x4*>(wdximg) +
                    (((wv * 8 + 4 * (blk & 1) + ntl) * 2 + s) * 2) * 64 + lane;
      wimg[0] = hi;
      wimg[64] = lo;
    }
  }
}
#endif   // SCL_DIAG

// ---------------------------------------------------------------------------------------
// dx16_kernel: grad_x on 16-location tiles (v_mfma_f32_16x16x4_f32), 2-3 workgroups per CU.
//   dxhat[n, d] = sum_{k<64} a[n,k] dU[b][d,k] + ds[n,k] W[d,k]        (K = 128)
//   grad_x[n,d] = rn[n] * (dxhat[n,d] - x[n,d] * rn[n] * rowdot[n])
// A operand (a | ds rows of the tile) stays in 8 registers per lane; the B operand
// [dU[b] | W] is streamed in 32-channel chunks, double-buffered through the same
// conflict-free plane image as rowtile16 (shared by the 4 waves); each finished
// [16 x 32] accumulator block is transposed through a per-wave LDS scratch so every lane
// loads / stores 8 consecutive channels of one location (16-byte bf16 accesses).
constexpr int DX_CH = 32;
constexpr int DX_PLANE = DX_CH * RT_LD;       // 1152 floats (= 18 x 64)
constexpr int DX_CHUNK = 4 * DX_PLANE;        // 4608 floats per buffer
constexpr int DX_SCR = 16 * RT_LD;            // per-wave transpose scratch
constexpr size_t kDx16Lds = (2 * (size_t)DX_CHUNK + 4 * DX_SCR) * sizeof(float);   // 46,080 B

template <typename T>
struct Elem8;
template <>
struct Elem8<float> {
  struct raw {
    f32x4 a, b;
  };
  static __device__ __forceinline__ raw ldraw(const float* p) {
    return raw{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)};
  }
  static __device__ __forceinline__ void cvt(const raw& r, float* v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = r.a[c];
      v[4 + c] = r.b[c];
    }
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <>
struct Elem8<unsigned short> {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 raw;
  static __device__ __forceinline__ raw ldraw(const unsigned short* p) {
    return *reinterpret_cast<const u32x4*>(p);
  }
  static __device__ __forceinline__ void cvt(const raw& w, float* v) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[2 * c] = __uint_as_float(w[c] << 16);
      v[2 * c + 1] = __uint_as_float(w[c] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void st(unsigned short* p, const float* v) {
    u32x4 w;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      w[c] = (unsigned)f32_to_bf16(v[2 * c]) | ((unsigned)f32_to_bf16(v[2 * c + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = w;
  }
};

// grid (ceil(ceil(N/16) / 4), B); block 256.
template <typename T>
__global__ __launch_bounds__(256, 2) void dx16_kernel(const void* __restrict__ xv,
                                                      const float* __restrict__ a,
                                                      const float* __restrict__ ds,
                                                      const float* __restrict__ rn,
                                                      const float* __restrict__ rowdot,
                                                      const float* __restrict__ du,
                                                      const float* __restrict__ w, int N,
                                                      int pre_l2, void* __restrict__ gxv) {
  extern __shared__ __attribute__((aligned(16))) float dx_lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x * 4 + wid) * 16;
  const bool active = n0 < N;
  float* scr = dx_lds + 2 * DX_CHUNK + wid * DX_SCR;
  const float* dub = du + (int64_t)b * D * K;

  // A operand: lane (i, g) holds a[n0+i][16t+4g..] (t < 4) and ds[n0+i][16(t-4)+4g..]
  f32x4 af[8];
  {
    const bool ok = active && n0 + i < N;
    const int64_t gr = (int64_t)b * N + (ok ? n0 + i : 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      af[t] = *reinterpret_cast<const f32x4*>(a + gr * K + 16 * t + 4 * g);
      af[4 + t] = *reinterpret_cast<const f32x4*>(ds + gr * K + 16 * t + 4 * g);
      if (!ok) {
        af[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        af[4 + t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  // epilogue role of this lane: location row_e, channels 8 seg .. 8 seg + 7 of each chunk
  const int row_e = lane >> 2, seg = lane & 3;
  const bool ok_e = active && n0 + row_e < N;
  const int64_t gr_e = (int64_t)b * N + (ok_e ? n0 + row_e : 0);
  const float rn_e = pre_l2 ? rn[gr_e] : 1.0f;
  const float rd_e = rowdot[gr_e];
  // x * rsqrt(max(ss, eps)): with the clamp active the op is a plain scale (no projection)
  const bool proj = pre_l2 && rn_e < 1.0e6f;
  const T* x = reinterpret_cast<const T*>(xv) + gr_e * D + seg * 8;
  T* gx = reinterpret_cast<T*>(gxv) + gr_e * D + seg * 8;

  f32x4 st[4];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int dl = idx >> 5, c4 = 4 * (idx & 7) + ((idx >> 3) & 3);
      const float* s = c4 < 16 ? dub + (int64_t)(chunk * DX_CH + dl) * K + 4 * c4
                               : w + (int64_t)(chunk * DX_CH + dl) * K + 4 * (c4 - 16);
      st[v] = *reinterpret_cast<const f32x4*>(s);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int dl = idx >> 5, c4 = 4 * (idx & 7) + ((idx >> 3) & 3);
      *reinterpret_cast<f32x4*>(&dx_lds[buf * DX_CHUNK + (c4 & 3) * DX_PLANE + dl * RT_LD +
                                        4 * (c4 >> 2)]) = st[v];
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < D / DX_CH; ++c) {
    const bool more = c + 1 < D / DX_CH;
    if (more) stage_load(c + 1);
    typename Elem8<T>::raw xraw{};     // raw prefetch: converted in the epilogue below
    if (active && proj) xraw = Elem8<T>::ldraw(x + c * DX_CH);
    if (active) {
      const float* wb = &dx_lds[(c & 1) * DX_CHUNK + g * DX_PLANE + i * RT_LD];
      f32x4 acc[2];
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B fragments double-buffered in registers (see rowtile16_kernel)
      f32x4 bq[2][2];
      bq[0][0] = *reinterpret_cast<const f32x4*>(wb);
      bq[0][1] = *reinterpret_cast<const f32x4*>(wb + 16 * RT_LD);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t + 1 < 8) {
          bq[(t + 1) & 1][0] = *reinterpret_cast<const f32x4*>(wb + 4 * (t + 1));
          bq[(t + 1) & 1][1] = *reinterpret_cast<const f32x4*>(wb + 16 * RT_LD + 4 * (t + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          acc[0] = mfma16(af[t][cc], bq[t & 1][0][cc], acc[0]);
          acc[1] = mfma16(af[t][cc], bq[t & 1][1][cc], acc[1]);
        }
      }
      // transpose the [16 x 32] block: accumulator (row 4g+j, channel 16dt+i) -> rows
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) scr[(4 * g + j) * RT_LD + 16 * dt + i] = acc[dt][j];
      __builtin_amdgcn_wave_barrier();
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(&scr[row_e * RT_LD + 8 * seg]);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(&scr[row_e * RT_LD + 8 * seg + 4]);
      __builtin_amdgcn_wave_barrier();
      if (ok_e) {
        float out[8];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          out[cc] = v0[cc];
          out[4 + cc] = v1[cc];
        }
        if (proj) {
          const float f = rn_e * rd_e;
          float xin[8];
          Elem8<T>::cvt(xraw, xin);
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) out[cc] -= xin[cc] * f;
        }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) out[cc] *= rn_e;
        Elem8<T>::st(gx + c * DX_CH, out);
      }
    }
    if (more) stage_store((c + 1) & 1);
    __syncthreads();
  }
}

// vlad_dx_kernel: grad_x of one (image, location slice) for a bf16 feature map,
//   dxhat[n, d] = sum_{k<64} a[n,k] dU[b][d,k] + ds[n,k] W[d,k],   grad_x = rn (dxhat - xhat <dxhat, xhat>)
// with the operand [dU[b] | W] (128 x 512, two bf16 planes: A.B = Ah.Bh + Ah.Bl + Al.Bh, three
// orders below the bf16 output rounding) RESIDENT IN REGISTERS: wave w owns channels 128 w .. + 127
// = 8 tiles x 4 k-steps x 2 planes x 4 registers = 256 per lane.  dx16b_kernel re-streamed that
// 256 KB operand through LDS for every 64 locations (115 MB of L2 -> LDS traffic per launch at
// 24 x 1200, with 2-way bank conflicts); here it is read once per workgroup.
// Per 16-location tile: the four waves convert [a | ds] to bf16 high / low fragments
// cooperatively (wave w = k-step w) into an LDS fragment buffer (lane-linear, conflict-free),
// one barrier, 96 MFMAs per wave with the operands swapped (lane = location, registers = four
// consecutive channels), then the wave's [16 loc][128 ch] float32 block goes through a per-wave
// LDS scratch so that x is loaded and grad_x stored as whole 256-byte row segments.
// grid (B images, S slices), block 256, one workgroup per CU.
constexpr int DXV_ABUF = 4 * 2 * 64 * 16;                 // bytes per fragment buffer (8 KB)
constexpr int DXV_SLD = 128 * 4 + 16;                     // bytes per scratch row (128 f32 + pad)
constexpr int DXV_SCR = 16 * DXV_SLD;                     // per wave (8,448 B)
constexpr size_t kVladDxLds = 2 * (size_t)DXV_ABUF + 4 * (size_t)DXV_SCR;   // 50,176 B
constexpr int DXV_TAIL_NB = 36;                           // slab loads in flight per thread in the tail

struct VladDxArgs {
  const unsigned short* x;      // [B][N][512] bf16
  const float* a;               // [B][N][64]
  const float* ds;              // [B][N][64]
  const float* rn;              // [B][N]
  const float* rowdot;          // [B][N]
  const unsigned short* dximg;  // [B] register images of dU (bwd_du_kernel)
  const unsigned short* wdximg; // the same image of W
  int N, pre_l2, steps_per_slice;
  unsigned short* gx;           // [B][N][512] bf16
  int dbg;                      // scl_debug_set_variant(917): clock stamps (scripts/vlad_stamps.py)
  unsigned long long* stamps;
  unsigned short* trash;        // 64 x 16 bytes for the stores of rows past the end
  // tail (wslab != NULL): this workgroup's share of the two parameter gradients
  const float* wslab;           // [nslab][8192 units][4]: vlad_bwd_kernel's dW slabs (complete: it
  int nslab;                    //   ran in front of this kernel)
  const float* du;              // [B][512][64]
  const float* save_vlad;       // [B][VROWS][64]: row 512 = asum
  int B;
  float* grad_w;                // [512][64]
  float* grad_c;                // [512][64]
};

__global__ __launch_bounds__(256, 1) void vlad_dx_kernel(VladDxArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dxv_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sl = blockIdx.y;
  const int nsteps_img = (p.N + VF_STEP - 1) / VF_STEP;
  const int st_lo = sl * p.steps_per_slice;
  const int st_hi = st_lo + p.steps_per_slice < nsteps_img ? st_lo + p.steps_per_slice : nsteps_img;
  const int n_lo = VF_STEP * st_lo;
  const int n_hi = VF_STEP * st_hi < p.N ? VF_STEP * st_hi : p.N;
  const int ntile = (n_hi - n_lo + 15) / 16;
  const unsigned lds0 = nv_lds_byte_of(dxv_lds);
  const unsigned scr0 = lds0 + 2 * DXV_ABUF + wid * DXV_SCR;
  unsigned long long* stp =
      SCL_DIAG_ONLY(p.dbg) && threadIdx.x == 0 ? p.stamps + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 32 : nullptr;
#define DX_STAMP(k)                                         \
  do {                                                      \
    if (SCL_DIAG_ONLY(p.dbg)) {                                            \
      __builtin_amdgcn_sched_barrier(0);                    \
      if (stp) stp[k] = __builtin_amdgcn_s_memtime();       \
      __builtin_amdgcn_sched_barrier(0);                    \
    }                                                       \
  } while (0)
  DX_STAMP(0);

  // ---- the wave's operand: bw[nt][s][pl] = [dU | W]_plane[ch 128 w + 16 nt + i][k 32 s + 8 g ..],
  // coalesced 16-byte loads from the register images bwd_du_kernel wrote
  u32x4 bw[8][4][2];
  {
    const u32x4* dimg = reinterpret_cast<const u32x4*>(p.dximg) + (int64_t)b * (4 * 8 * 2 * 2 * 64) + lane;
    const u32x4* wimg = reinterpret_cast<const u32x4*>(p.wdximg) + lane;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          bw[nt][s][pl] = (s < 2 ? dimg : wimg)[(((wid * 8 + nt) * 2 + (s & 1)) * 2 + pl) * 64];
  }

  // Addresses: wave-uniform 64-bit bases of the slice + 32-bit per-lane byte offsets (the tile
  // loop is paced by its vector instructions: no 64-bit address arithmetic inside it)
  const int nrows = n_hi - n_lo;
  const int64_t row0 = (int64_t)b * p.N + n_lo;
  // staging role: k-step wid of the tile: [a | ds][n0 + i][32 (wid & 1) + 8 g .. + 7]
  const char* abase = reinterpret_cast<const char*>((wid < 2 ? p.a : p.ds) + row0 * K);
  const unsigned acol = (32 * (wid & 1) + 8 * g) * 4;
  auto a_load = [&](int tile, f32x4& v0, f32x4& v1) {
    int r = 16 * tile + i;
    r = r < nrows ? r : nrows - 1;                         // rows past the end: results discarded
    const char* q = abase + ((unsigned)r * (K * 4) + acol);
    v0 = *reinterpret_cast<const f32x4*>(q);
    v1 = *reinterpret_cast<const f32x4*>(q + 16);
  };
  auto a_stage = [&](int buf, const f32x4& v0, const f32x4& v1) {
    u32x4 hi, lo;
    split2x8(v0, v1, hi, lo);
    const unsigned base = lds0 + buf * DXV_ABUF + (wid * 2) * 1024 + lane * 16;
    *(lds_u32x4*)(size_t)(base) = hi;
    *(lds_u32x4*)(size_t)(base + 1024) = lo;
  };
  // epilogue role: piece p = lane + 64 v of the tile's 256 (location, 8-channel chunk) pieces of
  // this wave's 128 channels: location (lane >> 4) + 4 v, chunk lane & 15
  const int el = lane >> 4, ec = lane & 15;
  const char* xbase = reinterpret_cast<const char*>(p.x + row0 * D);
  char* gbase = reinterpret_cast<char*>(p.gx + row0 * D);
  const char* rnbase = reinterpret_cast<const char*>(p.rn + row0);
  const char* rdbase = reinterpret_cast<const char*>(p.rowdot + row0);
  const unsigned xcol = (128 * wid + 8 * ec) * 2;

  // Software pipeline over the 16-location tiles: the matrix work of tile tt and the epilogue of
  // tile tt - 1 (scratch transpose, projection, stores — vector and LDS instructions only) are
  // independent and sit in one basic block, so they share the SIMD: an MFMA holds the vector
  // issue port for 8 of its 16 cycles, the epilogue fits into the other 8.
  struct EpiIn {
    u32x4 xr[4];
    float rn4[4], rd4[4];
    int r0;                                                 // first row of the tile in the slice
  };
  auto epi_load = [&](int tt, EpiIn& e) {
    e.r0 = 16 * tt;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      int r = e.r0 + el + 4 * v;
      r = r < nrows ? r : nrows - 1;
      e.xr[v] = *reinterpret_cast<const u32x4*>(xbase + ((unsigned)r * (D * 2) + xcol));
      e.rn4[v] = *reinterpret_cast<const float*>(rnbase + (unsigned)r * 4);   // (no branch: one
      e.rd4[v] = *reinterpret_cast<const float*>(rdbase + (unsigned)r * 4);   //  basic block per tile)
    }
  };
  auto tile_mfma = [&](int tt, f32x4 (&acc)[8]) {
    u32x4 af[4][2];
    const unsigned base = lds0 + (tt & 1) * DXV_ABUF + lane * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) af[s][pl] = vf_ldsr128(base + (s * 2 + pl) * 1024);
    // The MFMAs are written as asm with the resident operand CONSTRAINED to the accumulator
    // file ("a"): left to itself hipcc keeps what does not fit into 256 vector registers in
    // AGPRs as spill slots and copies every fragment back (four v_accvgpr_mov per MFMA: the
    // vector port, not the matrix pipe, then paces the tile).  hipcc pads no hazards around asm:
    // the first MFMA of a tile takes C = 0 as an immediate (no vector write in front of it), and
    // the accumulators are first read one tile later, or behind vlad_mfma_settle().
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[nt]) : "a"(bw[nt][0][1]), "v"(af[0][0]));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nt]) : "a"(bw[nt][0][0]), "v"(af[0][1]));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nt]) : "a"(bw[nt][0][0]), "v"(af[0][0]));
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nt]) : "a"(bw[nt][s][1]), "v"(af[s][0]));   // Bl . Ah
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nt]) : "a"(bw[nt][s][0]), "v"(af[s][1]));   // Bh . Al
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[nt]) : "a"(bw[nt][s][0]), "v"(af[s][0]));   // Bh . Ah
      }
    }
  };
  // wait states between the last asm MFMA and the first instruction that reads its result
  auto vlad_mfma_settle = [] { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); };
  auto epilogue = [&](const f32x4 (&acc)[8], const EpiIn& e) {
    // transpose through the wave's scratch: lane (i = location, g) holds channels
    // 16 nt + 4 g .. + 3 -> rows of 128 channels (LDS serves a wave's accesses in order)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      *(__attribute__((address_space(3))) f32x4*)(size_t)(scr0 + i * DXV_SLD + (16 * nt + 4 * g) * 4) = acc[nt];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const unsigned ra = scr0 + (el + 4 * v) * DXV_SLD + ec * 32;
      const f32x4 d0 = *(const __attribute__((address_space(3))) f32x4*)(size_t)(ra);
      const f32x4 d1 = *(const __attribute__((address_space(3))) f32x4*)(size_t)(ra + 16);
      float out[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
      const float rnv = p.pre_l2 ? e.rn4[v] : 1.0f;
      // x * rsqrt(max(ss, eps)): with the clamp active the op is a plain scale (no projection)
      const float f = (p.pre_l2 && rnv < 1.0e6f) ? rnv * e.rd4[v] : 0.f;
      float xv8[8];
      Elem8<unsigned short>::cvt(e.xr[v], xv8);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) out[cc] = (out[cc] - xv8[cc] * f) * rnv;
      // rows past the end go to the trash line: no exec-masked branch inside the tile's block
      const int r = e.r0 + el + 4 * v;
      Elem8<unsigned short>::st(r < nrows ? reinterpret_cast<unsigned short*>(gbase + ((unsigned)r * (D * 2) + xcol))
                                          : p.trash + 8 * lane, out);
    }
  };

  if (SCL_DIAG_ONLY(p.dbg)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DX_STAMP(1);
  }
  f32x4 pa0, pa1;
  a_load(0, pa0, pa1);
  a_stage(0, pa0, pa1);
  if (ntile > 1) a_load(1, pa0, pa1);
  __syncthreads();
  DX_STAMP(2);

  // One tile step: the matrix work of tile tt into (accc, ec2) and, HAND-INTERLEAVED with it, the
  // epilogue of tile tt - 1 from (accp, ep), the requests for tile tt + 1's epilogue inputs and
  // the staging of tile tt + 1's fragments.  A wave issues in order, so vector work only hides
  // under an MFMA if it stands right behind it in the instruction stream (an MFMA holds the vector
  // port for 8 of its 16 cycles); hipcc interleaves builtin MFMAs with vector code by itself but
  // not asm ones, and asm is what keeps the resident operand out of the spill path.  So the step
  // is written as 32 groups of three MFMAs (one (channel tile, k-step)), each followed by one
  // slice — about a 32nd — of the vector work; nothing in it branches.
  auto tile_step = [&](int tt, f32x4 (&accc)[8], EpiIn& ec2, const f32x4 (&accp)[8], const EpiIn& ep) {
    u32x4 af[4][2];
    {
      const unsigned base = lds0 + (tt & 1) * DXV_ABUF + lane * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) af[s][pl] = vf_ldsr128(base + (s * 2 + pl) * 1024);
    }
    f32x4 d0[4], d1[4];
    float outv[4][8];
    float rnv[4], fv[4];
    u32x4 sth, stl;
    ec2.r0 = 16 * tt;
    // (macros, not a generic lambda: clang does not capture asm operands inside one)
#define DX_M0(nt)                                                                                                 \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(accc[nt]) : "a"(bw[nt][0][1]), "v"(af[0][0]));    \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accc[nt]) : "a"(bw[nt][0][0]), "v"(af[0][1]));    \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accc[nt]) : "a"(bw[nt][0][0]), "v"(af[0][0]));
#define DX_MS(nt, sq)                                                                                              \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accc[nt]) : "a"(bw[nt][sq][1]), "v"(af[sq][0]));  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accc[nt]) : "a"(bw[nt][sq][0]), "v"(af[sq][1]));  \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accc[nt]) : "a"(bw[nt][sq][0]), "v"(af[sq][0]));
    auto slice = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      if constexpr (G < 4) {
        // previous tile's accumulators into the wave's scratch: lane (i = location, g) holds
        // channels 16 nt + 4 g .. + 3 (LDS serves a wave's accesses in order)
#pragma unroll
        for (int nt = 2 * G; nt < 2 * G + 2; ++nt)
          *(__attribute__((address_space(3))) f32x4*)(size_t)(scr0 + i * DXV_SLD + (16 * nt + 4 * g) * 4) = accp[nt];
      } else if constexpr (G < 8) {
        // this tile's epilogue inputs (used one step later)
        constexpr int v = G - 4;
        int r = ec2.r0 + el + 4 * v;
        r = r < nrows ? r : nrows - 1;
        ec2.xr[v] = *reinterpret_cast<const u32x4*>(xbase + ((unsigned)r * (D * 2) + xcol));
        ec2.rn4[v] = *reinterpret_cast<const float*>(rnbase + (unsigned)r * 4);
        ec2.rd4[v] = *reinterpret_cast<const float*>(rdbase + (unsigned)r * 4);
      } else if constexpr (G < 12) {
        constexpr int v = G - 8;                           // rows of 128 channels back from the scratch
        const unsigned ra = scr0 + (el + 4 * v) * DXV_SLD + ec * 32;
        d0[v] = *(const __attribute__((address_space(3))) f32x4*)(size_t)(ra);
        d1[v] = *(const __attribute__((address_space(3))) f32x4*)(size_t)(ra + 16);
        rnv[v] = p.pre_l2 ? ep.rn4[v] : 1.0f;
        // x * rsqrt(max(ss, eps)): with the clamp active the op is a plain scale (no projection)
        fv[v] = (p.pre_l2 && rnv[v] < 1.0e6f) ? rnv[v] * ep.rd4[v] : 0.f;
      } else if constexpr (G < 28) {
        constexpr int v = (G - 12) >> 2, qd = (G - 12) & 3;  // two of the row's eight values
        const unsigned xw = ep.xr[v][qd];
        const float x0 = __uint_as_float(xw << 16), x1 = __uint_as_float(xw & 0xffff0000u);
        const float a0 = qd < 2 ? d0[v][2 * qd] : d1[v][2 * qd - 4];
        const float a1 = qd < 2 ? d0[v][2 * qd + 1] : d1[v][2 * qd - 3];
        outv[v][2 * qd] = (a0 - x0 * fv[v]) * rnv[v];
        outv[v][2 * qd + 1] = (a1 - x1 * fv[v]) * rnv[v];
        if constexpr (qd == 3) {
          // rows past the end go to the trash line: no exec-masked branch inside the step
          const int r = ep.r0 + el + 4 * v;
          Elem8<unsigned short>::st(r < nrows ? reinterpret_cast<unsigned short*>(gbase + ((unsigned)r * (D * 2) + xcol))
                                              : p.trash + 8 * lane, outv[v]);
        }
      } else if constexpr (G == 28) {
        split2x8(pa0, pa1, sth, stl);                       // next tile's fragments (k-step wid)
      } else if constexpr (G == 29) {
        const unsigned base = lds0 + ((tt + 1) & 1) * DXV_ABUF + (wid * 2) * 1024 + lane * 16;
        *(lds_u32x4*)(size_t)(base) = sth;
        *(lds_u32x4*)(size_t)(base + 1024) = stl;
      } else if constexpr (G == 30) {
        a_load(tt + 2, pa0, pa1);                           // (rows clamped: harmless past the end)
      }
    };
#define DX_SL(G) slice(std::integral_constant<int, G>{});
#define DX_NT(nt)                                                                       \
  DX_M0(nt) DX_SL(4 * nt) DX_MS(nt, 1) DX_SL(4 * nt + 1) DX_MS(nt, 2) DX_SL(4 * nt + 2) \
  DX_MS(nt, 3) DX_SL(4 * nt + 3)
    DX_NT(0) DX_NT(1) DX_NT(2) DX_NT(3) DX_NT(4) DX_NT(5) DX_NT(6) DX_NT(7)
#undef DX_NT
#undef DX_SL
#undef DX_MS
#undef DX_M0
    vlad_mfma_settle();
    __syncthreads();
    if (tt < 9) DX_STAMP(3 + tt);
  };
  f32x4 acca[8], accb[8];
  EpiIn ea, eb;
  epi_load(0, ea);
  tile_mfma(0, acca);
  vlad_mfma_settle();
  if (ntile > 1) {
    a_stage(1, pa0, pa1);
    if (ntile > 2) a_load(2, pa0, pa1);
  }
  __syncthreads();
  int tt = 1;
#pragma unroll 1
  for (; tt + 1 < ntile; tt += 2) {                         // ping-pong: no register copies
    tile_step(tt, accb, eb, acca, ea);
    tile_step(tt + 1, acca, ea, accb, eb);
  }
  if (tt < ntile) {
    tile_step(tt, accb, eb, acca, ea);
    epilogue(accb, eb);
  } else {
    epilogue(acca, ea);
  }
  if (SCL_DIAG_ONLY(p.dbg)) {
    DX_STAMP(13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DX_STAMP(14);
  }
#undef DX_STAMP

  // ---- tail: the two parameter gradients, spread over the launch.  vlad_bwd_kernel (in front of
  // this kernel) left one dW slab per (slice, image); grad_w is their sum, grad_c[d,k] = sum_b
  // dU[b,d,k] asum[b,k].  They were two more launches (partial sums over 8 slab groups, then a
  // 32-workgroup finish): 10 us at the floor of a dependent kernel for 31 MB of reads.  Here every
  // workgroup sums a contiguous run of 16-byte units over ALL slabs (threads = units x slab
  // groups, groups combined through LDS in a fixed order: bitwise reproducible for a given grid).
  if (p.wslab) {
    __syncthreads();                                         // the scratch is free
    const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x;
    {
      const int upw = (8192 + nwg - 1) / nwg;                // units per workgroup
      const int u_lo = wg * upw, u_hi = u_lo + upw < 8192 ? u_lo + upw : 8192;
      const int chunk = upw < 256 ? upw : 256;               // units per pass
      int G = 256 / chunk;                                   // slab groups
      G = G > 8 ? 8 : G;
      f32x4* part = reinterpret_cast<f32x4*>(dxv_lds);       // [G][chunk] (<= 256 x 16 B)
      const int ul = tid % chunk, grp = tid / chunk;
      for (int u0 = u_lo; u0 < u_hi; u0 += chunk) {
        const int uidx = u0 + ul;
        const bool act = grp < G && uidx < u_hi;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (act) {
          const f32x4* src = reinterpret_cast<const f32x4*>(p.wslab) + uidx;
          for (int s0 = grp; s0 < p.nslab; s0 += DXV_TAIL_NB * G) {   // (one round at 24 x 1200: the
            f32x4 v[DXV_TAIL_NB];                                     //  tail is a round trip long)
#pragma unroll
            for (int j = 0; j < DXV_TAIL_NB; ++j) {          // branch-free: past the end re-reads
              const int sidx = s0 + j * G < p.nslab ? s0 + j * G : p.nslab - 1;
              v[j] = src[(int64_t)sidx * 8192];
            }
            __builtin_amdgcn_sched_barrier(0);               // every load requested before the first add
#pragma unroll
            for (int j = 0; j < DXV_TAIL_NB; ++j) acc += v[j] * (s0 + j * G < p.nslab ? 1.0f : 0.0f);
          }
          if (grp > 0) part[grp * chunk + ul] = acc;
        }
        __syncthreads();
        if (act && grp == 0) {
          for (int q2 = 1; q2 < G; ++q2) acc += part[q2 * chunk + ul];
          // unit ((w * 32 + ct) * 64 + 16 g + i) = dW[16 ct + 4 g + 0..3][16 w + i]
          const int ww = uidx >> 11, ct = (uidx >> 6) & 31, l2 = uidx & 63;
          const int kk = 16 * ww + (l2 & 15), d0 = 16 * ct + 4 * (l2 >> 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) p.grad_w[(d0 + j) * K + kk] = acc[j];
        }
        __syncthreads();
      }
    }
    {
      const int epw = (D * K + nwg - 1) / nwg;               // elements of grad_c per workgroup
      const int e_lo = wg * epw, e_hi = e_lo + epw < D * K ? e_lo + epw : D * K;
      for (int e = e_lo + tid; e < e_hi; e += 256) {
        const int kk = e & (K - 1);
        float gc = 0.f;
        for (int b0 = 0; b0 < p.B; b0 += 24) {               // (24 images: one round trip)
          float dv[24], as8[24];
#pragma unroll
          for (int bb = 0; bb < 24; ++bb) {
            const int b2 = b0 + bb < p.B ? b0 + bb : p.B - 1;
            dv[bb] = p.du[(int64_t)b2 * D * K + e];
            as8[bb] = p.save_vlad[((int64_t)b2 * VROWS + D) * K + kk];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int bb = 0; bb < 24; ++bb) gc = fmaf(dv[bb], b0 + bb < p.B ? as8[bb] : 0.f, gc);
        }
        p.grad_c[e] = gc;
      }
    }
  }
}

// grad_w[d,k] = sum_b sum_s slab[b][s][d,k];  grad_c[d,k] = sum_b dU[b,d,k] * asum[b,k].
// grid D*K/64, block 256: thread (j, q) sums the images b = q, q + 4, ... of element
// 64 * blockIdx.x + j; the four partials are combined in a fixed order (512 workgroups
// instead of 128 threads-per-element chains of 4 B loads each).
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ wpart,
                                                           const float* __restrict__ du,
                                                           const float* __restrict__ save_vlad,
                                                           int B, float* __restrict__ grad_w,
                                                           float* __restrict__ grad_c) {
  __shared__ float red[2][4][64];
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + j;
  const int k = idx % K;
  float gw = 0.f, gc = 0.f;
  for (int b = q; b < B; b += 4) {
#pragma unroll
    for (int s = 0; s < NSPLIT; ++s) gw += wpart[((int64_t)b * NSPLIT + s) * D * K + idx];
    gc = fmaf(du[(int64_t)b * D * K + idx], save_vlad[((int64_t)b * VROWS + D) * K + k], gc);
  }
  red[0][q][j] = gw;
  red[1][q][j] = gc;
  __syncthreads();
  if (q == 0) {
    grad_w[idx] = (red[0][0][j] + red[0][1][j]) + (red[0][2][j] + red[0][3][j]);
    grad_c[idx] = (red[1][0][j] + red[1][1][j]) + (red[1][2][j] + red[1][3][j]);
  }
}

// ---------------------------------------------------------------------- host side
template <typename T, int VAR>
void launch_variant_one(const RowTileArgs& a, dim3 grid, hipStream_t st) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowtile16_kernel<T, ASSIGN, VAR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowTile16Lds);
  SCL_LAUNCH("rowtile16_kernel<ASSIGN>", (rowtile16_kernel<T, ASSIGN, VAR>), grid, dim3(256), kRowTile16Lds,
             st, a);
}
template <typename T>
void launch_rowtile_variant(const RowTileArgs& a, dim3 grid, hipStream_t st) {
  switch (scl_variant() & 7) {
    case 1: launch_variant_one<T, 1>(a, grid, st); break;
    case 2: launch_variant_one<T, 2>(a, grid, st); break;
    case 3: launch_variant_one<T, 3>(a, grid, st); break;
    case 4: launch_variant_one<T, 4>(a, grid, st); break;
    case 5: launch_variant_one<T, 5>(a, grid, st); break;
    case 6: launch_variant_one<T, 6>(a, grid, st); break;
    default: launch_variant_one<T, 7>(a, grid, st); break;
  }
}

template <typename T, int MODE>
void launch_rowtile(const RowTileArgs& a, hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowtile16_kernel<T, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowTile16Lds);
  });
  const int tiles16 = (a.N + 15) / 16;
  const dim3 grid((tiles16 + 3) / 4, a.B);
  if (MODE == ASSIGN && scl_variant() >= 1 && scl_variant() <= 7) {   // ablate_rowtile.py
    launch_rowtile_variant<T>(a, grid, st);
    return;
  }
  SCL_LAUNCH(MODE == ASSIGN ? "rowtile16_kernel<ASSIGN>" : "rowtile16_kernel<DASSIGN>", (rowtile16_kernel<T, MODE>),
             grid, dim3(256), kRowTile16Lds, st, a);
}

// bf16 feature maps take the fused kernels; scl_debug_set_variant(1 .. 8) sends them through the
// float32-MFMA kernels instead (A/B timing and parity runs)
inline bool use_fused() { return scl_variant() < 1 || scl_variant() > 8; }

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base((char*)p) {}
  float* take(size_t floats) {
    float* ptr = (float*)(base + off);
    off += scl_round256(floats * sizeof(float));
    return ptr;
  }
};

// Slices of the fused kernels: steps of 32 locations per workgroup so that B * S workgroups
// cover the chip once (one workgroup per CU, 512 registers per lane).
struct VladPlan {
  int steps_per_slice, S;
};
inline int vlad_cus() {
  const int n = scl_device_cus();      // per device (scl_common.h)
  return n;
}
// images from which inference takes one workgroup per image with the fused finish (SCL_VLAD_FIN_IMAGES;
// measured, profiles/r06/netvlad_inference_sweep.txt)
inline int vlad_fin_images() {
  static const int n = [] {
    const char* e = getenv("SCL_VLAD_FIN_IMAGES");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 144;
  }();
  return n;
}
inline VladPlan vlad_plan(int B, int N) {
  const int nsteps = (N + VF_STEP - 1) / VF_STEP;
  VladPlan p;
  // (Round 5 tried "as many slices as keep B * S within ONE round of the chip" — 96 images: 2 slices
  // of 19 steps instead of 3 of 15 / 15 / 8, two partial VLADs per image instead of three.  Measured
  // equal to 2 % slower: the 32 workgroups of the short third slices start when the first short ones
  // end and finish with the long ones, so the cut by total steps / CUs was already balanced.)
  p.steps_per_slice = (int)(((int64_t)nsteps * B + vlad_cus() - 1) / vlad_cus());
  if (p.steps_per_slice < 1) p.steps_per_slice = 1;
  p.S = (nsteps + p.steps_per_slice - 1) / p.steps_per_slice;
  return p;
}

struct FwdWs {
  float *wt, *part, *colsum, *colsq, *vlad, *assign, *rnorm, *trash;
  unsigned long long* stamps;   // diagnostics: [B * S][32], the LAST bytes of the workspace
  unsigned short* wplanes;      // register images of W when the caller brought none (vlad_planes.h)
  unsigned long long* gran;     // [B][8] granules of the finish kernel's exchange
  size_t total;
};
inline FwdWs carve_fwd(void* ws, int B, int N) {
  Carver c(ws);
  FwdWs w;
  const int S = vlad_plan(B, N).S > NSPLIT ? vlad_plan(B, N).S : NSPLIT;
  w.wt = c.take((size_t)D * K);
  w.part = c.take((size_t)B * S * D * K);
  w.colsum = c.take((size_t)B * S * K);
  w.trash = c.take(256);
  w.colsq = c.take((size_t)B * 32 * K);
  w.vlad = c.take((size_t)B * VROWS * K);
  w.assign = c.take((size_t)B * N * K);
  w.rnorm = c.take((size_t)B * N);
  w.wplanes = (unsigned short*)c.take(VP_BYTES / 4);
  w.gran = (unsigned long long*)c.take((size_t)B * 8 * 2);
  w.stamps = (unsigned long long*)c.take((size_t)B * S * 32 * 2);
  w.total = c.off;
  return w;
}

struct BwdWs {
  float *du, *dut, *cdu, *ds, *rowdot, *wpart, *dots, *wpartial, *trash;
  unsigned short* duimg;     // [B] register images of dU^T for vlad_bwd_kernel
  unsigned short* dximg;     // [B] register images of dU for vlad_dx_kernel (2 planes)
  unsigned short* wdximg;    // the same image of W
  unsigned long long* stamps;   // diagnostics: [B * S][32], the LAST bytes of the workspace
  size_t total;
};
inline BwdWs carve_bwd(void* ws, int B, int N) {
  Carver c(ws);
  BwdWs w;
  w.du = c.take((size_t)B * D * K);
  w.dut = c.take((size_t)B * D * K);
  w.cdu = c.take((size_t)B * K);
  w.ds = c.take((size_t)B * N * K);
  w.rowdot = c.take((size_t)B * N);
  const int S = vlad_plan(B, N).S > NSPLIT ? vlad_plan(B, N).S : NSPLIT;
  w.wpart = c.take((size_t)B * S * D * K);
  w.wpartial = c.take((size_t)VW_GROUPS * D * K);
  w.trash = c.take(256);
  w.dots = c.take((size_t)B * 8 * 4 * K);
  w.duimg = (unsigned short*)c.take((size_t)B * VF_NPL * D * K / 2);
  w.dximg = (unsigned short*)c.take((size_t)B * 2 * D * K / 2);
  w.wdximg = (unsigned short*)c.take(VP_BYTES / 4);       // both images of W when the caller brought none
  w.stamps = (unsigned long long*)c.take((size_t)B * S * 32 * 2);
  w.total = c.off;
  return w;
}

inline bool shape_ok(int B, int N) { return B >= 1 && N >= 1 && B <= 65535 && N <= (1 << 22); }

}  // namespace

extern "C" size_t scl_netvlad_fwd_workspace_bytes(int B, int N) {
  if (!shape_ok(B, N)) return 0;
  return carve_fwd(nullptr, B, N).total;
}

// scl_debug_set_variant(920): round 3's launch structure (separate plane-split, finish_sum,
// finish_norm, bwd_dots, bwd_du, wgrad_partial, wgrad_finish kernels) for same-box A/B timing
inline bool old_launches() { return scl_variant() == 920; }
// scl_debug_set_variant(922): the four-wave fused kernels of round 3 instead of the eight-wave ones
inline bool four_waves() { return scl_variant() == 922 || scl_variant() == 920 || scl_variant() == 916; }

__global__ __launch_bounds__(256) void vlad_planes_kernel(const float* __restrict__ w,
                                                          unsigned short* __restrict__ planes) {
  vlad_planes_wave(w, planes, planes + VP_FWD_ELEMS, 4 * blockIdx.x + (threadIdx.x >> 6), threadIdx.x & 63);
}

extern "C" size_t scl_netvlad_planes_bytes(void) { return VP_BYTES; }

extern "C" int scl_netvlad_planes(const float* assign_w, void* planes, void* stream) {
  if (!assign_w || !planes) return SCL_E_NULL;
  if (!scl_aligned256(planes) || ((uintptr_t)assign_w % 16)) return SCL_E_SHAPE;
  SCL_LAUNCH("vlad_planes_kernel", vlad_planes_kernel, dim3(VP_WAVES / 4), dim3(256), 0, (hipStream_t)stream,
             assign_w, (unsigned short*)planes);
  return scl_launch_status();
}

extern "C" int scl_netvlad_fwd_p(const void* x, int x_dtype, const float* assign_w,
                                 const float* centers, const void* w_planes, int B, int N, int pre_l2,
                                 float* out, float* save_assign, float* save_logit, float* save_rnorm,
                                 float* save_vlad, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (!x || !assign_w || !centers || !out || !workspace) return SCL_E_NULL;
  if (!shape_ok(B, N)) return SCL_E_SHAPE;
  if (x_dtype != SCL_DT_F32 && x_dtype != SCL_DT_BF16) return SCL_E_KIND;
  if (((uintptr_t)x % 16) != 0) return SCL_E_SHAPE;
  if (w_planes && !scl_aligned256(w_planes)) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  FwdWs w = carve_fwd(workspace, B, N);
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* assign = save_assign ? save_assign : w.assign;
  float* rnorm = save_rnorm ? save_rnorm : w.rnorm;

  if (x_dtype == SCL_DT_BF16 && use_fused()) {
    // one pass over x: soft-assignment and aggregation fused; then ONE finish kernel
    static SclDeviceOnce once;
    scl_call_once(once, [] {
#ifdef SCL_DIAG
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVladFusedLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVladFusedLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd8_kernel<true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVlad8Lds);
#endif
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd8_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVlad8Lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd8_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVlad8Lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_fwd8_kernel<false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVlad8Lds);
    });
    VladPlan pl = vlad_plan(B, N);
    // Inference (nothing saved) at many images: ONE workgroup per image with the finish in its tail —
    // no partial VLADs written and re-read, one launch (evaluation/inference.py at large
    // images_per_pass).  From kVladFinImages images on; below that the slices keep the chip full.
    const bool fin = !save_assign && !save_rnorm && !save_vlad && !four_waves() && !old_launches() &&
                     (B >= vlad_fin_images() || scl_variant() == 924) &&    // 924: at any batch size (tests)
                     scl_variant() != 923;                                   // 923: the two launches, for A/B
    if (fin) {
      pl.steps_per_slice = (N + VF_STEP - 1) / VF_STEP;
      pl.S = 1;
    }
    const unsigned short* planes = (const unsigned short*)w_planes;
    if (!planes || old_launches()) {
#ifdef SCL_DIAG
      if (old_launches())
        SCL_LAUNCH("vlad_split_w_kernel", vlad_split_w_kernel, dim3(64, VF_WREP), dim3(64), 0, st,
                   assign_w, w.wplanes);
      else
#endif
        SCL_LAUNCH("vlad_planes_kernel", vlad_planes_kernel, dim3(VP_WAVES / 4), dim3(256), 0, st, assign_w,
                   w.wplanes);
      planes = w.wplanes;
    }
    VladFwdArgs fa{};
    fa.x = (const unsigned short*)x;
    fa.wimg = planes;
    fa.N = N;
    fa.pre_l2 = pre_l2 ? 1 : 0;
    fa.steps_per_slice = pl.steps_per_slice;
    fa.slab = w.part;
    fa.colsum = w.colsum;
    fa.trash = w.trash;
    fa.gran = w.gran;
    fa.dbg = (scl_variant() == 916 || scl_variant() == 918) ? 16 : 0;   // scripts/vlad_stamps.py
    fa.stamps = w.stamps;
    // (round 6: the eight-wave kernels neither write nor read logits — save_logit may be NULL; the
    // four-wave kernels of the diagnostic build still do)
    const bool save = save_assign && save_rnorm && (save_logit || !four_waves());
    if (save) {
      fa.assign = save_assign;
      fa.logit = save_logit;
      fa.rnorm = save_rnorm;
    }
    if (fin) {
      fa.fin_centers = centers;
      fa.fin_out = out;
      SCL_LAUNCH("vlad_fwd8_kernel<finish>", (vlad_fwd8_kernel<false, false, true>), dim3(B, 1), dim3(512),
                 kVlad8Lds, st, fa);
      return scl_launch_status();
    }
#ifdef SCL_DIAG
    if (four_waves()) {
      if (save)
        SCL_LAUNCH("vlad_fwd_kernel<true>", vlad_fwd_kernel<true>, dim3(B, pl.S), dim3(256),
                   kVladFusedLds, st, fa);
      else
        SCL_LAUNCH("vlad_fwd_kernel<false>", vlad_fwd_kernel<false>, dim3(B, pl.S), dim3(256),
                   kVladFusedLds, st, fa);
    } else if (save && scl_variant() == 918) {
      SCL_LAUNCH("vlad_fwd8_kernel<stamps>", (vlad_fwd8_kernel<true, true>), dim3(B, pl.S), dim3(512),
                 kVlad8Lds, st, fa);
    } else
#endif
    {
      if (save)
        SCL_LAUNCH("vlad_fwd8_kernel<true>", vlad_fwd8_kernel<true>, dim3(B, pl.S), dim3(512), kVlad8Lds,
                   st, fa);
      else
        SCL_LAUNCH("vlad_fwd8_kernel<false>", vlad_fwd8_kernel<false>, dim3(B, pl.S), dim3(512), kVlad8Lds,
                   st, fa);
    }
    float* vlad = save_vlad ? save_vlad : w.vlad;
#ifdef SCL_DIAG
    if (old_launches()) {
      SCL_LAUNCH("vlad_finish_sum_kernel", vlad_finish_sum_kernel, dim3(B, 32), dim3(256), 0, st,
                 (const float*)w.part, (const float*)w.colsum, centers, pl.S, vlad, w.colsq);
      SCL_LAUNCH("finish_norm_kernel", finish_norm_kernel, dim3(8, B), dim3(256), 0, st, vlad,
                 (const float*)w.colsq, 32, out);
      return scl_launch_status();
    }
#endif
    VladFinishArgs na{};
    na.slab = w.part;
    na.colsum = w.colsum;
    na.centers = centers;
    na.S = pl.S;
    na.B = B;
    na.vlad = vlad;
    na.out = out;
    na.gran = w.gran;
    na.spin_limit = spin_limit();
    SCL_LAUNCH("vlad_finish_kernel", vlad_finish_kernel, dim3(8, B), dim3(256), 0, st, na);
    return scl_launch_status();
  }
  // float32 feature maps (and bf16 ones under scl_debug_set_variant(1 .. 8)): float32-MFMA kernels
  SCL_LAUNCH("transpose_w_kernel", transpose_w_kernel, dim3(D * K / 256), dim3(256), 0, st, assign_w,
             w.wt);
  RowTileArgs a{};
  a.x = x;
  a.bt = w.wt;
  a.bt_stride = 0;
  a.B = B;
  a.N = N;
  a.pre_l2 = pre_l2 ? 1 : 0;
  a.assign = assign;
  a.logit = save_logit;
  a.rnorm = rnorm;
  if (x_dtype == SCL_DT_F32) {
    launch_rowtile<float, ASSIGN>(a, st);
    SCL_LAUNCH("aggregate_kernel<float>", aggregate_kernel<float>, dim3(D / 64, NSPLIT, B), dim3(256), 0,
               st, x, (const float*)assign, (const float*)rnorm, N, w.part, w.colsum);
  } else {
    launch_rowtile<unsigned short, ASSIGN>(a, st);
    SCL_LAUNCH("aggregate_kernel<bf16>", aggregate_kernel<unsigned short>, dim3(D / 64, NSPLIT, B),
               dim3(256), 0, st, x, (const float*)assign, (const float*)rnorm, N, w.part, w.colsum);
  }
  float* vlad = save_vlad ? save_vlad : w.vlad;
  SCL_LAUNCH("finish_sum_kernel", finish_sum_kernel, dim3(8, B), dim3(256), 0, st,
             (const float*)w.part, (const float*)w.colsum, centers, vlad, w.colsq);
  SCL_LAUNCH("finish_norm_kernel", finish_norm_kernel, dim3(8, B), dim3(256), 0, st, vlad,
             (const float*)w.colsq, 8, out);
  return scl_launch_status();
}

extern "C" int scl_netvlad_fwd(const void* x, int x_dtype, const float* assign_w,
                               const float* centers, int B, int N, int pre_l2, float* out,
                               float* save_assign, float* save_logit, float* save_rnorm,
                               float* save_vlad, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return scl_netvlad_fwd_p(x, x_dtype, assign_w, centers, nullptr, B, N, pre_l2, out, save_assign,
                           save_logit, save_rnorm, save_vlad, workspace, workspace_bytes, stream);
}

extern "C" size_t scl_netvlad_bwd_workspace_bytes(int B, int N) {
  if (!shape_ok(B, N)) return 0;
  return carve_bwd(nullptr, B, N).total;
}

extern "C" int scl_netvlad_bwd_p(const void* x, int x_dtype, const float* assign_w,
                                 const float* centers, const void* w_planes, const float* grad_out,
                                 const float* save_assign, const float* save_logit,
                                 const float* save_rnorm, float* save_vlad, int B, int N,
                                 int pre_l2, void* grad_x, float* grad_w, float* grad_c,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !assign_w || !centers || !grad_out || !save_assign || !save_rnorm ||
      !save_vlad || !grad_x || !grad_w || !grad_c || !workspace)
    return SCL_E_NULL;
  if (!shape_ok(B, N)) return SCL_E_SHAPE;
  if (x_dtype != SCL_DT_F32 && x_dtype != SCL_DT_BF16) return SCL_E_KIND;
  // the saved logits: float32 feature maps and the four-wave kernels of the diagnostic build only
  if (!save_logit && !(x_dtype == SCL_DT_BF16 && use_fused() && !four_waves())) return SCL_E_NULL;
  if (((uintptr_t)x % 16) != 0 || ((uintptr_t)save_vlad % 8) != 0) return SCL_E_SHAPE;
  if (w_planes && !scl_aligned256(w_planes)) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  BwdWs w = carve_bwd(workspace, B, N);
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  const bool fused = x_dtype == SCL_DT_BF16 && use_fused();
  const unsigned short* wdx = w_planes ? (const unsigned short*)w_planes + VP_FWD_ELEMS : nullptr;
#ifdef SCL_DIAG
  if (old_launches()) {
    SCL_LAUNCH("bwd_dots_kernel", bwd_dots_kernel, dim3(8, B), dim3(256), 0, st, (const float*)save_vlad,
               grad_out, centers, w.dots);
    SCL_LAUNCH("bwd_du_kernel", bwd_du_kernel, dim3(8, B), dim3(256), 0, st, (const float*)save_vlad,
               grad_out, (const float*)w.dots, w.du, fused ? (float*)nullptr : w.dut,
               fused ? w.duimg : (unsigned short*)nullptr, w.dximg, assign_w, w.wdximg, w.cdu);
    wdx = w.wdximg;
  } else
#endif
  {
    if (fused && !wdx) {
      SCL_LAUNCH("vlad_planes_kernel", vlad_planes_kernel, dim3(VP_WAVES / 4), dim3(256), 0, st, assign_w,
                 w.wdximg);
      wdx = w.wdximg + VP_FWD_ELEMS;
    }
    VladProArgs pa{};
    pa.save_vlad = save_vlad;
    pa.grad_out = grad_out;
    pa.centers = centers;
    pa.du = w.du;
    pa.dut = fused ? (float*)nullptr : w.dut;
    pa.duimg = fused ? w.duimg : (unsigned short*)nullptr;
    pa.dximg = w.dximg;
    pa.cdu = w.cdu;
    pa.spin_limit = spin_limit();
    SCL_LAUNCH("vlad_bwd_prologue_kernel", vlad_bwd_prologue_kernel, dim3(8, B), dim3(256), 0, st, pa);
  }
  if (fused) {
    static SclDeviceOnce once;
    scl_call_once(once, [] {
#ifdef SCL_DIAG
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_bwd_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVladFusedLds);
#endif
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_dx_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVladDxLds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vlad_bwd8_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVlad8Lds);
    });
    const VladPlan pl = vlad_plan(B, N);
    VladBwdArgs ba{};
    ba.x = (const unsigned short*)x;
    ba.duimg = w.duimg;
    ba.a = save_assign;
    ba.lg = save_logit;
    ba.rn = save_rnorm;
    ba.cdu = w.cdu;
    ba.N = N;
    ba.steps_per_slice = pl.steps_per_slice;
    ba.ds = w.ds;
    ba.rowdot = w.rowdot;
    ba.slab = w.wpart;
    ba.trash = w.trash;
#ifdef SCL_DIAG
    if (four_waves())
      SCL_LAUNCH("vlad_bwd_kernel", vlad_bwd_kernel, dim3(B, pl.S), dim3(256), kVladFusedLds, st, ba);
    else
#endif
      SCL_LAUNCH("vlad_bwd8_kernel", vlad_bwd8_kernel, dim3(B, pl.S), dim3(512), kVlad8Lds, st, ba);
    VladDxArgs da{};
    da.x = (const unsigned short*)x;
    da.a = save_assign;
    da.ds = w.ds;
    da.rn = save_rnorm;
    da.rowdot = w.rowdot;
    da.dximg = w.dximg;
    da.wdximg = wdx;
    da.N = N;
    da.pre_l2 = pre_l2 ? 1 : 0;
    da.steps_per_slice = pl.steps_per_slice;
    da.gx = (unsigned short*)grad_x;
    da.trash = (unsigned short*)w.trash;
    da.dbg = scl_variant() == 917 ? 1 : 0;
    da.stamps = w.stamps;
    if (!old_launches()) {               // the parameter gradients ride in the tail of this launch
      da.wslab = w.wpart;
      da.nslab = pl.S * B;
      da.du = w.du;
      da.save_vlad = save_vlad;
      da.B = B;
      da.grad_w = grad_w;
      da.grad_c = grad_c;
    }
    SCL_LAUNCH("vlad_dx_kernel", vlad_dx_kernel, dim3(B, pl.S), dim3(256), kVladDxLds, st, da);
#ifdef SCL_DIAG
    if (old_launches()) {
      SCL_LAUNCH("vlad_wgrad_partial_kernel", vlad_wgrad_partial_kernel, dim3(32, VW_GROUPS), dim3(256),
                 0, st, (const float*)w.wpart, pl.S * B, w.wpartial);
      SCL_LAUNCH("vlad_wgrad_finish_kernel", vlad_wgrad_finish_kernel, dim3(32), dim3(256), 0, st,
                 (const float*)w.wpartial, (const float*)w.du, (const float*)save_vlad, B, grad_w, grad_c);
    }
#endif
    return scl_launch_status();
  }
  RowTileArgs a{};
  a.x = x;
  a.bt = w.dut;
  a.bt_stride = (int64_t)D * K;
  a.B = B;
  a.N = N;
  a.pre_l2 = pre_l2 ? 1 : 0;
  a.a_in = save_assign;
  a.logit_in = save_logit;
  a.rn_in = save_rnorm;
  a.cdu = w.cdu;
  a.ds = w.ds;
  a.rowdot = w.rowdot;
  const dim3 dxgrid(((N + 15) / 16 + 3) / 4, B);
  if (x_dtype == SCL_DT_F32) {
    launch_rowtile<float, DASSIGN>(a, st);
    SCL_LAUNCH("aggregate_kernel<float>", aggregate_kernel<float>, dim3(D / 64, NSPLIT, B), dim3(256), 0,
               st, x, (const float*)w.ds, save_rnorm, N, w.wpart, (float*)nullptr);
    SCL_LAUNCH("dx16_kernel<float>", dx16_kernel<float>, dxgrid, dim3(256), kDx16Lds, st, x, save_assign,
               (const float*)w.ds, save_rnorm, (const float*)w.rowdot, (const float*)w.du, assign_w,
               N, pre_l2 ? 1 : 0, grad_x);
  } else {
    launch_rowtile<unsigned short, DASSIGN>(a, st);
    SCL_LAUNCH("aggregate_kernel<bf16>", aggregate_kernel<unsigned short>, dim3(D / 64, NSPLIT, B),
               dim3(256), 0, st, x, (const float*)w.ds, save_rnorm, N, w.wpart, (float*)nullptr);
    SCL_LAUNCH("dx16_kernel<bf16>", dx16_kernel<unsigned short>, dxgrid, dim3(256), kDx16Lds, st, x,
               save_assign, (const float*)w.ds, save_rnorm, (const float*)w.rowdot,
               (const float*)w.du, assign_w, N, pre_l2 ? 1 : 0, grad_x);
  }
  SCL_LAUNCH("wgrad_finish_kernel", wgrad_finish_kernel, dim3(D * K / 64), dim3(256), 0, st,
                     (const float*)w.wpart, (const float*)w.du, save_vlad, B, grad_w, grad_c);
  return scl_launch_status();
}

extern "C" int scl_netvlad_bwd(const void* x, int x_dtype, const float* assign_w,
                               const float* centers, const float* grad_out,
                               const float* save_assign, const float* save_logit,
                               const float* save_rnorm, float* save_vlad, int B, int N,
                               int pre_l2, void* grad_x, float* grad_w, float* grad_c,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return scl_netvlad_bwd_p(x, x_dtype, assign_w, centers, nullptr, grad_out, save_assign, save_logit,
                           save_rnorm, save_vlad, B, N, pre_l2, grad_x, grad_w, grad_c, workspace,
                           workspace_bytes, stream);
}
