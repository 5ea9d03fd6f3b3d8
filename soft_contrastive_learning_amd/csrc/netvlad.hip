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

#include "scl_common.h"

namespace {

constexpr int D = SCL_VLAD_D;   // 512
constexpr int K = SCL_VLAD_K;   // 64
constexpr int NSPLIT = 4;       // location splits with their own slab in aggregate_kernel

// ------------------------------------------------------------------ small kernels
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ w,
                                                          float* __restrict__ wt) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // over D*K, k fastest
  if (idx < D * K) wt[(idx % K) * D + idx / K] = w[idx];
}

// x = h1 + h2 + h3 exactly (up to the float32 subnormal range): three bf16 roundings.
__device__ __forceinline__ void split3_bf16(float x, unsigned short& h1, unsigned short& h2,
                                            unsigned short& h3) {
  h1 = f32_to_bf16(x);
  const float r1 = x - bf16_to_f32(h1);
  h2 = f32_to_bf16(r1);
  h3 = f32_to_bf16(r1 - bf16_to_f32(h2));
}

enum RowMode { ASSIGN = 0, DASSIGN = 1 };

struct RowTileArgs {
  const void* x;       // [B,N,512]
  const float* bt;     // ASSIGN: Wt [64][512];  DASSIGN: dUt [B][64][512]
  int64_t bt_stride;   // floats between images (0 for ASSIGN)
  const unsigned short* btp;   // bf16 input: the same operand as bf16x3 chunk images (split_w_kernel)
  int64_t btp_stride;          // elements between images (0 for ASSIGN)
  int dbg;                     // timing ablations (scl_debug_set_variant 21 / 22), 0 = production
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16b(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// Operand planes of the bf16x3 row-tile kernel.  The float32 operand O[k = cluster][c = channel]
// (W^T, or dU^T of an image) is stored as the kernel's LDS images, chunk after chunk:
//   16-byte unit ((chunk * 3 + plane) * 8 + piece) * 64 + k  =  O_plane[k][64 chunk + 8 piece .. + 7]
// so a 64-channel chunk (3 planes x 8 pieces x 64 clusters x 16 B = 24 KB) is copied into LDS
// by LDS-DMA exactly as it lies, and the B fragment of lane (i, g) for k-step s2 and cluster
// tile kt sits at unit (plane * 8 + 4 s2 + g) * 64 + 16 kt + i: inside every ds_read_b128
// service group ({0-3, 12-15, 20-27}, ... = each value of i once) the 16 lanes hit 16 different
// 16-byte slots — conflict-free, no padding.
constexpr int RC_UNITS = 3 * 8 * 64;                       // 16-byte units per chunk image
constexpr int RC_IMG = RC_UNITS * 8;                       // bf16 per chunk image (24 KB)

// W [512][64] float32 -> chunk images of W^T.  grid 16, block 256: thread = (8-channel piece
// c8, cluster k): eight strided reads (coalesced over k), one 16-byte store per plane.
__global__ __launch_bounds__(256) void split_w_kernel(const float* __restrict__ w,
                                                      unsigned short* __restrict__ planes) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // over 64 pieces x 64 clusters
  if (idx >= (D / 8) * K) return;
  const int k = idx % K, c8 = idx / K;
  unsigned short h[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3_bf16(w[(c8 * 8 + j) * K + k], h[0][j], h[1][j], h[2][j]);
  const int chunk = c8 >> 3, piece = c8 & 7;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    uint4 v;
    v.x = (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16);
    v.y = (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16);
    v.z = (unsigned)h[pl][4] | ((unsigned)h[pl][5] << 16);
    v.w = (unsigned)h[pl][6] | ((unsigned)h[pl][7] << 16);
    reinterpret_cast<uint4*>(planes)[((chunk * 3 + pl) * 8 + piece) * 64 + k] = v;
  }
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
__device__ __forceinline__ unsigned nv_lds_byte_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// rowtile_ring_kernel: [128 locations] x [512 channels] x [64 clusters] per workgroup; a wave
// owns 32 locations = two 16-row MFMA tiles that share every B fragment read.
//
// The kernel is bound by memory latency, not by matrix or LDS time (a compute step of 48 MFMAs
// is 0.3 us, a loaded HBM round trip 1-2 us), so EVERYTHING it reads arrives by LDS-DMA
// through one 3-stage ring, two 64-channel stages (80 KB per CU) ahead of the matrix work:
//   stage = operand chunk image (24 KB, see split_w_kernel) + the workgroup's x slice
//           [128 locations][64 channels] bf16 (16 KB);
//   x rows are 128 bytes; 16-byte piece q of row r is stored at position q ^ ((r >> 1) & 7)
//   (the DMA fetches piece j ^ ((r >> 1) & 7) into position j: the swizzle sits on the source
//   address, the LDS destination is lane-linear) so the 16 lanes of every ds_read_b128 service
//   group of an A fragment hit 16 different slots;
//   one queue, counted waits: s_waitcnt vmcnt(10) leaves the next stage in flight; one
//   barrier per chunk orders landed stages against readers and frees the stage read last.
// grid (ceil(N / 128), B); block 256; dynamic LDS 3 x 40 KB (the epilogue scratch reuses it).
constexpr int RG_STAGE = RC_IMG * 2 + 128 * 128;           // bytes per ring stage (40,960)
constexpr int RG_NST = 3;
constexpr size_t kRowTileRingLds = (size_t)RG_NST * RG_STAGE;
static_assert(kRowTileRingLds >= 4 * 16 * 68 * sizeof(float), "epilogue scratch must fit");

template <int MODE>
__global__ __launch_bounds__(256, 1) void rowtile_ring_kernel(RowTileArgs p) {
  extern __shared__ __attribute__((aligned(16))) float bt_lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * 128 + wid * 32;
  const bool active = n0 < p.N;          // wave-uniform; idle waves still stage and sync
  const unsigned short* src = p.btp + (int64_t)b * p.btp_stride;
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(p.x) + (int64_t)b * p.N * D;
  const unsigned lds0 = nv_lds_byte_of(bt_lds);
  const bool row_ok[2] = {n0 + i < p.N, n0 + 16 + i < p.N};

  // x DMA of this lane: instruction v covers the wave's rows 8 v .. 8 v + 7, lane = (r, j)
  const unsigned short* xsrc[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = 8 * v + (lane >> 3), j = lane & 7;
    int n = n0 + row;
    n = n < p.N ? n : p.N - 1;                             // rows past the end re-read the last
    xsrc[v] = xb + (int64_t)n * D + ((j ^ ((row >> 1) & 7)) << 3);
  }
  auto stage = [&](int chunk) {
    const unsigned base = lds0 + (chunk % RG_NST) * RG_STAGE;
#pragma unroll
    for (int v = 0; v < 6; ++v) {
      const int piece = v * 4 + wid;                       // 24 one-KB pieces per operand image
      nv_glds16(src + (int64_t)chunk * RC_IMG + piece * 512 + lane * 8, base + piece * 1024);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v)
      nv_glds16(xsrc[v] + chunk * 64, base + RC_IMG * 2 + (wid * 32 + 8 * v) * 128);
  };

  f32x4 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ss[2] = {0.f, 0.f};

  const int nchunks = p.dbg == 22 ? 0 : D / 64;          // 22: epilogue only
  if (nchunks) {
    stage(0);
    stage(1);
  }
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < D / 64)
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // stage c landed, c + 1 in flight
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < D / 64) stage(c + 2);
    if (active) {
      const char* sb = reinterpret_cast<const char*>(bt_lds) + (c % RG_NST) * RG_STAGE;
      // B fragment of (plane, k-step s2, cluster tile kt): unit (plane*8 + 4 s2 + g)*64 + 16 kt + i
      const char* wb = sb + (g * 64 + i) * 16;
      // A fragment of (tile t, k-step s2): row 16 t + i of the wave, piece 4 s2 + g, swizzled
      const char* xa_base = sb + RC_IMG * 2 + (wid * 32 + i) * 128;
      const int sw = (i >> 1) & 7;
      u32x4 xa[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          xa[t][s2] = *reinterpret_cast<const u32x4*>(xa_base + t * 16 * 128 + (((4 * s2 + g) ^ sw) << 4));
          if (!row_ok[t]) xa[t][s2] = u32x4{0u, 0u, 0u, 0u};
        }
      u32x4 wv[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        wv[0][pl] = *reinterpret_cast<const u32x4*>(wb + pl * 8 * 64 * 16);
#pragma unroll
      for (int q = 0; q < 8; ++q) {            // q = 4 * k-step + cluster tile
        const int s2 = q >> 2, kt = q & 3;
        if (q + 1 < 8) {
          const int s3 = (q + 1) >> 2, kt3 = (q + 1) & 3;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            wv[(q + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(
                wb + ((pl * 8 + 4 * s3) * 64 + 16 * kt3) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            acc[t][kt] = mfma16b(xa[t][s2], wv[q & 1][pl], acc[t][kt]);
          if (MODE == ASSIGN && kt == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned w = xa[t][s2][e];
              const float lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xffff0000u);
              ss[t] = fmaf(lo, lo, ss[t]);
              ss[t] = fmaf(hi, hi, ss[t]);
            }
          }
        }
      }
    }
  }
  if (!active) return;
  if (p.dbg == 21) {                                       // 21: main loop only
    float sacc = ss[0] + ss[1];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) sacc += acc[t][kt][0] + acc[t][kt][1] + acc[t][kt][2] + acc[t][kt][3];
    if (sacc == 1.2345e-7f) p.rnorm[0] = sacc;
    return;
  }
  // stage 0's bytes were last read in chunk 6 and every wave has passed chunk 7's barrier
  rowtile_epilogue<MODE>(p, acc[0], ss[0], bt_lds, b, n0);
  if (n0 + 16 < p.N) rowtile_epilogue<MODE>(p, acc[1], ss[1], bt_lds, b, n0 + 16);
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

// per-image normalisation state shared by finish (forward) and bwd_prep (backward)
// ---------------------------------------------------------------------------------------
// aggregate16b_kernel: V_part[b, split, d, k] = sum_{n in split} x[b,n,d] cf[b,n,k] for a bf16
// feature map on v_mfma_f32_16x16x32_bf16, cf = (a or ds) * rn.  The float32 coefficients are
// read as the row-tile kernel saved them ([n][64] rows) and split into three bf16 planes while
// they are staged — thread (cluster k, location octet) takes 8 strided values (a wave reads
// whole 256-byte rows), scales by rn, and writes one 16-byte piece per plane — so no plane
// copy of them ever goes through HBM (it was 11 MB written and read per pass).
// The contraction runs over n, the SLOW index of x[n][d]: each wave stages its [32 n][64 d]
// piece of x row-major in LDS and reads the A fragments with ds_read_b64_tr_b16 (4 rows x 16
// columns delivered column-major).
// grid (2 channel halves, NSPLIT, B); block 256: the four waves share a 32-location step
// (its cf planes are staged once per workgroup) and own 64 channels each, so no cross-wave
// reduction is needed.  LDS: 2 x ([3][64][40] cf + 4 x [32][72] x) bf16 = 67,584 B.
constexpr int AB_XLD = 72;                                 // bf16 per staged x row (64 + 8 pad)
constexpr int AB_CFLD = 40;                                // bf16 per staged cf row (32 n + 8 pad:
                                                           // 16 lanes -> 16 different 16-B slots)
constexpr int AB_CF = 3 * K * AB_CFLD;                     // bf16 per staged cf step
constexpr int AB_X = 32 * AB_XLD;                          // bf16 per wave x tile
constexpr int AB_BUF = AB_CF + 4 * AB_X;                   // bf16 per buffer
constexpr size_t kAgg16bLds = 2 * (size_t)AB_BUF * sizeof(unsigned short);   // 67,584 B
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void aggregate16b_kernel(
    const unsigned short* __restrict__ x, const float* __restrict__ coefn,
    const float* __restrict__ rn, int N, float* __restrict__ part,
    float* __restrict__ colsum_part) {
  extern __shared__ __attribute__((aligned(16))) unsigned short ab_lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int half = blockIdx.x, split = blockIdx.y, b = blockIdx.z;
  const int d0 = half * 256 + wid * 64;                    // this wave's 64 channels
  const int nsteps = (N + 31) / 32;
  const int per = (nsteps + NSPLIT - 1) / NSPLIT;
  const int s_begin = split * per;
  const int s_end = s_begin + per < nsteps ? s_begin + per : nsteps;

  // staging registers: this thread's 8 coefficients (cluster `lane`, locations 8 wid .. + 7
  // of the step) with their row norms, and 4 sixteen-byte pieces of the wave's x tile
  float st_a[8], st_r[8];
  u32x4 st_x[4];
  const float* cfb_g = coefn + (int64_t)b * N * K + lane;
  const float* rnb_g = rn + (int64_t)b * N;
  auto stage_load = [&](int s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int n = 32 * s + 8 * wid + j;
      n = n < N ? n : N - 1;                               // masked when it is consumed
      st_a[j] = cfb_g[(int64_t)n * K];
      st_r[j] = rnb_g[n];
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 64 + lane;                       // 256 pieces: [row 32][8]
      int n = 32 * s + (idx >> 3);
      if (n >= N) n = N - 1;                               // cf is zero there
      st_x[v] = *reinterpret_cast<const u32x4*>(x + ((int64_t)b * N + n) * D + d0 + 8 * (idx & 7));
    }
  };
  float cs_part = 0.f;                                     // column sum of a (cluster `lane`)
  auto stage_store = [&](int buf, int s) {
    unsigned short* cfb = ab_lds + buf * AB_BUF;
    // staged as [plane * 64 + k][32 n (+ pad)]: this thread's piece = locations 8 wid .. + 7
    unsigned short h[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = 32 * s + 8 * wid + j < N;
      const float av = ok ? st_a[j] : 0.f;
      cs_part += av;
      split3_bf16(av * st_r[j], h[0][j], h[1][j], h[2][j]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      u32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        v[c] = (unsigned)h[pl][2 * c] | ((unsigned)h[pl][2 * c + 1] << 16);
      *reinterpret_cast<u32x4*>(cfb + (pl * K + lane) * AB_CFLD + 8 * wid) = v;
    }
    unsigned short* xb = cfb + AB_CF + wid * AB_X;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 64 + lane;
      *reinterpret_cast<u32x4*>(xb + (idx >> 3) * AB_XLD + 8 * (idx & 7)) = st_x[v];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[mt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // column sums of a over this split (forward only) come out of the staging: wave w has
  // added the locations 8 w .. 8 w + 7 of every step; the four partials meet in LDS below
  const bool want_cs = colsum_part != nullptr && half == 0;

  if (s_begin < s_end) {
    stage_load(s_begin);
    stage_store(0, s_begin);
  }
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const bool more = s + 1 < s_end;
    if (more) stage_load(s + 1);
    const unsigned short* cfb = ab_lds + buf * AB_BUF;
    const unsigned short* xb = cfb + AB_CF + wid * AB_X;
    // A fragments: channel 16 mt + i, locations 8g .. 8g+7 of the step (two transposed reads:
    // lane 4q + p of a 16-lane group addresses row q, columns 4p .. 4p+3 of the 4 x 16 block)
    u32x4 af[4];
    {
      const int q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const unsigned short* a0 = xb + (8 * g + q) * AB_XLD + 16 * mt + 4 * pp;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(a0 + 4 * AB_XLD));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        af[mt] = u32x4{l2.x, l2.y, h2.x, h2.y};
      }
    }
    u32x4 bf[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      bf[0][pl] = *reinterpret_cast<const u32x4*>(cfb + (pl * K + i) * AB_CFLD + 8 * g);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      if (kt + 1 < 4) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bf[(kt + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(
              cfb + (pl * K + 16 * (kt + 1) + i) * AB_CFLD + 8 * g);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          acc[mt][kt] = mfma16b(af[mt], bf[kt & 1][pl], acc[mt][kt]);
    }
    if (more) stage_store(buf ^ 1, s + 1);
    __syncthreads();
  }

  // slab: rows d0 + 16 mt + 4 g + reg, columns 16 kt + i
  float* out = part + (((int64_t)b * NSPLIT + split) * D + d0) * K;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        out[(16 * mt + 4 * g + reg) * K + 16 * kt + i] = acc[mt][kt][reg];
  if (want_cs) {   // every wave is past the last barrier: the staging buffers are free
    float* red = reinterpret_cast<float*>(ab_lds);
    red[wid * K + lane] = cs_part;
    __syncthreads();
    if (wid == 0)
      colsum_part[((int64_t)b * NSPLIT + split) * K + lane] =
          (red[lane] + red[K + lane]) + (red[2 * K + lane] + red[3 * K + lane]);
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
    vlad[((int64_t)b * (D + 1) + d) * K + k] = u;
    ss = fmaf(u, u, ss);
  }
  if (blk == 0 && dq == 0) vlad[((int64_t)b * (D + 1) + D) * K + k] = asum;
  colbuf[dq * 64 + k] = ss;
  __syncthreads();
  if (dq == 0)
    colsq_part[((int64_t)b * 8 + blk) * K + k] =
        (colbuf[k] + colbuf[64 + k]) + (colbuf[128 + k] + colbuf[192 + k]);
}

__global__ __launch_bounds__(256) void finish_norm_kernel(const float* __restrict__ vlad,
                                                          const float* __restrict__ colsq_part,
                                                          float* __restrict__ out) {
  const int blk = blockIdx.x, b = blockIdx.y, k = threadIdx.x & 63, dq = threadIdx.x >> 6;
  float col = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) col += colsq_part[((int64_t)b * 8 + j) * K + k];
  // matconvnetNormalize: x / sqrt(sum x^2 + 1e-12), epsilon inside the sqrt
  const float q = 1.0f / sqrtf(col + 1e-12f);
  // every one of the 4 waves holds all 64 columns: a wave sum gives sum_k (q_k^2 col_k)
  const float tot = wave_sum(q * q * col);
  const float g = 1.0f / sqrtf(tot + 1e-12f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = blk * 64 + dq * 16 + i;
    out[(int64_t)b * D * K + d * K + k] = vlad[((int64_t)b * (D + 1) + d) * K + k] * q * g;
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
    const float u = save_vlad[((int64_t)b * (D + 1) + d) * K + k];
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

// bwd_du_kernel: grid (8, B), block 256: dU of one 64-channel block in both layouts (and as
// three bf16 planes of dU^T when dplanes != NULL), c.dU from block 0.
__global__ __launch_bounds__(256) void bwd_du_kernel(const float* __restrict__ save_vlad,
                                                     const float* __restrict__ grad_out,
                                                     const float* __restrict__ dots,
                                                     float* __restrict__ du,
                                                     float* __restrict__ dut,
                                                     unsigned short* __restrict__ dplanes,
                                                     unsigned short* __restrict__ du2,
                                                     const float* __restrict__ assign_w,
                                                     unsigned short* __restrict__ w2,
                                                     float* __restrict__ cdu) {
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
    const float u = save_vlad[((int64_t)b * (D + 1) + d) * K + k];
    const float go = grad_out[(int64_t)b * D * K + d * K + k];
    vals[i] = cg * go - cu * u;
    du[((int64_t)b * D + d) * K + k] = vals[i];
  }
  if (dut) {   // transposed float32 copy (float32-MFMA row-tile kernel only)
    float* trow = dut + ((int64_t)b * K + k) * D + blk * 64 + dq * 16;
#pragma unroll
    for (int i = 0; i < 16; i += 4)
      *reinterpret_cast<f32x4*>(trow + i) = f32x4{vals[i], vals[i + 1], vals[i + 2], vals[i + 3]};
  }
  if (dplanes) {
    unsigned short h[3][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) split3_bf16(vals[i], h[0][i], h[1][i], h[2][i]);
    // chunk images (see split_w_kernel): chunk = blk, pieces 2 dq and 2 dq + 1, cluster k
    uint4* img = reinterpret_cast<uint4*>(dplanes + (int64_t)b * 3 * D * K);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      unsigned w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        w[i] = (unsigned)h[pl][2 * i] | ((unsigned)h[pl][2 * i + 1] << 16);
      img[((blk * 3 + pl) * 8 + 2 * dq) * 64 + k] = make_uint4(w[0], w[1], w[2], w[3]);
      img[((blk * 3 + pl) * 8 + 2 * dq + 1) * 64 + k] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
  if (du2) {
    // operands of dx16b_kernel: dU[b] and (from image 0's workgroups) W as high / low bf16
    // planes [plane][d][k]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int d = blk * 64 + dq * 16 + i;
      unsigned short h = f32_to_bf16(vals[i]);
      du2[((int64_t)b * 2 * D + d) * K + k] = h;
      du2[((int64_t)b * 2 * D + D + d) * K + k] = f32_to_bf16(vals[i] - bf16_to_f32(h));
      if (b == 0) {
        const float wv = assign_w[d * K + k];
        h = f32_to_bf16(wv);
        w2[d * K + k] = h;
        w2[(D + d) * K + k] = f32_to_bf16(wv - bf16_to_f32(h));
      }
    }
  }
  if (blk == 0 && dq == 0) cdu[b * K + k] = cg * bk - cu * dk;
}

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

// ---------------------------------------------------------------------------------------
// dx16b_kernel: the same tile for a bf16 feature map on v_mfma_f32_16x16x32_bf16.  grad_x is
// stored as bf16 (8 mantissa bits), so both operands are taken as TWO bf16 planes and
// A.B = A_hi.B_hi + A_hi.B_lo + A_lo.B_hi (|error| <= 1.2e-5 of the term magnitudes, three
// orders below the output rounding): 24 MFMAs of 16 cycles per 32-channel chunk instead of
// 64 of 32.  A = [a | ds] rows of the tile, split in registers; B = [dU[b] | W] planes
// [plane][d][k] written by bwd_du_kernel, staged per chunk as [plane][32 d][128 k (+8 pad)].
constexpr int DXB_LD = 136;                               // bf16 per staged row (128 k + 8 pad)
constexpr int DXB_CH = 64;                                // channels per staged chunk / barrier
constexpr int DXB_CHUNK = 2 * DXB_CH * DXB_LD;            // bf16 per buffer (two planes)
constexpr size_t kDx16bLds = 2 * (size_t)DXB_CHUNK * sizeof(unsigned short) +
                             4 * (size_t)DX_SCR * sizeof(float);   // 69,632 + 9,216 B

__device__ __forceinline__ void split2x8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float v0 = c < 2 ? a[2 * c] : b[2 * c - 4], v1 = c < 2 ? a[2 * c + 1] : b[2 * c - 3];
    const unsigned short h0 = f32_to_bf16(v0), h1 = f32_to_bf16(v1);
    hi[c] = (unsigned)h0 | ((unsigned)h1 << 16);
    lo[c] = (unsigned)f32_to_bf16(v0 - bf16_to_f32(h0)) |
            ((unsigned)f32_to_bf16(v1 - bf16_to_f32(h1)) << 16);
  }
}

__global__ __launch_bounds__(256, 2) void dx16b_kernel(const unsigned short* __restrict__ xin,
                                                       const float* __restrict__ a,
                                                       const float* __restrict__ ds,
                                                       const float* __restrict__ rn,
                                                       const float* __restrict__ rowdot,
                                                       const unsigned short* __restrict__ du2,
                                                       const unsigned short* __restrict__ w2,
                                                       int N, int pre_l2,
                                                       unsigned short* __restrict__ gxo) {
  extern __shared__ __attribute__((aligned(16))) float dx_lds[];
  unsigned short* bl = reinterpret_cast<unsigned short*>(dx_lds);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.y;
  const int n0 = (blockIdx.x * 4 + wid) * 16;
  const bool active = n0 < N;
  float* scr = dx_lds + (2 * DXB_CHUNK) / 2 + wid * DX_SCR;
  const unsigned short* dub = du2 + (int64_t)b * 2 * D * K;

  // A operand: k-step s (32 of the 128 contraction indices): lane (i, g) holds
  // [a | ds][n0 + i][32 s + 8 g .. + 7] as packed high and low bf16
  u32x4 ah[4], al[4];
  {
    const bool ok = active && n0 + i < N;
    const int64_t gr = (int64_t)b * N + (ok ? n0 + i : 0);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const float* src = (s2 < 2 ? a : ds) + gr * K + 32 * (s2 & 1) + 8 * g;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
      if (!ok) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      split2x8(v0, v1, ah[s2], al[s2]);
    }
  }
  // epilogue role of this lane: location row_e, channels 8 seg .. 8 seg + 7 of each chunk
  const int row_e = lane >> 2, seg = lane & 3;
  const bool ok_e = active && n0 + row_e < N;
  const int64_t gr_e = (int64_t)b * N + (ok_e ? n0 + row_e : 0);
  const float rn_e = pre_l2 ? rn[gr_e] : 1.0f;
  const float rd_e = rowdot[gr_e];
  const bool proj = pre_l2 && rn_e < 1.0e6f;
  const unsigned short* x = xin + gr_e * D + seg * 8;
  unsigned short* gx = gxo + gr_e * D + seg * 8;

  // staging: 2 planes x 64 channels x 16 sixteen-byte pieces (8 of dU, 8 of W) = 2048 pieces
  u32x4 st[8];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int pl = idx >> 10, dl = (idx >> 4) & 63, c = idx & 15;
      const unsigned short* src =
          c < 8 ? dub + ((int64_t)pl * D + chunk * DXB_CH + dl) * K + 8 * c
                : w2 + ((int64_t)pl * D + chunk * DXB_CH + dl) * K + 8 * (c - 8);
      st[v] = *reinterpret_cast<const u32x4*>(src);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int pl = idx >> 10, dl = (idx >> 4) & 63, c = idx & 15;
      *reinterpret_cast<u32x4*>(bl + buf * DXB_CHUNK + (pl * DXB_CH + dl) * DXB_LD + 8 * c) = st[v];
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < D / DXB_CH; ++c) {
    const bool more = c + 1 < D / DXB_CH;
    if (more) stage_load(c + 1);
    u32x4 xraw2[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    if (active && proj) {
      xraw2[0] = *reinterpret_cast<const u32x4*>(x + c * DXB_CH);
      xraw2[1] = *reinterpret_cast<const u32x4*>(x + c * DXB_CH + DX_CH);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {      // two 32-channel blocks per staged chunk
     const u32x4 xraw = xraw2[sub];
     if (active) {
      // B fragment of (channel tile dt, k-step s, plane pl): channel 16 dt + i, k 32 s + 8 g ..
      const unsigned short* wb =
          bl + (c & 1) * DXB_CHUNK + (DX_CH * sub + i) * DXB_LD + 8 * g;
      f32x4 acc[2];
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragments of step q + 1 (q = 2 * k-step + channel tile) fly under the MFMAs of step q
      u32x4 bq[2][2];
      bq[0][0] = *reinterpret_cast<const u32x4*>(wb);
      bq[0][1] = *reinterpret_cast<const u32x4*>(wb + DXB_CH * DXB_LD);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int s2 = q >> 1, dt = q & 1;
        if (q + 1 < 8) {
          const int s3 = (q + 1) >> 1, d3 = (q + 1) & 1;
          bq[(q + 1) & 1][0] =
              *reinterpret_cast<const u32x4*>(wb + (16 * d3) * DXB_LD + 32 * s3);
          bq[(q + 1) & 1][1] =
              *reinterpret_cast<const u32x4*>(wb + (DXB_CH + 16 * d3) * DXB_LD + 32 * s3);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[dt] = mfma16b(ah[s2], bq[q & 1][0], acc[dt]);
        acc[dt] = mfma16b(ah[s2], bq[q & 1][1], acc[dt]);
        acc[dt] = mfma16b(al[s2], bq[q & 1][0], acc[dt]);
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
          float xv8[8];
          Elem8<unsigned short>::cvt(xraw, xv8);
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) out[cc] -= xv8[cc] * f;
        }
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) out[cc] *= rn_e;
        Elem8<unsigned short>::st(gx + c * DXB_CH + DX_CH * sub, out);
      }
     }
    }
    if (more) stage_store((c + 1) & 1);
    __syncthreads();
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
    gc = fmaf(du[(int64_t)b * D * K + idx], save_vlad[((int64_t)b * (D + 1) + D) * K + k], gc);
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
  SCL_LAUNCH("rowtile_assign", (rowtile16_kernel<T, ASSIGN, VAR>), grid, dim3(256), kRowTile16Lds,
             st, a);
}
template <typename T>
void launch_rowtile_variant(const RowTileArgs& a, dim3 grid, hipStream_t st) {
  switch (scl_debug_variant & 7) {
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
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowtile16_kernel<T, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowTile16Lds);
  });
  const int tiles16 = (a.N + 15) / 16;
  const dim3 grid((tiles16 + 3) / 4, a.B);
  if (MODE == ASSIGN && scl_debug_variant >= 1 && scl_debug_variant <= 7) {   // ablate_rowtile.py
    launch_rowtile_variant<T>(a, grid, st);
    return;
  }
  SCL_LAUNCH(MODE == ASSIGN ? "rowtile_assign" : "rowtile_dassign", (rowtile16_kernel<T, MODE>),
             grid, dim3(256), kRowTile16Lds, st, a);
}

// bf16 feature maps: the bf16x3 kernel (scl_debug_set_variant(8) forces the float32-MFMA
// kernel on them for A/B timing and parity runs)
template <int MODE>
void launch_rowtile_b3(const RowTileArgs& a, hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowtile_ring_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRowTileRingLds);
  });
  RowTileArgs ad = a;
  ad.dbg = (scl_debug_variant == 21 || scl_debug_variant == 22) ? scl_debug_variant : 0;
  SCL_LAUNCH(MODE == ASSIGN ? "rowtile_assign" : "rowtile_dassign", (rowtile_ring_kernel<MODE>),
             dim3((a.N + 127) / 128, a.B), dim3(256), kRowTileRingLds, st, ad);
}
inline void launch_aggregate_b3(const char* name, const void* x, const float* coefn,
                                const float* rn, int B, int N, float* part, float* colsum,
                                hipStream_t st) {
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aggregate16b_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAgg16bLds);
  });
  SCL_LAUNCH(name, aggregate16b_kernel, dim3(2, NSPLIT, B), dim3(256), kAgg16bLds, st,
             (const unsigned short*)x, coefn, rn, N, part, colsum);
}
inline bool use_b3() { return scl_debug_variant < 1 || scl_debug_variant > 8; }

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

struct FwdWs {
  float *wt, *part, *colsum, *colsq, *vlad, *assign, *rnorm;
  unsigned short* wplanes;   // bf16x3 chunk images of W^T (3 * 64 * 512 bf16)
  size_t total;
};
inline FwdWs carve_fwd(void* ws, int B, int N) {
  Carver c(ws);
  FwdWs w;
  w.wt = c.take((size_t)D * K);
  w.part = c.take((size_t)B * NSPLIT * D * K);
  w.colsum = c.take((size_t)B * NSPLIT * K);
  w.colsq = c.take((size_t)B * 8 * K);
  w.vlad = c.take((size_t)B * (D + 1) * K);
  w.assign = c.take((size_t)B * N * K);
  w.rnorm = c.take((size_t)B * N);
  w.wplanes = (unsigned short*)c.take((size_t)3 * D * K / 2);
  w.total = c.off;
  return w;
}

struct BwdWs {
  float *du, *dut, *cdu, *ds, *rowdot, *wpart, *dots;
  unsigned short* dplanes;   // [B] bf16x3 chunk images of dU^T
  unsigned short* du2;       // [B][2][512][64] bf16 planes of dU
  unsigned short* w2;        // [2][512][64] bf16 planes of W
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
  w.wpart = c.take((size_t)B * NSPLIT * D * K);
  w.dplanes = (unsigned short*)c.take((size_t)B * 3 * D * K / 2);
  w.dots = c.take((size_t)B * 8 * 4 * K);
  w.du2 = (unsigned short*)c.take((size_t)B * 2 * D * K / 2);
  w.w2 = (unsigned short*)c.take((size_t)2 * D * K / 2);
  w.total = c.off;
  return w;
}

inline bool shape_ok(int B, int N) { return B >= 1 && N >= 1 && B <= 65535 && N <= (1 << 22); }

}  // namespace

extern "C" size_t scl_netvlad_fwd_workspace_bytes(int B, int N) {
  if (!shape_ok(B, N)) return 0;
  return carve_fwd(nullptr, B, N).total;
}

extern "C" int scl_netvlad_fwd(const void* x, int x_dtype, const float* assign_w,
                               const float* centers, int B, int N, int pre_l2, float* out,
                               float* save_assign, float* save_logit, float* save_rnorm,
                               float* save_vlad, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!x || !assign_w || !centers || !out || !workspace) return SCL_E_NULL;
  if (!shape_ok(B, N)) return SCL_E_SHAPE;
  if (x_dtype != SCL_DT_F32 && x_dtype != SCL_DT_BF16) return SCL_E_KIND;
  if (((uintptr_t)x % 16) != 0) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  FwdWs w = carve_fwd(workspace, B, N);
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* assign = save_assign ? save_assign : w.assign;
  float* rnorm = save_rnorm ? save_rnorm : w.rnorm;

  const bool fwd_b3 = x_dtype == SCL_DT_BF16 && use_b3();
  if (!fwd_b3)
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
    SCL_LAUNCH("aggregate_kernel", aggregate_kernel<float>, dim3(D / 64, NSPLIT, B), dim3(256), 0, st, x,
                       (const float*)assign, (const float*)rnorm, N, w.part, w.colsum);
  } else {
    if (fwd_b3) {
      SCL_LAUNCH("split_w_kernel", split_w_kernel, dim3((D / 8) * K / 256), dim3(256), 0, st,
                 assign_w, w.wplanes);
      a.btp = w.wplanes;
      a.btp_stride = 0;
      launch_rowtile_b3<ASSIGN>(a, st);
      launch_aggregate_b3("aggregate_kernel", x, (const float*)assign, (const float*)rnorm, B, N,
                          w.part, w.colsum, st);
    } else {
      launch_rowtile<unsigned short, ASSIGN>(a, st);
      SCL_LAUNCH("aggregate_kernel", aggregate_kernel<unsigned short>, dim3(D / 64, NSPLIT, B), dim3(256), 0, st,
                         x, (const float*)assign, (const float*)rnorm, N, w.part, w.colsum);
    }
  }
  float* vlad = save_vlad ? save_vlad : w.vlad;
  SCL_LAUNCH("finish_sum_kernel", finish_sum_kernel, dim3(8, B), dim3(256), 0, st,
             (const float*)w.part, (const float*)w.colsum, centers, vlad, w.colsq);
  SCL_LAUNCH("finish_norm_kernel", finish_norm_kernel, dim3(8, B), dim3(256), 0, st,
             (const float*)vlad, (const float*)w.colsq, out);
  return scl_launch_status();
}

extern "C" size_t scl_netvlad_bwd_workspace_bytes(int B, int N) {
  if (!shape_ok(B, N)) return 0;
  return carve_bwd(nullptr, B, N).total;
}

extern "C" int scl_netvlad_bwd(const void* x, int x_dtype, const float* assign_w,
                               const float* centers, const float* grad_out,
                               const float* save_assign, const float* save_logit,
                               const float* save_rnorm, const float* save_vlad, int B, int N,
                               int pre_l2, void* grad_x, float* grad_w, float* grad_c,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !assign_w || !centers || !grad_out || !save_assign || !save_logit || !save_rnorm ||
      !save_vlad || !grad_x || !grad_w || !grad_c || !workspace)
    return SCL_E_NULL;
  if (!shape_ok(B, N)) return SCL_E_SHAPE;
  if (x_dtype != SCL_DT_F32 && x_dtype != SCL_DT_BF16) return SCL_E_KIND;
  if (((uintptr_t)x % 16) != 0) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  BwdWs w = carve_bwd(workspace, B, N);
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;

  const bool b3 = x_dtype == SCL_DT_BF16 && use_b3();
  SCL_LAUNCH("bwd_dots_kernel", bwd_dots_kernel, dim3(8, B), dim3(256), 0, st, save_vlad, grad_out,
             centers, w.dots);
  SCL_LAUNCH("bwd_du_kernel", bwd_du_kernel, dim3(8, B), dim3(256), 0, st, save_vlad, grad_out,
             (const float*)w.dots, w.du, b3 ? (float*)nullptr : w.dut,
             b3 ? w.dplanes : (unsigned short*)nullptr,
             b3 ? w.du2 : (unsigned short*)nullptr, assign_w, w.w2, w.cdu);
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
    SCL_LAUNCH("aggregate_dw", aggregate_kernel<float>, dim3(D / 64, NSPLIT, B), dim3(256), 0, st, x,
                       (const float*)w.ds, save_rnorm, N, w.wpart, (float*)nullptr);
    SCL_LAUNCH("dx_kernel", dx16_kernel<float>, dxgrid, dim3(256), kDx16Lds, st, x, save_assign,
                       (const float*)w.ds, save_rnorm, (const float*)w.rowdot,
                       (const float*)w.du, assign_w, N, pre_l2 ? 1 : 0, grad_x);
  } else {
    if (b3) {
      a.btp = w.dplanes;
      a.btp_stride = (int64_t)3 * D * K;
      launch_rowtile_b3<DASSIGN>(a, st);
      launch_aggregate_b3("aggregate_dw", x, (const float*)w.ds, save_rnorm, B, N, w.wpart,
                          nullptr, st);
    } else {
      launch_rowtile<unsigned short, DASSIGN>(a, st);
      SCL_LAUNCH("aggregate_dw", aggregate_kernel<unsigned short>, dim3(D / 64, NSPLIT, B), dim3(256), 0, st,
                         x, (const float*)w.ds, save_rnorm, N, w.wpart, (float*)nullptr);
    }
    if (b3) {
      SCL_LAUNCH("dx_kernel", dx16b_kernel, dxgrid, dim3(256), kDx16bLds, st,
                 (const unsigned short*)x, save_assign, (const float*)w.ds, save_rnorm,
                 (const float*)w.rowdot, (const unsigned short*)w.du2,
                 (const unsigned short*)w.w2, N, pre_l2 ? 1 : 0, (unsigned short*)grad_x);
    } else {
      SCL_LAUNCH("dx_kernel", dx16_kernel<unsigned short>, dxgrid, dim3(256), kDx16Lds, st, x, save_assign,
                         (const float*)w.ds, save_rnorm, (const float*)w.rowdot,
                         (const float*)w.du, assign_w, N, pre_l2 ? 1 : 0, grad_x);
    }
  }
  SCL_LAUNCH("wgrad_finish_kernel", wgrad_finish_kernel, dim3(D * K / 64), dim3(256), 0, st,
                     (const float*)w.wpart, (const float*)w.du, save_vlad, B, grad_w, grad_c);
  return scl_launch_status();
}
