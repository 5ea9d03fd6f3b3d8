// Elementwise glue of the VGG16 backbone on gfx950 (channels-last activations).
//
// The reference's layers are tf.layers.conv2d / max_pooling2d / tf.nn.relu with TF autodiff
// (model/nets.py:27-63).  In the bf16 step every convolution is an own kernel (conv64.hip,
// convh.hip, convg.hip) that fuses most of this glue into its epilogue; the passes below serve
// what is left (the un-pooling by window index, the bias-gradient column sums where no
// weight-gradient kernel produces them) and the float32 / small-map mode, where the
// convolutions run in the library.  Around library convolutions PyTorch
// launches one kernel per elementwise op — bias add, ReLU, max-pool, their backward ops and
// a per-channel reduction for every bias gradient — and each streams the activation map
// (up to 943 MB at 24 x 480 x 640 x 64 bf16) through HBM again; pool backward additionally
// reads int64 indices.  These kernels are HBM-bound, so the fix is fewer passes:
//   bias_act_kernel       y = [relu](y + bias) in place                    (1 R + 1 W)
//   act_bwd_kernel        gz = g * [a > 0]  and per-channel partial sums   (2 R + 1 W)
//   pool_fwd_kernel       a = relu(maxpool2x2(z) + bias)                   (1 R + 1/4 W)
//   pool_bwd_kernel       gz = route(g * [a > 0]) to the window's first max, recomputed
//                         from z (no index tensor), + partial sums         (1.5 R + 1 W)
//   colsum_kernel         partial sums -> bias gradient
// All accesses are 16 bytes per lane along the channel axis (8 bf16 / 4+4 f32).
// max-pool and ReLU commute exactly, so relu(pool(z) + b) == the reference's
// pool(z + b) followed by ReLU (model/nets.py:40-42).
#include "scl_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Vec8;
template <>
struct Vec8<float> {
  static __device__ __forceinline__ void ld(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c] = a[c];
      v[4 + c] = b[c];
    }
  }
  static __device__ __forceinline__ void st(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
template <>
struct Vec8<unsigned short> {
  static __device__ __forceinline__ void ld(const unsigned short* p, float* v) {
    const u32x4 w = *reinterpret_cast<const u32x4*>(p);
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

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;   // 4 workgroups per CU; also the rows of the partial buffer

// Every thread keeps one fixed 8-channel group: C/8 divides 256 and the grid stride is a
// multiple of 256, so bias is loaded once and channel partial sums stay in registers.
template <typename T>
__global__ __launch_bounds__(kThreads) void bias_act_kernel(T* __restrict__ y,
                                                            const float* __restrict__ bias,
                                                            int64_t nvec, int c8n, int relu) {
  const int c8 = threadIdx.x % c8n;
  float b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) b[c] = bias[c8 * 8 + c];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * kThreads) {
    float v[8];
    Vec8<T>::ld(y + i * 8, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      v[c] += b[c];
      if (relu) v[c] = fmaxf(v[c], 0.f);
    }
    Vec8<T>::st(y + i * 8, v);
  }
}

// block-level reduction of per-thread channel partials -> partial[blockIdx][C]
__device__ __forceinline__ void write_partials(const float* acc, int c8n, float* lds,
                                               float* __restrict__ partial, int C) {
  // lds: [kThreads][8]
#pragma unroll
  for (int c = 0; c < 8; ++c) lds[threadIdx.x * 8 + c] = acc[c];
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += kThreads) {
    const int c8 = ch >> 3, c = ch & 7;
    float s = 0.f;
    for (int t = c8; t < kThreads; t += c8n) s += lds[t * 8 + c];
    partial[(int64_t)blockIdx.x * C + ch] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void act_bwd_kernel(const T* __restrict__ g,
                                                           const T* __restrict__ a,
                                                           T* __restrict__ gz, int64_t nvec,
                                                           int c8n, int C,
                                                           float* __restrict__ partial) {
  __shared__ float lds[kThreads * 8];
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * kThreads) {
    float gv[8];
    Vec8<T>::ld(g + i * 8, gv);
    if (a) {
      float av[8];
      Vec8<T>::ld(a + i * 8, av);
#pragma unroll
      for (int c = 0; c < 8; ++c) gv[c] = av[c] > 0.f ? gv[c] : 0.f;
      Vec8<T>::st(gz + i * 8, gv);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] += gv[c];
  }
  write_partials(acc, c8n, lds, partial, C);
}

// a[b, ho, wo, :] = relu(max over the 2x2 window of z + bias); 'valid' pooling.
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_fwd_kernel(const T* __restrict__ z,
                                                            const float* __restrict__ bias,
                                                            int H, int W, int Ho, int Wo, int c8n,
                                                            int64_t nvec_out, T* __restrict__ a) {
  const int c8 = threadIdx.x % c8n;
  float b[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) b[c] = bias[c8 * 8 + c];
  const int64_t C = (int64_t)c8n * 8;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec_out;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t pos = i / c8n;                 // (b, ho, wo)
    const int wo = (int)(pos % Wo);
    const int64_t bh = pos / Wo;
    const int ho = (int)(bh % Ho);
    const int64_t bb = bh / Ho;
    const T* src = z + ((bb * H + 2 * ho) * W + 2 * wo) * C + c8 * 8;
    float m[8], v[8];
    Vec8<T>::ld(src, m);
    Vec8<T>::ld(src + C, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], v[c]);
    Vec8<T>::ld(src + (int64_t)W * C, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = fmaxf(m[c], v[c]);
    Vec8<T>::ld(src + (int64_t)W * C + C, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = fmaxf(fmaxf(m[c], v[c]) + b[c], 0.f);
    Vec8<T>::st(a + i * 8, m);
  }
}

// gz over the 2x2 window of every pooled position: the gradient g * [a > 0] goes to the
// FIRST maximal element in raster order (what max_pool2d's argmax does); the rest get 0.
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_bwd_kernel(const T* __restrict__ g,
                                                            const T* __restrict__ a,
                                                            const T* __restrict__ z, int H, int W,
                                                            int Ho, int Wo, int c8n, int Cn,
                                                            int64_t nvec_out, T* __restrict__ gz,
                                                            float* __restrict__ partial) {
  __shared__ float lds[kThreads * 8];
  const int c8 = threadIdx.x % c8n;
  const int64_t C = (int64_t)c8n * 8;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec_out;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t pos = i / c8n;
    const int wo = (int)(pos % Wo);
    const int64_t bh = pos / Wo;
    const int ho = (int)(bh % Ho);
    const int64_t bb = bh / Ho;
    const int64_t base = ((bb * H + 2 * ho) * W + 2 * wo) * C + c8 * 8;
    float gv[8], av[8], z0[8], z1[8], z2[8], z3[8];
    Vec8<T>::ld(g + i * 8, gv);
    Vec8<T>::ld(a + i * 8, av);
    Vec8<T>::ld(z + base, z0);
    Vec8<T>::ld(z + base + C, z1);
    Vec8<T>::ld(z + base + (int64_t)W * C, z2);
    Vec8<T>::ld(z + base + (int64_t)W * C + C, z3);
    float o0[8], o1[8], o2[8], o3[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float gg = av[c] > 0.f ? gv[c] : 0.f;
      acc[c] += gg;
      const float m = fmaxf(fmaxf(z0[c], z1[c]), fmaxf(z2[c], z3[c]));
      const bool h0 = z0[c] == m;
      const bool h1 = !h0 && z1[c] == m;
      const bool h2 = !h0 && !h1 && z2[c] == m;
      const bool h3 = !h0 && !h1 && !h2;
      o0[c] = h0 ? gg : 0.f;
      o1[c] = h1 ? gg : 0.f;
      o2[c] = h2 ? gg : 0.f;
      o3[c] = h3 ? gg : 0.f;
    }
    Vec8<T>::st(gz + base, o0);
    Vec8<T>::st(gz + base + C, o1);
    Vec8<T>::st(gz + base + (int64_t)W * C, o2);
    Vec8<T>::st(gz + base + (int64_t)W * C + C, o3);
  }
  write_partials(acc, c8n, lds, partial, Cn);
}

// The same routing from a stored window position (idx in 0..3 = 2 * dy + dx, one byte per
// pooled element, written by the convolution's pooling epilogue): no read of the full-size z.
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_bwd_idx_kernel(
    const T* __restrict__ g, const T* __restrict__ a, const unsigned char* __restrict__ idx,
    int H, int W, int Ho, int Wo, int c8n, int Cn, int64_t nvec_out, T* __restrict__ gz,
    float* __restrict__ partial) {
  __shared__ float lds[kThreads * 8];
  const int c8 = threadIdx.x % c8n;
  const int64_t C = (int64_t)c8n * 8;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec_out;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t pos = i / c8n;
    const int wo = (int)(pos % Wo);
    const int64_t bh = pos / Wo;
    const int ho = (int)(bh % Ho);
    const int64_t bb = bh / Ho;
    const int64_t base = ((bb * H + 2 * ho) * W + 2 * wo) * C + c8 * 8;
    float gv[8], av[8];
    Vec8<T>::ld(g + i * 8, gv);
    if (a) {
      Vec8<T>::ld(a + i * 8, av);
    } else {                                  // g arrives masked (the producer's epilogue did it)
#pragma unroll
      for (int c = 0; c < 8; ++c) av[c] = 1.f;
    }
    const uint2 iw = *reinterpret_cast<const uint2*>(idx + i * 8);
    float o0[8], o1[8], o2[8], o3[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float gg = av[c] > 0.f ? gv[c] : 0.f;
      acc[c] += gg;
      const unsigned k = ((c < 4 ? iw.x : iw.y) >> (8 * (c & 3))) & 3u;
      o0[c] = k == 0 ? gg : 0.f;
      o1[c] = k == 1 ? gg : 0.f;
      o2[c] = k == 2 ? gg : 0.f;
      o3[c] = k == 3 ? gg : 0.f;
    }
    Vec8<T>::st(gz + base, o0);
    Vec8<T>::st(gz + base + C, o1);
    Vec8<T>::st(gz + base + (int64_t)W * C, o2);
    Vec8<T>::st(gz + base + (int64_t)W * C + C, o3);
  }
  write_partials(acc, c8n, lds, partial, Cn);
}

// rows / columns that no 2x2 window covers (odd H or W) get a zero gradient
template <typename T>
__global__ __launch_bounds__(kThreads) void pool_bwd_border_kernel(T* __restrict__ gz, int B,
                                                                   int H, int W, int Ho, int Wo,
                                                                   int C) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t pos = i / C;
    const int w = (int)(pos % W), h = (int)((pos / W) % H);
    if (h >= 2 * Ho || w >= 2 * Wo) gz[i] = (T)0;
  }
}

// partial[nblocks][C] -> out[C].  grid C/32 (C % 32 == 0 is not required), block 1024:
// thread (rg = t >> 5, ch = t & 31) adds the rows rg, rg + 32, ... of its channel, then the
// 32 row groups are combined in fixed order (a single serial loop per channel took 0.5 ms).
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ partial,
                                                      int nblocks, int C,
                                                      float* __restrict__ out) {
  __shared__ float red[32][33];
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
  float s = 0.f;
  if (ch < C) {
#pragma unroll 8
    for (int b = rg; b < nblocks; b += 32) s += partial[(int64_t)b * C + ch];
  }
  red[rg][threadIdx.x & 31] = s;
  __syncthreads();
  if (rg == 0 && ch < C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][threadIdx.x & 31];
    out[ch] = t;
  }
}

inline int blocks_for(int64_t nvec) {
  int64_t b = (nvec + kThreads - 1) / kThreads;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return b < 1 ? 1 : (int)b;
}
inline bool channels_ok(int C) { return C >= 8 && C % 8 == 0 && kThreads % (C / 8) == 0; }
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace

extern "C" size_t scl_vgg_workspace_bytes(int C) {
  if (!channels_ok(C)) return 0;
  return scl_round256((size_t)kMaxBlocks * C * sizeof(float));
}

extern "C" int scl_vgg_bias_act(void* y, int dtype, const float* bias, int64_t M, int C, int relu,
                                void* stream) {
  if (!y || !bias) return SCL_E_NULL;
  if (M < 1 || !channels_ok(C) || !aligned16(y)) return SCL_E_SHAPE;
  const int64_t nvec = M * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SCL_DT_F32)
    SCL_LAUNCH("bias_act_kernel", bias_act_kernel<float>, dim3(blocks_for(nvec)), dim3(kThreads), 0,
               st, (float*)y, bias, nvec, C / 8, relu);
  else if (dtype == SCL_DT_BF16)
    SCL_LAUNCH("bias_act_kernel", bias_act_kernel<unsigned short>, dim3(blocks_for(nvec)),
               dim3(kThreads), 0, st, (unsigned short*)y, bias, nvec, C / 8, relu);
  else
    return SCL_E_KIND;
  return scl_launch_status();
}

extern "C" int scl_vgg_act_bwd(const void* g, const void* a, int dtype, int64_t M, int C, void* gz,
                               float* bias_grad, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (!g || !bias_grad || !workspace || (a && !gz)) return SCL_E_NULL;
  if (M < 1 || !channels_ok(C) || !aligned16(g) || !aligned16(a) || !aligned16(gz))
    return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_vgg_workspace_bytes(C))
    return SCL_E_WORKSPACE;
  const int64_t nvec = M * (C / 8);
  const int nb = blocks_for(nvec);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  if (dtype == SCL_DT_F32)
    SCL_LAUNCH("act_bwd_kernel", act_bwd_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
               (const float*)g, (const float*)a, (float*)gz, nvec, C / 8, C, partial);
  else if (dtype == SCL_DT_BF16)
    SCL_LAUNCH("act_bwd_kernel", act_bwd_kernel<unsigned short>, dim3(nb), dim3(kThreads), 0, st,
               (const unsigned short*)g, (const unsigned short*)a, (unsigned short*)gz, nvec,
               C / 8, C, partial);
  else
    return SCL_E_KIND;
  SCL_LAUNCH("colsum_kernel", colsum_kernel, dim3((C + 31) / 32), dim3(1024), 0,
             st, (const float*)partial, nb, C, bias_grad);
  return scl_launch_status();
}

extern "C" int scl_vgg_pool_fwd(const void* z, int dtype, const float* bias, int B, int H, int W,
                                int C, void* a, void* stream) {
  if (!z || !bias || !a) return SCL_E_NULL;
  if (B < 1 || H < 2 || W < 2 || !channels_ok(C) || !aligned16(z) || !aligned16(a))
    return SCL_E_SHAPE;
  const int Ho = H / 2, Wo = W / 2;
  const int64_t nvec = (int64_t)B * Ho * Wo * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SCL_DT_F32)
    SCL_LAUNCH("pool_fwd_kernel", pool_fwd_kernel<float>, dim3(blocks_for(nvec)), dim3(kThreads), 0,
               st, (const float*)z, bias, H, W, Ho, Wo, C / 8, nvec, (float*)a);
  else if (dtype == SCL_DT_BF16)
    SCL_LAUNCH("pool_fwd_kernel", pool_fwd_kernel<unsigned short>, dim3(blocks_for(nvec)),
               dim3(kThreads), 0, st, (const unsigned short*)z, bias, H, W, Ho, Wo, C / 8, nvec,
               (unsigned short*)a);
  else
    return SCL_E_KIND;
  return scl_launch_status();
}

extern "C" int scl_vgg_pool_bwd(const void* g, const void* a, const void* z, int dtype, int B,
                                int H, int W, int C, void* gz, float* bias_grad, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!g || !a || !z || !gz || !bias_grad || !workspace) return SCL_E_NULL;
  if (B < 1 || H < 2 || W < 2 || !channels_ok(C) || !aligned16(g) || !aligned16(a) ||
      !aligned16(z) || !aligned16(gz))
    return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_vgg_workspace_bytes(C))
    return SCL_E_WORKSPACE;
  const int Ho = H / 2, Wo = W / 2;
  const int64_t nvec = (int64_t)B * Ho * Wo * (C / 8);
  const int nb = blocks_for(nvec);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const bool border = (H & 1) || (W & 1);
  if (dtype == SCL_DT_F32) {
    if (border)
      SCL_LAUNCH("pool_bwd_border_kernel", pool_bwd_border_kernel<float>, dim3(kMaxBlocks),
                 dim3(kThreads), 0, st, (float*)gz, B, H, W, Ho, Wo, C);
    SCL_LAUNCH("pool_bwd_kernel", pool_bwd_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
               (const float*)g, (const float*)a, (const float*)z, H, W, Ho, Wo, C / 8, C, nvec,
               (float*)gz, partial);
  } else if (dtype == SCL_DT_BF16) {
    if (border)
      SCL_LAUNCH("pool_bwd_border_kernel", pool_bwd_border_kernel<unsigned short>, dim3(kMaxBlocks),
                 dim3(kThreads), 0, st, (unsigned short*)gz, B, H, W, Ho, Wo, C);
    SCL_LAUNCH("pool_bwd_kernel", pool_bwd_kernel<unsigned short>, dim3(nb), dim3(kThreads), 0, st,
               (const unsigned short*)g, (const unsigned short*)a, (const unsigned short*)z, H, W,
               Ho, Wo, C / 8, C, nvec, (unsigned short*)gz, partial);
  } else {
    return SCL_E_KIND;
  }
  SCL_LAUNCH("colsum_kernel", colsum_kernel, dim3((C + 31) / 32), dim3(1024), 0,
             st, (const float*)partial, nb, C, bias_grad);
  return scl_launch_status();
}

extern "C" int scl_vgg_pool_bwd_idx(const void* g, const void* a, const void* idx, int dtype,
                                    int B, int H, int W, int C, void* gz, float* bias_grad,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !idx || !gz || !bias_grad || !workspace) return SCL_E_NULL;
  if (B < 1 || H < 2 || W < 2 || !channels_ok(C) || !aligned16(g) || (a && !aligned16(a)) ||
      (((uintptr_t)idx) & 7u) || !aligned16(gz))
    return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_vgg_workspace_bytes(C))
    return SCL_E_WORKSPACE;
  const int Ho = H / 2, Wo = W / 2;
  const int64_t nvec = (int64_t)B * Ho * Wo * (C / 8);
  const int nb = blocks_for(nvec);
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const bool border = (H & 1) || (W & 1);
  if (dtype == SCL_DT_F32) {
    if (border)
      SCL_LAUNCH("pool_bwd_border_kernel", pool_bwd_border_kernel<float>, dim3(kMaxBlocks),
                 dim3(kThreads), 0, st, (float*)gz, B, H, W, Ho, Wo, C);
    SCL_LAUNCH("pool_bwd_idx_kernel", pool_bwd_idx_kernel<float>, dim3(nb), dim3(kThreads), 0, st,
               (const float*)g, (const float*)a, (const unsigned char*)idx, H, W, Ho, Wo, C / 8, C,
               nvec, (float*)gz, partial);
  } else if (dtype == SCL_DT_BF16) {
    if (border)
      SCL_LAUNCH("pool_bwd_border_kernel", pool_bwd_border_kernel<unsigned short>, dim3(kMaxBlocks),
                 dim3(kThreads), 0, st, (unsigned short*)gz, B, H, W, Ho, Wo, C);
    SCL_LAUNCH("pool_bwd_idx_kernel", pool_bwd_idx_kernel<unsigned short>, dim3(nb), dim3(kThreads),
               0, st, (const unsigned short*)g, (const unsigned short*)a,
               (const unsigned char*)idx, H, W, Ho, Wo, C / 8, C, nvec, (unsigned short*)gz,
               partial);
  } else {
    return SCL_E_KIND;
  }
  SCL_LAUNCH("colsum_kernel", colsum_kernel, dim3((C + 31) / 32), dim3(1024), 0,
             st, (const float*)partial, nb, C, bias_grad);
  return scl_launch_status();
}
