// bf16 plane images of the NetVLAD assignment weights W [512][64] (float32), shared by
// netvlad.hip (which reads them) and conv_pack.hip (which writes them in the same launch as the
// packed convolution weights: the weights change once per step, train/train.py:877-879).
//
// A float32 operand o is carried as bf16 planes o = o1 + o2 (+ o3): every product of two bf16
// values is exact in float32, two planes leave 2^-17 |o|.
#pragma once
#include "scl_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = h1 + h2 + h3 exactly (up to the float32 subnormal range): three bf16 roundings.
__device__ __forceinline__ void split3_bf16(float x, unsigned short& h1, unsigned short& h2,
                                            unsigned short& h3) {
  h1 = f32_to_bf16(x);
  const float r1 = x - bf16_to_f32(h1);
  h2 = f32_to_bf16(r1);
  h3 = f32_to_bf16(r1 - bf16_to_f32(h2));
}

// float32 x 8 -> packed bf16 high and low halves (v = hi + lo + O(2^-17 |v|))
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

constexpr int VP_NPL = 2;                               // planes of the forward image
constexpr int VP_FWD_ELEMS = VP_NPL * SCL_VLAD_D * SCL_VLAD_K;   // bf16 elements (131,072 B)
constexpr int VP_DX_ELEMS = 2 * SCL_VLAD_D * SCL_VLAD_K;         // bf16 elements (131,072 B)
constexpr size_t VP_BYTES = (size_t)(VP_FWD_ELEMS + VP_DX_ELEMS) * 2;
constexpr int VP_WAVES = 64 + 32;                       // wave-sized jobs that write both images

// Wave job `job` (0 .. VP_WAVES - 1) of the two register images of W:
//   jobs 0..63   forward image (vlad_fwd_kernel; was vlad_split_w_kernel): 16-byte unit
//                ((w * 16 + s) * VP_NPL + plane) * 64 + lane = W_plane[ch 32 s + 8 g + e][cluster
//                16 w + i], e = 0..7, lane = 16 g + i; job = 16 w + s;
//   jobs 64..95  grad_x image (vlad_dx_kernel; was written by image 0's bwd_du workgroups): unit
//                (((wv * 8 + nt) * 2 + s) * 2 + plane) * 64 + lane = W_plane[ch 128 wv + 16 nt + i]
//                [cluster 32 s + 8 g + e]; job - 64 = (wv * 8 + nt), both s.
__device__ __forceinline__ void vlad_planes_wave(const float* __restrict__ w, unsigned short* __restrict__ fwd_img,
                                                 unsigned short* __restrict__ dx_img, int job, int lane) {
  constexpr int K = SCL_VLAD_K;
  const int i = lane & 15, g = lane >> 4;
  if (job < 64) {
    const int wv = job >> 4, s = job & 15;
    unsigned short h[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) split3_bf16(w[(32 * s + 8 * g + e) * K + 16 * wv + i], h[0][e], h[1][e], h[2][e]);
#pragma unroll
    for (int pl = 0; pl < VP_NPL; ++pl) {
      uint4 v;
      v.x = (unsigned)h[pl][0] | ((unsigned)h[pl][1] << 16);
      v.y = (unsigned)h[pl][2] | ((unsigned)h[pl][3] << 16);
      v.z = (unsigned)h[pl][4] | ((unsigned)h[pl][5] << 16);
      v.w = (unsigned)h[pl][6] | ((unsigned)h[pl][7] << 16);
      reinterpret_cast<uint4*>(fwd_img)[(((wv * 16 + s) * VP_NPL + pl) * 64) + lane] = v;
    }
    return;
  }
  const int t = job - 64;                                   // wv * 8 + nt
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* wr = w + (int64_t)(16 * t + i) * K + 32 * s + 8 * g;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
    u32x4 hi, lo;
    split2x8(w0, w1, hi, lo);
    u32x4* img = reinterpret_cast<u32x4*>(dx_img) + ((t * 2 + s) * 2) * 64 + lane;
    img[0] = hi;
    img[64] = lo;
  }
}
