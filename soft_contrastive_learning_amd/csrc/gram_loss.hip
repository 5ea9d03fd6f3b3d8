// Gram-matrix loss family on gfx950: wms_loss / ms_loss forward + backward and
// _pairwise_squared_distances.
//
// Reference semantics: model/losses.py:5-60 (wms_loss), :76-122 (ms_loss),
// :656-661 (_pairwise_squared_distances).  The reference runs ~40 TF ops on B×B
// temporaries plus one [B,E]x[E,B] matmul; here the path is
//   1. gram_partial_kernel   split-K raw Gram  G = E E^T  on the f32 MFMA
//                            (v_mfma_f32_32x32x2_f32: exact f32, SURVEY H2 forbids
//                            a low-precision Gram), upper-triangular 32x32 tiles,
//                            one [32x32] slab per (K-split, tile pair);
//   2. gram_rows_kernel      one workgroup per row: slab reduction, row norms from
//                            the Gram diagonal, masks, MS mining, row loss and
//                            d loss / d S row;
//   3. gram_coef_kernel      loss mean + the matrix M with d loss / d E = M E
//                            (folds (G+G^T) and the l2_normalize Jacobian);
//   4. gram_bwd_kernel       grad_E[rows] = g * M[rows,:] E   on the f32 MFMA.
// All reductions are fixed-order (no float atomics): results are bitwise
// reproducible run to run.
#include <mutex>

#include "scl_common.h"

namespace {

constexpr int kTile = 32;
constexpr int kMaxB = 1024;  // rows kernel LDS and the backward's M tile are sized for this
constexpr int kMaxSqdist = 4096;  // _pairwise_squared_distances only needs the slab pass

struct GramPlan {
  int tiles;    // ceil(B / 32)
  int npairs;   // tiles * (tiles + 1) / 2
  int splits;   // K-splits
  int kchunk;   // floats of E per split (multiple of 32)
};

inline GramPlan make_plan(int B, int E) {
  GramPlan p;
  p.tiles = (B + kTile - 1) / kTile;
  p.npairs = p.tiles * (p.tiles + 1) / 2;
  // K-splits: enough workgroups to cover the chip, but every split costs the row kernel one
  // more slab entry per pair (B=24: 64 splits x 4 waves already stream the 3 MB in ~4 us)
  int s = 1024 / p.npairs;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  int kc = (E + s - 1) / s;
  kc = (kc + 31) / 32 * 32;
  p.kchunk = kc;
  p.splits = (E + kc - 1) / kc;
  return p;
}

// pair index -> (ti <= tj), row-major over the upper triangle
__device__ __forceinline__ void decode_pair(int pair, int tiles, int& ti, int& tj) {
  int i = 0, rem = pair;
  while (rem >= tiles - i) {
    rem -= tiles - i;
    ++i;
  }
  ti = i;
  tj = i + rem;
}
__device__ __forceinline__ int pair_index(int ti, int tj, int tiles) {
  // ti <= tj
  return ti * tiles - ti * (ti - 1) / 2 + (tj - ti);
}

__device__ __forceinline__ f32x4 load4_guard(const float* row, int e, int E, bool row_ok,
                                             bool vec_ok) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!row_ok) return v;
  if (vec_ok && e + 4 <= E) return *reinterpret_cast<const f32x4*>(row + e);
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (e + c < E) v[c] = row[e + c];
  return v;
}

// tf.tanh of a float32 CPU tensor (Eigen's generic_fast_tanh_float, restated op for op in
// oracle/losses_np.py::eigen_fast_tanh_f32): clamp to [-9, 9], rational approximation, every
// operation individually rounded.  1 - tanh(d / d_beta) decides pair membership through
// mask_pos > 0, so the last bit near saturation matters; ocml's tanhf differs there.
__device__ __forceinline__ float eigen_fast_tanh(float a) {
  // HIP's __fmul_rn / __fadd_rn are plain operators, so only this pragma keeps hipcc from
  // contracting the Horner steps into FMAs (Eigen's pmadd is mul-then-add on non-FMA builds)
#pragma clang fp contract(off)
  const float x = fmaxf(-9.0f, fminf(9.0f, a));
  const float x2 = x * x;
  float p = x2 * -2.76076847742355e-16f + 2.00018790482477e-13f;
  p = x2 * p + -8.60467152213735e-11f;
  p = x2 * p + 5.12229709037114e-08f;
  p = x2 * p + 1.48572235717979e-05f;
  p = x2 * p + 6.37261928875436e-04f;
  p = x2 * p + 4.89352455891786e-03f;
  p = x * p;
  float q = x2 * 1.19825839466702e-06f + 1.18534705686654e-04f;
  q = x2 * q + 2.26843463243900e-03f;
  q = x2 * q + 4.89352518554385e-03f;
  return __fdiv_rn(p, q);
}

// grid (splits, npairs, batch); block 256.  Slab layout: [batch][split][pair][32*32] row-major.
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ emb,
                                                           int64_t ld, int64_t batch_stride,
                                                           int B, int E, int tiles, int kchunk,
                                                           int vec_ok, float* __restrict__ slabs) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  int ti, tj;
  decode_pair(blockIdx.y, tiles, ti, tj);
  const float* base = emb + (int64_t)blockIdx.z * batch_stride;
  const int rowi = ti * kTile + r, rowj = tj * kTile + r;
  const bool oki = rowi < B, okj = rowj < B;
  const float* pi = base + (int64_t)rowi * ld;
  const float* pj = base + (int64_t)rowj * ld;
  const int kw = kchunk >> 2;  // per-wave span, multiple of 8
  const int e_begin = blockIdx.x * kchunk + wid * kw;
  int e_end = e_begin + kw;
  if (e_end > E) e_end = E;
  f32x16 acc = zero16();
  const bool same = ti == tj;
  // Contraction order inside one 8-float group is permuted (half h takes floats
  // 4h..4h+3): A and B use the same permutation, so the product is unchanged.
#pragma unroll 4
  for (int e = e_begin; e < e_end; e += 8) {
    f32x4 a = load4_guard(pi, e + 4 * h, E, oki, vec_ok);
    f32x4 b = same ? a : load4_guard(pj, e + 4 * h, E, okj, vec_ok);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = mfma32(a[c], b[c], acc);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wid][q][lane] = acc[q];
  __syncthreads();
  float* slab = slabs + (((int64_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) *
                            (kTile * kTile);
  for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
    const int q = idx >> 6, l = idx & 63;
    const float v = (red[0][q][l] + red[1][q][l]) + (red[2][q][l] + red[3][q][l]);
    slab[acc_row(q, l >> 5) * kTile + (l & 31)] = v;
  }
}

// fixed-order sum over the K-split slabs of Gram entry (i, j)
__device__ __forceinline__ float slab_entry(const float* slabs, int splits, int npairs, int tiles,
                                            int i, int j, int s_begin, int s_step) {
  int ti = i / kTile, tj = j / kTile, ri = i % kTile, rj = j % kTile;
  if (ti > tj) {
    int t = ti; ti = tj; tj = t;
    t = ri; ri = rj; rj = t;
  }
  const float* p = slabs + (int64_t)pair_index(ti, tj, tiles) * (kTile * kTile) + ri * kTile + rj;
  float acc = 0.f;
  for (int s = s_begin; s < splits; s += s_step)
    acc += p[(int64_t)s * npairs * (kTile * kTile)];
  return acc;
}

struct LossParams {
  int mask_kind, dist_rank3, ms_mining, sum_kind;
  float d_alpha, d_beta, alpha, beta, lamb, eps;
};

constexpr int kRowThreads = 512;

// floats of scratch at the head of gram_rows_kernel's LDS
__host__ __device__ inline int rows_part_floats(int B) {
  const int Bp = (B + 63) / 64 * 64;
  const int a = 2 * (Bp > kRowThreads ? Bp : kRowThreads), b = 4 * Bp;
  return a > b ? a : b;
}

// One workgroup per (reduction) row i.  Dynamic LDS: part[rows_part_floats] | gdiag[Bp] | grow[Bp]
// | S[Bp] | mp[Bp] | mn[Bp]  with Bp = B rounded up to 64.
__global__ __launch_bounds__(kRowThreads) void gram_rows_kernel(
    const float* __restrict__ slabs, int splits, int npairs, int tiles, int B,
    const float* __restrict__ distances, const int64_t* __restrict__ labels, LossParams lp,
    float* __restrict__ gn_out, float* __restrict__ gc_out, float* __restrict__ rn_out,
    float* __restrict__ rowloss_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float scratch[32];
  const int Bp = (B + 63) / 64 * 64;
  const int PN = rows_part_floats(B);    // >= 2 * max(kRowThreads, Bp) and >= 4 * Bp
  float* part = lds;                     // [PN] scratch: slab partial sums, later pair terms
  float* gdiag = part + PN;              // raw G[j,j]
  float* grow = gdiag + Bp;              // raw G[i,j], later normalised Gn[i,j]
  float* sS = grow + Bp;
  float* sMp = sS + Bp;
  float* sMn = sMp + Bp;
  const int i = blockIdx.x;

  // --- slab reduction.  All 512 threads take part: thread (g, jj) sums the residue class g
  //     (mod G) of the K-splits for column jj — row entry and diagonal entry in the same loop
  //     so their loads overlap — and the G partial sums are then added in fixed order.
  {
    const int Bq = B < kRowThreads ? B : kRowThreads;   // columns handled per pass
    const int G = kRowThreads / Bq;                     // split classes (1 when B >= 512)
    const int g = threadIdx.x / Bq, jj = threadIdx.x % Bq;
    float* pdiag = part + PN / 2;
    const int64_t stride = (int64_t)npairs * (kTile * kTile);
    if (g < G) {
      for (int j = jj; j < B; j += Bq) {                // one j per thread unless B > 512
        int ti = i / kTile, tj = j / kTile, ri = i % kTile, rj = j % kTile;
        if (ti > tj) {
          int t = ti; ti = tj; tj = t;
          t = ri; ri = rj; rj = t;
        }
        const float* pr =
            slabs + (int64_t)pair_index(ti, tj, tiles) * (kTile * kTile) + ri * kTile + rj;
        const int td = j / kTile, rd = j % kTile;
        const float* pd =
            slabs + (int64_t)pair_index(td, td, tiles) * (kTile * kTile) + rd * kTile + rd;
        float ar = 0.f, ad = 0.f;
#pragma unroll 4
        for (int s = g; s < splits; s += G) {
          ar += pr[s * stride];
          ad += pd[s * stride];
        }
        // slot: [g][j] when G > 1 (then j < Bq), [j] when G == 1
        part[g * Bq + j] = ar;
        pdiag[g * Bq + j] = ad;
      }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < B; j += kRowThreads) {
      float ar = 0.f, ad = 0.f;
      if (G == 1) {
        ar = part[j];
        ad = pdiag[j];
      } else {
        for (int gg = 0; gg < G; ++gg) {
          ar += part[gg * Bq + j];
          ad += pdiag[gg * Bq + j];
        }
      }
      grow[j] = ar;
      gdiag[j] = ad;
    }
    __syncthreads();
  }

  // --- tf.nn.l2_normalize folded into the Gram: Gn = G * rn_i * rn_j,
  //     rn = rsqrt(max(sum x^2, 1e-12))  (model/losses.py:7,82)
  const float rni = 1.0f / sqrtf(fmaxf(gdiag[i], 1e-12f));
  const int64_t labi = labels ? labels[i] : 0;
  float vmaxN = -INFINITY, vmaxP = -INFINITY;
  for (int j = threadIdx.x; j < B; j += kRowThreads) {
    const float rnj = 1.0f / sqrtf(fmaxf(gdiag[j], 1e-12f));
    const float gn = grow[j] * rni * rnj;
    const float s = fmaxf(gn, 0.f);
    float mp, mn;
    if (lp.mask_kind == SCL_MASK_LABELS) {
      const bool adj = labels[j] == labi;
      mp = adj ? 1.f : 0.f;
      mn = adj ? 0.f : 1.f;
    } else {
      const float d = lp.dist_rank3 ? distances[(int64_t)j * B + i] : distances[(int64_t)i * B + j];
      if (lp.mask_kind == SCL_MASK_WMS_LIN) {
        mp = d < lp.d_beta ? 1.0f - d / lp.d_beta : 0.f;
        mn = d < lp.d_beta ? d / lp.d_beta : 1.f;
      } else if (lp.mask_kind == SCL_MASK_WMS_TANH) {
        const float t = eigen_fast_tanh(__fdiv_rn(d, lp.d_beta));
        mp = 1.0f - t;
        mn = t;
      } else {
        mp = 1.0f / (1.0f + expf(lp.d_alpha * (d - lp.d_beta)));
        mn = 1.0f / (1.0f + expf(lp.d_alpha * (lp.d_beta - d)));
      }
    }
    if (j == i) mp -= 1.0f;  // mask_pos - eye (model/losses.py:22,91)
    grow[j] = gn;
    sS[j] = s;
    sMp[j] = mp;
    sMn[j] = mn;
    vmaxN = fmaxf(vmaxN, s * mn);
    vmaxP = fmaxf(vmaxP, s * mp);
  }
  float max_val = 0.f, min_val = 0.f;
  if (lp.ms_mining) {
    max_val = block_reduce<1>(vmaxN, scratch);
    const float tmp = block_reduce<1>(vmaxP, scratch);
    float vmin = INFINITY;
    for (int j = threadIdx.x; j < B; j += kRowThreads) vmin = fminf(vmin, (sS[j] - tmp) * sMp[j]);
    min_val = block_reduce<2>(vmin, scratch) + tmp;
  }
  __syncthreads();

  // --- selected pair terms
  float ps = 0.f, ns = 0.f;
  for (int j = threadIdx.x; j < B; j += kRowThreads) {
    const float s = sS[j], mp = sMp[j], mn = sMn[j];
    const float P = s * mp, Nm = s * mn;
    float kp = mp, kn = mn;
    if (lp.ms_mining) {
      kp = P < max_val + lp.eps ? mp : 0.f;
      kn = Nm > min_val - lp.eps ? mn : 0.f;
    }
    float pe, ne;
    if (lp.sum_kind == SCL_SUM_PLAIN) {
      pe = kp > 0.f ? P : 0.f;
      ne = kn > 0.f ? Nm : 0.f;
    } else {
      pe = kp > 0.f ? expf(-lp.alpha * (P - lp.lamb)) : 0.f;
      ne = kn > 0.f ? expf(lp.beta * (Nm - lp.lamb)) : 0.f;
    }
    // re-use sS / part as per-pair term storage for the gradient pass
    part[j] = pe;
    part[Bp + j] = ne;
    part[2 * Bp + j] = kp > 0.f ? 1.f : 0.f;
    part[3 * Bp + j] = kn > 0.f ? 1.f : 0.f;
    ps += pe;
    ns += ne;
  }
  ps = block_reduce<0>(ps, scratch);
  ns = block_reduce<0>(ns, scratch);
  float rowloss;
  if (lp.sum_kind == SCL_SUM_PLAIN)
    rowloss = ns - ps;
  else
    rowloss = logf(1.0f + ps) / lp.alpha + logf(1.0f + ns) / lp.beta;
  if (threadIdx.x == 0) {
    rowloss_out[i] = rowloss;
    rn_out[i] = rni;
  }
  // --- d loss / d sim_mat row (masks are constants: they only enter through
  //     non-differentiable where-conditions; tf.maximum passes grad where x >= 0)
  const float invB = 1.0f / (float)B;
  for (int j = threadIdx.x; j < B; j += kRowThreads) {
    const float gn = grow[j];
    float g;
    if (lp.sum_kind == SCL_SUM_PLAIN)
      g = part[3 * Bp + j] * sMn[j] - part[2 * Bp + j] * sMp[j];
    else
      g = part[Bp + j] / (1.0f + ns) * sMn[j] - part[j] / (1.0f + ps) * sMp[j];
    g = gn >= 0.f ? g * invB : 0.f;
    gn_out[(int64_t)i * B + j] = gn;
    gc_out[(int64_t)i * B + j] = g;
  }
}

// One workgroup per row i: M[i,:] and, in block 0, the loss mean.
__global__ __launch_bounds__(256) void gram_coef_kernel(const float* __restrict__ gn,
                                                        const float* __restrict__ gc,
                                                        const float* __restrict__ rn,
                                                        const float* __restrict__ rowloss, int B,
                                                        float* __restrict__ coef,
                                                        float* __restrict__ loss_out) {
  __shared__ float scratch[32];
  const int i = blockIdx.x;
  if (i == 0) {
    float a = 0.f;
    for (int j = threadIdx.x; j < B; j += 256) a += rowloss[j];
    a = block_reduce<0>(a, scratch);
    if (threadIdx.x == 0) *loss_out = a / (float)B;
  }
  if (!coef) return;
  float c = 0.f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float gs = gc[(int64_t)i * B + j] + gc[(int64_t)j * B + i];
    c += gs * gn[(int64_t)i * B + j];
  }
  c = block_reduce<0>(c, scratch);
  const float rni = rn[i];
  // rn == 1e6 means the row norm was clamped (sum x^2 < 1e-12): l2_normalize is
  // then a plain scale and has no projection term.
  const bool clamped = rni >= 1.0e6f;
  for (int j = threadIdx.x; j < B; j += 256) {
    const float gs = gc[(int64_t)i * B + j] + gc[(int64_t)j * B + i];
    float m = rni * rn[j] * gs;
    if (j == i && !clamped) m -= rni * rni * c;
    coef[(int64_t)i * B + j] = m;
  }
}

// grad[r, e] = g * sum_j M[row_begin + r, j] * emb[j, e].
// grid (ceil(E/512), row tiles); block 256.  The workgroup's [32 x B] slice of M sits in LDS
// (odd row stride: conflict-free ds_read_b32); every wave owns 128 columns as 4 accumulator
// tiles whose columns are interleaved (tile t holds columns e0 + 4 i + t), so one 16-byte
// load per lane per contraction step feeds 4 MFMAs and a half-wave reads 512 contiguous
// bytes of the embedding row.
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ emb, int64_t ld,
                                                       int B, int E, const float* __restrict__ coef,
                                                       const float* __restrict__ grad_loss,
                                                       int row_begin, int row_count, int vec_ok,
                                                       float* __restrict__ grad, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) float mt[];   // [32][B | 1]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ldm = B | 1;
  for (int idx = threadIdx.x; idx < kTile * B; idx += 256) {
    const int rr = idx / B, j = idx % B;
    const int lr = blockIdx.y * kTile + rr;
    mt[rr * ldm + j] = lr < row_count ? coef[(int64_t)(row_begin + lr) * B + j] : 0.f;
  }
  __syncthreads();
  const int e0 = (blockIdx.x * 4 + wid) * 128;
  if (e0 >= E) return;
  const int e = e0 + 4 * r;                 // this lane's 4 columns (one per tile)
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = zero16();
  const float* ma = mt + r * ldm;
#pragma unroll 4
  for (int j0 = 0; j0 < B; j0 += 2) {
    const int j = j0 + h;
    const bool jok = j < B;
    const float a = jok ? ma[j] : 0.f;
    const f32x4 b = load4_guard(emb + (int64_t)(jok ? j : 0) * ld, e, E, jok, vec_ok);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma32(a, b[t], acc[t]);
  }
  const float g = grad_loss ? *grad_loss : 1.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int orow = blockIdx.y * kTile + acc_row(q, h);
    if (orow >= row_count) continue;
    float* dst = grad + (int64_t)orow * ldg + e;
    if (vec_ok && e + 4 <= E) {
      *reinterpret_cast<f32x4*>(dst) =
          f32x4{g * acc[0][q], g * acc[1][q], g * acc[2][q], g * acc[3][q]};
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (e + t < E) dst[t] = g * acc[t][q];
    }
  }
}

// out[t,i,j] = r_i - 2 G_ij + r_j  from the slabs of batch t.  grid (S rows, T); block 256.
__global__ __launch_bounds__(256) void sqdist_finish_kernel(const float* __restrict__ slabs,
                                                            int splits, int npairs, int tiles,
                                                            int S, float* __restrict__ out) {
  const int i = blockIdx.x, t = blockIdx.y;
  const float* sl = slabs + (int64_t)t * splits * npairs * (kTile * kTile);
  const float ri = slab_entry(sl, splits, npairs, tiles, i, i, 0, 1);
  for (int j = threadIdx.x; j < S; j += 256) {
    const float rj = slab_entry(sl, splits, npairs, tiles, j, j, 0, 1);
    const float g = slab_entry(sl, splits, npairs, tiles, i, j, 0, 1);
    out[((int64_t)t * S + i) * S + j] = ri - 2.0f * g + rj;
  }
}

struct GramWs {
  float *slabs, *gn, *gc, *rn, *rowloss;
  size_t total;
};

inline GramWs carve(void* ws, int B, const GramPlan& p, int batch) {
  GramWs w;
  char* c = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    float* ptr = (float*)(c + off);
    off += scl_round256(bytes);
    return ptr;
  };
  w.slabs = take((size_t)batch * p.splits * p.npairs * kTile * kTile * sizeof(float));
  w.gn = take((size_t)B * B * sizeof(float));
  w.gc = take((size_t)B * B * sizeof(float));
  w.rn = take((size_t)B * sizeof(float));
  w.rowloss = take((size_t)B * sizeof(float));
  w.total = off;
  return w;
}


}  // namespace

extern "C" size_t scl_gram_loss_workspace_bytes(int B, int E) {
  if (B < 1 || E < 1 || B > kMaxB) return 0;
  return carve(nullptr, B, make_plan(B, E), 1).total;
}

extern "C" int scl_gram_loss_fwd(const float* emb, int64_t ld_emb, int B, int E, int mask_kind,
                                 const float* distances, int dist_rank3, float d_alpha,
                                 float d_beta, const int64_t* labels, float alpha, float beta,
                                 float lamb, float eps, int ms_mining, int sum_kind,
                                 float* loss_out, float* coef, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!emb || !loss_out || !workspace) return SCL_E_NULL;
  if (B < 1 || E < 1 || B > kMaxB || ld_emb < E) return SCL_E_SHAPE;
  if (mask_kind < SCL_MASK_WMS_EXP || mask_kind > SCL_MASK_LABELS) return SCL_E_KIND;
  if (sum_kind != SCL_SUM_MS && sum_kind != SCL_SUM_PLAIN) return SCL_E_KIND;
  if (mask_kind == SCL_MASK_LABELS ? !labels : !distances) return SCL_E_NULL;
  const GramPlan p = make_plan(B, E);
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  GramWs w = carve(workspace, B, p, 1);
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int vec_ok = (ld_emb % 4 == 0) && ((uintptr_t)emb % 16 == 0);
  SCL_LAUNCH("gram_partial_kernel", gram_partial_kernel, dim3(p.splits, p.npairs, 1), dim3(256), 0, st, emb,
                     ld_emb, (int64_t)0, B, E, p.tiles, p.kchunk, vec_ok, w.slabs);
  LossParams lp;
  lp.mask_kind = mask_kind;
  lp.dist_rank3 = dist_rank3 ? 1 : 0;
  lp.ms_mining = ms_mining ? 1 : 0;
  lp.sum_kind = sum_kind;
  lp.d_alpha = d_alpha;
  lp.d_beta = d_beta;
  lp.alpha = alpha;
  lp.beta = beta;
  lp.lamb = lamb;
  lp.eps = eps;
  const int Bp = (B + 63) / 64 * 64;
  const size_t lds_bytes = ((size_t)rows_part_floats(B) + 5 * (size_t)Bp) * sizeof(float);
  SCL_LAUNCH("gram_rows_kernel", gram_rows_kernel, dim3(B), dim3(kRowThreads), lds_bytes, st, w.slabs,
                     p.splits, p.npairs, p.tiles, B, distances, labels, lp, w.gn, w.gc, w.rn,
                     w.rowloss);
  SCL_LAUNCH("gram_coef_kernel", gram_coef_kernel, dim3(coef ? B : 1), dim3(256), 0, st, w.gn, w.gc, w.rn,
                     w.rowloss, B, coef, loss_out);
  return scl_launch_status();
}

extern "C" int scl_gram_loss_bwd(const float* emb, int64_t ld_emb, int B, int E, const float* coef,
                                 const float* grad_loss, int row_begin, int row_count,
                                 float* grad_emb, int64_t ld_grad, void* stream) {
  if (!emb || !coef || !grad_emb) return SCL_E_NULL;
  if (B < 1 || E < 1 || ld_emb < E || ld_grad < E) return SCL_E_SHAPE;
  if (row_begin < 0 || row_count < 1 || row_begin + row_count > B || B > kMaxB) return SCL_E_SHAPE;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kTile * (kMaxB | 1) * sizeof(float)));
  });
  const int vec_ok = (ld_emb % 4 == 0) && (ld_grad % 4 == 0) && ((uintptr_t)emb % 16 == 0) &&
                     ((uintptr_t)grad_emb % 16 == 0);
  const size_t lds = (size_t)kTile * (B | 1) * sizeof(float);
  dim3 grid((E + 511) / 512, (row_count + kTile - 1) / kTile);
  SCL_LAUNCH("gram_bwd_kernel", gram_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, emb,
             ld_emb, B, E, coef, grad_loss, row_begin, row_count, vec_ok, grad_emb, ld_grad);
  return scl_launch_status();
}

extern "C" size_t scl_pairwise_sqdist_workspace_bytes(int T, int S, int E) {
  if (T < 1 || S < 1 || E < 1 || S > kMaxSqdist) return 0;
  const GramPlan p = make_plan(S, E);
  return scl_round256((size_t)T * p.splits * p.npairs * kTile * kTile * sizeof(float));
}

extern "C" int scl_pairwise_sqdist(const float* feats, int T, int S, int E, float* out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!feats || !out || !workspace) return SCL_E_NULL;
  if (T < 1 || S < 1 || E < 1 || S > kMaxSqdist || T > 65535) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_pairwise_sqdist_workspace_bytes(T, S, E))
    return SCL_E_WORKSPACE;
  const GramPlan p = make_plan(S, E);
  hipStream_t st = (hipStream_t)stream;
  const int vec_ok = (E % 4 == 0) && ((uintptr_t)feats % 16 == 0);
  SCL_LAUNCH("gram_partial_kernel", gram_partial_kernel, dim3(p.splits, p.npairs, T), dim3(256), 0, st, feats,
                     (int64_t)E, (int64_t)S * E, S, E, p.tiles, p.kchunk, vec_ok,
                     (float*)workspace);
  SCL_LAUNCH("sqdist_finish_kernel", sqdist_finish_kernel, dim3(S, T), dim3(256), 0, st, (const float*)workspace,
                     p.splits, p.npairs, p.tiles, S, out);
  return scl_launch_status();
}
