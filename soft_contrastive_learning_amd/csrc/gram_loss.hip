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

// grad[r, e] = g * sum_j M[row_begin + r, j] emb[j, e] for 32 < B <= 1024 on aligned rows:
// gram_bwd_kernel without a branch around any load (a guarded load makes hipcc wait for each
// one before issuing the next) and with the work cut so that every SIMD gets the same number
// of waves: a wave owns 32 * TILES columns (TILES interleaved accumulator tiles fed by one
// 4 * TILES-byte load per lane and step), four waves per workgroup.
// grid (E / (128 * TILES), row tiles); block 256; LDS [32][B | 1] floats (the M tile).
template <int TILES>
__global__ __launch_bounds__(256) void gram_bwd_fast_kernel(const float* __restrict__ emb,
                                                            int64_t ld, int B, int E,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ grad_loss,
                                                            int row_begin, int row_count,
                                                            float* __restrict__ grad, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) float mt[];   // [32][B | 1]
  typedef float vec_t __attribute__((ext_vector_type(TILES)));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ldm = B | 1;
  for (int idx = threadIdx.x; idx < kTile * B; idx += 256) {
    const int rr = idx / B, j = idx - rr * B;
    const int lr = blockIdx.y * kTile + rr;
    mt[rr * ldm + j] = lr < row_count ? coef[(int64_t)(row_begin + lr) * B + j] : 0.f;
  }
  __syncthreads();
  const int e = (blockIdx.x * 4 + wid) * 32 * TILES + TILES * r;   // this lane's TILES columns
  f32x16 acc[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) acc[t] = zero16();
  const float* ma = mt + r * ldm + h;
  const float* eb = emb + (int64_t)h * ld + e;
  const int steps = B >> 1;                                       // pairs of contraction rows
  constexpr int U = 8;        // (32 for the narrow cut: 19 -> 32 us, the wave count drops)
  int s0 = 0;
  for (; s0 + U <= steps; s0 += U) {
    vec_t bv[U];
    float av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bv[u] = *reinterpret_cast<const vec_t*>(eb + (int64_t)2 * (s0 + u) * ld);
      av[u] = ma[2 * (s0 + u)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < TILES; ++t) acc[t] = mfma32(av[u], bv[u][t], acc[t]);
  }
  for (; 2 * s0 < B; ++s0) {                                      // tail (and an odd last row)
    const int j = 2 * s0 + h;
    const bool jok = j < B;
    const vec_t bvt = *reinterpret_cast<const vec_t*>(emb + (int64_t)(jok ? j : B - 1) * ld + e);
    const float a = jok ? mt[r * ldm + j] : 0.f;
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = mfma32(a, bvt[t], acc[t]);
  }
  const float g = grad_loss ? *grad_loss : 1.0f;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int orow = blockIdx.y * kTile + acc_row(q, h);
    if (orow >= row_count) continue;
    vec_t o;
#pragma unroll
    for (int t = 0; t < TILES; ++t) o[t] = g * acc[t][q];
    *reinterpret_cast<vec_t*>(grad + (int64_t)orow * ldg + e) = o;
  }
}

// gram_bwd_rows_kernel (round 4): the backward for ONE row tile (<= 32 rows) — what every rank of a
// data-parallel run asks for after the all-gather (its own 24 rows of the 192, parallel.wms_loss_dp).
// That is a 25 MB read of E for 19 MFLOP per KB: a stream, and gram_bwd_fast_kernel<1> ran it with
// 4-byte loads per lane (32-column waves were the only cut that put a wave on every SIMD).  Here
// a workgroup owns 128 columns and its four waves split the CONTRACTION rows: every lane loads 16
// bytes per step (a half-wave reads 512 contiguous bytes of an embedding row), twelve steps in
// flight, and the four partial [32 x 128] tiles meet in LDS in a fixed order.
// grid E / 128, block 256; LDS [32][B | 1] (M tile) + [4][32][132] floats.
constexpr int GBR_LD = 132;
__global__ __launch_bounds__(256) void gram_bwd_rows_kernel(const float* __restrict__ emb, int64_t ld,
                                                            int B, int E, const float* __restrict__ coef,
                                                            const float* __restrict__ grad_loss,
                                                            int row_begin, int row_count,
                                                            float* __restrict__ grad, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) float mt[];   // [32][B | 1] | part[4][32][GBR_LD]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int ldm = B | 1;
  float* part = mt + ((kTile * ldm + 3) & ~3);
  for (int idx = threadIdx.x; idx < kTile * B; idx += 256) {
    const int rr = idx / B, j = idx - rr * B;
    mt[rr * ldm + j] = rr < row_count ? coef[(int64_t)(row_begin + rr) * B + j] : 0.f;
  }
  __syncthreads();
  const int e = blockIdx.x * 128 + 4 * r;                        // this lane's four columns
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = zero16();
  const int steps = (B + 1) >> 1;                                // pairs of contraction rows
  const int per = (steps + 3) >> 2;
  const int s_lo = wid * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
  constexpr int U = 12;
  for (int s0 = s_lo; s0 < s_hi; s0 += U) {
    f32x4 bv[U];
    float av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                                // branch-free: clamped rows, zero weight
      const int j = 2 * (s0 + u) + h;
      const bool ok = s0 + u < s_hi && j < B;
      bv[u] = *reinterpret_cast<const f32x4*>(emb + (int64_t)(ok ? j : B - 1) * ld + e);
      av[u] = ok ? mt[r * ldm + j] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);                           // every load requested before the products
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma32(av[u], bv[u][t], acc[t]);
  }
  // partial tile of this wave: element (row acc_row(q, h), column 4 r + t)
  float* mine = part + wid * kTile * GBR_LD;
#pragma unroll
  for (int q = 0; q < 16; ++q)
    *reinterpret_cast<f32x4*>(mine + acc_row(q, h) * GBR_LD + 4 * r) =
        f32x4{acc[0][q], acc[1][q], acc[2][q], acc[3][q]};
  __syncthreads();
  const float g = grad_loss ? *grad_loss : 1.0f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int unit = it * 256 + threadIdx.x, row = unit >> 5, c4 = unit & 31;
    const float* p0 = part + row * GBR_LD + 4 * c4;
    const f32x4 v = (*reinterpret_cast<const f32x4*>(p0) + *reinterpret_cast<const f32x4*>(p0 + kTile * GBR_LD)) +
                    (*reinterpret_cast<const f32x4*>(p0 + 2 * kTile * GBR_LD) +
                     *reinterpret_cast<const f32x4*>(p0 + 3 * kTile * GBR_LD));
    if (row < row_count)
      *reinterpret_cast<f32x4*>(grad + (int64_t)row * ldg + blockIdx.x * 128 + 4 * c4) = v * g;
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


// =======================================================================================
// Fast path for B <= 256 (the trainer's batches: 24 / 25 rows per GPU, 192 on eight).
//
//   gram16_kernel        split-K raw Gram on v_mfma_f32_16x16x4_f32 (exact f32), 16-row tiles,
//                        upper-triangular tile pairs only.  A workgroup stages its whole
//                        [B x kchunk] slice of E in LDS with ONE burst of 16-byte loads (a
//                        single exposed memory latency) and its waves split the work as
//                        (pair ranges) x (k ranges); fragments are ds_read_b128 from padded
//                        rows, the k index inside a group of 16 permuted identically on both
//                        operands so that one 16-byte read feeds four MFMAs.
//   gram_final32_kernel  B <= 32: ONE workgroup sums the slabs and does everything else —
//                        norms, masks, mining, row losses, d loss / d S, the loss mean and the
//                        matrix M — so the forward is two launches (a second launch costs
//                        less than an in-kernel release / ticket / acquire hand-off).
//   gram_reduce_kernel   32 < B <= 256: slab sums -> full raw Gram matrix;
//   gram_rows_wave_kernel  one WAVE per row (lanes across the columns, shuffle reductions)
//                        instead of one 512-thread workgroup per row.
//   gram_bwd32_kernel    B <= 32: grad = g M E with a wave per 32 columns (1024 waves).
// =======================================================================================
constexpr int kFastB = 256;
constexpr int kG16 = 16;

__device__ __forceinline__ f32x4 mfma16f(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct Gram16Plan {
  int T;        // 16-row tiles
  int P;        // tile pairs (upper triangle)
  int S;        // K-splits = workgroups
  int kchunk;   // columns per workgroup (multiple of 16)
  int KS;       // k-ranges among the 4 waves (4, 2 or 1); pair ranges = 4 / KS
};

inline Gram16Plan make_plan16(int B, int E) {
  Gram16Plan p;
  p.T = (B + 15) / 16;
  p.P = p.T * (p.T + 1) / 2;
  p.KS = p.P <= 3 ? 4 : (p.P <= 10 ? 2 : 1);
  // LDS holds [16 T][kchunk + 4] floats (<= 132 KB, the kernels' dynamic limit: [64][516] floats
  // is the largest) and a thread stages <= 48 float4
  int kc_max = (132 * 1024 / 4) / (16 * p.T) - 4;
  kc_max = kc_max / 64 * 64;
  if (kc_max > 512) kc_max = 512;
  // small batches: fewer, larger splits (the finishing workgroup reads every slab); large ones:
  // one workgroup per CU.  (Round 5 measured 64 splits up to B = 64 for a one-launch forward there:
  // the Gram on 64 CUs takes 11.7 / 18.3 us at B = 48 / 64 against 7.8 / 9.7 on 256 — not kept.)
  int s = B <= 32 ? 64 : 256;
  int ov = scl_variant() / 100000;          // tuning override: splits = ov
  if (ov > 0) s = ov;
  int kc = 64;                                   // power of two: staging indices by shifts
  while (kc < (E + s - 1) / s && 2 * kc <= kc_max) kc *= 2;
  p.kchunk = kc;
  p.S = (E + kc - 1) / kc;
  return p;
}

// pair index (row-major over the upper triangle of T x T tiles) -> (ti, tj)
__device__ __forceinline__ void decode_pair16(int pair, int T, int& ti, int& tj) {
  int i = 0, rem = pair;
  while (rem >= T - i) {
    rem -= T - i;
    ++i;
  }
  ti = i;
  tj = i + rem;
}

// grid S; block 256; dynamic LDS [16 T][kchunk + 4] floats (reused for the cross-wave sums).
// slabs: [S][P][256]: accumulator register j of lane l at [l * 4 + j] = Gram entry
// (row 16 ti + 4 (l >> 4) + j, column 16 tj + (l & 15)).
// FULL: every wave owns exactly PWMAX pairs (no guard in the pair loop).
// write-through (sc1) 16-byte store / L1-bypassing (sc1) load: the hand-off of the fused finish
// (MI355X_MICROARCH.md, "Valid forms": every handed-off byte stored sc1 and drained before ONE lane's
// agent-scope ticket add, every load of them sc1 behind the last ticket holder's add — no release,
// no acquire).
__device__ __forceinline__ void st_sc1_x4(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// (the loads go through the buffer-load builtin with aux = 16 = sc1, so that hipcc tracks their
// destinations: an asm load's destination register is unprotected until the wait, and a copy
// hipcc placed right behind one read stale bits — found as a launch failure)
__device__ __forceinline__ f32x4 ld_sc1_x4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16));
}

// Arguments of the finish that rides in the LAST workgroup of gram16_kernel to arrive (B <= 32,
// round 4): what gram_final32_kernel takes, plus the arrival counter.
struct FinalArgs {
  unsigned* counter;          // one word of the caller's sync block: zero on entry, zero on return
  const float* distances;
  const int64_t* labels;
  LossParams lp;
  float* coef;
  float* loss_out;
};
template <int NT>
__device__ void final32_body(float* lds, const float* slabs, int S, int T, int P, int B,
                             const FinalArgs& fa);
template <int NT>
__device__ void final64_body(float* lds, const float* slabs, int S, int T, int P, int B,
                             const FinalArgs& fa);

// FUSE (gram16_fused_kernel, 1024 threads): the first four waves run the Gram exactly as the
// 256-thread kernel does, the other twelve only keep its barriers company until the finish, which
// (in the last workgroup to arrive) wants sixteen waves: a half-wave per row, all rows at once.
// PERSIST (gram16_persist_kernel, round 6): write-through slab stores and no early return — the
// caller continues with persist_tail.
template <int PWMAX, bool FULL, bool FUSE, bool PERSIST = false>
__device__ __forceinline__ void gram16_body(const float* __restrict__ emb, int64_t ld, int B, int E, int T,
                                            int P, int kchunk, int KS, int vec_ok,
                                            float* __restrict__ slabs, const FinalArgs& fa) {
  extern __shared__ __attribute__((aligned(16))) float g16_lds[];
  // the wave index must be PROVABLY wave-uniform: everything derived from it (pair range, k
  // range) then lives in scalar registers and the pair loop has scalar branches; as a plain
  // threadIdx expression hipcc treats it as divergent and wraps every step in exec masks and
  // accumulator copies
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, gq = lane >> 4;
  const int LD = kchunk + 4;
  const int Bp = 16 * T;
  const int k0 = blockIdx.x * kchunk;
  const int qshift = __ffs(kchunk) - 3;              // log2(float4 per staged row)
  const int qmask = (1 << qshift) - 1;
  const int total_q = Bp << qshift;

  // (FUSE: waves 4..15 only match the barriers of the Gram part: one behind the staging, two more
  // around the cross-wave sums when the k ranges are split.  The barriers of the two branches sit
  // at DIFFERENT program points: valid because s_barrier on this hardware counts arrivals of the
  // workgroup's waves — it does not require them to arrive at the same instruction — and both
  // branches execute the same NUMBER of barriers, 1 or 3, chosen by the uniform KS.)
  const bool worker = !FUSE || wid < 4;                       // (wid is a scalar: a scalar branch)
  if (!worker) {
    __syncthreads();
    if (KS != 1) {
      __syncthreads();
      __syncthreads();
    }
  } else {
  // ---- stage the slice: every load is issued before the first LDS write.  The common case
  //      (aligned rows, slice inside E) has NO branch around a load: a guarded load makes hipcc
  //      wait for each one before issuing the next (16 serial round trips measured here).
  constexpr int RND = 16;
  const bool fast = vec_ok && k0 + kchunk <= E;
  for (int base = 0; base < total_q; base += 256 * RND) {
    f32x4 v[RND];
    if (fast) {
#pragma unroll
      for (int u = 0; u < RND; ++u) {
        const int q = base + u * 256 + threadIdx.x;
        int row = q >> qshift;
        row = row < B ? row : B - 1;                    // padding rows re-read the last row
        v[u] = *reinterpret_cast<const f32x4*>(emb + (int64_t)row * ld + k0 + 4 * (q & qmask));
      }
    } else {
#pragma unroll
      for (int u = 0; u < RND; ++u) {
        const int q = base + u * 256 + threadIdx.x;
        const int row = q >> qshift, c4 = q & qmask;
        const int e = k0 + 4 * c4;
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < total_q && row < B) {
          const float* src = emb + (int64_t)row * ld + e;
          if (vec_ok && e + 4 <= E) {
            v[u] = *reinterpret_cast<const f32x4*>(src);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (e + c < E) v[u][c] = src[c];
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RND; ++u) {
      const int q = base + u * 256 + threadIdx.x;
      const int row = q >> qshift, c4 = q & qmask;
      if (q < total_q)
        *reinterpret_cast<f32x4*>(&g16_lds[row * LD + 4 * c4]) =
            row < B ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();

  // ---- this wave's share: pairs [p_begin, p_end) x 16-column groups [g_begin, g_end)
  const int PS = 4 / KS;
  const int kq = wid % KS, pq = wid / KS;
  const int p_begin = (int)(((long)P * pq) / PS), p_end = (int)(((long)P * (pq + 1)) / PS);
  const int np = p_end - p_begin;
  const int G = kchunk / kG16;
  const int g_begin = (G * kq) / KS, g_end = (G * (kq + 1)) / KS;
  int ti0, tj0;
  decode_pair16(p_begin < P ? p_begin : 0, T, ti0, tj0);

  f32x4 acc[PWMAX];
#pragma unroll
  for (int lp = 0; lp < PWMAX; ++lp) acc[lp] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (group, pair) is one flat stream: the fragments of the NEXT step — the next pair of this
  // group or the first pair of the next group — are read under this step's MFMAs
  const float* col0 = &g16_lds[i * LD + 4 * gq];
  f32x4 a = *reinterpret_cast<const f32x4*>(col0 + kG16 * g_begin + 16 * ti0 * LD);
  f32x4 b = *reinterpret_cast<const f32x4*>(col0 + kG16 * g_begin + 16 * tj0 * LD);
  for (int g = g_begin; g < g_end; ++g) {
    const float* col = col0 + kG16 * g;
    const float* col_next = g + 1 < g_end ? col + kG16 : col;
    int ti = ti0, tj = tj0;
#pragma unroll
    for (int lp = 0; lp < PWMAX; ++lp) {
      if (FULL || lp < np) {
        const bool last = FULL ? lp + 1 == PWMAX : lp + 1 == np;
        int tj_n = tj + 1, ti_n = ti;
        if (tj_n == T) {
          ti_n = ti + 1;
          tj_n = ti_n;
        }
        ti_n = last ? ti0 : ti_n;
        tj_n = last ? tj0 : tj_n;
        const float* cn = last ? col_next : col;
        const f32x4 a_n = *reinterpret_cast<const f32x4*>(cn + 16 * ti_n * LD);
        const f32x4 b_n = *reinterpret_cast<const f32x4*>(cn + 16 * tj_n * LD);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[lp] = mfma16f(a[c], b[c], acc[lp]);
        a = a_n;
        b = b_n;
        ti = ti_n;
        tj = tj_n;
      }
    }
  }

  float* slab = slabs + (int64_t)blockIdx.x * P * 256;
  if (KS == 1) {
#pragma unroll
    for (int lp = 0; lp < PWMAX; ++lp)
      if (FULL || lp < np) {
        if (FUSE || PERSIST)
          st_sc1_x4(slab + (int64_t)(p_begin + lp) * 256 + 4 * lane, acc[lp]);
        else
          *reinterpret_cast<f32x4*>(slab + (int64_t)(p_begin + lp) * 256 + 4 * lane) = acc[lp];
      }
    if (!FUSE && !PERSIST) return;
  } else {
    // k ranges on different waves: fixed-order sum through LDS (the staged slice is dead)
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(g16_lds);          // [KS][P][64]
#pragma unroll
    for (int lp = 0; lp < PWMAX; ++lp)
      if (FULL || lp < np) red[((int64_t)kq * P + p_begin + lp) * 64 + lane] = acc[lp];
    __syncthreads();
    for (int idx = threadIdx.x; idx < P * 64; idx += 256) {
      f32x4 v = red[idx];
      if (KS == 2) {
        v += red[P * 64 + idx];
      } else {
        const f32x4 v1 = red[P * 64 + idx], v2 = red[2 * P * 64 + idx], v3 = red[3 * P * 64 + idx];
        v = (v + v1) + (v2 + v3);
      }
      if (FUSE || PERSIST)
        st_sc1_x4(slab + 4 * (int64_t)idx, v);
      else
        *reinterpret_cast<f32x4*>(slab + 4 * (int64_t)idx) = v;
    }
  }
  }
  if constexpr (FUSE) {
    // ---- the workgroup whose slab arrives last finishes the loss (round 4: the forward at B <= 32
    // was this kernel + a one-workgroup finish kernel, each at the floor of a dependent launch).
    // Write-through slab stores -> every wave drains its stores -> barrier -> lane 0 draws a ticket
    // (agent-scope add).  The last ticket holder runs the finish on all slabs with L1-bypassing loads
    // and puts the counter back to zero behind it (the caller's word is zero again when the call
    // returns) — the same fixed-order sums as gram_final32_kernel: same
    // bits.  (A first version with plain stores, a release fence per workgroup and an acquire in
    // the last one took 20 us against 14 for the two launches: the fences cost more than a launch.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(g16_lds);
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(fa.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = old == gridDim.x - 1;
      // a ticket past the grid size: the word was not zero on entry (see below) — word 2 of the
      // block records it for good
      if (old >= gridDim.x) __hip_atomic_store(fa.counter + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();
    if (!last) return;
    if (B <= 32)
      final32_body<1024>(g16_lds, slabs, gridDim.x, T, P, B, fa);
    else
      final64_body<1024>(g16_lds, slabs, gridDim.x, T, P, B, fa);
    // The word goes back to zero at the END of the finish.  Under the contract (zero on entry) every
    // workgroup has drawn its ticket by now, so it reads exactly gridDim.x.  A word that was NOT zero
    // on entry (an aborted earlier launch, a caller's bug) makes a middle workgroup take itself for
    // the last one: that cannot be repaired and — ADVICE rounds 4 and 5 — cannot be detected here
    // with certainty either (the late workgroups may not have drawn their tickets yet when this
    // check runs).  What is certain: each of them draws a ticket >= gridDim.x and sets the STICKY
    // error word (word 2 of the block), and leaves word 0 non-zero again behind this reset.  So the
    // violating call returns NaN if it can see the damage and every later call on the block returns
    // NaN until the host zeroes the block: loud, and not "healed" by the library.
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned seen = __hip_atomic_load(fa.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned sticky = __hip_atomic_load(fa.counter + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen != gridDim.x || sticky != 0u) *fa.loss_out = __builtin_nanf("");
      __hip_atomic_store(fa.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int PWMAX, bool FULL>
__global__ __launch_bounds__(256) void gram16_kernel(const float* __restrict__ emb, int64_t ld,
                                                     int B, int E, int T, int P, int kchunk,
                                                     int KS, int vec_ok,
                                                     float* __restrict__ slabs) {
  gram16_body<PWMAX, FULL, false>(emb, ld, B, E, T, P, kchunk, KS, vec_ok, slabs, FinalArgs{});
}
template <int PWMAX, bool FULL>
__global__ __launch_bounds__(1024) void gram16_fused_kernel(const float* __restrict__ emb, int64_t ld,
                                                            int B, int E, int T, int P, int kchunk,
                                                            int KS, int vec_ok,
                                                            float* __restrict__ slabs, FinalArgs fa) {
  gram16_body<PWMAX, FULL, true>(emb, ld, B, E, T, P, kchunk, KS, vec_ok, slabs, fa);
}


// ---------------------------------------------------------------------------------------
// gram16x6_kernel: the raw Gram for 64 < B <= 256 on the bf16 matrix cores with BOTH operands
// split into three bf16 planes, e = e1 + e2 + e3 exactly (24 mantissa bits), and the six
// products whose weight is >= 2^-16 of the leading one:
//   e.f ~= e1.f1 + e1.f2 + e2.f1 + e1.f3 + e3.f1 + e2.f2     (dropped: 2^-24 relative)
// every bf16 x bf16 product is exact in float32 and the sum runs in float32, smallest terms
// first, so the result is float32-equivalent at 6 x 16 cycles per 32-deep step
// instead of 8 x 32 for the float32 MFMA.  Same decomposition, slab format and finishing
// kernels as gram16_kernel.
//   * the workgroup's [B x kchunk] slice of E is loaded ONCE (one burst into registers), split
//     once — each element of E is split exactly once on the whole chip — and written to LDS
//     64 columns at a time as plane images [plane][16-byte piece][row (+ 1 pad)]: the fragment
//     of lane (i, g) for k-step ks and tile t is unit (plane * 8 + 4 ks + g) * (Bp + 1) + 16 t
//     + i.  The pad unit per piece keeps the 8-byte staging stores of 16 consecutive lanes
//     (one row, its 16 column quads = 8 pieces) on 32 different banks — with a piece stride of
//     Bp units they all fell on the same four (PMC: 2.1 M conflict cycles per launch) — at the
//     price of one 2-way slot per fragment read;
//   * A and B fragments are the same thing (rows of E), a pair (ti, tj) costs 3 fragment
//     reads (tile tj; tile ti is kept while the wave stays in row ti) and 6 MFMAs per k-step.
// grid S; block 256; dynamic LDS 3 * 8 * (Bp + 1) * 16 bytes.
constexpr int kX6Sub = 64;       // columns staged per pass

__device__ __forceinline__ void split3_bf16x(float x, unsigned& h1, unsigned& h2, unsigned& h3) {
  const unsigned short a = f32_to_bf16(x);
  const float r1 = x - bf16_to_f32(a);
  const unsigned short b = f32_to_bf16(r1);
  h1 = a;
  h2 = b;
  h3 = f32_to_bf16(r1 - bf16_to_f32(b));
}
typedef unsigned gx_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16bf(gx_u32x4 a, gx_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gx_bf16x8, a),
                                                 __builtin_bit_cast(gx_bf16x8, b), c, 0, 0, 0);
}

// DUAL: two accumulation chains per pair (needs 2 x 4 x PWMAX accumulator registers)
template <int PWMAX, int NSUB, bool DUAL>   // pairs per wave (max), 64-column passes per workgroup
__global__ __launch_bounds__(256) void gram16x6_kernel(const float* __restrict__ emb, int64_t ld,
                                                       int B, int E, int T, int P,
                                                       float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned x6_lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Bp = 16 * T;
  const int k0 = blockIdx.x * (NSUB * kX6Sub);
  // ---- one burst: thread u-th float4 = row (tid + 256 u) >> 4, columns 4 ((tid + 256 u) & 15)
  //      of every pass (requires aligned rows and the slice inside E: checked by the host)
  constexpr int RMAX = 16;                                  // Bp * 16 / 256 <= 16 for Bp <= 256
  f32x4 v[NSUB][RMAX];
  const int nq = Bp * 16;
#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int q = u * 256 + threadIdx.x;                    // no branch around a load:
      int row = q >> 4;                                        // rows past the end re-read B - 1
      row = row < B ? row : B - 1;
      v[sb][u] = *reinterpret_cast<const f32x4*>(emb + (int64_t)row * ld + k0 + sb * kX6Sub + 4 * (q & 15));
    }
  // this wave's pairs
  const int p_begin = (int)(((long)P * wid) / 4), p_end = (int)(((long)P * (wid + 1)) / 4);
  const int np = p_end - p_begin;
  int ti0, tj0;
  decode_pair16(p_begin < P ? p_begin : 0, T, ti0, tj0);
  // two accumulation chains per pair (three products each): a chain of dependent 16x16x32
  // MFMAs issues slower than independent ones
  f32x4 acc[PWMAX], acc2[DUAL ? PWMAX : 1];
#pragma unroll
  for (int lp = 0; lp < PWMAX; ++lp) acc[lp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int lp = 0; lp < (DUAL ? PWMAX : 1); ++lp) acc2[lp] = f32x4{0.f, 0.f, 0.f, 0.f};
  const gx_u32x4* img = reinterpret_cast<const gx_u32x4*>(x6_lds);

#pragma unroll
  for (int sb = 0; sb < NSUB; ++sb) {
    if (sb) __syncthreads();                                // everyone is done with the last pass
    // split and store: float4 (row, c4) -> planes: unit (plane * 8 + c4 / 2) * Bp + row, half c4 & 1
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int q = u * 256 + threadIdx.x;
      if (q < nq) {
        const int row = q >> 4, c4 = q & 15;
        unsigned h[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = row < B ? v[sb][u][c] : 0.f;
          split3_bf16x(x, h[0][c], h[1][c], h[2][c]);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          uint2 w2;
          w2.x = h[pl][0] | (h[pl][1] << 16);
          w2.y = h[pl][2] | (h[pl][3] << 16);
          *reinterpret_cast<uint2*>(&x6_lds[(((pl * 8 + (c4 >> 1)) * (Bp + 1) + row) << 2) + ((c4 & 1) << 1)]) = w2;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const gx_u32x4* fb = img + (4 * ks + g) * (Bp + 1) + i;   // + plane * 8 * (Bp + 1) + 16 * tile
      int ti = ti0, tj = tj0;
      gx_u32x4 a[3], b[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[pl] = fb[pl * 8 * (Bp + 1) + 16 * ti];
        b[pl] = fb[pl * 8 * (Bp + 1) + 16 * tj];
      }
#pragma unroll
      for (int lp = 0; lp < PWMAX; ++lp) {
        if (lp < np) {
          int tj_n = tj + 1, ti_n = ti;
          if (tj_n == T) {
            ti_n = ti + 1 < T ? ti + 1 : ti;
            tj_n = ti_n;
          }
          gx_u32x4 a_n[3], b_n[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            b_n[pl] = fb[pl * 8 * (Bp + 1) + 16 * tj_n];
            a_n[pl] = fb[pl * 8 * (Bp + 1) + 16 * ti_n];
          }
          __builtin_amdgcn_sched_barrier(0);     // next pair's reads fly under these MFMAs
          if constexpr (DUAL) {
            f32x4 c = acc[lp], c2 = acc2[lp];
            c2 = mfma16bf(a[2], b[0], c2);         // chain 2: the small terms
            c = mfma16bf(a[1], b[0], c);
            c2 = mfma16bf(a[0], b[2], c2);
            c = mfma16bf(a[0], b[1], c);
            c2 = mfma16bf(a[1], b[1], c2);
            c = mfma16bf(a[0], b[0], c);
            acc[lp] = c;
            acc2[lp] = c2;
          } else {
            f32x4 c = acc[lp];
            c = mfma16bf(a[2], b[0], c);
            c = mfma16bf(a[0], b[2], c);
            c = mfma16bf(a[1], b[1], c);
            c = mfma16bf(a[1], b[0], c);
            c = mfma16bf(a[0], b[1], c);
            c = mfma16bf(a[0], b[0], c);
            acc[lp] = c;
          }
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            a[pl] = a_n[pl];
            b[pl] = b_n[pl];
          }
          ti = ti_n;
          tj = tj_n;
        }
      }
    }
  }
  float* slab = slabs + (int64_t)blockIdx.x * P * 256;
#pragma unroll
  for (int lp = 0; lp < PWMAX; ++lp)
    if (lp < np)
      *reinterpret_cast<f32x4*>(slab + (int64_t)(p_begin + lp) * 256 + 4 * lane) =
          DUAL ? acc[lp] + acc2[DUAL ? lp : 0] : acc[lp];
}

// ---- pair terms of one (row, column) entry, shared by every finishing kernel --------------
struct PairEval {
  float s, mp, mn;   // clamped similarity and the two masks (mp already has the -eye term)
};
__device__ __forceinline__ PairEval pair_masks(float gn, float d, int same_label, bool diag,
                                               const LossParams& lp) {
  PairEval r;
  r.s = fmaxf(gn, 0.f);
  if (lp.mask_kind == SCL_MASK_LABELS) {
    r.mp = same_label ? 1.f : 0.f;
    r.mn = same_label ? 0.f : 1.f;
  } else if (lp.mask_kind == SCL_MASK_WMS_LIN) {
    r.mp = d < lp.d_beta ? 1.0f - d / lp.d_beta : 0.f;
    r.mn = d < lp.d_beta ? d / lp.d_beta : 1.f;
  } else if (lp.mask_kind == SCL_MASK_WMS_TANH) {
    const float t = eigen_fast_tanh(__fdiv_rn(d, lp.d_beta));
    r.mp = 1.0f - t;
    r.mn = t;
  } else {
    r.mp = 1.0f / (1.0f + expf(lp.d_alpha * (d - lp.d_beta)));
    r.mn = 1.0f / (1.0f + expf(lp.d_alpha * (lp.d_beta - d)));
  }
  if (diag) r.mp -= 1.0f;   // mask_pos - eye (model/losses.py:22,91)
  return r;
}

// One wave evaluates reduction row i.  Lane l holds columns j = l + 64 c (c < C); gn[c] is the
// normalised Gram entry, d[c] / lab[c] the mask input.  Returns the row loss (all lanes) and
// fills g[c] = d loss / d S[i, j] (before the 1 / B of the mean is applied by the caller's
// invB).  Semantics as gram_rows_kernel / model/losses.py:25-58, 94-120.
template <int C, bool HALF = false>
__device__ __forceinline__ float wave_row_eval(int i, int B, int lane, const float (&gn)[C],
                                               const float (&d)[C], const int (&same)[C],
                                               const LossParams& lp, float invB, float (&g)[C]) {
  // HALF: `lane` is the column inside a 32-lane half that owns the row; reductions stay in it
  auto rmax = [](float v) { return HALF ? half_max(v) : wave_max(v); };
  auto rmin = [](float v) { return HALF ? -half_max(-v) : wave_min(v); };
  auto rsum = [](float v) { return HALF ? half_sum(v) : wave_sum(v); };
  PairEval pe[C];
  float vmaxN = -INFINITY, vmaxP = -INFINITY;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = lane + 64 * c;
    pe[c] = pair_masks(gn[c], d[c], same[c], j == i, lp);
    if (j < B) {
      vmaxN = fmaxf(vmaxN, pe[c].s * pe[c].mn);
      vmaxP = fmaxf(vmaxP, pe[c].s * pe[c].mp);
    }
  }
  float max_val = 0.f, min_val = 0.f;
  if (lp.ms_mining) {
    max_val = rmax(vmaxN);
    const float tmp = rmax(vmaxP);
    float vmin = INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (lane + 64 * c < B) vmin = fminf(vmin, (pe[c].s - tmp) * pe[c].mp);
    min_val = rmin(vmin) + tmp;
  }
  float ps = 0.f, ns = 0.f, pterm[C], nterm[C];
  bool selp[C], seln[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float Pm = pe[c].s * pe[c].mp, Nm = pe[c].s * pe[c].mn;
    float kp = pe[c].mp, kn = pe[c].mn;
    if (lp.ms_mining) {
      kp = Pm < max_val + lp.eps ? pe[c].mp : 0.f;
      kn = Nm > min_val - lp.eps ? pe[c].mn : 0.f;
    }
    const bool ok = lane + 64 * c < B;
    selp[c] = ok && kp > 0.f;
    seln[c] = ok && kn > 0.f;
    if (lp.sum_kind == SCL_SUM_PLAIN) {
      pterm[c] = selp[c] ? Pm : 0.f;
      nterm[c] = seln[c] ? Nm : 0.f;
    } else {
      pterm[c] = selp[c] ? expf(-lp.alpha * (Pm - lp.lamb)) : 0.f;
      nterm[c] = seln[c] ? expf(lp.beta * (Nm - lp.lamb)) : 0.f;
    }
    ps += pterm[c];
    ns += nterm[c];
  }
  ps = rsum(ps);
  ns = rsum(ns);
  float rowloss;
  if (lp.sum_kind == SCL_SUM_PLAIN)
    rowloss = ns - ps;
  else
    rowloss = logf(1.0f + ps) / lp.alpha + logf(1.0f + ns) / lp.beta;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float v;
    if (lp.sum_kind == SCL_SUM_PLAIN)
      v = (seln[c] ? pe[c].mn : 0.f) - (selp[c] ? pe[c].mp : 0.f);
    else
      v = nterm[c] / (1.0f + ns) * pe[c].mn - pterm[c] / (1.0f + ps) * pe[c].mp;
    g[c] = gn[c] >= 0.f ? v * invB : 0.f;
  }
  return rowloss;
}

// B <= 32: the whole finish in one 1024-thread workgroup.
//   1. G = sum of the S slabs (fixed order), raw Gram in LDS (both triangles)
//   2. rn_j from the diagonal; 3. one 32-lane HALF-wave per row: masks, mining, row loss,
//      d loss / d S (all 32 rows in one round);
//   4. loss mean; M = rn_i rn_j (g_ij + g_ji) - [i == j] rn_i^2 sum_j (g_ij + g_ji) Gn_ij
// Every global load (slabs, distances / labels) is issued before the first wait.
__global__ __launch_bounds__(1024) void gram_final32_kernel(
    const float* __restrict__ slabs, int S, int T, int P, int B,
    const float* __restrict__ distances, const int64_t* __restrict__ labels, LossParams lp,
    float* __restrict__ coef, float* __restrict__ loss_out) {
  __shared__ float Gr[32][33];       // raw Gram, then normalised
  __shared__ float Gc[32][33];       // d loss / d S
  __shared__ f32x4 part[4][192];     // partial slab sums
  __shared__ float rn[32], rowloss[32];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ri = 2 * wid + (lane >> 5), rj = lane & 31;      // this lane's (row, column)
  const bool rok = ri < B && rj < B;
  // mask inputs of entry (ri, rj): in flight under the slab sums
  float dval = 0.f;
  int same = 0;
  if (rok) {
    if (lp.mask_kind == SCL_MASK_LABELS)
      same = labels[rj] == labels[ri];
    else
      dval = lp.dist_rank3 ? distances[(int64_t)rj * B + ri] : distances[(int64_t)ri * B + rj];
  }
  {
    const int grp = threadIdx.x >> 8, idx = threadIdx.x & 255;
    constexpr int U = 16;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    if (idx < P * 64) {
      const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + idx;
      for (int s0 = grp; s0 < S; s0 += 4 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int sidx = s0 + 4 * u;
          v[u] = sidx < S ? src[(int64_t)sidx * P * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc4 += v[u];
      }
      part[grp][idx] = acc4;
    }
    __syncthreads();
    if (threadIdx.x < P * 64) {
      const f32x4 v = (part[0][idx] + part[1][idx]) + (part[2][idx] + part[3][idx]);
      const int pair = idx >> 6, l = idx & 63;
      int ti, tj;
      decode_pair16(pair, T, ti, tj);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 16 * ti + 4 * (l >> 4) + j, c = 16 * tj + (l & 15);
        Gr[r][c] = v[j];
        if (ti != tj) Gr[c][r] = v[j];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) rn[threadIdx.x] = 1.0f / sqrtf(fmaxf(Gr[threadIdx.x][threadIdx.x], 1e-12f));
  __syncthreads();
  // 3. rows: half-wave (wid, lane >> 5) owns row ri
  const float invB = 1.0f / (float)B;
  float gnv[1], dv[1], gv[1];
  int sv[1];
  gnv[0] = rok ? Gr[ri][rj] * rn[ri] * rn[rj] : 0.f;
  dv[0] = dval;
  sv[0] = same;
  // rows >= B run on zeros (their results are never stored); B is passed as the column count
  const float rl = wave_row_eval<1, true>(ri, B, rj, gnv, dv, sv, lp, invB, gv);
  if (rok) Gc[ri][rj] = gv[0];
  if (rj == 0 && ri < B) rowloss[ri] = rl;
  __syncthreads();
  if (wid == 0) {
    float a = lane < B ? rowloss[lane] : 0.f;
    a = wave_sum(a);
    if (lane == 0) *loss_out = a / (float)B;
  }
  if (!coef) return;
  const float gs = rok ? Gc[ri][rj] + Gc[rj][ri] : 0.f;
  const float c = half_sum(rok ? gs * gnv[0] : 0.f);
  if (rok) {
    const float rni = rn[ri];
    const bool clamped = rni >= 1.0e6f;   // see gram_coef_kernel
    float m = rni * rn[rj] * gs;
    if (rj == ri && !clamped) m -= rni * rni * c;
    coef[(int64_t)ri * B + rj] = m;
  }
}

// The finish of gram_final32_kernel as a device function for the 1024 threads of the last gram16
// workgroup: the same sums in the same order (four slab groups s = g, g + 4, .. added in batches
// of 16, then (g0 + g1) + (g2 + g3)), the same row evaluation (a half-wave per row, all rows at
// once).  Slabs come in through L1-bypassing loads (see st_sc1_x4).  LDS (>= 24 KB of the caller's
// dynamic block): Gr[32][33] | Gc[32][33] | rn[32] | rowloss[32] | part[4][192] x 16 B.
template <int NT>
__device__ void final32_body(float* lds, const float* slabs, int S, int T, int P, int B,
                             const FinalArgs& fa) {
  static_assert(NT == 1024, "a half-wave per row");
  float(*Gr)[33] = reinterpret_cast<float(*)[33]>(lds);
  float(*Gc)[33] = reinterpret_cast<float(*)[33]>(lds + 32 * 33);
  float* rn = lds + 2 * 32 * 33;
  float* rowloss = rn + 32;
  f32x4* part = reinterpret_cast<f32x4*>(lds + 2 * 32 * 33 + 64);       // [4][192]
  const LossParams& lp = fa.lp;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ri = 2 * wid + (lane >> 5), rj = lane & 31;      // this lane's (row, column)
  const bool rok = ri < B && rj < B;
  float dval = 0.f;
  int same = 0;
  if (rok) {
    if (lp.mask_kind == SCL_MASK_LABELS)
      same = fa.labels[rj] == fa.labels[ri];
    else
      dval = lp.dist_rank3 ? fa.distances[(int64_t)rj * B + ri] : fa.distances[(int64_t)ri * B + rj];
  }
  {
    const int grp = threadIdx.x >> 8, idx = threadIdx.x & 255;
    constexpr int U = 16;
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    if (idx < P * 64) {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(slabs), 0, S * P * 1024, 0x00020000);
      for (int s0 = grp; s0 < S; s0 += 4 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                          // (branch-free: clamped, masked below)
          const int sidx = s0 + 4 * u < S ? s0 + 4 * u : S - 1;
          v[u] = ld_sc1_x4(rsrc, (unsigned)((sidx * P * 64 + idx) * 16));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc4 += s0 + 4 * u < S ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      part[grp * 192 + idx] = acc4;
    }
    __syncthreads();
    if (threadIdx.x < P * 64) {
      const f32x4 v = (part[idx] + part[192 + idx]) + (part[2 * 192 + idx] + part[3 * 192 + idx]);
      const int pair = idx >> 6, l = idx & 63;
      int ti, tj;
      decode_pair16(pair, T, ti, tj);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 16 * ti + 4 * (l >> 4) + j, c = 16 * tj + (l & 15);
        Gr[r][c] = v[j];
        if (ti != tj) Gr[c][r] = v[j];
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 32) rn[threadIdx.x] = 1.0f / sqrtf(fmaxf(Gr[threadIdx.x][threadIdx.x], 1e-12f));
  __syncthreads();
  const float invB = 1.0f / (float)B;
  float gnv[1], dv[1], gv[1];
  int sv[1];
  gnv[0] = rok ? Gr[ri][rj] * rn[ri] * rn[rj] : 0.f;
  dv[0] = dval;
  sv[0] = same;
  const float rl = wave_row_eval<1, true>(ri, B, rj, gnv, dv, sv, lp, invB, gv);
  if (rok) Gc[ri][rj] = gv[0];
  if (rj == 0 && ri < B) rowloss[ri] = rl;
  __syncthreads();
  if (wid == 0) {
    float a = lane < B ? rowloss[lane] : 0.f;
    a = wave_sum(a);
    if (lane == 0) *fa.loss_out = a / (float)B;
  }
  if (!fa.coef) return;
  const float gs = rok ? Gc[ri][rj] + Gc[rj][ri] : 0.f;
  const float c = half_sum(rok ? gs * gnv[0] : 0.f);
  if (rok) {
    const float rni = rn[ri];
    const bool clamped = rni >= 1.0e6f;   // see gram_coef_kernel
    float m = rni * rn[rj] * gs;
    if (rj == ri && !clamped) m -= rni * rni * c;
    fa.coef[(int64_t)ri * B + rj] = m;
  }
}

// Round 5: the same for 32 < B <= 64 — what gram_reduce_kernel + gram_rows_wave_kernel<1> +
// gram_coef_kernel do in three launches behind gram16_kernel, by the 1024 threads of the last gram16
// workgroup to arrive, with the SAME sums in the SAME order (same bits):
//   1. per tile pair, four waves add the slabs w, w + 4, .. in batches of 16 and combine as
//      (p0 + p1) + (p2 + p3) — sixteen waves take four pairs at a time;
//   2. a wave per row: normalised entries, masks, mining, row loss, d loss / d S (wave_row_eval<1>);
//   3. loss mean and M = coef, a wave per row (gram_coef_kernel's block_reduce over 256 threads of
//      which only the first 64 hold a term: the wave sum, then three additions of zero).
// LDS (>= 67 KB of the caller's dynamic block): G | GN | GC [64][65] | rn[64] | rowloss[64] |
// part[16][64] x 16 B.
constexpr int kFinal64Lds = (3 * 64 * 65 + 128) * 4 + 16 * 64 * 16;
template <int NT>
__device__ void final64_body(float* lds, const float* slabs, int S, int T, int P, int B,
                             const FinalArgs& fa) {
  static_assert(NT == 1024, "sixteen waves");
  constexpr int LD = 65;
  float* G = lds;
  float* GN = G + 64 * LD;
  float* GC = GN + 64 * LD;
  float* rn = GC + 64 * LD;
  float* rowloss = rn + 64;
  f32x4* part = reinterpret_cast<f32x4*>(rowloss + 64);
  const LossParams& lp = fa.lp;
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slabs), 0, S * P * 1024, 0x00020000);
    const int w = wid & 3, pg = wid >> 2;
    constexpr int U = 16;
    for (int p0 = 0; p0 < P; p0 += 4) {
      const int pair = p0 + pg;
      f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
      if (pair < P) {
        for (int s0 = w; s0 < S; s0 += 4 * U) {
          f32x4 v[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {                        // (branch-free: clamped, masked below)
            const int sidx = s0 + 4 * u < S ? s0 + 4 * u : S - 1;
            v[u] = ld_sc1_x4(rsrc, (unsigned)(((sidx * P + pair) * 64 + lane) * 16));
          }
#pragma unroll
          for (int u = 0; u < U; ++u) acc4 += s0 + 4 * u < S ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      part[wid * 64 + lane] = acc4;
      __syncthreads();
      if (w == 0 && pair < P) {
        const f32x4 v = (part[(4 * pg) * 64 + lane] + part[(4 * pg + 1) * 64 + lane]) +
                        (part[(4 * pg + 2) * 64 + lane] + part[(4 * pg + 3) * 64 + lane]);
        int ti, tj;
        decode_pair16(pair, T, ti, tj);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 16 * ti + 4 * (lane >> 4) + j, c = 16 * tj + (lane & 15);
          G[r * LD + c] = v[j];
          if (ti != tj) G[c * LD + r] = v[j];
        }
      }
      __syncthreads();
    }
  }
  const float invB = 1.0f / (float)B;
  for (int i = wid; i < B; i += 16) {                          // (wave-uniform)
    const float rni = 1.0f / sqrtf(fmaxf(G[i * LD + i], 1e-12f));
    const int j = lane;
    float gn[1] = {0.f}, d[1] = {0.f}, g[1];
    int same[1] = {0};
    if (j < B) {
      const float rnj = 1.0f / sqrtf(fmaxf(G[j * LD + j], 1e-12f));
      gn[0] = G[i * LD + j] * rni * rnj;
      if (lp.mask_kind == SCL_MASK_LABELS)
        same[0] = fa.labels[j] == fa.labels[i];
      else
        d[0] = lp.dist_rank3 ? fa.distances[(int64_t)j * B + i] : fa.distances[(int64_t)i * B + j];
    }
    const float rl = wave_row_eval<1>(i, B, lane, gn, d, same, lp, invB, g);
    if (j < B) {
      GN[i * LD + j] = gn[0];
      GC[i * LD + j] = g[0];
    }
    if (lane == 0) {
      rowloss[i] = rl;
      rn[i] = rni;
    }
  }
  __syncthreads();
  if (wid == 0) {
    float a = 0.f;
    if (lane < B) a += rowloss[lane];
    a = wave_sum(a);
    a = ((a + 0.f) + 0.f) + 0.f;                               // gram_coef_kernel's empty waves
    if (lane == 0) *fa.loss_out = a / (float)B;
  }
  if (!fa.coef) return;
  for (int i = wid; i < B; i += 16) {
    const int j = lane;
    float gs = 0.f, c = 0.f;
    if (j < B) {
      gs = GC[i * LD + j] + GC[j * LD + i];
      c += gs * GN[i * LD + j];
    }
    c = wave_sum(c);
    c = ((c + 0.f) + 0.f) + 0.f;
    const float rni = rn[i];
    const bool clamped = rni >= 1.0e6f;                        // see gram_coef_kernel
    if (j < B) {
      float m = rni * rn[j] * gs;
      if (j == i && !clamped) m -= rni * rni * c;
      fa.coef[(int64_t)i * B + j] = m;
    }
  }
}

// grid P; block 256: full raw Gram (both triangles) = fixed-order sum of the S slabs of one
// tile pair.  Wave w adds slabs s = w, w + 4, ...; the four partial sums meet in LDS.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ slabs, int S,
                                                          int T, int P, int B,
                                                          float* __restrict__ gfull) {
  __shared__ f32x4 part[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int pair = blockIdx.x;
  const f32x4* src = reinterpret_cast<const f32x4*>(slabs) + (int64_t)pair * 64 + lane;
  // sixteen slab loads of a wave in flight at once (the kernel is a latency-bound reader of
  // S x 1 KB per pair: with four in flight it took 7.7 us for 20 MB), summed in a fixed order
  constexpr int U = 16;
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = wid; s0 < S; s0 += 4 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sidx = s0 + 4 * u;
      v[u] = sidx < S ? src[(int64_t)sidx * P * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc4 += v[u];
  }
  part[wid][lane] = acc4;
  __syncthreads();
  if (wid != 0) return;
  const f32x4 v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  int ti, tj;
  decode_pair16(pair, T, ti, tj);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 16 * ti + 4 * (lane >> 4) + j, c = 16 * tj + (lane & 15);
    if (r < B && c < B) {
      // (diagonal tiles hold both (r, c) and (c, r) themselves: each is written by its owner only —
      // the bf16x6 Gram adds the cross products of the two in a different order, so the two need
      // not agree in the last bit, and which of two stores lands last must not decide a result)
      gfull[(int64_t)r * B + c] = v[j];
      if (ti != tj) gfull[(int64_t)c * B + r] = v[j];
    }
  }
}

// grid ceil(B / 4); block 256: wave w evaluates reduction row 4 blockIdx.x + w from the full
// raw Gram matrix.  C = ceil(B / 64) columns per lane.
template <int C>
__global__ __launch_bounds__(256) void gram_rows_wave_kernel(
    const float* __restrict__ gfull, int B, const float* __restrict__ distances,
    const int64_t* __restrict__ labels, LossParams lp, float* __restrict__ gn_out,
    float* __restrict__ gc_out, float* __restrict__ rn_out, float* __restrict__ rowloss_out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wid;
  if (i >= B) return;                                    // wave-uniform
  const float rni = 1.0f / sqrtf(fmaxf(gfull[(int64_t)i * B + i], 1e-12f));
  const int64_t labi = labels ? labels[i] : 0;
  float gn[C], d[C], g[C];
  int same[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = lane + 64 * c;
    gn[c] = 0.f;
    d[c] = 0.f;
    same[c] = 0;
    if (j < B) {
      const float rnj = 1.0f / sqrtf(fmaxf(gfull[(int64_t)j * B + j], 1e-12f));
      gn[c] = gfull[(int64_t)i * B + j] * rni * rnj;
      if (lp.mask_kind == SCL_MASK_LABELS)
        same[c] = labels[j] == labi;
      else
        d[c] = lp.dist_rank3 ? distances[(int64_t)j * B + i] : distances[(int64_t)i * B + j];
    }
  }
  const float rl = wave_row_eval<C>(i, B, lane, gn, d, same, lp, 1.0f / (float)B, g);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int j = lane + 64 * c;
    if (j < B) {
      gn_out[(int64_t)i * B + j] = gn[c];
      gc_out[(int64_t)i * B + j] = g[c];
    }
  }
  if (lane == 0) {
    rowloss_out[i] = rl;
    rn_out[i] = rni;
  }
}

// B <= 32: grad[r, e] = g * sum_j M[row_begin + r, j] emb[j, e] on v_mfma_f32_32x32x2_f32 with
// ONE wave per 32 columns (grid E / 128 workgroups of four waves: 1024 waves at E = 32768, so
// every CU streams; the previous kernel ran 64 workgroups).  All loads of a wave are issued
// before its first MFMA.
__global__ __launch_bounds__(256) void gram_bwd32_kernel(const float* __restrict__ emb, int64_t ld,
                                                         int B, int E,
                                                         const float* __restrict__ coef,
                                                         const float* __restrict__ grad_loss,
                                                         int row_begin, int row_count,
                                                         float* __restrict__ grad, int64_t ldg) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int e = (blockIdx.x * 4 + wid) * 32 + r;
  if ((blockIdx.x * 4 + wid) * 32 >= E) return;          // wave-uniform
  const bool col_ok = e < E;
  float a[16], b[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = 2 * t + h;
    const bool jok = j < B;
    a[t] = (jok && r < row_count) ? coef[(int64_t)(row_begin + r) * B + j] : 0.f;
    b[t] = (jok && col_ok) ? emb[(int64_t)j * ld + e] : 0.f;
  }
  f32x16 acc = zero16();
#pragma unroll
  for (int t = 0; t < 16; ++t)
    if (2 * t < B) acc = mfma32(a[t], b[t], acc);
  const float g = grad_loss ? *grad_loss : 1.0f;
  if (!col_ok) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int orow = acc_row(q, h);
    if (orow < row_count) grad[(int64_t)orow * ldg + e] = g * acc[q];
  }
}

// =======================================================================================
// Round 6: the forward for 32 < B <= 208 as ONE launch (was four: Gram | slab sums | rows | M).
//
//   gram16x6p_body   the bf16x6 Gram of gram16x6_kernel with BOTH 64-column passes of the
//                    workgroup's slice resident in LDS (2 x 3 planes x 8 pieces x (Bp + 1) x 16 B =
//                    145 KB at B = 192) and the PAIR loop outermost: a pair's 24 MFMAs (2 passes x 2
//                    k-steps x 6 products, the same products in the same order into the same two
//                    chains: same bits) finish it, and its 1 KB of slab goes out while the next
//                    pair's MFMAs run — the 78 KB of slab per workgroup used to leave in one tail
//                    burst after the last MFMA (store-issue-bound, ~4 us of a 22 us kernel).
//   persist_tail     behind a grid barrier: phase 2, the slab sums, spread over EVERY workgroup
//                    (item = one 16-byte unit of a pair tile; workgroup w owns items [N w / W, N (w +
//                    1) / W), its four waves are gram_reduce_kernel's four chains s = w, w + 4, ..:
//                    the same sums in the same order); barrier; phase 3, a wave per row
//                    (wave_row_eval, as gram_rows_wave_kernel); barrier; phase 4, a workgroup per row
//                    of M (as gram_coef_kernel).  Same bits as the four launches.
//
// Cross-workgroup visibility (MI355X_MICROARCH.md, "Valid forms", first row of the table): every
// handed-off byte is stored sc1 (write-through), every storing wave drains its stores, a workgroup
// barrier, ONE lane's agent-scope add to the arrival counter; consumers poll the counter with an
// sc1 load, join a workgroup barrier and read every handed-off byte with sc1 loads.  No fence.
//
// The barrier SPINS, so the grid must be co-resident: the host only takes this path with at most
// one workgroup per CU of the device.  Correctness still never depends on it: the spin is bounded
// (a few ms), a workgroup that runs out of patience raises the ABORT bit, every workgroup that sees
// it leaves at once, and the LAST workgroup to leave the kernel — by then every slab is complete —
// runs phases 2-4 alone with the same routines (partition 0 of 1: the same sums in the same order,
// the same bits).  That is what happens when two such kernels from two streams each hold part of
// the chip, or when another kernel holds CUs for longer than the limit.
// sync words (8-byte aligned, zero on entry, zero on return): [0] arrivals | ABORT bit, [1] leavers.
// =======================================================================================
#ifdef SCL_DIAG
typedef __attribute__((address_space(1))) unsigned* u32_gptr_t;
constexpr unsigned kAbortBit = 0x80000000u;
constexpr int kPersistSpinLimit = 20000;        // x (poll + s_sleep) ~ 5-10 ms

struct PersistArgs {
  unsigned* sync;
  const float* distances;
  const int64_t* labels;
  LossParams lp;
  const float* slabs;        // [S][P][256]
  float* gsum;               // [P][256] summed pair tiles
  float *gn, *gc, *rn, *rowloss;
  float* coef;
  float* loss_out;
  int S, T, P, B;
  int spin_limit;
  unsigned long long* stamps;   // diagnostics (scl_debug_set_variant(40), scripts/loss_stamps.py):
                                // [workgroup][16] shader-clock stamps of thread 0; null otherwise
};
#define PSTAMP(a, k)                                                                         \
  do {                                                                                       \
    if (SCL_DIAG_ONLY((a).stamps != nullptr) && threadIdx.x == 0)                            \
      (a).stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();                      \
  } while (0)

__device__ __forceinline__ void st_sc1_f32(float* p, float v) {
  __hip_atomic_store((__attribute__((address_space(1))) float*)p, v, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1_f32(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off, 0, 16));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Arrive at the grid barrier number `phase` (1, 2, 3) and wait for the others.  false: aborted.
// flag: one LDS word.  Every wave's stores are drained before the arrival is counted.
__device__ __forceinline__ bool persist_barrier(unsigned* sync, unsigned target, int limit, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 0;
    for (int spins = 0;; ++spins) {
      const unsigned x = __hip_atomic_load((u32_gptr_t)sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (x & kAbortBit) break;
      if (x >= target) {
        ok = 1;
        break;
      }
      if (spins >= limit) {
        __hip_atomic_fetch_or(sync, kAbortBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    *flag = ok;
  }
  __syncthreads();
  const int ok = *flag;
  __syncthreads();
  return ok != 0;
}

// entry (r, c) of the summed Gram: tile (min, max) of the upper triangle, lane 16 (row >> 2) + col,
// register row & 3 (gram16 / gram16x6 accumulator layout).  Diagonal tiles are read directly, as
// gram_reduce_kernel writes them.
__device__ __forceinline__ unsigned gsum_offset(int r, int c, int T) {
  int tr = r >> 4, tc = c >> 4, rr = r & 15, cc = c & 15;
  if (tr > tc) {
    const int t = tr;
    tr = tc;
    tc = t;
    const int q = rr;
    rr = cc;
    cc = q;
  }
  const int pair = tr * T - tr * (tr - 1) / 2 + (tc - tr);
  return (unsigned)(((pair * 64 + 16 * (rr >> 2) + cc) * 4 + (rr & 3)) * 4);
}

// phase 2: partition w of W.  lds: [4][64] f32x4.
__device__ __forceinline__ void persist_reduce(const PersistArgs& a, int w, int W, f32x4* part) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = 64 * a.P;
  const int lo = (int)(((long)N * w) / W), hi = (int)(((long)N * (w + 1)) / W);
  const __amdgpu_buffer_rsrc_t rsrc = rsrc_of(a.slabs, (unsigned)((size_t)a.S * a.P * 1024));
  constexpr int U = 16;                                        // gram_reduce_kernel's batches
  for (int base = lo; base < hi; base += 64) {
    const int item = base + lane < hi ? base + lane : hi - 1;   // (idle lanes re-read the last item)
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = wid; s0 < a.S; s0 += 4 * U) {
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {                            // (branch-free: clamped, masked below)
        const int sidx = s0 + 4 * u < a.S ? s0 + 4 * u : a.S - 1;
        v[u] = ld_sc1_x4(rsrc, (unsigned)((sidx * a.P * 64 + item) * 16));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc4 += s0 + 4 * u < a.S ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    part[wid * 64 + lane] = acc4;
    __syncthreads();
    if (wid == 0 && base + lane < hi) {
      const f32x4 v = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
      st_sc1_x4(a.gsum + (int64_t)(base + lane) * 4, v);
    }
    __syncthreads();
  }
}

// phase 3: rows i = w + W wid, + 4 W, ..: one wave per row, as gram_rows_wave_kernel<C>.
template <int C>
__device__ __forceinline__ void persist_rows(const PersistArgs& a, int w, int W) {
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int B = a.B;
  const __amdgpu_buffer_rsrc_t gs = rsrc_of(a.gsum, (unsigned)(a.P * 1024));
  const LossParams& lp = a.lp;
  for (int i = w + W * wid; i < B; i += 4 * W) {               // (wave-uniform)
    const float gii = ld_sc1_f32(gs, gsum_offset(i, i, a.T));
    float gij[C], gjj[C], d[C], gn[C], g[C];
    int same[C];
    const int64_t labi = a.labels ? a.labels[i] : 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      const int jj = j < B ? j : B - 1;                        // (clamped: no branch around a load)
      gij[c] = ld_sc1_f32(gs, gsum_offset(i, jj, a.T));
      gjj[c] = ld_sc1_f32(gs, gsum_offset(jj, jj, a.T));
      d[c] = 0.f;
      same[c] = 0;
      if (j < B) {
        if (lp.mask_kind == SCL_MASK_LABELS)
          same[c] = a.labels[j] == labi;
        else
          d[c] = lp.dist_rank3 ? a.distances[(int64_t)j * B + i] : a.distances[(int64_t)i * B + j];
      }
    }
    const float rni = 1.0f / sqrtf(fmaxf(gii, 1e-12f));
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      const float rnj = 1.0f / sqrtf(fmaxf(gjj[c], 1e-12f));
      gn[c] = j < B ? gij[c] * rni * rnj : 0.f;
    }
    const float rl = wave_row_eval<C>(i, B, lane, gn, d, same, lp, 1.0f / (float)B, g);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      if (j < B) {
        st_sc1_f32(a.gn + (int64_t)i * B + j, gn[c]);
        st_sc1_f32(a.gc + (int64_t)i * B + j, g[c]);
      }
    }
    if (lane == 0) {
      st_sc1_f32(a.rowloss + i, rl);
      st_sc1_f32(a.rn + i, rni);
    }
  }
}

// phase 4: rows i = w, w + W, ..: the whole workgroup per row, as gram_coef_kernel (B <= 256:
// thread j holds column j).  scratch: 32 floats.
__device__ __forceinline__ void persist_coef(const PersistArgs& a, int w, int W, float* scratch) {
  const int B = a.B, j = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rgn = rsrc_of(a.gn, (unsigned)(B * B * 4)), rgc = rsrc_of(a.gc, (unsigned)(B * B * 4)),
                               rrn = rsrc_of(a.rn, (unsigned)(B * 4)), rrl = rsrc_of(a.rowloss, (unsigned)(B * 4));
  const int jj = j < B ? j : B - 1;
  if (w == 0) {
    float s = j < B ? ld_sc1_f32(rrl, (unsigned)(jj * 4)) : 0.f;
    s = block_reduce<0>(s, scratch);
    if (threadIdx.x == 0) *a.loss_out = s / (float)B;
  }
  if (!a.coef) return;
  const float rnj = ld_sc1_f32(rrn, (unsigned)(jj * 4));
  for (int i = w; i < B; i += W) {
    const float gs = ld_sc1_f32(rgc, (unsigned)((i * B + jj) * 4)) + ld_sc1_f32(rgc, (unsigned)((jj * B + i) * 4));
    float c = j < B ? gs * ld_sc1_f32(rgn, (unsigned)((i * B + jj) * 4)) : 0.f;
    c = block_reduce<0>(c, scratch);
    const float rni = ld_sc1_f32(rrn, (unsigned)(i * 4));
    const bool clamped = rni >= 1.0e6f;                        // see gram_coef_kernel
    if (j < B) {
      float m = rni * rnj * gs;
      if (j == i && !clamped) m -= rni * rni * c;
      a.coef[(int64_t)i * B + j] = m;
    }
    __syncthreads();                                           // scratch is reused by the next row
  }
}

__device__ __forceinline__ void persist_rows_any(const PersistArgs& a, int w, int W) {
  if (a.B <= 64)
    persist_rows<1>(a, w, W);
  else if (a.B <= 128)
    persist_rows<2>(a, w, W);
  else if (a.B <= 192)
    persist_rows<3>(a, w, W);
  else
    persist_rows<4>(a, w, W);
}

// Everything behind the Gram phase of a 256-thread workgroup.  lds: >= 4.5 KB, dead Gram data.
// One loop body serves both the normal run (partition blockIdx of gridDim, grid barriers between
// the phases) and the repair run of the last workgroup out (partition 0 of 1, its own barriers).
__device__ __forceinline__ void persist_tail(const PersistArgs& a, float* lds) {
  f32x4* part = reinterpret_cast<f32x4*>(lds);                 // [4][64]
  float* scratch = lds + 1024;                                 // [32]
  int* flag = reinterpret_cast<int*>(lds + 1024 + 32);
  int W = gridDim.x, w = blockIdx.x;
  bool solo = false;
  for (;;) {
    auto barrier = [&](unsigned k) -> bool {
      if (solo) {                                              // own stores -> own sc1 loads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return true;
      }
      return persist_barrier(a.sync, k * gridDim.x, a.spin_limit, flag);
    };
    PSTAMP(a, 2);
    bool ok = barrier(1u);
    PSTAMP(a, 3);
    if (ok) {
      persist_reduce(a, w, W, part);
      PSTAMP(a, 4);
      ok = barrier(2u);
      PSTAMP(a, 5);
    }
    if (ok) {
      persist_rows_any(a, w, W);
      PSTAMP(a, 6);
      ok = barrier(3u);
      PSTAMP(a, 7);
    }
    if (ok) persist_coef(a, w, W, scratch);
    PSTAMP(a, 8);
    if (solo) break;
    // ---- leave.  The last workgroup out repairs an aborted run; it puts the words back to zero.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned left = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int role = 0;
      if (left == gridDim.x - 1) {
        const unsigned x = __hip_atomic_load((u32_gptr_t)a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = (x & kAbortBit) ? 2 : 1;
      }
      *flag = role;
    }
    __syncthreads();
    const int role = *flag;
    __syncthreads();
    if (role == 0) return;
    if (role == 1) break;
    solo = true;                                               // every slab is complete by now
    w = 0;
    W = 1;
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

#endif   // SCL_DIAG (the persistent tail)

// grid E / 128; block 256; dynamic LDS 2 * 3 * 8 * (Bp + 1) * 16 bytes (Bp = 16 T <= 208).
// gram16x6_kernel's products in gram16x6_kernel's order (same bits per pair), scheduled differently:
//   * STRIPS: a wave walks strips of TWO tile rows (r0, r0 + 1) from the diagonal to the right; a
//     step = one tile column tj = the pairs (r0, tj) and (r0 + 1, tj).  The A fragments of both rows
//     stay in registers along the strip and one read of the B fragments feeds two pairs: ~6.5
//     fragment reads per pair and k-step instead of ~14 — the pair loop was bound by the LDS read
//     rate (stamps, round 6: 18.5 k cycles for 7.7 k cycles of MFMA), not by the matrix pipe.
//     The flat list of steps is cut into four ranges of (almost) equal pair counts, one per wave.
//   * TWO PASSES, the second finishing pair by pair: pass 0 (columns 0..63 of the slice) runs while
//     the loads of pass 1 are still in flight and keeps every pair's partial sums in registers;
//     pass 1 runs step-outermost, so a step's pairs are complete after its 24 MFMAs and their 2 KB
//     of slab leave while the next step computes (the 78 KB per workgroup used to go out in one
//     burst behind the last MFMA: store-issue-bound).
// DUAL as gram16x6_kernel (two accumulation chains per pair; the host picks it by the same rule).
// SMAX: steps per wave (T <= 12: 11; T = 13: 13).  SC1: write-through slab stores.
template <bool DUAL>
__device__ __forceinline__ void x6_pair_mfmas(const gx_u32x4 (&A)[3], const gx_u32x4 (&Bf)[3], f32x4& c,
                                              f32x4& c2) {
  if constexpr (DUAL) {
    c2 = mfma16bf(A[2], Bf[0], c2);         // chain 2: the small terms
    c = mfma16bf(A[1], Bf[0], c);
    c2 = mfma16bf(A[0], Bf[2], c2);
    c = mfma16bf(A[0], Bf[1], c);
    c2 = mfma16bf(A[1], Bf[1], c2);
    c = mfma16bf(A[0], Bf[0], c);
  } else {
    c = mfma16bf(A[2], Bf[0], c);
    c = mfma16bf(A[0], Bf[2], c);
    c = mfma16bf(A[1], Bf[1], c);
    c = mfma16bf(A[1], Bf[0], c);
    c = mfma16bf(A[0], Bf[1], c);
    c = mfma16bf(A[0], Bf[0], c);
  }
}

template <bool DUAL, int SMAX, bool SC1>
__device__ __forceinline__ void gram16x6p_body(const float* __restrict__ emb, int64_t ld, int B, int T, int P,
                                               float* __restrict__ slabs,
                                               unsigned long long* stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned x6p_lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int Bp = 16 * T;
  const int IMG = 3 * 8 * (Bp + 1);                            // 16-byte units per pass image
  const int k0 = blockIdx.x * (2 * kX6Sub);
  constexpr int RMAX = 13;                                     // Bp * 16 / 256 <= 13 for Bp <= 208
  f32x4 v[2][RMAX];
  const int nq = Bp * 16;
#pragma unroll
  for (int sb = 0; sb < 2; ++sb)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int q = u * 256 + threadIdx.x;                     // no branch around a load:
      int row = q >> 4;                                        // rows past the end re-read B - 1
      row = row < B ? row : B - 1;
      v[sb][u] = *reinterpret_cast<const f32x4*>(emb + (int64_t)row * ld + k0 + sb * kX6Sub + 4 * (q & 15));
    }
  // ---- this wave's range of steps (scalar work under the loads): steps in strip-major order, a
  //      step belongs to the wave whose pair range [P w / 4, P (w + 1) / 4) holds its first pair
  int s_first = 0, tj_first = 0, nsteps = 0;
  {
    const int lo = (int)(((long)P * wid) / 4), hi = (int)(((long)P * (wid + 1)) / 4);
    int c = 0;
    for (int st = 0; 2 * st < T; ++st) {
      const int r0 = 2 * st;
      for (int tj = r0; tj < T; ++tj) {
        if (c >= lo && c < hi) {
          if (nsteps == 0) {
            s_first = st;
            tj_first = tj;
          }
          ++nsteps;
        }
        c += (r0 + 1 < T && tj >= r0 + 1) ? 2 : 1;
      }
    }
  }
  auto split_store = [&](int sb) {
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
      const int q = u * 256 + threadIdx.x;
      if (q < nq) {
        const int row = q >> 4, c4 = q & 15;
        unsigned h[3][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = row < B ? v[sb][u][c] : 0.f;
          split3_bf16x(x, h[0][c], h[1][c], h[2][c]);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          uint2 w2;
          w2.x = h[pl][0] | (h[pl][1] << 16);
          w2.y = h[pl][2] | (h[pl][3] << 16);
          *reinterpret_cast<uint2*>(
              &x6p_lds[((sb * IMG + (pl * 8 + (c4 >> 1)) * (Bp + 1) + row) << 2) + ((c4 & 1) << 1)]) = w2;
        }
      }
    }
  };
  const gx_u32x4* img = reinterpret_cast<const gx_u32x4*>(x6p_lds) + g * (Bp + 1) + i;
  // fragment (pass sb, k-step ks, plane pl, tile t) = img[sb * IMG + (pl * 8 + 4 * ks) * (Bp + 1) + 16 * t]
  auto frag3 = [&](gx_u32x4 (&f)[3], int sb, int ks, int tile) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = img[sb * IMG + (pl * 8 + 4 * ks) * (Bp + 1) + 16 * tile];
  };
  auto next_step = [&](int st, int tj, int& st_n, int& tj_n) {
    st_n = st;
    tj_n = tj + 1;
    if (tj_n == T) {
      st_n = 2 * (st + 1) < T ? st + 1 : st;                   // (past the end: stay, harmless re-read)
      tj_n = 2 * st_n;
    }
  };
  f32x4 acc[2 * SMAX], acc2[DUAL ? 2 * SMAX : 1];
#pragma unroll
  for (int q = 0; q < 2 * SMAX; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < (DUAL ? 2 * SMAX : 1); ++q) acc2[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- pass 0: columns 0..63, k-step outermost, every pair's sums stay in registers
  split_store(0);
  __syncthreads();
  if (SCL_DIAG_ONLY(stamps != nullptr) && threadIdx.x == 0)
    stamps[blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    int st = s_first, tj = tj_first;
    gx_u32x4 a0[3], a1[3], b[3];
    frag3(a0, 0, ks, 2 * st);
    frag3(a1, 0, ks, 2 * st + 1 < T ? 2 * st + 1 : 2 * st);
    frag3(b, 0, ks, tj);
#pragma unroll
    for (int ls = 0; ls < SMAX; ++ls) {
      if (ls < nsteps) {
        int st_n, tj_n;
        next_step(st, tj, st_n, tj_n);
        gx_u32x4 b_n[3];
        frag3(b_n, 0, ks, tj_n);
        __builtin_amdgcn_sched_barrier(0);       // the next step's reads fly under these MFMAs
        x6_pair_mfmas<DUAL>(a0, b, acc[2 * ls], acc2[DUAL ? 2 * ls : 0]);
        if (2 * st + 1 < T && tj >= 2 * st + 1)
          x6_pair_mfmas<DUAL>(a1, b, acc[2 * ls + 1], acc2[DUAL ? 2 * ls + 1 : 0]);
        if (st_n != st) {                        // (wave-uniform: the next strip)
          frag3(a0, 0, ks, 2 * st_n);
          frag3(a1, 0, ks, 2 * st_n + 1 < T ? 2 * st_n + 1 : 2 * st_n);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b[pl] = b_n[pl];
        st = st_n;
        tj = tj_n;
      }
    }
  }
  // ---- pass 1: columns 64..127, step outermost: a step's pairs are final after its MFMAs
  split_store(1);
  __syncthreads();
  if (SCL_DIAG_ONLY(stamps != nullptr) && threadIdx.x == 0)
    stamps[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();
  float* slab = slabs + (int64_t)blockIdx.x * P * 256;
  {
    int st = s_first, tj = tj_first;
    gx_u32x4 a0[2][3], a1[2][3], b[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag3(a0[ks], 1, ks, 2 * st);
      frag3(a1[ks], 1, ks, 2 * st + 1 < T ? 2 * st + 1 : 2 * st);
      frag3(b[ks], 1, ks, tj);
    }
#pragma unroll
    for (int ls = 0; ls < SMAX; ++ls) {
      if (ls < nsteps) {
        int st_n, tj_n;
        next_step(st, tj, st_n, tj_n);
        gx_u32x4 b_n[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) frag3(b_n[ks], 1, ks, tj_n);
        __builtin_amdgcn_sched_barrier(0);
        const bool two = 2 * st + 1 < T && tj >= 2 * st + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          x6_pair_mfmas<DUAL>(a0[ks], b[ks], acc[2 * ls], acc2[DUAL ? 2 * ls : 0]);
          if (two) x6_pair_mfmas<DUAL>(a1[ks], b[ks], acc[2 * ls + 1], acc2[DUAL ? 2 * ls + 1 : 0]);
        }
        {
          const int r0 = 2 * st;
          const int pair = r0 * T - r0 * (r0 - 1) / 2 + (tj - r0);
          const f32x4 out = DUAL ? acc[2 * ls] + acc2[DUAL ? 2 * ls : 0] : acc[2 * ls];
          if (SC1)
            st_sc1_x4(slab + (int64_t)pair * 256 + 4 * lane, out);
          else
            *reinterpret_cast<f32x4*>(slab + (int64_t)pair * 256 + 4 * lane) = out;
        }
        if (two) {
          const int r1 = 2 * st + 1;
          const int pair = r1 * T - r1 * (r1 - 1) / 2 + (tj - r1);
          const f32x4 out = DUAL ? acc[2 * ls + 1] + acc2[DUAL ? 2 * ls + 1 : 0] : acc[2 * ls + 1];
          if (SC1)
            st_sc1_x4(slab + (int64_t)pair * 256 + 4 * lane, out);
          else
            *reinterpret_cast<f32x4*>(slab + (int64_t)pair * 256 + 4 * lane) = out;
        }
        if (st_n != st) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            frag3(a0[ks], 1, ks, 2 * st_n);
            frag3(a1[ks], 1, ks, 2 * st_n + 1 < T ? 2 * st_n + 1 : 2 * st_n);
          }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b[ks][pl] = b_n[ks][pl];
        st = st_n;
        tj = tj_n;
      }
    }
  }
}

template <bool DUAL, int SMAX>
__global__ __launch_bounds__(256) void gram16x6p_kernel(const float* __restrict__ emb, int64_t ld, int B,
                                                        int T, int P, float* __restrict__ slabs) {
  gram16x6p_body<DUAL, SMAX, false>(emb, ld, B, T, P, slabs);
}
#ifdef SCL_DIAG
template <bool DUAL, int SMAX>
__global__ __launch_bounds__(256) void gram16x6_persist_kernel(const float* __restrict__ emb, int64_t ld,
                                                               float* __restrict__ slabs, PersistArgs pa) {
  extern __shared__ __attribute__((aligned(16))) unsigned x6p_lds[];
  PSTAMP(pa, 0);
  gram16x6p_body<DUAL, SMAX, true>(emb, ld, pa.B, pa.T, pa.P, slabs, SCL_DIAG_ONLY(pa.stamps));
  persist_tail(pa, reinterpret_cast<float*>(x6p_lds));
}

// 32 < B <= 64: the exact-float32 Gram of gram16_kernel in front of the same tail.
template <int PWMAX, bool FULL>
__global__ __launch_bounds__(256) void gram16_persist_kernel(const float* __restrict__ emb, int64_t ld,
                                                             int E, int kchunk, int KS, int vec_ok,
                                                             float* __restrict__ slabs, PersistArgs pa) {
  extern __shared__ __attribute__((aligned(16))) float g16_lds[];
  PSTAMP(pa, 0);
  gram16_body<PWMAX, FULL, false, true>(emb, ld, pa.B, E, pa.T, pa.P, kchunk, KS, vec_ok, slabs, FinalArgs{});
  persist_tail(pa, g16_lds);
}
#endif   // SCL_DIAG

struct GramWs {
  float *slabs, *gn, *gc, *rn, *rowloss, *gfull, *gsum;
  unsigned long long* stamps;   // diagnostic build: [256][16], the LAST bytes of the workspace
  size_t total;
};

inline GramWs carve(void* ws, int B, size_t slab_floats) {
  GramWs w;
  char* c = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    float* ptr = (float*)(c + off);
    off += scl_round256(bytes);
    return ptr;
  };
  w.slabs = take(slab_floats * sizeof(float));
  w.gn = take((size_t)B * B * sizeof(float));
  w.gc = take((size_t)B * B * sizeof(float));
  w.rn = take((size_t)B * sizeof(float));
  w.rowloss = take((size_t)B * sizeof(float));
  w.gfull = take((size_t)B * B * sizeof(float));
  const size_t t16 = (size_t)(B + 15) / 16;
  w.gsum = take(t16 * (t16 + 1) / 2 * 256 * sizeof(float));    // summed pair tiles (one-launch forward)
  w.stamps = nullptr;
#ifdef SCL_DIAG
  w.stamps = (unsigned long long*)take(256 * 16 * sizeof(unsigned long long));
#endif
  w.total = off;
  return w;
}

inline size_t slab_floats_for(int B, int E) {
  if (B <= kFastB) {
    const Gram16Plan p = make_plan16(B, E);
    size_t n = (size_t)p.S * p.P * 256;
    if (B > 64 && E % 128 == 0 && (size_t)(E / 128) * p.P * 256 > n) n = (size_t)(E / 128) * p.P * 256;
    return n;
  }
  const GramPlan p = make_plan(B, E);
  return (size_t)p.splits * p.npairs * kTile * kTile;
}

// bf16x6 route: aligned rows and E a multiple of the 128-column slice
inline bool use_x6(int B, int E, int64_t ld, const float* emb) {
  return B > 64 && B <= kFastB && E % 128 == 0 && ld % 4 == 0 && ((uintptr_t)emb % 16) == 0 &&
         scl_variant() != 31;                             // 31: force the float32-MFMA Gram
}
template <int PWMAX, bool DUAL>
void launch_gram16x6(int T, int P, const float* emb, int64_t ld, int B, int E, float* slabs,
                     hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16x6_kernel<PWMAX, 2, DUAL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  });
  const size_t lds = (size_t)3 * 8 * (16 * T + 1) * 16;
  SCL_LAUNCH("gram16x6_kernel", (gram16x6_kernel<PWMAX, 2, DUAL>), dim3(E / 128), dim3(256), lds, st, emb,
             ld, B, E, T, P, slabs);
}

template <int PWMAX, bool FULL = false>
void launch_gram16(const Gram16Plan& p, const float* emb, int64_t ld, int B, int E, int vec_ok,
                   float* slabs, hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16_kernel<PWMAX, FULL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
  });
  size_t lds = (size_t)16 * p.T * (p.kchunk + 4) * sizeof(float);
  const size_t red = p.KS > 1 ? (size_t)p.KS * p.P * 64 * sizeof(f32x4) : 0;
  if (red > lds) lds = red;
  SCL_LAUNCH("gram16_kernel", (gram16_kernel<PWMAX, FULL>), dim3(p.S), dim3(256), lds, st, emb, ld, B, E,
             p.T, p.P, p.kchunk, p.KS, vec_ok, slabs);
}
// B <= 32 with the finish inside: the last workgroup to arrive runs it
template <int PWMAX, bool FULL>
void launch_gram16_fused(const Gram16Plan& p, const float* emb, int64_t ld, int B, int E, int vec_ok,
                         float* slabs, const FinalArgs& fa, hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16_fused_kernel<PWMAX, FULL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
  });
  size_t lds = (size_t)16 * p.T * (p.kchunk + 4) * sizeof(float);
  const size_t red = p.KS > 1 ? (size_t)p.KS * p.P * 64 * sizeof(f32x4) : 0;
  if (red > lds) lds = red;
  if (lds < 24 * 1024) lds = 24 * 1024;                       // final32_body's tables
  if (B > 32 && lds < (size_t)kFinal64Lds) lds = kFinal64Lds; // final64_body's
  SCL_LAUNCH("gram16_fused_kernel", (gram16_fused_kernel<PWMAX, FULL>), dim3(p.S), dim3(1024), lds, st,
             emb, ld, B, E, p.T, p.P, p.kchunk, p.KS, vec_ok, slabs, fa);
}

#ifdef SCL_DIAG
template <bool DUAL, int SMAX>
void launch_x6_persist(const float* emb, int64_t ld, int E, float* slabs, const PersistArgs& pa,
                       hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16x6_persist_kernel<DUAL, SMAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)2 * 3 * 8 * (16 * pa.T + 1) * 16;
  SCL_LAUNCH("gram16x6_persist_kernel", (gram16x6_persist_kernel<DUAL, SMAX>), dim3(E / 128), dim3(256), lds, st,
             emb, ld, slabs, pa);
}
#endif
template <bool DUAL, int SMAX>
void launch_x6p(const float* emb, int64_t ld, int B, int E, int T, int P, float* slabs, hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16x6p_kernel<DUAL, SMAX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)2 * 3 * 8 * (16 * T + 1) * 16;
  SCL_LAUNCH("gram16x6p_kernel", (gram16x6p_kernel<DUAL, SMAX>), dim3(E / 128), dim3(256), lds, st, emb, ld, B, T,
             P, slabs);
}
#ifdef SCL_DIAG
template <int PWMAX, bool FULL>
void launch_gram16_persist(const Gram16Plan& p, const float* emb, int64_t ld, int E, int vec_ok,
                           float* slabs, const PersistArgs& pa, hipStream_t st) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram16_persist_kernel<PWMAX, FULL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024);
  });
  size_t lds = (size_t)16 * p.T * (p.kchunk + 4) * sizeof(float);
  const size_t red = p.KS > 1 ? (size_t)p.KS * p.P * 64 * sizeof(f32x4) : 0;
  if (red > lds) lds = red;
  if (lds < 8 * 1024) lds = 8 * 1024;                         // persist_tail's tables
  SCL_LAUNCH("gram16_persist_kernel", (gram16_persist_kernel<PWMAX, FULL>), dim3(p.S), dim3(256), lds, st,
             emb, ld, E, p.kchunk, p.KS, vec_ok, slabs, pa);
}
#endif
constexpr int kX6pMaxRows = 208;     // both passes of the slice in LDS: 2 * 3 * 8 * (Bp + 1) * 16 B <= 160 KB

}  // namespace

extern "C" size_t scl_gram_loss_workspace_bytes(int B, int E) {
  if (B < 1 || E < 1 || B > kMaxB) return 0;
  return carve(nullptr, B, slab_floats_for(B, E)).total;
}

extern "C" int scl_gram_loss_fwd(const float* emb, int64_t ld_emb, int B, int E, int mask_kind,
                                 const float* distances, int dist_rank3, float d_alpha,
                                 float d_beta, const int64_t* labels, float alpha, float beta,
                                 float lamb, float eps, int ms_mining, int sum_kind,
                                 float* loss_out, float* coef, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return scl_gram_loss_fwd_s(emb, ld_emb, B, E, mask_kind, distances, dist_rank3, d_alpha, d_beta, labels,
                             alpha, beta, lamb, eps, ms_mining, sum_kind, loss_out, coef, workspace,
                             workspace_bytes, nullptr, stream);
}

extern "C" int scl_gram_loss_fwd_s(const float* emb, int64_t ld_emb, int B, int E, int mask_kind,
                                   const float* distances, int dist_rank3, float d_alpha,
                                   float d_beta, const int64_t* labels, float alpha, float beta,
                                   float lamb, float eps, int ms_mining, int sum_kind,
                                   float* loss_out, float* coef, void* workspace,
                                   size_t workspace_bytes, void* sync_words, void* stream) {
  if (!emb || !loss_out || !workspace) return SCL_E_NULL;
  if (sync_words && ((uintptr_t)sync_words % 4)) return SCL_E_SHAPE;
  if (B < 1 || E < 1 || B > kMaxB || ld_emb < E) return SCL_E_SHAPE;
  if (mask_kind < SCL_MASK_WMS_EXP || mask_kind > SCL_MASK_LABELS) return SCL_E_KIND;
  if (sum_kind != SCL_SUM_MS && sum_kind != SCL_SUM_PLAIN) return SCL_E_KIND;
  if (mask_kind == SCL_MASK_LABELS ? !labels : !distances) return SCL_E_NULL;
  if (!scl_aligned256(workspace)) return SCL_E_WORKSPACE;
  GramWs w = carve(workspace, B, slab_floats_for(B, E));
  if (workspace_bytes < w.total) return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int vec_ok = (ld_emb % 4 == 0) && ((uintptr_t)emb % 16 == 0);
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
  if (B <= kFastB) {
    Gram16Plan p = make_plan16(B, E);
    const int ps = 4 / p.KS;
    const int pw = (p.P + ps - 1) / ps;
    const bool full = p.P % ps == 0;
    // ---- round 6.  Product path for 64 < B <= 208: the strip-scheduled Gram kernel (gram16x6p_kernel)
    // in front of the three finishing launches.  The ONE-launch persistent form (persist_tail) was
    // built, is bit-identical and deadlock-free, and is SLOWER (B = 192: 52 us against 35; stamps in
    // profiles/r06/loss_one_launch_persistent.txt: three grid barriers at 6-10 us each with their
    // imbalance, and phases whose latency chains cost what the small kernels cost): it lives in the
    // diagnostic build only — scl_debug_set_variant(41), + 39 no patience (repair path), 40 stamps.
    // 37 = round 5's four launches (the old Gram kernel).
    const bool x6 = use_x6(B, E, ld_emb, emb);
    const bool x6p = x6 && 16 * p.T <= kX6pMaxRows;
#ifdef SCL_DIAG
    const int grid = x6 ? E / 128 : p.S;
    const int pv = scl_variant();
    if (B > 32 && sync_words && ((uintptr_t)sync_words % 8) == 0 && (x6p || (!x6 && B <= 64)) &&
        grid <= scl_device_cus() && (pv == 39 || pv == 40 || pv == 41)) {
      PersistArgs pa;
      pa.sync = (unsigned*)sync_words;
      pa.distances = distances;
      pa.labels = labels;
      pa.lp = lp;
      pa.slabs = w.slabs;
      pa.gsum = w.gsum;
      pa.gn = w.gn;
      pa.gc = w.gc;
      pa.rn = w.rn;
      pa.rowloss = w.rowloss;
      pa.coef = coef;
      pa.loss_out = loss_out;
      pa.S = grid;
      pa.T = p.T;
      pa.P = p.P;
      pa.B = B;
      pa.spin_limit = scl_variant() == 39 ? 0 : kPersistSpinLimit;   // 39: no patience (the repair path)
      pa.stamps = scl_variant() == 40 ? w.stamps : nullptr;          // 40: clock stamps
      bool launched = true;
      if (x6p) {
        if ((p.P + 3) / 4 <= 20)
          launch_x6_persist<true, 11>(emb, ld_emb, E, w.slabs, pa, st);
        else
          launch_x6_persist<false, 13>(emb, ld_emb, E, w.slabs, pa, st);
      } else if (pw == 3 && full)
        launch_gram16_persist<3, true>(p, emb, ld_emb, E, vec_ok, w.slabs, pa, st);
      else if (pw <= 3)
        launch_gram16_persist<3, false>(p, emb, ld_emb, E, vec_ok, w.slabs, pa, st);
      else if (pw <= 5)
        launch_gram16_persist<5, false>(p, emb, ld_emb, E, vec_ok, w.slabs, pa, st);
      else
        launched = false;
      if (launched) return scl_launch_status();
    }
#endif
    if (x6p && scl_variant() != 37) {
      p.S = E / 128;
      if ((p.P + 3) / 4 <= 20)
        launch_x6p<true, 11>(emb, ld_emb, B, E, p.T, p.P, w.slabs, st);
      else
        launch_x6p<false, 13>(emb, ld_emb, B, E, p.T, p.P, w.slabs, st);
    } else if (use_x6(B, E, ld_emb, emb)) {
      p.S = E / 128;                                          // slabs of the bf16x6 kernel
#ifdef SCL_DIAG                                               // (B <= 208 arrives here under variant 37 only)
      if (pw <= 9)
        launch_gram16x6<9, true>(p.T, p.P, emb, ld_emb, B, E, w.slabs, st);
      else if (pw <= 20)
        launch_gram16x6<20, true>(p.T, p.P, emb, ld_emb, B, E, w.slabs, st);
      else
#endif
        launch_gram16x6<34, false>(p.T, p.P, emb, ld_emb, B, E, w.slabs, st);
    } else if ((B <= 32 || (B <= 64 && scl_variant() == 36)) && sync_words && full &&
               (pw == 1 || pw == 3 || pw == 5) && scl_variant() != 32) {
      // (32: the two-launch forward, for A/B.  36: the one-launch forward for 32 < B <= 64 as well —
      // final64_body; bit-identical to the four launches and SLOWER, so not the product path: in
      // device time 24.8 us against 17.0 at B = 48 — one workgroup reads every slab and walks 48 rows
      // where three small kernels use the chip; profiles/r05/loss_one_launch_above_32.txt)
      // (32: the two-launch forward, for A/B)  one launch: the Gram and, in its last workgroup, the finish
      FinalArgs fa;
      fa.counter = (unsigned*)sync_words;
      fa.distances = distances;
      fa.labels = labels;
      fa.lp = lp;
      fa.coef = coef;
      fa.loss_out = loss_out;
      if (pw == 1)
        launch_gram16_fused<1, true>(p, emb, ld_emb, B, E, vec_ok, w.slabs, fa, st);
      else if (pw == 3)
        launch_gram16_fused<3, true>(p, emb, ld_emb, B, E, vec_ok, w.slabs, fa, st);
      else
        launch_gram16_fused<5, true>(p, emb, ld_emb, B, E, vec_ok, w.slabs, fa, st);
      return scl_launch_status();
    } else if (pw == 1 && full)
      launch_gram16<1, true>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else if (pw == 3 && full)
      launch_gram16<3, true>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else if (pw <= 3)
      launch_gram16<3>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else if (pw <= 5)
      launch_gram16<5>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else if (pw <= 9)
      launch_gram16<9>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else if (pw <= 20)
      launch_gram16<20>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    else
      launch_gram16<34>(p, emb, ld_emb, B, E, vec_ok, w.slabs, st);
    if (B <= 32) {
      SCL_LAUNCH("gram_final32_kernel", gram_final32_kernel, dim3(1), dim3(1024), 0, st,
                 (const float*)w.slabs, p.S, p.T, p.P, B, distances, labels, lp, coef, loss_out);
      return scl_launch_status();
    }
    SCL_LAUNCH("gram_reduce_kernel", gram_reduce_kernel, dim3(p.P), dim3(256), 0, st,
               (const float*)w.slabs, p.S, p.T, p.P, B, w.gfull);
    const dim3 rg((B + 3) / 4);
#define SCL_ROWS(C)                                                                          \
  SCL_LAUNCH("gram_rows_wave_kernel", gram_rows_wave_kernel<C>, rg, dim3(256), 0, st,             \
             (const float*)w.gfull, B, distances, labels, lp, w.gn, w.gc, w.rn, w.rowloss)
    if (B <= 64)
      SCL_ROWS(1);
    else if (B <= 128)
      SCL_ROWS(2);
    else if (B <= 192)
      SCL_ROWS(3);
    else
      SCL_ROWS(4);
#undef SCL_ROWS
  } else {
    const GramPlan p = make_plan(B, E);
    SCL_LAUNCH("gram_partial_kernel", gram_partial_kernel, dim3(p.splits, p.npairs, 1), dim3(256),
               0, st, emb, ld_emb, (int64_t)0, B, E, p.tiles, p.kchunk, vec_ok, w.slabs);
    const int Bp = (B + 63) / 64 * 64;
    const size_t lds_bytes = ((size_t)rows_part_floats(B) + 5 * (size_t)Bp) * sizeof(float);
    SCL_LAUNCH("gram_rows_kernel", gram_rows_kernel, dim3(B), dim3(kRowThreads), lds_bytes, st,
               w.slabs, p.splits, p.npairs, p.tiles, B, distances, labels, lp, w.gn, w.gc, w.rn,
               w.rowloss);
  }
  SCL_LAUNCH("gram_coef_kernel", gram_coef_kernel, dim3(coef ? B : 1), dim3(256), 0, st, w.gn, w.gc, w.rn,
                     w.rowloss, B, coef, loss_out);
  return scl_launch_status();
}

// ---------------------------------------------------------------------------------------
// gram_bwd_planes_kernel (round 4): grad[r, :] = g * sum_j M[row_begin + r, j] E[j, :] for MANY rows
// (a single-process run at B = 192 asks for all of them) with BOTH operands as three bf16 planes
// and the six products M1E1 + M1E2 + M2E1 + M1E3 + M3E1 + M2E2 on v_mfma_f32_16x16x32_bf16 —
// float32-equivalent like gram16x6 (what is dropped is 2^-24 relative), at 2.7 x the rate of the
// float32 MFMA the older kernels use (gram_bwd_fast_kernel<2>: 40 us at B = 192 = 0.39 of that
// pipe's peak); measured 42.5 -> 28.6 us there.
constexpr int GBP_LDM = 40;      // bf16 per row of a k-step's M planes (32 + 8 pad = 80 bytes)

// grid E / 128; block 512; dynamic LDS max(3 (Rp GBP_LDM + 32 GBP_LDE) * 2, Rp * GBP_LDO * 4) bytes,
// Rp = 16 NRT (a multiple of 32).  Workgroup = 128 columns x ALL the rows, k-step by k-step: the M
// rows [Rp][32] of the step and the E slab [32][128] are split into their planes on the way into
// LDS; both are requested into registers one k-step ahead.  Wave w = columns 16 w .. + 15: two
// transposed reads per plane give its E^T fragments, then one ds_read_b128 per plane and row tile
// and six MFMAs (two chains of three) per row tile.  (Versions on the way — 64-column workgroups
// with the whole E tile in LDS, a pre-split image of M from a launch of its own, two k-steps per
// stage: DESIGN.md section 5, Round 4.)
constexpr int GBP_LDE = 144;     // bf16 per row of an E slab plane in LDS (128 + 16 pad = 288 bytes)
constexpr int GBP_LDO = 132;     // floats per staged output row (128 + 4 pad)
template <int NRT>
__global__ __launch_bounds__(512) void gram_bwd_planes_kernel(const float* __restrict__ emb, int64_t ld,
                                                              int B, int Bp, int E,
                                                              const float* __restrict__ coef,
                                                              int row_begin,
                                                              const float* __restrict__ grad_loss,
                                                              int R, float* __restrict__ grad,
                                                              int64_t ldg) {
  constexpr int Rp = 16 * NRT;
  constexpr int NPC = (Rp * 8 + 511) / 512;           // float4 pieces of a k-step's M rows per thread
  extern __shared__ __attribute__((aligned(16))) unsigned short gbp_lds[];
  unsigned short* mp = gbp_lds;                       // [3][Rp][GBP_LDM]
  unsigned short* ep = gbp_lds + 3 * Rp * GBP_LDM;    // [3][32][GBP_LDE]
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ii = lane & 15, g_ = lane >> 4, q_ = (lane >> 2) & 3, p_ = lane & 3;
  const int e0 = blockIdx.x * 128;
  const int KS = Bp >> 5;
  f32x4 pre[NPC];                    // (clang vectors: an array of HIP's uint4 structs went to scratch)
  f32x4 ev[2];
  auto request = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NPC; ++k) {                    // M rows [Rp][32] of the k-step, float4 pieces
      const int idx = threadIdx.x + 512 * k, r = idx >> 3, c4 = idx & 7;
      const int j = 32 * ks + 4 * c4;                  // (B % 4 == 0: a piece is inside or outside)
      const bool in_ = r < R && j < B;
      pre[k] = *reinterpret_cast<const f32x4*>(coef + (int64_t)(row_begin + (in_ ? r : 0)) * B + (in_ ? j : 0));
      if (!in_) pre[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {                      // E slab: 32 rows x 32 float4
      const int idx = threadIdx.x + 512 * k, jj = idx >> 5, c4 = idx & 31;
      const int j = 32 * ks + jj, jc = j < B ? j : B - 1;
      ev[k] = *reinterpret_cast<const f32x4*>(emb + (int64_t)jc * ld + e0 + 4 * c4);
      if (j >= B) ev[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto publish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int idx = threadIdx.x + 512 * k, r = idx >> 3, c4 = idx & 7;
      unsigned h[3][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split3_bf16x(pre[k][c], h[0][c], h[1][c], h[2][c]);
      if (r < Rp) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<uint2*>(mp + (pl * Rp + r) * GBP_LDM + 4 * c4) =
              make_uint2(h[pl][0] | (h[pl][1] << 16), h[pl][2] | (h[pl][3] << 16));
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = threadIdx.x + 512 * k, jj = idx >> 5, c4 = idx & 31;
      unsigned h[3][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) split3_bf16x(ev[k][c], h[0][c], h[1][c], h[2][c]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<uint2*>(ep + (pl * 32 + jj) * GBP_LDE + 4 * c4) =
            make_uint2(h[pl][0] | (h[pl][1] << 16), h[pl][2] | (h[pl][3] << 16));
    }
  };
  request(0);
  f32x4 acc0[NRT], acc1[NRT];
#pragma unroll
  for (int t = 0; t < NRT; ++t) acc0[t] = acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef short gbp_s16x4 __attribute__((ext_vector_type(4)));
  for (int ks = 0; ks < KS; ++ks) {
    __syncthreads();                                   // the previous k-step's fragments are read
    publish();
    __syncthreads();
    if (ks + 1 < KS) request(ks + 1);                  // lands under this k-step's products
    // E^T fragments of the wave's 16 columns: lane (group g, 4 q + p) addresses slab row
    // 8 g + 4 half + q, columns 16 w + 4 p .. + 3; lane (g, i) receives column i of the four rows —
    // k index 8 g + 4 half + q, the natural order of the A fragment's 16 bytes
    gx_u32x4 bfr[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned short* pb = ep + (pl * 32 + 8 * g_ + q_) * GBP_LDE + 16 * wid + 4 * p_;
      const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                     (gbp_s16x4 __attribute__((address_space(3)))*)(pb)));
      const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                     (gbp_s16x4 __attribute__((address_space(3)))*)(pb + 4 * GBP_LDE)));
      bfr[pl] = gx_u32x4{lo.x, lo.y, hi.x, hi.y};
    }
#pragma unroll
    for (int t = 0; t < NRT; ++t) {
      gx_u32x4 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[pl] = *reinterpret_cast<const gx_u32x4*>(mp + (pl * Rp + 16 * t + ii) * GBP_LDM + 8 * g_);
      acc0[t] = mfma16bf(a[1], bfr[1], acc0[t]);       // small terms first, two chains
      acc1[t] = mfma16bf(a[0], bfr[2], acc1[t]);
      acc0[t] = mfma16bf(a[2], bfr[0], acc0[t]);
      acc1[t] = mfma16bf(a[0], bfr[1], acc1[t]);
      acc0[t] = mfma16bf(a[1], bfr[0], acc0[t]);
      acc1[t] = mfma16bf(a[0], bfr[0], acc1[t]);
    }
  }
  // register q of lane (column i, group g) = row 16 t + 4 g + q.  The tile leaves through LDS so
  // that a store instruction writes whole 512-byte row segments.
  const float g = grad_loss ? *grad_loss : 1.0f;
  float* ob = reinterpret_cast<float*>(gbp_lds);
  __syncthreads();                                     // every wave is done with the planes
#pragma unroll
  for (int t = 0; t < NRT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ob[(16 * t + 4 * g_ + q) * GBP_LDO + 16 * wid + ii] = g * (acc0[t][q] + acc1[t][q]);
  __syncthreads();
  for (int idx = threadIdx.x; idx < Rp * 32; idx += 512) {
    const int row = idx >> 5, pc = idx & 31;
    if (row < R)
      *reinterpret_cast<f32x4*>(grad + (int64_t)row * ldg + e0 + 4 * pc) =
          *reinterpret_cast<const f32x4*>(ob + row * GBP_LDO + 4 * pc);
  }
}

extern "C" size_t scl_gram_loss_bwd_workspace_bytes(int B, int row_count) {
  (void)B;
  (void)row_count;
  return 256;                                   // (no workspace is needed any more: kept for the ABI)
}

extern "C" int scl_gram_loss_bwd(const float* emb, int64_t ld_emb, int B, int E, const float* coef,
                                 const float* grad_loss, int row_begin, int row_count,
                                 float* grad_emb, int64_t ld_grad, void* stream) {
  if (!emb || !coef || !grad_emb) return SCL_E_NULL;
  if (B < 1 || E < 1 || ld_emb < E || ld_grad < E) return SCL_E_SHAPE;
  if (row_begin < 0 || row_count < 1 || row_begin + row_count > B || B > kMaxB) return SCL_E_SHAPE;
  if (B <= 32) {
    SCL_LAUNCH("gram_bwd32_kernel", gram_bwd32_kernel, dim3((E + 127) / 128), dim3(256), 0,
               (hipStream_t)stream, emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count,
               grad_emb, ld_grad);
    return scl_launch_status();
  }
  {
    // many rows (>= 128; a single-process run at B = 192) on bf16 planes; one row tile (a
    // data-parallel rank's own rows) stays with gram_bwd_rows_kernel, everything else with the
    // float32-MFMA kernels
    const int Bp = (B + 31) & ~31, Rp = (row_count + 31) & ~31;
    const size_t lds_k = ((size_t)3 * Rp * GBP_LDM + (size_t)3 * 32 * GBP_LDE) * sizeof(unsigned short);
    const size_t lds_o = (size_t)Rp * GBP_LDO * sizeof(float);
    const size_t lds = lds_k > lds_o ? lds_k : lds_o;
    const bool al = (ld_emb % 4 == 0) && ((uintptr_t)emb % 16 == 0) && (ld_grad % 4 == 0) &&
                    ((uintptr_t)grad_emb % 16 == 0) && ((uintptr_t)coef % 16 == 0) && B % 4 == 0;
    if (B > 64 && B <= 256 && row_count >= 96 && E % 128 == 0 && al && lds <= 160 * 1024 &&
        scl_variant() != 32 && scl_variant() != 34) {      // (34: the float32-MFMA kernels, for A/B)
      hipStream_t st = (hipStream_t)stream;
#define SCL_GBP_CASE(N)                                                                          \
  if (Rp / 16 == N) {                                                                            \
    static SclDeviceOnce once##N;                                                               \
    scl_call_once(once##N, [] {                                                                 \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_planes_kernel<N>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);         \
    });                                                                                          \
    SCL_LAUNCH("gram_bwd_planes_kernel", gram_bwd_planes_kernel<N>, dim3(E / 128), dim3(512), lds, \
               st, emb, ld_emb, B, Bp, E, coef, row_begin, grad_loss, row_count, grad_emb,       \
               ld_grad);                                                                         \
    return scl_launch_status();                                                                  \
  }
      SCL_GBP_CASE(6)
      SCL_GBP_CASE(8)
      SCL_GBP_CASE(10)
      SCL_GBP_CASE(12)
      SCL_GBP_CASE(14)
      SCL_GBP_CASE(16)
#undef SCL_GBP_CASE
    }
  }
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kTile * (kMaxB | 1) * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_fast_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kTile * (kMaxB | 1) * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_fast_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kTile * (kMaxB | 1) * sizeof(float)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_fast_kernel<4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kTile * (kMaxB | 1) * sizeof(float)));
  });
  {
    const bool al = (ld_emb % 4 == 0) && (ld_grad % 4 == 0) && ((uintptr_t)emb % 16 == 0) &&
                    ((uintptr_t)grad_emb % 16 == 0);
    const size_t ldsf = (size_t)kTile * (B | 1) * sizeof(float);
    const int rt = (row_count + kTile - 1) / kTile;
    // waves per SIMD (1024 SIMDs) with 32-, 64- or 128-column waves: take the best balanced cut
    // (ties: the wider wave, which re-reads M less).  A rank of a data-parallel run asks for its
    // own 24 rows of 192 (parallel.wms_loss_dp): one row tile — 32-column waves are then the only
    // cut that puts a wave on every SIMD (64-column waves left half the chip idle: 21 us for a
    // 25 MB read).
    if (al && rt == 1 && E % 128 == 0 && B >= 64 && B <= 512 && scl_variant() != 32 &&
        scl_variant() != 33) {                            // (33: the 32-column waves, for A/B)
      static SclDeviceOnce once_rows;
      scl_call_once(once_rows, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_bwd_rows_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
      });
      const size_t lds_rows = ((size_t)((kTile * (B | 1) + 3) & ~3) + 4 * kTile * GBR_LD) * sizeof(float);
      SCL_LAUNCH("gram_bwd_rows_kernel", gram_bwd_rows_kernel, dim3(E / 128), dim3(256), lds_rows,
                 (hipStream_t)stream, emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count, grad_emb,
                 ld_grad);
      return scl_launch_status();
    }
    if (al && E % 512 == 0 && scl_variant() != 32) {
      const long w4 = (long)(E / 128) * rt, w2 = 2 * w4, w1 = 4 * w4;
      const long r4 = (w4 + 1023) / 1024 * 4, r2 = (w2 + 1023) / 1024 * 2, r1 = (w1 + 1023) / 1024;
      if (r1 < r2 && r1 < r4)
        SCL_LAUNCH("gram_bwd_fast_kernel<1>", gram_bwd_fast_kernel<1>, dim3(E / 128, rt), dim3(256), ldsf,
                   (hipStream_t)stream, emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count,
                   grad_emb, ld_grad);
      else if (r2 < r4)
        SCL_LAUNCH("gram_bwd_fast_kernel<2>", gram_bwd_fast_kernel<2>, dim3(E / 256, rt), dim3(256), ldsf,
                   (hipStream_t)stream, emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count,
                   grad_emb, ld_grad);
      else
        SCL_LAUNCH("gram_bwd_fast_kernel<4>", gram_bwd_fast_kernel<4>, dim3(E / 512, rt), dim3(256), ldsf,
                   (hipStream_t)stream, emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count,
                   grad_emb, ld_grad);
      return scl_launch_status();
    }
  }
  const int vec_ok = (ld_emb % 4 == 0) && (ld_grad % 4 == 0) && ((uintptr_t)emb % 16 == 0) &&
                     ((uintptr_t)grad_emb % 16 == 0);
  const size_t lds = (size_t)kTile * (B | 1) * sizeof(float);
  dim3 grid((E + 511) / 512, (row_count + kTile - 1) / kTile);
  SCL_LAUNCH("gram_bwd_kernel", gram_bwd_kernel, grid, dim3(256), lds, (hipStream_t)stream, emb,
             ld_emb, B, E, coef, grad_loss, row_begin, row_count, vec_ok, grad_emb, ld_grad);
  return scl_launch_status();
}

extern "C" int scl_gram_loss_bwd_w(const float* emb, int64_t ld_emb, int B, int E, const float* coef,
                                   const float* grad_loss, int row_begin, int row_count,
                                   float* grad_emb, int64_t ld_grad, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  (void)workspace;                              // (an earlier form pre-split M into it; not needed)
  (void)workspace_bytes;
  return scl_gram_loss_bwd(emb, ld_emb, B, E, coef, grad_loss, row_begin, row_count, grad_emb, ld_grad,
                           stream);
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
