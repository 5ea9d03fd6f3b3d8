// Tuple losses on gfx950: triplet / quadruplet families and log-ratio.
//
// Reference semantics: pointnetvlad_cls.{triplet,lazy_triplet,quadruplet,
// lazy_quadruplet}_loss (external; call sites train/train.py:700-712), the in-tree
// twins evil_triplet_loss / evil_quadruplet_loss / worst_pos_distance
// (model/losses.py:63-73,197-222) and logratio_loss (model/losses.py:125-135).
//
// The reference tiles the anchor P (or N) times and materialises three [T,N,E]
// temporaries; these losses only ever need the S-1 (or S-1+N) squared distances per
// tuple, so the path is HBM-bound on one read of the [T,S,E] rows:
//   1. tuple_sqd_kernel      one workgroup per (tuple, pair): sum (a-b)^2, float4 loads
//   2. tuple_finish_kernel   hinges, min/max over positives, sum/max over negatives,
//                            loss and d loss / d sqd (TF tie conventions)
//      logratio_finish_kernel  the log-ratio variant
//      distance_term_kernel    the (Huber) distance term of distance_{triplet,quadruplet}_loss
//   3. tuple_bwd_kernel      one pass writing grad_q / grad_pos / grad_neg / grad_other
// Fixed-order reductions only: bitwise reproducible.
#include "scl_common.h"

namespace {

struct TupleView {
  const float* q;
  const float* pos;
  const float* neg;
  const float* other;
  int64_t q_ts, pos_ts, neg_ts, other_ts;
};

// grid (P + N + (other ? N : 0), T); block 256.
__global__ __launch_bounds__(256) void tuple_sqd_kernel(TupleView v, int P, int N, int E,
                                                        int vec_ok, float* __restrict__ sqd) {
  __shared__ float scratch[32];
  const int t = blockIdx.y, idx = blockIdx.x;
  const int width = P + 2 * N;
  const float *a, *b;
  int slot;
  if (idx < P) {
    a = v.q + t * v.q_ts;
    b = v.pos + t * v.pos_ts + (int64_t)idx * E;
    slot = idx;
  } else if (idx < P + N) {
    a = v.q + t * v.q_ts;
    b = v.neg + t * v.neg_ts + (int64_t)(idx - P) * E;
    slot = idx;
  } else {
    a = v.other + t * v.other_ts;
    b = v.neg + t * v.neg_ts + (int64_t)(idx - P - N) * E;
    slot = idx;
  }
  float acc = 0.f;
  if (vec_ok) {
    for (int e = threadIdx.x * 4; e < E; e += 256 * 4) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(a + e);
      const f32x4 y = *reinterpret_cast<const f32x4*>(b + e);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = y[c] - x[c];
        acc += d * d;
      }
    }
  } else {
    for (int e = threadIdx.x; e < E; e += 256) {
      const float d = b[e] - a[e];
      acc += d * d;
    }
  }
  acc = block_reduce<0>(acc, scratch);
  if (threadIdx.x == 0) sqd[(int64_t)t * width + slot] = acc;
}

// One thread per tuple, one workgroup.  coef is d loss / d sqd, already divided by T.
__global__ __launch_bounds__(256) void tuple_finish_kernel(int kind, int T, int P, int N, float m1,
                                                           float m2, const float* __restrict__ sqd,
                                                           float* __restrict__ coef,
                                                           float* __restrict__ loss_out) {
  __shared__ float scratch[32];
  const bool pos_max = kind == SCL_TUPLE_EVIL_TRIPLET || kind == SCL_TUPLE_EVIL_QUADRUPLET;
  const bool lazy = kind == SCL_TUPLE_LAZY_TRIPLET || kind == SCL_TUPLE_LAZY_QUADRUPLET;
  const bool quad = kind >= SCL_TUPLE_QUADRUPLET;
  const int width = P + 2 * N;
  const float invT = 1.0f / (float)T;
  float total = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float* s = sqd + (int64_t)t * width;
    float* c = coef + (int64_t)t * width;
    // best (min) / worst (max) positive; reduce_min/max gradients are split evenly over ties
    float ref = s[0];
    for (int p = 1; p < P; ++p) ref = pos_max ? fmaxf(ref, s[p]) : fminf(ref, s[p]);
    int ties = 0;
    for (int p = 0; p < P; ++p) ties += s[p] == ref;
    float cref = 0.f;  // d loss_t / d ref
    float loss_t = 0.f;
    const int nterms = quad ? 2 : 1;
    for (int term = 0; term < nterms; ++term) {
      const float m = term == 0 ? m1 : m2;
      const float* d = s + P + term * N;
      float* cd = c + P + term * N;
      float hmax = -INFINITY, hsum = 0.f;
      for (int n = 0; n < N; ++n) {
        const float hv = fmaxf(m + (ref - d[n]), 0.f);
        hmax = fmaxf(hmax, hv);
        hsum += hv;
      }
      int hties = 0;
      if (lazy)
        for (int n = 0; n < N; ++n) hties += fmaxf(m + (ref - d[n]), 0.f) == hmax;
      for (int n = 0; n < N; ++n) {
        const float x = m + (ref - d[n]);
        const float hv = fmaxf(x, 0.f);
        float w = lazy ? (hv == hmax ? 1.0f / (float)hties : 0.f) : 1.0f;
        w = x >= 0.f ? w : 0.f;  // tf.maximum(x, 0): gradient to x where x >= 0
        cd[n] = -w * invT;
        cref += w;
      }
      loss_t += lazy ? hmax : hsum;
    }
    if (!quad)
      for (int n = 0; n < N; ++n) c[P + N + n] = 0.f;
    for (int p = 0; p < P; ++p) c[p] = s[p] == ref ? cref * invT / (float)ties : 0.f;
    total += loss_t;
  }
  total = block_reduce<0>(total, scratch);
  if (threadIdx.x == 0) *loss_out = total * invT;
}

// Distance-term add-on of distance_triplet_loss / distance_quadruplet_loss
// (model/losses.py:225-307, 664-690).  Runs after tuple_finish_kernel of the plain (lazy)
// triplet: adds lam * mean_{t,p} term(sf, sd) with sf = sqd_pos / f_max, sd = d_dists / d_max
// (term = squared difference, or Huber with delta 1) and, for the quadruplet form,
// mean_t max_n max(m2 + min_p term - sqd_other_neg / f_max, 0), to the loss and their
// derivatives to coef.  One thread per tuple, one workgroup.
__global__ __launch_bounds__(256) void distance_term_kernel(int huber, int quad, int T, int P,
                                                            int N, float m2, float lam,
                                                            float d_max, float f_max,
                                                            const float* __restrict__ d_dists,
                                                            const float* __restrict__ sqd,
                                                            float* __restrict__ coef,
                                                            float* __restrict__ loss_inout) {
  __shared__ float scratch[32];
  const int width = P + 2 * N;
  const float invT = 1.0f / (float)T, invTP = 1.0f / ((float)T * (float)P);
  float total = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float* s = sqd + (int64_t)t * width;
    float* c = coef + (int64_t)t * width;
    float tsum = 0.f, best = INFINITY;
    for (int p = 0; p < P; ++p) {
      const float err = s[p] / f_max - d_dists[(int64_t)t * P + p] / d_max;
      float v, dv;
      if (huber) {
        const float a = fabsf(err), qd = fminf(a, 1.0f);
        v = 0.5f * qd * qd + (a - qd);
        dv = fminf(fmaxf(err, -1.0f), 1.0f);
      } else {
        v = err * err;
        dv = 2.0f * err;
      }
      tsum += v;
      best = fminf(best, v);
      c[p] += lam * dv * invTP / f_max;
    }
    float second = 0.f;
    if (quad) {
      const float* on = s + P + N;
      float* con = c + P + N;
      float hmax = -INFINITY;
      for (int n = 0; n < N; ++n) hmax = fmaxf(hmax, fmaxf(m2 + (best - on[n] / f_max), 0.f));
      int hties = 0;
      for (int n = 0; n < N; ++n) hties += fmaxf(m2 + (best - on[n] / f_max), 0.f) == hmax;
      float cbest = 0.f;
      for (int n = 0; n < N; ++n) {
        const float x = m2 + (best - on[n] / f_max);
        float w = fmaxf(x, 0.f) == hmax ? 1.0f / (float)hties : 0.f;
        w = x >= 0.f ? w : 0.f;
        con[n] += -w * invT / f_max;
        cbest += w;
      }
      // reduce_min over positives: gradient split evenly over ties
      int ties = 0;
      for (int p = 0; p < P; ++p) {
        const float err = s[p] / f_max - d_dists[(int64_t)t * P + p] / d_max;
        const float a = fabsf(err), qd = fminf(a, 1.0f);
        ties += (huber ? 0.5f * qd * qd + (a - qd) : err * err) == best;
      }
      for (int p = 0; p < P; ++p) {
        const float err = s[p] / f_max - d_dists[(int64_t)t * P + p] / d_max;
        const float a = fabsf(err), qd = fminf(a, 1.0f);
        const float v = huber ? 0.5f * qd * qd + (a - qd) : err * err;
        const float dv = huber ? fminf(fmaxf(err, -1.0f), 1.0f) : 2.0f * err;
        if (v == best) c[p] += cbest * invT * dv / ((float)ties * f_max);
      }
      second = hmax;
    }
    total += lam * tsum * invTP + second * invT;
  }
  total = block_reduce<0>(total, scratch);
  if (threadIdx.x == 0) *loss_inout += total;
}

// logratio_loss, T == 1, P == N.  Single workgroup.
//   loss = mean_{i<N, j<P} (log(pr_j / nr_i) - log(spd_i / snd_i))^2
__global__ __launch_bounds__(256) void logratio_finish_kernel(int P, int N,
                                                              const float* __restrict__ sq_pos_d,
                                                              const float* __restrict__ sq_neg_d,
                                                              const float* __restrict__ sqd,
                                                              float* __restrict__ coef,
                                                              float* __restrict__ loss_out) {
  __shared__ float scratch[32];
  const float* pr = sqd;
  const float* nr = sqd + P;
  const float inv = 1.0f / ((float)N * (float)P);
  float total = 0.f;
  for (int idx = threadIdx.x; idx < N * P; idx += 256) {
    const int i = idx / P, j = idx % P;
    const float diff = logf(pr[j] / nr[i]) - logf(sq_pos_d[i] / sq_neg_d[i]);
    total += diff * diff;
  }
  total = block_reduce<0>(total, scratch);
  if (threadIdx.x == 0) *loss_out = total * inv;
  // d/d pr_j = 2/(NP) * sum_i diff_ij / pr_j ;  d/d nr_i = -2/(NP) * sum_j diff_ij / nr_i
  for (int j = threadIdx.x; j < P; j += 256) {
    float a = 0.f;
    for (int i = 0; i < N; ++i) a += logf(pr[j] / nr[i]) - logf(sq_pos_d[i] / sq_neg_d[i]);
    coef[j] = 2.0f * inv * a / pr[j];
  }
  for (int i = threadIdx.x; i < N; i += 256) {
    float a = 0.f;
    const float dr = logf(sq_pos_d[i] / sq_neg_d[i]);
    for (int j = 0; j < P; ++j) a += logf(pr[j] / nr[i]) - dr;
    coef[P + i] = -2.0f * inv * a / nr[i];
    coef[P + N + i] = 0.f;
  }
}

struct TupleGrads {
  float* q;
  float* pos;
  float* neg;
  float* other;
};

// sqd_ab = sum (b - a)^2:  d/da = -2 c (b - a),  d/db = 2 c (b - a).
// grid (ceil(E / 256), T); one thread per feature column e of tuple t.
__global__ __launch_bounds__(256) void tuple_bwd_kernel(TupleView v, TupleGrads g, int P, int N,
                                                        int E, const float* __restrict__ coef,
                                                        const float* __restrict__ grad_loss) {
  const int t = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int width = P + 2 * N;
  const float* c = coef + (int64_t)t * width;
  const float gl = 2.0f * (grad_loss ? *grad_loss : 1.0f);
  const float qv = v.q[t * v.q_ts + e];
  const bool has_other = v.other != nullptr;
  const float ov = has_other ? v.other[t * v.other_ts + e] : 0.f;
  float gq = 0.f, go = 0.f;
  for (int p = 0; p < P; ++p) {
    const int64_t off = t * v.pos_ts + (int64_t)p * E + e;
    const float d = gl * c[p] * (v.pos[off] - qv);
    g.pos[off] = d;
    gq -= d;
  }
  for (int n = 0; n < N; ++n) {
    const int64_t off = t * v.neg_ts + (int64_t)n * E + e;
    const float nv = v.neg[off];
    const float d1 = gl * c[P + n] * (nv - qv);
    float d2 = 0.f;
    if (has_other) d2 = gl * c[P + N + n] * (nv - ov);
    g.neg[off] = d1 + d2;
    gq -= d1;
    go -= d2;
  }
  g.q[t * v.q_ts + e] = gq;
  if (has_other && g.other) g.other[t * v.other_ts + e] = go;
}

inline bool vec4_ok(const float* p, int64_t ts, int E) {
  return p == nullptr || (((uintptr_t)p % 16 == 0) && (ts % 4 == 0) && (E % 4 == 0));
}

}  // namespace

extern "C" int scl_tuple_loss_fwd(int kind, const float* q, int64_t q_tstride, const float* pos,
                                  int64_t pos_tstride, const float* neg, int64_t neg_tstride,
                                  const float* other, int64_t other_tstride, int T, int P, int N,
                                  int E, float m1, float m2, float* loss_out, float* sqd,
                                  float* coef, void* stream) {
  if (kind < SCL_TUPLE_TRIPLET || kind > SCL_TUPLE_EVIL_QUADRUPLET) return SCL_E_KIND;
  const bool quad = kind >= SCL_TUPLE_QUADRUPLET;
  if (!q || !pos || !neg || !loss_out || !sqd || !coef || (quad && !other)) return SCL_E_NULL;
  if (T < 1 || P < 1 || N < 1 || E < 1 || T > 65535) return SCL_E_SHAPE;
  TupleView v{q, pos, neg, quad ? other : nullptr, q_tstride, pos_tstride, neg_tstride,
              other_tstride};
  const int vec = vec4_ok(q, q_tstride, E) && vec4_ok(pos, pos_tstride, E) &&
                  vec4_ok(neg, neg_tstride, E) && vec4_ok(v.other, other_tstride, E);
  hipStream_t st = (hipStream_t)stream;
  SCL_LAUNCH("tuple_sqd_kernel", tuple_sqd_kernel, dim3(P + N + (quad ? N : 0), T), dim3(256), 0, st, v, P, N,
                     E, vec, sqd);
  SCL_LAUNCH("tuple_finish_kernel", tuple_finish_kernel, dim3(1), dim3(256), 0, st, kind, T, P, N, m1, m2,
                     (const float*)sqd, coef, loss_out);
  return scl_launch_status();
}

extern "C" int scl_distance_tuple_loss_fwd(int kind, int quad, int huber, const float* q,
                                           int64_t q_tstride, const float* pos,
                                           int64_t pos_tstride, const float* neg,
                                           int64_t neg_tstride, const float* other,
                                           int64_t other_tstride, int T, int P, int N, int E,
                                           float m1, float m2, float lam,
                                           const float* sq_d_dists, float d_max_squared,
                                           float f_max_squared, float* loss_out, float* sqd,
                                           float* coef, void* stream) {
  if (kind != SCL_TUPLE_TRIPLET && kind != SCL_TUPLE_LAZY_TRIPLET) return SCL_E_KIND;
  if (!q || !pos || !neg || !sq_d_dists || !loss_out || !sqd || !coef || (quad && !other))
    return SCL_E_NULL;
  if (T < 1 || P < 1 || N < 1 || E < 1 || T > 65535) return SCL_E_SHAPE;
  if (!(d_max_squared > 0.f) || !(f_max_squared > 0.f)) return SCL_E_SHAPE;
  TupleView v{q, pos, neg, quad ? other : nullptr, q_tstride, pos_tstride, neg_tstride,
              other_tstride};
  const int vec = vec4_ok(q, q_tstride, E) && vec4_ok(pos, pos_tstride, E) &&
                  vec4_ok(neg, neg_tstride, E) && vec4_ok(v.other, other_tstride, E);
  hipStream_t st = (hipStream_t)stream;
  SCL_LAUNCH("tuple_sqd_kernel", tuple_sqd_kernel, dim3(P + N + (quad ? N : 0), T), dim3(256), 0,
             st, v, P, N, E, vec, sqd);
  SCL_LAUNCH("tuple_finish_kernel", tuple_finish_kernel, dim3(1), dim3(256), 0, st, kind, T, P, N,
             m1, 0.0f, (const float*)sqd, coef, loss_out);
  SCL_LAUNCH("distance_term_kernel", distance_term_kernel, dim3(1), dim3(256), 0, st, huber ? 1 : 0,
             quad ? 1 : 0, T, P, N, m2, lam, d_max_squared, f_max_squared, sq_d_dists,
             (const float*)sqd, coef, loss_out);
  return scl_launch_status();
}

extern "C" int scl_tuple_loss_bwd(const float* q, int64_t q_tstride, const float* pos,
                                  int64_t pos_tstride, const float* neg, int64_t neg_tstride,
                                  const float* other, int64_t other_tstride, int T, int P, int N,
                                  int E, const float* coef, const float* grad_loss, float* grad_q,
                                  float* grad_pos, float* grad_neg, float* grad_other,
                                  void* stream) {
  if (!q || !pos || !neg || !coef || !grad_q || !grad_pos || !grad_neg) return SCL_E_NULL;
  if (T < 1 || P < 1 || N < 1 || E < 1 || T > 65535) return SCL_E_SHAPE;
  TupleView v{q, pos, neg, other, q_tstride, pos_tstride, neg_tstride, other_tstride};
  TupleGrads g{grad_q, grad_pos, grad_neg, grad_other};
  SCL_LAUNCH("tuple_bwd_kernel", tuple_bwd_kernel, dim3((E + 255) / 256, T), dim3(256), 0,
                     (hipStream_t)stream, v, g, P, N, E, coef, grad_loss);
  return scl_launch_status();
}

extern "C" int scl_logratio_fwd(const float* a, const float* pos, const float* neg, int P, int N,
                                int E, const float* sq_pos_d, const float* sq_neg_d,
                                float* loss_out, float* sqd, float* coef, void* stream) {
  if (!a || !pos || !neg || !sq_pos_d || !sq_neg_d || !loss_out || !sqd || !coef)
    return SCL_E_NULL;
  // the reference's literal broadcasting (model/losses.py:130-133) needs P == N
  if (P < 1 || N < 1 || P != N || E < 1) return SCL_E_SHAPE;
  TupleView v{a, pos, neg, nullptr, 0, 0, 0, 0};
  const int vec = vec4_ok(a, 0, E) && vec4_ok(pos, 0, E) && vec4_ok(neg, 0, E);
  hipStream_t st = (hipStream_t)stream;
  SCL_LAUNCH("tuple_sqd_kernel", tuple_sqd_kernel, dim3(P + N, 1), dim3(256), 0, st, v, P, N, E, vec, sqd);
  SCL_LAUNCH("logratio_finish_kernel", logratio_finish_kernel, dim3(1), dim3(256), 0, st, P, N, sq_pos_d, sq_neg_d,
                     (const float*)sqd, coef, loss_out);
  return scl_launch_status();
}
