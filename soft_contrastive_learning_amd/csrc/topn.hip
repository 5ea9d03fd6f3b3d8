// Exact Euclidean top-n retrieval on gfx950.
//
// Reference semantics: KDTree(ref_f).query(query_f, k=N, return_distance=True,
// sort_results=True) (evaluation/top-n.py:103-108; train/train.py:1181-1182) — exact
// L2 nearest neighbours in ascending order, float64 inside scikit-learn.  At d=256 a
// tree degenerates to brute force; here the brute force is a fused contraction +
// selection so the Q x R distance matrix (4 GB at 10k x 100k) never exists:
//   refnorm_kernel     ||r||^2 per reference row
//   topn_scan_kernel   per (128-query tile, reference split): score = ||r||^2 - 2 q.r on
//                      v_mfma_f32_32x32x2_f32 (queries resident in registers, the reference
//                      tile in LDS with a register prefetch of the next one); every query
//                      keeps a sorted list of its best 32 and a score under its threshold
//                      is inserted by the whole wave in O(1)
//   topn_rerank_kernel one wave per query: merges the per-split lists by f32 score,
//                      recomputes the best 32 candidates' distances exactly in float64 as
//                      sum (q - r)^2, sorts by (distance, index) and emits the first n
// The f32 pass only nominates candidates (n <= 25 of 32 kept); the emitted
// order and distances are float64-exact.
#include <limits.h>
#include <math.h>

#include <mutex>

#include "scl_common.h"

namespace {

constexpr int KEEP = 32;   // candidates kept per (query, split)
constexpr int QW = 4;      // waves (32-query tiles) per workgroup
constexpr int kMaxN = 25;

// grid ceil(R / 64), block 256: wave w takes rows 64 b + 16 w .. + 15.  rmax_bits (optional,
// zeroed by the caller): max_j ||r_j||^2 as its float bit pattern (non-negative floats order
// like unsigned integers), one atomic per workgroup — the exactness certificate's norm bound.
__global__ __launch_bounds__(256) void refnorm_kernel(const float* __restrict__ ref, int R, int d,
                                                      float* __restrict__ out,
                                                      unsigned* __restrict__ rmax_bits) {
  __shared__ float wmax[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float mx = 0.f;
  for (int t = 0; t < 16; ++t) {
    const int row = blockIdx.x * 64 + wid * 16 + t;
    if (row >= R) break;                         // wave-uniform
    const float* p = ref + (int64_t)row * d;
    float s = 0.f;
    for (int e = lane * 4; e < d; e += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + e);
      s = fmaf(v[0], v[0], s);
      s = fmaf(v[1], v[1], s);
      s = fmaf(v[2], v[2], s);
      s = fmaf(v[3], v[3], s);
    }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
    mx = fmaxf(mx, s);
  }
  if (!rmax_bits) return;
  if (lane == 0) wmax[wid] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(rmax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// x = hi + lo + O(2^-17 |x|): hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even.
__device__ __forceinline__ void split_bf16(float x, unsigned& hi, unsigned& lo) {
  const unsigned short h = f32_to_bf16(x);
  hi = h;
  lo = f32_to_bf16(x - bf16_to_f32(h));
}
// 8 consecutive floats -> 8 packed bf16 high parts and 8 packed low parts
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    unsigned h0, l0, h1, l1;
    split_bf16(a[2 * c], h0, l0);
    split_bf16(a[2 * c + 1], h1, l1);
    hi[c] = h0 | (h1 << 16);
    lo[c] = l0 | (l1 << 16);
    split_bf16(b[2 * c], h0, l0);
    split_bf16(b[2 * c + 1], h1, l1);
    hi[2 + c] = h0 | (h1 << 16);
    lo[2 + c] = l0 | (l1 << 16);
  }
}

// Reference rows as two bf16 planes for the bf16x3 scoring mode (one pass, 8 floats/thread).
__global__ __launch_bounds__(256) void ref_split_kernel(const float* __restrict__ ref,
                                                        int64_t n8, u32x4* __restrict__ hi,
                                                        u32x4* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 a = *reinterpret_cast<const f32x4*>(ref + 8 * i);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ref + 8 * i + 4);
  u32x4 h, l;
  split8(a, b, h, l);
  hi[i] = h;
  lo[i] = l;
}

// LDS-DMA (global_load_lds_dwordx4): one wave instruction copies 64 x 16 bytes from a wave-uniform base
// + per-lane byte offsets straight into 1 KB of consecutive LDS — no staging registers, no ds_write pass.
// Inline asm so that hipcc does not order it against the reads of the other buffer; the kernel waits
// vmcnt(0) itself before the barrier that hands the buffer over (csrc/conv64.hip has the same helper).
__device__ __forceinline__ void tn_glds16(const void* base, unsigned off_bytes, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off_bytes), "s"(base), "s"(lds_byte)
      : "memory");
}
__device__ __forceinline__ unsigned tn_lds_byte_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// Rank-count sort of one query's KEEP-entry list by one wave (entry e = lane & 31):
// afterwards the list is ascending by (score, index).
__device__ __forceinline__ void sort_list(float* sc, int* ix, int lane) {
  const int e = lane & (KEEP - 1);
  const float ms = sc[e];
  const int mi = ix[e];
  int rank = 0;
  for (int o = 0; o < KEEP; ++o) {
    const float os = sc[o];
    const int oi = ix[o];
    rank += (os < ms) || (os == ms && oi < mi);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < KEEP) {
    sc[rank] = ms;
    ix[rank] = mi;
  }
  __builtin_amdgcn_wave_barrier();
}

// grid (ceil(Q/128), splits); block 256.  Dynamic LDS:
//   tile[32][d+4] | refn[32] | lsc[QW][32][KEEP] | lix[QW][32][KEEP] | tau[QW][32]
// Every query keeps a SORTED list of its KEEP best (score, index) pairs.  The first tile of
// a split fills the lists (one rank-count sort each); afterwards a score below the query's
// threshold (its current KEEP-th best) is inserted by the whole wave in O(1): position by
// ballot + popcount, shift by one lane, write back — no per-candidate sort.  Single tile
// buffer + register prefetch (65 KB of LDS: two workgroups per CU hide each other's LDS /
// HBM latencies; their MFMA and VALU work does not overlap on a SIMD).
// dbg (scl_debug_set_variant / 1000, diagnostics only): bit 0 no selection, bit 1 no staging,
// bit 2 no MFMAs — timing ablations, results are then meaningless.
// BF = 1 (bf16x3 scoring): q.r = q_hi.r_hi + q_hi.r_lo + q_lo.r_hi on v_mfma_f32_32x32x16_bf16
// (every product of two 8-bit mantissas is exact in f32; what is dropped is q_lo.r_lo and the
// 2^-17 split residue, |error| <= 1.2e-5 |q||r|).  `ref` then points at the high plane and
// `ref_lo` at the low plane written by ref_split_kernel; the LDS tile holds both planes with
// rows of d/2 + 4 dwords.  The float64 re-rank still reads the float32 rows.
//
// Round 6 — SEL = 1, the THRESHOLD scan: every query comes with a score threshold tau_q that at
// least KEEP references are known to meet (the KEEP-th best score of a pre-pass over every 16th
// reference tile, topn_tau_kernel).  The scan then keeps no lists at all: a score <= tau_q is
// APPENDED to the query's candidate buffer [Q][cap] (one atomic add per (query, tile) that has any;
// ~16 KEEP = 512 of R references qualify per query whatever R is) and the re-rank selects the best
// KEEP of them.  The nominated set is the same as with the lists — the KEEP best scores of the whole
// reference set all lie at or below tau_q — so the certificate and the emitted lists do not change;
// what goes is the sorted insertion (~32 ln(tiles) per query and split at ~240 cycles each: as long
// as the products in the bf16x3 mode) and 64 KB of list LDS per workgroup.  A query whose buffer
// overflows is handed to the exact fallback like an uncertified one.
// SEL = 2, the PRE-PASS that produces tau_q: over a 1 / kTauStride sample of the tiles every lane
// keeps, per query row, the TWO smallest scores of its own reference column (3 vector ops per
// score, no cross-lane work, no lists): 64 scores of 64 DIFFERENT references per (query, split);
// topn_tau_kernel takes the KEEP-th smallest of the splits' union.
// Interleaved splits (refs_per_split == 0; SEL 1 and 2): split s takes the tiles s, s + S, s + 2 S,
// .. (x tile_stride), so that every split sees the whole reference set thinly — references in
// driving order put a query's neighbours in consecutive tiles, which contiguous splits would all
// hand to one workgroup's buffers.
template <int D8, int BF, int SEL = 0>
__global__ __launch_bounds__(256, 2) void topn_scan_kernel(const void* __restrict__ refv,
                                                           const void* __restrict__ ref_lov,
                                                           const float* __restrict__ refnorm,
                                                           int R, const float* __restrict__ query,
                                                           int Q, int refs_per_split, int dbg_arg,
                                                           float* __restrict__ cand_sc,
                                                           int* __restrict__ cand_ix, int tile_stride,
                                                           const float* __restrict__ tau_q,
                                                           int* __restrict__ cand_cnt, int cap,
                                                           int tile_first) {
  const int dbg = SCL_DIAG_ONLY(dbg_arg);      // timing ablations: diagnostic build only
  constexpr int d = D8 * 8;
  constexpr int LD = d + 4;          // f32 tile: floats per row
  constexpr int LDB = d / 2 + 4;     // bf16 planes: dwords per row
  constexpr int KS = d / 16;         // bf16 k-steps
  constexpr int TILE_DW = BF ? 2 * 32 * LDB : 32 * LD;
  const float* ref = reinterpret_cast<const float*>(refv);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // SEL 1 / 2 (no lists in LDS): the tile arrives by LDS-DMA into one of TWO unpadded, XOR-swizzled
  // buffers (16-byte slot s of row r lies at slot s ^ (r & (SLOTS - 1)): the 32 rows a fragment read
  // touches fall on different banks without row padding, which a lane-linear DMA cannot write) — no
  // staging registers (the register-staged form spilled), no store pass, one barrier per tile.  Same
  // box, configs[4]: bf16x3 scan 1585 -> 1455-1475 us, float32 4851-4887 -> 4700-4727 us.  (Two
  // buffers with REGISTER staging had measured slower: profiles/r06/topn_threshold_scan.txt.)
  // SEL 0 keeps the padded single buffer (its lists fill the LDS).
  constexpr bool DMA = SEL != 0;
  constexpr int ROWB = BF ? d * 2 : d * 4;             // bytes per row (per plane)
  constexpr int SLOTS = ROWB / 16;                     // 16-byte slots per row
  constexpr int PLANE_B = 32 * ROWB;                   // bytes per plane image of a tile
  constexpr int DMA_TILE_DW = (BF ? 2 : 1) * PLANE_B / 4;
  constexpr int BUF_DW = (DMA ? DMA_TILE_DW : TILE_DW) + 32;   // + refn[32]
  float* tile = lds;                                   // (SEL 0) TILE_DW | refn
  float* refn = tile + TILE_DW;                        // 32
  float* lsc = lds + (DMA ? 2 : 1) * BUF_DW;           // QW * 32 * KEEP      (SEL 0 only)
  int* lix = reinterpret_cast<int*>(lsc + QW * 32 * KEEP);
  float* tau = SEL == 0 ? reinterpret_cast<float*>(lix + QW * 32 * KEEP) : lsc;   // QW * 32

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q0 = (blockIdx.x * QW + wid) * 32;
  const int split = blockIdx.y;
  // contiguous splits (refs_per_split > 0) or interleaved ones: first tile split * tile_first,
  // then every tile_stride-th
  const int r_begin = refs_per_split > 0 ? split * refs_per_split : split * tile_first * 32;
  int r_end = refs_per_split > 0 ? r_begin + refs_per_split : R;
  if (r_end > R) r_end = R;
  const int ntiles = r_begin < r_end ? ((r_end - r_begin + 31) / 32 + tile_stride - 1) / tile_stride : 0;

  // query fragments.  f32: lane (r, h) keeps q[q0 + r][8t + 4h .. +3] for every t.
  // bf16x3: q[q0 + r][16u + 8h .. +7] split into packed high / low parts for every k-step u.
  f32x4 qf[BF ? 1 : D8];
  u32x4 qh[BF ? KS : 1], ql[BF ? KS : 1];
  {
    const int qrow = q0 + r;
    const float* qp = query + (int64_t)(qrow < Q ? qrow : 0) * d;
    if constexpr (BF) {
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * u + 8 * h);
        f32x4 b = *reinterpret_cast<const f32x4*>(qp + 16 * u + 8 * h + 4);
        if (qrow >= Q) a = b = f32x4{0.f, 0.f, 0.f, 0.f};
        split8(a, b, qh[u], ql[u]);
      }
    } else {
#pragma unroll
      for (int t = 0; t < D8; ++t) {
        qf[t] = *reinterpret_cast<const f32x4*>(qp + 4 * h + 8 * t);
        if (qrow >= Q) qf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  float* my_sc = lsc + wid * 32 * KEEP;
  int* my_ix = lix + wid * 32 * KEEP;
  float* my_tau = tau + wid * 32;

  // staging: 256 threads move one tile as 16-byte pieces.  f32: [32][d] floats; bf16x3: the
  // two [32][d] bf16 planes (same byte count).
  constexpr int V4 = (32 * d / 4 + 255) / 256;
  constexpr int ROW16 = BF ? d / 8 : d / 4;            // 16-byte pieces per row (per plane)
  f32x4 stage[V4];
  float stage_n = 0.f;
  auto stage_load = [&](int t) {
    const int rb = r_begin + t * 32 * tile_stride;
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int prow = idx / ROW16, c = idx % ROW16;   // prow: row, or plane * 32 + row
      const int row = BF ? (prow & 31) : prow;
      const int rr = rb + row;
      const float* src;
      if constexpr (BF)
        src = reinterpret_cast<const float*>(prow < 32 ? refv : ref_lov) +
              ((int64_t)rr * d) / 2 + c * 4;
      else
        src = ref + (int64_t)rr * d + c * 4;
      const bool ok = idx < 32 * d / 4 && rr < r_end;
      stage[v] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (threadIdx.x < 32) {
      const int rr = rb + threadIdx.x;
      stage_n = rr < r_end ? refnorm[rr] : INFINITY;  // padding rows never qualify
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int prow = idx / ROW16, c = idx % ROW16;
      if (idx < 32 * d / 4)
        *reinterpret_cast<f32x4*>(&tile[prow * (BF ? LDB : LD) + c * 4]) = stage[v];
    }
    if (threadIdx.x < 32) refn[threadIdx.x] = stage_n;
  };

  // DMA path: chunk c of a plane image = its bytes [1024 c, 1024 c + 1024); lane l owns byte 1024 c +
  // 16 l = (row, physical slot) and fetches the row's LOGICAL slot phys ^ (row & (SLOTS - 1)).  Rows
  // past the end of the split re-read its last row (their norm is +inf: they never qualify).
  constexpr int NCH = PLANE_B / 1024, NCHT = (BF ? 2 : 1) * NCH;      // chunks per plane / per tile
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  auto dma_issue = [&](int t) {
    const int rb = r_begin + t * 32 * tile_stride;
    const int last = r_end - 1 - rb < 31 ? r_end - 1 - rb : 31;       // last valid row of the tile
    const unsigned dst0 = tn_lds_byte_of(lds) + (unsigned)(t & 1) * (BUF_DW * 4);
#pragma unroll
    for (int k = 0; k < (NCHT + QW - 1) / QW; ++k) {
      const int c = wid_s + QW * k;                                   // (wave-uniform)
      if (c < NCHT) {
        const int pl = c / NCH, cc = c % NCH;
        const int o = 1024 * cc + 16 * lane;
        int row = o / ROWB;
        const int phys = (o % ROWB) / 16;
        const int logical = phys ^ (row & (SLOTS - 1));
        row = row < last ? row : last;
        const char* base = reinterpret_cast<const char*>(BF && pl ? ref_lov : refv) + (int64_t)rb * ROWB;
        tn_glds16(base, (unsigned)(row * ROWB + logical * 16), dst0 + (unsigned)(pl * PLANE_B + 1024 * cc));
      }
    }
    if (threadIdx.x < 32) {
      const int rr = rb + threadIdx.x;
      stage_n = rr < r_end ? refnorm[rr] : INFINITY;
    }
  };
  auto dma_publish = [&](int t) {      // own chunks landed; the norms of tile t into its buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x < 32) lds[(t & 1) * BUF_DW + DMA_TILE_DW + threadIdx.x] = stage_n;
  };
  if (ntiles > 0) {
    if constexpr (DMA) {
      dma_issue(0);
      dma_publish(0);
    } else {
      stage_load(0);
      stage_store();
    }
  }
  __syncthreads();
  float tq[SEL == 0 ? 16 : 1];
#pragma unroll
  for (int q = 0; q < (SEL == 0 ? 16 : 1); ++q) tq[q] = INFINITY;
  // SEL 1: the rows' thresholds live in LDS (16 more registers would spill), their candidate
  // counters in registers: row acc_row(q, h) belongs to this half-wave alone — no atomics
  int cntv[SEL == 1 ? 16 : 1];
  float m1[SEL == 2 ? 16 : 1], m2[SEL == 2 ? 16 : 1];
  if constexpr (SEL == 1) {
    if (lane < 32) my_tau[lane] = q0 + lane < Q ? tau_q[q0 + lane] : -INFINITY;   // (rows past the end: nothing)
#pragma unroll
    for (int q = 0; q < 16; ++q) cntv[q] = 0;
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (SEL == 2) {
#pragma unroll
    for (int q = 0; q < 16; ++q) m1[q] = m2[q] = INFINITY;
  }

  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles && !(dbg & 2)) {
      if constexpr (DMA) dma_issue(t + 1); else stage_load(t + 1);
    }
    f32x16 acc = zero16();
    // DMA buffers: row r starts at byte r * ROWB; the lane's slot 2 u + h lies at (2 u) ^ cm
    const char* dbuf = reinterpret_cast<const char*>(lds) + (t & 1) * (BUF_DW * 4);
    const int cm = h ^ (r & (SLOTS - 1));
    if constexpr (BF) {
      // planes: hi rows at tile[row * LDB], lo rows at tile[(32 + row) * LDB]; lane (r, h) reads
      // the 16 bytes of reference row r holding elements 16u + 8h .. +7
      const float* bh = &tile[r * LDB + 4 * h];
      const float* bl = bh + 32 * LDB;
      const char* dh = dbuf + r * ROWB;
      auto rd_h = [&](int u) {
        return DMA ? *reinterpret_cast<const u32x4*>(dh + 16 * ((2 * u) ^ cm))
                   : *reinterpret_cast<const u32x4*>(bh + 8 * u);
      };
      auto rd_l = [&](int u) {
        return DMA ? *reinterpret_cast<const u32x4*>(dh + PLANE_B + 16 * ((2 * u) ^ cm))
                   : *reinterpret_cast<const u32x4*>(bl + 8 * u);
      };
      u32x4 vh[2], vl[2];
      vh[0] = rd_h(0);
      vl[0] = rd_l(0);
      if (!(dbg & 4)) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
          if (u + 1 < KS) {
            vh[(u + 1) & 1] = rd_h(u + 1);
            vl[(u + 1) & 1] = rd_l(u + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc = mfma_bf16(qh[u], vh[u & 1], acc);
          acc = mfma_bf16(qh[u], vl[u & 1], acc);
          acc = mfma_bf16(ql[u], vh[u & 1], acc);
        }
      }
    } else {
      const float* bp = &tile[r * LD + 4 * h];
      const char* dp = dbuf + r * ROWB;
      auto rd = [&](int u) {
        return DMA ? *reinterpret_cast<const f32x4*>(dp + 16 * ((2 * u) ^ cm))
                   : *reinterpret_cast<const f32x4*>(bp + 8 * u);
      };
      // reference fragments run two steps ahead of the MFMAs in a 3-deep register ring
      f32x4 bv[3];
      bv[0] = rd(0);
      if (D8 > 1) bv[1] = rd(1);
      if (!(dbg & 4)) {
#pragma unroll
        for (int u = 0; u < D8; ++u) {
          if (u + 2 < D8) bv[(u + 2) % 3] = rd(u + 2);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc = mfma32(qf[u][c], bv[u % 3][c], acc);
        }
      }
    }
    const float rnj = DMA ? lds[(t & 1) * BUF_DW + DMA_TILE_DW + r] : refn[r];
    const int ridx = r_begin + t * 32 * tile_stride + r;
    if constexpr (!DMA) {
      __syncthreads();                     // every wave is done reading the tile
      if (t + 1 < ntiles && !(dbg & 2)) stage_store();   // refill it under the selection below
    }

    if constexpr (SEL == 1) {
      // threshold scan: half-wave h holds row acc_row(q, h) of register q against the tile's 32
      // references; the qualifying ones take consecutive slots of the row's region [split][cap]
      if (!(dbg & 1)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float sc = rnj - 2.0f * acc[q];
          const bool hit = sc <= my_tau[acc_row(q, h)];
          const unsigned long long m = __ballot(hit);
          if (m) {                                             // (wave-uniform, rare)
            const unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;
            const int slot = cntv[q] + __popc(mh & ((1u << r) - 1u));
            cntv[q] += __popc(mh);
            if (hit && slot < cap) {
              const int qrow = q0 + acc_row(q, h);
              const int64_t o = ((int64_t)qrow * gridDim.y + split) * cap + slot;
              cand_sc[o] = sc;
              cand_ix[o] = ridx;
            }
          }
        }
      }
    } else if constexpr (SEL == 2) {
      // pre-pass: the two smallest scores of this lane's reference column, per row
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float sc = rnj - 2.0f * acc[q];
        const float t2 = fmaxf(m1[q], sc);
        m1[q] = fminf(m1[q], sc);
        m2[q] = fminf(m2[q], t2);
      }
    } else if (t == 0) {
      // first tile of the split: every list takes the 32 scores as they are, then sorts
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = acc_row(q, h);
        my_sc[row * KEEP + r] = rnj - 2.0f * acc[q];
        my_ix[row * KEEP + r] = ridx;
      }
      __builtin_amdgcn_wave_barrier();
      for (int row = 0; row < 32; ++row) sort_list(my_sc + row * KEEP, my_ix + row * KEEP, lane);
      if (lane < 32) my_tau[lane] = my_sc[lane * KEEP + KEEP - 1];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 16; ++q) tq[q] = my_tau[acc_row(q, h)];
    } else if (!(dbg & 1)) {
      // Register q of the two half-waves holds two DIFFERENT queries (rows acc_row(q, 0) and
      // acc_row(q, 1)) against the tile's 32 references, and an insertion needs 32 lanes (one per
      // list entry): each half-wave inserts its own row's candidates, both at once (round 4; one
      // candidate per step for the whole wave before).  Per list the order of insertion —
      // ascending reference index — is unchanged, and so is every list.
      bool any = false;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float sc = rnj - 2.0f * acc[q];
        const unsigned long long m = __ballot(sc < tq[q]);
        unsigned mh = h ? (unsigned)(m >> 32) : (unsigned)m;    // this half's candidates
        if (m) any = true;
        const int row = acc_row(q, h);
        float* ls = my_sc + row * KEEP;
        int* li = my_ix + row * KEEP;
        const int e = lane & (KEEP - 1);
        while (__ballot(mh != 0)) {
          const bool act = mh != 0;
          const int L = act ? __ffs((int)mh) - 1 : 0;
          mh &= mh - 1;
          const float s = __shfl(sc, 32 * h + L, 64);
          const int id = r_begin + t * 32 * tile_stride + L;
          // sorted insertion by the half-wave: its 32 lanes mirror the list entries
          const float cs = ls[e];
          const int ci = li[e];
          const bool before = (cs < s) || (cs == s && ci < id);
          const unsigned long long bb = __ballot(before);
          const int pos = __popc(h ? (unsigned)(bb >> 32) : (unsigned)bb);
          const float ps = __shfl_up(cs, 1, 64);           // (entry 0 never takes its neighbour's)
          const int pi = __shfl_up(ci, 1, 64);
          const float ns = e < pos ? cs : (e == pos ? s : ps);
          const int ni = e < pos ? ci : (e == pos ? id : pi);
          __builtin_amdgcn_wave_barrier();
          if (act && pos < KEEP) {
            ls[e] = ns;
            li[e] = ni;
            if (e == KEEP - 1) my_tau[row] = ns;
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (any) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = my_tau[acc_row(q, h)];
      }
    }
    if constexpr (DMA) {
      if (t + 1 < ntiles && !(dbg & 2)) dma_publish(t + 1);
    }
    __syncthreads();                       // the next tile visible to everyone (DMA: and this one free)
  }

  if constexpr (SEL == 1) {
    // the rows' candidate counts: lane r == 0 of each half holds them
    if (r == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int qrow = q0 + acc_row(q, h);
        if (qrow < Q) cand_cnt[(int64_t)qrow * gridDim.y + split] = cntv[q];
      }
    }
    return;
  }
  if constexpr (SEL == 2) {
    // [Q][splits][64]: the row's two smallest per reference column (+inf where the split had no tile)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int qrow = q0 + acc_row(q, h);
      if (qrow < Q) {
        float* o = cand_sc + ((int64_t)qrow * gridDim.y + split) * 64;
        o[r] = m1[q];
        o[32 + r] = m2[q];
      }
    }
    return;
  }
  // hand-off: [Q][splits][KEEP], already sorted; +inf scores mark empty slots
  for (int row = 0; row < 32; ++row) {
    const int qrow = q0 + row;
    if (qrow < Q && lane < KEEP) {
      const int64_t o = ((int64_t)qrow * gridDim.y + split) * KEEP + lane;
      const float s = ntiles > 0 ? my_sc[row * KEEP + lane] : INFINITY;
      cand_sc[o] = s;
      cand_ix[o] = s < INFINITY ? my_ix[row * KEEP + lane] : -1;
    }
  }
}

// tau_q = the KEEP-th smallest of the pre-pass values [Q][S][64] (S <= 8: 8 per lane), one wave per
// query, radix select on the order-preserving integer image of the scores.  Every value is the
// score of a different reference, so at least KEEP references score <= tau_q; +inf entries (no
// tile) lose, and tau_q = +inf when fewer than KEEP values exist (everything is appended then: the
// buffers overflow and the exact pass takes the query).
__global__ __launch_bounds__(256) void topn_tau_kernel(const float* __restrict__ pre, int Q, int S,
                                                       float* __restrict__ tau_out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int qi = blockIdx.x * 4 + wid;
  if (qi >= Q) return;                                     // wave-uniform
  const int M = S * 64;
  unsigned key[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = k * 64 + lane;
    const float v = e < M ? pre[(int64_t)qi * M + e] : INFINITY;
    const unsigned bits = __float_as_uint(v);
    key[k] = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);
  }
  unsigned prefix = 0;
  int need = KEEP;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned hi_mask = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
    int c0 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) c0 += ((key[k] & hi_mask) == (prefix & hi_mask)) && !((key[k] >> bit) & 1u);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) c0 += __shfl_xor(c0, m, 64);
    if (need > c0) {
      need -= c0;
      prefix |= 1u << bit;
    }
  }
  const unsigned bits = (prefix & 0x80000000u) ? (prefix ^ 0x80000000u) : ~prefix;
  if (lane == 0) tau_out[qi] = __uint_as_float(bits);
}

// One WAVE per query (4 queries per workgroup).  Per-wave LDS: sc[M] | ix[M] | best_ix[KEEP]
// | best_d[KEEP] with M = splits * KEEP.  NE = entries per lane (M <= 64 * NE).
//   stage 1  best KEEP of the M per-split candidates by f32 score (rank counting)
//   stage 2  exact float64 sum (q - r)^2 of those KEEP: four 16-lane groups work on four
//            candidates at a time and all loads of a round are issued before any reduction,
//            so the 32 scattered reference rows are fetched with 4-8 rows in flight
//   stage 3  order by (distance, index), emit the first n
// APPEND (round 6): the candidates are the threshold scan's regions [Q][splits][cap] with
// cnt[q][split] entries each, in no particular order; what lies outside the nominated KEEP is
// either a dropped candidate (>= the KEEP-th best score) or was never appended (> tau_q).
template <int NE, bool APPEND = false>
__global__ __launch_bounds__(256) void topn_rerank_kernel(const float* __restrict__ ref,
                                                          const float* __restrict__ query, int Q,
                                                          int d, int splits, int n,
                                                          int64_t idx_offset,
                                                          const float* __restrict__ cand_sc,
                                                          const int* __restrict__ cand_ix,
                                                          int64_t* __restrict__ idx_out,
                                                          double* __restrict__ dist_out,
                                                          const unsigned* __restrict__ rmax_bits,
                                                          float eps_q, float eps_r,
                                                          unsigned char* __restrict__ uncertified,
                                                          double* __restrict__ bound_sq,
                                                          const int* __restrict__ cand_cnt = nullptr,
                                                          const float* __restrict__ tau_q = nullptr,
                                                          int cap = 0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // APPEND: the regions are mostly empty (16 KEEP candidates expected in splits x cap slots): the
  // wave works on the COMPACTED list of at most M = 64 NE entries — position p belongs to the split
  // whose running count covers it (binary search over the splits' prefix sums in LDS), so every
  // lane's NE loads are independent and only occupied slots are read
  const int M = APPEND ? 64 * NE : splits * KEEP;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int per_wave = 2 * M + KEEP + 2 * (KEEP + 1) + 2 + 66;  // floats (best_d: KEEP + 1 doubles, 8-byte aligned)
  float* sc = lds + wid * per_wave;
  int* ix = reinterpret_cast<int*>(sc + M);
  int* best_ix = ix + M;
  double* best_d = reinterpret_cast<double*>(best_ix + KEEP);
  int* pre_off = reinterpret_cast<int*>(best_d + KEEP + 1);     // [65] exclusive prefix sums (APPEND)
  const int qi = blockIdx.x * 4 + wid;
  if (qi >= Q) return;                                   // wave-uniform; no block barriers below

  float es[NE];
  int ei[NE];
  bool overflow = false;
  int total = M;
  if constexpr (APPEND) {                                  // (splits <= 64)
    const int craw = lane < splits ? cand_cnt[(int64_t)qi * splits + lane] : 0;
    overflow = craw > cap;
    const int c = craw < cap ? craw : cap;
    int incl = c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const int o = __shfl_up(incl, m, 64);
      if (lane >= m) incl += o;
    }
    pre_off[lane + 1] = incl;
    if (lane == 0) pre_off[0] = 0;
    total = __shfl(incl, 63, 64);
    overflow = __any(overflow) || total > M;               // (more candidates than the compact list holds)
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const int e = k * 64 + lane;
    bool valid = e < M;
    int64_t src = (int64_t)qi * M + e;
    if constexpr (APPEND) {
      valid = e < total && e < M;
      int lo = 0, hi = splits;                             // largest s with pre_off[s] <= e
#pragma unroll
      for (int it = 0; it < 7; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        if (mid <= splits && pre_off[mid < 64 ? mid : 64] <= e) lo = mid; else hi = mid - 1;
      }
      src = ((int64_t)qi * splits + lo) * cap + (e - pre_off[lo]);
    }
    es[k] = valid ? cand_sc[src] : INFINITY;
    ei[k] = valid ? cand_ix[src] : -1;
    if (e < M) {
      sc[e] = es[k];
      ix[e] = ei[k];
    }
  }
  if (lane < KEEP) best_ix[lane] = -1;
  __builtin_amdgcn_wave_barrier();
  // Stage 1 is a radix select of the KEEP-th smallest score (32 bit-steps of a wave-wide
  // count) instead of rank counting, which costs M^2/64 compares per wave (the dominant
  // cost of this kernel at M = 608).  Keys: order-preserving map f32 -> u32; empty = max.
  unsigned eu[NE];
  int nvalid = 0;
#pragma unroll
  for (int k = 0; k < NE; ++k) {
    const unsigned bits = __float_as_uint(es[k]);
    eu[k] = ei[k] >= 0 ? (bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u)) : 0xffffffffu;
    nvalid += ei[k] >= 0;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) nvalid += __shfl_xor(nvalid, m, 64);
  unsigned tau_key = 0xffffffffu;                         // fewer than KEEP entries: all win
  if (nvalid > KEEP) {
    unsigned prefix = 0;
    int need = KEEP;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned hi_mask = bit == 31 ? 0u : (0xffffffffu << (bit + 1));
      int c0 = 0;
#pragma unroll
      for (int k = 0; k < NE; ++k)
        c0 += ((eu[k] & hi_mask) == (prefix & hi_mask)) && !((eu[k] >> bit) & 1u);
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) c0 += __shfl_xor(c0, m, 64);
      if (need > c0) {
        need -= c0;
        prefix |= 1u << bit;
      }
    }
    tau_key = prefix;                                     // key of the KEEP-th smallest
  }
  // Smallest approximate score any reference OUTSIDE the nominated KEEP can have: the merge
  // threshold when the merge dropped candidates, and the last entry of every full per-split
  // list (a split only drops what scores no better than its KEEP-th best).
  float tau_excl = INFINITY;
  if (uncertified) {
    if (nvalid > KEEP) {
      const unsigned bits = (tau_key & 0x80000000u) ? (tau_key ^ 0x80000000u) : ~tau_key;
      tau_excl = __uint_as_float(bits);
    }
    if constexpr (APPEND) {
      tau_excl = fminf(tau_excl, tau_q[qi]);                // never appended: score > tau_q
    } else {
#pragma unroll
      for (int k = 0; k < NE; ++k)
        if ((lane & (KEEP - 1)) == KEEP - 1 && ei[k] >= 0) tau_excl = fminf(tau_excl, es[k]);
    }
    tau_excl = wave_min(tau_excl);
  }
  {
    const unsigned long long lt = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int k = 0; k < NE; ++k) {                        // strictly below the threshold
      const bool sel = ei[k] >= 0 && eu[k] < tau_key;
      const unsigned long long m = __ballot(sel);
      if (sel) best_ix[base + __popcll(m & lt)] = ei[k];
      base += __popcll(m);
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {                        // ties at the threshold fill the rest
      const bool sel = ei[k] >= 0 && eu[k] == tau_key;
      const unsigned long long m = __ballot(sel);
      const int pos = base + __popcll(m & lt);
      if (sel && pos < KEEP) best_ix[pos] = ei[k];
      base += __popcll(m);
    }
  }
  __builtin_amdgcn_wave_barrier();

  const int grp = lane >> 4, gl = lane & 15;
  const float* qp = query + (int64_t)qi * d;
#pragma unroll 2
  for (int t = 0; t < KEEP / 4; ++t) {
    const int c = 4 * t + grp;
    const int ri = best_ix[c];
    double s = 0.0;
    if (ri >= 0) {
      const float* rp = ref + (int64_t)ri * d;
      for (int e = 4 * gl; e < d; e += 64) {
        const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + e);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + e);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const double df = (double)qv[cc] - (double)rv[cc];
          s = fma(df, df, s);
        }
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
    if (gl == 0) best_d[c] = ri >= 0 ? s : INFINITY;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < KEEP) {
    const double md = best_d[lane];
    const int mi = best_ix[lane];
    if (mi >= 0) {
      int rk = 0;
      for (int o = 0; o < KEEP; ++o) {
        const int oi = best_ix[o];
        const double od = best_d[o];
        rk += oi >= 0 && ((od < md) || (od == md && oi < mi));
      }
      if (rk < n) {
        idx_out[(int64_t)qi * n + rk] = (int64_t)mi + idx_offset;
        dist_out[(int64_t)qi * n + rk] = sqrt(md);
      }
      if (rk == n - 1) best_d[KEEP] = md;                 // exact n-th smallest of the nominated
    }
  }
  if (!uncertified) return;
  // Exactness certificate.  Every reference outside the nominated set has approximate score
  // >= tau_excl, so its exact squared distance is >= tau_excl + |q|^2 - eps, eps bounding
  // |approximate - exact score| (rounding of the f32 / bf16x3 contraction and of ||r||^2, as
  // eps_q |q| Rmax + eps_r Rmax^2 with Rmax = max_j ||r_j||).  If the n-th exact distance of
  // the nominated set is strictly below that, no outsider can enter (or tie with) the top n.
  double qq = 0.0;
  for (int e = lane; e < d; e += 64) {
    const double v = (double)qp[e];
    qq = fma(v, v, qq);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) qq += __shfl_xor(qq, m, 64);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    const double dn = best_d[KEEP];
    const double rmax = sqrt((double)__uint_as_float(*rmax_bits));
    const double eps = (double)eps_q * sqrt(qq) * rmax + (double)eps_r * rmax * rmax;
    bool ok = !(tau_excl < INFINITY) || dn < (double)tau_excl + qq - eps;
    if (APPEND && overflow) ok = false;                     // a region overflowed: candidates lost
    uncertified[qi] = ok ? 0 : 1;
    bound_sq[qi] = dn;
  }
}

// Exact fallback for the (rare) queries whose certificate failed: float64 sum (q - r)^2 against
// EVERY reference; the ones within the query's bound (the n-th exact distance of its nominated
// set, an upper bound of the true n-th distance) are appended to its candidate list.  The host
// sorts each list by (distance, index).  grid (ceil(R / 1024), nq); block 256: four 16-lane
// groups per wave, one reference row each per round.
__global__ __launch_bounds__(256) void topn_exact_filter_kernel(
    const float* __restrict__ ref, int R, const float* __restrict__ query, int d,
    const int* __restrict__ qlist, const double* __restrict__ bound_sq, int cap,
    int* __restrict__ count, double* __restrict__ cand_d, int* __restrict__ cand_i) {
  const int f = blockIdx.y, qi = qlist[f];
  const double U = bound_sq[qi];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int grp = lane >> 4, gl = lane & 15;
  const float* qp = query + (int64_t)qi * d;
  f32x4 qv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    qv[j] = 4 * gl + 64 * j < d ? *reinterpret_cast<const f32x4*>(qp + 4 * gl + 64 * j)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.x * 1024 + wid * 256;
  for (int t = 0; t < 64; ++t) {
    const int ri = r0 + 4 * t + grp;
    double s = 0.0;
    if (ri < R) {
      const float* rp = ref + (int64_t)ri * d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (4 * gl + 64 * j < d) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(rp + 4 * gl + 64 * j);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const double df = (double)qv[j][cc] - (double)rv[cc];
            s = fma(df, df, s);
          }
        }
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m, 64);
    if (gl == 0 && ri < R && s <= U) {
      const int slot = atomicAdd(&count[f], 1);
      if (slot < cap) {
        cand_d[(int64_t)f * cap + slot] = s;
        cand_i[(int64_t)f * cap + slot] = ri;
      }
    }
  }
}

struct TopnPlan {
  int qtiles, splits, refs_per_split;
};
inline TopnPlan topn_plan(int R, int Q, int bf) {
  TopnPlan p;
  p.qtiles = (Q + 32 * QW - 1) / (32 * QW);
  const int max_splits = (R + 32 * KEEP - 1) / (32 * KEEP);  // keep >= 32 tiles per split
  int best = 1;
  double best_cost = 1e30;
  // <= 32 splits: the re-rank keeps 32 * splits candidates per query in LDS (one wave each)
  for (int s = 1; s <= 32 && s <= (max_splits < 1 ? 1 : max_splits); ++s) {
    const long wgs = (long)p.qtiles * s;
    const long rounds = (wgs + 511) / 512;   // two resident workgroups per CU
    // Cycles per workgroup ~ MFMA time of its tiles + list insertions.  Every split restarts
    // its thresholds at +inf, and a wave inserts ~1024 ln(tiles) candidates per split at
    // ~240 cycles each (measured: 19 splits spent as long inserting as multiplying), so
    // more splits buy parallelism with extra selection work.
    const double tiles = ((double)R / s) / 32.0;
    // per tile: 128 f32 MFMAs x 64 cycles, or 48 bf16 MFMAs x 32 cycles (+ barriers, LDS latency)
    const double tile_cycles = bf ? 3000.0 : 8192.0;
    const double cost =
        (double)rounds * (tile_cycles * tiles + 245760.0 * log(tiles + 1.0)) * (1.0 + 0.002 * s);
    if (cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  const int tv = scl_variant() < 8000 ? scl_variant() : 0;   // larger values: other kernels
  if (tv % 1000 >= 100) {   // tuning override (microbench.py --topn-splits)
    best = tv % 1000 - 100;
    if (best > 32) best = 32;
    if (best > max_splits) best = max_splits;
    if (best < 1) best = 1;
  }
  p.splits = best;
  int per = (R + best - 1) / best;
  per = (per + 31) / 32 * 32;
  p.refs_per_split = per;
  p.splits = (R + per - 1) / per;
  return p;
}

inline size_t scan_lds_bytes(int d, int bf, int sel = 0) {
  const size_t tile = bf ? (size_t)2 * 32 * (d / 2 + 4) : (size_t)32 * (d + 4);
  if (sel)   // two unpadded DMA buffers (+ norms) and the thresholds
    return (2 * ((size_t)(bf ? 2 : 1) * 32 * (bf ? d / 2 : d) + 32) + QW * 32) * sizeof(float);
  return (tile + 32 + (sel ? 0 : (size_t)QW * 32 * KEEP * 2) + QW * 32) * sizeof(float);
}

// The threshold scheme (SEL = 1 scan behind the SEL = 2 pre-pass over every kTauStride-th tile).
constexpr int kTauStride = 16;        // the pre-pass scores 1 / 16 of the references
constexpr int kTauCapSplit = 128;     // candidate slots per (query, split): ~16 KEEP = 512 candidates per query
constexpr int kTauSplits = 32;        // are expected over the 32 interleaved splits (16 each, sd 4) when the
                                      // references come in random order; in DRIVING order a query's
                                      // neighbours are one run of consecutive tiles, of which the strided
                                      // pre-pass sees one in sixteen — counts then vary by a factor of 2-3
constexpr int kTauCompact = 1024;     // candidates the re-rank takes per query (more: the exact pass)
constexpr int kTauMinRefs = 32768;    // below: the sorted-list scan
// (diagnostic build: SCL_TAU_STRIDE = 4 .. 64 samples more or fewer tiles — scripts/topn_tau_stride_ab.py)
inline int tau_stride() {
#ifdef SCL_DIAG
  static const int v = [] {
    const char* e = getenv("SCL_TAU_STRIDE");
    const int s = e ? atoi(e) : kTauStride;
    return s >= 4 && s <= 64 ? s : kTauStride;
  }();
  return v;
#else
  return kTauStride;
#endif
}
struct TauPlan {
  int qtiles, pre_splits, main_splits;
};
inline TauPlan tau_plan(int R, int Q) {
  TauPlan t;
  t.qtiles = (Q + 32 * QW - 1) / (32 * QW);
  const int tiles = (R + 31) / 32;
  const int sampled = (tiles + tau_stride() - 1) / tau_stride();
  // pre-pass: interleaved splits (<= 8: topn_tau_kernel keeps 8 values per lane) that fill the chip
  // once, each with >= 8 sampled tiles
  int sp = (512 + t.qtiles - 1) / t.qtiles;
  if (sp > 8) sp = 8;
  if (sp > sampled / 8) sp = sampled / 8;
  t.pre_splits = sp < 1 ? 1 : sp;
  t.main_splits = kTauSplits < tiles ? kTauSplits : tiles;
  return t;
}
inline bool use_tau(int R, bool certified, int bf) {
  // Both score modes (configs[4], 100k x 10k x 256, same box: bf16x3 2.78 -> 2.00 ms, float32 5.75 ->
  // 5.00 ms per call; profiles/r06/topn_threshold_scan.txt).
  // 8100: the sorted-list scan, for A/B; 9000 + bits: the same with the timing ablations `bits`
  (void)bf;
  if (scl_variant() == 8100 || scl_variant() / 1000 == 9) return false;
  return certified && R >= kTauMinRefs;
}

template <int D8, int BF, int SEL = 0>
void launch_scan(const TopnPlan& p, const void* ref, const void* ref_lo, const float* refnorm,
                 int R, const float* query, int Q, float* cs, int* ci, hipStream_t st,
                 int tile_stride = 1, const float* tau = nullptr, int* cnt = nullptr, int cap = 0,
                 int tile_first = 0) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topn_scan_kernel<D8, BF, SEL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)scan_lds_bytes(D8 * 8, BF, SEL));
  });
  SCL_LAUNCH(SEL == 1   ? (BF ? "topn_scan_kernel<BF=1,tau>" : "topn_scan_kernel<BF=0,tau>")
             : SEL == 2 ? (BF ? "topn_scan_kernel<BF=1,pre>" : "topn_scan_kernel<BF=0,pre>")
                        : (BF ? "topn_scan_kernel<BF=1>" : "topn_scan_kernel<BF=0>"),
             (topn_scan_kernel<D8, BF, SEL>), dim3(p.qtiles, p.splits), dim3(256),
             scan_lds_bytes(D8 * 8, BF, SEL), st, ref, ref_lo, refnorm, R, query, Q, p.refs_per_split,
             scl_variant() < 8000 ? scl_variant() / 1000 : (scl_variant() / 1000 == 9 ? scl_variant() % 1000 : 0),
             cs, ci, tile_stride, tau, cnt, cap, tile_first);
}
template <int SEL>
void launch_scan_d(int d, int bf, const TopnPlan& p, const void* ref, const void* ref_lo,
                   const float* refnorm, int R, const float* query, int Q, float* cs, int* ci,
                   hipStream_t st, int tile_stride = 1, const float* tau = nullptr, int* cnt = nullptr,
                   int cap = 0, int tile_first = 0) {
#define SCL_SCAN(D8, BF)                                                                                  \
  launch_scan<D8, BF, SEL>(p, ref, ref_lo, refnorm, R, query, Q, cs, ci, st, tile_stride, tau, cnt, cap, \
                           tile_first)
  if (bf) {
    switch (d) {
      case 32: SCL_SCAN(4, 1); break;
      case 64: SCL_SCAN(8, 1); break;
      case 128: SCL_SCAN(16, 1); break;
      default: SCL_SCAN(32, 1); break;
    }
  } else {
    switch (d) {
      case 32: SCL_SCAN(4, 0); break;
      case 64: SCL_SCAN(8, 0); break;
      case 128: SCL_SCAN(16, 0); break;
      default: SCL_SCAN(32, 0); break;
    }
  }
#undef SCL_SCAN
}

inline bool topn_shape_ok(int R, int Q, int d, int n) {
  const bool d_ok = d == 32 || d == 64 || d == 128 || d == 256;
  return R >= 1 && Q >= 1 && d_ok && n >= 1 && n <= kMaxN && n <= R;
}

}  // namespace

extern "C" size_t scl_topn_l2_ex_workspace_bytes(int R, int Q, int d, int n, int flags) {
  if (!topn_shape_ok(R, Q, d, n) || (flags & ~SCL_TOPN_SCORE_BF16X3)) return 0;
  const int bf = flags & SCL_TOPN_SCORE_BF16X3;
  const TopnPlan p = topn_plan(R, Q, bf);
  size_t lists = 2 * scl_round256((size_t)Q * p.splits * KEEP * sizeof(float));
  if (R >= kTauMinRefs) {             // the threshold scheme's buffers (whichever is larger serves both)
    const TauPlan t = tau_plan(R, Q);
    const size_t tau_bytes = scl_round256((size_t)Q * t.pre_splits * 64 * sizeof(float)) +
                             2 * scl_round256((size_t)Q * t.main_splits * kTauCapSplit * sizeof(float)) +
                             scl_round256((size_t)Q * sizeof(float)) +
                             scl_round256((size_t)Q * t.main_splits * sizeof(int));
    if (tau_bytes > lists) lists = tau_bytes;
  }
  return 256 + scl_round256((size_t)R * sizeof(float)) + lists +
         (bf ? 2 * scl_round256((size_t)R * d * sizeof(unsigned short)) : 0);
}

extern "C" size_t scl_topn_l2_workspace_bytes(int R, int Q, int d, int n) {
  return scl_topn_l2_ex_workspace_bytes(R, Q, d, n, 0);
}

extern "C" int scl_topn_l2_cert(const float* ref, int R, const float* query, int Q, int d, int n,
                                int64_t idx_offset, int64_t* idx_out, double* dist_out,
                                unsigned char* uncertified, double* bound_sq, void* workspace,
                                size_t workspace_bytes, int flags, void* stream) {
  if (!ref || !query || !idx_out || !dist_out || !workspace) return SCL_E_NULL;
  if ((uncertified == nullptr) != (bound_sq == nullptr)) return SCL_E_NULL;
  if (!topn_shape_ok(R, Q, d, n)) return SCL_E_SHAPE;
  if (flags & ~SCL_TOPN_SCORE_BF16X3) return SCL_E_KIND;
  if (((uintptr_t)ref % 16) || ((uintptr_t)query % 16)) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) ||
      workspace_bytes < scl_topn_l2_ex_workspace_bytes(R, Q, d, n, flags))
    return SCL_E_WORKSPACE;
  const int bf = flags & SCL_TOPN_SCORE_BF16X3;
  const TopnPlan p = topn_plan(R, Q, bf);
  char* base = (char*)workspace;
  unsigned* rmax_bits = (unsigned*)base;
  base += 256;
  float* refnorm = (float*)base;
  base += scl_round256((size_t)R * sizeof(float));
  hipStream_t st = (hipStream_t)stream;
  const bool tau_mode = use_tau(R, uncertified != nullptr, bf);
  const TauPlan tp = tau_mode ? tau_plan(R, Q) : TauPlan{};
  // sorted-list scan: cs / ci [Q][splits][KEEP].  threshold scheme: pre-pass values [Q][S'][64],
  // candidate regions [Q][S][kTauCapSplit] x 2, tau [Q], counts [Q][S]
  float *cs, *pre = nullptr, *tau = nullptr;
  int *ci, *cnt = nullptr;
  if (tau_mode) {
    pre = (float*)base;
    base += scl_round256((size_t)Q * tp.pre_splits * 64 * sizeof(float));
    cs = (float*)base;
    base += scl_round256((size_t)Q * tp.main_splits * kTauCapSplit * sizeof(float));
    ci = (int*)base;
    base += scl_round256((size_t)Q * tp.main_splits * kTauCapSplit * sizeof(float));
    tau = (float*)base;
    base += scl_round256((size_t)Q * sizeof(float));
    cnt = (int*)base;
    base += scl_round256((size_t)Q * tp.main_splits * sizeof(int));
  } else {
    cs = (float*)base;
    base += scl_round256((size_t)Q * p.splits * KEEP * sizeof(float));
    ci = (int*)base;
    base += scl_round256((size_t)Q * p.splits * KEEP * sizeof(float));
  }
  // (the bf16 planes lie behind whichever of the two layouts is larger)
  {
    const size_t used = (size_t)(base - (char*)workspace);
    const size_t planes = bf ? 2 * scl_round256((size_t)R * d * sizeof(unsigned short)) : 0;
    base = (char*)workspace + (scl_topn_l2_ex_workspace_bytes(R, Q, d, n, flags) - planes);
    if ((size_t)(base - (char*)workspace) < used) return SCL_E_WORKSPACE;
  }
  if (uncertified) {
    const hipError_t e = hipMemsetAsync(rmax_bits, 0, 16, st);
    if (e != hipSuccess) return (int)e;
  }
  SCL_LAUNCH("refnorm_kernel", refnorm_kernel, dim3((R + 63) / 64), dim3(256), 0, st, ref, R, d,
             refnorm, uncertified ? rmax_bits : (unsigned*)nullptr);
  const void* scan_ref = ref;
  const void* scan_lo = nullptr;
  if (bf) {
    u32x4* hi = (u32x4*)base;
    base += scl_round256((size_t)R * d * sizeof(unsigned short));
    u32x4* lo = (u32x4*)base;
    const int64_t n8 = (int64_t)R * d / 8;
    SCL_LAUNCH("ref_split_kernel", ref_split_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256),
               0, st, ref, n8, hi, lo);
    scan_ref = hi;
    scan_lo = lo;
  }
  if (tau_mode) {
    // pre-pass over every kTauStride-th tile (interleaved splits) -> tau_q -> threshold scan
    TopnPlan pp;
    pp.qtiles = tp.qtiles;
    pp.refs_per_split = 0;                                   // interleaved
    pp.splits = tp.pre_splits;
    launch_scan_d<2>(d, bf, pp, scan_ref, scan_lo, refnorm, R, query, Q, pre, (int*)nullptr, st,
                     tau_stride() * tp.pre_splits, nullptr, nullptr, 0, tau_stride());
    SCL_LAUNCH("topn_tau_kernel", topn_tau_kernel, dim3((Q + 3) / 4), dim3(256), 0, st, (const float*)pre, Q,
               tp.pre_splits, tau);
    pp.splits = tp.main_splits;
    launch_scan_d<1>(d, bf, pp, scan_ref, scan_lo, refnorm, R, query, Q, cs, ci, st, tp.main_splits, tau, cnt,
                     kTauCapSplit, 1);
  } else {
    launch_scan_d<0>(d, bf, p, scan_ref, scan_lo, refnorm, R, query, Q, cs, ci, st);
  }
  // |approximate - exact score| <= eps_q |q| Rmax + eps_r Rmax^2 (u = 2^-24; factor 2 of
  // safety on rigorous worst-case bounds):
  //   ||r||^2 by f32 FMAs + the final f32 subtraction        (d + 16) u (Rmax^2 + 2 |q| Rmax)
  //   -2 q.r, f32 MFMA chain of d terms                       2 d u |q| Rmax
  //   -2 q.r, bf16x3: dropped q_lo.r_lo + split residues      2 * 3 * 2^-18 |q| Rmax
  //                   + f32 accumulation of 3 d products      2 * 3 d u |q| Rmax
  const double u = 5.9604644775390625e-8;
  const double eps_r = 2.0 * (d + 16) * u;
  const double eps_q = 2.0 * (2.0 * (d + 16) * u +
                              (bf ? 6.0 / 262144.0 + 6.0 * d * u : 2.0 * d * u));
  const int M = tau_mode ? kTauCompact : p.splits * KEEP;
  const size_t lds = (size_t)4 * (2 * M + 3 * KEEP + 4 + 66) * sizeof(float);
  const dim3 rgrid((Q + 3) / 4);
  if (tau_mode) {
    static SclDeviceOnce once;
    scl_call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topn_rerank_kernel<kTauCompact / 64, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    });
    SCL_LAUNCH("topn_rerank_kernel<append>", (topn_rerank_kernel<kTauCompact / 64, true>), rgrid, dim3(256), lds, st, ref,
               query, Q, d, tp.main_splits, n, idx_offset, (const float*)cs, (const int*)ci, idx_out, dist_out,
               (const unsigned*)rmax_bits, (float)eps_q, (float)eps_r, uncertified, bound_sq,
               (const int*)cnt, (const float*)tau, kTauCapSplit);
    return scl_launch_status();
  }
#define SCL_RERANK(NE)                                                                           \
  SCL_LAUNCH("topn_rerank_kernel", topn_rerank_kernel<NE>, rgrid, dim3(256), lds, st, ref, query, \
             Q, d, p.splits, n, idx_offset, (const float*)cs, (const int*)ci, idx_out, dist_out,  \
             (const unsigned*)rmax_bits, (float)eps_q, (float)eps_r, uncertified, bound_sq)
  if (M <= 256)
    SCL_RERANK(4);
  else if (M <= 512)
    SCL_RERANK(8);
  else
    SCL_RERANK(16);
#undef SCL_RERANK
  return scl_launch_status();
}

extern "C" int scl_topn_l2_ex(const float* ref, int R, const float* query, int Q, int d, int n,
                              int64_t idx_offset, int64_t* idx_out, double* dist_out,
                              void* workspace, size_t workspace_bytes, int flags, void* stream) {
  return scl_topn_l2_cert(ref, R, query, Q, d, n, idx_offset, idx_out, dist_out, nullptr, nullptr,
                          workspace, workspace_bytes, flags, stream);
}

extern "C" int scl_topn_l2(const float* ref, int R, const float* query, int Q, int d, int n,
                           int64_t idx_offset, int64_t* idx_out, double* dist_out, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return scl_topn_l2_ex(ref, R, query, Q, d, n, idx_offset, idx_out, dist_out, workspace,
                        workspace_bytes, 0, stream);
}

extern "C" int scl_topn_exact_filter(const float* ref, int R, const float* query, int d,
                                     const int* qlist, int nq, const double* bound_sq, int cap,
                                     int* count, double* cand_d, int* cand_i, void* stream) {
  if (!ref || !query || !qlist || !bound_sq || !count || !cand_d || !cand_i) return SCL_E_NULL;
  if (R < 1 || nq < 1 || nq > 65535 || cap < 1 || d < 4 || d > 256 || d % 4) return SCL_E_SHAPE;
  if (((uintptr_t)ref % 16) || ((uintptr_t)query % 16)) return SCL_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(count, 0, (size_t)nq * sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  SCL_LAUNCH("topn_exact_filter_kernel", topn_exact_filter_kernel, dim3((R + 1023) / 1024, nq),
             dim3(256), 0, st, ref, R, query, d, qlist, bound_sq, cap, count, cand_d, cand_i);
  return scl_launch_status();
}

// ---------------------------------------------------------------------------------------
// topn_dots_kernel: the inner products q.r of a block of queries with a block of references for
// descriptors of ANY width (round 4: the in-training localisation check runs on the raw
// 32768-wide descriptors, train/train.py:1181-1182; evaluation/top-n.py sweeps d up to 4096).
// The nomination of evaluation/retrieval._topn_wide used the library's float64 GEMM here.
//
// Arithmetic chosen for a WIDTH-INDEPENDENT error bound: inside a chunk of 256 features the
// products run on the float32 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered chain of fused
// multiply-adds, one rounding each), the chunk sums are added in float64.  For any summation
// order of a length-k chain |fl(x.y) - x.y| <= gamma_k |x||y|, gamma_k = k u / (1 - k u), u =
// 2^-24; summed over chunks (Cauchy-Schwarz) the float32 part is <= gamma_256 |q||r| = 1.53e-5
// |q||r| whatever d is, the float64 part 2^-53 (d / 256 + 1) relative.  retrieval._topn_wide turns
// that into its per-query certificate.
// grid (ceil(R / 64), ceil(Q / 64)), block 256: wave w = 32 queries x 32 references of the 64 x 64
// tile; both operand tiles staged through LDS 128 features at a time (rows of 129 floats: the
// 32 lanes of a fragment read hit 32 banks).
constexpr int DT_CH = 128, DT_LD = DT_CH + 1, DT_CHUNK = 256;
constexpr size_t kDotsLds = 2 * (size_t)64 * DT_LD * sizeof(float);

// blockIdx.z = split of the feature axis (whole chunks of 256 each): its partial sums go to
// out[z][Q][R]; the caller adds the splits in float64 (a few dozen queries against a few thousand
// references are 32 tiles: without the split the launch leaves 7/8 of the chip idle).
__global__ __launch_bounds__(256) void topn_dots_kernel(const float* __restrict__ ref, int R,
                                                        const float* __restrict__ qry, int Q, int d,
                                                        int chunks_per_split, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float dt_lds[];
  float* qs = dt_lds;
  float* rs = dt_lds + 64 * DT_LD;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, q0 = blockIdx.y * 64;
  const int qi = (wid >> 1) * 32, rj = (wid & 1) * 32;
  const bool vec = (d & 3) == 0 && (((uintptr_t)ref | (uintptr_t)qry) & 15) == 0;
  double acc64[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc64[k] = 0.0;
  const int d_lo = blockIdx.z * chunks_per_split * DT_CHUNK;
  const int d_hi = d_lo + chunks_per_split * DT_CHUNK < d ? d_lo + chunks_per_split * DT_CHUNK : d;
  out += (int64_t)blockIdx.z * Q * R;
  for (int c0 = d_lo; c0 < d_hi; c0 += DT_CHUNK) {
    f32x16 acc = zero16();
    const int c1 = c0 + DT_CHUNK < d_hi ? c0 + DT_CHUNK : d_hi;
    for (int s0 = c0; s0 < c1; s0 += DT_CH) {
      __syncthreads();
      // stage [64 rows][128 features] of both operands (zeros past the matrix / the chunk)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = (threadIdx.x >> 5) + 8 * it, col = (threadIdx.x & 31) * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float* src = m ? ref : qry;
          const int grow = (m ? r0 : q0) + row, nrows = m ? R : Q;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          if (grow < nrows) {
            const float* p = src + (int64_t)grow * d + s0 + col;
            if (vec && s0 + col + 4 <= c1) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(p);
              v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (s0 + col + e < c1) v[e] = p[e];
            }
          }
          float* dst = (m ? rs : qs) + row * DT_LD + col;
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[e] = v[e];
        }
      }
      __syncthreads();
      const float* qa = qs + (qi + (lane & 31)) * DT_LD + (lane >> 5);
      const float* rb = rs + (rj + (lane & 31)) * DT_LD + (lane >> 5);
#pragma unroll 16
      for (int k = 0; k < DT_CH; k += 2) acc = mfma32(qa[k], rb[k], acc);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc64[k] += (double)acc[k];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int q = q0 + qi + acc_row(k, lane >> 5), r = r0 + rj + (lane & 31);
    if (q < Q && r < R) out[(int64_t)q * R + r] = acc64[k];
  }
}

extern "C" int scl_topn_dots(const float* ref, int R, const float* query, int Q, int d, int splits,
                             double* out, void* stream) {
  if (!ref || !query || !out) return SCL_E_NULL;
  if (R < 1 || Q < 1 || d < 1 || splits < 1 || splits > 65535) return SCL_E_SHAPE;
  const int chunks = (d + DT_CHUNK - 1) / DT_CHUNK;
  const int per = (chunks + splits - 1) / splits;
  if ((int64_t)(splits - 1) * per >= chunks && splits > 1) return SCL_E_SHAPE;   // an empty split
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&topn_dots_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDotsLds);
  });
  SCL_LAUNCH("topn_dots_kernel", topn_dots_kernel, dim3((R + 63) / 64, (Q + 63) / 64, splits), dim3(256),
             kDotsLds, (hipStream_t)stream, ref, R, query, Q, d, per, out);
  return scl_launch_status();
}
