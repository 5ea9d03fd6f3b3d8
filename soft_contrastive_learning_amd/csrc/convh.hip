// The LDS-weights 3x3 convolution of convg.hip on v_mfma_f32_16x16x32_bf16.
//
// Same decomposition (workgroup = [12 or 8 rows x 40 cols] of output pixels x 128 output
// channels, eight waves, K loop over (32-channel chunk, tap) in groups of three, windows and
// weights by LDS-DMA, persistent workgroups that prefetch the next tile's first stage under the
// epilogue), rebuilt around what the timing ablations (scripts/conv_ab.py --variants) showed:
// with every DMA removed the loop runs at the rate of a bare LDS-fed MFMA loop (1.56 of 1.68
// PFLOP/s), and what the real kernel loses on top of that is the DMA stream (L2 -> LDS runs at
// 11-13 bytes per cycle and CU with every CU streaming: 43 KB per group against 3,072 MFMA
// cycles) and the barrier at the end of every group (the first fragment reads of the next group
// wait behind it with the MFMA pipe idle).  Hence:
//   * 16x16x32: one MFMA k-step = the whole 32-channel chunk of a tap: a pixel fragment of lane
//     (i, g) = the 16 bytes (channels 8 g .. + 7) of pixel i of a 16-pixel m-tile, a weight
//     fragment = 16 bytes of a weight row; the chip also holds a higher clock on this shape
//     (scripts/mfma_shape_bf16.hip);
//   * 64-byte window pixels without padding (37.6 KB per window instead of 47 with the 80-byte
//     pixels of convg.hip): the four 16-byte pieces of pixel (row, col) are stored at piece ^
//     f(row, col), f = ((col >> 2) & 1) | ((row & 1) << 1) — the DMA applies it on the source
//     side, its LDS destination stays lane-linear.  An m-tile is 2 rows x 8 columns, lanes 0-3 /
//     12-15 the upper row, 4-11 the lower; with 42-column windows every ds_read_b128 service
//     group ({0-3, 12-15, 20-27}, ...) then hits 16 different 16-byte slots for every tap and
//     tile column (checked exhaustively);
//   * weights piece-major — unit (piece g * 128 + row) of 16 bytes — so the slot of a weight
//     fragment is the row modulo 16 = the lane's i: conflict-free without padding, 8 KB per step;
//   * THREE weight buffers and the barrier in the MIDDLE of a group: before it a wave waits for
//     its own share of the next group's weights (requested one group earlier), after it the
//     request for the group after next goes out, and the first fragments of the next group are
//     read at the end of the current one — no LDS read ever waits behind a barrier.  The next
//     window travels in two parts after the first two barriers of a chunk;
//   * LDS: 2 x 40 KB windows + 3 x 24 KB weights = 152 KB (12-row blocks).
#include <mutex>

#include "scl_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBW = 40;                      // output block width
constexpr int HWC = HBW + 2;                 // window columns
constexpr int HPIX = 32;                     // bf16 per staged pixel (64 B, pieces swizzled)
constexpr int HCCH = 32;                     // channels per chunk = one MFMA k-step
constexpr int HNB = 128;                     // output channels per workgroup
constexpr int HTPB = 3;                      // (chunk, tap) steps per group
constexpr int HWT = 4 * HNB * 8;             // bf16 per weight step image (8 KB)
constexpr int HTHR = 512;
constexpr int HX = 1;                        // half step of a group behind whose MFMAs the barrier sits
                                             // (odd: its pixel and weight fragments are dead by then)

// BHv = block rows, or 24 = 12 rows cut 2 x 4 (below)
template <int BHv>
struct HCfg {
  static constexpr int BH = BHv == 24 ? 12 : BHv;
  static constexpr int WR = BH + 2;                          // window rows
  static constexpr int SLOTS = WR * HWC * 4;                 // 16-byte slots of a window
  static constexpr int NI = (SLOTS + 511) / 512;             // 1-KB DMA chunks per wave: 5 / 4
  static constexpr int NA = (NI + 1) / 2;                    // ... of them in the first part
  static constexpr int WIN = NI * 8 * 512;                   // bf16 per window buffer
  static constexpr int NMT = BH / 2 * 5;                     // 2 x 8 m-tiles of the block
  // waves (mg, ng): 12 rows 4 x 2 (8 m-tiles x 64 channels each), 8 rows 2 x 4 (10 x 32),
  // 6 rows 4 x 2 (4 x 64: half the accumulators, for maps whose 8- or 12-row blocks fill the
  // chip badly — 30 x 40: 480 blocks of 6 rows = 1.9 rounds of 256 against 1.5 of 8 rows)
  // 24: 12 rows 2 x 4 (15 x 32 — three whole row pairs per wave group: all 30 m-tiles of the
  // block are computed once (4 x 8 = 32 slots recompute two), and m-tile j sits at a compile-time
  // offset from the group's first, see REG in the kernel; a tap runs in three sub-steps of a row
  // pair each)
  static constexpr int MT = BHv == 24 ? 15 : BH == 12 ? 8 : BH == 8 ? 10 : 4;   // m-tiles per wave
  static constexpr int NT = (BHv == 24 || BH == 8) ? 2 : 4;  // 16-channel n-tiles per wave
  static constexpr int SPT = BHv == 24 ? 3 : 2;              // sub-steps per tap
  static constexpr int MH = MT / SPT;                        // m-tiles per sub-step
  static constexpr size_t LDS = (2 * (size_t)WIN + 3 * HTPB * (size_t)HWT) * 2;
};

template <int V>
struct HConst {
  static constexpr int value = V;
};

// pixel of m-tile row i: lanes 0-3 and 12-15 the upper row, 4-11 the lower (see header)
__device__ __forceinline__ int ht_row(int i) { return (i >= 4 && i < 12) ? 1 : 0; }
__device__ __forceinline__ int ht_col(int i) { return i < 4 ? i : i < 8 ? i - 4 : i < 12 ? i - 4 : i - 8; }

__device__ __forceinline__ f32x4 mfma16h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Weights -> [n-block][chunk][tap][piece g][k 128][8 channels] bf16: the LDS image of a step
// (8 KB), three consecutive steps = one group, copied by LDS-DMA as they lie.
__global__ __launch_bounds__(256) void convh_pack_kernel(const void* __restrict__ w, int64_t sk,
                                                         int64_t sc, int64_t sh, int64_t sw,
                                                         int flags, int cin, int kout,
                                                         unsigned short* __restrict__ packed) {
  const int transposed = flags & 1, wf32 = flags & 2;   // SCL_CONV_TRANSPOSED | SCL_W_F32
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)9 * (cin / HCCH) * kout * HCCH;
  if (idx >= total) return;
  const int e = idx & 7, k = (idx >> 3) & 127, g = (idx >> 10) & 3;
  const int64_t rest = idx >> 12;                      // (nb * CC + cc) * 9 + tap
  const int tap = rest % 9;
  const int cc = (rest / 9) % (cin / HCCH), nb = rest / 9 / (cin / HCCH);
  const int kh = tap / 3, kw = tap % 3;
  const int ci = HCCH * cc + 8 * g + e, co = HNB * nb + k;
  int64_t off;
  if (!transposed)
    off = co * sk + ci * sc + kh * sh + kw * sw;
  else
    off = ci * sk + co * sc + (2 - kh) * sh + (2 - kw) * sw;
  packed[idx] = weight_bf16(w, off, wf32);
}

__device__ uint4 h_zero_block[4];                      // never written: zeros

__device__ __forceinline__ void hglds16(const unsigned short* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_byte)
      : "memory");
}
__device__ __forceinline__ unsigned h_lds_byte_of(const unsigned short* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned short*)p;
}

// grid: persistent workgroups over `vblocks` virtual blocks; block 512.  EPI as convg_kernel:
// 0 plain, 1 + bias (+ ReLU), 2 out = conv * [mask > 0], 3 pooled + window index.
//
// EPI 0..2 run the MFMA with the operands swapped (weights as A, pixels as B): accumulator
// register q of lane (p = lane & 15, g' = lane >> 4) is then pixel p, channel 4 g' + q of the
// n-tile — four consecutive channels, i.e. 8 bytes of the channels-last output, stored straight
// from the registers (a 16-pixel store instruction writes 16 whole 32-byte sectors); no LDS
// transpose, no wave barriers.  EPI 3 keeps pixels as A: register q is pixel 4 g' + q, channel
// lane & 15, which puts a pooling window into registers (q, q + 1) of lanes l and l ^ 16.
//
// STAMP (scl_debug_set_variant(53024 + w), <1, 12> only; scripts/convh_stamps.py): wave w of every
// workgroup writes s_memtime stamps of its second and third tile BEHIND the end of `out` (the
// caller allocates 2 x 24 x 8 bytes per workgroup more) — slots: 0 tile start, 1 first stage
// landed + barrier; 2 + 9 (cc & 1) + 3 gi + {0 own share waited for, 1 barrier passed, 2 next
// requests issued} for the last chunk pair of the tile; 20 K loop left, 21 barrier, 22 epilogue
// done.  (Each stamp waits for its own result: the compiler does not know that s_memtime writes
// its registers late, and re-used them — one stamp landed in a pointer.)
// FULL: H and W are multiples of the block — no tile crosses the image's lower or right border, the
// epilogue carries no border code at all (every bench shape).
template <int EPI, int BHv, bool STAMP = false, bool FULL = false>
__global__ __launch_bounds__(HTHR, 1) void convh_kernel(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ packed,
                                                       int B, int H, int W, int cin, int kout,
                                                       unsigned short* __restrict__ out,
                                                       const float* __restrict__ bias, int relu,
                                                       const unsigned short* __restrict__ mask,
                                                       unsigned char* __restrict__ pidx,
                                                       int vblocks) {
  using G = HCfg<BHv>;
  constexpr int BH = G::BH, WR = G::WR, WIN = G::WIN, NMT = G::NMT, MT = G::MT, NT = G::NT;
  constexpr int MH = G::MH, NI = G::NI, NA = G::NA, SPT = G::SPT;
  constexpr bool WEAVE = BHv == 24 || BHv == 6 || BHv == 8;
  constexpr int HS = 9 * SPT;                  // sub-steps per chunk; odd: the roles of af[] flip too
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* win = lds;
  unsigned short* wts = lds + 2 * WIN;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int kb = kout / HNB;
  const int blocks_x = (W + HBW - 1) / HBW, blocks_y = (H + BH - 1) / BH;
  const int CC = cin / HCCH, S = 9 * CC;                 // CC is even (host check)
  const unsigned short* zeros = reinterpret_cast<const unsigned short*>(h_zero_block);
  int nb = 0, b = 0, y0 = 0, x0 = 0;
  // LDS slot 64 (wid + 8 n) + lane of a window = piece `slot & 3` of pixel `slot >> 2`, filled
  // with the pixel's piece (slot & 3) ^ f(row, col).  One packed register per chunk —
  // (row << 16) | (col << 8) | source piece, negative past the window (zeros are fetched: every
  // wave issues the same number of DMAs) — unpacked at every use behind an opaque asm: left to
  // itself the compiler keeps every derived value live across the K loop and spills.
  int rel[NI];
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    const int slot = 64 * (wid + 8 * n) + lane;
    const int pix = slot >> 2, row = pix / HWC, col = pix - row * HWC;
    const int piece = (slot & 3) ^ (((col >> 2) & 1) | ((row & 1) << 1));
    rel[n] = pix >= WR * HWC ? -1 : (row << 16) | (col << 8) | piece;
  }
  int woff[NI];             // element offset at chunk 0; -1: zeros (outside the image / window)
  auto locate = [&](int vb) -> bool {
    const int grp = vb / (8 * kb), rem = vb - grp * 8 * kb;
    const int pblk = grp * 8 + (rem & 7);
    nb = rem >> 3;
    if (pblk >= B * blocks_x * blocks_y) return false;
    b = pblk / (blocks_x * blocks_y);
    const int t2 = pblk % (blocks_x * blocks_y);
    y0 = (t2 / blocks_x) * BH;
    x0 = (t2 % blocks_x) * HBW;
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      int r = rel[n];
      asm volatile("" : "+v"(r));
      const int y = y0 - 1 + (r >> 16), xx = x0 - 1 + ((r >> 8) & 255);
      const bool inimg = r >= 0 && y >= 0 && y < H && xx >= 0 && xx < W;
      woff[n] = inimg ? ((b * H + y) * W + xx) * cin + 8 * (r & 255) : -1;
    }
    return true;
  };
  auto issue_win = [&](int cc, int buf, int n0, int n1) {
    const unsigned base = h_lds_byte_of(win) + buf * WIN * 2;
#pragma unroll
    for (int n = 0; n < NI; ++n)
      if (n >= n0 && n < n1)
        hglds16(woff[n] >= 0 ? x + woff[n] + HCCH * cc : zeros, base + (wid + 8 * n) * 1024);
  };
  auto issue_wts = [&](int grp, int buf) {              // three steps = 24 chunks, 3 per wave
    const unsigned short* src = packed + ((int64_t)nb * S + HTPB * grp) * HWT + lane * 8;
    const unsigned base = h_lds_byte_of(wts) + buf * HTPB * HWT * 2;
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      const int j = wid + 8 * n;
      hglds16(src + j * 512, base + j * 1024);
    }
  };
  auto issue_wts_one = [&](int grp, int buf, int n) {   // one of a wave's three chunks of a group
    const unsigned short* src = packed + ((int64_t)nb * S + HTPB * grp) * HWT + lane * 8;
    const unsigned base = h_lds_byte_of(wts) + buf * HTPB * HWT * 2;
    const int j = wid + 8 * n;
    hglds16(src + j * 512, base + j * 1024);
  };
  auto stage_first = [&]() {                            // a tile's first two groups and window
    issue_wts(0, 0);
    issue_wts(1, 1);
    issue_win(0, 0, 0, NI);
  };
  // wave (mg, ng): m-tiles MT mg .. + MT - 1 (2 x 8 pixels each, 5 per row pair), channels
  // 16 NT ng .. + 16 NT - 1
  const int mg = NT == 4 ? wid >> 1 : wid >> 2, ng = NT == 4 ? wid & 1 : wid & 3;
  // REG: a wave group's m-tiles are whole row pairs — m-tile j sits at a compile-time offset from
  // the group's first one, which rides in the ds_read's offset field (no address add per read)
  constexpr bool REG = MT % 5 == 0;
  constexpr bool ALLT = NMT % MT == 0;         // every m-tile slot of every wave group is a real m-tile
  int aoff[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int mt = MT * mg + j < NMT ? MT * mg + j : NMT - 1;
    aoff[j] = ((2 * (mt / 5)) * HWC + 8 * (mt % 5)) * HPIX;
  }
  // pixel fragment of this lane for tap t = (kh, kw), relative to the m-tile's corner:
  // lane_a + tap_const(t) + 8 * piece(t), piece(t) = g ^ f(row + kh, col + kw) — the nine
  // two-bit pieces packed into one register
  const int lane_a = (ht_row(i) * HWC + ht_col(i)) * HPIX + (REG ? 2 * (MT / 5) * mg * HWC * HPIX : 0);
  int pieces = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = ht_row(i) + t / 3, c = ht_col(i) + t % 3;
    pieces |= (g ^ (((c >> 2) & 1) | ((r & 1) << 1))) << (2 * t);
  }
  const int lane_b = (g * HNB + 16 * NT * ng + i) * 8;

  const int relu_in = relu;
  const int dbg = STAMP ? 0 : SCL_DIAG_ONLY(relu >> 1);   // timing diagnostics (dv 3020 + bits), results meaningless
  relu &= 1;
  const short relu_floor = relu ? (short)0 : (short)-32768;   // packed ReLU: max with 0, or with the least int16
  bool staged = false;
  // (experiment, dv 3040: static priority for the younger half — MI355X_MICROARCH.md, two waves
  // per SIMD, item 4)
  if (!STAMP && (dbg & 4) && wid >= 4) __builtin_amdgcn_s_setprio(1);
  uint64_t stamp[STAMP ? 24 : 1];
  int tcount = 0;
#define HSTAMP(k)                                                                     \
  if (STAMP)                                                                          \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[STAMP ? (k) : 0])::"memory")
  for (int vb = blockIdx.x; vb < vblocks; vb += gridDim.x) {
    HSTAMP(0);
    if (!staged) {
      if (!locate(vb)) continue;
      stage_first();
    }
    staged = false;
    const int nb_t = nb, b_t = b, y0_t = y0, x0_t = x0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[j][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Half step h = 0 .. 17 of a chunk: tap t = h / 2 (group t / 3 -> weight buffer t / 3), m-tile
    // half h % 2.  Pixel fragments alternate between af[0] and af[1] by half step, weight
    // fragments (one set per step) between bf[0] and bf[1] by step; a chunk has nine steps, so
    // the roles of bf[] flip from chunk to chunk (PAR).
    u32x4 af[2][MH], bf[2][NT];
    int tb = 0;            // the lane's fragment offset for the tap in flight (set at its first sub-step)
    auto load_half = [&](const unsigned short* wbase, int h, int par, int apar) {
      const int t = h / SPT, mh = h % SPT;
      if (mh == 0) {
        const unsigned short* wbp = wts + t * HWT + lane_b;    // (t / 3) * HTPB + t % 3 = t
#pragma unroll
        for (int n = 0; n < NT; ++n)
          bf[(t + par) & 1][n] = *reinterpret_cast<const u32x4*>(wbp + 16 * n * 8);
        // once per tap (it was once per sub-step: four vector instructions each)
        // (opaque: the compiler would otherwise keep all 9 x MT address sums in registers)
        int pk = pieces;
        asm volatile("" : "+v"(pk));
        tb = lane_a + ((pk >> (2 * t)) & 3) * 8;
      }
#pragma unroll
      for (int j = 0; j < MH; ++j)
        af[(h + apar) & 1][j] = *reinterpret_cast<const u32x4*>(
            wbase + tb +
            (REG ? (2 * ((MH * mh + j) / 5) * HWC + 8 * ((MH * mh + j) % 5)) * HPIX : aoff[MH * mh + j]) +
            ((t / 3) * HWC + t % 3) * HPIX);
    };
    // Straight-line on purpose (a data-dependent branch in here costs the accumulators their
    // registers): the last chunk of a tile requests, and reads ahead, like every other one — the
    // clamped group / chunk indices make those requests re-fetch what the buffers already hold.
    const int NG = 3 * CC;
    auto chunk = [&](int cc, auto par_c) {
      constexpr int PAR = decltype(par_c)::value;
      constexpr int APAR = PAR * (HS & 1);
      const unsigned short* wcur = win + (cc & 1) * WIN;
      const unsigned short* wnext = win + ((cc + 1) & 1) * WIN;
      const int ccn = cc + 1 < CC ? cc + 1 : cc;
      // request k = 0 .. 5 of the group behind barrier gi: the three weight chunks, then the
      // window chunks of that barrier's part
      auto issue_op = [&](int gi, int k) {
        if (k < 3) {
          const int gn = 3 * cc + gi + 2 < NG ? 3 * cc + gi + 2 : NG - 3 + (gi + 2) % 3;
          issue_wts_one(gn, (gi + 2) % 3, k);
        } else {
          const int n = (gi == 0 ? 0 : NA) + k - 3;
          if (gi == 0 && n < NA) issue_win(ccn, (cc + 1) & 1, n, n + 1);
          if (gi == 1 && n < NI) issue_win(ccn, (cc + 1) & 1, n, n + 1);
        }
      };
#pragma unroll
      for (int h = 0; h < HS; ++h) {
        // the fragments of the next sub-step fly under the MFMAs of this one
        if (h + 1 < HS)
          load_half(wcur, h + 1, PAR, APAR);
        else
          load_half(wnext, 0, PAR ^ 1, APAR ^ (HS & 1));
        __builtin_amdgcn_sched_barrier(0);
        const int t = h / SPT, mh = h % SPT;
        // WEAVE: the requests that follow a barrier go out BETWEEN the products of the next
        // sub-step (one behind every m-tile's products) instead of in one block in front of them
        const bool after_barrier = WEAVE && (h % (3 * SPT) == HX + 1);   // (compile-time once unrolled)
#pragma unroll
        for (int j = 0; j < MH; ++j) {
#pragma unroll
          for (int n = 0; n < NT; ++n)
            acc[MH * mh + j][n] =
                EPI == 3 ? mfma16h(af[(h + APAR) & 1][j], bf[(t + PAR) & 1][n], acc[MH * mh + j][n])
                         : mfma16h(bf[(t + PAR) & 1][n], af[(h + APAR) & 1][j], acc[MH * mh + j][n]);
          if (after_barrier) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = j * 6 / MH; k < (j + 1) * 6 / MH; ++k) issue_op(h / (3 * SPT), k);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (h % (3 * SPT) == HX) {
          __builtin_amdgcn_sched_barrier(0);
          // own share of what the NEXT group reads has landed; the barrier publishes everybody's
          const int gi = h / (3 * SPT);
          if (gi == 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");   // first window part may fly on
          else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          HSTAMP(2 + 9 * PAR + 3 * gi);
          __builtin_amdgcn_s_barrier();
          HSTAMP(2 + 9 * PAR + 3 * gi + 1);
          // ... and everybody has left the previous group: its weight buffer takes the group
          // after next, the other window buffer the next chunk's window (in two parts)
          if (!WEAVE) {
            const int gn = 3 * cc + gi + 2 < NG ? 3 * cc + gi + 2 : NG - 3 + (gi + 2) % 3;
            issue_wts(gn, (gi + 2) % 3);
            if (gi == 0) issue_win(ccn, (cc + 1) & 1, 0, NA);
            if (gi == 1) issue_win(ccn, (cc + 1) & 1, NA, NI);
          }
          HSTAMP(2 + 9 * PAR + 3 * gi + 2);
        }
      }
    };

    if (!(dbg & 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    HSTAMP(1);
    load_half(win, 0, 0, 0);
#pragma unroll 1
    for (int cc = 0; cc < CC; cc += 2) {
      chunk(cc, HConst<0>());
      chunk(cc + 1, HConst<1>());
    }

    // (the tile coordinates go through an opaque asm: otherwise every epilogue address is
    // computed before the K loop and spilled across it)
    int nb_e = nb_t, b_e = b_t, y0_e = y0_t, x0_e = x0_t;
    asm volatile("" : "+s"(nb_e), "+s"(b_e), "+s"(y0_e), "+s"(x0_e));
    // Every wave has left the K loop: window buffer 1 (the last chunk's window; the next tile's
    // first stage goes to buffer 0 and weight buffers 0 / 1) now serves as epilogue scratch.
    HSTAMP(20);
    __builtin_amdgcn_s_barrier();
    HSTAMP(21);
    // Output through a per-wave LDS transpose so that a store instruction writes WHOLE 128-byte
    // lines (the 64 channels of the wave for 8 pixels): measured on the first-layer kernel, whose
    // only problem is its output, 16-byte pieces scattered as quarter lines ran 338-349 us and
    // whole lines 250-259 — the L2 merges partial lines, but not for free.  The transpose is four
    // ds_write_b64 and two ds_read_b128 per m-tile (the swapped-operand accumulators hold four
    // consecutive channels per lane).
    constexpr int SLD = 16 * NT + 8;                       // bf16 per scratch row (pixel)
    constexpr int PCS = 2 * NT;                            // 16-byte pieces per pixel of the wave
    constexpr int RR = 16 * PCS / 64;                      // pieces per lane and m-tile: 2 (or 1)
    // per wave: 16 pixels x SLD bf16 (+ 16 x (16 NT + 16) index bytes for the pooling epilogue);
    // waves 0-3 in window buffer 1, waves 4-7 in weight buffer 2 (the last group's: as dead)
    constexpr int SCRW = 16 * SLD + 8 * (16 * NT + 16);
    static_assert(4 * SCRW <= WIN && 4 * SCRW <= HTPB * HWT, "epilogue scratch must fit");
    unsigned short* scr = wid < 4 ? lds + WIN + wid * SCRW : wts + 2 * HTPB * HWT + (wid - 4) * SCRW;
    const int ch_w = HNB * nb_e + 16 * NT * ng;           // first channel of the wave
    f32x4 bias4[EPI == 1 ? NT : 1];
    float bias1[EPI == 3 ? NT : 1];
    if (EPI == 1) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
        bias4[n] = *reinterpret_cast<const f32x4*>(bias + ch_w + 16 * n + 4 * g);
    }
    if (EPI == 3) {
#pragma unroll
      for (int n = 0; n < NT; ++n) bias1[n] = bias[ch_w + 16 * n + i];
    }
    // piece p = lane + 64 rr of an m-tile: pixel p / PCS (row ht_row, column ht_col), channels
    // 8 (p % PCS) .. + 7 of the wave; element offset from the m-tile's corner pixel:
    int piece_o[RR], piece_px[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int pp = lane + 64 * rr;
      piece_px[rr] = pp / PCS;
      piece_o[rr] = (ht_row(pp / PCS) * W + ht_col(pp / PCS)) * kout + 8 * (pp % PCS);
    }
    // EPI 2: the mask pieces of m-tile j + PD are requested after the stores of m-tile j, the
    // first PD m-tiles up front (more in flight would not fit the register file next to the
    // accumulators); the next tile's first stage goes out once every mask load has been issued
    constexpr int PD = MT / 2;
    u32x4 mk[EPI == 2 ? MT : 1][EPI == 2 ? RR : 1];
    // the tile's corner pixel, this wave's channels (element offset into out / mask); a tile that
    // lies inside the image whole (every tile of the bench shapes) loads and stores without the
    // per-lane border tests
    const int64_t tile_off = (((int64_t)b_e * H + y0_e) * W + x0_e) * kout + ch_w;
    const bool full_tile = FULL ? !(dbg & 2) : (y0_e + BH <= H && x0_e + HBW <= W && !(dbg & 2));
    auto load_mask = [&](int j) {
      const int mt = MT * mg + j;
      const int oy = y0_e + 2 * (mt / 5), ox = x0_e + 8 * (mt % 5);
      const unsigned short* mb = mask + tile_off + ((2 * (mt / 5)) * W + 8 * (mt % 5)) * kout;
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        // (every element is assigned: a conditionally assigned one would stay live around the
        // whole tile loop)
        if (FULL ? (ALLT || mt < NMT) : (full_tile && mt < NMT)) {   // wave-uniform
          mk[EPI == 2 ? j : 0][EPI == 2 ? rr : 0] = *reinterpret_cast<const u32x4*>(mb + piece_o[rr]);
        } else if (FULL) {
          mk[EPI == 2 ? j : 0][EPI == 2 ? rr : 0] = u32x4{0u, 0u, 0u, 0u};
        } else {
          const bool inside = mt < NMT && oy + ht_row(piece_px[rr]) < H && ox + ht_col(piece_px[rr]) < W;
          mk[EPI == 2 ? j : 0][EPI == 2 ? rr : 0] =
              inside ? *reinterpret_cast<const u32x4*>(mb + piece_o[rr]) : u32x4{0u, 0u, 0u, 0u};
        }
      }
    };
    if (EPI == 2) {
#pragma unroll
      for (int j = 0; j < PD; ++j) load_mask(j);
    }

    // next tile's first stage under this tile's epilogue (see convg.hip)
    auto stage_next = [&]() {
      const int vn = vb + gridDim.x;
      if (vn < vblocks && locate(vn)) {
        // (the K loop's last requests target the same buffers)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stage_first();
        staged = true;
      }
    };
    if (EPI != 2) stage_next();

    if (EPI == 3) {
      // register q of lane (c = i, g' = g): pixel 4 g' + q of the m-tile, channel 16 n + c.
      // Registers (0,1) and (2,3) are the two pooling windows of this lane's row quad; the other
      // row of both windows sits in lane l ^ 16 under the same registers.  The pooled values and
      // window indices of FOUR m-tiles (16 pooled pixels x the wave's 64 channels) are gathered
      // in the scratch and leave as whole lines: 128 bytes of bf16 / 64 bytes of indices per
      // pooled pixel.
      const int PH = H / 2, PW = W / 2;
      const bool upper = g == 0 || g == 3;
      const int cb = g < 2 ? 0 : 4;                        // column base of this lane's quad
      unsigned char* scr8 = reinterpret_cast<unsigned char*>(scr + 16 * SLD);   // [16][64 + 16]
      constexpr int SLD8 = 16 * NT + 16;
#pragma unroll
      for (int jb = 0; jb < MT; jb += 4) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int j = jb + jj < MT ? jb + jj : MT - 1;   // (MT = 10: the last group is half full)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {
              const float v0 = acc[j][n][q0], v1 = acc[j][n][q0 + 1];
              const float lm = fmaxf(v0, v1);
              const int li = v0 >= v1 ? 0 : 1;
              const float om = __shfl_xor(lm, 16);
              const int oi = __shfl_xor(li, 16);
              const float mu = upper ? lm : om, ml = upper ? om : lm;
              const int iu = upper ? li : oi, il = upper ? oi : li;
              const float m = fmaxf(mu, ml);
              const int k = mu >= ml ? iu : 2 + il;        // first maximum in raster order
              // pooled pixel of the group of four m-tiles: 4 jj + (cb + q0) / 2
              const int pl = 4 * jj + ((cb + q0) >> 1);
              if (upper) {
                scr[pl * SLD + 16 * n + i] = f32_to_bf16(fmaxf(m + bias1[n], 0.f));
                scr8[pl * SLD8 + 16 * n + i] = (unsigned char)k;
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        // 16 pooled pixels: values 16 x PCS pieces of 16 bytes, indices 16 x NT pieces
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
          const int pp = lane + 64 * rr, pl = pp / PCS, pc = pp % PCS;
          const int mt = MT * mg + jb + (pl >> 2);
          const int py = (y0_e >> 1) + mt / 5, px = (x0_e >> 1) + 4 * (mt % 5) + (pl & 3);
          const u32x4 v = *reinterpret_cast<const u32x4*>(scr + pl * SLD + 8 * pc);
          if (jb + (pl >> 2) < MT && (FULL ? (ALLT || mt < NMT) : (mt < NMT && py < PH && px < PW && !(dbg & 2))))
            *reinterpret_cast<u32x4*>(out + (((int64_t)b_e * PH + py) * PW + px) * kout + ch_w + 8 * pc) = v;
        }
        if (lane < 16 * NT) {
          const int pl = lane / NT, pc = lane % NT;
          const int mt = MT * mg + jb + (pl >> 2);
          const int py = (y0_e >> 1) + mt / 5, px = (x0_e >> 1) + 4 * (mt % 5) + (pl & 3);
          const u32x4 v = *reinterpret_cast<const u32x4*>(scr8 + pl * SLD8 + 16 * pc);
          if (jb + (pl >> 2) < MT && (FULL ? (ALLT || mt < NMT) : (mt < NMT && py < PH && px < PW && !(dbg & 2))))
            *reinterpret_cast<u32x4*>(pidx + (((int64_t)b_e * PH + py) * PW + px) * kout + ch_w + 16 * pc) = v;
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      unsigned short* out_tile = out + tile_off;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int mt = MT * mg + j;
        if (ALLT || mt < NMT) {                            // wave-uniform
          const int oy = y0_e + 2 * (mt / 5), ox = x0_e + 8 * (mt % 5);
          unsigned short* ob = out_tile + ((2 * (mt / 5)) * W + 8 * (mt % 5)) * kout;
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            f32x4 v = acc[j][n];
            if (EPI == 1) v += bias4[n];                   // (packed adds)
            // (scalars, not elements of pk: a bit cast of a vector ELEMENT read element 0 for
            // both halves — hipcc 7.2)
            unsigned p0 = pack2_bf16(v[0], v[1]), p1 = pack2_bf16(v[2], v[3]);
            if (EPI == 1) {
              p0 = max2_i16(p0, relu_floor);               // ReLU on the rounded pairs, or the identity
              p1 = max2_i16(p1, relu_floor);
            }
            const u32x2 pk = {p0, p1};
            *reinterpret_cast<u32x2*>(scr + i * SLD + 16 * n + 4 * g) = pk;
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int rr = 0; rr < RR; ++rr) {
            u32x4 v = *reinterpret_cast<const u32x4*>(scr + piece_px[rr] * SLD + 8 * ((lane + 64 * rr) % PCS));
            if (EPI == 2) v = relu_mask(v, mk[EPI == 2 ? j : 0][EPI == 2 ? rr : 0]);
            if (full_tile) {                               // (wave-uniform: no per-lane test)
              *reinterpret_cast<u32x4*>(ob + piece_o[rr]) = v;
            } else if (!FULL) {
              const bool inside = oy + ht_row(piece_px[rr]) < H && ox + ht_col(piece_px[rr]) < W && !(dbg & 2);
              if (inside) *reinterpret_cast<u32x4*>(ob + piece_o[rr]) = v;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (EPI == 2) {
          __builtin_amdgcn_sched_barrier(0);               // keep the loads where they are
          if (j + PD < MT) load_mask(j + PD);
          if (j + PD == MT - 1) stage_next();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (STAMP) {
      HSTAMP(22);
      if (threadIdx.x == 64 * ((relu_in >> 4) & 7) && (tcount == 1 || tcount == 2)) {
        uint64_t* o = reinterpret_cast<uint64_t*>(out + (int64_t)B * H * W * kout) +
                      (int64_t)(blockIdx.x * 2 + tcount - 1) * 24;
#pragma unroll
        for (int k = 0; k < 23; ++k) o[k] = stamp[STAMP ? k : 0];
      }
      ++tcount;
    }
  }   // persistent loop
#undef HSTAMP
}

int convh_cus() {
  const int n = scl_device_cus();      // per device (scl_common.h)
  return scl_usable_cus(n);
}

}  // namespace

// Same contract as convg_dispatch (convg.hip), which validates the arguments and calls this
// (dv = the diagnostic variant: 3006 / 3008 / 3012 block height, 3099 one tile per workgroup,
// 3100 + g grid of g + 1 groups).  The packed weights fit the workspace
// of scl_convg_workspace_bytes (8 KB per step instead of 10).
int scl_convh_dispatch(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                       int64_t w_stride_h, int64_t w_stride_w, int flags, int B, int H, int W,
                       int cin, int kout, void* out, const float* bias, int relu, const void* mask,
                       void* pidx, void* workspace, int dv, void* stream) {
  static SclDeviceOnce once;
  scl_call_once(once, [] {
#define SCL_CONVH_ATTR(E, BHV)                                                                 \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convh_kernel<E, BHV>),              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)HCfg<BHV>::LDS);  \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convh_kernel<E, BHV, false, true>), \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)HCfg<BHV>::LDS);
    SCL_CONVH_ATTR(0, 12) SCL_CONVH_ATTR(1, 12) SCL_CONVH_ATTR(2, 12) SCL_CONVH_ATTR(3, 12)
    SCL_CONVH_ATTR(0, 24) SCL_CONVH_ATTR(1, 24) SCL_CONVH_ATTR(2, 24) SCL_CONVH_ATTR(3, 24)
    SCL_CONVH_ATTR(0, 8) SCL_CONVH_ATTR(1, 8) SCL_CONVH_ATTR(2, 8) SCL_CONVH_ATTR(3, 8)
    SCL_CONVH_ATTR(0, 6) SCL_CONVH_ATTR(1, 6) SCL_CONVH_ATTR(2, 6) SCL_CONVH_ATTR(3, 6)
#undef SCL_CONVH_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convh_kernel<1, 12, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HCfg<12>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convh_kernel<1, 24, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)HCfg<24>::LDS);
  });
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* packed = (const unsigned short*)workspace;
  if (flags & SCL_W_PACKED) {
    packed = (const unsigned short*)w;                 // scl_conv_pack_batch wrote it
  } else {
    const int64_t total = (int64_t)9 * (cin / HCCH) * kout * HCCH;
    SCL_LAUNCH("convh_pack_kernel", convh_pack_kernel, dim3((unsigned)((total + 255) / 256)),
               dim3(256), 0, st, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, flags, cin, kout,
               (unsigned short*)workspace);
  }
  const int bx = (W + HBW - 1) / HBW, kb = kout / HNB;
  // block height: the one with the fewest (rounds of one workgroup per CU) x (window rows)
  const int cus = convh_cus();
  int bh = 12;
  int64_t best = -1, pblocks = 0;
  for (int cand : {12, 8, 6}) {
    const int64_t wg = (int64_t)B * ((H + cand - 1) / cand) * bx * kb;
    const int64_t cost = ((wg + cus - 1) / cus) * (cand + 2);
    if (best < 0 || cost < best) {
      best = cost;
      bh = cand;
    }
  }
  if (dv == 3012 || dv == 3008 || dv == 3006) bh = dv - 3000;
  if (dv == 3013) bh = 12;
  const bool cut2x4 = dv != 3012;      // 12 rows: 2 x 4 waves (HCfg<24>); 3012 pins the 4 x 2 cut, 3013 this one
  pblocks = (int64_t)B * ((H + bh - 1) / bh) * bx;
  const int vblocks = (int)(((pblocks + 7) / 8) * 8 * kb);
  int groups = cus / (8 * kb) > 0 ? cus / (8 * kb) : 1;
  if (dv >= 3100 && dv < 3200) groups = dv - 3100 + 1;
  int gsize = groups * 8 * kb;
  if (gsize > vblocks) gsize = vblocks;
  if (dv == 3099) gsize = vblocks;
  const dim3 grid((unsigned)gsize);
  // 3020 + bits: 1 no wait / barrier at the top of a tile, 2 no output stores
  // 3040: EXPERIMENT (correct results): s_setprio 1 for waves 4..7 before the tile loop
  const int dbgbits = (dv >= 3020 && dv < 3024) ? (dv - 3020) << 1 : (dv == 3040 ? 4 << 1 : 0);
#define SCL_CONVH_LAUNCH_F(E, BHV, FULLV, BIAS, RELU, MASK)                                   \
  SCL_LAUNCH("convh_kernel", (convh_kernel<E, BHV, false, FULLV>), grid, dim3(HTHR),           \
             HCfg<BHV>::LDS, st, (const unsigned short*)x, (const unsigned short*)packed, B, H, \
             W, cin, kout, (unsigned short*)out, BIAS, RELU, (const unsigned short*)MASK,      \
             (unsigned char*)pidx, vblocks)
#define SCL_CONVH_LAUNCH(E, BHV, BIAS, RELU, MASK)                                             \
  do {                                                                                         \
    if (H % HCfg<BHV>::BH == 0 && W % HBW == 0)                                                \
      SCL_CONVH_LAUNCH_F(E, BHV, true, BIAS, RELU, MASK);                                      \
    else                                                                                       \
      SCL_CONVH_LAUNCH_F(E, BHV, false, BIAS, RELU, MASK);                                     \
  } while (0)
#define SCL_CONVH_BH(E, BIAS, RELU, MASK)                                                      \
  do {                                                                                         \
    if (bh == 12 && cut2x4) SCL_CONVH_LAUNCH(E, 24, BIAS, RELU, MASK);                         \
    else if (bh == 12) SCL_CONVH_LAUNCH(E, 12, BIAS, RELU, MASK);                              \
    else if (bh == 8) SCL_CONVH_LAUNCH(E, 8, BIAS, RELU, MASK);                                \
    else SCL_CONVH_LAUNCH(E, 6, BIAS, RELU, MASK);                                             \
  } while (0)
  // stamps of wave dv - 3024 (2 x 4 cut) / dv - 3032 (4 x 2 cut)
  if (dv >= 3024 && dv < 3040 && bias && !mask && !pidx && bh == 12) {
    if (dv < 3032)
      SCL_LAUNCH("convh_kernel", (convh_kernel<1, 24, true>), grid, dim3(HTHR), HCfg<24>::LDS, st,
                 (const unsigned short*)x, (const unsigned short*)packed, B, H, W, cin, kout,
                 (unsigned short*)out, bias, (relu ? 1 : 0) | ((dv - 3024) << 4),
                 (const unsigned short*)nullptr, (unsigned char*)nullptr, vblocks);
    else
      SCL_LAUNCH("convh_kernel", (convh_kernel<1, 12, true>), grid, dim3(HTHR), HCfg<12>::LDS, st,
                 (const unsigned short*)x, (const unsigned short*)packed, B, H, W, cin, kout,
                 (unsigned short*)out, bias, (relu ? 1 : 0) | ((dv - 3032) << 4),
                 (const unsigned short*)nullptr, (unsigned char*)nullptr, vblocks);
    return scl_launch_status();
  }
  if (pidx)
    SCL_CONVH_BH(3, bias, dbgbits, nullptr);
  else if (mask)
    SCL_CONVH_BH(2, bias, dbgbits, mask);
  else if (bias)
    SCL_CONVH_BH(1, bias, (relu ? 1 : 0) | dbgbits, nullptr);
  else
    SCL_CONVH_BH(0, bias, 0, nullptr);
#undef SCL_CONVH_BH
#undef SCL_CONVH_LAUNCH
#undef SCL_CONVH_LAUNCH_F
  return scl_launch_status();
}
