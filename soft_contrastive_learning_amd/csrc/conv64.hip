// 3x3 / stride 1 / same-padding convolution with 64 input and 64 output channels on bf16
// channels-last activations — VGG16's conv1_2 at full resolution (model/nets.py:41-42), the
// one layer where the library kernels run at half the rate of its equal-FLOP siblings
// (DESIGN.md §7).  Used for the forward pass and, with the weights transposed and the taps
// flipped, for the backward-data pass.
//
// Implicit GEMM on v_mfma_f32_32x32x16_bf16: out[p][k] = sum_{tap, c} x[p + s_tap][c] W[tap][c][k],
// M = pixels, N = 64, K = 9 taps x 64 channels = 36 k-steps of 16.  The contraction index c is
// the FAST index of x, so an A fragment (pixel r, 8 consecutive channels) is one ds_read_b128
// out of a [10 rows][34 cols][64 ch] halo window in LDS at a per-tap offset — no im2col.
//   * persistent grid, one 256-thread workgroup per CU; each loops over 8 x 32-pixel tiles;
//   * wave (nt, half) owns output channels 32 nt .. +31 of tile rows 4 half .. +3; its whole
//     weight slice [36 k-steps][8 bf16 per lane] = 144 VGPRs stays in REGISTERS for the life
//     of the kernel (occupancy 1: 512 registers per lane), so LDS holds only the window;
//   * the next tile's window is loaded into registers before the MFMAs of the current one and
//     written to the other LDS buffer after them (zero padding at image borders by
//     predication); one barrier per tile;
//   * epilogue: f32 accumulators -> bf16 through a per-wave LDS transpose -> 16-byte stores.
#include <mutex>

#include "scl_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C64 = 64;
constexpr int TH = 8, TW = 32;                       // output tile
constexpr int WR = TH + 2, WC = TW + 2;              // halo window
constexpr int SCR_LD = 40;                           // bf16 per scratch row (32 ch + 8 pad)
constexpr int SCR = 32 * SCR_LD;                     // per-wave epilogue scratch (one tile row)

__device__ __forceinline__ f32x16 mfma32b(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef float f32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_ mfma16b(u32x4 a, u32x4 b, f32x4_ c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
typedef short s16x4_ __attribute__((ext_vector_type(4)));
// two transposed reads (ds_read_b64_tr_b16), `step4` elements apart: 8 consecutive rows of one column
__device__ __forceinline__ u32x4 tr_pair_early(const unsigned short* a0, int step4) {
  const s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4_ __attribute__((address_space(3)))*)(a0));
  const s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4_ __attribute__((address_space(3)))*)(a0 + step4));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return u32x4{l2.x, l2.y, h2.x, h2.y};
}

// LDS-DMA staging (global_load_lds_dwordx4): one wave-instruction copies 64 x 16 bytes from
// per-lane global addresses to 1 KB of CONSECUTIVE LDS — no staging registers, no ds_write
// pass, many more bytes in flight per CU.  Out-of-image halo pixels read a zero block.
__device__ uint4 zero_block[4];                      // never written: zeros

// Issued as inline asm so that hipcc does not order it against the LDS reads of the OTHER
// buffer (with the builtin it waits vmcnt(0) before the next ds_read: no overlap at all);
// the kernel waits vmcnt(0) itself before the barrier that hands the buffer over.
// lds_byte: wave-uniform LDS byte address of the 1-KB chunk; src: this lane's 16 bytes.
__device__ __forceinline__ void glds16(const unsigned short* src, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_byte)
      : "memory");
}
__device__ __forceinline__ unsigned lds_byte_of(const unsigned short* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned short*)p;
}
// ... with a wave-uniform base pointer and a 32-bit byte offset per lane: no vector instruction
// for the address at all
__device__ __forceinline__ void glds16_s(const unsigned short* base, unsigned off_bytes,
                                         unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off_bytes), "s"(base), "s"(lds_byte)
      : "memory");
}
// LDS-DMA through a BUFFER RESOURCE (buffer_load_dwordx4 ... offen lds): the same 1-KB copy, with
// the hardware's bounds check on every lane's byte offset — an offset at or beyond num_records,
// or a negative one (it wraps to > 2^31), delivers ZEROS (scripts/buffer_lds_probe.hip).  With one
// resource per image the rows of a halo window that lie above or below the image need no per-lane
// test, no zero block and no select: one v_add per chunk instead of ~11 vector instructions.
// rsrc: {base lo, base hi (stride 0), num_records in bytes, 0x00020000}, wave-uniform.
__device__ __forceinline__ void blds16(u32x4 rsrc, unsigned voff_bytes, unsigned lds_byte) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(rsrc), "s"(lds_byte)
      : "memory");
}
__device__ __forceinline__ u32x4 image_rsrc(const unsigned short* base, int64_t first_elem, unsigned bytes) {
  const unsigned long long a = (unsigned long long)(base + first_elem);
  return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
               (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
               (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
}
// Round 5: wrw64_kernel takes the buffer path for tiles whose window columns lie inside the image
// (DBG & 8 = the round-4 staging, for A/B in the diagnostic build).
constexpr bool kWrwBufPath = true;

// Generalisation to the other VGG shapes whose weight slice still fits the register file:
//   (CIN, KOUT) in {(64,64) conv1_2 fwd+bwd, (64,128) conv2_1 fwd, (128,64) conv2_1 bwd,
//   (128,128) conv2_2 fwd+bwd}.  A wave always owns 32 output channels (KOUT / 32 n-tiles)
// and, when KOUT = 64, one half of the tile's rows; its weights are 9 * CIN / 16 fragments =
// 144 (CIN = 64) or 288 (CIN = 128) VGPRs.  CIN = 128 uses 4-row tiles so that two halo
// windows fit LDS.
// GEO = 1 (round 6, (64, 64) only): 4-row tiles and FOUR waves per workgroup, so that TWO workgroups
// share a CU (2 x 69 KB of LDS, 256 registers per lane as before).  The two waves of a SIMD then
// belong to different workgroups: no barrier ties them together, so one runs its epilogue (matrix
// pipe idle) while the other is in its K loop — with eight waves in one workgroup the tile barrier
// phase-locked them (both in the K loop, then both in the epilogue: 8,100 cycles per tile against
// 4,608 of MFMA issue).
template <int CIN, int KOUT, int GEO = 0>
struct ConvCfg {
  static_assert(GEO == 0 || (CIN == 64 && KOUT == 64), "GEO 1 is conv1_2's geometry");
  static constexpr int TH_ = (CIN == 64 && GEO == 0) ? 8 : 4;
  static constexpr int WR_ = TH_ + 2;
  static constexpr int PIX = CIN + 8;                  // bf16 per staged pixel
  static constexpr int WIN_ = WR_ * WC * PIX;          // bf16 per window buffer
  static constexpr int PPP = CIN / 8;                  // 16-byte pieces per pixel
  static constexpr int PIECES_ = WR_ * WC * PPP;
  // (64, 64) runs eight waves (two per SIMD: 144 weight + 32 accumulator registers fit 256),
  // the other shapes four (their register budget needs occupancy 1)
  static constexpr int WAVES = (CIN == 64 && KOUT == 64 && GEO == 0) ? 8 : 4;
  static constexpr int NTHR = 64 * WAVES;
  static constexpr int SPP = PPP + 1;                  // 16-byte slots per staged pixel (+ pad)
  static constexpr int CHUNKS = (WR_ * WC * SPP + 63) / 64;   // 1-KB DMA chunks per window
  static constexpr int NI = (CHUNKS + WAVES - 1) / WAVES;     // chunks per wave
  static constexpr int SPT = CIN / 16;                 // k-steps per tap
  static constexpr int KS = 9 * SPT;                   // 36 or 72
  static constexpr int NT = KOUT / 32;                 // n-tiles: 2 or 4
  static constexpr int PARTS = WAVES / NT;             // row groups of the tile: 4, 2 or 1
  static constexpr int MT = TH_ / PARTS;               // tile rows per wave
  static constexpr size_t LDS =
      (2 * (size_t)WIN_ + WAVES * (size_t)SCR) * sizeof(unsigned short);
};

// Weights -> register image [nt][ks][lane 64][8 bf16].
//   transposed = 0 (forward):       B[c][k] = w[k][c][kh][kw]
//   transposed = 1 (backward-data): B[k][c] = w[k][c][2-kh][2-kw]  (contraction over k)
// w is addressed through its element strides (OIHW logical, any memory format); flags =
// SCL_CONV_TRANSPOSED | SCL_W_F32 (include/scl_hip.h).
template <int CIN, int KOUT>
__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const void* __restrict__ w,
                                                           int64_t sk, int64_t sc, int64_t sh,
                                                           int64_t sw, int flags,
                                                           unsigned short* __restrict__ packed) {
  using Cfg = ConvCfg<CIN, KOUT>;
  const int transposed = flags & 1, wf32 = flags & 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;      // over NT * KS * 64 * 8
  if (idx >= Cfg::NT * Cfg::KS * 512) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, ks = (idx >> 9) % Cfg::KS, nt = idx / (Cfg::KS * 512);
  const int j = lane & 31, h = lane >> 5;
  const int tap = ks / Cfg::SPT, kh = tap / 3, kw = tap % 3;
  const int cin = 16 * (ks % Cfg::SPT) + 8 * h + e;    // contraction index
  const int cout = 32 * nt + j;                        // output channel of this pass
  int64_t off;
  if (!transposed)
    off = cout * sk + cin * sc + kh * sh + kw * sw;
  else
    off = cin * sk + cout * sc + (2 - kh) * sh + (2 - kw) * sw;
  packed[idx] = weight_bf16(w, off, wf32);
}

// grid = number of CUs (persistent); block 256.  EPI (compile-time, so that the plain kernel
// keeps its register allocation): 0 plain, 1 + bias (+ ReLU), 2 raw + pooled output,
// 3 out = conv * [mask > 0] (the ReLU' of the layer below, for backward-data),
// 4 pooled output + the window position of each maximum (one byte), no full-size output.
//
// PL = 1 (backward-data of a layer that ends in the 2x2 max-pooling): x is the POOLED gradient
// [B][H/2][W/2][CIN] and uidx its window positions; the halo window of the full-size gradient is
// built in LDS by the threads (16 bytes of the pooled map + 8 index bytes per (pooled pixel,
// 8 channels), loaded when the tile starts, un-pooled and written in slices between the k-steps
// of the second half of the K loop) — the full-size map is never in memory.  H and W even.
//
// FW = 1 (round 5; with EPI 3, PL 1 and (64, 64) only: conv1_2's backward-data pass): the masked
// result — the gradient at conv1_1's pre-activation, 944 MB at 24 x 640x480 — is NOT written.  Its
// only consumer is the first layer's weight / bias / mean gradient (conv1_1 has no backward-data
// pass: its input is the image), a GEMM over pixels with 27 + 5 columns per pixel
// (conv_first_wrw_kernel, below): that product runs HERE, on the tile while it is in LDS —
//   * after the K loop and one more barrier the tile's window buffer is dead; the epilogue puts the
//     masked bf16 tile there as two 32-channel planes [256 px][32 ch] (the layout of the
//     weight-gradient kernels: transposed reads give 8 pixels of a channel per lane);
//   * the im2col matrix of the tile's x0 window ([10][34][3 + 1] bf16, its own small double-buffered
//     LDS area) is built ONCE per tile, by all 512 threads, into the epilogue scratch (dead by then)
//     as ready B operands [16 steps][32 columns][16 pixels]: column 27 = 1 (bias gradient), 28 .. 31
//     = first / last row / column indicators (the border sums of the mean gradient's closed form)
//     exactly as in conv_first_wrw_kernel.  (A first version let every wave gather its own operands
//     for v_mfma_f32_16x16x32_bf16 — 512 two-byte gather instructions per tile: 792 us for the
//     kernel against 505 + 216 for the two it replaces);
//   * wave (mt = wid & 1, pq = wid >> 1) multiplies channels 32 mt .. + 31 of tile rows 2 pq, 2 pq + 1
//     (four steps of 16 pixels) on v_mfma_f32_32x32x16_bf16 — one transposed read pair and one
//     ds_read_b128 per product; its partial [32 ch][32 cols] (16 registers) lives in LDS between
//     tiles (4 KB per wave: the kernel has no registers to spare), one slab [64][32] per workgroup
//     at the end, conv_first_wrw_reduce_kernel sums the slabs as before; the four corner pixels of
//     every image (the davg kernel's X term) go to a compact side buffer.
// `pooled` carries x0 and `pidx` the float32 slab / corner workspace in this mode (EPI 3 uses
// neither); saves the 944 MB store here and the 944 MB read + 217 us of conv_first_wrw_kernel.
template <int CIN, int KOUT, int EPI, int PL = 0, int FW = 0, int GEO = 0>
__global__ __launch_bounds__((ConvCfg<CIN, KOUT, GEO>::NTHR), (GEO ? 2 : 1)) void conv3x3_kernel(const unsigned short* __restrict__ x,
                                                         const unsigned short* __restrict__ packed,
                                                         int B, int H, int W,
                                                         unsigned short* __restrict__ out,
                                                         const float* __restrict__ bias, int relu,
                                                         unsigned short* __restrict__ pooled,
                                                         const unsigned short* __restrict__ mask,
                                                         unsigned char* __restrict__ pidx,
                                                         const unsigned char* __restrict__ uidx) {
  using Cfg = ConvCfg<CIN, KOUT, GEO>;
  constexpr int TH_ = Cfg::TH_, PIX = Cfg::PIX, WIN_ = Cfg::WIN_, KS = Cfg::KS, MT = Cfg::MT;
  static_assert(!FW || GEO == 0, "the fused first-layer gradients are written for the 8-row tile");
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nt = wid % Cfg::NT, part = wid / Cfg::NT;
  unsigned short* scr = lds + 2 * WIN_ + wid * SCR;
  // FW: [8 waves][2 halves][64 lanes][4] float32 partials | two x0 windows [10][34][4] bf16
  static_assert(!FW || (CIN == 64 && KOUT == 64 && EPI == 3 && PL == 1), "FW: conv1_2 backward-data only");
  constexpr int FWPL = 32, FGPL = TH * TW * FWPL, FXW = WR * WC * 4;
  float* dwl = reinterpret_cast<float*>(lds + 2 * WIN_ + Cfg::WAVES * SCR);
  unsigned short* xwl = reinterpret_cast<unsigned short*>(dwl + Cfg::WAVES * 1024);
  unsigned short* imc = lds + 2 * WIN_;     // im2col tile [16][32][16] bf16 = 16 KB of the 20 KB scratch
  const unsigned short* fx0 = reinterpret_cast<const unsigned short*>(pooled);
  float* fslabs = reinterpret_cast<float*>(pidx);
  // epilogue fusions (lane r <-> output channel 32 nt + r):
  //   pooled == NULL: out = acc (+ bias) (ReLU if relu)            conv + bias + activation
  //   pooled != NULL: out = acc raw, pooled = relu(max2x2(acc) + bias)   conv + pool + ReLU
  const float bias_r = (EPI == 1 || EPI == 2 || EPI == 4) ? bias[32 * nt + r] : 0.f;
  // SWAP (the epilogues without pooling): the MFMA runs with the operands swapped — weights as A,
  // pixels as B; products and summation order do not change — so that accumulator register q of
  // lane (pixel r, half h) is channel acc_row(q, h) of the wave's 32: four consecutive channels
  // per register quad, i.e. two packed conversions and one 8-byte LDS write instead of four
  // conversions and four 2-byte writes.  The pooled epilogues keep pixels as A (their 2 x 2
  // windows are register pairs of one lane).
  constexpr bool SWAP = EPI == 0 || EPI == 1 || EPI == 3;
  f32x4 bias_q[4];                     // SWAP + bias: channels 8 j + 4 h .. + 3 of the n-tile
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bias_q[j] = (SWAP && EPI == 1) ? *reinterpret_cast<const f32x4*>(bias + 32 * nt + 8 * j + 4 * h)
                                   : f32x4{0.f, 0.f, 0.f, 0.f};

  // the wave's weight slice: KS fragments of 16 bytes per lane
  u32x4 wf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wf[ks] = *reinterpret_cast<const u32x4*>(packed + (((int64_t)nt * KS + ks) * 64 + lane) * 8);

  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH_ - 1) / TH_;
  const int per_img = tiles_x * tiles_y;
  const int ntiles = B * per_img;

  // The window [pixel][CIN + 8] is filled by LDS-DMA: wave wid issues chunks wid, wid + WAVES,
  // ...; lane l of chunk j owns slot 64 j + l = (pixel, piece) with piece == PPP the pad (not
  // fetched).  The slot's place in the window does not depend on the tile: computed once.
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  int rel[Cfg::NI], roff[Cfg::NI];
#pragma unroll
  for (int i = 0; i < Cfg::NI; ++i) {
    const int slot = 64 * (wid_s + Cfg::WAVES * i) + lane;
    const int pix = slot / Cfg::SPP, piece = slot % Cfg::SPP;
    rel[i] = (piece < Cfg::PPP && pix < Cfg::WR_ * WC) ? ((pix / WC) << 8) | (pix % WC) : -1;
    roff[i] = ((pix / WC) * W + pix % WC) * CIN + 8 * piece;
  }
  const unsigned short* zeros = reinterpret_cast<const unsigned short*>(zero_block);
  auto stage_issue = [&](int tile, int buf) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int y0 = (t2 / tiles_x) * TH_ - 1, x0 = (t2 % tiles_x) * TW - 1;
    const int xo = ((b * H + y0) * W + x0) * CIN;        // < 2^31 elements: host check
    const unsigned base = lds_byte_of(lds) + buf * WIN_ * 2;
#pragma unroll
    for (int i = 0; i < Cfg::NI; ++i) {
      const int j = wid_s + Cfg::WAVES * i;
      if (j < Cfg::CHUNKS) {                             // wave-uniform
        const int y = y0 + (rel[i] >> 8), xx = x0 + (rel[i] & 255);
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
        const unsigned short* src = ok ? x + (xo + roff[i]) : zeros;
        if (rel[i] >= 0) glds16(src, base + j * 1024);
      }
    }
  };
  // PL: the window [WR_][34] of the full-size gradient starts at the odd pixel (ty - 1, tx - 1):
  // it is covered by (TH_ / 2 + 2) x 18 pooling windows, the outer ones with one row / column
  // inside.  Task (round rd, thread) = (pooled pixel of that grid, 8 channels).
  constexpr int PWR = TH_ / 2 + 2, PWC = TW / 2 + 2, PPP = Cfg::PPP;
  constexpr int PNR = PL ? (PWR * PWC * PPP + Cfg::NTHR - 1) / Cfg::NTHR : 1;
  // The rounds travel in NB batches of PLIVE rounds, one after the other through the same
  // registers: batch g is loaded at k-step g * SPAN - 1 (the first one when the tile starts) and
  // written in the last 4 * PLIVE + 2 .. 2 k-steps of its span.  (64, 64): two batches of one round
  // — six registers live, the kernel sits at the 256 of two waves per SIMD.
  constexpr int PLIVE = CIN == 64 ? 1 : PNR, NB = PNR / PLIVE, SPAN = KS / NB;
  static_assert(!PL || (PNR % PLIVE == 0 && 4 * PLIVE + 2 < SPAN), "un-pooling schedule");
  const int Ho = H >> 1, Wo = W >> 1;
  u32x4 pg[PLIVE];
  uint2 pi[PLIVE];
  const int p_piece = threadIdx.x % PPP;
  // pooled (row << 8 | column) of this thread's task in round rd, -1: none
  // (FW: re-derived at every use from an opaque copy of the thread index — kept across the K loop
  // these per-thread constants were what the fused kernel spilled, and every scratch reload waits
  // vmcnt(0), i.e. for the next tile's pooled gradient, in the middle of the products)
  auto p_rc = [&](int rd) {
    int tid_ = threadIdx.x;
    if (FW || GEO) asm volatile("" : "+v"(tid_));
    const int pp = (tid_ + Cfg::NTHR * rd) / PPP;
    return pp < PWR * PWC ? ((pp / PWC) << 8) | (pp % PWC) : -1;
  };
  auto pool_issue = [&](int tile, int g) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int py0 = (t2 / tiles_x) * (TH_ / 2) - 1, px0 = (t2 % tiles_x) * (TW / 2) - 1;
#pragma unroll
    for (int i = 0; i < PLIVE; ++i) {
      const int rc = p_rc(g * PLIVE + i);
      const int py = py0 + (rc >> 8), px = px0 + (rc & 255);
      const int off = ((b * Ho + py) * Wo + px) * CIN + 8 * p_piece;
      if (rc >= 0 && (unsigned)py < (unsigned)Ho && (unsigned)px < (unsigned)Wo) {
        pg[i] = *reinterpret_cast<const u32x4*>(x + off);
        pi[i] = *reinterpret_cast<const uint2*>(uidx + off);
      } else {
        pg[i] = u32x4{0u, 0u, 0u, 0u};
        pi[i] = uint2{0u, 0u};
      }
    }
  };
  // slice (round i of batch g, pos): window pixel (2 pr + dy - 1, 2 pc + dx - 1) if inside
  auto pool_write = [&](int g, int i, int pos, int buf) {
    const int rc = p_rc(g * PLIVE + i);
    const int wy = 2 * (rc >> 8) + (pos >> 1) - 1, wx = 2 * (rc & 255) + (pos & 1) - 1;
    const UnpoolFlags fl = unpool_flags(pi[i].x, pi[i].y);
    const u32x4 o = unpool8(pg[i], fl, pos);
    if (rc >= 0 && (unsigned)wy < (unsigned)Cfg::WR_ && (unsigned)wx < (unsigned)WC)
      *reinterpret_cast<u32x4*>(lds + buf * WIN_ + (wy * WC + wx) * PIX + 8 * p_piece) = o;
  };

  const int dbg = SCL_DIAG_ONLY(relu >> 1);            // timing diagnostics (scl_debug_set_variant(60000 + bits))
  relu &= 1;
  const short relu_floor = relu ? (short)0 : (short)-32768;   // packed ReLU: max with 0, or with the least int16
  // (experiment, 60004: static priority for the younger half of an eight-wave workgroup)
  if ((dbg & 4) && (threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);
  // FW: this thread's two elements of a tile's x0 window (340 pixels x 3 channels)
  unsigned short xreg[FW ? 2 : 1];     // (two registers, written by the loads and by nothing else: packed into
                                       // one the shift / or had to wait for the load right behind it)
  // (thread index through an opaque copy per call: its quotients by 3 and 34 must not be hoisted
  // out of the tile loop — the K loop has no register for them)
  auto xwin_load = [&](int tile_) {
    const int b_ = tile_ / per_img, t2_ = tile_ % per_img;
    const int ty_ = (t2_ / tiles_x) * TH_, tx_ = (t2_ % tiles_x) * TW;
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int idx = v * Cfg::NTHR + tid2;
      const int pix = idx / 3, c = idx - 3 * pix;
      const int y = ty_ - 1 + pix / WC, xx = tx_ - 1 + pix % WC;
      const bool ok = idx < 3 * WR * WC && y >= 0 && y < H && xx >= 0 && xx < W;
      xreg[v] = ok ? fx0[(((int64_t)b_ * H + y) * W + xx) * 3 + c] : (unsigned short)0;
    }
  };
  auto xwin_store = [&](int par) {
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int idx = v * Cfg::NTHR + tid2;
      const int pix = idx / 3, c = idx - 3 * pix;
      if (idx < 3 * WR * WC) xwl[par * FXW + pix * 4 + c] = xreg[v];
    }
  };
  int tile = blockIdx.x;
  if (FW) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
      *reinterpret_cast<f32x4_*>(dwl + wid * 1024 + v * 256 + lane * 4) = f32x4_{0.f, 0.f, 0.f, 0.f};
    if (tile < ntiles) {
      xwin_load(tile);
      xwin_store(0);
    }
  }
  int xpar = 0;
  // dbg & 64 (scl_debug_set_variant(61064), FW only): wave `dbg >> 7` of every workgroup writes
  // s_memtime stamps of its tiles 2 .. 5 behind the slabs ([workgroup][tile][12] at float offset
  // 1024 * 2048 of the workspace): 0 tile start, 1 K loop done, 2 own loads landed, 3 barrier 1
  // passed, 4 epilogue rows in LDS, 5 barrier 2, 6 im2col built, 7 barrier 3, 8 products done,
  // 9 barrier 4 (scripts/first_wrw_fused_ablate.py --stamps)
  uint64_t fstamp[10];
  int ftile = 0;
#define FW_STAMP(k)                                                                     \
  if (FW && (dbg & 64))                                                                 \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fstamp[k])::"memory")
  if (tile < ntiles) {
    if (PL) {
#pragma unroll
      for (int g = 0; g < NB; ++g) {
        pool_issue(tile, g);
#pragma unroll
        for (int sl = 0; sl < 4 * PLIVE; ++sl) pool_write(g, sl >> 2, sl & 3, 0);
      }
    } else {
      stage_issue(tile, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    FW_STAMP(0);
    const int next = (dbg & 1) ? ntiles : tile + gridDim.x;
    if (next < ntiles) {                                 // lands under the whole K loop
      if (PL) pool_issue(next, 0); else stage_issue(next, buf ^ 1);
    }
    // EPI 3: the tile's mask values are fetched now, under the K loop (an epilogue that waits
    // for them exposes the HBM latency once per tile row)
    u32x4 mk[EPI == 3 ? MT : 1][2];
    if (EPI == 3) {
      const int b_ = tile / per_img, t2_ = tile % per_img;
      const int my0 = (t2_ / tiles_x) * TH_ + MT * part, mx = (t2_ % tiles_x) * TW + (lane >> 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bool in_ = my0 + mt < H && mx < W;
        const int64_t mo = (((int64_t)b_ * H + my0 + mt) * W + mx) * KOUT + 32 * nt + 8 * (lane & 1);
        mk[mt][0] = in_ ? *reinterpret_cast<const u32x4*>(mask + mo) : u32x4{0u, 0u, 0u, 0u};
        mk[mt][1] = in_ ? *reinterpret_cast<const u32x4*>(mask + mo + 16) : u32x4{0u, 0u, 0u, 0u};
      }
    }

    // A fragment of (tile row mt, k-step ks): window pixel (MT part + mt + kh, r + kw),
    // channels 16 (ks % SPT) + 8 h .. + 7
    const unsigned short* wb = lds + buf * WIN_ + ((MT * part) * WC + r) * PIX + 8 * h;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero16();
    // The K loop walks (16-channel slice c, window row wr, tap column kw): the fragment of window
    // pixel row wr at column offset kw is read ONCE and multiplied into every tile row mt that
    // sees it as tap row kh = wr - mt in 0..2 — (MT + 2) * 3 fragment reads per slice instead of
    // 9 MT (MT = 2: 12 for 18, MT = 4: 18 for 36, MT = 8: 30 for 72).  Measured: within noise on
    // conv1_2, -2 % on conv2_2 — the LDS pipe is 25 % busy in this kernel (SQ_LDS_IDX_ACTIVE,
    // profiles/r03), it was never the bound; kept for the traffic it saves.
    constexpr int WRW = MT + 2, NST = Cfg::SPT * WRW * 3, RING = 4, AHEAD = 3;
    auto frag_at = [&](int st) {
      const int c = st / (WRW * 3), wr = (st / 3) % WRW, kw = st % 3;
      return *reinterpret_cast<const u32x4*>(wb + (wr * WC + kw) * PIX + 16 * c);
    };
    u32x4 af[RING];
#pragma unroll
    for (int st = 0; st < AHEAD; ++st) af[st] = frag_at(st);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (st + AHEAD < NST) af[(st + AHEAD) % RING] = frag_at(st + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      const int c = st / (WRW * 3), wr = (st / 3) % WRW, kw = st % 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int kh = wr - mt;
        if (kh >= 0 && kh < 3)
          acc[mt] = SWAP ? mfma32b(wf[(3 * kh + kw) * Cfg::SPT + c], af[st % RING], acc[mt])
                         : mfma32b(af[st % RING], wf[(3 * kh + kw) * Cfg::SPT + c], acc[mt]);
      }
      // the un-pooling slices keep their places on the scale of the KS k-steps
      const int ks = st * KS / NST;
      if (PL && (st == 0 || ks != (st - 1) * KS / NST)) {
        // (also after the last tile: stale registers into a buffer nobody reads — no branch)
        constexpr int W0 = SPAN - 4 * PLIVE - 2;
        const int g = ks / SPAN, kk = ks % SPAN;
        if (kk >= W0 && kk < W0 + 4 * PLIVE) pool_write(g, (kk - W0) >> 2, (kk - W0) & 3, buf ^ 1);
        if (kk == SPAN - 1 && g + 1 < NB && next < ntiles) pool_issue(next, g + 1);
      }
    }

    // this wave's share of the next window has landed (waited for here, before the epilogue's
    // own stores join the queue); the barrier below publishes it
    FW_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FW_STAMP(2);
    if (FW) {
      __syncthreads();             // every wave is done with the window: its buffer takes the tile
      FW_STAMP(3);
      // the next tile's x0 window: requested here, stored behind the products below.  (Requested
      // at the top of the tile instead — its one register alive across the K loop — the kernel
      // measured 844 us against 765: profiles/r05/first_wrw_fused_notes.txt)
      if (next < ntiles) xwin_load(next);
    }

    // epilogue: tile row MT part + mt, accumulator register q <-> column acc_row(q, h),
    // lane r <-> output channel 32 nt + r
    const int b = tile / per_img, t2 = tile % per_img;
    const int oy0 = (t2 / tiles_x) * TH_ + MT * part, ox0 = (t2 % tiles_x) * TW;
#pragma unroll
    for (int mt = 0; mt < (EPI == 4 ? 0 : MT); ++mt) {
      // 32 pixels x 64 bytes in two store instructions: lane -> pixel lane >> 1; instruction i
      // writes channels 16 i + 8 (lane & 1) .. + 7, so a lane pair covers 32 contiguous bytes
      // (whole 32-byte sectors — interleaving the two lanes' 16-byte pieces writes half sectors)
      const int px = lane >> 1, hf = lane & 1;
      const int oy = oy0 + mt, ox = ox0 + px;
      const bool inside = oy < H && ox < W;
      const int64_t o_off = (((int64_t)b * H + oy) * W + ox) * KOUT + 32 * nt + 8 * hf;
      const u32x4 y0v = mk[EPI == 3 ? mt : 0][0], y1v = mk[EPI == 3 ? mt : 0][1];
      if (SWAP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = {acc[mt][4 * j], acc[mt][4 * j + 1], acc[mt][4 * j + 2], acc[mt][4 * j + 3]};
          if (EPI == 1) v += bias_q[j];
          unsigned p0 = pack2_bf16(v[0], v[1]), p1 = pack2_bf16(v[2], v[3]);
          if (EPI == 1) {                                  // ReLU on the rounded pairs, or the identity
            p0 = max2_i16(p0, relu_floor);
            p1 = max2_i16(p1, relu_floor);
          }
          *reinterpret_cast<uint2*>(scr + r * SCR_LD + 8 * j + 4 * h) = make_uint2(p0, p1);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) scr[acc_row(q, h) * SCR_LD + r] = f32_to_bf16(acc[mt][q]);   // (EPI 2: raw)
      }
      __builtin_amdgcn_wave_barrier();
      u32x4 v0 = *reinterpret_cast<const u32x4*>(scr + px * SCR_LD + 8 * hf);
      u32x4 v1 = *reinterpret_cast<const u32x4*>(scr + px * SCR_LD + 8 * hf + 16);
      __builtin_amdgcn_wave_barrier();
      if (EPI == 3) {
        v0 = relu_mask(v0, y0v);
        v1 = relu_mask(v1, y1v);
      }
      if (FW && (dbg & 32)) {
        // (timing ablation, scl_debug_set_variant(61032): the tile is not written to LDS)
      } else if (FW) {
        // (outside the image the mask was loaded as zero: v0 = v1 = 0 there already)
        unsigned short* gzp = lds + buf * WIN_ + nt * FGPL + ((MT * part + mt) * TW + px) * FWPL + 8 * hf;
        *reinterpret_cast<u32x4*>(gzp) = v0;
        *reinterpret_cast<u32x4*>(gzp + 16) = v1;
        // the four corner pixels of the image: what conv_first_davg_kernel reads of this map
        const bool cy0 = oy == 0, cy1 = oy == H - 1, cx0 = ox == 0, cx1 = ox == W - 1;
        if ((cy0 || cy1) && (cx0 || cx1)) {
          unsigned short* cp = reinterpret_cast<unsigned short*>(fslabs + (size_t)1536 * 2048) +
                               ((int64_t)b * 4 + (cy1 ? 2 : 0) + (cx1 ? 1 : 0)) * C64 + 32 * nt + 8 * hf;
          *reinterpret_cast<u32x4*>(cp) = v0;
          *reinterpret_cast<u32x4*>(cp + 16) = v1;
        }
      } else if (inside && !(dbg & 2)) {
        *reinterpret_cast<u32x4*>(out + o_off) = v0;
        *reinterpret_cast<u32x4*>(out + o_off + 16) = v1;
      }
    }
    if (FW) {
      FW_STAMP(4);
      __syncthreads();             // the whole masked tile is in LDS; the epilogue scratch is dead
      FW_STAMP(5);
      // (the thread's / lane's roles are re-derived per tile from opaque copies: hoisted out of
      // the tile loop they stay live across the K loop, which has no register left)
      int tid2 = threadIdx.x;
      asm volatile("" : "+v"(tid2));
      const unsigned short* xw = xwl + xpar * FXW;
      const int ty_ = oy0 - MT * part, tx_ = ox0;
      // 1. im2col: unit u = (step s, column n, pixel half hh) = 8 pixels of one column, two per thread
      // (dbg & 8 / & 16, scl_debug_set_variant(61008 / 61016): timing ablations without this pass /
      // without the products)
#pragma unroll
      for (int v = 0; v < ((dbg & 8) ? 0 : 2); ++v) {
        const int u = v * Cfg::NTHR + tid2;                   // 0 .. 1023
        const int hh = u & 1, n = (u >> 1) & 31, s_ = u >> 6;
        const int row = s_ >> 1, col0 = 16 * (s_ & 1) + 8 * hh;   // first pixel of the unit
        const int nc = n < 27 ? n : 26, tap = nc / 3, c = nc - 3 * tap;
        const unsigned short* wp = xw + ((row + tap / 3) * WC + col0 + tap % 3) * 4 + c;
        unsigned e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = wp[4 * j];
        if (n >= 27) {
          const int y = ty_ + row, xb = tx_ + col0;
          const bool rowflag = n == 27 || (n == 28 && y == 0) || (n == 29 && y == H - 1);
          const int colx = n == 30 ? 0 : (n == 31 ? W - 1 : -1);
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = (rowflag || xb + j == colx) ? 0x3f80u : 0u;
        }
        *reinterpret_cast<u32x4*>(imc + (s_ * 32 + n) * 16 + 8 * hh) =
            u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16)};
      }
      FW_STAMP(6);
      __syncthreads();
      FW_STAMP(7);
      // 2. the products: wave (mt, pq), steps 4 pq .. 4 pq + 3
      int lane2 = lane, wid2 = wid;
      asm volatile("" : "+v"(lane2), "+v"(wid2));
      const int mt_ = wid2 & 1, pq = wid2 >> 1;
      const int r2 = lane2 & 31, h2 = lane2 >> 5;
      const int q4 = (lane2 >> 2) & 3, p4 = lane2 & 3, gq = lane2 >> 4;
      // A (gz^T): the lane addresses pixel 8 (gq >> 1) + q4 of the step, channels 16 (gq & 1) + 4 p4 ..
      const unsigned short* ga = lds + buf * WIN_ + mt_ * FGPL + (8 * (gq >> 1) + q4) * FWPL +
                                 16 * (gq & 1) + 4 * p4;
      f32x16 dacc;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4_ t = *reinterpret_cast<const f32x4_*>(dwl + wid2 * 1024 + v * 256 + lane2 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) dacc[4 * v + j] = t[j];
      }
#pragma unroll
      for (int u = 0; u < ((dbg & 16) ? 0 : 4); ++u) {
        const int s_ = 4 * pq + u;                                   // 16 pixels: row s_ / 2, half s_ & 1
        const u32x4 a = tr_pair_early(ga + s_ * 16 * FWPL, 4 * FWPL);
        const u32x4 bfr = *reinterpret_cast<const u32x4*>(imc + (s_ * 32 + r2) * 16 + 8 * h2);
        dacc = mfma32b(a, bfr, dacc);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
        *reinterpret_cast<f32x4_*>(dwl + wid2 * 1024 + v * 256 + lane2 * 4) =
            f32x4_{dacc[4 * v], dacc[4 * v + 1], dacc[4 * v + 2], dacc[4 * v + 3]};
      if (next < ntiles) xwin_store(xpar ^ 1);
      xpar ^= 1;
      FW_STAMP(8);
    }
    if (EPI == 2 || EPI == 4) {
      // 2x2 / stride 2 max-pool of the raw outputs, lane-local: rows mt, mt + 1 are two
      // accumulators, columns acc_row(q, h), acc_row(q + 1, h) two registers (q even)
      const int PH2 = H / 2, PW2 = W / 2;
      unsigned char* scr8 = reinterpret_cast<unsigned char*>(scr + 16 * SCR_LD);   // [16][32]
#pragma unroll
      for (int mp = 0; mp < MT / 2; ++mp) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          const float m0 = fmaxf(acc[2 * mp][q], acc[2 * mp][q + 1]);
          const float m1 = fmaxf(acc[2 * mp + 1][q], acc[2 * mp + 1][q + 1]);
          const float m = fmaxf(m0, m1);
          scr[(acc_row(q, h) >> 1) * SCR_LD + r] = f32_to_bf16(fmaxf(m + bias_r, 0.f));
          if (EPI == 4) {
            // first maximum in raster order (0,0), (0,1), (1,0), (1,1)
            const int k = acc[2 * mp][q] == m ? 0 : acc[2 * mp][q + 1] == m ? 1
                          : acc[2 * mp + 1][q] == m ? 2 : 3;
            scr8[(acc_row(q, h) >> 1) * 32 + r] = (unsigned char)k;
          }
        }
        __builtin_amdgcn_wave_barrier();
        // 16 pooled pixels x 64 bytes: lane -> pixel lane >> 2, 16-byte quarter lane & 3
        const int px = lane >> 2, qu = lane & 3;
        const u32x4 v = *reinterpret_cast<const u32x4*>(scr + px * SCR_LD + 8 * qu);
        uint2 kv = uint2{0u, 0u};
        if (EPI == 4) kv = *reinterpret_cast<const uint2*>(scr8 + px * 32 + 8 * qu);
        __builtin_amdgcn_wave_barrier();
        const int py = (oy0 >> 1) + mp, pxg = (ox0 >> 1) + px;
        if (py < PH2 && pxg < PW2) {
          const int64_t po = (((int64_t)b * PH2 + py) * PW2 + pxg) * KOUT + 32 * nt + 8 * qu;
          *reinterpret_cast<u32x4*>(pooled + po) = v;
          if (EPI == 4) *reinterpret_cast<uint2*>(pidx + po) = kv;
        }
      }
    }

    __syncthreads();
    if (FW && (dbg & 64)) {
      FW_STAMP(9);
      if ((int)threadIdx.x == 64 * (dbg >> 7) && ftile >= 2 && ftile < 6) {
        uint64_t* o = reinterpret_cast<uint64_t*>(fslabs + (size_t)1024 * 2048) +
                      ((int64_t)blockIdx.x * 4 + (ftile - 2)) * 12;
#pragma unroll
        for (int k = 0; k < 10; ++k) o[k] = fstamp[k];
      }
      ++ftile;
    }
    buf ^= 1;
  }
#undef FW_STAMP
  if (FW) {
    // slab [64 ch][32 cols] of the workgroup = the four pixel quarters in a fixed order; the
    // partials pass through the (free) window buffers as [pq][64][32]
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    const int mt_ = wid & 1, pq = wid >> 1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const f32x4_ t = *reinterpret_cast<const f32x4_*>(dwl + wid * 1024 + v * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(pq * 64 + 32 * mt_ + acc_row(4 * v + j, h)) * 32 + r] = t[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2048; e += Cfg::NTHR)
      fslabs[(int64_t)blockIdx.x * 2048 + e] = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
  }
}

// ---------------------------------------------------------------------------------------
// wrw64_kernel: weight gradient of the same layer,
//   dW[k][c][kh][kw] = sum_{b,y,x} gz[b,y,x,k] * x[b, y+kh-1, x+kw-1, c].
// The contraction runs over pixels — the SLOW index of both channels-last operands — so both
// MFMA operands are read with ds_read_b64_tr_b16 out of row-major LDS tiles (the x halo
// window of conv64_kernel and the gz tile); a tap row is an address offset, the three tap
// columns of a row are one 12-pixel fragment shifted inside the lane (round 3, see the loop).
//   * v_mfma_f32_32x32x16_bf16 with M = 32 input channels, N = 32 output channels, 16 pixels
//     per step; wave (mt, nt) keeps all nine taps of its 32 x 32 block: 144 accumulators;
//   * persistent grid, one workgroup per CU accumulating over all its tiles, then ONE slab
//     [9][64][64] float32 per workgroup; wrw64_reduce_kernel sums the slabs in a fixed order
//     into the weight's own layout (bf16).
// LDS layout: each staged tile is TWO planes of 32 channels with 64-byte pixels and no padding.
// A ds_read_b64_tr_b16 is served 32 lanes at a time (8-byte units, 32 of them per cycle); the
// 32 lanes of a group address 4 pixels x (2 x 16 channels): unit = 8 * pixel + 4 * (channel
// half) + piece, all distinct mod 32 — conflict-free.  (One 64-channel plane at 144 B per
// pixel, fine for ds_read_b128, makes every transposed read 2-way conflicted: unit 18 * pixel.)
// The planes are filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass, 80 KB in flight per CU): one wave-instruction writes 1 KB = 16 pixels of a
// plane, lane -> (pixel lane >> 2, 16-byte piece lane & 3) — the image is lane-linear as the
// DMA requires.  Out-of-image halo pixels are fetched from a zero block.  Round 3: the DMA
// instructions of a tile go out one at a time between the products (back to back they parked the
// wave for 2,100 cycles), with a scalar base + constant lane offset for tiles off the border;
// the grid is 1-D with the workgroups that share tiles on one XCD; PL = 1 builds the gz tile from
// the pooled gradient + window index instead.
constexpr int WPL = 32;                              // bf16 per pixel and plane
constexpr int GPLANE = TH * TW * WPL;                // one 32-channel plane of a 256-pixel gz tile
                                                     // (conv_first_wrw_kernel; wrw64: WrwCfg)


typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 tr_pair(const unsigned short* a0, int step4) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(a0 + step4));
  const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
  return u32x4{l2.x, l2.y, h2.x, h2.y};
}

// grid (pixel splits P, C / 64, K / 64): workgroup (p, cb, kb) accumulates the [9][64][64]
// block (input channels 64 cb .., output channels 64 kb ..) over the tiles p, p + P, ...
// DBG (scl_debug_set_variant(2002), timing diagnostics only — wrong results): 2 = no staging
// after the first tile.
// Geometry of a variant: TWv = tile width (32 or 8), NKB = 64-channel output blocks per
// workgroup (1: [64 c] x [64 k], waves = 2 x 2 blocks x 2 pixel halves of a 256-pixel tile;
// 2: [64 c] x [128 k], waves = 2 x 4 blocks over a whole 128-pixel tile — the x window is
// fetched once for twice the output channels: 24 % fewer bytes per FLOP, which is what the
// kernel is bound by once the staging is DMA: 76 KB per 72 MFMAs of every wave).
template <int TWv, int NKB>
struct WrwCfg {
  static constexpr int PIXELS = 256 / NKB;
  static constexpr int THv = PIXELS / TWv, WCv = TWv + 2, WRv = THv + 2;
  static constexpr int XCHv = (WRv * WCv + 15) / 16;   // 1-KB chunks per x plane
  static constexpr int GCHv = PIXELS / 16;             // per gz plane
  static constexpr int XPL = XCHv * 16 * WPL, GPL = GCHv * 16 * WPL;
  static constexpr int BUF = 2 * XPL + 2 * NKB * GPL;  // one staged (x window, gz tile) pair
  static constexpr int CHUNKS = 2 * XCHv + 2 * NKB * GCHv;
  static constexpr size_t LDS = 2 * (size_t)BUF * sizeof(unsigned short);
};

// TWv: tile width, 32 (8 x 32 tiles, a step = 16 pixels of a row) or 8 (32 x 8 tiles for
// narrow maps, a step = two rows of 8): the same 256 pixels, 340-pixel windows and LDS image
// either way — only the pixel <-> address maps differ; the host takes the shape that pads the
// map less (W = 80: 96 -> 80 columns; 30 x 40: 32 x 64 -> 32 x 40).
//
// PL = 1: the layer ends in a 2x2 max-pooling, and `gz` is the POOLED gradient [B][H/2][W/2][K]
// with `pidx` the window position (2 dy + dx) of every maximum (scl_conv3x3_pool_idx): the
// full-size gradient — one non-zero per window and channel — is never in memory.  Its tile is
// built in LDS instead: every thread loads 16 bytes of the pooled gradient and their 8 index
// bytes for the NEXT tile when the tile starts (512 threads = the tile's pooled pixels x planes x
// pieces exactly), and writes the four 16-byte pixels of the window at the tile's end.  The gz
// part of the stream shrinks from 2 to 0.75 bytes per element; H and W are even (host check),
// so a window is inside or outside the image as a whole.
template <int DBG, int TWv, int NKB, int PL = 0>
__global__ __launch_bounds__(512, 1) void wrw64_kernel(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ gz,
                                                       int B, int H, int W, int C, int K,
                                                       float* __restrict__ slabs,
                                                       float* __restrict__ bslabs,
                                                       const unsigned char* __restrict__ pidx) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  using Cfg = WrwCfg<TWv, NKB>;
  // Workgroup -> (pixel split p, input block cb, output block kz).  The 1-D grid is dealt to the
  // eight XCDs round robin (workgroup L runs on XCD L % 8) and every workgroup is resident at
  // once.  The n_cb * n_kz workgroups of one p walk the same tiles at the same time — each x
  // window is read by n_kz of them, each gz tile by n_cb — so they are put on ONE XCD: its L2
  // then serves all but the first request (33 instead of 13 bytes per cycle and CU into the LDS;
  // with p on the fast grid axis the siblings sat on different XCDs and every one of them
  // fetched from the Infinity Cache: the staging stream, 58 KB per tile, was the kernel's bound).
  const int n_cb = C / C64, n_kz = K / (NKB * C64), n_sib = n_cb * n_kz;
  const int n_p = gridDim.x / n_sib;
  int wg_p, wg_sib;
  if (n_p % 8 == 0) {
    const int s_ = blockIdx.x >> 3;
    wg_sib = s_ % n_sib;
    wg_p = (blockIdx.x & 7) + 8 * (s_ / n_sib);
  } else {
    wg_p = blockIdx.x % n_p;
    wg_sib = blockIdx.x / n_p;
  }
  const int wg_cb = wg_sib % n_cb, wg_kz = wg_sib / n_cb;
  constexpr int XCH = Cfg::XCHv, GCH = Cfg::GCHv, XPLANE = Cfg::XPL, GPLANE = Cfg::GPL;
  constexpr int WBUF = Cfg::BUF, WCHUNKS = Cfg::CHUNKS;
  // eight waves, two per SIMD.  NKB = 1: wave (mt, nt, ph) accumulates block (mt, nt) over the
  // steps of pixel half ph of the tile; the two halves write separate slabs.  NKB = 2: wave
  // (mt, nt of 4) over the whole tile.  Eight steps of 16 pixels per wave and tile either way.
  const int mt = wid & 1, nt = NKB == 1 ? (wid >> 1) & 1 : wid >> 1, ph = NKB == 1 ? wid >> 2 : 0;
  // transposed-read role of this lane: 16-lane group gq = lane >> 4 covers channels
  // 16 (gq & 1) .. + 15 and pixels 8 (gq >> 1) + q (+ 4); lane 4q + p addresses row q,
  // columns 4p .. 4p + 3
  constexpr int THv = Cfg::THv, WCv = Cfg::WCv, WRv = Cfg::WRv;
  const int q = (lane >> 2) & 3, pp = lane & 3, gq = lane >> 4;
  const int ch0 = 16 * (gq & 1) + 4 * pp;
  // the lane's pixel inside a step: (row, column) relative to the step's first pixel
  const int lrow = TWv == 32 ? 0 : (gq >> 1), lcol = TWv == 32 ? 8 * (gq >> 1) + q : q;

  const int tiles_x = (W + TWv - 1) / TWv, tiles_y = (H + THv - 1) / THv;
  const int per_img = tiles_x * tiles_y;
  const int ntiles = B * per_img;

  // wave wid issues chunks j = wid, wid + 8, ... of the 76: [x plane 0][x plane 1][gz 0][gz 1]
  // (PL: of the x planes only).  Which pixel of the window / tile a lane fetches for its i-th
  // chunk does not depend on the tile: (row << 8 | column) once, -1 for the slack behind the
  // window.  A wave whose last index is past the end issues the last chunk once more (the same
  // bytes to the same place as the wave that owns it): stage_chunk has no branches, it sits
  // between the products of the unrolled tile loop.
  constexpr int NCH = PL ? 2 * XCH : WCHUNKS;
  constexpr int NI = (NCH + 7) / 8;
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);
  const int pl = lane >> 2, piece = lane & 3;
  int rel[NI], roff[NI];      // roff: element offset from the tile's first (halo) pixel
  int ldst[NI];               // (wave-uniform) LDS byte offset of the chunk inside a buffer
  bool is_x[NI];              // (wave-uniform) x window or gz tile
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int j = wid_s + 8 * i < NCH ? wid_s + 8 * i : NCH - 1;
    is_x[i] = j < 2 * XCH;
    if (j < 2 * XCH) {
      const int plane = j >= XCH ? 1 : 0;
      const int pix = 16 * (j - XCH * plane) + pl;
      rel[i] = pix < WRv * WCv ? ((pix / WCv) << 8) | (pix % WCv) : -1;
      // (the slack lanes behind the window point at its first pixel: never read, always mapped)
      const int pv = pix < WRv * WCv ? pix : 0;
      roff[i] = ((pv / WCv) * W + pv % WCv) * C + C64 * wg_cb + 8 * piece + 32 * plane;
      ldst[i] = (plane * XPLANE + (j - XCH * plane) * 512) * 2;
    } else {
      const int jj = j - 2 * XCH;
      const int plane = jj / GCH;
      const int pix = 16 * (jj - GCH * plane) + pl;
      rel[i] = ((pix / TWv) << 8) | (pix % TWv);
      roff[i] = ((pix / TWv) * W + pix % TWv) * K + NKB * C64 * wg_kz + 8 * piece + 32 * plane;
      ldst[i] = (2 * XPLANE + plane * GPLANE + (jj - GCH * plane) * 512) * 2;
    }
  }
  const unsigned short* zeros = reinterpret_cast<const unsigned short*>(zero_block);
  // One chunk (i) of a tile's staging; the whole of it = stage_issue.  Inside the tile loop the
  // chunks go out one at a time between the products (stage_chunk at every fifth): the s_memtime
  // stamps (scripts/wrw_stamps.py) showed a wave that issues its ten DMA instructions back to back
  // blocked for 2,100 cycles — the queue in front of the LDS takes them at the rate the data
  // arrives (58 KB at 13 B per cycle), and the wave's 72 products only started afterwards.
  struct TilePos {
    int ty, tx, xo, go;      // first pixel; element offsets of the window's / tile's first pixel
    int mode;                // 0: the whole halo window lies inside the image; 1: its COLUMNS do
                             // (rows may not): the buffer path; 2: per-lane border tests
    int b, xi, gi;           // image; the same offsets relative to the image's first element
  };
  auto tile_pos = [&](int tile) {
    const int b = tile / per_img, t2 = tile % per_img;
    TilePos t;
    t.ty = (t2 / tiles_x) * THv;
    t.tx = (t2 % tiles_x) * TWv;
    // (< 2^31: host check; the window's may be negative at the image border — those lanes are
    // masked by `ok`)
    t.xo = ((b * H + t.ty - 1) * W + t.tx - 1) * C;
    t.go = ((b * H + t.ty) * W + t.tx) * K;
    const bool colin = t.tx >= 1 && t.tx + TWv < W;
    // (one scalar: as two bools hipcc rebuilt them as lane masks at every chunk)
    t.mode = __builtin_amdgcn_readfirstlane(
        colin && t.ty >= 1 && t.ty + THv < H ? 0 : (colin && kWrwBufPath && !(DBG & 8) ? 1 : 2));
    t.b = b;
    t.xi = ((t.ty - 1) * W + t.tx - 1) * C;      // (negative in the first tile row)
    t.gi = (t.ty * W + t.tx) * K;
    return t;
  };
  auto stage_chunk = [&](const TilePos& tp, int buf, int i) __attribute__((always_inline)) {
    const int ty = tp.ty, tx = tp.tx, xo = tp.xo, go = tp.go;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_of(lds) + buf * WBUF * 2 + ldst[i]);
    if (tp.mode == 0) {
      // (wave-uniform) no border in sight: scalar base + the lane's constant offset
      glds16_s(is_x[i] ? x + xo : gz + go, 2u * (unsigned)roff[i], dst);
    } else if (tp.mode == 1) {
      // (wave-uniform) rows above / below the image: the resource of image b bounds every lane's
      // offset; a negative one wraps and is out of range too — zeros, like the zero block's
      const u32x4 rs = is_x[i] ? image_rsrc(x, (int64_t)tp.b * H * W * C, (unsigned)(H * W * C) * 2u)
                               : image_rsrc(gz, (int64_t)tp.b * H * W * K, (unsigned)(H * W * K) * 2u);
      blds16(rs, 2u * (unsigned)((is_x[i] ? tp.xi : tp.gi) + roff[i]), dst);
    } else {
      const int halo = is_x[i] ? 1 : 0;
      const int y = ty - halo + (rel[i] >> 8), xx = tx - halo + (rel[i] & 255);
      const bool ok = rel[i] >= 0 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
      const unsigned short* src = (is_x[i] ? x : gz) + ((is_x[i] ? xo : go) + roff[i]);
      glds16(ok ? src : zeros, dst);
    }
  };
  auto stage_issue = [&](int tile, int buf) __attribute__((always_inline)) {
    const TilePos tp = tile_pos(tile);
#pragma unroll
    for (int i = 0; i < NI; ++i) stage_chunk(tp, buf, i);
  };
  // PL: this thread's window of the tile = pooled pixel (pr, pc), 8 channels (plane, piece)
  constexpr int NPC = TWv / 2, NPR = THv / 2;
  const int p_piece = threadIdx.x & 3, p_pc = (threadIdx.x >> 2) % NPC;
  const int p_pr = ((threadIdx.x >> 2) / NPC) % NPR, p_plane = (threadIdx.x >> 2) / (NPC * NPR);
  const int Ho = H >> 1, Wo = W >> 1;
  u32x4 pg = {0u, 0u, 0u, 0u};
  uint2 pi = {0u, 0u};
  auto pool_issue = [&](int tile) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int py = (t2 / tiles_x) * NPR + p_pr, px = (t2 % tiles_x) * NPC + p_pc;
    const int off = ((b * Ho + py) * Wo + px) * K + NKB * C64 * wg_kz + 32 * p_plane + 8 * p_piece;
    if (py < Ho && px < Wo) {
      pg = *reinterpret_cast<const u32x4*>(gz + off);
      pi = *reinterpret_cast<const uint2*>(pidx + off);
    } else {
      pg = u32x4{0u, 0u, 0u, 0u};
      pi = uint2{0u, 0u};
    }
  };
  auto pool_write = [&](int buf) {
    unsigned short* dst = lds + buf * WBUF + 2 * XPLANE + p_plane * GPLANE +
                          (2 * p_pr * TWv + 2 * p_pc) * WPL + 8 * p_piece;
    const UnpoolFlags fl = unpool_flags(pi.x, pi.y);
#pragma unroll
    for (int pos = 0; pos < 4; ++pos)
      *reinterpret_cast<u32x4*>(dst + ((pos >> 1) * TWv + (pos & 1)) * WPL) = unpool8(pg, fl, pos);
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero16();
  // Bias gradient on the side: the B fragment of a step is 8 pixels of gz channel r of this
  // wave's n-tile — summed into one register per lane (v_dot2c_f32_bf16 against (1, 1): four
  // instructions per nine MFMAs, in every wave so that the loop stays branch-free; only the
  // waves mt == 0 of the workgroups cb == 0 write theirs out).
  float bsum = 0.f;
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

  // The wave's 72 (step, tap) products of a tile run as ONE software pipeline: the window-row
  // fragment a product opens is requested six products earlier (an LDS transposed read returns
  // long after one 32-cycle MFMA), the B operand of the next step while the current one runs.
  // The tiles alternate between two LDS buffers: the DMA of tile n + 1 runs under the whole
  // of tile n; one barrier per tile (it drains the DMA: hipcc waits vmcnt(0) there).
  // DBG & 4 (scl_debug_set_variant(2004), diagnostics): wave 0 of every workgroup writes s_memtime
  // stamps of its first four tiles to bslabs (no bias partials then): [workgroup][tile][8] —
  // 0 tile start, 1 staging issued, 2 first step done, 3 fourth step done, 4 last product
  // issued, 5 own DMA landed, 6 barrier passed.
  uint64_t stamp[7];
#define WRW_STAMP(k)                                                   \
  if (DBG & 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k])::"memory")
  int tile = wg_p;
  if (tile < ntiles) {
    stage_issue(tile, 0);
    if (PL) {
      pool_issue(tile);
      pool_write(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0, tcount = 0;
  for (; tile < ntiles; tile += n_p) {
    WRW_STAMP(0);
    const int next = (DBG & 2) ? ntiles : tile + n_p;
    // (after the last tile the current one is staged again, into the buffer nobody reads: the
    // product loop stays free of branches)
    const TilePos stg = tile_pos(next < ntiles ? next : tile);
    if (PL && next < ntiles) pool_issue(next);
    WRW_STAMP(1);
    const unsigned short* xl =
        lds + buf * WBUF + mt * XPLANE + (lrow * WCv + lcol) * WPL + ch0;
    const unsigned short* gl =
        lds + buf * WBUF + 2 * XPLANE + nt * GPLANE + (lrow * TWv + lcol) * WPL + ch0;
    // product i = 9 * s + t: step s of this wave's eight (first pixel (ry, cx)), tap t = 3 kh + kw.
    // A lane's A operand is 8 consecutive pixels of a window row for its channel; the three kw
    // taps of a row are the same pixels shifted by 0 / 1 / 2 — so ONE fragment of 12 pixels (three
    // transposed reads: pixels 0-3, 4-7, 8-11 of the lane's block) serves all three: kw = 1 is four
    // v_alignbit, kw = 2 a renaming of registers.  And a window row serves the kh taps of three
    // tile rows (wide tiles) / two steps (tall tiles): it is read once and stays in registers.
    // LDS reads per wave and tile: 36 (52 tall) + 16 for B instead of 144 + 16 — the transposed
    // reads of nine separate taps kept the LDS pipe busy for more cycles than the MFMAs take.
    struct XFrag {
      uint2 lo, hi, ex;
    };
    auto x_frag = [&](int wrow, int cx) {
      const unsigned short* p = xl + (wrow * WCv + cx) * WPL;
      XFrag f;
      f.lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                           (s16x4 __attribute__((address_space(3)))*)(p)));
      f.hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                           (s16x4 __attribute__((address_space(3)))*)(p + 4 * WPL)));
      f.ex = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                           (s16x4 __attribute__((address_space(3)))*)(p + 8 * WPL)));
      return f;
    };
    // window row (relative to the wave's first) and column slot of step s_, tap row kh
    // Wide tiles walk the left 16 columns of the wave's four rows, then the right ones (steps
    // 0-3, 4-7): three window rows of ONE column half are live at a time.
    constexpr int NFR = TWv == 32 ? 6 : 18, NFC = TWv == 32 ? 2 : 1;
    auto f_row = [](int s_, int kh) { return TWv == 32 ? (s_ & 3) + kh : 2 * s_ + kh; };
    auto f_col = [](int s_) { return TWv == 32 ? (s_ >> 2) : 0; };
    // is (s_, kh) the first product that touches its fragment?
    auto f_first = [](int s_, int kh) { return TWv == 32 ? ((s_ & 3) == 0 || kh == 2) : (s_ == 0 || kh >= 1); };
    const int row_w = TWv == 32 ? 4 * ph : 16 * ph;        // the wave's first window row
    XFrag fr[NFR][NFC];
    auto f_load = [&](int i) {                            // the fragment product i (kw = 0) opens
      const int s_ = i / 9, kh = (i % 9) / 3;
      fr[f_row(s_, kh)][f_col(s_)] = x_frag(row_w + f_row(s_, kh), 16 * f_col(s_));
    };
    auto b_of = [&](int s_) {
      const int ry = TWv == 32 ? 4 * ph + (s_ & 3) : 16 * ph + 2 * s_;
      return tr_pair(gl + (ry * TWv + 16 * f_col(s_)) * WPL, 4 * WPL);
    };
    constexpr int LEAD = 6;                                // products between a read and its first use
    constexpr int STG = 5;                                 // products between two staging chunks
    static_assert(STG * NI <= 9 * 8 - 12, "the last chunk needs time to land");
    u32x4 bf[2];
#pragma unroll
    for (int i = 0; i < LEAD; i += 3)
      if (f_first(i / 9, (i % 9) / 3)) f_load(i);
    bf[0] = b_of(0);
#pragma unroll
    for (int i = 0; i < 9 * 8; ++i) {
      const int s_ = i / 9, kh = (i % 9) / 3, kw = i % 3;
      if (i + LEAD < 9 * 8 && (i + LEAD) % 3 == 0 && f_first((i + LEAD) / 9, ((i + LEAD) % 9) / 3))
        f_load(i + LEAD);
      if (i % 9 == 2 && s_ + 1 < 8) bf[(s_ + 1) & 1] = b_of(s_ + 1);
      if (!(DBG & 2) && i % STG == 0 && i / STG < NI) stage_chunk(stg, buf ^ 1, i / STG);
      __builtin_amdgcn_sched_barrier(0);
      const XFrag& f = fr[f_row(s_, kh)][f_col(s_)];
      u32x4 a;
      if (kw == 0)
        a = u32x4{f.lo.x, f.lo.y, f.hi.x, f.hi.y};
      else if (kw == 1)
        a = u32x4{__builtin_amdgcn_alignbit(f.lo.y, f.lo.x, 16), __builtin_amdgcn_alignbit(f.hi.x, f.lo.y, 16),
                  __builtin_amdgcn_alignbit(f.hi.y, f.hi.x, 16), __builtin_amdgcn_alignbit(f.ex.x, f.hi.y, 16)};
      else
        a = u32x4{f.lo.y, f.hi.x, f.hi.y, f.ex.x};
      if constexpr ((DBG & 16) != 0) {
        // timing ablation (RESULTS MEANINGLESS): the same FLOPs, operand registers and accumulator
        // registers on v_mfma_f32_16x16x32_bf16 — what would the other MFMA shape be worth here?
        f32x16& c_ = acc[i % 9];
        const int o_ = 8 * (i & 1);
        f32x4_ q0 = {c_[o_], c_[o_ + 1], c_[o_ + 2], c_[o_ + 3]};
        f32x4_ q1 = {c_[o_ + 4], c_[o_ + 5], c_[o_ + 6], c_[o_ + 7]};
        q0 = mfma16b(a, bf[s_ & 1], q0);
        q1 = mfma16b(a, bf[s_ & 1], q1);
        c_[o_] = q0[0]; c_[o_ + 1] = q0[1]; c_[o_ + 2] = q0[2]; c_[o_ + 3] = q0[3];
        c_[o_ + 4] = q1[0]; c_[o_ + 5] = q1[1]; c_[o_ + 6] = q1[2]; c_[o_ + 7] = q1[3];
      } else {
        acc[i % 9] = mfma32b(a, bf[s_ & 1], acc[i % 9]);
      }
      if (i == 8) WRW_STAMP(2);
      if (i == 35) WRW_STAMP(3);
      if (i == 71) WRW_STAMP(4);
      if (i % 9 == 4) {
        const u32x4 bw = bf[s_ & 1];
        const bf16x2 one2 = __builtin_bit_cast(bf16x2, 0x3f803f80u);
        // (element by element: indexing the vector in a loop made hipcc feed word 0 four times)
        const unsigned w0 = bw.x, w1 = bw.y, w2 = bw.z, w3 = bw.w;
        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w0), one2, bsum, false);
        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w1), one2, bsum, false);
        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w2), one2, bsum, false);
        bsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w3), one2, bsum, false);
      }
    }
    if (PL && next < ntiles) pool_write(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA chunks have landed
    WRW_STAMP(5);
    __syncthreads();
    WRW_STAMP(6);
    if (DBG & 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (threadIdx.x == 0 && tcount < 4) {
        uint64_t* o = reinterpret_cast<uint64_t*>(bslabs) + ((int64_t)((int)blockIdx.x) * 4 + tcount) * 8;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[k] = stamp[k];
      }
      ++tcount;
    }
    buf ^= 1;
  }
#undef WRW_STAMP

  // slab[s][cb][kb][tap][c][k] with 64 x 64 blocks (cb, kb): s = 2 p + ph (NKB = 1) or p, kb =
  // the workgroup's block or its pair 2 z + nt / 2; accumulator register qq <-> c = 32 mt +
  // acc_row(qq, h), lane r <-> k
  const int slab = NKB == 1 ? 2 * wg_p + ph : wg_p;
  const int kb = NKB == 1 ? wg_kz : 2 * wg_kz + (nt >> 1);
  float* out = slabs + (((int64_t)slab * n_cb + wg_cb) * (NKB * n_kz) + kb) * 9 * C64 * C64;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int qq = 0; qq < 16; ++qq)
      out[(t * C64 + 32 * mt + acc_row(qq, h)) * C64 + 32 * (nt & 1) + r] = acc[t][qq];
  // bias partials [slab][K]: the two pixel halves h of a step combined in a fixed order
  const float bs = bsum + __shfl_xor(bsum, 32);
  if (bslabs && !(DBG & 4) && wg_cb == 0 && mt == 0 && h == 0)
    bslabs[(int64_t)slab * K + C64 * kb + 32 * (nt & 1) + r] = bs;
}

// dW element (k, c, kh, kw) = sum over the pixel-split slabs of its (cb, kb) block, written
// at the weight's strides (bf16 or float32).  grid (9*64*64/256, CB * KB), block 64 x RG: thread
// (j, g) sums the slabs p = g, g + RG, ... of the FOUR elements 4 (64 * blockIdx.x + j) .. + 3
// (16-byte loads); the RG partials are combined in a fixed order.  RG = 4, or 16 where few
// (cb, kb) blocks face many slabs (conv1_2 / conv2_1: 256 and 128 slabs for 1 and 2 blocks —
// with four groups the 37 MB took 75 us).
template <int RG>
__global__ __launch_bounds__(64 * RG) void wrw64_reduce_kernel(const float* __restrict__ slabs,
                                                              int nsplit, int KB, int64_t sk,
                                                              int64_t sc, int64_t sh, int64_t sw,
                                                              void* __restrict__ gw, int gw_f32,
                                                              const float* __restrict__ bslabs,
                                                              float* __restrict__ gb) {
  __shared__ f32x4 red[RG][64];
  const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int idx = 4 * (blockIdx.x * 64 + j);           // over 9 * 64 * 64, k fastest
  const int blk = blockIdx.y, nblk = gridDim.y;        // blk = cb * KB + kb
  const int64_t stride = (int64_t)nblk * 9 * C64 * C64;
  const float* base = slabs + (int64_t)blk * 9 * C64 * C64 + idx;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  int i = g;
  for (; i + RG < nsplit; i += 2 * RG) {
    s0 += *reinterpret_cast<const f32x4*>(base + (int64_t)i * stride);
    s1 += *reinterpret_cast<const f32x4*>(base + (int64_t)(i + RG) * stride);
  }
  if (i < nsplit) s0 += *reinterpret_cast<const f32x4*>(base + (int64_t)i * stride);
  red[g][j] = s0 + s1;
  __syncthreads();
  if (g == 0) {
    f32x4 s = red[0][j];
#pragma unroll
    for (int q = 1; q < RG; ++q) s += red[q][j];
    const int c = 64 * (blk / KB) + ((idx >> 6) & 63), t = idx >> 12;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 64 * (blk % KB) + ((idx + e) & 63);
      store_weight_grad(gw, k * sk + c * sc + (t / 3) * sh + (t % 3) * sw, s[e], gw_f32);
    }
  }
  // bias gradient: the blocks (0, cb = 0, kb) add up the partials of their 64 channels — slab
  // groups g, g + RG, ... in parallel, the RG partial sums combined in a fixed order (one wave
  // walking all slabs alone took 64 us of conv2_1's 70 us reduce: a dependent load per slab)
  if (gb && blockIdx.x == 0 && blk / KB == 0) {       // block-uniform
    const int K = 64 * KB, k = 64 * (blk % KB) + j;
    float acc_b = 0.f;
#pragma unroll 4
    for (int sl = g; sl < nsplit; sl += RG) acc_b += bslabs[(int64_t)sl * K + k];
    __syncthreads();                                   // red[] is free again
    reinterpret_cast<float*>(red)[g * 64 + j] = acc_b;
    __syncthreads();
    if (g == 0) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RG; ++q) t += reinterpret_cast<float*>(red)[q * 64 + j];
      gb[k] = t;
    }
  }
}


// ---------------------------------------------------------------------------------------
// conv_first_kernel: the whole first layer in one pass (model/nets.py:22-24, 39),
//   x0 = bf16(img - average_rgb),  y = relu(conv3x3(x0, w) + bias),
// from the float32 NHWC image: 3 input channels, 64 output channels.  K = 27 (padded to 32 =
// two k-steps of v_mfma_f32_32x32x16_bf16); an A fragment is gathered from a [10][34][3+1]
// bf16 halo window in LDS (eight 2-byte reads per lane per k-step — the kernel is bound by
// writing its 64-channel output, not by this).  x0 is written out for the weight gradient.
constexpr int F_PIX = 4;                              // bf16 per staged pixel (3 + 1 pad)
constexpr int F_WIN = WR * WC * F_PIX;                // 1360 bf16
constexpr size_t kFirstLds = ((size_t)F_WIN + 4 * 2 * (size_t)SCR) * sizeof(unsigned short);

__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ img,
                                                         const float* __restrict__ avg,
                                                         const void* __restrict__ w, int w_f32,
                                                         int64_t sk, int64_t sc, int64_t sh,
                                                         int64_t sw, const float* __restrict__ bias,
                                                         int B, int H, int W,
                                                         unsigned short* __restrict__ x0,
                                                         unsigned short* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  const int dbg = SCL_DIAG_ONLY(w_f32 >> 4);            // timing diagnostics (scl_debug_set_variant(70000 + bits))
  w_f32 &= 1;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  unsigned short* scr = lds + F_WIN + wid * 2 * SCR;

  // weights: B[k][n], k = 3 * tap + c (k >= 27: zero); lane (j, h) holds k = 16 ks + 8 h + e
  // (Round 6: the weights go through LDS as a [64 n][32 k] bf16 image — thread t fetches the eight values
  // (n = t / 4, k = 8 (t % 4) .. + 7) with INDEPENDENT loads (padding slots k >= 27 re-read k = 26 and are
  // zeroed; the float32 / bf16 branch is taken once around all of them) and a lane's four fragments are four
  // ds_read_b128.  As `k < 27 ? weight_bf16(..) : 0` per element, each of a lane's 32 loads sat in its own
  // branch behind an `s_waitcnt vmcnt(0)`: 32 dependent round trips before a workgroup's first tile, with
  // eight workgroups per CU in the grid.)
  u32x4 wf[2][2];
  {
    unsigned short* wimg = lds + F_WIN;                 // 4 KB of the epilogue scratch, free until the first tile
    const int n = threadIdx.x >> 2, kc = 8 * (threadIdx.x & 3);
    unsigned short v[8];
    if (w_f32) {
      float wl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kc + e < 27 ? kc + e : 26, tap = k / 3, c = k % 3;
        wl[e] = static_cast<const float*>(w)[n * sk + c * sc + (tap / 3) * sh + (tap % 3) * sw];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = kc + e < 27 ? f32_to_bf16(wl[e]) : (unsigned short)0;
    } else {
      unsigned short wl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kc + e < 27 ? kc + e : 26, tap = k / 3, c = k % 3;
        wl[e] = static_cast<const unsigned short*>(w)[n * sk + c * sc + (tap / 3) * sh + (tap % 3) * sw];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = kc + e < 27 ? wl[e] : (unsigned short)0;
    }
    *reinterpret_cast<u32x4*>(wimg + n * 32 + kc) =
        u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
              (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        wf[ks][nt] = *reinterpret_cast<const u32x4*>(wimg + (32 * nt + r) * 32 + 16 * ks + 8 * h);
    __syncthreads();                                    // (the scratch is the waves' again)
  }
  // A gather offsets (bf16 units, relative to the lane's pixel): window (kh, kw), channel c
  int goff[2][8];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 8 * h + e;
      const int tap = k < 27 ? k / 3 : 0, c = k < 27 ? k % 3 : 3;     // pad slot holds zero
      goff[ks][e] = ((tap / 3) * WC + tap % 3) * F_PIX + c;
    }
  const float a0 = avg[0], a1 = avg[1], a2 = avg[2];
  // (Round 6: the products run with the operands SWAPPED — weights as A, pixels as B — so that lane
  // (pixel r, half h) holds in registers 4 g .. 4 g + 3 the four consecutive channels 8 g + 4 h .. + 3 of its
  // pixel: two v_cvt_pk_bf16_f32 and ONE ds_write_b64 per group instead of four ds_write_b16 — the
  // epilogue's 32 two-byte LDS stores per m-tile were a quarter of the kernel's time without any global
  // traffic.  The bias of those channels sits in 32 registers.)
  float biasv[2][16];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) biasv[nt][q] = bias[32 * nt + acc_row(q, h)];

  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int per_img = tiles_x * tiles_y;
  const int ntiles = B * per_img;
  // The image pixels of tile t + 1 are requested before tile t is computed and stored (the
  // float32 image needs a conversion, so it cannot come by LDS-DMA): a thread owns window
  // pixels threadIdx.x and threadIdx.x + 256 (340 per window).  Without this the loads of a
  // tile — 88 MB in all — cost 115 of the kernel's 390 us: nothing else of the workgroup runs
  // while they are in flight.
  float pre[2][3];
  int64_t pre_p[2];        // pixel index, -1: outside the image (zeros) or no such window pixel
  auto prefetch = [&](int tile) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pix = threadIdx.x + 256 * it;
      const int wy = pix / WC, wx = pix % WC;
      const int yy = ty - 1 + wy, xx = tx - 1 + wx;
      pre_p[it] = -1;
      pre[it][0] = pre[it][1] = pre[it][2] = 0.f;
      if (pix < WR * WC && tile < ntiles && yy >= 0 && yy < H && xx >= 0 && xx < W && !(dbg & 1)) {
        const int64_t p = ((int64_t)b * H + yy) * W + xx;
        pre_p[it] = p;
        pre[it][0] = img[3 * p];
        pre[it][1] = img[3 * p + 1];
        pre[it][2] = img[3 * p + 2];
      }
    }
  };
  prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;
    __syncthreads();                                   // previous tile's window is consumed
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int pix = threadIdx.x + 256 * it;
      if (pix < WR * WC) {
        const int wy = pix / WC, wx = pix % WC;
        unsigned short v0 = 0, v1 = 0, v2 = 0;
        if (pre_p[it] >= 0) {
          const int64_t p = pre_p[it];
          v0 = f32_to_bf16(pre[it][0] - a0);
          v1 = f32_to_bf16(pre[it][1] - a1);
          v2 = f32_to_bf16(pre[it][2] - a2);
          if (wy >= 1 && wy <= TH && wx >= 1 && wx <= TW && !(dbg & 2)) {   // interior: this tile owns it
            x0[3 * p] = v0;
            x0[3 * p + 1] = v1;
            x0[3 * p + 2] = v2;
          }
        }
        *reinterpret_cast<uint2*>(lds + pix * F_PIX) =
            make_uint2((unsigned)v0 | ((unsigned)v1 << 16), (unsigned)v2);
      }
    }
    __syncthreads();
    prefetch(tile + gridDim.x);                        // in flight under the MFMAs and stores

    // wave w: tile rows 2w, 2w + 1 (two m-tiles of 32 pixels) x both n-tiles
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const unsigned short* base = lds + ((2 * wid + mt) * WC + r) * F_PIX;
      f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = base[goff[ks][e]];
        const u32x4 af = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                               (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        acc0 = mfma32b(wf[ks][0], af, acc0);
        acc1 = mfma32b(wf[ks][1], af, acc1);
      }
      const int oy = ty + 2 * wid + mt;
      // both n-tiles through the scratch (two planes of 32 channels), then whole 128-byte pixels
      // per store instruction: lane -> (pixel 8 k + lane / 8, 16-byte piece lane % 8) — a store
      // instruction writes 8 complete lines instead of 32 quarter lines
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            v[j] = fmaxf((nt ? acc1[4 * g + j] : acc0[4 * g + j]) + biasv[nt][4 * g + j], 0.f);
          *reinterpret_cast<uint2*>(scr + (32 * nt + r) * SCR_LD + 8 * g + 4 * h) =
              make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int piece = lane & 7;
      u32x4 vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int px = 8 * k + (lane >> 3);
        vv[k] = *reinterpret_cast<const u32x4*>(scr + (32 * (piece >> 2) + px) * SCR_LD + 8 * (piece & 3));
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ox = tx + 8 * k + (lane >> 3);
        if (oy < H && ox < W && !(dbg & 4))
          *reinterpret_cast<u32x4*>(y + (((int64_t)b * H + oy) * W + ox) * C64 + 8 * piece) = vv[k];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// conv12_kernel: conv1_1 and conv1_2 of the forward pass in ONE kernel (model/nets.py:22-24,
// 39-42): image -> x0 = bf16(img - average_rgb) -> y1 = relu(conv1_1(x0) + b1) -> pooled =
// relu(maxpool2x2(conv1_2(y1)) + b2) + the window index.  conv1_2's forward is bound by reading
// its 944 MB input with 1.33 x halo amplification; here a tile's
// [10][34] halo window of y1 is COMPUTED into LDS from a [12][36] window of the image (44 more
// MFMAs per tile next to conv1_2's 576) instead of fetched, and y1 — which the backward pass
// needs (weight gradient of conv1_2, ReLU' of conv1_1) — is written once from that window
// (interior pixels only: every y1 pixel has exactly one owner).  Replaces conv_first_kernel +
// conv3x3_kernel<64, 64, 4>: 1.25 GB less read per step, one launch less.
//
// The arithmetic is that of the two kernels it replaces, value for value: conv1_1 with the
// same K order (k = 3 tap + c, two k-steps), float32 bias, ReLU, bf16 rounding (the MFMA's
// operands are swapped — weights as A — so that a lane holds four consecutive channels of a
// pixel: products and summation order do not change); conv1_2 with the register-resident
// weight slices, K-loop order and pooled epilogue of conv3x3_kernel.  tests/test_gpu_backbone.py
// compares x0, y1, pooled map and index bit for bit with the two-kernel path.
//
// Per tile and workgroup (512 threads): [S2] waves 0..7 take the 11 32-pixel groups of the y1
// window: im2col gather from the image window, 4 MFMAs, bias / ReLU / zero outside the image
// (conv1_2's padding is zero in y1, not conv1_1 of a padded image), 8-byte LDS writes;
// barrier; [S3] the interior of the window leaves as whole 128-byte pixels; the next tile's
// image pixels (requested one tile earlier) go into the image window; conv1_2's K loop and
// pooled epilogue; barrier.  One window buffer: nothing is staged under the K loop.
constexpr int IWR = TH + 4, IWC = TW + 4;            // image window of a tile: 12 x 36 pixels
constexpr int C12_IW = IWR * IWC * F_PIX;            // bf16 (1728)
constexpr int C12_W1 = 2 * 2 * 64 * 8;               // conv1_1 fragments [k-step][ch tile][lane][8]
constexpr int C12_PT = (WR * WC + 31) / 32;          // 32-pixel groups of the y1 window: 11
constexpr size_t kConv12Lds =
    ((size_t)ConvCfg<64, 64>::WIN_ + C12_IW + C12_W1 + 128 + 8 * (size_t)SCR) * sizeof(unsigned short);

__global__ __launch_bounds__(512, 1) void conv12_kernel(
    const float* __restrict__ img, const float* __restrict__ avg, const void* __restrict__ w1,
    int w1_f32, int64_t sk, int64_t sc, int64_t sh, int64_t sw, const float* __restrict__ b1,
    const unsigned short* __restrict__ packed2, const float* __restrict__ b2, int B, int H, int W,
    unsigned short* __restrict__ x0, unsigned short* __restrict__ y1,
    unsigned short* __restrict__ pooled, unsigned char* __restrict__ pidx) {
  using Cfg = ConvCfg<64, 64>;
  constexpr int PIX = Cfg::PIX, WIN_ = Cfg::WIN_, KS = Cfg::KS, MT = Cfg::MT;
  static_assert(MT == 2 && Cfg::NTHR == 512, "conv12_kernel is conv3x3_kernel<64, 64>'s geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* win = lds;
  unsigned short* iw = lds + WIN_;
  unsigned short* w1i = iw + C12_IW;
  float* b1s = reinterpret_cast<float*>(w1i + C12_W1);
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nt = wid % Cfg::NT, part = wid / Cfg::NT;
  unsigned short* scr = w1i + C12_W1 + 128 + wid * SCR;
  const float bias_r = b2[32 * nt + r];

  // conv1_2: the wave's weight slice, KS fragments of 16 bytes per lane
  u32x4 wf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wf[ks] = *reinterpret_cast<const u32x4*>(packed2 + (((int64_t)nt * KS + ks) * 64 + lane) * 8);

  // conv1_1: fragment (k-step ks, channel tile ct) of lane (j, hh) = channel 32 ct + j,
  // k = 16 ks + 8 hh + e = 3 tap + c (k >= 27: zero) — conv_first_kernel's wf[ks][nt]
  if (threadIdx.x < 256) {
    const int ks = threadIdx.x >> 7, ct = (threadIdx.x >> 6) & 1, hh = (threadIdx.x >> 5) & 1;
    const int j = threadIdx.x & 31;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * ks + 8 * hh + e;
      const int tap = k / 3, c = k % 3;
      v[e] = k < 27 ? weight_bf16(w1, (32 * ct + j) * sk + c * sc + (tap / 3) * sh + (tap % 3) * sw, w1_f32)
                    : (unsigned short)0;
    }
    *reinterpret_cast<u32x4*>(w1i + threadIdx.x * 8) =
        u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
              (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
  }
  if (threadIdx.x < 64) b1s[threadIdx.x] = b1[threadIdx.x];
  const float a0 = avg[0], a1 = avg[1], a2 = avg[2];

  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int per_img = tiles_x * tiles_y;
  const int ntiles = B * per_img;

  // image window: thread t < 432 owns pixel (t / 36, t % 36) of the [12][36] window, whose origin
  // is the tile's corner minus (2, 2); its float32 values travel one tile ahead in registers
  const int iwy = (int)threadIdx.x / IWC, iwx = (int)threadIdx.x % IWC;
  const bool iw_owner = threadIdx.x < IWR * IWC;
  const bool iw_interior = iwy >= 2 && iwy < 2 + TH && iwx >= 2 && iwx < 2 + TW;
  float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f;
  int64_t pre_p = -1;
  auto prefetch = [&](int t) {
    pre_p = -1;
    pre0 = pre1 = pre2 = 0.f;
    if (t < ntiles && iw_owner) {
      const int b = t / per_img, t2 = t % per_img;
      const int yy = (t2 / tiles_x) * TH - 2 + iwy, xx = (t2 % tiles_x) * TW - 2 + iwx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const int64_t p = ((int64_t)b * H + yy) * W + xx;
        pre_p = p;
        pre0 = img[3 * p];
        pre1 = img[3 * p + 1];
        pre2 = img[3 * p + 2];
      }
    }
  };
  auto write_iw = [&]() {
    if (iw_owner) {
      unsigned short v0 = 0, v1 = 0, v2 = 0;
      if (pre_p >= 0) {
        v0 = f32_to_bf16(pre0 - a0);
        v1 = f32_to_bf16(pre1 - a1);
        v2 = f32_to_bf16(pre2 - a2);
        if (iw_interior) {                               // this tile owns the pixel
          x0[3 * pre_p] = v0;
          x0[3 * pre_p + 1] = v1;
          x0[3 * pre_p + 2] = v2;
        }
      }
      *reinterpret_cast<uint2*>(iw + threadIdx.x * F_PIX) =
          make_uint2((unsigned)v0 | ((unsigned)v1 << 16), (unsigned)v2);
    }
  };
  // gather offset (bf16 units from the lane's window pixel) of contraction index k
  auto goff_of = [](int k) {
    const int tap = k < 27 ? k / 3 : 0, c = k < 27 ? k % 3 : 3;      // the pad slot holds zero
    return ((tap / 3) * IWC + tap % 3) * F_PIX + c;
  };

  int tile = blockIdx.x;
  prefetch(tile);
  write_iw();
  prefetch(tile + gridDim.x);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;

    // [S2] y1 window: group pt = window pixels 32 pt .. + 31 (row-major over [10][34])
    for (int pt = wid; pt < C12_PT; pt += 8) {
      const int pix = 32 * pt + r;
      const bool live = pix < WR * WC;
      const int pc = live ? pix : WR * WC - 1;
      const int wy = pc / WC, wx = pc % WC;
      const unsigned short* base = iw + (wy * IWC + wx) * F_PIX;
      f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        unsigned short v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = base[h ? goff_of(16 * ks + 8 + e) : goff_of(16 * ks + e)];
        const u32x4 af = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                               (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        const u32x4 wa0 = *reinterpret_cast<const u32x4*>(w1i + ((ks * 2 + 0) * 64 + lane) * 8);
        const u32x4 wa1 = *reinterpret_cast<const u32x4*>(w1i + ((ks * 2 + 1) * 64 + lane) * 8);
        acc0 = mfma32b(wa0, af, acc0);
        acc1 = mfma32b(wa1, af, acc1);
      }
      // register q of lane (pixel r, half h) = channel 32 ct + acc_row(q, h): four consecutive
      // channels per register quad
      const int yy = ty - 1 + wy, xx = tx - 1 + wx;
      const bool inimg = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(b1s + 32 * ct + 8 * j + 4 * h);
          unsigned short o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float a = ct ? acc1[4 * j + i] : acc0[4 * j + i];
            o[i] = f32_to_bf16(fmaxf(a + bb[i], 0.f));
          }
          uint2 pk = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
          if (!inimg) pk = make_uint2(0u, 0u);           // conv1_2 pads y1 with zeros
          if (live) *reinterpret_cast<uint2*>(win + pix * PIX + 32 * ct + 8 * j + 4 * h) = pk;
        }
    }
    __syncthreads();

    // [S3] the tile's own 8 x 32 pixels of y1: 16-byte piece id -> (pixel id / 8, piece id % 8),
    // eight lanes write one 128-byte pixel
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int id = (int)threadIdx.x + 512 * k;
      const int px = id >> 3, piece = id & 7;
      const int py = px / TW, pxx = px % TW;
      const u32x4 v = *reinterpret_cast<const u32x4*>(win + ((py + 1) * WC + pxx + 1) * PIX + 8 * piece);
      const int oy = ty + py, ox = tx + pxx;
      if (oy < H && ox < W)
        *reinterpret_cast<u32x4*>(y1 + (((int64_t)b * H + oy) * W + ox) * C64 + 8 * piece) = v;
    }
    // the next tile's image window (its pixels were requested a tile ago), then the request for
    // the tile after it
    write_iw();
    prefetch(tile + 2 * gridDim.x);

    // conv1_2: K loop and pooled epilogue of conv3x3_kernel<64, 64, 4>
    const unsigned short* wb = win + ((MT * part) * WC + r) * PIX + 8 * h;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero16();
    constexpr int WRW = MT + 2, NST = Cfg::SPT * WRW * 3, RING = 4, AHEAD = 3;
    auto frag_at = [&](int st) {
      const int c = st / (WRW * 3), wr = (st / 3) % WRW, kw = st % 3;
      return *reinterpret_cast<const u32x4*>(wb + (wr * WC + kw) * PIX + 16 * c);
    };
    u32x4 af[RING];
#pragma unroll
    for (int st = 0; st < AHEAD; ++st) af[st] = frag_at(st);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (st + AHEAD < NST) af[(st + AHEAD) % RING] = frag_at(st + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      const int c = st / (WRW * 3), wr = (st / 3) % WRW, kw = st % 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int kh = wr - mt;
        if (kh >= 0 && kh < 3)
          acc[mt] = mfma32b(af[st % RING], wf[(3 * kh + kw) * Cfg::SPT + c], acc[mt]);
      }
    }

    const int oy0 = ty + MT * part, ox0 = tx;
    const int PH2 = H / 2, PW2 = W / 2;
    unsigned char* scr8 = reinterpret_cast<unsigned char*>(scr + 16 * SCR_LD);   // [16][32]
#pragma unroll
    for (int q = 0; q < 16; q += 2) {
      const float m0 = fmaxf(acc[0][q], acc[0][q + 1]);
      const float m1 = fmaxf(acc[1][q], acc[1][q + 1]);
      const float m = fmaxf(m0, m1);
      scr[(acc_row(q, h) >> 1) * SCR_LD + r] = f32_to_bf16(fmaxf(m + bias_r, 0.f));
      // first maximum in raster order (0,0), (0,1), (1,0), (1,1)
      const int k = acc[0][q] == m ? 0 : acc[0][q + 1] == m ? 1 : acc[1][q] == m ? 2 : 3;
      scr8[(acc_row(q, h) >> 1) * 32 + r] = (unsigned char)k;
    }
    __builtin_amdgcn_wave_barrier();
    {
      // 16 pooled pixels x 64 bytes: lane -> pixel lane >> 2, 16-byte quarter lane & 3
      const int px = lane >> 2, qu = lane & 3;
      const u32x4 v = *reinterpret_cast<const u32x4*>(scr + px * SCR_LD + 8 * qu);
      const uint2 kv = *reinterpret_cast<const uint2*>(scr8 + px * 32 + 8 * qu);
      __builtin_amdgcn_wave_barrier();
      const int py = oy0 >> 1, pxg = (ox0 >> 1) + px;
      if (py < PH2 && pxg < PW2) {
        const int64_t po = (((int64_t)b * PH2 + py) * PW2 + pxg) * C64 + 32 * nt + 8 * qu;
        *reinterpret_cast<u32x4*>(pooled + po) = v;
        *reinterpret_cast<uint2*>(pidx + po) = kv;
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------
// conv_first_wrw_kernel: weight AND bias gradient of the first layer in ONE pass over its
// 64-channel gradient map (the layer's only large operand: 944 MB at 24 x 480 x 640),
//   gw[k][c][kh][kw] = sum_p gz[p][k] * x0[p + (kh-1, kw-1)][c],     gb[k] = sum_p gz[p][k],
// as a GEMM over pixels: A = gz^T (64 channels = two m-tiles, read with ds_read_b64_tr_b16
// from the two LDS-DMA planes of wrw64_kernel), B[p][n] = im2col of x0 gathered from a
// [10][34][3+1] halo window, n = 3 * tap + c < 27, column 27 = 1.0 (its product is the bias
// gradient), columns 28 .. 31 = the indicators of the image's first / last row and first /
// last column (their products are the border sums the closed-form gradient of the trainable
// mean needs, conv_first_davg_kernel).  Wave w takes tile rows 2w, 2w + 1; per-workgroup
// slabs [64][32] float32 are summed in a fixed order by conv_first_wrw_reduce_kernel.
constexpr int FW_WIN = WR * WC * F_PIX;               // bf16 per x0 window (1360)
constexpr int FW_GBUF = 2 * GPLANE;                   // bf16 per staged gz tile (16,384)
constexpr size_t kFirstWrwLds = (2 * (size_t)FW_GBUF + 2 * (size_t)FW_WIN) * sizeof(unsigned short);

__global__ __launch_bounds__(256, 2) void conv_first_wrw_kernel(
    const unsigned short* __restrict__ x0, const unsigned short* __restrict__ gz, int B, int H,
    int W, float* __restrict__ slabs) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  unsigned short* wl = lds + 2 * FW_GBUF;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q = (lane >> 2) & 3, pp = lane & 3, gq = lane >> 4;
  const int pix0 = 8 * (gq >> 1) + q, ch0 = 16 * (gq & 1) + 4 * pp;
  // this lane's im2col column: tap (kh, kw), input channel c
  const int tap = r / 3, cc = r % 3, kh = tap / 3, kw = tap % 3;
  const unsigned ones = r == 27 ? 0x3f803f80u : 0u;

  const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
  const int per_img = tiles_x * tiles_y;
  const int ntiles = B * per_img;
  const unsigned short* zeros = reinterpret_cast<const unsigned short*>(zero_block);
  const int wid_s = __builtin_amdgcn_readfirstlane(wid);

  // gz tile: 32 chunks of 1 KB (2 planes x 16), eight per wave
  auto gz_issue = [&](int tile, int buf) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;
    const unsigned base = lds_byte_of(lds) + buf * FW_GBUF * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = wid_s + 4 * i, plane = j >> 4, chunk = j & 15;
      const int pix = 16 * chunk + (lane >> 2);
      const int y = ty + pix / TW, xx = tx + pix % TW;
      const bool ok = y < H && xx < W;
      const int off = ((b * H + y) * W + xx) * C64 + 32 * plane + 8 * (lane & 3);
      glds16(ok ? gz + off : zeros, base + (plane * GPLANE + chunk * 512) * 2);
    }
  };
  // x0 window: 340 pixels x 3 channels, four 2-byte elements per thread
  unsigned short wreg[4];
  auto win_load = [&](int tile) {
    const int b = tile / per_img, t2 = tile % per_img;
    const int ty = (t2 / tiles_x) * TH, tx = (t2 % tiles_x) * TW;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int pix = idx / 3, c = idx - 3 * pix;
      const int y = ty - 1 + pix / WC, xx = tx - 1 + pix % WC;
      const bool ok = idx < 3 * WR * WC && y >= 0 && y < H && xx >= 0 && xx < W;
      wreg[v] = ok ? x0[(((int64_t)b * H + y) * W + xx) * 3 + c] : (unsigned short)0;
    }
  };
  auto win_store = [&](int buf) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int idx = v * 256 + threadIdx.x;
      const int pix = idx / 3, c = idx - 3 * pix;
      if (idx < 3 * WR * WC) wl[buf * FW_WIN + pix * F_PIX + c] = wreg[v];
    }
  };

  f32x16 acc[2] = {zero16(), zero16()};
  int tile = blockIdx.x;
  if (tile < ntiles) {
    gz_issue(tile, 0);
    win_load(tile);
    win_store(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    if (next < ntiles) {
      gz_issue(next, buf ^ 1);
      win_load(next);
    }
    const unsigned short* gl = lds + buf * FW_GBUF + pix0 * WPL + ch0;
    const unsigned short* wb = wl + buf * FW_WIN + (kh * WC + kw + 8 * h) * F_PIX + cc;
    const int t2c = tile % per_img;
    const int ty = (t2c / tiles_x) * TH, tx = (t2c % tiles_x) * TW;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s_ = 4 * wid + u, ry = s_ >> 1, cx = 16 * (s_ & 1);
      const u32x4 a0 = tr_pair(gl + (ry * TW + cx) * WPL, 4 * WPL);
      const u32x4 a1 = tr_pair(gl + GPLANE + (ry * TW + cx) * WPL, 4 * WPL);
      const unsigned short* wp = wb + (ry * WC + cx) * F_PIX;
      u32x4 bf;
      if (r < 27) {
        bf.x = (unsigned)wp[0 * F_PIX] | ((unsigned)wp[1 * F_PIX] << 16);
        bf.y = (unsigned)wp[2 * F_PIX] | ((unsigned)wp[3 * F_PIX] << 16);
        bf.z = (unsigned)wp[4 * F_PIX] | ((unsigned)wp[5 * F_PIX] << 16);
        bf.w = (unsigned)wp[6 * F_PIX] | ((unsigned)wp[7 * F_PIX] << 16);
      } else if (r == 27) {
        bf = u32x4{ones, ones, ones, ones};
      } else {
        // border indicators of pixels (ty + ry, tx + cx + 8 h + j), j = 0 .. 7
        const int y = ty + ry, x0c = tx + cx + 8 * h;
        const unsigned rowhit = (r == 28 ? y == 0 : r == 29 ? y == H - 1 : false) ? 0x3f80u : 0u;
        const int xc = r == 30 ? 0 : r == 31 ? W - 1 : -1;        // column whose pixels count
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (r < 30) ? rowhit : (x0c + j == xc ? 0x3f80u : 0u);
        bf = u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
      }
      acc[0] = mfma32b(a0, bf, acc[0]);
      acc[1] = mfma32b(a1, bf, acc[1]);
    }
    if (next < ntiles) win_store(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }

  // fixed-order sum of the four waves' partials through LDS (the staging buffers are free)
  float* red = reinterpret_cast<float*>(lds);            // [4][64][32]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int qq = 0; qq < 16; ++qq)
      red[(wid * 64 + 32 * mt + acc_row(qq, h)) * 32 + r] = acc[mt][qq];
  __syncthreads();
  float* out = slabs + (int64_t)blockIdx.x * 2048;
  for (int e = threadIdx.x; e < 2048; e += 256)
    out[e] = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
}

// gw / gb from the slabs: grid 32, block 256 = 64 elements x 4 slab groups; element
// e = (k, n); n < 27: weight tap (kh, kw), input channel c, written as bf16 at the weight's
// strides; n == 27: bias gradient.  Fixed summation order.
__global__ __launch_bounds__(256) void conv_first_wrw_reduce_kernel(
    const float* __restrict__ slabs, int nslab, int64_t sk, int64_t sc, int64_t sh, int64_t sw,
    void* __restrict__ gw, int w_f32, float* __restrict__ gb, float* __restrict__ aux) {
  __shared__ float red[4][64];
  const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + j;
  float s0 = 0.f, s1 = 0.f;
  int i = g;
  for (; i + 4 < nslab; i += 8) {
    s0 += slabs[(int64_t)i * 2048 + e];
    s1 += slabs[(int64_t)(i + 4) * 2048 + e];
  }
  if (i < nslab) s0 += slabs[(int64_t)i * 2048 + e];
  red[g][j] = s0 + s1;
  __syncthreads();
  if (g == 0) {
    const float s = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    const int k = e >> 5, n = e & 31;
    if (n < 27) {
      const int tap = n / 3, c = n % 3;
      store_weight_grad(gw, k * sk + c * sc + (tap / 3) * sh + (tap % 3) * sw, s, w_f32);
    } else if (n == 27) {
      gb[k] = s;
    } else {
      aux[(n - 28) * 64 + k] = s;      // first row, last row, first column, last column sums
    }
  }
}

// Gradient of the trainable mean (model/nets.py:22-24) without the first layer's backward-data
// pass: x0 = img - avg feeds a 3x3 same-padding conv, so
//   d loss / d avg[c] = - sum_{o,kh,kw} w[o][c][kh][kw] * S[o][kh][kw],
//   S = (sum of gz over the positions whose tap stays inside the image)
//     = T[o] - R_kh[o] - C_kw[o] + X_khkw[o]
// with T the bias gradient, R / C the first or last row / column sums (aux) and X the four
// corner pixels (read here).  One block of 64 threads, thread = output channel; fixed order.
__global__ __launch_bounds__(64) void conv_first_davg_kernel(
    const unsigned short* __restrict__ gz, const void* __restrict__ w, int w_f32, int64_t sk,
    int64_t sc, int64_t sh, int64_t sw, const float* __restrict__ gb,
    const float* __restrict__ aux, int B, int H, int W, float* __restrict__ davg, int compact) {
  __shared__ float part[3][64];
  const int o = threadIdx.x;
  float corner[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  // (unrolled by 8: the loads of eight images go out together — one image per round trip made this
  // 64-thread kernel 24 + 27 dependent loads long; the sums keep their order)
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    if (compact) {           // gz = [B][4 corners: (0,0) (0,W-1) (H-1,0) (H-1,W-1)][64] — the fused
                             // conv1_2 backward's side buffer (H, W >= 2 there: four distinct pixels)
      const unsigned short* c4 = gz + (int64_t)b * 4 * C64 + o;
      corner[0][0] += bf16_to_f32(c4[0 * C64]);
      corner[0][2] += bf16_to_f32(c4[1 * C64]);
      corner[2][0] += bf16_to_f32(c4[2 * C64]);
      corner[2][2] += bf16_to_f32(c4[3 * C64]);
      continue;
    }
    const int64_t img = (int64_t)b * H * W;
    corner[0][0] += bf16_to_f32(gz[(img + 0) * C64 + o]);
    corner[0][2] += bf16_to_f32(gz[(img + W - 1) * C64 + o]);
    corner[2][0] += bf16_to_f32(gz[(img + (int64_t)(H - 1) * W) * C64 + o]);
    corner[2][2] += bf16_to_f32(gz[(img + (int64_t)(H - 1) * W + W - 1) * C64 + o]);
  }
  const float rows[3] = {aux[0 * 64 + o], 0.f, aux[1 * 64 + o]};
  const float cols[3] = {aux[2 * 64 + o], 0.f, aux[3 * 64 + o]};
  float acc[3] = {0.f, 0.f, 0.f};
  float wv[3][3][3];                   // the 27 weights of channel o: independent loads, one branch
  if (w_f32) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          wv[kh][kw][c] = static_cast<const float*>(w)[o * sk + c * sc + kh * sh + kw * sw];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 3; ++c) wv[kh][kw][c] = bf16_to_f32(f32_to_bf16(wv[kh][kw][c]));
  } else {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          wv[kh][kw][c] =
              bf16_to_f32(static_cast<const unsigned short*>(w)[o * sk + c * sc + kh * sh + kw * sw]);
  }
  const float gbo = gb[o];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float s_ = gbo - rows[kh] - cols[kw] + corner[kh][kw];
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wv[kh][kw][c] * s_;
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) part[c][o] = acc[c];
  __syncthreads();
  if (o < 3) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += part[o][i];
    davg[o] = -t;
  }
}

}  // namespace

static int conv64_cus();

// conv1_2 with two 4-wave workgroups per CU (ConvCfg GEO 1) — DIAGNOSTIC BUILD, SCL_CONV64_TWO_WG=1.
// Round 6 measured it (scripts/conv12_geo_ab.py, profiles/r06/conv12_two_workgroups_per_cu.txt):
// bit-identical, forward 488 -> 478 us, backward-data with the un-pooling window 555 -> 686 us,
// bias + ReLU forward 510 -> 510.  Taking the tile barrier out from between the two waves of a
// SIMD does not speed the K loop up: the 8,100 cycles per tile are not the phase lock DESIGN.md
// section 7 suspected, so the product keeps the one 8-wave workgroup.
static bool conv64_two_wg() {
#ifdef SCL_DIAG
  static const bool on = [] {
    const char* e = getenv("SCL_CONV64_TWO_WG");
    return e && e[0] == '1';
  }();
  return on;
#else
  return false;
#endif
}

template <int CIN, int KOUT, int GEO = 0>
int launch_conv3x3(const void* x, const void* w, int64_t sk, int64_t sc, int64_t sh, int64_t sw,
                   int transposed, int B, int H, int W, void* out, const float* bias, int relu,
                   void* pooled, const void* mask, void* pidx, const void* uidx, void* workspace,
                   hipStream_t st) {
  using Cfg = ConvCfg<CIN, KOUT, GEO>;
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, KOUT, 0, 0, 0, GEO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, KOUT, 1, 0, 0, GEO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, KOUT, 2, 0, 0, GEO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, KOUT, 3, 0, 0, GEO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, KOUT, 4, 0, 0, GEO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
    if (CIN == KOUT)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<CIN, CIN, 3, 1, 0, GEO>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS);
  });
  const int cus = conv64_cus() * (GEO ? 2 : 1);
  const unsigned short* packed = (const unsigned short*)workspace;
  if (transposed & SCL_W_PACKED)
    packed = (const unsigned short*)w;                   // scl_conv_pack_batch wrote it
  else
    SCL_LAUNCH("conv3x3_pack_kernel", (conv3x3_pack_kernel<CIN, KOUT>),
               dim3(Cfg::NT * Cfg::KS * 512 / 256), dim3(256), 0, st, w, sk, sc, sh, sw, transposed,
               (unsigned short*)workspace);
  const int tiles = B * ((H + Cfg::TH_ - 1) / Cfg::TH_) * ((W + TW - 1) / TW);
  const dim3 grid(tiles < cus ? tiles : cus);
  if (scl_variant() / 1000 == 60) relu |= (scl_variant() & 7) << 1;   // bit 2: setprio experiment
  if (pidx)
    SCL_LAUNCH("conv3x3_kernel", (conv3x3_kernel<CIN, KOUT, 4, 0, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS, st,
               (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)nullptr, bias, relu & ~1, (unsigned short*)pooled,
               (const unsigned short*)nullptr, (unsigned char*)pidx, (const unsigned char*)nullptr);
  else if (mask && uidx && CIN == KOUT)
    SCL_LAUNCH("conv3x3_kernel<pooled>", (conv3x3_kernel<CIN, CIN, 3, 1, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS,
               st, (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)out, bias, relu & ~1, (unsigned short*)nullptr,
               (const unsigned short*)mask, (unsigned char*)nullptr, (const unsigned char*)uidx);
  else if (mask)
    SCL_LAUNCH("conv3x3_kernel", (conv3x3_kernel<CIN, KOUT, 3, 0, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS, st,
               (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)out, bias, relu & ~1, (unsigned short*)nullptr,
               (const unsigned short*)mask, (unsigned char*)nullptr, (const unsigned char*)nullptr);
  else if (pooled)
    SCL_LAUNCH("conv3x3_kernel", (conv3x3_kernel<CIN, KOUT, 2, 0, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS, st,
               (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)out, bias, relu, (unsigned short*)pooled,
               (const unsigned short*)nullptr, (unsigned char*)nullptr, (const unsigned char*)nullptr);
  else if (bias)
    SCL_LAUNCH("conv3x3_kernel", (conv3x3_kernel<CIN, KOUT, 1, 0, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS, st,
               (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)out, bias, relu, (unsigned short*)pooled,
               (const unsigned short*)nullptr, (unsigned char*)nullptr, (const unsigned char*)nullptr);
  else
    SCL_LAUNCH("conv3x3_kernel", (conv3x3_kernel<CIN, KOUT, 0, 0, 0, GEO>), grid, dim3(Cfg::NTHR), Cfg::LDS, st,
               (const unsigned short*)x, (const unsigned short*)packed, B, H, W,
               (unsigned short*)out, bias, relu, (unsigned short*)pooled,
               (const unsigned short*)nullptr, (unsigned char*)nullptr, (const unsigned char*)nullptr);
  return scl_launch_status();
}

extern "C" size_t scl_conv3x3_workspace_bytes(void) {
  return scl_round256((size_t)4 * 72 * 512 * sizeof(unsigned short));   // largest packed image
}

static int conv3x3_dispatch(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                            int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H,
                            int W, int cin, int kout, void* out, const float* bias, int relu,
                            void* pooled, const void* mask, void* pidx, void* workspace,
                            size_t workspace_bytes, void* stream, const void* uidx = nullptr) {
  if (!x || !w || (!out && !pidx) || !workspace) return SCL_E_NULL;
  // un-pooling window staging: the masked backward-data pass of a cin == kout layer, even H, W
  if (uidx && (!mask || cin != kout || ((H | W) & 1) || ((uintptr_t)uidx % 8))) return SCL_E_SHAPE;
  if (pidx && (!pooled || !bias || mask || ((uintptr_t)pidx % 8))) return SCL_E_NULL;
  if (mask && (bias || pooled || ((uintptr_t)mask % 16))) return SCL_E_NULL;
  if ((int64_t)B * H * W * cin >= (int64_t)1 << 31) return SCL_E_SHAPE;   // 32-bit element offsets
  if (pooled && (!bias || ((uintptr_t)pooled % 16))) return SCL_E_NULL;
  if (B < 1 || H < 1 || W < 1 || (int64_t)B * H * W > (int64_t)1 << 30) return SCL_E_SHAPE;
  if (((uintptr_t)x % 16) || (out && ((uintptr_t)out % 16))) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_conv3x3_workspace_bytes())
    return SCL_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
#define SCL_CONV_CASE(CI, KO)                                                                  \
  if (cin == CI && kout == KO)                                                                 \
    return launch_conv3x3<CI, KO>(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w,        \
                                  transposed, B, H, W, out, bias, relu ? 1 : 0, pooled,        \
                                  mask, pidx, uidx, workspace, st);
#ifdef SCL_DIAG
  if (cin == 64 && kout == 64 && conv64_two_wg())
    return launch_conv3x3<64, 64, 1>(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, transposed, B, H,
                                     W, out, bias, relu ? 1 : 0, pooled, mask, pidx, uidx, workspace, st);
#endif
  SCL_CONV_CASE(64, 64)
  SCL_CONV_CASE(64, 128)
  SCL_CONV_CASE(128, 64)
  SCL_CONV_CASE(128, 128)
#undef SCL_CONV_CASE
  return SCL_E_SHAPE;
}

extern "C" int scl_conv3x3_fused(const void* x, const void* w, int64_t w_stride_k,
                                 int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                                 int transposed, int B, int H, int W, int cin, int kout,
                                 void* out, const float* bias, int relu, void* pooled,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return conv3x3_dispatch(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, transposed, B, H,
                          W, cin, kout, out, bias, relu, pooled, nullptr, nullptr, workspace,
                          workspace_bytes, stream);
}

extern "C" int scl_conv3x3_pool_idx(const void* x, const void* w, int64_t w_stride_k,
                                    int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                                    int flags, int B, int H, int W, int cin, int kout,
                                    const float* bias, void* pooled, void* pool_idx,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!pool_idx || !pooled || !bias) return SCL_E_NULL;
  return conv3x3_dispatch(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, flags & 6, B, H, W, cin,
                          kout, nullptr, bias, 0, pooled, nullptr, pool_idx, workspace,
                          workspace_bytes, stream);
}

extern "C" int scl_conv3x3_masked(const void* x, const void* w, int64_t w_stride_k,
                                  int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                                  int transposed, int B, int H, int W, int cin, int kout,
                                  void* out, const void* mask, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (!mask) return SCL_E_NULL;
  return conv3x3_dispatch(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, transposed, B, H,
                          W, cin, kout, out, nullptr, 0, nullptr, mask, nullptr, workspace,
                          workspace_bytes, stream);
}

extern "C" int scl_conv3x3_masked_pooled(const void* g_pooled, const void* pool_idx, const void* w,
                                         int64_t w_stride_k, int64_t w_stride_c,
                                         int64_t w_stride_h, int64_t w_stride_w, int flags, int B,
                                         int H, int W, int cin, int kout, void* out,
                                         const void* mask, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!mask || !pool_idx) return SCL_E_NULL;
  return conv3x3_dispatch(g_pooled, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, flags, B, H,
                          W, cin, kout, out, nullptr, 0, nullptr, mask, nullptr, workspace,
                          workspace_bytes, stream, pool_idx);
}

extern "C" int scl_conv3x3(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                           int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H,
                           int W, int cin, int kout, void* out, void* workspace,
                           size_t workspace_bytes, void* stream) {
  return scl_conv3x3_fused(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, transposed, B, H,
                           W, cin, kout, out, nullptr, 0, nullptr, workspace, workspace_bytes,
                           stream);
}

extern "C" size_t scl_conv64_workspace_bytes(void) { return scl_conv3x3_workspace_bytes(); }

extern "C" int scl_conv64(const void* x, const void* w, int64_t w_stride_k, int64_t w_stride_c,
                          int64_t w_stride_h, int64_t w_stride_w, int transposed, int B, int H,
                          int W, void* out, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return scl_conv3x3(x, w, w_stride_k, w_stride_c, w_stride_h, w_stride_w, transposed, B, H, W,
                     64, 64, out, workspace, workspace_bytes, stream);
}

// CUs the persistent grids may fill: the hardware's count (asked once), minus the reserve AS IT
// IS NOW — scl_set_reserve_cus may change between calls (bench.py tries 0 and 8 at N > 1)
static int conv64_cus() {
  const int n = scl_device_cus();      // per device (scl_common.h)
  const int u = scl_usable_cus(n);
  return u > 1024 ? 1024 : u;
}

static int wrw_splits(int C, int K, int tiles, int cus) {
  const int blocks = (C / 64) * (K / 64);
  int p = (cus + blocks - 1) / blocks;                 // about one workgroup per CU
  if (p > tiles) p = tiles;
  if (p > 1024) p = 1024;
  return p < 1 ? 1 : p;
}

extern "C" size_t scl_wrw3x3_workspace_bytes(int cin, int kout) {
  if (cin < 64 || kout < 64 || cin % 64 || kout % 64 || cin > 1024 || kout > 1024) return 0;
  // slabs: at most (1024 CUs rounded up to whole (cb, kb) block sets) x 147,456 B
  const size_t blocks = (size_t)(cin / 64) * (kout / 64);
  const size_t p = (1024 + blocks - 1) / blocks;
  // two slabs per workgroup + the bias partials [slab][kout] of scl_wrw3x3_bias
  return scl_round256(2 * p * blocks * 9 * 64 * 64 * sizeof(float)) +
         scl_round256((2 * p + 2) * (size_t)kout * sizeof(float));
}
static size_t wrw_bias_slab_offset(int cin, int kout) {
  const size_t blocks = (size_t)(cin / 64) * (kout / 64);
  const size_t p = (1024 + blocks - 1) / blocks;
  return scl_round256(2 * p * blocks * 9 * 64 * 64 * sizeof(float));
}

// pidx != nullptr: gz is the pooled gradient [B][H/2][W/2][kout], pidx its window positions
static int wrw3x3_run(const void* x, const void* gz, const unsigned char* pidx, int B, int H, int W,
                      int cin, int kout, void* gw, int64_t w_stride_k, int64_t w_stride_c,
                      int64_t w_stride_h, int64_t w_stride_w, int gw_f32, float* grad_bias,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !gz || !gw || !workspace) return SCL_E_NULL;
  if (pidx && ((H | W) & 1 || (uintptr_t)pidx % 8)) return SCL_E_SHAPE;
  const size_t need = scl_wrw3x3_workspace_bytes(cin, kout);
  if (need == 0 || B < 1 || H < 1 || W < 1 || (int64_t)B * H * W > (int64_t)1 << 30)
    return SCL_E_SHAPE;
  if ((int64_t)B * H * W * (cin > kout ? cin : kout) >= (int64_t)1 << 31)
    return SCL_E_SHAPE;                                 // 32-bit element offsets in the kernel
  if (((uintptr_t)x % 16) || ((uintptr_t)gz % 16)) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < need) return SCL_E_WORKSPACE;
  static SclDeviceOnce once;
  scl_call_once(once, [] {
#define SCL_WRW_ATTR(D, T, N, PL)                                                              \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wrw64_kernel<D, T, N, PL>),         \
                            hipFuncAttributeMaxDynamicSharedMemorySize,                        \
                            (int)WrwCfg<T, N>::LDS);
    SCL_WRW_ATTR(0, 32, 1, 0) SCL_WRW_ATTR(2, 32, 1, 0)
    SCL_WRW_ATTR(0, 8, 1, 0) SCL_WRW_ATTR(0, 32, 2, 0) SCL_WRW_ATTR(0, 8, 2, 0)
    SCL_WRW_ATTR(0, 32, 1, 1) SCL_WRW_ATTR(0, 8, 1, 1) SCL_WRW_ATTR(0, 32, 2, 1) SCL_WRW_ATTR(0, 8, 2, 1)
    SCL_WRW_ATTR(4, 32, 2, 0) SCL_WRW_ATTR(4, 8, 2, 0) SCL_WRW_ATTR(4, 32, 2, 1) SCL_WRW_ATTR(4, 8, 2, 1)
    SCL_WRW_ATTR(6, 32, 2, 0) SCL_WRW_ATTR(6, 8, 2, 0)
#ifdef SCL_DIAG
    SCL_WRW_ATTR(8, 32, 2, 0) SCL_WRW_ATTR(8, 8, 2, 0) SCL_WRW_ATTR(8, 32, 2, 1) SCL_WRW_ATTR(8, 8, 2, 1)
    SCL_WRW_ATTR(16, 32, 2, 0) SCL_WRW_ATTR(16, 8, 2, 0) SCL_WRW_ATTR(16, 32, 2, 1) SCL_WRW_ATTR(16, 8, 2, 1)
#endif
#undef SCL_WRW_ATTR
  });
  const int cus = conv64_cus();
  // [64 c] x [128 k] blocks wherever the output channels allow (scl_debug_set_variant(2100)
  // pins the 64 x 64 variant); tile shape: wide, or tall where that pads the map less
  const int dbg = scl_variant() / 1000 == 2 ? scl_variant() & 3 : 0;
  const bool stamps = (scl_variant() == 2004 || scl_variant() == 2006) && kout % 128 == 0;   // (needs >= 64 KB of bias slabs)
  const int nkb = (kout % 128 == 0 && dbg == 0 && scl_variant() != 2100) ? 2 : 1;
  const int th_w = nkb == 1 ? 8 : 4, th_t = nkb == 1 ? 32 : 16;
  const int tiles_wide = B * ((H + th_w - 1) / th_w) * ((W + 31) / 32);
  const int tiles_tall = B * ((H + th_t - 1) / th_t) * ((W + 7) / 8);
  const bool tall = tiles_tall < tiles_wide && dbg == 0;
  const int tiles = tall ? tiles_tall : tiles_wide;
  const int P = wrw_splits(cin, kout, tiles, cus);
  hipStream_t st = (hipStream_t)stream;
#define SCL_WRW_LAUNCH(D, T, N, PL)                                                            \
  SCL_LAUNCH(PL ? "wrw64_kernel<pooled>" : "wrw64_kernel", (wrw64_kernel<D, T, N, PL>),        \
             dim3(PP * (cin / 64) * (kout / (64 * N))), dim3(512), (WrwCfg<T, N>::LDS), st,    \
             (const unsigned short*)x, (const unsigned short*)gz, B, H, W, cin, kout,          \
             (float*)workspace, bslabs, pidx)
  float* bslabs = grad_bias ? (float*)((char*)workspace + wrw_bias_slab_offset(cin, kout)) : nullptr;
  int PP = P;
  if (stamps) {
    PP = wrw_splits(cin, kout / 2, tiles, cus);
    bslabs = (float*)((char*)workspace + wrw_bias_slab_offset(cin, kout));
    if ((size_t)PP * (cin / 64) * (kout / 128) * 256 > need - wrw_bias_slab_offset(cin, kout))
      return SCL_E_WORKSPACE;
    if (scl_variant() == 2006) {        // ... without staging after the first tile
      if (tall) SCL_WRW_LAUNCH(6, 8, 2, 0); else SCL_WRW_LAUNCH(6, 32, 2, 0);
    } else if (pidx) {
      if (tall) SCL_WRW_LAUNCH(4, 8, 2, 1); else SCL_WRW_LAUNCH(4, 32, 2, 1);
    } else {
      if (tall) SCL_WRW_LAUNCH(4, 8, 2, 0); else SCL_WRW_LAUNCH(4, 32, 2, 0);
    }
    return scl_launch_status();
  }
#ifdef SCL_DIAG
  if (nkb == 2 && scl_variant() == 2200) {     // round 4's staging (no buffer path): A/B partner, CORRECT results
    PP = wrw_splits(cin, kout / 2, tiles, cus);
    if (pidx) {
      if (tall) SCL_WRW_LAUNCH(8, 8, 2, 1); else SCL_WRW_LAUNCH(8, 32, 2, 1);
    } else {
      if (tall) SCL_WRW_LAUNCH(8, 8, 2, 0); else SCL_WRW_LAUNCH(8, 32, 2, 0);
    }
  } else if (nkb == 2 && scl_variant() == 2300) {   // 16x16x32 timing ablation: RESULTS MEANINGLESS
    PP = wrw_splits(cin, kout / 2, tiles, cus);
    if (pidx) {
      if (tall) SCL_WRW_LAUNCH(16, 8, 2, 1); else SCL_WRW_LAUNCH(16, 32, 2, 1);
    } else {
      if (tall) SCL_WRW_LAUNCH(16, 8, 2, 0); else SCL_WRW_LAUNCH(16, 32, 2, 0);
    }
  } else
#endif
  if (nkb == 2) {
    PP = wrw_splits(cin, kout / 2, tiles, cus);
    if (pidx) {
      if (tall) SCL_WRW_LAUNCH(0, 8, 2, 1); else SCL_WRW_LAUNCH(0, 32, 2, 1);
    } else {
      if (tall) SCL_WRW_LAUNCH(0, 8, 2, 0); else SCL_WRW_LAUNCH(0, 32, 2, 0);
    }
  } else if (pidx) {
    if (tall) SCL_WRW_LAUNCH(0, 8, 1, 1); else SCL_WRW_LAUNCH(0, 32, 1, 1);
  } else if (tall) SCL_WRW_LAUNCH(0, 8, 1, 0);
  else if (dbg == 0) SCL_WRW_LAUNCH(0, 32, 1, 0);
  else SCL_WRW_LAUNCH(2, 32, 1, 0);
#undef SCL_WRW_LAUNCH
  const int nslab = nkb == 1 ? 2 * PP : PP, nblk = (cin / 64) * (kout / 64);
#define SCL_WRW_REDUCE(RG)                                                                     \
  SCL_LAUNCH("wrw64_reduce_kernel", wrw64_reduce_kernel<RG>, dim3(9 * 64 * 64 / 256, nblk),    \
             dim3(64 * RG), 0, st, (const float*)workspace, nslab, kout / 64, w_stride_k,      \
             w_stride_c, w_stride_h, w_stride_w, gw, gw_f32 ? 1 : 0, (const float*)bslabs,     \
             grad_bias)
  if (nblk <= 4 && nslab >= 32) SCL_WRW_REDUCE(16); else SCL_WRW_REDUCE(4);
#undef SCL_WRW_REDUCE
  return scl_launch_status();
}

extern "C" int scl_wrw3x3_bias(const void* x, const void* gz, int B, int H, int W, int cin,
                               int kout, void* gw, int64_t w_stride_k, int64_t w_stride_c,
                               int64_t w_stride_h, int64_t w_stride_w, int gw_f32,
                               float* grad_bias, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return wrw3x3_run(x, gz, nullptr, B, H, W, cin, kout, gw, w_stride_k, w_stride_c, w_stride_h,
                    w_stride_w, gw_f32, grad_bias, workspace, workspace_bytes, stream);
}

extern "C" int scl_wrw3x3_pooled(const void* x, const void* g_pooled, const void* pool_idx, int B,
                                 int H, int W, int cin, int kout, void* gw, int64_t w_stride_k,
                                 int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                                 int gw_f32, float* grad_bias, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!pool_idx) return SCL_E_NULL;
  return wrw3x3_run(x, g_pooled, (const unsigned char*)pool_idx, B, H, W, cin, kout, gw,
                    w_stride_k, w_stride_c, w_stride_h, w_stride_w, gw_f32, grad_bias, workspace,
                    workspace_bytes, stream);
}

extern "C" int scl_wrw3x3_ex(const void* x, const void* gz, int B, int H, int W, int cin, int kout,
                             void* gw, int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                             int64_t w_stride_w, int gw_f32, void* workspace,
                             size_t workspace_bytes, void* stream) {
  return scl_wrw3x3_bias(x, gz, B, H, W, cin, kout, gw, w_stride_k, w_stride_c, w_stride_h,
                         w_stride_w, gw_f32, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int scl_wrw3x3(const void* x, const void* gz, int B, int H, int W, int cin, int kout,
                          void* gw, int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                          int64_t w_stride_w, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return scl_wrw3x3_ex(x, gz, B, H, W, cin, kout, gw, w_stride_k, w_stride_c, w_stride_h,
                       w_stride_w, 0, workspace, workspace_bytes, stream);
}

extern "C" size_t scl_wrw64_workspace_bytes(void) { return scl_wrw3x3_workspace_bytes(64, 64); }

extern "C" int scl_wrw64(const void* x, const void* gz, int B, int H, int W, void* gw,
                         int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                         int64_t w_stride_w, void* workspace, size_t workspace_bytes,
                         void* stream) {
  return scl_wrw3x3(x, gz, B, H, W, 64, 64, gw, w_stride_k, w_stride_c, w_stride_h, w_stride_w,
                    workspace, workspace_bytes, stream);
}

extern "C" int scl_conv_first(const float* img, const float* avg, const void* w, int64_t w_stride_k,
                              int64_t w_stride_c, int64_t w_stride_h, int64_t w_stride_w,
                              int w_f32, const float* bias, int B, int H, int W, void* x0,
                              void* y, void* stream) {
  if (!img || !avg || !w || !bias || !x0 || !y) return SCL_E_NULL;
  if (B < 1 || H < 1 || W < 1 || (int64_t)B * H * W > (int64_t)1 << 30) return SCL_E_SHAPE;
  if ((uintptr_t)y % 16) return SCL_E_SHAPE;
  const int cus = conv64_cus();
  const int tiles = B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  // (8 workgroups per CU although four are resident at a time: measured 337-348 us against
  // 351-361 with 4 — the tail is finer)
  const int grid = tiles < 8 * cus ? tiles : 8 * cus;
  SCL_LAUNCH("conv_first_kernel", conv_first_kernel, dim3(grid), dim3(256), kFirstLds,
             (hipStream_t)stream, img, avg, w,
             (w_f32 ? 1 : 0) | (scl_variant() / 1000 == 70 ? (scl_variant() & 7) << 4 : 0),
             w_stride_k, w_stride_c,
             w_stride_h, w_stride_w, bias, B, H, W, (unsigned short*)x0, (unsigned short*)y);
  return scl_launch_status();
}

extern "C" int scl_conv_first_pool_idx(const float* img, const float* avg, const void* w1,
                                       int64_t w1_stride_k, int64_t w1_stride_c,
                                       int64_t w1_stride_h, int64_t w1_stride_w, int w1_f32,
                                       const float* bias1, const void* w2, int64_t w2_stride_k,
                                       int64_t w2_stride_c, int64_t w2_stride_h,
                                       int64_t w2_stride_w, int w2_flags, const float* bias2, int B,
                                       int H, int W, void* x0, void* y1, void* pooled,
                                       void* pool_idx, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  if (!img || !avg || !w1 || !bias1 || !w2 || !bias2 || !x0 || !y1 || !pooled || !pool_idx ||
      !workspace)
    return SCL_E_NULL;
  if (B < 1 || H < 1 || W < 1 || (int64_t)B * H * W * 64 >= (int64_t)1 << 31) return SCL_E_SHAPE;
  if (((uintptr_t)y1 % 16) || ((uintptr_t)pooled % 16) || ((uintptr_t)pool_idx % 8)) return SCL_E_SHAPE;
  if (w2_flags & SCL_CONV_TRANSPOSED) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_conv3x3_workspace_bytes())
    return SCL_E_WORKSPACE;
  using Cfg = ConvCfg<64, 64>;
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kConv12Lds);
  });
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* packed = (const unsigned short*)workspace;
  if (w2_flags & SCL_W_PACKED)
    packed = (const unsigned short*)w2;                  // scl_conv_pack_batch wrote it
  else
    SCL_LAUNCH("conv3x3_pack_kernel", (conv3x3_pack_kernel<64, 64>),
               dim3(Cfg::NT * Cfg::KS * 512 / 256), dim3(256), 0, st, w2, w2_stride_k, w2_stride_c,
               w2_stride_h, w2_stride_w, w2_flags & SCL_W_F32, (unsigned short*)workspace);
  const int cus = conv64_cus();
  const int tiles = B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  SCL_LAUNCH("conv12_kernel", conv12_kernel, dim3(tiles < cus ? tiles : cus), dim3(512), kConv12Lds,
             st, img, avg, w1, w1_f32 ? 1 : 0, w1_stride_k, w1_stride_c, w1_stride_h, w1_stride_w,
             bias1, packed, bias2, B, H, W, (unsigned short*)x0, (unsigned short*)y1,
             (unsigned short*)pooled, (unsigned char*)pool_idx);
  return scl_launch_status();
}

static int first_wrw_grid(int tiles, int cus) { return tiles < 2 * cus ? tiles : 2 * cus; }

extern "C" size_t scl_conv_first_wrw_workspace_bytes(void) {
  // up to 2048 slabs [64][32] + the four border-sum rows
  return scl_round256(((size_t)2 * 1024 * 2048 + 256) * sizeof(float));
}

extern "C" int scl_conv_first_wrw(const void* x0, const void* gz, int B, int H, int W, void* gw,
                                  int64_t w_stride_k, int64_t w_stride_c, int64_t w_stride_h,
                                  int64_t w_stride_w, int w_f32, float* gb, const void* w,
                                  float* davg, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  if (!x0 || !gz || !gw || !gb || !workspace || (davg && !w)) return SCL_E_NULL;
  if (B < 1 || H < 1 || W < 1 || (int64_t)B * H * W * 64 >= (int64_t)1 << 31) return SCL_E_SHAPE;
  if ((uintptr_t)gz % 16) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_conv_first_wrw_workspace_bytes())
    return SCL_E_WORKSPACE;
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_first_wrw_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFirstWrwLds);
  });
  const int cus = conv64_cus();
  const int tiles = B * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
  const int grid = first_wrw_grid(tiles, cus);
  hipStream_t st = (hipStream_t)stream;
  float* aux = (float*)workspace + (size_t)2 * 1024 * 2048;
  SCL_LAUNCH("conv_first_wrw_kernel", conv_first_wrw_kernel, dim3(grid), dim3(256), kFirstWrwLds,
             st, (const unsigned short*)x0, (const unsigned short*)gz, B, H, W, (float*)workspace);
  SCL_LAUNCH("conv_first_wrw_reduce_kernel", conv_first_wrw_reduce_kernel, dim3(32), dim3(256), 0,
             st, (const float*)workspace, grid, w_stride_k, w_stride_c, w_stride_h, w_stride_w,
             gw, w_f32 ? 1 : 0, gb, aux);
  if (davg)      // w has the strides of gw (the layer's bf16 weight)
    SCL_LAUNCH("conv_first_davg_kernel", conv_first_davg_kernel, dim3(1), dim3(64), 0, st,
               (const unsigned short*)gz, w, w_f32 ? 1 : 0, w_stride_k, w_stride_c, w_stride_h,
               w_stride_w, (const float*)gb, (const float*)aux, B, H, W, davg, 0);
  return scl_launch_status();
}

// conv1_2's backward-data pass WITHOUT its output, the first layer's weight / bias / mean gradient
// out of the same launch (conv3x3_kernel<64, 64, 3, 1, FW = 1>; then the two small kernels of
// scl_conv_first_wrw).  g_pooled / pool_idx / w2 / mask as scl_conv3x3_masked_pooled (64 -> 64
// channels, H and W even); x0 / gw1 / gb1 / w1 / davg / fw_workspace as scl_conv_first_wrw.
extern "C" int scl_conv3x3_masked_pooled_first_wrw(
    const void* g_pooled, const void* pool_idx, const void* w2, int64_t w2_stride_k, int64_t w2_stride_c,
    int64_t w2_stride_h, int64_t w2_stride_w, int flags, int B, int H, int W, const void* mask,
    const void* x0, void* gw1, int64_t w1_stride_k, int64_t w1_stride_c, int64_t w1_stride_h,
    int64_t w1_stride_w, int w1_f32, float* gb1, const void* w1, float* davg, void* workspace,
    size_t workspace_bytes, void* fw_workspace, size_t fw_workspace_bytes, void* stream) {
  if (!g_pooled || !pool_idx || !w2 || !mask || !x0 || !gw1 || !gb1 || !workspace || !fw_workspace ||
      (davg && !w1))
    return SCL_E_NULL;
  if (B < 1 || B > 8192 || H < 2 || W < 2 || ((H | W) & 1) ||
      (int64_t)B * H * W * 64 >= (int64_t)1 << 31)
    return SCL_E_SHAPE;
  if (((uintptr_t)g_pooled % 16) || ((uintptr_t)mask % 16) || ((uintptr_t)pool_idx % 8)) return SCL_E_SHAPE;
  if (!scl_aligned256(workspace) || workspace_bytes < scl_conv3x3_workspace_bytes()) return SCL_E_WORKSPACE;
  if (!scl_aligned256(fw_workspace) || fw_workspace_bytes < scl_conv_first_wrw_workspace_bytes())
    return SCL_E_WORKSPACE;
  using Cfg = ConvCfg<64, 64>;
  constexpr size_t kLds = Cfg::LDS + (size_t)Cfg::WAVES * 1024 * sizeof(float) +
                          2 * (size_t)WR * WC * 4 * sizeof(unsigned short);
  static SclDeviceOnce once;
  scl_call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<64, 64, 3, 1, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds);
  });
  hipStream_t st = (hipStream_t)stream;
  const int cus = conv64_cus();
  const unsigned short* packed = (const unsigned short*)workspace;
  const int transposed = (flags & (SCL_W_F32 | SCL_W_PACKED)) | SCL_CONV_TRANSPOSED;
  if (flags & SCL_W_PACKED)
    packed = (const unsigned short*)w2;
  else
    SCL_LAUNCH("conv3x3_pack_kernel", (conv3x3_pack_kernel<64, 64>), dim3(Cfg::NT * Cfg::KS * 512 / 256),
               dim3(256), 0, st, w2, w2_stride_k, w2_stride_c, w2_stride_h, w2_stride_w, transposed,
               (unsigned short*)workspace);
  const int tiles = B * ((H + Cfg::TH_ - 1) / Cfg::TH_) * ((W + TW - 1) / TW);
  const int grid = tiles < cus ? tiles : cus;
  float* aux = (float*)fw_workspace + (size_t)2 * 1024 * 2048;
  const unsigned short* corners = (const unsigned short*)((float*)fw_workspace + (size_t)1536 * 2048);
  SCL_LAUNCH("conv3x3_kernel<pooled+first_wrw>", (conv3x3_kernel<64, 64, 3, 1, 1>), dim3(grid),
             dim3(Cfg::NTHR), kLds, st, (const unsigned short*)g_pooled, packed, B, H, W,
             (unsigned short*)nullptr, (const float*)nullptr,
             scl_variant() / 1000 == 61 ? (scl_variant() % 1000) << 1 : 0,     // timing ablations / stamps (diagnostic build)
             (unsigned short*)x0, (const unsigned short*)mask, (unsigned char*)fw_workspace,
             (const unsigned char*)pool_idx);
  SCL_LAUNCH("conv_first_wrw_reduce_kernel", conv_first_wrw_reduce_kernel, dim3(32), dim3(256), 0,
             st, (const float*)fw_workspace, grid, w1_stride_k, w1_stride_c, w1_stride_h, w1_stride_w,
             gw1, w1_f32 ? 1 : 0, gb1, aux);
  if (davg)
    SCL_LAUNCH("conv_first_davg_kernel", conv_first_davg_kernel, dim3(1), dim3(64), 0, st, corners, w1,
               w1_f32 ? 1 : 0, w1_stride_k, w1_stride_c, w1_stride_h, w1_stride_w, (const float*)gb1,
               (const float*)aux, B, H, W, davg, 1);
  return scl_launch_status();
}
