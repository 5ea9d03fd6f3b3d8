// All weight images of a step in ONE launch.
//
// Every 3x3 convolution kernel of this library reads its weights from a packed image that a
// small kernel writes in front of it (conv3x3_pack_kernel in conv64.hip, convh_pack_kernel in
// convh.hip): 24 launches of 4-10 us per training step, each with a kernel boundary on either
// side, on the critical path.  The weights change once per step, so the caller can have all the
// images written up front by scl_conv_pack_batch — one launch — and hand them to the
// convolutions with SCL_W_PACKED.  The two layouts are restated here; the parity tests compare
// a convolution fed this way with the same convolution packing for itself, bit for bit.
#include "scl_common.h"
#include "vlad_planes.h"

namespace {

constexpr int kMaxJobs = 32;
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct PackJobs {
  const void* w[kMaxJobs];
  unsigned short* packed[kMaxJobs];
  int64_t sk[kMaxJobs], sc[kMaxJobs], sh[kMaxJobs], sw[kMaxJobs];
  int flags[kMaxJobs];       // SCL_CONV_TRANSPOSED | SCL_W_F32 | 8: register-weights layout |
                             // SCL_PACK_VLAD_W: the NetVLAD plane images (vlad_planes.h)
  int cin[kMaxJobs], kout[kMaxJobs];
  int first_block[kMaxJobs + 1];   // prefix sums of 256-element blocks
  int n;
};

__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const PackJobs jobs) {
  int job = 0;
  while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[job + 1]) ++job;
  const int64_t idx = (int64_t)(blockIdx.x - jobs.first_block[job]) * 256 + threadIdx.x;
  const int cin = jobs.cin[job], kout = jobs.kout[job], flags = jobs.flags[job];
  if (flags & SCL_PACK_VLAD_W) {
    vlad_planes_wave((const float*)jobs.w[job], jobs.packed[job], jobs.packed[job] + VP_FWD_ELEMS,
                     (int)(idx >> 6), threadIdx.x & 63);
    return;
  }
  const int transposed = flags & 1, wf32 = flags & 2;
  const int64_t total = (int64_t)9 * cin * kout;
  if (idx >= total) return;
  int ci, co, kh, kw;
  if (flags & 8) {
    // conv64.hip: [n-tile kout / 32][k-step 9 * cin / 16][lane 64][8]; lane (j, h): output
    // channel 32 nt + j, contraction index 16 (ks % SPT) + 8 h + e of tap ks / SPT
    const int spt = cin / 16, ks_n = 9 * spt;
    const int e = idx & 7, lane = (idx >> 3) & 63, ks = (int)((idx >> 9) % ks_n);
    const int nt = (int)(idx / ((int64_t)ks_n * 512));
    const int tap = ks / spt;
    kh = tap / 3;
    kw = tap % 3;
    ci = 16 * (ks % spt) + 8 * (lane >> 5) + e;
    co = 32 * nt + (lane & 31);
  } else {
    // convh.hip: [n-block kout / 128][chunk cin / 32][tap 9][piece 4][k 128][8]
    const int e = idx & 7, k = (idx >> 3) & 127, g = (idx >> 10) & 3;
    const int64_t rest = idx >> 12;
    const int tap = rest % 9, cc_n = cin / 32;
    const int cc = (rest / 9) % cc_n, nb = rest / 9 / cc_n;
    kh = tap / 3;
    kw = tap % 3;
    ci = 32 * cc + 8 * g + e;
    co = 128 * nb + k;
  }
  int64_t off;
  if (!transposed)
    off = co * jobs.sk[job] + ci * jobs.sc[job] + kh * jobs.sh[job] + kw * jobs.sw[job];
  else
    off = ci * jobs.sk[job] + co * jobs.sc[job] + (2 - kh) * jobs.sh[job] + (2 - kw) * jobs.sw[job];
  jobs.packed[job][idx] = weight_bf16(jobs.w[job], off, wf32);
}

// The same images from a contiguous OIHW weight (the float32 master, or a contiguous bf16 copy)
// through an LDS tile: the gather above reads 4 (or 2) bytes per lane at a stride of 36 (18)
// bytes and took 255 us per step for the 29 M elements of VGG16 — it runs at every forward pass
// (prepack forces it: an in-place optimizer need not touch a tensor's version), so it matters.
// A workgroup takes a [32 output channels] x [32 contraction channels] x 9 taps patch: 32 source
// rows of 288 contiguous elements in, 1152 units of 16 bytes out, 32 consecutive units (512
// bytes) per destination run in either layout.
__global__ __launch_bounds__(256) void conv_pack_tiled_kernel(const PackJobs jobs) {
  __shared__ unsigned short tile[9][32][40];          // [tap][co][ci + pad]
  int job = 0;
  while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[job + 1]) ++job;
  const int t = blockIdx.x - jobs.first_block[job];
  const int cin = jobs.cin[job], flags = jobs.flags[job];
  if (flags & SCL_PACK_VLAD_W) {     // (uniform over the workgroup)
    vlad_planes_wave((const float*)jobs.w[job], jobs.packed[job], jobs.packed[job] + VP_FWD_ELEMS,
                     4 * t + (threadIdx.x >> 6), threadIdx.x & 63);
    return;
  }
  const int transposed = flags & 1, wf32 = flags & 2;
  const int cibn = cin / 32;
  const int cob = t / cibn, cib = t % cibn;
  // source rows: forward: output channel co (row stride sk), 288 = (ci, tap) contiguous from
  // ci0 * 9; transposed: contraction channel ci = source k (row stride sk), 288 = (co, tap)
  const int64_t sk = jobs.sk[job];
  const int row0 = transposed ? 32 * cib : 32 * cob, col0 = (transposed ? 32 * cob : 32 * cib) * 9;
  for (int e = threadIdx.x; e < 32 * 288; e += 256) {
    const int row = e / 288, col = e - row * 288;
    const unsigned short v = weight_bf16(jobs.w[job], (row0 + row) * sk + col0 + col, wf32);
    const int inner = col / 9, tap = col - inner * 9;
    if (!transposed)
      tile[tap][row][inner] = v;                      // row = co, inner = ci
    else
      tile[8 - tap][inner][row] = v;                  // row = ci, inner = co; taps flipped
  }
  __syncthreads();
  unsigned short* packed = jobs.packed[job];
  for (int u = threadIdx.x; u < 9 * 4 * 32; u += 256) {
    const int kl = u & 31, g = (u >> 5) & 3, tap = u >> 7;       // co_local, 8-channel piece
    const u32x4s v = *reinterpret_cast<const u32x4s*>(&tile[tap][kl][8 * g]);
    int64_t idx;
    if (flags & 8) {
      // conv64.hip: ((nt * KS + ks) * 64 + lane) * 8, ks = tap * SPT + ci / 16, lane = co % 32 + 32 h
      const int spt = cin / 16, ks = tap * spt + 2 * cib + (g >> 1);
      idx = (((int64_t)cob * 9 * spt + ks) * 64 + kl + 32 * (g & 1)) * 8;
    } else {
      // convh.hip: ((nb * CC + cc) * 9 + tap) * 4096 + (g * 128 + k) * 8
      const int co = 32 * cob + kl;
      idx = (((int64_t)(co >> 7) * cibn + cib) * 9 + tap) * 4096 + (g * 128 + (co & 127)) * 8;
    }
    *reinterpret_cast<u32x4s*>(packed + idx) = v;
  }
}

inline bool reg_shape(int cin, int kout) {
  return (cin == 64 || cin == 128) && (kout == 64 || kout == 128);
}

}  // namespace

extern "C" size_t scl_conv_packed_bytes(int cin, int kout) {
  if (reg_shape(cin, kout)) return scl_round256((size_t)9 * cin * kout * sizeof(unsigned short));
  if (cin < 64 || kout < 128 || cin % 64 || kout % 128 || cin > 1024 || kout > 1024) return 0;
  return scl_round256((size_t)9 * cin * kout * sizeof(unsigned short));
}

extern "C" int scl_conv_pack_batch(const SclPackJob* jobs, int njobs, void* stream) {
  if (!jobs) return SCL_E_NULL;
  if (njobs < 1) return SCL_OK;
  for (int base = 0; base < njobs; base += kMaxJobs) {
    PackJobs pj;
    pj.n = njobs - base < kMaxJobs ? njobs - base : kMaxJobs;
    int blocks = 0;
    for (int i = 0; i < pj.n; ++i) {
      const SclPackJob& j = jobs[base + i];
      if (!j.w || !j.packed) return SCL_E_NULL;
      const bool vlad = (j.flags & SCL_PACK_VLAD_W) != 0;
      if (vlad) {      // assign_w [512][64] float32 contiguous -> scl_netvlad_planes_bytes() bytes
        if (j.flags != SCL_PACK_VLAD_W) return SCL_E_KIND;
        if (j.cin != SCL_VLAD_D || j.kout != SCL_VLAD_K || ((uintptr_t)j.packed % 256) || ((uintptr_t)j.w % 16))
          return SCL_E_SHAPE;
      } else {
        if (scl_conv_packed_bytes(j.cin, j.kout) == 0 || ((uintptr_t)j.packed % 256)) return SCL_E_SHAPE;
        if (j.flags & ~3) return SCL_E_KIND;
      }
      pj.w[i] = j.w;
      pj.packed[i] = (unsigned short*)j.packed;
      pj.sk[i] = j.w_stride_k;
      pj.sc[i] = j.w_stride_c;
      pj.sh[i] = j.w_stride_h;
      pj.sw[i] = j.w_stride_w;
      pj.flags[i] = vlad ? j.flags : (j.flags | (reg_shape(j.cin, j.kout) ? 8 : 0));
      pj.cin[i] = j.cin;
      pj.kout[i] = j.kout;
      pj.first_block[i] = blocks;
      blocks += vlad ? VP_WAVES / 4 : (9 * j.cin * j.kout + 255) / 256;
    }
    pj.first_block[pj.n] = blocks;
    // contiguous OIHW sources (the float32 masters): the tiled kernel; anything else: the gather
    bool tiled = true;
    for (int i = 0; i < pj.n; ++i) {
      const SclPackJob& j = jobs[base + i];
      if (j.flags & SCL_PACK_VLAD_W) continue;              // served by either kernel
      const int src_c = (j.flags & 1) ? j.kout : j.cin;      // the source tensor's dimension 1
      tiled = tiled && j.w_stride_w == 1 && j.w_stride_h == 3 && j.w_stride_c == 9 &&
              j.w_stride_k == (int64_t)9 * src_c && j.cin % 32 == 0 && j.kout % 32 == 0 &&
              ((uintptr_t)j.packed % 16) == 0;
    }
    if (tiled) {
      int tiles = 0;
      for (int i = 0; i < pj.n; ++i) {
        pj.first_block[i] = tiles;
        tiles += (pj.flags[i] & SCL_PACK_VLAD_W) ? VP_WAVES / 4 : (pj.cin[i] / 32) * (pj.kout[i] / 32);
      }
      pj.first_block[pj.n] = tiles;
      SCL_LAUNCH("conv_pack_tiled_kernel", conv_pack_tiled_kernel, dim3((unsigned)tiles), dim3(256), 0,
                 (hipStream_t)stream, pj);
    } else {
      SCL_LAUNCH("conv_pack_batch_kernel", conv_pack_batch_kernel, dim3((unsigned)blocks), dim3(256), 0,
                 (hipStream_t)stream, pj);
    }
  }
  return scl_launch_status();
}
