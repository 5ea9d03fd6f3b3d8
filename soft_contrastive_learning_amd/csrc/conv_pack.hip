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

namespace {

constexpr int kMaxJobs = 32;

struct PackJobs {
  const void* w[kMaxJobs];
  unsigned short* packed[kMaxJobs];
  int64_t sk[kMaxJobs], sc[kMaxJobs], sh[kMaxJobs], sw[kMaxJobs];
  int flags[kMaxJobs];       // SCL_CONV_TRANSPOSED | SCL_W_F32 | 8: register-weights layout
  int cin[kMaxJobs], kout[kMaxJobs];
  int first_block[kMaxJobs + 1];   // prefix sums of 256-element blocks
  int n;
};

__global__ __launch_bounds__(256) void conv_pack_batch_kernel(const PackJobs jobs) {
  int job = 0;
  while (job + 1 < jobs.n && (int)blockIdx.x >= jobs.first_block[job + 1]) ++job;
  const int64_t idx = (int64_t)(blockIdx.x - jobs.first_block[job]) * 256 + threadIdx.x;
  const int cin = jobs.cin[job], kout = jobs.kout[job], flags = jobs.flags[job];
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
      if (scl_conv_packed_bytes(j.cin, j.kout) == 0 || ((uintptr_t)j.packed % 256)) return SCL_E_SHAPE;
      if (j.flags & ~3) return SCL_E_KIND;
      pj.w[i] = j.w;
      pj.packed[i] = (unsigned short*)j.packed;
      pj.sk[i] = j.w_stride_k;
      pj.sc[i] = j.w_stride_c;
      pj.sh[i] = j.w_stride_h;
      pj.sw[i] = j.w_stride_w;
      pj.flags[i] = j.flags | (reg_shape(j.cin, j.kout) ? 8 : 0);
      pj.cin[i] = j.cin;
      pj.kout[i] = j.kout;
      pj.first_block[i] = blocks;
      blocks += (9 * j.cin * j.kout + 255) / 256;
    }
    pj.first_block[pj.n] = blocks;
    SCL_LAUNCH("conv_pack_batch_kernel", conv_pack_batch_kernel, dim3((unsigned)blocks), dim3(256), 0,
               (hipStream_t)stream, pj);
  }
  return scl_launch_status();
}
