// ABI bookkeeping entry points of libscl_hip.so (declared in include/scl_hip.h).
#include "scl_common.h"

extern "C" int scl_abi_version(void) { return SCL_ABI_VERSION; }

extern "C" const char* scl_error_string(int code) {
  switch (code) {
    case SCL_OK: return "ok";
    case SCL_E_SHAPE: return "unsupported or inconsistent shape";
    case SCL_E_KIND: return "unknown loss / mask / dtype selector";
    case SCL_E_NULL: return "required pointer is NULL";
    case SCL_E_WORKSPACE: return "workspace too small or not 256-byte aligned";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

// ---- ablation switch for diagnostic kernel variants (scripts/ablate_rowtile.py) ----------
#ifdef SCL_DIAG
volatile int scl_debug_variant = 0;
extern "C" int scl_debug_set_variant(int v) {
  const int old = scl_debug_variant;
  scl_debug_variant = v;
  return old;
}
extern "C" int scl_build_is_diag(void) { return 1; }
#else
// the product library has no variants: 0 is accepted (and is what it always runs)
extern "C" int scl_debug_set_variant(int v) { return v == 0 ? 0 : SCL_E_KIND; }
extern "C" int scl_build_is_diag(void) { return 0; }
#endif

// ---- CUs left free by the persistent convolution grids (scl_usable_cus in scl_common.h) ----
volatile int scl_reserve_cus = -1;
extern "C" int scl_set_reserve_cus(int n) {
  const int old = scl_reserve_cus;
  scl_reserve_cus = n < 0 ? -1 : n;          // negative: read SCL_RESERVE_CUS again
  return old < 0 ? 0 : old;
}

extern "C" int scl_get_reserve_cus(void) {
  (void)scl_usable_cus(256);                  // resolves the environment variable if nobody has yet
  return scl_reserve_cus < 0 ? 0 : scl_reserve_cus;
}

// ---- per-kernel timing sink (diagnostics; see SCL_LAUNCH in scl_common.h) ---------------
SclProfSink* volatile scl_prof_sink = nullptr;

extern "C" int scl_prof_begin(int capacity) {
  if (scl_prof_sink || capacity < 1) return SCL_E_SHAPE;
  SclProfSink* s = new SclProfSink;
  s->capacity = capacity;
  s->count = 0;
  s->ev = new hipEvent_t[2 * (size_t)capacity];
  s->name = new const char*[capacity];
  for (int i = 0; i < capacity; ++i) s->name[i] = nullptr;
  for (int i = 0; i < 2 * capacity; ++i) {
    hipError_t e = hipEventCreate(&s->ev[i]);
    if (e != hipSuccess) return (int)e;
  }
  scl_prof_sink = s;
  return SCL_OK;
}

extern "C" int scl_prof_count(void) {
  SclProfSink* s = scl_prof_sink;
  if (!s) return 0;
  const int c = __atomic_load_n(&s->count, __ATOMIC_RELAXED);
  return c < s->capacity ? c : s->capacity;
}

// Call after the launching threads are quiescent (e.g. after a device synchronize).  Waits
// for the recorded launches; fills ms[i] and names[i], destroys the sink and returns the
// number of entries.
extern "C" int scl_prof_end(float* ms, const char** names, int capacity) {
  SclProfSink* s = scl_prof_sink;
  if (!s) return 0;
  scl_prof_sink = nullptr;
  int used = __atomic_load_n(&s->count, __ATOMIC_RELAXED);
  if (used > s->capacity) used = s->capacity;
  int n = 0;
  for (int i = 0; i < used; ++i) {
    (void)hipEventSynchronize(s->ev[2 * i + 1]);
    if (i < capacity && ms && names) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, s->ev[2 * i], s->ev[2 * i + 1]);
      ms[n] = t;
      names[n] = s->name[i];
      ++n;
    }
  }
  for (int i = 0; i < 2 * s->capacity; ++i) (void)hipEventDestroy(s->ev[i]);
  delete[] s->ev;
  delete[] s->name;
  delete s;
  return n;
}

// A kernel that does nothing, launched like every other kernel of the library (256 workgroups of
// 256 threads): what the event bracket of SCL_LAUNCH measures beyond a kernel's own duration is
// the bracket of THIS launch minus its device time (rocprofv3: profiles/r04/null_kernel_bracket.txt).
// bench.py subtracts that from every event-bracketed duration it reports.
__global__ void scl_null_kernel(int* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 0;
}
extern "C" int scl_prof_null(void* stream) {
  SCL_LAUNCH("scl_null_kernel", scl_null_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (int*)nullptr);
  return scl_launch_status();
}

// ---- CRC-32C for checkpoint bundles (host only; slice-by-8 tables) ------------------------
namespace {
struct Crc32cTables {
  unsigned t[8][256];
  Crc32cTables() {
    for (unsigned i = 0; i < 256; ++i) {
      unsigned c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      t[0][i] = c;
    }
    for (unsigned i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
  }
};
}  // namespace

extern "C" unsigned scl_crc32c(unsigned crc, const void* data, size_t n) {
  static const Crc32cTables tab;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  unsigned c = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) {
    c = tab.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    --n;
  }
  while (n >= 8) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    w ^= c;
    c = tab.t[7][w & 0xff] ^ tab.t[6][(w >> 8) & 0xff] ^ tab.t[5][(w >> 16) & 0xff] ^
        tab.t[4][(w >> 24) & 0xff] ^ tab.t[3][(w >> 32) & 0xff] ^ tab.t[2][(w >> 40) & 0xff] ^
        tab.t[1][(w >> 48) & 0xff] ^ tab.t[0][(w >> 56) & 0xff];
    p += 8;
    n -= 8;
  }
  while (n--) c = tab.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
  return ~c;
}
