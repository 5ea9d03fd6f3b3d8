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
