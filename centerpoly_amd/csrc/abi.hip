// ABI bookkeeping entry points of libcenterpoly_hip.so.
#include "cp_common.h"

extern "C" int cp_abi_version(void) { return CP_ABI_VERSION; }

extern "C" const char* cp_build_arch(void) { return "gfx950"; }

extern "C" const char* cp_strerror(int code) {
  switch (code) {
    case CP_OK: return "ok";
    case CP_EINVAL: return "invalid argument";
    case CP_EUNSUPPORTED: return "unsupported shape or option";
    case CP_EWORKSPACE: return "workspace too small";
    case CP_EHIP: return "HIP runtime / launch failure";
    default: return "unknown error";
  }
}
