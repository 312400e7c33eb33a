// libp2l_hip: version / error strings.
#include "p2l_common.h"

thread_local int g_p2l_last_hip_error = 0;

extern "C" int p2l_version(void) { return 101; }

extern "C" int p2l_last_hip_error(void) { return g_p2l_last_hip_error; }

extern "C" const char* p2l_strerror(int rc) {
  switch (rc) {
    case P2L_OK: return "ok";
    case P2L_EINVAL: return "invalid argument (shape / alignment / null pointer)";
    case P2L_ELAUNCH: return "HIP kernel launch failed";
    case P2L_EWS: return "workspace too small";
    case P2L_EUNSUP: return "unsupported combination";
    default: return "unknown p2l error";
  }
}
