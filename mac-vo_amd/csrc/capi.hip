// Library-level entry points of libmacvo_hip.so (see include/macvo_hip.h).
#include "common.h"

extern "C" int mv_abi_version(void) { return 5; }

extern "C" const char* mv_error_string(int code) {
    switch (code) {
        case MV_OK: return "ok";
        case MV_ERR_INVALID_ARG: return "invalid argument";
        case MV_ERR_UNSUPPORTED: return "unsupported configuration";
        case MV_ERR_LAUNCH: return "HIP launch failed";
        case MV_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
