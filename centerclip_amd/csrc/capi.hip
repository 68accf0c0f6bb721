// Version / status strings of the C ABI (include/centerclip_hip.h).
#include "cc_common.h"

extern "C" {

const char* cc_version(void) { return "centerclip_hip 0.1.0 (gfx950)"; }

const char* cc_status_string(int status) {
    switch (status) {
        case CC_OK: return "ok";
        case CC_ERR_INVALID: return "invalid argument";
        case CC_ERR_UNSUPPORTED: return "unsupported configuration";
        case CC_ERR_WORKSPACE: return "workspace missing or too small";
        case CC_ERR_HIP: return "HIP runtime error";
        default: return "unknown status";
    }
}

}  // extern "C"
