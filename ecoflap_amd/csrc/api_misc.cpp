// api_misc.cpp — version / error strings of the C ABI (host only).
#include <hip/hip_runtime_api.h>

#include "../../include/ecoflap_hip.h"

extern "C" const char* ecoflap_version(void) { return "ecoflap_hip 0.1 (gfx950)"; }

extern "C" const char* ecoflap_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case ECOFLAP_EDTYPE: return "unsupported dtype code";
        case ECOFLAP_ENULL: return "null or aliased pointer argument";
        case ECOFLAP_ESIZE: return "size / rank argument out of range";
        case ECOFLAP_EMODE: return "unknown mode";
        case ECOFLAP_EALIGN: return "pointer not 16-byte aligned";
        case ECOFLAP_EWORKSPACE: return "workspace too small";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}
