// et_abi.hip -- version / status helpers of the C ABI (include/eigentraj.h).
#include "et_common.h"

extern "C" int et_abi_version(void) { return ET_ABI_VERSION; }

extern "C" const char *et_compiled_arch(void) { return "gfx950"; }

extern "C" const char *et_status_string(int status) {
    switch (status) {
        case ET_OK: return "ok";
        case ET_ERR_INVALID_ARG: return "invalid argument";
        case ET_ERR_HIP: return "HIP runtime error";
        case ET_ERR_UNSUPPORTED: return "unsupported dimensions";
        case ET_ERR_WORKSPACE: return "workspace missing or too small";
        case ET_ERR_BAD_DATA: return "k-means input contains NaN/Inf";
        case ET_ERR_RCCL: return "RCCL not loadable or a collective failed";
        default: return "unknown status";
    }
}
