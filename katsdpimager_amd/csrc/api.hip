// Version / error strings of the libkimg C ABI.
#include "kimg_common.h"

extern "C" int kimg_version(void) { return KIMG_VERSION; }

extern "C" const char *kimg_error_string(int code)
{
    switch (code) {
    case 0: return "success";
    case KIMG_EINVAL: return "invalid argument";
    case KIMG_EUNSUPPORTED: return "unsupported parameter combination";
    case KIMG_EWORKSPACE: return "workspace too small";
    case KIMG_ETIMEOUT: return "persistent kernel timed out waiting for its peer workgroups";
    default:
        if (code < 0 && code > -10000)
            return hipGetErrorString((hipError_t) (-code));
        return "unknown error";
    }
}
