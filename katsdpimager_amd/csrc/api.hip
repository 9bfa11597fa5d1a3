// Version / error strings of the libkimg C ABI.
#include "kimg_common.h"

#include <atomic>

extern "C" int kimg_version(void) { return KIMG_VERSION; }

namespace {
std::atomic<int> window_cus{256};
}

int kimg_window_cus_now() { return window_cus.load(std::memory_order_relaxed); }

extern "C" int kimg_set_window_cus(int cus)
{
    KIMG_CHECK_ARG(cus >= 0 && cus <= 256);
    window_cus.store(cus == 0 ? 256 : cus, std::memory_order_relaxed);
    return 0;
}

extern "C" int kimg_get_window_cus(void) { return kimg_window_cus_now(); }

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
