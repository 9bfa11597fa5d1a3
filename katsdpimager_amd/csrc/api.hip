// Version / error strings of the libkimg C ABI.
#include "kimg_common.h"

#include <atomic>

extern "C" int kimg_version(void) { return KIMG_VERSION; }

namespace {
std::atomic<int> window_cus{256};       // the process default (kimg_set_window_cus: deprecated)
thread_local int window_cus_call = 0;   // the value of the kimg_grid / kimg_degrid call in progress
}

// What the window kernels of the call in progress on this thread may fill: the call's own value
// (bits 8-16 of its `variant` argument) if it brought one, else the process default.
int kimg_window_cus_now()
{
    return window_cus_call > 0 ? window_cus_call : window_cus.load(std::memory_order_relaxed);
}

kimg_window_cus_scope::kimg_window_cus_scope(int cus) : before(window_cus_call)
{
    window_cus_call = cus > 0 ? cus : before;
}

kimg_window_cus_scope::~kimg_window_cus_scope() { window_cus_call = before; }

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
