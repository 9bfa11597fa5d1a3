// Version / error strings of the libkimg C ABI.
#include "kimg_common.h"

#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <vector>

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

namespace {
std::mutex dynamic_lds_mutex;
std::map<std::pair<const void *, int>, size_t> dynamic_lds_set;
}

int kimg_dynamic_lds(const void *fn, size_t bytes)
{
    int device = 0;
    KIMG_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(dynamic_lds_mutex);
    size_t &have = dynamic_lds_set[std::make_pair(fn, device)];
    if (bytes > have) {
        KIMG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
        have = bytes;
    }
    return 0;
}

// The runtime loads a code object when the first kernel of it is launched.  Channels imaged on several
// host threads reach the first launches of one code object together, and a launch that meets the
// load of its own code object half way has been seen to do nothing (the first kernel of the weights
// stage of ONE of four channels started together, about one run in three: an all-zero weights grid,
// a PSF of zeros, NaN from there on).  kimg_preload loads every code object of the library on the
// current device, in the calling thread, before anybody launches anything.
namespace {
std::vector<const void *> &preload_list()
{
    static std::vector<const void *> list;
    return list;
}
std::mutex preload_mutex;
std::set<int> preloaded_devices;
}

hipStream_t kimg_capture_stream()
{
    // (per thread and device; lives as long as the process)
    thread_local std::map<int, hipStream_t> streams;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
        return nullptr;
    hipStream_t &s = streams[device];
    if (s == nullptr && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
        s = nullptr;
    return s;
}

int kimg_register_kernel(const void *fn)
{
    preload_list().push_back(fn);
    return 0;
}

namespace {
std::vector<void (*)()> &touch_list()
{
    static std::vector<void (*)()> list;
    return list;
}
}

int kimg_register_touch(void (*launch)())
{
    touch_list().push_back(launch);
    return 0;
}

extern "C" int kimg_preload(void)
{
    int device = 0;
    KIMG_HIP(hipGetDevice(&device));
    std::lock_guard<std::mutex> lock(preload_mutex);
    if (preloaded_devices.count(device))
        return 0;
    for (const void *fn : preload_list()) {
        hipFuncAttributes attr;
        KIMG_HIP(hipFuncGetAttributes(&attr, fn));
    }
    for (void (*launch)() : touch_list()) {
        launch();
        KIMG_HIP(hipGetLastError());
    }
    KIMG_HIP(hipDeviceSynchronize());
    preloaded_devices.insert(device);
    return 0;
}

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
