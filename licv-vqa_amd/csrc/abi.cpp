// Error plumbing + version of the C-ABI (host only).
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <dlfcn.h>
#include "licv_hip.h"

static thread_local char g_err[512] = "";

int licv_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int licv_version(void) { return LICV_ABI_VERSION; }
extern "C" const char* licv_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// The one collective of the path: the sum / mean of the ICV parameters' gradients over the data-parallel
// ranks - what Lightning DDP / DeepSpeed ZeRO-2 do for the reference (ref:config/trainer/ddp.yaml:5, zero2.yaml:5).  A thin wrapper
// over the RCCL the calling process has ALREADY loaded and initialised (the Python host: torch.distributed's; a C++ host: its own):
// the library opens no communicator of its own and does not link RCCL - ncclAllReduce is looked up at the first call.
// ------------------------------------------------------------------------------------------------
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, void*);
extern "C" int licv_allreduce_small(void* nccl_comm, float* data, int64_t n, int average, void* stream) {
    if (!nccl_comm || !data || n < 0) return licv_set_error(LICV_E_BADARG, "allreduce_small: null communicator / buffer or negative count");
    static nccl_allreduce_fn fn = nullptr;
    if (!fn) {
        fn = (nccl_allreduce_fn)dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!fn) {
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (h) fn = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
        }
        if (!fn) return licv_set_error(LICV_E_UNSUPPORTED, "allreduce_small: no RCCL (ncclAllReduce) in this process");
    }
    if (n == 0) return LICV_OK;
    const int rc = fn(data, data, (size_t)n, /* ncclFloat32 */ 7, average ? /* ncclAvg */ 4 : /* ncclSum */ 0, nccl_comm, stream);
    if (rc != 0) return licv_set_error(LICV_E_HIP, "allreduce_small: ncclAllReduce returned %d", rc);
    return LICV_OK;
}
