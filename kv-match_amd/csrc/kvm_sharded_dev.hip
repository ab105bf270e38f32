// kvm_sharded_dev.hip -- the device layer of the group entry points (kvm_sharded_dev.h) on HIP and RCCL.
// RCCL is NOT a link dependency of the library: librccl.so.1 (573 MB of kernels the runtime registers at load) is opened the
// first time a group needs a communicator, so that a single-GPU host never pays for it.  The name asked for is the soname:
// found through this library's RUNPATH in ROCm's install -- or, in a process that has already loaded a copy under that
// soname (torch ships one as torch/lib/librccl.so, soname librccl.so.1), that copy: one RCCL per process, bound like this
// library itself to the one HIP runtime of the process (libamdhip64.so.7, whichever copy was loaded first).
#include <dlfcn.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/kvmatch.h"
#include "kvm_sharded_dev.h"

static thread_local std::string t_err;
const char *kvm_sd_last_error() { return t_err.c_str(); }

static int fail_hip(const char *what, hipError_t e) {
    try {
        t_err = std::string(what) + ": " + hipGetErrorString(e);
    } catch (...) {
    }
    return KVM_ERR_DEVICE;
}
#define SDCHK(call)                                  \
    do {                                             \
        hipError_t e_ = (call);                      \
        if (e_ != hipSuccess) return fail_hip(#call, e_); \
    } while (0)

int kvm_sd_stream_synchronize(int device, void *stream) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipStreamSynchronize((hipStream_t)stream));
    return KVM_OK;
}
int kvm_sd_stream_wait_event(int device, void *stream, void *event) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return KVM_OK;
}
int kvm_sd_event_create(int device, void **event) {
    SDCHK(hipSetDevice(device));
    hipEvent_t e = nullptr;
    SDCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return KVM_OK;
}
void kvm_sd_event_destroy(int device, void *event) {
    if (event && hipSetDevice(device) == hipSuccess) (void)hipEventDestroy((hipEvent_t)event);
}
int kvm_sd_event_record(int device, void *event, void *stream) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return KVM_OK;
}
int kvm_sd_event_synchronize(void *event) {
    // Poll first: hipEventSynchronize parks the thread on the event's signal, and being woken costs 50 - 60 us -- more than a
    // whole early-abandoning query (profiles/r06_group_stream_probe.log: 72 us per query with the blocking wait alone).  In a
    // stream that keeps the GPU busy the event has usually completed before anyone asks.
    for (int spin = 0; spin < 4000; spin++) {
        const hipError_t e = hipEventQuery((hipEvent_t)event);
        if (e == hipSuccess) return KVM_OK;
        if (e != hipErrorNotReady) return fail_hip("hipEventQuery", e);
        if (spin >= 64) std::this_thread::yield();
    }
    SDCHK(hipEventSynchronize((hipEvent_t)event));
    return KVM_OK;
}
int kvm_sd_malloc(int device, void **p, size_t bytes) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipMalloc(p, bytes ? bytes : 8));
    return KVM_OK;
}
void kvm_sd_free(int device, void *p) {
    if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p);
}
int kvm_sd_host_alloc(int device, void **p, size_t bytes) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault));
    return KVM_OK;
}
void kvm_sd_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
int kvm_sd_copy_d2h_async(int device, void *dst_host, const void *src_device, size_t bytes, void *stream) {
    SDCHK(hipSetDevice(device));
    SDCHK(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return KVM_OK;
}

__global__ void __launch_bounds__(256) kvm_sd_readback_kernel(double *__restrict__ dst, const double *__restrict__ src, size_t block_doubles) {
    const size_t base = (size_t)blockIdx.x * block_doubles;
    const double c = src[base];
    const size_t cap = block_doubles / 2 - 1;
    const size_t n = c > 0 ? (c < (double)cap ? (size_t)c : cap) : 0;
    for (size_t i = threadIdx.x; i < 2 * (n + 1); i += blockDim.x) dst[base + i] = src[base + i];
}
int kvm_sd_readback_blocks(int device, double *dst_host, const double *src_device, int blocks, size_t block_doubles, void *stream) {
    SDCHK(hipSetDevice(device));
    hipLaunchKernelGGL(kvm_sd_readback_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst_host, src_device, block_doubles);
    SDCHK(hipGetLastError());
    return KVM_OK;
}

// ---- RCCL, opened on demand -------------------------------------------------------------------------------
namespace {
struct rccl_api {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;  // why it could not be opened
};
rccl_api *rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            const char *e = dlerror();
            api.why += std::string(api.why.empty() ? "" : "; ") + (e ? e : nm);
        }
        if (!api.lib) return;
        bool ok = true;
        auto sym = [&](const char *nm) {
            void *p = dlsym(api.lib, nm);
            if (!p) {
                ok = false;
                api.why += std::string("missing symbol ") + nm + " ";
            }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(api.lib);
            api.lib = nullptr;
        }
    });
    return &api;
}
int need_rccl(rccl_api *&a) {
    a = rccl();
    if (a->lib) return KVM_OK;
    try {
        t_err = "RCCL is not available (librccl.so.1): " + a->why;
    } catch (...) {
    }
    return KVM_ERR_DEVICE;
}
int fail_nccl(rccl_api *a, const char *what, ncclResult_t r) {
    try {
        t_err = std::string(what) + ": " + a->GetErrorString(r);
    } catch (...) {
    }
    return KVM_ERR_DEVICE;
}
// RCCL prints a version banner on STDOUT when its first communicator comes up (NCCL_DEBUG=VERSION style); a host that keeps
// stdout for its own protocol (bench.py's one JSON line, a REPL) must not see it: file descriptor 1 points at descriptor 2
// for the duration of the communicator's creation.
struct stdout_to_stderr {
    int saved = -1;
    stdout_to_stderr() {
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0) dup2(2, 1);
    }
    ~stdout_to_stderr() {
        if (saved >= 0) {
            fflush(stdout);
            dup2(saved, 1);
            close(saved);
        }
    }
};
}  // namespace
#define NCHK(a, call)                                      \
    do {                                                   \
        ncclResult_t r_ = (call);                          \
        if (r_ != ncclSuccess) return fail_nccl(a, #call, r_); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == KVM_GROUP_ID_BYTES, "KVM_GROUP_ID_BYTES is the size of an ncclUniqueId");

int kvm_sd_unique_id(void *id, int32_t id_bytes) {
    rccl_api *a;
    if (int rc = need_rccl(a)) return rc;
    if (!id || id_bytes != (int32_t)sizeof(ncclUniqueId)) {
        t_err = "kvm_group_unique_id: id must hold KVM_GROUP_ID_BYTES bytes";
        return KVM_ERR_ILLEGAL_ARGUMENT;
    }
    ncclUniqueId u;
    NCHK(a, a->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return KVM_OK;
}
int kvm_sd_comm_init_all(const int *devices, int n, void **comms) {
    rccl_api *a;
    if (int rc = need_rccl(a)) return rc;
    stdout_to_stderr quiet;
    static_assert(sizeof(ncclComm_t) == sizeof(void *), "ncclComm_t is a pointer");
    NCHK(a, a->CommInitAll((ncclComm_t *)comms, n, devices));
    return KVM_OK;
}
int kvm_sd_comm_init_rank(int device, int rank, int world, const void *id, void **comm) {
    rccl_api *a;
    if (int rc = need_rccl(a)) return rc;
    SDCHK(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    stdout_to_stderr quiet;
    NCHK(a, a->CommInitRank((ncclComm_t *)comm, world, u, rank));
    return KVM_OK;
}
void kvm_sd_comm_destroy(int device, void *comm) {
    rccl_api *a = rccl();
    if (comm && a->lib && hipSetDevice(device) == hipSuccess) (void)a->CommDestroy((ncclComm_t)comm);
}
int kvm_sd_all_gather(int n, const int *devices, void *const *comms, const void *const *send, void *const *recv, size_t doubles,
                      void *const *streams) {
    rccl_api *a;
    if (int rc = need_rccl(a)) return rc;
    if (n > 1) NCHK(a, a->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int i = 0; i < n; i++) {
        if (n > 1 && hipSetDevice(devices[i]) != hipSuccess && first == ncclSuccess) first = ncclUnhandledCudaError;
        ncclResult_t r = a->AllGather(send[i], recv[i], doubles, ncclDouble, (ncclComm_t)comms[i], (hipStream_t)streams[i]);
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    if (n > 1) {
        ncclResult_t r = a->GroupEnd();  // (always closed, also after a failing member: the group must not stay open)
        if (r != ncclSuccess && first == ncclSuccess) first = r;
    }
    if (first != ncclSuccess) return fail_nccl(a, "ncclAllGather", first);
    return KVM_OK;
}
