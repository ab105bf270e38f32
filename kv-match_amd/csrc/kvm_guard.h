// kvm_guard.h -- the exception barrier of the C ABI.  The entry points build std::string / std::vector objects and start
// threads; a C++ exception (std::bad_alloc on a 1e7-interval list, std::system_error) must not unwind into a JVM or ctypes
// frame: every extern "C" function that can allocate runs its body through kvm_guard and returns a status instead.
#pragma once
#include <exception>
#include <new>
#include <string>

#include "../../include/kvmatch.h"

struct kvm_ctx;
void kvm_set_error_noexcept(kvm_ctx *ctx, const char *msg) noexcept;  // kvm_api.hip (never allocates more than the message)

template <typename R, typename F>
static inline R kvm_guard(kvm_ctx *ctx, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        kvm_set_error_noexcept(ctx, "out of host memory (std::bad_alloc)");
        return (R)KVM_ERR_NO_MEMORY;
    } catch (const std::exception &e) {
        kvm_set_error_noexcept(ctx, e.what());
        return (R)KVM_ERR_DEVICE;
    } catch (...) {
        kvm_set_error_noexcept(ctx, "unknown C++ exception");
        return (R)KVM_ERR_DEVICE;
    }
}
