// Context, scratch management and error reporting.
#include "fd_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace fd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int scratch(fd_ctx* ctx, Slot slot, size_t bytes, void** out) {
    if (!ctx) {
        set_error("null context");
        return FD_ERR_ARG;
    }
    Buf& b = ctx->slots[slot];
    if (bytes == 0) bytes = 256;
    if (b.cap < bytes) {
        size_t want = bytes + bytes / 2;          // geometric growth bounds the parked memory
        want = (want + 255) & ~size_t(255);
        void* p = nullptr;
        FD_HIP(hipSetDevice(ctx->device));
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("scratch slot %d: hipMalloc(%zu bytes) failed: %s", (int)slot, want, hipGetErrorString(e));
            return FD_ERR_HIP;
        }
        if (b.ptr) ctx->parked.push_back(b.ptr);
        b.ptr = p;
        b.cap = want;
        ctx->bytes += (int64_t)want;
    }
    *out = b.ptr;
    return FD_OK;
}

}  // namespace fd

extern "C" {

int fd_version(void) { return 100; }

const char* fd_last_error(void) { return fd::g_err; }

int fd_ctx_create(int device, fd_ctx** out) {
    if (!out) {
        fd::set_error("fd_ctx_create: out is NULL");
        return FD_ERR_ARG;
    }
    int n = 0;
    FD_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) {
        fd::set_error("fd_ctx_create: device %d out of range (%d visible)", device, n);
        return FD_ERR_ARG;
    }
    FD_HIP(hipSetDevice(device));
    fd_ctx* c = new fd_ctx();
    c->device = device;
    *out = c;
    return FD_OK;
}

void fd_ctx_destroy(fd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& b : ctx->slots)
        if (b.ptr) (void)hipFree(b.ptr);
    for (void* p : ctx->parked) (void)hipFree(p);
    delete ctx;
}

int64_t fd_ctx_scratch_bytes(const fd_ctx* ctx) { return ctx ? ctx->bytes : 0; }

int fd_generate_anchors(double base_size, const double* ratios, int num_ratios, const double* scales, int num_scales,
                        double* anchors) {
    FD_REQUIRE(ratios && scales && anchors && num_ratios > 0 && num_scales > 0, "fd_generate_anchors: bad arguments");
    // Windows share the centre of (0, 0, base-1, base-1); the side lengths are rounded
    // half-to-even once per ratio and then multiplied by each scale.
    const double ctr = 0.5 * (base_size - 1.0);
    const double area = base_size * base_size;
    int row = 0;
    for (int r = 0; r < num_ratios; ++r) {
        const double w0 = std::nearbyint(std::sqrt(area / ratios[r]));
        const double h0 = std::nearbyint(w0 * ratios[r]);
        for (int s = 0; s < num_scales; ++s, ++row) {
            const double w = w0 * scales[s], h = h0 * scales[s];
            anchors[row * 4 + 0] = ctr - 0.5 * (w - 1.0);
            anchors[row * 4 + 1] = ctr - 0.5 * (h - 1.0);
            anchors[row * 4 + 2] = ctr + 0.5 * (w - 1.0);
            anchors[row * 4 + 3] = ctr + 0.5 * (h - 1.0);
        }
    }
    return FD_OK;
}

}  // extern "C"
