// Context, scratch management and error reporting.
#include "fd_common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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

// ---------------------------------------------------------------------------------
// Event profiler (process-wide; one profiling session at a time).
struct ProfRec {
    hipEvent_t a, b;
    int kernel;
    double flops, bytes;
};
struct Profiler {
    bool active = false;
    std::string only;                    // non-empty: bracket this kernel group only
    std::vector<std::string> names;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
};
static Profiler g_prof;

bool prof_active() { return g_prof.active; }

static hipEvent_t prof_event() {
    if (!g_prof.pool.empty()) {
        hipEvent_t e = g_prof.pool.back();
        g_prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

int prof_open(hipStream_t stream, const char* kernel, double flops, double bytes) {
    if (!g_prof.only.empty() && g_prof.only != kernel) return -1;
    int id = -1;
    for (size_t i = 0; i < g_prof.names.size(); ++i)
        if (g_prof.names[i] == kernel) id = (int)i;
    if (id < 0) {
        g_prof.names.push_back(kernel);
        id = (int)g_prof.names.size() - 1;
    }
    ProfRec r;
    r.a = prof_event();
    r.b = prof_event();
    r.kernel = id;
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.a, stream);
    g_prof.recs.push_back(r);
    return (int)g_prof.recs.size() - 1;
}

void prof_close(hipStream_t stream, int token) { (void)hipEventRecord(g_prof.recs[token].b, stream); }

}  // namespace fd

extern "C" {

int fd_profile_begin(void) {
    fd::g_prof.only.clear();
    fd::g_prof.active = true;
    return FD_OK;
}

int fd_profile_begin_only(const char* kernel) {
    FD_REQUIRE(kernel && kernel[0], "fd_profile_begin_only: empty kernel name");
    fd::g_prof.only = kernel;
    fd::g_prof.active = true;
    return FD_OK;
}

int fd_profile_end(fd_profile_row* rows, int capacity, int* n_rows) {
    FD_REQUIRE(n_rows, "fd_profile_end: n_rows is NULL");
    fd::Profiler& P = fd::g_prof;
    P.active = false;
    FD_HIP(hipDeviceSynchronize());
    const int nk = (int)P.names.size();
    std::vector<fd_profile_row> acc(nk);
    for (int k = 0; k < nk; ++k) {
        memset(&acc[k], 0, sizeof(fd_profile_row));
        snprintf(acc[k].kernel, sizeof(acc[k].kernel), "%s", P.names[k].c_str());
    }
    for (fd::ProfRec& r : P.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            acc[r.kernel].launches += 1;
            acc[r.kernel].ms += ms;
            acc[r.kernel].flops += r.flops;
            acc[r.kernel].bytes += r.bytes;
        }
        P.pool.push_back(r.a);
        P.pool.push_back(r.b);
    }
    P.recs.clear();
    P.names.clear();
    *n_rows = nk;
    for (int k = 0; k < nk && k < capacity; ++k)
        if (rows) rows[k] = acc[k];
    return FD_OK;
}

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
