// Shared host-side plumbing of the detection library: error reporting, the
// per-GPU context and its grow-only scratch slots.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>

#include "../../include/facedet.h"

namespace fd {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define FD_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t fd_e_ = (call);                                                                \
        if (fd_e_ != hipSuccess) {                                                                \
            ::fd::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(fd_e_)); \
            return FD_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define FD_TRY(expr)                  \
    do {                              \
        int fd_rc_ = (expr);          \
        if (fd_rc_ != FD_OK) return fd_rc_; \
    } while (0)

#define FD_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            ::fd::set_error(__VA_ARGS__);  \
            return FD_ERR_ARG;             \
        }                                  \
    } while (0)

// Launch check: catches bad launch configurations at the call site.
#define FD_LAUNCH_CHECK() FD_HIP(hipGetLastError())

// Scratch slots.  Each slot is one device allocation that only ever grows; a
// replaced allocation is parked (kernels already queued on the stream may still be
// using it) and freed at the next safe point, see park().
enum Slot {
    SLOT_SORT_KEYS_A = 0,
    SLOT_SORT_KEYS_B,
    SLOT_SORT_VALS_A,
    SLOT_SORT_VALS_B,
    SLOT_SORT_HIST,
    SLOT_BOXES,        // decoded boxes of every anchor / roi
    SLOT_SCORES,       // score of every anchor / roi
    SLOT_CAND_BOXES,   // boxes of the sorted candidates
    SLOT_CAND_SCORES,
    SLOT_CAND_INDEX,
    SLOT_COUNTERS,     // small int32 block: n_valid / n_cand / n_keep per image
    SLOT_NMS_MASK,
    SLOT_KEEP,
    SLOT_ANCHORS,
    SLOT_IMINFO,
    SLOT_MISC0,
    SLOT_MISC1,
    SLOT_MISC2,
    SLOT_NET0,         // network activations (ping-pong + side buffers)
    SLOT_NET1,
    SLOT_NET2,
    SLOT_NET3,
    SLOT_NET4,
    SLOT_NET5,
    SLOT_NET6,
    SLOT_NET7,
    SLOT_COUNT
};

struct Buf {
    void* ptr = nullptr;
    size_t cap = 0;
};

}  // namespace fd

struct fd_ctx {
    int device = 0;
    fd::Buf slots[fd::SLOT_COUNT];
    std::vector<fd::Buf> parked;       // replaced allocations that queued kernels may still use
    size_t parked_bytes = 0;
    int64_t bytes = 0;
    uint64_t epoch = 0;                // counts replaced allocations: device addresses recorded earlier (a captured
                                       // graph's kernel arguments) are stale once it has moved
};

namespace fd {

// Returns a device buffer of at least `bytes` for `slot` (contents undefined).
int scratch(fd_ctx* ctx, Slot slot, size_t bytes, void** out);

// Hands a replaced allocation to the context.  It is freed at the next safe point: once the parked
// allocations exceed PARK_LIMIT bytes the device is synchronised (nothing queued can still use
// them) and all of them are released, so a stream of growing shapes does not accumulate memory.
constexpr size_t PARK_LIMIT = size_t(256) << 20;
int park(fd_ctx* ctx, void* ptr, size_t bytes);

template <typename T>
inline int scratch_as(fd_ctx* ctx, Slot slot, size_t count, T** out) {
    void* p = nullptr;
    int rc = scratch(ctx, slot, count * sizeof(T), &p);
    *out = static_cast<T*>(p);
    return rc;
}

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// Optional per-kernel timing with HIP events (fd_profile_begin / fd_profile_end).  A scope
// records one event before and one after the launches it brackets, on the launch stream,
// and carries the algorithmic FLOPs / bytes of that work.  Inactive scopes cost one branch.
bool prof_active();
// A token names a record of one profiling session (session number << 32 | record index; negative: none),
// so a scope that outlives its session cannot write into a record of the next one.
long long prof_open(hipStream_t stream, const char* kernel, double flops, double bytes);
void prof_close(hipStream_t stream, long long token);
struct ProfScope {
    hipStream_t stream;
    long long token;
    ProfScope(hipStream_t s, const char* kernel, double flops, double bytes)
        : stream(s), token(prof_active() ? prof_open(s, kernel, flops, bytes) : -1) {}
    ~ProfScope() {
        if (token >= 0) prof_close(stream, token);
    }
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace fd
