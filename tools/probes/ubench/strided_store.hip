// Does the order in which a tile's output bytes leave matter to HBM?  Each workgroup owns 128 rows of a [M][256]
// float32 tensor (1 KB per row) and adds 1 to it (read + write, like a residual epilogue), either row by row in whole
// 1 KB rows (mode 0) or as four column chunks of 256 B per row with a pause between the chunks (mode 1: the order of
// the fused tails' 1x1 chunk loop), or chunked without the pause (mode 2).
//   hipcc --offload-arch=gfx950 -O3 -o strided_store strided_store.hip && ./strided_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void touch(float* __restrict__ y, const float* __restrict__ r, long long rows, int mode, int pause) {
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 128;
    if (mode == 0) {
        // a wave instruction covers 4 whole rows (64 lanes x 16 B = 1 KB each)
        for (int i = 0; i < 32; ++i) {
            const long long row = row0 + i * 4 + (tid >> 6);
            if (row >= rows) continue;
            const long long o = row * 256 + (tid & 63) * 4;
            f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r + o));
            v += 1.f;
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y + o));
        }
    } else {
        for (int ch = 0; ch < 4; ++ch) {
            // a wave instruction covers 16 rows x 64 B... as the epilogue does: 8 lanes x 32 B = 256 B per row, 8 rows
            for (int i = 0; i < 4; ++i) {
                const long long row = row0 + i * 32 + (tid >> 3);
                if (row >= rows) continue;
                const long long o = row * 256 + ch * 64 + (tid & 7) * 8;
                f32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r + o));
                f32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r + o + 4));
                v0 += 1.f; v1 += 1.f;
                __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(y + o));
                __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(y + o + 4));
            }
            if (mode == 1) for (int p = 0; p < pause; ++p) __builtin_amdgcn_s_sleep(127);
        }
    }
}

int main() {
    const long long rows = 16ll * 270 * 480;      // the 64-channel stage's map, 16 frames
    float *y, *r;
    if (hipMalloc(&y, rows * 1024) != hipSuccess || hipMalloc(&r, rows * 1024) != hipSuccess) return 1;
    (void)hipMemset(r, 0, rows * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (int)((rows + 127) / 128);
    for (int mode = 0; mode < 3; ++mode)
        for (int pause = (mode == 1 ? 1 : 0); pause <= (mode == 1 ? 2 : 0); ++pause) {
            for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, 0, y, r, rows, mode, pause);
            (void)hipEventRecord(e0, 0);
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, 0, y, r, rows, mode, pause);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d pause %d: %.1f us per pass, %.2f TB/s (read + write)\n", mode, pause, ms / 5 * 1e3, 2.0 * rows * 1024 / (ms / 5 * 1e-3) / 1e12);
        }
    return 0;
}
