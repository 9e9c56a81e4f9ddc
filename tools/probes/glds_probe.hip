// Probe: buffer_load ... lds (16 bytes per lane) on gfx950 -- LDS placement and out-of-range behaviour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned voff = ((threadIdx.x ^ 5) * 4) * 4;      // permuted source
    if (lane == 3) voff = 0x7FFFFFF0u;               // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)&lds[wave * 256], 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 v = *reinterpret_cast<f32x4*>(&lds[threadIdx.x * 4]);
    *reinterpret_cast<f32x4*>(y + threadIdx.x * 4) = v;
}
int main() {
    const int n = 1024;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *dx, *dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<1, 256>>>(dx, dy, n);
    hipMemcpy(o.data(), dy, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        float expect0 = ((t & 63) == 3) ? 0.f : (float)((t ^ 5) * 4);
        if (o[t * 4] != expect0) { if (bad < 8) printf("t=%d got %g expect %g\n", t, o[t * 4], expect0); ++bad; }
    }
    printf("glds probe: %d mismatches; lane3 value %g (0 = zero-filled, -7 = untouched)\n", bad, o[3 * 4]);
    return bad != 0;
}
