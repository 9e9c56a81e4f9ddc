// Row-panel kernel for the K = 128 1x1 convolutions of the float32 pass (the last convolution of
// every 128-channel bottleneck).
//
// Such a layer moves about as many bytes per multiply as the chip can stream (its residual input and
// its output are each four times as large as its activation operand), and in the general kernel a tile
// spends more time in its prologue and epilogue than in its four or eight K-steps.  Here a workgroup
// owns a panel of output rows for ALL output channels:
//   * the panel's activation rows (BM x K floats = 32 KB) are brought into LDS once and stay there;
//   * the weights stream through a ring of three 16 KB buffers, one chunk of BC output channels each
//     (80 KB of LDS per workgroup: two workgroups per CU, each covering the other's prologue);
//   * the eight waves form two groups of four that alternate: while one group runs the MFMAs of chunk
//     c out of LDS, the other writes out chunk c - 1 (transpose through the ring buffer chunk c - 1 has
//     just released, residual add, stores) and requests the weights of chunk c + 1.  A chunk's residual
//     values are requested when its MFMA phase begins, a whole phase before they are added, and the
//     stores of an epilogue stay in flight into the next phases.  Every SIMD hosts a wave of each group
//     of both resident workgroups, so its matrix pipe always has a wave in its MFMA phase, and the
//     memory traffic of the epilogues is spread evenly under the MFMAs.  (A workgroup that alternates
//     between a matrix phase and a memory phase of about equal length is one customer of two stations;
//     with N of them per CU the kernel runs at about N / (N + 1) of its floor.  The two groups make a
//     workgroup two customers: four per CU here, against three of the general kernel's 128 x 64 tile.)
// Operand images in LDS, fragment addressing, the order in which an output element accumulates over K
// and the arithmetic of the epilogue are those of conv_split_glds_kernel<.., F32>: results are
// bit-identical to that kernel's.
#include "fd_conv_split_dev.h"

#include <cstdlib>
#include <string>

namespace fd {

template <int K>
struct PanelGeom {
    static constexpr int BM = 8192 / K;         // rows of the panel: 32 KB of activations
    static constexpr int BC = 4096 / K;         // output channels per chunk: 16 KB of weights
    static constexpr int NKS = K / 32;          // K-steps
    static constexpr int GWN = BC / 32;         // wave columns of a group (a wave is 32 channels wide)
    static constexpr int GWM = 4 / GWN;         // wave rows
    static constexpr int WM = BM / GWM;         // rows per wave
    static constexpr int AM = WM / 16, AN = 2;  // 16x16 blocks per wave
    static constexpr int RB = 1;                // row blocks per staging pass (4 waves x 16 rows x LDC floats fit the
                                                // 16 KB ring buffer the staging borrows)
    static constexpr int NG = AM / RB;          // staging passes per chunk
    static constexpr int LDC = 36;              // staging row: 32 floats + 4 pad
    static constexpr int RA = BM / 64;          // activation rows per lane (LDS-DMA)
    static constexpr int RW = BC / 32;          // weight rows per lane (LDS-DMA)
};

template <int K>
__global__ __launch_bounds__(512, 4) void conv_f32_panel_kernel(SplitKArgs a, unsigned x_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = PanelGeom<K>;
    constexpr int BM = G::BM, BC = G::BC, NKS = G::NKS, AM = G::AM, AN = G::AN, RB = G::RB, NG = G::NG, LDC = G::LDC;
    constexpr int A_FLOATS = BM * K, W_FLOATS = BC * K;
    static_assert(A_FLOATS == 8192 && W_FLOATS == 4096 && BM % 64 == 0 && BC % 32 == 0, "panel geometry");
    static_assert(4 * 16 * RB * LDC <= W_FLOATS, "the epilogue staging of a group must fit one ring buffer");
    __shared__ __attribute__((aligned(16))) float smem[A_FLOATS + 3 * W_FLOATS];
    float* As = smem;
    float* Ws = smem + A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int wm = wq / G::GWN, wn = wq % G::GWN;
    int tile_m;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nchunks = a.Cout / BC;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    // ---- LDS-DMA addressing (8 rows x 128 B per wave instruction, chunk c of row r at position c ^ ((r >> 1) & 7))
    const int lrow = lane >> 3, lpos = lane & 7;
    const bool linear_rows = a.stride == 1 && a.Ho == a.H && a.Wo == a.W;
    unsigned avo[G::RA];
#pragma unroll
    for (int j = 0; j < G::RA; ++j) {
        const int row = (wave + 8 * j) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        const int m = tile_m * BM + row;
        if (m >= a.M) { avo[j] = OOB; continue; }
        if (linear_rows) {
            avo[j] = (unsigned)(m * a.Cin * 4 + chunk * 16);
        } else {
            const int hw = a.Ho * a.Wo;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int ho = rem / a.Wo;
            const int wo = rem - ho * a.Wo;
            avo[j] = (unsigned)(((img * a.H + ho * a.stride) * a.W + wo * a.stride) * a.Cin * 4 + chunk * 16);
        }
    }
    unsigned bvo[G::RW];
#pragma unroll
    for (int j = 0; j < G::RW; ++j) {
        const int row = (wq + 4 * j) * 8 + lrow;
        const int chunk = lpos ^ ((row >> 1) & 7);
        bvo[j] = (unsigned)(row * K * 4 + chunk * 16);
    }
    auto issue_weights = [&](int c) {
        float* buf = Ws + (c % 3) * W_FLOATS;
#pragma unroll
        for (int j = 0; j < G::RW; ++j)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    wr, (__attribute__((address_space(3))) void*)(buf + (ks * BC + (wq + 4 * j) * 8) * 32), 16, bvo[j],
                    c * (BC * K * 4) + ks * 128, 0, 0);
    };

    // ---- MFMA fragments (as in conv_split_glds_kernel: row R with R & 15 == lane & 15, chunk part * 4 + (lane >> 4))
    const int frow = lane & 15, fh = lane >> 4;
    const int sw = (frow >> 1) & 7;
    int a_off[2], b_off[2];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int pos = ((part * 4 + fh) ^ sw) * 16;
        a_off[part] = (wm * G::WM + frow) * 128 + pos;
        b_off[part] = (wn * 32 + frow) * 128 + pos;
    }

    // ---- epilogue geometry: after the transpose a lane owns 8 consecutive channels of a row; 4 lanes per row
    const int erow = lane >> 2, ecol = (lane & 3) * 8;
    const bool stream_out = (long long)a.M * a.Cout > (80ll << 20);
    const bool full_panel = (tile_m + 1) * BM <= a.M;
    f32x4 res[NG][RB][2];
    auto load_residual = [&](int c, f32x4 (&res)[NG][RB][2]) {
        if (!a.res) return;
        const int gcol = c * BC + wn * 32 + ecol;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int ps = 0; ps < RB; ++ps) {
                const int row = tile_m * BM + wm * G::WM + gi * (16 * RB) + ps * 16 + erow;
                res[gi][ps][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                res[gi][ps][1] = res[gi][ps][0];
                if (row < a.M) {
                    const float* rp = a.res + (size_t)row * a.Cout + gcol;
                    res[gi][ps][0] = *reinterpret_cast<const f32x4*>(rp);
                    res[gi][ps][1] = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
    };

    f32x4 acc[AM][AN];

    auto mma_chunk = [&](int c) {
#pragma unroll
        for (int mi = 0; mi < AM; ++mi)
#pragma unroll
            for (int ni = 0; ni < AN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* abase = reinterpret_cast<const char*>(As);
        const char* wbase = reinterpret_cast<const char*>(Ws + (c % 3) * W_FLOATS);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const char* ak = abase + ks * (BM * 128);
            const char* wk = wbase + ks * (BC * 128);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                f32x4 av[AM], bv[AN];
#pragma unroll
                for (int mi = 0; mi < AM; ++mi) av[mi] = *reinterpret_cast<const f32x4*>(ak + a_off[part] + mi * 2048);
#pragma unroll
                for (int ni = 0; ni < AN; ++ni) bv[ni] = *reinterpret_cast<const f32x4*>(wk + b_off[part] + ni * 2048);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < AM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < AN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][t], bv[ni][t], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    };

    auto epilogue = [&](int c) {
        // staging: this wave's slice of the ring buffer chunk c has released (the group read it one phase ago)
        float* Cs = Ws + (c % 3) * W_FLOATS + wq * (16 * RB * LDC);
        const int gcol = c * BC + wn * 32 + ecol;
        float bias_v[AN];
#pragma unroll
        for (int ni = 0; ni < AN; ++ni) bias_v[ni] = a.bias ? a.bias[c * BC + wn * 32 + ni * 16 + (lane & 15)] : 0.f;
        f32x4 ps0 = {1.f, 1.f, 1.f, 1.f}, ps1 = ps0, pt0 = {0.f, 0.f, 0.f, 0.f}, pt1 = pt0;
        const bool post = a.ysplit && a.post_scale;
        if (post) {
            ps0 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol);
            ps1 = *reinterpret_cast<const f32x4*>(a.post_scale + gcol + 4);
            pt0 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol);
            pt1 = *reinterpret_cast<const f32x4*>(a.post_shift + gcol + 4);
        }
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            if (gi > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int mb = 0; mb < RB; ++mb)
#pragma unroll
                for (int ni = 0; ni < AN; ++ni)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[gi * RB + mb][ni][r] * a.w_inv_scale + bias_v[ni];
                        if (a.relu) v = fmaxf(v, 0.f);
                        Cs[(mb * 16 + (lane >> 4) * 4 + r) * LDC + ni * 16 + (lane & 15)] = v;
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < RB; ++ps) {
                const int lrow2 = ps * 16 + erow;
                const int row = tile_m * BM + wm * G::WM + gi * (16 * RB) + lrow2;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol]);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(&Cs[lrow2 * LDC + ecol + 4]);
                if (a.res) { v0 += res[gi][ps][0]; v1 += res[gi][ps][1]; }
                if (row < a.M) {
                    if (a.y32) {
                        float* yp = a.y32 + (size_t)row * a.Cout + gcol;
                        if (stream_out) {
                            __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(yp));
                            __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(yp + 4));
                        } else {
                            *reinterpret_cast<f32x4*>(yp) = v0;
                            *reinterpret_cast<f32x4*>(yp + 4) = v1;
                        }
                    }
                    if (a.ysplit) {
                        if (post) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                v0[j] = fmaxf(v0[j] * ps0[j] + pt0[j], 0.f);
                                v1[j] = fmaxf(v1[j] * ps1[j] + pt1[j], 0.f);
                            }
                        }
                        float* ap = reinterpret_cast<float*>(a.ysplit) + (size_t)row * a.Cout + gcol;
                        *reinterpret_cast<f32x4*>(ap) = v0;
                        *reinterpret_cast<f32x4*>(ap + 4) = v1;
                    }
                }
            }
        }
    };

    // ---- prologue: the panel's activation rows, the first chunk of each group, their residual values
#pragma unroll
    for (int j = 0; j < G::RA; ++j)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                xr, (__attribute__((address_space(3))) void*)(As + (ks * BM + (wave + 8 * j) * 8) * 32), 16, avo[j], ks * 128,
                0, 0);
    if (grp < nchunks) issue_weights(grp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pipeline_barrier();

    // ---- phases: chunk p is multiplied by group p & 1 in phase p and written out by it in phase p + 1.  The
    // residual values of a chunk are requested when its MFMA phase begins, a whole phase before they are added;
    // the stores of an epilogue stay in flight into the following phases.  What a group waits for at the end of
    // its epilogue phase is only its next chunk of weights, which comes out of L2.
    const int n_out = (a.y32 ? 1 : 0) + (a.ysplit ? 1 : 0);
    for (int p = 0; p <= nchunks; ++p) {
        if ((p & 1) == grp) {
            if (p < nchunks) {
                load_residual(p, res);
                mma_chunk(p);
            }
        } else if (p >= 1) {
            if (p + 1 < nchunks) issue_weights(p + 1);
            epilogue(p - 1);
            // (every lane of a full panel issues every store, so the count holds; the ragged last panel waits for all)
            if (!full_panel) wait_vmcnt<0>();
            else if (n_out == 1) wait_vmcnt<NG * RB * 2>();
            else wait_vmcnt<NG * RB * 4>();
        }
        pipeline_barrier();
    }
#endif
}

template <int K>
static int launch_panel(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    using G = PanelGeom<K>;
    static const std::string name = std::string("conv_f32_panel_k") + std::to_string(K);
    ProfScope scope(stream, name.c_str(), 2.0 * a.M * a.Cout * (double)a.K,
                    (double)a.H * a.W * (a.M / ((double)a.Ho * a.Wo)) * 4.0 * a.Cin + 4.0 * a.Cout * (double)a.K +
                        4.0 * a.M * a.Cout * ((a.res ? 1.0 : 0.0) + (a.y32 ? 1.0 : 0.0) + (a.ysplit ? 1.0 : 0.0)));
    const int tiles_m = ceil_div(a.M, G::BM);
    hipLaunchKernelGGL((conv_f32_panel_kernel<K>), dim3(tiles_m), dim3(512), 0, stream, a, (unsigned)x_bytes, (unsigned)w_bytes);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

bool conv_panel_applies(const SplitKArgs& a) {
    // FD_F32_PANEL=0 keeps these layers on the general kernel; read at every call so that a test can compare the two
    // kernels in one process
    const char* env = getenv("FD_F32_PANEL");
    if ((env && atoi(env) == 0) || a.f32 != 1 || a.pre_scale || a.ksplit != 1 || a.K != a.Cin || a.pad != 0) return false;
    // (K = 64, the projection shortcut of the 64-channel stage, was built as well: 128-row panels, 64-channel
    // chunks; that layer is bound by its output stream alone and ran as fast as on the general kernel)
    if (a.K != 128) return false;
    const int bc = 4096 / a.K, bm = 8192 / a.K;
    if (a.Cout % bc != 0 || a.Cout / bc < 2) return false;
    // (the batched K = 128 GEMMs of the Winograd layers -- four chunks, no residual -- ran 13 % slower here than on
    // the three-workgroup tile: 391 -> 443 us)
    if (a.wbatch_tiles > 0) return false;
    // from most of one round of resident workgroups (2 x 256) on: a single 1080p frame's 507 panels take 46 us here
    // and 53-58 us as 1 016 tiles of the general kernel (FD_F32_PANEL_MIN moves the limit for probes)
    static const int panel_min = getenv("FD_F32_PANEL_MIN") ? atoi(getenv("FD_F32_PANEL_MIN")) : 448;
    return ceil_div(a.M, bm) >= panel_min;
}

int conv_panel_forward(const SplitKArgs& a, size_t x_bytes, size_t w_bytes, hipStream_t stream) {
    return launch_panel<128>(a, x_bytes, w_bytes, stream);
}

}  // namespace fd
