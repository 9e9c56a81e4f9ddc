// Box-stage kernels for gfx950: delta decode / clip / min-size filter, stable segmented
// radix sort, bitmask IoU NMS, RPN proposal assembly, ROI max pooling and the detector
// tail.  All of this is integer / latency / HBM-bound work: 64-wide wave ballots for
// ranking and counting, LDS for the per-block digit tables and the suppression bitmap,
// no matrix cores.
//
// Floating-point contraction is off for the whole file: the IoU and decode arithmetic
// has to round exactly where NumPy rounds (one rounding per +, -, *, /), so no a*b+c may
// be fused.  (The build also passes -ffp-contract=off.)
#include "fd_boxes.h"
#include "fd_split.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>

#pragma clang fp contract(off)

namespace fd {

constexpr long long ceil_div_c(long long a, long long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------
// Sort keys: ascending key order == descending score order.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t desc_key(float s) {
    if (s == 0.0f) s = 0.0f;   // -0.0 and +0.0 compare equal and must tie
    uint32_t u = __float_as_uint(s);
    uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~asc;
}
__device__ __forceinline__ uint64_t desc_key(double s) {
    if (s == 0.0) s = 0.0;
    uint64_t u = (uint64_t)__double_as_longlong(s);
    uint64_t asc = (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
    return ~asc;
}
constexpr uint32_t KEY_INVALID32 = 0xFFFFFFFFu;   // sorts behind every real score

// ---------------------------------------------------------------------------------
// Stable segmented LSD radix sort, 8-bit digits.
// One tile = 256 threads x 8 rounds; in round r thread t owns element tile*2048 + r*256 + t,
// so (round, wave, lane) order is element order and ranks can be taken with wave ballots.
// ---------------------------------------------------------------------------------
constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = 8;
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;

template <typename K>
__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const K* __restrict__ keys, int n, int shift,
                                                                uint32_t* __restrict__ hist, int nblk) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x, blk = blockIdx.x, seg = blockIdx.y;
    h[tid] = 0;
    __syncthreads();
    const K* k = keys + (size_t)seg * n;
    const int base = blk * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int i = base + r * SORT_THREADS + tid;
        if (i < n) atomicAdd(&h[(uint32_t)(k[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[((size_t)seg * 256 + tid) * nblk + blk] = h[tid];
}

template <typename K>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const K* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   K* __restrict__ keys_out,
                                                                   uint32_t* __restrict__ vals_out, int n, int shift,
                                                                   const uint32_t* __restrict__ hist, int nblk) {
    __shared__ uint32_t digit_base[256];
    __shared__ uint32_t wave_cnt[4][256];
    __shared__ uint32_t scan[256];
    const int tid = threadIdx.x, blk = blockIdx.x, seg = blockIdx.y;
    const int lane = tid & 63, wave = tid >> 6;

    // Where does this tile's first element of digit `tid` go?  Elements of smaller digits
    // (all tiles) come first, then the same digit in earlier tiles.
    uint32_t total = 0, before = 0;
    const uint32_t* row = hist + ((size_t)seg * 256 + tid) * nblk;
    for (int b = 0; b < nblk; ++b) {
        const uint32_t c = row[b];
        if (b < blk) before += c;
        total += c;
    }
    scan[tid] = total;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive Hillis-Steele scan over the 256 digits
        uint32_t v = (tid >= off) ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    digit_base[tid] = scan[tid] - total + before;
    wave_cnt[0][tid] = wave_cnt[1][tid] = wave_cnt[2][tid] = wave_cnt[3][tid] = 0;
    __syncthreads();

    const size_t seg_off = (size_t)seg * n;
    const int base = blk * SORT_TILE;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const int i = base + r * SORT_THREADS + tid;
        const bool valid = i < n;
        K key = 0;
        uint32_t val = 0, d = 0;
        if (valid) {
            key = keys_in[seg_off + i];
            val = vals_in[seg_off + i];
            d = (uint32_t)(key >> shift) & 255u;
        }
        // lanes of this wave holding the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t m = __ballot(valid && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt_mask);
        if (valid && rank == 0) wave_cnt[wave][d] = __popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = digit_base[d] + rank;
            for (int w = 0; w < wave; ++w) pos += wave_cnt[w][d];
            keys_out[seg_off + pos] = key;
            vals_out[seg_off + pos] = val;
        }
        __syncthreads();
        digit_base[tid] += wave_cnt[0][tid] + wave_cnt[1][tid] + wave_cnt[2][tid] + wave_cnt[3][tid];
        wave_cnt[0][tid] = wave_cnt[1][tid] = wave_cnt[2][tid] = wave_cnt[3][tid] = 0;
        __syncthreads();
    }
}

// Short segments (the detector tail sorts 10 rois per image): one workgroup per segment ranks
// every element against all others in LDS -- rank = number of smaller keys + number of equal keys
// at lower positions, i.e. the same stable order as the radix passes -- in a single launch
// instead of eight.
constexpr int SORT_SMALL_MAX = 2048;
template <typename K>
__global__ __launch_bounds__(256) void sort_small_kernel(K* __restrict__ keys, uint32_t* __restrict__ vals, int n) {
    __shared__ K sk[SORT_SMALL_MAX];
    __shared__ uint32_t sv[SORT_SMALL_MAX];
    const int seg = blockIdx.x, tid = threadIdx.x;
    K* k = keys + (size_t)seg * n;
    uint32_t* v = vals + (size_t)seg * n;
    for (int i = tid; i < n; i += 256) { sk[i] = k[i]; sv[i] = v[i]; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const K ki = sk[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const K kj = sk[j];
            rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
        }
        k[rank] = ki;
        v[rank] = sv[i];
    }
}

// `hist`: nseg * 256 * ceil(n / SORT_TILE) counters (unused for n <= SORT_SMALL_MAX)
template <typename K>
static int sort_pairs_ws(K* keys_a, uint32_t* vals_a, K* keys_b, uint32_t* vals_b, int n, int nseg, uint32_t* hist,
                         hipStream_t stream) {
    if (n <= 0 || nseg <= 0) return FD_OK;
    if (n <= SORT_SMALL_MAX) {
        ProfScope scope(stream, "sort_small", 0.0, 2.0 * (sizeof(K) + 4.0) * n * nseg);
        hipLaunchKernelGGL(sort_small_kernel<K>, dim3(nseg), dim3(256), 0, stream, keys_a, vals_a, n);
        FD_LAUNCH_CHECK();
        return FD_OK;
    }
    const int nblk = ceil_div(n, SORT_TILE);
    dim3 grid(nblk, nseg);
    ProfScope scope(stream, "radix_sort", 0.0, (double)sizeof(K) * 8 / 8 * 2.0 * (sizeof(K) + 4.0) * n * nseg);
    K* src_k = keys_a;
    uint32_t* src_v = vals_a;
    K* dst_k = keys_b;
    uint32_t* dst_v = vals_b;
    for (int shift = 0; shift < (int)sizeof(K) * 8; shift += 8) {
        hipLaunchKernelGGL(sort_hist_kernel<K>, grid, dim3(SORT_THREADS), 0, stream, src_k, n, shift, hist, nblk);
        FD_LAUNCH_CHECK();
        hipLaunchKernelGGL(sort_scatter_kernel<K>, grid, dim3(SORT_THREADS), 0, stream, src_k, src_v, dst_k, dst_v, n,
                           shift, hist, nblk);
        FD_LAUNCH_CHECK();
        K* tk = src_k; src_k = dst_k; dst_k = tk;
        uint32_t* tv = src_v; src_v = dst_v; dst_v = tv;
    }
    return FD_OK;   // an even number of passes: the result is back in (keys_a, vals_a)
}

template <typename K>
static int sort_pairs(fd_ctx* ctx, K* keys_a, uint32_t* vals_a, K* keys_b, uint32_t* vals_b, int n, int nseg,
                      hipStream_t stream) {
    uint32_t* hist = nullptr;
    if (n > SORT_SMALL_MAX && nseg > 0)
        FD_TRY(scratch_as(ctx, SLOT_SORT_HIST, (size_t)nseg * 256 * ceil_div(n, SORT_TILE), &hist));
    return sort_pairs_ws<K>(keys_a, vals_a, keys_b, vals_b, n, nseg, hist, stream);
}

int sort_pairs_u32(fd_ctx* ctx, uint32_t* ka, uint32_t* va, uint32_t* kb, uint32_t* vb, int n, int nseg,
                   hipStream_t stream) {
    return sort_pairs<uint32_t>(ctx, ka, va, kb, vb, n, nseg, stream);
}
int sort_pairs_u64(fd_ctx* ctx, uint64_t* ka, uint32_t* va, uint64_t* kb, uint32_t* vb, int n, int nseg,
                   hipStream_t stream) {
    return sort_pairs<uint64_t>(ctx, ka, va, kb, vb, n, nseg, stream);
}

// ---------------------------------------------------------------------------------
// Bitmask NMS, in row panels.
// mask[seg][i - panel start][w] bit b == 1  <=>  candidate j = 64 w + b (j > i) is suppressed
// by i, i.e. NOT (iou(i, j) <= thresh), with the reference's arithmetic:
//   area = (x2 - x1 + 1) * (y2 - y1 + 1)
//   iw = max(0, min(x2i, x2j) - max(x1i, x1j) + 1), ih likewise, inter = iw * ih
//   iou = inter / (area_i + area_j - inter)
// Only the upper triangle (w >= i / 64) is written and read.  Candidates are processed in
// panels of NMS_PANEL_CHUNKS * 64 rows: mask kernel, then the serial reduce for those rows,
// which carries the `removed` bitmap and the keep count to the next panel through global
// memory.  Once max_keep boxes are kept the remaining panels' kernels exit at once, so a
// proposal stage that wants 6 000 of 100 000 candidates computes only the rows it needs, and
// the mask buffer is one panel (<= 4096 x cap/8 bytes) instead of cap^2/8 bytes.
// ---------------------------------------------------------------------------------
constexpr int NMS_PANEL_CHUNKS = 64;

__device__ __forceinline__ uint64_t wave_uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <typename T>
__global__ __launch_bounds__(64) void nms_mask_kernel(const T* __restrict__ boxes, const int32_t* __restrict__ counters,
                                                      int cap, int words, int chunk0, int panel_rows, int max_keep,
                                                      T thresh, const uint64_t* __restrict__ removed_g,
                                                      uint64_t* __restrict__ mask) {
    const int col = chunk0 + blockIdx.x, rowc = chunk0 + blockIdx.y, seg = blockIdx.z;
    if (col < rowc) return;
    if (max_keep > 0 && counters[seg * CNT_STRIDE + CNT_KEEP] >= max_keep) return;
    const int n = counters[seg * CNT_STRIDE + CNT_CAND];
    if (rowc * 64 >= n || col * 64 >= n) return;
    // Candidates already suppressed by boxes kept in earlier panels need neither a mask row (only rows of
    // kept boxes are ever read) nor mask bits as columns (they are removed whatever the bit says): a block
    // whose 64 rows or 64 columns are all gone exits, the others skip dead rows and visit live columns only.
    // Words this leaves unwritten or incomplete are read back only for dead rows, or OR-ed into bitmap words
    // whose live bits they cannot change.
    uint64_t dead_rows = 0ull, dead_cols = 0ull;
    if (removed_g) {
        dead_rows = removed_g[(size_t)seg * words + rowc];
        dead_cols = removed_g[(size_t)seg * words + col];
        if (dead_rows == ~0ull || dead_cols == ~0ull) return;
    }
    __shared__ T cb[64][5];
    const int t = threadIdx.x;
    const T* segb = boxes + (size_t)seg * cap * 4;
    {
        const int j = col * 64 + t;
        T x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (j < n) {
            x1 = segb[(size_t)j * 4 + 0];
            y1 = segb[(size_t)j * 4 + 1];
            x2 = segb[(size_t)j * 4 + 2];
            y2 = segb[(size_t)j * 4 + 3];
        }
        cb[t][0] = x1; cb[t][1] = y1; cb[t][2] = x2; cb[t][3] = y2;
        cb[t][4] = (x2 - x1 + T(1)) * (y2 - y1 + T(1));
    }
    __syncthreads();
    const int i = rowc * 64 + t;
    if (i >= n || ((dead_rows >> t) & 1ull)) return;
    const T x1 = segb[(size_t)i * 4 + 0], y1 = segb[(size_t)i * 4 + 1];
    const T x2 = segb[(size_t)i * 4 + 2], y2 = segb[(size_t)i * 4 + 3];
    const T area = (x2 - x1 + T(1)) * (y2 - y1 + T(1));
    const int jn = min(64, n - col * 64);
    const int j0 = (col == rowc) ? t + 1 : 0;
    uint64_t bits = 0;
    const bool fast_ok = thresh > T(0);
    const T margin_lo = sizeof(T) == 4 ? T(1.0 - 1.0 / (1 << 20)) : T(1.0 - 1.0 / (1ll << 48));
    const T margin_hi = sizeof(T) == 4 ? T(1.0 + 1.0 / (1 << 20)) : T(1.0 + 1.0 / (1ll << 48));
    // live columns of this block (uniform over the wave: a scalar loop over the set bits)
    uint64_t live = ~dead_cols;
    if (jn < 64) live &= ~(~0ull << jn);
    live = wave_uniform_u64(live);
    while (live) {
        const int j = __ffsll((long long)live) - 1;
        live &= live - 1;
        if (j < j0) continue;
        const T xx1 = (x1 > cb[j][0]) ? x1 : cb[j][0];
        const T yy1 = (y1 > cb[j][1]) ? y1 : cb[j][1];
        const T xx2 = (x2 < cb[j][2]) ? x2 : cb[j][2];
        const T yy2 = (y2 < cb[j][3]) ? y2 : cb[j][3];
        const T dw = xx2 - xx1 + T(1), dh = yy2 - yy1 + T(1);
        const T iw = (dw > T(0)) ? dw : T(0);
        const T ih = (dh > T(0)) ? dh : T(0);
        const T inter = iw * ih;
        const T den = area + cb[j][4] - inter;
        // The reference's test is  inter / den <= thresh  with a rounded quotient.  Far from the threshold
        // the division is not needed: the rounded product p = thresh * den is within one rounding of the
        // true one, so inter below p by more than a few units in the last place means a true quotient below
        // thresh (and a rounded one <= thresh), inter above it by as much a rounded quotient > thresh.
        // Only the pairs in between (and degenerate denominators) take the exact division.
        const T p = thresh * den;
        bool sup;
        if (fast_ok && den > T(0) && inter < p * margin_lo) sup = false;
        else if (fast_ok && den > T(0) && inter > p * margin_hi) sup = true;
        else sup = !(inter / den <= thresh);
        if (sup) bits |= 1ull << j;
    }
    mask[((size_t)seg * panel_rows + (i - chunk0 * 64)) * words + col] = bits;
}

// Serial part, one workgroup per image and panel.  Chunk c (64 candidates) of the panel can be
// resolved once it is known which earlier boxes are kept:
//   rem_c = removed_g[c]                                   (kept boxes of earlier panels)
//         | OR over kept rows r of this panel before chunk c of mask[r][c]
//         | the chunk's own diagonal word, resolved bit by bit by one wave (v_readlane).
// The middle term is pulled: every thread owns up to four panel rows and loads their word of
// column c+1 while wave 0 is still resolving chunk c -- the loads do not depend on the result,
// only the masking with the kept bits does -- so no global load sits on the serial chain.
// Kept bits per chunk go to keptbits[] for the spread kernel; the running count stays in
// counters[CNT_KEEP].
constexpr int REDUCE_THREADS = 1024;
constexpr int NMS_PANEL_ROWS = NMS_PANEL_CHUNKS * 64;
constexpr int ROWS_PER_THREAD = NMS_PANEL_ROWS / REDUCE_THREADS;
__global__ __launch_bounds__(REDUCE_THREADS) void nms_reduce_kernel(const uint64_t* __restrict__ mask,
                                                                   int32_t* __restrict__ counters,
                                                                   const uint64_t* __restrict__ removed_g,
                                                                   uint64_t* __restrict__ keptbits, int cap, int words,
                                                                   int chunk0, int panel_rows, int max_keep,
                                                                   int32_t* __restrict__ keep) {
    __shared__ uint64_t s_kept[NMS_PANEL_CHUNKS];
    __shared__ unsigned long long s_rem;
    __shared__ int s_count;
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    uint64_t* segkb = keptbits + (size_t)seg * words;
    const int count0 = counters[seg * CNT_STRIDE + CNT_KEEP];
    const int n = counters[seg * CNT_STRIDE + CNT_CAND];
    const int W = (n + 63) >> 6;
    const int cend = min(chunk0 + panel_rows / 64, W);
    // chunks this launch does not reach keep no box
    for (int c = chunk0 + tid; c < chunk0 + panel_rows / 64 && c < words; c += REDUCE_THREADS) segkb[c] = 0ull;
    if ((max_keep > 0 && count0 >= max_keep) || chunk0 >= W) return;
    const uint64_t* segm = mask + (size_t)seg * panel_rows * words;
    const uint64_t* segr = removed_g + (size_t)seg * words;
    int32_t* segk = keep + (size_t)seg * cap;
    if (tid < NMS_PANEL_CHUNKS) s_kept[tid] = 0ull;
    if (tid == 0) { s_count = count0; s_rem = 0ull; }
    uint64_t pv[ROWS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; ++j) pv[j] = 0ull;
    __syncthreads();
    for (int c = chunk0; c < cend; ++c) {
        const int lrow0 = (c - chunk0) * 64;        // first panel row of the chunk
        // pull: pv[] holds column c of this thread's rows (loaded during the previous chunk)
        {
            uint64_t acc = 0ull;
#pragma unroll
            for (int j = 0; j < ROWS_PER_THREAD; ++j) {
                const int r = tid + j * REDUCE_THREADS;
                if (r < lrow0 && ((s_kept[r >> 6] >> (r & 63)) & 1ull)) acc |= pv[j];
            }
            if (acc) atomicOr(&s_rem, (unsigned long long)acc);
        }
        __syncthreads();
        // prefetch column c + 1 for every row up to and including this chunk's
        if (c + 1 < cend) {
#pragma unroll
            for (int j = 0; j < ROWS_PER_THREAD; ++j) {
                const int r = tid + j * REDUCE_THREADS;
                pv[j] = (r < lrow0 + 64) ? segm[(size_t)r * words + c + 1] : 0ull;
            }
        }
        if (tid < 64) {
            const int i = c * 64 + lane;
            const uint64_t diag = (i < n) ? segm[(size_t)(lrow0 + lane) * words + c] : 0ull;
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            uint64_t rem = wave_uniform_u64((uint64_t)s_rem | (chunk0 == 0 ? 0ull : segr[c]));
            const int left = n - c * 64;
            if (left < 64) rem |= ~0ull << left;
            uint64_t kept = 0;
#pragma unroll
            for (int b = 0; b < 64; ++b) {
                if (!((rem >> b) & 1ull)) {
                    kept |= 1ull << b;
                    const uint32_t lo = __builtin_amdgcn_readlane(dlo, b);
                    const uint32_t hi = __builtin_amdgcn_readlane(dhi, b);
                    rem |= ((uint64_t)hi << 32) | lo;
                }
            }
            const int count = s_count;
            if ((kept >> lane) & 1ull) {
                const uint64_t below = (lane == 0) ? 0ull : (kept & (~0ull >> (64 - lane)));
                const int pos = count + __popcll(below);
                if (max_keep <= 0 || pos < max_keep) segk[pos] = i;
            }
            if (lane == 0) {
                s_kept[c - chunk0] = kept;
                segkb[c] = kept;
                s_count = count + __popcll(kept);
                s_rem = 0ull;
            }
        }
        __syncthreads();
        if (max_keep > 0 && s_count >= max_keep) break;
    }
    if (tid == 0) {
        int count = s_count;
        if (max_keep > 0 && count > max_keep) count = max_keep;
        counters[seg * CNT_STRIDE + CNT_KEEP] = count;
    }
}

// After a panel: OR the mask rows of its kept boxes into removed_g for all later columns.
// grid (column blocks of 256 words, groups of 4 chunks of the panel, images).
constexpr int SPREAD_CHUNKS = 4;
__global__ __launch_bounds__(256) void nms_spread_kernel(const uint64_t* __restrict__ mask,
                                                        const int32_t* __restrict__ counters,
                                                        const uint64_t* __restrict__ keptbits,
                                                        uint64_t* __restrict__ removed_g, int words, int chunk0,
                                                        int panel_rows, int max_keep) {
    const int seg = blockIdx.z;
    if (max_keep > 0 && counters[seg * CNT_STRIDE + CNT_KEEP] >= max_keep) return;
    const int n = counters[seg * CNT_STRIDE + CNT_CAND];
    const int W = (n + 63) >> 6;
    const int cend = chunk0 + panel_rows / 64;
    const int w = cend + blockIdx.x * 256 + threadIdx.x;
    const int cfirst = chunk0 + blockIdx.y * SPREAD_CHUNKS;
    if (cfirst >= W) return;
    const uint64_t* segm = mask + (size_t)seg * panel_rows * words;
    const uint64_t* segkb = keptbits + (size_t)seg * words;
    uint64_t acc = 0ull;
    for (int c = cfirst; c < cfirst + SPREAD_CHUNKS && c < cend; ++c) {
        uint64_t k = wave_uniform_u64(segkb[c]);
        const uint64_t* rows = segm + (size_t)(c - chunk0) * 64 * words;
        while (k) {
            const int b = __ffsll((long long)k) - 1;
            k &= k - 1;
            if (w < W) acc |= rows[(size_t)b * words + w];
        }
    }
    if (w < W && acc) atomicOr(reinterpret_cast<unsigned long long*>(removed_g + (size_t)seg * words + w),
                               (unsigned long long)acc);
}

// Limited by the size of one mask panel (4096 x cap / 8 bytes) and by 32-bit indexing.
constexpr long long NMS_MAX_CAND = 1 << 20;

// `mask`: nseg * panel_rows * words words (one row panel), `bits`: nseg * words * 2 words
// (removed_g | keptbits), with words = ceil(cap / 64), panel_rows = 64 * min(words, NMS_PANEL_CHUNKS).
template <typename T>
static int nms_sorted_ws(const T* cand_boxes, int32_t* counters, int cap, int nseg, T thresh, int max_keep,
                         int32_t* keep, uint64_t* mask, uint64_t* bits, hipStream_t stream) {
    const int words = ceil_div(cap, 64);
    const int panel_chunks = words < NMS_PANEL_CHUNKS ? words : NMS_PANEL_CHUNKS;
    const int panel_rows = panel_chunks * 64;
    uint64_t* removed_g = bits;
    uint64_t* keptbits = bits + (size_t)nseg * words;
    const bool multi = words > panel_chunks;
    if (multi) FD_HIP(hipMemsetAsync(removed_g, 0, (size_t)nseg * words * sizeof(uint64_t), stream));
    for (int chunk0 = 0; chunk0 < words; chunk0 += panel_chunks) {
        const int rows_here = (words - chunk0 < panel_chunks ? words - chunk0 : panel_chunks);
        {
            // upper-triangle part of this panel: rows x remaining columns / 8 bytes written
            const double cols = (double)(words - chunk0) * 64.0;
            ProfScope scope(stream, "nms_mask", 0.0,
                            (double)nseg * (rows_here * 64.0 * cols / 8.0 * 0.5 + 4.0 * sizeof(T) * cols));
            hipLaunchKernelGGL(nms_mask_kernel<T>, dim3(words - chunk0, rows_here, nseg), dim3(64), 0, stream,
                               cand_boxes, counters, cap, words, chunk0, panel_rows, max_keep, thresh,
                               (multi && chunk0 > 0) ? removed_g : nullptr, mask);
            FD_LAUNCH_CHECK();
        }
        {
            ProfScope scope(stream, "nms_reduce", 0.0, (double)nseg * rows_here * 64.0 * rows_here * 8.0 * 0.5);
            hipLaunchKernelGGL(nms_reduce_kernel, dim3(nseg), dim3(REDUCE_THREADS), 0, stream, mask, counters,
                               removed_g, keptbits, cap, words, chunk0, panel_rows, max_keep, keep);
            FD_LAUNCH_CHECK();
        }
        const int later = words - (chunk0 + panel_chunks);
        if (later > 0) {
            ProfScope scope(stream, "nms_spread", 0.0, (double)nseg * rows_here * 64.0 * later * 8.0);
            hipLaunchKernelGGL(nms_spread_kernel, dim3(ceil_div(later, 256), ceil_div(panel_chunks, SPREAD_CHUNKS), nseg),
                               dim3(256), 0, stream, mask, counters, keptbits, removed_g, words, chunk0, panel_rows,
                               max_keep);
            FD_LAUNCH_CHECK();
        }
    }
    return FD_OK;
}

template <typename T>
int nms_sorted(fd_ctx* ctx, const T* cand_boxes, int32_t* counters, int cap, int nseg, T thresh, int max_keep,
               int32_t* keep, hipStream_t stream) {
    if (cap <= 0 || nseg <= 0) return FD_OK;
    if (cap > NMS_MAX_CAND) {
        set_error("nms: %d candidates per image exceed the limit of %lld", cap, NMS_MAX_CAND);
        return FD_ERR_LIMIT;
    }
    const int words = ceil_div(cap, 64);
    const int panel_chunks = words < NMS_PANEL_CHUNKS ? words : NMS_PANEL_CHUNKS;
    uint64_t *mask = nullptr, *bits = nullptr;
    FD_TRY(scratch_as(ctx, SLOT_NMS_MASK, (size_t)nseg * panel_chunks * 64 * words, &mask));
    FD_TRY(scratch_as(ctx, SLOT_MISC2, (size_t)nseg * words * 2, &bits));
    return nms_sorted_ws<T>(cand_boxes, counters, cap, nseg, thresh, max_keep, keep, mask, bits, stream);
}

template int nms_sorted<float>(fd_ctx*, const float*, int32_t*, int, int, float, int, int32_t*, hipStream_t);
template int nms_sorted<double>(fd_ctx*, const double*, int32_t*, int, int, double, int, int32_t*, hipStream_t);

// ---------------------------------------------------------------------------------
// Candidate gather: boxes / scores of the first n_cand sorted entries.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_candidates_kernel(const uint32_t* __restrict__ sorted_idx,
                                                               const T* __restrict__ boxes,
                                                               const float* __restrict__ scores, int n, int cap,
                                                               int32_t* __restrict__ counters, T* __restrict__ cand_boxes,
                                                               float* __restrict__ cand_scores,
                                                               int32_t* __restrict__ cand_index) {
    const int seg = blockIdx.y;
    const int n_valid = counters[seg * CNT_STRIDE + CNT_VALID];
    const int n_cand = min(n_valid, cap);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) counters[seg * CNT_STRIDE + CNT_CAND] = n_cand;
    if (r >= n_cand) return;
    const uint32_t idx = sorted_idx[(size_t)seg * n + r];
    const T* b = boxes + ((size_t)seg * n + idx) * 4;
    T* o = cand_boxes + ((size_t)seg * cap + r) * 4;
    o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
    if (cand_scores) cand_scores[(size_t)seg * cap + r] = scores[(size_t)seg * n + idx];
    cand_index[(size_t)seg * cap + r] = (int32_t)idx;
}

// ---------------------------------------------------------------------------------
// Shared decode arithmetic (processing.py:47-76): float32 deltas, exp() in float32,
// everything else in T.
// ---------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void decode_box(T bx1, T by1, T bx2, T by2, float dx, float dy, float dw, float dh,
                                           T& ox1, T& oy1, T& ox2, T& oy2) {
    const T w = bx2 - bx1 + T(1);
    const T h = by2 - by1 + T(1);
    const T cx = bx1 + T(0.5) * (w - T(1));
    const T cy = by1 + T(0.5) * (h - T(1));
    const T pcx = T(dx) * w + cx;
    const T pcy = T(dy) * h + cy;
    const T pw = T(expf(dw)) * w;
    const T ph = T(expf(dh)) * h;
    ox1 = pcx - T(0.5) * (pw - T(1));
    oy1 = pcy - T(0.5) * (ph - T(1));
    ox2 = pcx + T(0.5) * (pw - T(1));
    oy2 = pcy + T(0.5) * (ph - T(1));
}
__device__ __forceinline__ float tmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double tmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float tmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double tmin(double a, double b) { return fmin(a, b); }
template <typename T>
__device__ __forceinline__ T clampT(T v, T hi) { return tmax(tmin(v, hi), T(0)); }

// Anchor table and image info travel as kernel arguments: no host->device copy whose
// source would have to outlive the call.
constexpr int MAX_ANCHORS = 32;
template <typename T>
struct AnchorTable {
    T v[MAX_ANCHORS * 4];
};

// ---------------------------------------------------------------------------------
// RPN proposal: decode + clip + min-size filter for every anchor (proposal.py:60-103).
// Entry N-1-idx of the sort input holds anchor idx, so the stable sort breaks score
// ties towards the higher anchor index.
// ---------------------------------------------------------------------------------
constexpr int DECODE_PER_BLOCK = 1024;

template <typename T>
__global__ __launch_bounds__(256) void proposal_decode_kernel(const float* __restrict__ cls,
                                                             const float* __restrict__ bbox, RpnLayout lay, int softmax,
                                                             AnchorTable<T> anchors, int A, int H, int W,
                                                             int feat_stride, const float* __restrict__ im_info,
                                                             float info_h, float info_w, float info_s,
                                                             float min_size, T* __restrict__ boxes,
                                                             float* __restrict__ scores, uint32_t* __restrict__ keys,
                                                             uint32_t* __restrict__ vals, int32_t* __restrict__ counters,
                                                             int32_t* __restrict__ nan_flag) {
    // A workgroup handles DECODE_PER_BLOCK consecutive anchors and adds its count of valid ones to
    // the image's counter once: one same-address atomic per workgroup instead of one per wave
    // (the atomics to a single address serialise; per wave they dominated this kernel's time).
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = 0;
    __syncthreads();
    const int seg = blockIdx.y;
    const int N = H * W * A;
    int nvalid = 0;
    const bool vec_deltas = lay.box_chan == 1 && ((lay.box_batch | lay.box_off | lay.box_h | lay.box_w) & 3) == 0 &&
                            (reinterpret_cast<uintptr_t>(bbox) & 15u) == 0;
#pragma unroll 1
    for (int it = 0; it < DECODE_PER_BLOCK / 256; ++it) {
    const int idx = blockIdx.x * DECODE_PER_BLOCK + it * 256 + threadIdx.x;
    bool valid = false;
    if (idx < N) {
        const int a = idx % A;
        const int w = (idx / A) % W;
        const int h = idx / (A * W);
        const long long cbase = seg * lay.cls_batch + a * lay.cls_anchor + h * lay.cls_h + w * lay.cls_w;
        float score = cls[cbase + lay.cls_fg_off];
        if (softmax) {
            const float bg = cls[cbase + lay.cls_bg_off];
            const float m = fmaxf(bg, score);
            const float eb = expf(bg - m), ef = expf(score - m);
            score = ef / (eb + ef);
        }
        const long long bbase = seg * lay.box_batch + lay.box_off + h * lay.box_h + w * lay.box_w;
        float dx, dy, dw, dh;
        if (vec_deltas) {
            // channels-last head: the four deltas of an anchor are 16 contiguous, aligned bytes
            const f32x4 d = *reinterpret_cast<const f32x4*>(bbox + bbase + 4 * a);
            dx = d[0]; dy = d[1]; dw = d[2]; dh = d[3];
        } else {
            dx = bbox[bbase + (4 * a + 0) * lay.box_chan];
            dy = bbox[bbase + (4 * a + 1) * lay.box_chan];
            dw = bbox[bbase + (4 * a + 2) * lay.box_chan];
            dh = bbox[bbase + (4 * a + 3) * lay.box_chan];
        }
        if (nan_flag && (isnan(score) || isnan(dx) || isnan(dy) || isnan(dw) || isnan(dh))) atomicOr(nan_flag, 1);
        const T sx = T(w * feat_stride), sy = T(h * feat_stride);
        T x1, y1, x2, y2;
        decode_box<T>(anchors.v[a * 4 + 0] + sx, anchors.v[a * 4 + 1] + sy, anchors.v[a * 4 + 2] + sx,
                      anchors.v[a * 4 + 3] + sy, dx, dy, dw, dh, x1, y1, x2, y2);
        const float im_h = im_info ? im_info[seg * 3 + 0] : info_h;
        const float im_w = im_info ? im_info[seg * 3 + 1] : info_w;
        const float im_s = im_info ? im_info[seg * 3 + 2] : info_s;
        const T wmax = T(im_w - 1.0f), hmax = T(im_h - 1.0f);
        x1 = clampT(x1, wmax); y1 = clampT(y1, hmax);
        x2 = clampT(x2, wmax); y2 = clampT(y2, hmax);
        const T lim = T(min_size * im_s);
        valid = ((x2 - x1 + T(1)) >= lim) && ((y2 - y1 + T(1)) >= lim);
        typedef T box_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<box_t*>(boxes + ((size_t)seg * N + idx) * 4) = box_t{x1, y1, x2, y2};    // one 16- or 32-byte store
        scores[(size_t)seg * N + idx] = score;
        const size_t p = (size_t)seg * N + (N - 1 - idx);
        keys[p] = valid ? desc_key(score) : KEY_INVALID32;
        vals[p] = (uint32_t)idx;
    }
    nvalid += (int)__popcll(__ballot(valid));
    }
    if ((threadIdx.x & 63) == 0 && nvalid) atomicAdd(&s_valid, nvalid);
    __syncthreads();
    if (threadIdx.x == 0 && s_valid) atomicAdd(&counters[seg * CNT_STRIDE + CNT_VALID], s_valid);
}

template <typename T>
__global__ __launch_bounds__(256) void proposal_output_kernel(const T* __restrict__ cand_boxes,
                                                             const float* __restrict__ cand_scores,
                                                             const int32_t* __restrict__ keep,
                                                             const int32_t* __restrict__ counters, int cap, int post_n,
                                                             float* __restrict__ rois, float* __restrict__ roi_scores,
                                                             int32_t* __restrict__ counts_out) {
    const int seg = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nk = min(counters[seg * CNT_STRIDE + CNT_KEEP], post_n);
    if (i == 0 && counts_out) {
        counts_out[seg * 3 + 0] = counters[seg * CNT_STRIDE + CNT_VALID];
        counts_out[seg * 3 + 1] = counters[seg * CNT_STRIDE + CNT_CAND];
        counts_out[seg * 3 + 2] = nk;
    }
    if (i >= post_n) return;
    float* o = rois + ((size_t)seg * post_n + i) * 5;
    float sc = 0.0f;
    if (nk == 0) {
        o[0] = (float)seg; o[1] = o[2] = o[3] = o[4] = 0.0f;
    } else {
        const int r = keep[(size_t)seg * cap + (i % nk)];   // cyclic padding of a short keep list
        const T* b = cand_boxes + ((size_t)seg * cap + r) * 4;
        o[0] = (float)seg;
        o[1] = (float)b[0]; o[2] = (float)b[1]; o[3] = (float)b[2]; o[4] = (float)b[3];
        sc = cand_scores[(size_t)seg * cap + r];
    }
    if (roi_scores) roi_scores[(size_t)seg * post_n + i] = sc;
}

template <typename T>
static int proposal_run_t(fd_ctx* ctx, const fd_proposal_cfg* cfg, const float* cls, const float* bbox,
                          const RpnLayout& lay, int softmax, const float* im_info_dev, const float* info_uniform,
                          int batch, int H, int W,
                          float* rois, float* roi_scores, int32_t* counts, int32_t* nan_flag, hipStream_t stream) {
    const int A = cfg->num_ratios * cfg->num_scales;
    const long long N_ll = (long long)H * W * A;
    if (N_ll > (1ll << 24)) {
        set_error("proposal: %lld anchors per image exceed the limit of %d", N_ll, 1 << 24);
        return FD_ERR_LIMIT;
    }
    const int N = (int)N_ll;
    const int cap = (cfg->pre_nms_top_n > 0 && cfg->pre_nms_top_n < N) ? cfg->pre_nms_top_n : N;
    const int post_n = cfg->post_nms_top_n;

    // anchor table (host, float64, exact) -> device in T
    double ratios[8], scales[8], anc[64 * 4];
    for (int i = 0; i < cfg->num_ratios; ++i) ratios[i] = cfg->ratios[i];
    for (int i = 0; i < cfg->num_scales; ++i) scales[i] = cfg->scales[i];
    FD_TRY(fd_generate_anchors((double)cfg->feat_stride, ratios, cfg->num_ratios, scales, cfg->num_scales, anc));
    AnchorTable<T> anchors;
    memset(&anchors, 0, sizeof(anchors));
    for (int i = 0; i < A * 4; ++i) anchors.v[i] = (T)anc[i];

    T* boxes = nullptr; float* scores = nullptr;
    uint32_t *ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr;
    int32_t* counters = nullptr;
    T* cand_boxes = nullptr; float* cand_scores = nullptr; int32_t* cand_index = nullptr; int32_t* keep = nullptr;
    const size_t tot = (size_t)batch * N, ctot = (size_t)batch * cap;
    FD_TRY(scratch_as(ctx, SLOT_BOXES, tot * 4, &boxes));
    FD_TRY(scratch_as(ctx, SLOT_SCORES, tot, &scores));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_A, tot, &ka));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_B, tot, &kb));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_A, tot, &va));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_B, tot, &vb));
    FD_TRY(scratch_as(ctx, SLOT_COUNTERS, (size_t)batch * CNT_STRIDE, &counters));
    FD_TRY(scratch_as(ctx, SLOT_CAND_BOXES, ctot * 4, &cand_boxes));
    FD_TRY(scratch_as(ctx, SLOT_CAND_SCORES, ctot, &cand_scores));
    FD_TRY(scratch_as(ctx, SLOT_CAND_INDEX, ctot, &cand_index));
    FD_TRY(scratch_as(ctx, SLOT_KEEP, ctot, &keep));
    FD_HIP(hipMemsetAsync(counters, 0, sizeof(int32_t) * batch * CNT_STRIDE, stream));
    if (nan_flag) FD_HIP(hipMemsetAsync(nan_flag, 0, sizeof(int32_t), stream));

    {
        // per anchor: 2 scores + 4 deltas read, box + score + sort pair written
        ProfScope scope(stream, "proposal_decode", 0.0, (double)batch * N * (24.0 + 4.0 * sizeof(T) + 12.0));
        hipLaunchKernelGGL(proposal_decode_kernel<T>, dim3(ceil_div(N, DECODE_PER_BLOCK), batch), dim3(256), 0, stream, cls, bbox,
                           lay, softmax, anchors, A, H, W, cfg->feat_stride, im_info_dev, info_uniform[0],
                           info_uniform[1], info_uniform[2], cfg->min_size, boxes, scores, ka, va, counters, nan_flag);
        FD_LAUNCH_CHECK();
    }
    FD_TRY(sort_pairs_u32(ctx, ka, va, kb, vb, N, batch, stream));
    hipLaunchKernelGGL(gather_candidates_kernel<T>, dim3(ceil_div(cap, 256), batch), dim3(256), 0, stream, va, boxes,
                       scores, N, cap, counters, cand_boxes, cand_scores, cand_index);
    FD_LAUNCH_CHECK();
    FD_TRY(nms_sorted<T>(ctx, cand_boxes, counters, cap, batch, (T)cfg->nms_thresh, post_n, keep, stream));
    hipLaunchKernelGGL(proposal_output_kernel<T>, dim3(ceil_div(post_n, 256), batch), dim3(256), 0, stream, cand_boxes,
                       cand_scores, keep, counters, cap, post_n, rois, roi_scores, counts);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int proposal_run(fd_ctx* ctx, const fd_proposal_cfg* cfg, const float* cls, const float* bbox, const RpnLayout& lay,
                 int softmax, const float* im_info_dev, const float* info_uniform, int batch, int height, int width,
                 float* rois, float* roi_scores, int32_t* counts, int32_t* nan_flag, hipStream_t stream) {
    FD_REQUIRE(ctx && cfg && cls && bbox && rois && (im_info_dev || info_uniform), "proposal: NULL argument");
    static const float zero3[3] = {0.f, 0.f, 0.f};
    if (!info_uniform) info_uniform = zero3;
    FD_REQUIRE(batch > 0 && height > 0 && width > 0, "proposal: bad shape %d x %d x %d", batch, height, width);
    FD_REQUIRE(cfg->num_ratios > 0 && cfg->num_ratios <= 8 && cfg->num_scales > 0 && cfg->num_scales <= 8 &&
                   cfg->num_ratios * cfg->num_scales <= MAX_ANCHORS,
               "proposal: 1..8 ratios and scales, at most %d anchors per cell", MAX_ANCHORS);
    FD_REQUIRE(cfg->post_nms_top_n > 0, "proposal: post_nms_top_n must be > 0");
    FD_REQUIRE(cfg->feat_stride > 0, "proposal: feat_stride must be > 0");
    if (cfg->box_f64)
        return proposal_run_t<double>(ctx, cfg, cls, bbox, lay, softmax, im_info_dev, info_uniform, batch, height,
                                      width, rois, roi_scores, counts, nan_flag, stream);
    return proposal_run_t<float>(ctx, cfg, cls, bbox, lay, softmax, im_info_dev, info_uniform, batch, height, width,
                                 rois, roi_scores, counts, nan_flag, stream);
}

// ---------------------------------------------------------------------------------
// Stand-alone nms(dets, thresh).
// ---------------------------------------------------------------------------------
template <typename T, typename K>
__global__ __launch_bounds__(256) void nms_prepare_kernel(const T* __restrict__ dets, int n, T* __restrict__ boxes,
                                                         K* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T* d = dets + (size_t)i * 5;
    boxes[(size_t)i * 4 + 0] = d[0];
    boxes[(size_t)i * 4 + 1] = d[1];
    boxes[(size_t)i * 4 + 2] = d[2];
    boxes[(size_t)i * 4 + 3] = d[3];
    keys[n - 1 - i] = desc_key(d[4]);
    vals[n - 1 - i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void set_counters_kernel(int32_t* counters, int n_valid) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counters[CNT_VALID] = n_valid;
        counters[CNT_CAND] = 0;
        counters[CNT_KEEP] = 0;
    }
}

__global__ __launch_bounds__(256) void nms_finish_kernel(const int32_t* __restrict__ keep_ranks,
                                                        const int32_t* __restrict__ cand_index,
                                                        const int32_t* __restrict__ counters, int n,
                                                        int32_t* __restrict__ keep_out, int32_t* __restrict__ n_keep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nk = counters[CNT_KEEP];
    if (i == 0) *n_keep = nk;
    if (i < nk) keep_out[i] = cand_index[keep_ranks[i]];
}

// Every buffer nms(dets, thresh) needs beyond its arguments, for one call of n boxes.
template <typename T, typename K>
struct NmsWorkspace {
    T *boxes, *cand_boxes;            // n * 4 each
    K *ka, *kb;                       // n each
    uint32_t *va, *vb, *hist;         // n, n, 256 * ceil(n / SORT_TILE)
    int32_t *counters, *cand_index, *keep;   // CNT_STRIDE, n, n
    uint64_t *mask, *bits;            // one row panel; 2 * ceil(n / 64)
};

// The launch sequence of nms(dets, thresh) on a prepared workspace: allocates nothing.
template <typename T, typename K>
static int nms_standalone_ws(const NmsWorkspace<T, K>& w, const T* dets, int n, T thresh, int32_t* keep_out,
                             int32_t* n_keep, hipStream_t stream) {
    const int nb = ceil_div(n, 256);
    hipLaunchKernelGGL((nms_prepare_kernel<T, K>), dim3(nb), dim3(256), 0, stream, dets, n, w.boxes, w.ka, w.va);
    FD_LAUNCH_CHECK();
    hipLaunchKernelGGL(set_counters_kernel, dim3(1), dim3(256), 0, stream, w.counters, n);
    FD_LAUNCH_CHECK();
    FD_TRY((sort_pairs_ws<K>(w.ka, w.va, w.kb, w.vb, n, 1, w.hist, stream)));
    hipLaunchKernelGGL(gather_candidates_kernel<T>, dim3(nb, 1), dim3(256), 0, stream, w.va, w.boxes,
                       (const float*)nullptr, n, n, w.counters, w.cand_boxes, (float*)nullptr, w.cand_index);
    FD_LAUNCH_CHECK();
    FD_TRY(nms_sorted_ws<T>(w.cand_boxes, w.counters, n, 1, thresh, 0, w.keep, w.mask, w.bits, stream));
    hipLaunchKernelGGL(nms_finish_kernel, dim3(nb), dim3(256), 0, stream, w.keep, w.cand_index, w.counters, n, keep_out,
                       n_keep);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

template <typename T, typename K>
static int nms_standalone(fd_ctx* ctx, const T* dets, int n, T thresh, int32_t* keep_out, int32_t* n_keep,
                          hipStream_t stream) {
    FD_REQUIRE(ctx && n_keep && n >= 0, "nms: bad arguments");
    if (n == 0) {
        FD_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), stream));
        return FD_OK;
    }
    FD_REQUIRE(dets && keep_out, "nms: NULL dets / keep");
    if (n > NMS_MAX_CAND) {
        set_error("nms: %d boxes exceed the limit of %lld", n, NMS_MAX_CAND);
        return FD_ERR_LIMIT;
    }
    NmsWorkspace<T, K> w;
    const int words = ceil_div(n, 64);
    const int panel_chunks = words < NMS_PANEL_CHUNKS ? words : NMS_PANEL_CHUNKS;
    FD_TRY(scratch_as(ctx, SLOT_BOXES, (size_t)n * 4, &w.boxes));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_A, (size_t)n, &w.ka));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_B, (size_t)n, &w.kb));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_A, (size_t)n, &w.va));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_B, (size_t)n, &w.vb));
    FD_TRY(scratch_as(ctx, SLOT_SORT_HIST, (size_t)256 * ceil_div(n, SORT_TILE), &w.hist));
    FD_TRY(scratch_as(ctx, SLOT_COUNTERS, (size_t)CNT_STRIDE, &w.counters));
    FD_TRY(scratch_as(ctx, SLOT_CAND_BOXES, (size_t)n * 4, &w.cand_boxes));
    FD_TRY(scratch_as(ctx, SLOT_CAND_INDEX, (size_t)n, &w.cand_index));
    FD_TRY(scratch_as(ctx, SLOT_KEEP, (size_t)n, &w.keep));
    FD_TRY(scratch_as(ctx, SLOT_NMS_MASK, (size_t)panel_chunks * 64 * words, &w.mask));
    FD_TRY(scratch_as(ctx, SLOT_MISC2, (size_t)words * 2, &w.bits));
    return nms_standalone_ws<T, K>(w, dets, n, thresh, keep_out, n_keep, stream);
}

// ---------------------------------------------------------------------------------
// nms(dets, thresh) for callers that must not allocate: the workspace is a fixed device array of
// this library, one copy per GPU, carved up for at most NOALLOC_MAX_BOXES boxes (one mask panel).
// ---------------------------------------------------------------------------------
constexpr int NOALLOC_MAX_BOXES = NMS_PANEL_ROWS;                       // 4096
constexpr size_t NOALLOC_MASK_BYTES = (size_t)NOALLOC_MAX_BOXES * (NOALLOC_MAX_BOXES / 64) * 8;   // 2 MiB
constexpr size_t NOALLOC_VEC = (size_t)NOALLOC_MAX_BOXES * 4;          // one 32-bit value per box
constexpr size_t NOALLOC_OFF_BOXES = NOALLOC_MASK_BYTES;               // 4 floats per box
constexpr size_t NOALLOC_OFF_CAND = NOALLOC_OFF_BOXES + 4 * NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_KA = NOALLOC_OFF_CAND + 4 * NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_KB = NOALLOC_OFF_KA + NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_VA = NOALLOC_OFF_KB + NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_VB = NOALLOC_OFF_VA + NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_INDEX = NOALLOC_OFF_VB + NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_KEEP = NOALLOC_OFF_INDEX + NOALLOC_VEC;
constexpr size_t NOALLOC_OFF_HIST = NOALLOC_OFF_KEEP + NOALLOC_VEC;    // 256 * ceil(4096 / 2048) counters
constexpr size_t NOALLOC_OFF_BITS = NOALLOC_OFF_HIST + 256 * 2 * 4;    // 2 * 64 words
constexpr size_t NOALLOC_OFF_COUNTERS = NOALLOC_OFF_BITS + 2 * 64 * 8;
constexpr size_t NOALLOC_BYTES = NOALLOC_OFF_COUNTERS + 256;
static_assert(NOALLOC_MAX_BOXES <= NMS_PANEL_ROWS && ceil_div_c(NOALLOC_MAX_BOXES, SORT_TILE) == 2, "workspace plan");

}  // namespace fd

// File scope, outside the namespace (which already has a function of this name).
static __device__ __align__(256) unsigned char scratch[fd::NOALLOC_BYTES];

namespace fd {

// Device address of `scratch` on each GPU (every GPU holds its own copy at its own address).  In
// host code the array's name is only a handle for hipGetSymbolAddress, never an address.
constexpr int NOALLOC_MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;

// Resolves the array's address on `device` (which must be the current device); idempotent.
int noalloc_init_device(int device) {
    FD_REQUIRE(device >= 0 && device < NOALLOC_MAX_DEVICES, "no-alloc nms: device index %d out of range", device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return FD_OK;
    void* p = nullptr;
    FD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(::scratch)));
    FD_REQUIRE(p != nullptr, "no-alloc nms: hipGetSymbolAddress returned a null address");
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// bbox_pred (+ clip_boxes) as a stand-alone stage.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_pred_kernel(const float* __restrict__ boxes, const float* __restrict__ deltas,
                                                       int n, int C, int clip, double hmax, double wmax,
                                                       double* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n * C) return;
    const int i = t / C, c = t % C;
    const float* b = boxes + (size_t)i * 4;
    const float* d = deltas + (size_t)i * 4 * C + 4 * c;
    double x1, y1, x2, y2;
    decode_box<double>((double)b[0], (double)b[1], (double)b[2], (double)b[3], d[0], d[1], d[2], d[3], x1, y1, x2, y2);
    if (clip) {
        x1 = clampT(x1, wmax); y1 = clampT(y1, hmax);
        x2 = clampT(x2, wmax); y2 = clampT(y2, hmax);
    }
    double* o = out + (size_t)i * 4 * C + 4 * c;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
}

__global__ __launch_bounds__(256) void clip_boxes_kernel(double* __restrict__ boxes, int total, double hmax,
                                                        double wmax) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    boxes[t] = clampT(boxes[t], (t & 1) ? hmax : wmax);   // columns x1,y1,x2,y2 repeat with period 4
}

// ---------------------------------------------------------------------------------
// Detector tail (detection.py:47-58).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_decode_kernel(const float* __restrict__ rois, int roi_stride,
                                                         const float* __restrict__ scores,
                                                         const float* __restrict__ deltas, int R, double hmax,
                                                         double wmax, float thresh, float* __restrict__ boxes,
                                                         float* __restrict__ out_scores, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals, int32_t* __restrict__ counters) {
    const int seg = blockIdx.y;
    const int r = blockIdx.x * 256 + threadIdx.x;
    bool valid = false;
    if (r < R) {
        const size_t g = (size_t)seg * R + r;
        const float* b = rois + g * roi_stride + (roi_stride - 4);
        const float* d = deltas + g * 8 + 4;   // class 1
        double x1, y1, x2, y2;
        decode_box<double>((double)b[0], (double)b[1], (double)b[2], (double)b[3], d[0], d[1], d[2], d[3], x1, y1, x2,
                           y2);
        x1 = clampT(x1, wmax); y1 = clampT(y1, hmax);
        x2 = clampT(x2, wmax); y2 = clampT(y2, hmax);
        const float s = scores[g * 2 + 1];
        valid = s >= thresh;
        boxes[g * 4 + 0] = (float)x1; boxes[g * 4 + 1] = (float)y1;
        boxes[g * 4 + 2] = (float)x2; boxes[g * 4 + 3] = (float)y2;
        out_scores[g] = s;
        const size_t p = (size_t)seg * R + (R - 1 - r);
        keys[p] = valid ? desc_key(s) : KEY_INVALID32;
        vals[p] = (uint32_t)r;
    }
    const uint64_t m = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&counters[seg * CNT_STRIDE + CNT_VALID], (int)__popcll(m));
}

__global__ __launch_bounds__(256) void dets_output_kernel(const float* __restrict__ cand_boxes,
                                                         const float* __restrict__ cand_scores,
                                                         const int32_t* __restrict__ keep,
                                                         const int32_t* __restrict__ counters, int R, int do_round,
                                                         float* __restrict__ dets, int32_t* __restrict__ n_dets) {
    const int seg = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nk = counters[seg * CNT_STRIDE + CNT_KEEP];
    if (i == 0) n_dets[seg] = nk;
    if (i >= R) return;
    float* o = dets + ((size_t)seg * R + i) * 5;
    if (i >= nk) {
        o[0] = o[1] = o[2] = o[3] = o[4] = 0.0f;
        return;
    }
    const int r = keep[(size_t)seg * R + i];
    const float* b = cand_boxes + ((size_t)seg * R + r) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = b[k];
        if (do_round) v = (float)(int)rintf(v);   // round half to even, through int32 as the reference does
        o[k] = v;
    }
    o[4] = cand_scores[(size_t)seg * R + r];
}

int postprocess_run(fd_ctx* ctx, const float* rois, int roi_stride, const float* scores, const float* deltas,
                    int batch, int R, int im_h, int im_w, float thresh, float nms_thresh, int do_round, float* dets,
                    int32_t* n_dets, hipStream_t stream) {
    FD_REQUIRE(ctx && n_dets && batch > 0 && R >= 0, "postprocess: bad arguments");
    FD_REQUIRE(roi_stride == 4 || roi_stride == 5, "postprocess: roi rows must have 4 or 5 columns");
    if (R == 0) {
        FD_HIP(hipMemsetAsync(n_dets, 0, sizeof(int32_t) * batch, stream));
        return FD_OK;
    }
    FD_REQUIRE(rois && scores && deltas && dets, "postprocess: NULL argument");
    float *boxes = nullptr, *sc = nullptr, *cand_boxes = nullptr, *cand_scores = nullptr;
    uint32_t *ka = nullptr, *kb = nullptr, *va = nullptr, *vb = nullptr;
    int32_t *counters = nullptr, *cand_index = nullptr, *keep = nullptr;
    const size_t tot = (size_t)batch * R;
    FD_TRY(scratch_as(ctx, SLOT_BOXES, tot * 4, &boxes));
    FD_TRY(scratch_as(ctx, SLOT_SCORES, tot, &sc));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_A, tot, &ka));
    FD_TRY(scratch_as(ctx, SLOT_SORT_KEYS_B, tot, &kb));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_A, tot, &va));
    FD_TRY(scratch_as(ctx, SLOT_SORT_VALS_B, tot, &vb));
    FD_TRY(scratch_as(ctx, SLOT_COUNTERS, (size_t)batch * CNT_STRIDE, &counters));
    FD_TRY(scratch_as(ctx, SLOT_CAND_BOXES, tot * 4, &cand_boxes));
    FD_TRY(scratch_as(ctx, SLOT_CAND_SCORES, tot, &cand_scores));
    FD_TRY(scratch_as(ctx, SLOT_CAND_INDEX, tot, &cand_index));
    FD_TRY(scratch_as(ctx, SLOT_KEEP, tot, &keep));
    FD_HIP(hipMemsetAsync(counters, 0, sizeof(int32_t) * batch * CNT_STRIDE, stream));
    dim3 grid(ceil_div(R, 256), batch);
    hipLaunchKernelGGL(head_decode_kernel, grid, dim3(256), 0, stream, rois, roi_stride, scores, deltas, R,
                       (double)(im_h - 1), (double)(im_w - 1), thresh, boxes, sc, ka, va, counters);
    FD_LAUNCH_CHECK();
    FD_TRY(sort_pairs_u32(ctx, ka, va, kb, vb, R, batch, stream));
    hipLaunchKernelGGL(gather_candidates_kernel<float>, grid, dim3(256), 0, stream, va, boxes, sc, R, R, counters,
                       cand_boxes, cand_scores, cand_index);
    FD_LAUNCH_CHECK();
    FD_TRY(nms_sorted<float>(ctx, cand_boxes, counters, R, batch, nms_thresh, 0, keep, stream));
    hipLaunchKernelGGL(dets_output_kernel, grid, dim3(256), 0, stream, cand_boxes, cand_scores, keep, counters, R,
                       do_round, dets, n_dets);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

// ---------------------------------------------------------------------------------
// ROI max pooling (MXNet ROIPooling semantics: box corners scaled and rounded half away
// from zero, bins split with floor/ceil edges, empty bin -> 0).
// ---------------------------------------------------------------------------------
struct RoiBin {
    int b, hstart, hend, wstart, wend;
    bool empty;
};
__device__ __forceinline__ RoiBin roi_bin(const float* roi, float scale, int ph, int pw, int PH, int PW, int H, int W,
                                          int batch) {
    RoiBin r;
    r.b = (int)roi[0];
    const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
    const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
    const int rh = max(eh - sh + 1, 1), rw = max(ew - sw + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw);
    int he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
    r.hstart = min(max(hs + sh, 0), H);
    r.hend = min(max(he + sh, 0), H);
    r.wstart = min(max(ws + sw, 0), W);
    r.wend = min(max(we + sw, 0), W);
    r.empty = (r.hend <= r.hstart) || (r.wend <= r.wstart) || r.b < 0 || r.b >= batch;
    return r;
}

__global__ __launch_bounds__(256) void roipool_nchw_kernel(const float* __restrict__ data, int batch, int C, int H, int W,
                                                          const float* __restrict__ rois, int num_rois, int PH, int PW,
                                                          float scale, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)num_rois * C * PH * PW;
    if (t >= total) return;
    const int pw = (int)(t % PW);
    const int ph = (int)((t / PW) % PH);
    const int c = (int)((t / ((long long)PW * PH)) % C);
    const int n = (int)(t / ((long long)PW * PH * C));
    const RoiBin r = roi_bin(rois + (size_t)n * 5, scale, ph, pw, PH, PW, H, W, batch);
    float m = 0.0f;
    if (!r.empty) {
        m = -FLT_MAX;
        const float* p = data + ((size_t)r.b * C + c) * H * W;
        for (int h = r.hstart; h < r.hend; ++h)
            for (int w = r.wstart; w < r.wend; ++w) m = fmaxf(m, p[h * W + w]);
    }
    out[t] = m;
}

// channels-last: one block per (roi, bin); threads sweep the channels 4 at a time
__global__ __launch_bounds__(256) void roipool_nhwc_kernel(const float* __restrict__ data, int batch, int C, int H, int W,
                                                          const float* __restrict__ rois, int num_rois, int PH, int PW,
                                                          float scale, float* __restrict__ out) {
    const int bin = blockIdx.x, n = blockIdx.y;
    const int ph = bin / PW, pw = bin % PW;
    const RoiBin r = roi_bin(rois + (size_t)n * 5, scale, ph, pw, PH, PW, H, W, batch);
    float4* o = reinterpret_cast<float4*>(out + ((size_t)n * PH * PW + bin) * C);
    for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!r.empty) {
            m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
            for (int h = r.hstart; h < r.hend; ++h)
                for (int w = r.wstart; w < r.wend; ++w) {
                    const float4 v =
                        reinterpret_cast<const float4*>(data + (((size_t)r.b * H + h) * W + w) * C)[c4];
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
        }
        o[c4] = m;
    }
}

// The same pooling written as split(relu(max * bn_scale[c] + bn_shift[c])): the form the ROI head's
// first convolutions read.  One workgroup per (roi, bin) as above (one workgroup per roi sweeping
// its bins measured 1.7x slower: too few loads in flight); threads sweep the channels 8 at a time,
// each writing one 16-byte hi and one 16-byte lo chunk.
__global__ __launch_bounds__(128) void roipool_nhwc_split_kernel(const float* __restrict__ data, int batch, int C, int H,
                                                                int W, const float* __restrict__ rois, int PH, int PW,
                                                                float scale, const float* __restrict__ bn_scale,
                                                                const float* __restrict__ bn_shift,
                                                                uint8_t* __restrict__ out, unsigned* __restrict__ clamped,
                                                                int f32) {
    const int bin = blockIdx.x, n = blockIdx.y;
    const RoiBin r = roi_bin(rois + (size_t)n * 5, scale, bin / PW, bin % PW, PH, PW, H, W, batch);
    uint8_t* o = out + ((size_t)n * PH * PW + bin) * C * 4;
    unsigned bad = 0;
    for (int c8 = threadIdx.x; c8 < C / 8; c8 += 128) {
        f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0;
        if (!r.empty) {
            m0 = f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
            m1 = m0;
            for (int h = r.hstart; h < r.hend; ++h) {
                const f32x4* row = reinterpret_cast<const f32x4*>(data + ((size_t)r.b * H + h) * W * C) + c8 * 2;
                for (int w = r.wstart; w < r.wend; ++w) {
                    const f32x4* px = row + (size_t)w * (C / 4);
                    const f32x4 v0 = px[0], v1 = px[1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { m0[j] = fmaxf(m0[j], v0[j]); m1[j] = fmaxf(m1[j], v1[j]); }
                }
            }
        }
        const f32x4* sc = reinterpret_cast<const f32x4*>(bn_scale) + c8 * 2;
        const f32x4* sh = reinterpret_cast<const f32x4*>(bn_shift) + c8 * 2;
        const f32x4 s0 = sc[0], s1 = sc[1], t0 = sh[0], t1 = sh[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // fused multiply-add, as the standalone float32 -> split conversion rounds
            m0[j] = fmaxf(__builtin_fmaf(m0[j], s0[j], t0[j]), 0.f);
            m1[j] = fmaxf(__builtin_fmaf(m1[j], s1[j], t1[j]), 0.f);
        }
        if (f32) {
            float* ap = reinterpret_cast<float*>(o) + c8 * 8;
            *reinterpret_cast<f32x4*>(ap) = m0;
            *reinterpret_cast<f32x4*>(ap + 4) = m1;
            continue;
        }
        f16x8 hi, lo;
        split8(m0, m1, hi, lo, bad);
        uint8_t* sp = o + split_offset(c8 * 8);
        *reinterpret_cast<f16x8*>(sp) = hi;
        *reinterpret_cast<f16x8*>(sp + 64) = lo;
    }
    split_report(clamped, bad);
}

int roipool_nhwc_split(const float* data, int batch, int C, int H, int W, const float* rois, int num_rois, int PH,
                       int PW, float scale, const float* bn_scale, const float* bn_shift, void* out, unsigned* clamped,
                       hipStream_t stream, bool f32) {
    FD_REQUIRE(data && rois && out && bn_scale && bn_shift, "roipool: NULL argument");
    FD_REQUIRE(C % 32 == 0 && C > 0, "roipool (split output): channel count %d must be a multiple of 32", C);
    FD_REQUIRE(PH > 0 && PW > 0, "roipool (split output): bad bin grid");
    if (num_rois <= 0) return FD_OK;
    ProfScope scope(stream, "roipool", 0.0, 2.0 * 4.0 * (double)num_rois * PH * PW * C);
    hipLaunchKernelGGL(roipool_nhwc_split_kernel, dim3(PH * PW, num_rois), dim3(128), 0, stream, data, batch, C, H, W,
                       rois, PH, PW, scale, bn_scale, bn_shift, static_cast<uint8_t*>(out), clamped, f32 ? 1 : 0);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int roipool_nhwc(const float* data, int batch, int C, int H, int W, const float* rois, int num_rois, int PH, int PW,
                 float scale, float* out, hipStream_t stream) {
    FD_REQUIRE(data && rois && out, "roipool: NULL argument");
    FD_REQUIRE(C % 4 == 0 && C > 0, "roipool (channels-last): channel count %d must be a multiple of 4", C);
    if (num_rois <= 0) return FD_OK;
    ProfScope scope(stream, "roipool", 0.0, 2.0 * 4.0 * (double)num_rois * PH * PW * C);
    hipLaunchKernelGGL(roipool_nhwc_kernel, dim3(PH * PW, num_rois), dim3(256), 0, stream, data, batch, C, H, W, rois,
                       num_rois, PH, PW, scale, out);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

}  // namespace fd

// =================================================================================
// C ABI
// =================================================================================
extern "C" {

int fd_bbox_pred_f64(fd_ctx* ctx, const float* boxes, const float* deltas, int n, int num_classes, int clip,
                     double im_h, double im_w, double* out, void* stream) {
    (void)ctx;
    FD_REQUIRE(n >= 0 && num_classes > 0, "fd_bbox_pred_f64: bad shape");
    if (n == 0) return FD_OK;
    FD_REQUIRE(boxes && deltas && out, "fd_bbox_pred_f64: NULL argument");
    hipLaunchKernelGGL(fd::bbox_pred_kernel, dim3(fd::ceil_div((long long)n * num_classes, 256)), dim3(256), 0,
                       fd::as_stream(stream), boxes, deltas, n, num_classes, clip, im_h - 1.0, im_w - 1.0, out);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int fd_clip_boxes_f64(fd_ctx* ctx, double* boxes, int n, int num_classes, double im_h, double im_w, void* stream) {
    (void)ctx;
    FD_REQUIRE(n >= 0 && num_classes > 0, "fd_clip_boxes_f64: bad shape");
    if (n == 0) return FD_OK;
    FD_REQUIRE(boxes, "fd_clip_boxes_f64: NULL argument");
    const int total = n * num_classes * 4;
    hipLaunchKernelGGL(fd::clip_boxes_kernel, dim3(fd::ceil_div(total, 256)), dim3(256), 0, fd::as_stream(stream),
                       boxes, total, im_h - 1.0, im_w - 1.0);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int fd_nms_f32(fd_ctx* ctx, const float* dets, int n, float thresh, int32_t* keep, int32_t* n_keep, void* stream) {
    return fd::nms_standalone<float, uint32_t>(ctx, dets, n, thresh, keep, n_keep, fd::as_stream(stream));
}

int fd_nms_f32_noalloc(const float* dets, int n, float thresh, int32_t* keep, int32_t* n_keep, void* stream) {
    using namespace fd;
    FD_REQUIRE(n_keep && n >= 0, "fd_nms_f32_noalloc: bad arguments");
    if (n > NOALLOC_MAX_BOXES) {
        set_error("fd_nms_f32_noalloc: %d boxes exceed the fixed workspace's limit of %d; use fd_nms_f32", n,
                  NOALLOC_MAX_BOXES);
        return FD_ERR_LIMIT;
    }
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        FD_HIP(hipMemsetAsync(n_keep, 0, sizeof(int32_t), st));
        return FD_OK;
    }
    FD_REQUIRE(dets && keep, "fd_nms_f32_noalloc: NULL dets / keep");
    int device = -1;
    FD_HIP(hipGetDevice(&device));
    FD_REQUIRE(device >= 0 && device < NOALLOC_MAX_DEVICES, "fd_nms_f32_noalloc: device index %d out of range", device);
    FD_TRY(noalloc_init_device(device));       // resolves the address on the first call on this GPU (under the mutex)
    unsigned char* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        base = g_scratch_ptr[device];
    }
    NmsWorkspace<float, uint32_t> w;
    w.mask = reinterpret_cast<uint64_t*>(base);
    w.boxes = reinterpret_cast<float*>(base + NOALLOC_OFF_BOXES);
    w.cand_boxes = reinterpret_cast<float*>(base + NOALLOC_OFF_CAND);
    w.ka = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_KA);
    w.kb = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_KB);
    w.va = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_VA);
    w.vb = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_VB);
    w.cand_index = reinterpret_cast<int32_t*>(base + NOALLOC_OFF_INDEX);
    w.keep = reinterpret_cast<int32_t*>(base + NOALLOC_OFF_KEEP);
    w.hist = reinterpret_cast<uint32_t*>(base + NOALLOC_OFF_HIST);
    w.bits = reinterpret_cast<uint64_t*>(base + NOALLOC_OFF_BITS);
    w.counters = reinterpret_cast<int32_t*>(base + NOALLOC_OFF_COUNTERS);
    // Nothing in the workspace has to start at zero: the counters are set by the first kernels, the
    // histogram, keys, mask words and kept bits a call reads are all written by that call first.
    return nms_standalone_ws<float, uint32_t>(w, dets, n, thresh, keep, n_keep, st);
}

int fd_nms_f64(fd_ctx* ctx, const double* dets, int n, double thresh, int32_t* keep, int32_t* n_keep, void* stream) {
    return fd::nms_standalone<double, uint64_t>(ctx, dets, n, thresh, keep, n_keep, fd::as_stream(stream));
}

int fd_proposal(fd_ctx* ctx, const fd_proposal_cfg* cfg, const float* cls_prob, const float* bbox_pred,
                const float* im_info_host, int batch, int height, int width, float* rois, float* roi_scores,
                int32_t* counts, int32_t* nan_flag, void* stream) {
    FD_REQUIRE(ctx && cfg && im_info_host && batch > 0, "fd_proposal: bad arguments");
    const int A = cfg->num_ratios * cfg->num_scales;
    const long long HW = (long long)height * width;
    fd::RpnLayout lay;
    lay.cls_batch = 2ll * A * HW; lay.cls_anchor = HW; lay.cls_h = width; lay.cls_w = 1;
    lay.cls_bg_off = 0; lay.cls_fg_off = (long long)A * HW;
    lay.box_batch = 4ll * A * HW; lay.box_chan = HW; lay.box_h = width; lay.box_w = 1; lay.box_off = 0;
    bool uniform = true;
    for (int b = 1; b < batch && uniform; ++b)
        for (int k = 0; k < 3; ++k) uniform = uniform && (im_info_host[b * 3 + k] == im_info_host[k]);
    float* d_info = nullptr;
    if (!uniform) {
        // per-image sizes: a blocking copy, so the caller's array need not outlive the call
        FD_TRY(fd::scratch_as(ctx, fd::SLOT_IMINFO, (size_t)batch * 3, &d_info));
        FD_HIP(hipStreamSynchronize(fd::as_stream(stream)));
        FD_HIP(hipMemcpy(d_info, im_info_host, sizeof(float) * 3 * batch, hipMemcpyHostToDevice));
    }
    return fd::proposal_run(ctx, cfg, cls_prob, bbox_pred, lay, 0, d_info, im_info_host, batch, height, width, rois,
                            roi_scores, counts, nan_flag, fd::as_stream(stream));
}

int fd_roipool(fd_ctx* ctx, const float* data, int batch, int channels, int height, int width, const float* rois,
               int num_rois, int pooled_h, int pooled_w, float spatial_scale, float* out, void* stream) {
    (void)ctx;
    FD_REQUIRE(batch > 0 && channels > 0 && height > 0 && width > 0 && pooled_h > 0 && pooled_w > 0 && num_rois >= 0,
               "fd_roipool: bad shape");
    if (num_rois == 0) return FD_OK;
    FD_REQUIRE(data && rois && out, "fd_roipool: NULL argument");
    const long long total = (long long)num_rois * channels * pooled_h * pooled_w;
    hipLaunchKernelGGL(fd::roipool_nchw_kernel, dim3(fd::ceil_div(total, 256)), dim3(256), 0, fd::as_stream(stream),
                       data, batch, channels, height, width, rois, num_rois, pooled_h, pooled_w, spatial_scale, out);
    FD_LAUNCH_CHECK();
    return FD_OK;
}

int fd_postprocess(fd_ctx* ctx, const float* rois, const float* scores, const float* deltas, int num_rois, int im_h,
                   int im_w, float thresh, float nms_thresh, int do_round, float* dets, int32_t* n_dets,
                   void* stream) {
    return fd::postprocess_run(ctx, rois, 4, scores, deltas, 1, num_rois, im_h, im_w, thresh, nms_thresh, do_round,
                               dets, n_dets, fd::as_stream(stream));
}

}  // extern "C"
