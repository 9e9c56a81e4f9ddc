/* Plain-C restatement of the reference's greedy NMS (test infrastructure; see oracle/__init__.py).
 *
 * Follows code/detection/symbol/processing.py:114-148 of the reference: areas with the +1
 * pixel-extent convention, boxes visited by descending score, a box survives a kept one iff
 * inter / (area_i + area_j - inter) <= thresh, arithmetic in the dtype of the input.  Equal
 * scores are visited higher index first (the order a reversed stable argsort gives; the
 * reference's own order for ties is undefined).
 *
 * Used where the NumPy oracle is too slow (100k+ boxes) and as the timed CPU baseline of the
 * NMS stage.  Parity: pinned through tests/test_oracle_golden.py against the vectors the
 * reference's own module produced.  Build: make -C oracle  (-ffp-contract=off: no fused a*b+c). */
#include <stdint.h>
#include <stdlib.h>

#define DEFINE_NMS(NAME, T)                                                                   \
    typedef struct { T score; int32_t idx; } NAME##_item;                                     \
    static int NAME##_cmp(const void* a, const void* b) {                                     \
        const NAME##_item* x = (const NAME##_item*)a;                                         \
        const NAME##_item* y = (const NAME##_item*)b;                                         \
        if (x->score > y->score) return -1;                                                   \
        if (x->score < y->score) return 1;                                                    \
        return (x->idx > y->idx) ? -1 : (x->idx < y->idx);                                    \
    }                                                                                         \
    int NAME(const T* dets, int n, T thresh, int32_t* keep) {                                 \
        if (n <= 0) return 0;                                                                 \
        NAME##_item* order = (NAME##_item*)malloc(sizeof(NAME##_item) * (size_t)n);           \
        T* area = (T*)malloc(sizeof(T) * (size_t)n);                                          \
        unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);                           \
        for (int i = 0; i < n; ++i) {                                                         \
            const T* d = dets + (size_t)i * 5;                                                \
            area[i] = (d[2] - d[0] + (T)1) * (d[3] - d[1] + (T)1);                            \
            order[i].score = d[4];                                                            \
            order[i].idx = i;                                                                 \
        }                                                                                     \
        qsort(order, (size_t)n, sizeof(NAME##_item), NAME##_cmp);                             \
        int nk = 0;                                                                           \
        for (int p = 0; p < n; ++p) {                                                         \
            const int i = order[p].idx;                                                       \
            if (dead[i]) continue;                                                            \
            keep[nk++] = i;                                                                   \
            const T* a = dets + (size_t)i * 5;                                                \
            for (int q = p + 1; q < n; ++q) {                                                 \
                const int j = order[q].idx;                                                   \
                if (dead[j]) continue;                                                        \
                const T* b = dets + (size_t)j * 5;                                            \
                const T xx1 = a[0] > b[0] ? a[0] : b[0];                                      \
                const T yy1 = a[1] > b[1] ? a[1] : b[1];                                      \
                const T xx2 = a[2] < b[2] ? a[2] : b[2];                                      \
                const T yy2 = a[3] < b[3] ? a[3] : b[3];                                      \
                T w = xx2 - xx1 + (T)1, h = yy2 - yy1 + (T)1;                                 \
                if (w < (T)0) w = (T)0;                                                       \
                if (h < (T)0) h = (T)0;                                                       \
                const T inter = w * h;                                                        \
                const T iou = inter / (area[i] + area[j] - inter);                            \
                if (!(iou <= thresh)) dead[j] = 1;                                            \
            }                                                                                 \
        }                                                                                     \
        free(order); free(area); free(dead);                                                  \
        return nk;                                                                            \
    }

DEFINE_NMS(oracle_nms_f32, float)
DEFINE_NMS(oracle_nms_f64, double)
