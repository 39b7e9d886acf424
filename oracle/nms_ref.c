/* CPU ORACLE -- test infrastructure only (see oracle/yolo2_ref.py header).
 *
 * Plain-C restatement of utils/postprocess.py:21-51 (iou + non_max_suppress) of
 * ruiminshen/yolo-tf, for parity runs at sizes where the Python loop takes minutes and for
 * bench.py's cpu_baseline leg.  fp32 arithmetic in the reference's operation order
 * ((a1+a2)-inter, floor 1e-10, `>=` on the IoU, `<=` on the score); compile with
 * -ffp-contract=off so the compiler cannot fuse w*h+a2.
 * Pinned against the golden vectors produced by the reference itself (tests/golden/nms_*.npz).
 */
#include <stdlib.h>
#include <string.h>

static float iou_f32(const float *mn1, const float *mx1, const float *mn2, const float *mx2) {
    /* utils/postprocess.py:28-36 */
    float a1 = (mx1[0] - mn1[0]) * (mx1[1] - mn1[1]);
    float a2 = (mx2[0] - mn2[0]) * (mx2[1] - mn2[1]);
    float lo0 = mn1[0] > mn2[0] ? mn1[0] : mn2[0];
    float lo1 = mn1[1] > mn2[1] ? mn1[1] : mn2[1];
    float hi0 = mx1[0] < mx2[0] ? mx1[0] : mx2[0];
    float hi1 = mx1[1] < mx2[1] ? mx1[1] : mx2[1];
    float w = hi0 - lo0; if (!(w > 0.0f)) w = 0.0f;
    float h = hi1 - lo1; if (!(h > 0.0f)) h = 0.0f;
    float inter = w * h;
    float u = (a1 + a2) - inter;
    if (!(u > 1e-10f)) u = 1e-10f;
    return inter / u;
}

/* stable merge sort of `order` by key descending (Python list.sort(reverse=True) is stable) */
static void msort(long *order, long *tmp, const float *key, long n) {
    if (n < 2) return;
    long h = n / 2;
    msort(order, tmp, key, h);
    msort(order + h, tmp, key, n - h);
    long i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (key[order[j]] > key[order[i]]) tmp[k++] = order[j++];   /* strictly greater moves ahead */
        else tmp[k++] = order[i++];
    }
    while (i < h) tmp[k++] = order[i++];
    while (j < n) tmp[k++] = order[j++];
    memcpy(order, tmp, (size_t)n * sizeof(long));
}

/* conf [n][classes] (mutated in place), xy_min/xy_max [n][2]; order_out [n] receives the
 * reference's returned list order (after the last class's sort).  Returns 0. */
int nms_ref(float *conf, const float *xy_min, const float *xy_max, long n, long classes,
            float threshold, float threshold_iou, long *order_out) {
    long *order = order_out;
    long *tmp = (long *)malloc((size_t)n * sizeof(long));
    float *key = (float *)malloc((size_t)n * sizeof(float));
    if (!tmp || !key) return 1;
    for (long i = 0; i < n; ++i) order[i] = i;
    for (long c = 0; c < classes; ++c) {
        for (long i = 0; i < n; ++i) key[i] = conf[i * classes + c];
        msort(order, tmp, key, n);                               /* :43 */
        for (long p = 0; p + 1 < n; ++p) {                       /* :44 */
            long i = order[p];
            if (conf[i * classes + c] <= threshold) continue;    /* :46-47 */
            for (long q = p + 1; q < n; ++q) {                   /* :48 */
                long j = order[q];
                if (iou_f32(xy_min + 2 * i, xy_max + 2 * i, xy_min + 2 * j, xy_max + 2 * j) >= threshold_iou)
                    conf[j * classes + c] = 0.0f;                /* :49-50 */
            }
        }
    }
    free(tmp); free(key);
    return 0;
}
