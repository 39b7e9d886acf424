// Batched per-class greedy NMS on gfx950, bit-exact with the reference's
// utils/postprocess.py:39-51 (iou :21-36).
//
// One workgroup per (image, class).  The reference's `boxes.sort(key=conf[c], reverse=True)` is a
// STABLE sort applied to the list order left by the previous class, so the effective key of class
// c is lexicographic (conf[:,c] desc, conf[:,c-1] desc, ..., conf[:,0] desc, box index asc) on the
// ORIGINAL scores -- which makes every class independent.  Phase 1 ranks the N boxes with that
// comparator (rank = number of boxes that sort before me; O(N^2/256) per thread from LDS
// broadcasts).  Phase 2 is the greedy scan, done by ONE wavefront: for each surviving box above
// the threshold, every lane tests one sorted position per 64-box chunk and __ballot() turns the
// 64 IoU compares into one 64-bit word of the overlap bitmask that is OR-ed into the chunk's
// `removed` word (kept in the lane that owns the chunk).  IoU is evaluated in fp32 in the
// reference's operation order ((a1+a2)-inter, floor 1e-10, `>=`) with FP contraction off.
// Finally the removed boxes' scores in column c are zeroed in place, as the reference mutates
// its input.
#include "common.h"
#pragma clang fp contract(off)

__device__ __forceinline__ bool sorts_before(const float *__restrict__ conf0, long jrow, long irow, int j, int i, int c, int C, float kj, float ki) {
    if (kj != ki) return kj > ki;
    for (int cc = c - 1; cc >= 0; --cc) {  // carried order of the earlier classes' stable sorts
        float a = conf0[jrow * C + cc], b = conf0[irow * C + cc];
        if (a != b) return a > b;
    }
    return j < i;
}

__global__ __launch_bounds__(256) void nms_kernel(float *__restrict__ conf, const float *__restrict__ conf0, const float *__restrict__ xy_min,
                                                  const float *__restrict__ xy_max, int *__restrict__ order_out, int N, int C,
                                                  float thr, float thr_iou) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *key = reinterpret_cast<float *>(smem_raw);      // [N] original scores of this class (box order)
    float *skey = key + N;                                  // [N] scores in sorted order
    int *sidx = reinterpret_cast<int *>(skey + N);          // [N] sorted position -> box index
    f32x4 *sbox = reinterpret_cast<f32x4 *>(sidx + N + ((4 - (3 * N) % 4) % 4));  // [N] (minx,miny,maxx,maxy), 16-B aligned

    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const long base = (long)b * N;
    const int tid = threadIdx.x;

    for (int i = tid; i < N; i += 256) key[i] = conf0[(base + i) * C + c];
    __syncthreads();
    auto place = [&](int rank, int i) {
        sidx[rank] = i;
        skey[rank] = key[i];
        f32x4 bx;
        bx[0] = xy_min[(base + i) * 2]; bx[1] = xy_min[(base + i) * 2 + 1];
        bx[2] = xy_max[(base + i) * 2]; bx[3] = xy_max[(base + i) * 2 + 1];
        sbox[rank] = bx;
    };
    // The exact order of the boxes at or below the threshold never matters to the scan (they cannot suppress, and they all
    // sort after every box that can), so only the K boxes above it are ranked -- K^2 instead of N^2 comparisons (the full
    // sort was 19 % of a batch-256 detect).  The complete order is still produced where it is observable: for the class
    // whose order is reported (order_out), and for negative thresholds.
    const bool full = !(0.0f <= thr) || (order_out && c == C - 1);
    if (full) {
        for (int i = tid; i < N; i += 256) {
            const float ki = key[i];
            int rank = 0;
            for (int j = 0; j < N; ++j) rank += sorts_before(conf0, base + j, base + i, j, i, c, C, key[j], ki) ? 1 : 0;
            place(rank, i);
        }
    } else {
        int *cidx = reinterpret_cast<int *>(sbox + N);       // [N] box indices of the candidates, in box order
        int *cnt = cidx + N;                                  // [256] candidates per thread range
        const int per = (N + 255) / 256;
        const int lo = min(tid * per, N), hi = min(lo + per, N);
        int nc = 0;
        for (int i = lo; i < hi; ++i) nc += key[i] > thr ? 1 : 0;
        cnt[tid] = nc;
        __syncthreads();
        int before = 0, K = 0;
        for (int t = 0; t < 256; ++t) {
            const int v = cnt[t];
            before += t < tid ? v : 0;
            K += v;
        }
        int ci = before, ni = K + (lo - before);             // candidates keep box order in cidx; the others fill positions K..N-1
        for (int i = lo; i < hi; ++i) {
            if (key[i] > thr) cidx[ci++] = i;
            else place(ni++, i);
        }
        __syncthreads();
        for (int q = tid; q < K; q += 256) {
            const int i = cidx[q];
            const float ki = key[i];
            int rank = 0;
            for (int r = 0; r < K; ++r) {
                const int j = cidx[r];
                rank += sorts_before(conf0, base + j, base + i, j, i, c, C, key[j], ki) ? 1 : 0;
            }
            place(rank, i);
        }
    }
    __syncthreads();
    if (c == C - 1 && order_out)
        for (int p = tid; p < N; p += 256) order_out[base + p] = sidx[p];
    if (tid >= 64) return;

    // ---- greedy scan by one wavefront ----
    const int lane = tid;
    const int nchunks = (N + 63) >> 6;           // <= 64 (N <= 4096)
    unsigned long long myword = 0ull;            // lane k owns the `removed` word of chunk k
    for (int p = 0; p + 1 < N; ++p) {
        const float kp = skey[p];
        if (kp <= thr && 0.0f <= thr) break;     // sorted descending: nothing later can suppress
        const unsigned long long wp = __shfl(myword, p >> 6, 64);
        const bool removed = (wp >> (p & 63)) & 1ull;
        const float cur = removed ? 0.0f : kp;
        if (cur <= thr) continue;                // utils/postprocess.py:46-47
        const f32x4 bp = sbox[p];
        const float a1 = (bp[2] - bp[0]) * (bp[3] - bp[1]);
        for (int k = p >> 6; k < nchunks; ++k) {
            const int pos = (k << 6) + lane;
            bool hit = false;
            if (pos > p && pos < N) {
                const f32x4 bq = sbox[pos];
                const float a2 = (bq[2] - bq[0]) * (bq[3] - bq[1]);
                const float w = fmaxf(fminf(bp[2], bq[2]) - fmaxf(bp[0], bq[0]), 0.0f);
                const float h = fmaxf(fminf(bp[3], bq[3]) - fmaxf(bp[1], bq[1]), 0.0f);
                const float inter = w * h;
                const float iou = inter / fmaxf((a1 + a2) - inter, 1e-10f);
                hit = iou >= thr_iou;             // :49
            }
            const unsigned long long word = __ballot(hit);
            if (lane == k) myword |= word;
        }
    }
    for (int k = 0; k < nchunks; ++k) {
        const unsigned long long word = __shfl(myword, k, 64);
        const int pos = (k << 6) + lane;
        if (pos < N && ((word >> lane) & 1ull)) conf[(base + sidx[pos]) * C + c] = 0.0f;   // :50
    }
}

extern "C" int yolo2_nms(float *conf, const float *xy_min, const float *xy_max, int *order_out, int *ws, int B, int N, int C,
                         float threshold, float threshold_iou, void *stream) {
    Y2_CHECK_ARG(conf && xy_min && xy_max && ws);
    Y2_CHECK_ARG(B > 0 && N > 0 && N <= 4096 && C > 0);
    hipStream_t st = (hipStream_t)stream;
    // snapshot of the original scores: the class blocks read neighbours' columns for tie-breaks
    // while those columns are being zeroed by their own blocks
    if (hipMemcpyAsync(ws, conf, sizeof(float) * (size_t)B * N * C, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        yolo2_set_error("nms: snapshot copy failed");
        return YOLO2_E_LAUNCH;
    }
    const size_t lds = sizeof(float) * 3 * (size_t)N + 16 + sizeof(float) * 4 * (size_t)N + sizeof(int) * ((size_t)N + 256);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        yolo2_set_error("nms: cannot reserve %zu bytes of LDS", lds);
        return YOLO2_E_LAUNCH;
    }
    nms_kernel<<<B * C, 256, lds, st>>>(conf, (const float *)ws, xy_min, xy_max, order_out, N, C, threshold, threshold_iou);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
