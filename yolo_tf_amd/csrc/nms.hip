// Batched per-class greedy NMS on gfx950, bit-exact with the reference's
// utils/postprocess.py:39-51 (iou :21-36).
//
// One workgroup per (image, class).  The reference's `boxes.sort(key=conf[c], reverse=True)` is a
// STABLE sort applied to the list order left by the previous class, so the effective key of class
// c is lexicographic (conf[:,c] desc, conf[:,c-1] desc, ..., conf[:,0] desc, box index asc) on the
// ORIGINAL scores -- which makes every class independent.  Phase 1 sorts the boxes with that
// comparator: a bitonic network over box indices in LDS (round 3; the first version counted, for every
// box, the boxes that sort before it -- O(N^2/256) comparisons per thread, 11 % of a sparse batch-256
// detect for the one class whose full order is observable).  Phase 2 is the greedy scan, 64 candidates
// (boxes above the threshold, in sorted order) at a time: all four waves build the group's rows of the
// overlap bitmask -- every lane tests one sorted position per 64-box chunk and __ballot() turns the 64
// IoU compares into one 64-bit word -- and one wave then walks the group, OR-ing the rows of the
// candidates that are still alive into the `removed` words (kept in the lane that owns the chunk): the
// serial part is one LDS row per candidate instead of an IoU loop (the first version's whole scan ran on
// one wavefront: 4.9 ms of a dense batch-256 detect).  IoU is evaluated in fp32 in the
// reference's operation order ((a1+a2)-inter, floor 1e-10, `>=`) with FP contraction off.
// Finally the removed boxes' scores in column c are zeroed in place, as the reference mutates
// its input.
#include "common.h"
#include <atomic>
#include <stdlib.h>
#pragma clang fp contract(off)

__device__ __forceinline__ bool sorts_before(const float *__restrict__ conf0, long jrow, long irow, int j, int i, int c, int C, float kj, float ki) {
    if (kj != ki) return kj > ki;
    for (int cc = c - 1; cc >= 0; --cc) {  // carried order of the earlier classes' stable sorts
        float a = conf0[jrow * C + cc], b = conf0[irow * C + cc];
        if (a != b) return a > b;
    }
    return j < i;
}

#define Y2_NMS_GROUP 64      // candidates whose overlap rows are built together (one sorted 64-box chunk)
__global__ __launch_bounds__(256) void nms_kernel(float *__restrict__ conf, const float *__restrict__ conf0, const float *__restrict__ xy_min,
                                                  const float *__restrict__ xy_max, int *__restrict__ order_out, int N, int C,
                                                  float thr, float thr_iou, int NP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *skey = reinterpret_cast<float *>(smem_raw);     // [N] scores in sorted order
    int *sidx = reinterpret_cast<int *>(skey + N);          // [N] sorted position -> box index
    f32x4 *sbox = reinterpret_cast<f32x4 *>(sidx + N + ((4 - (2 * N) % 4) % 4));  // [N] (minx,miny,maxx,maxy), 16-B aligned
    float *key = reinterpret_cast<float *>(sbox + N);       // [N] original scores of this class (box order): sort phase only
    int *perm = reinterpret_cast<int *>(key + N);           // [NP] box indices being sorted (-1 = padding, sorts last); NP = pow2 >= N: sort phase only
    int *cnt = perm + NP;                                   // [256] candidates per thread range: sort phase only
    // [Y2_NMS_GROUP][nchunks] overlap words of one candidate group: the scan's only scratch, laid over the sort phase's (key, perm, cnt),
    // which are dead by then (4 N + 4 NP + 1024 >= 8 (64 nchunks + 1) bytes; with its own storage N = 4096 asked for 166 KB of the 160 KB LDS)
    unsigned long long *rows = reinterpret_cast<unsigned long long *>(key);

    const int b = blockIdx.x / C, c = blockIdx.x % C;
    const long base = (long)b * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nchunks = (N + 63) >> 6;           // <= 64 (N <= 4096)

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
    // bitonic network on perm[0..np): the comparator is a strict total order (ties end at the box index), so the result is THE sorted
    // sequence whatever the network's exchange order
    auto bitonic = [&](int np) {
        for (int k = 2; k <= np; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                __syncthreads();
                for (int t = tid; t < (np >> 1); t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const int a = perm[i], bb = perm[l];
                    // which of the two belongs first?  padding (-1) belongs after everything
                    const bool b_first = bb >= 0 && (a < 0 || sorts_before(conf0, base + bb, base + a, bb, a, c, C, key[bb], key[a]));
                    const bool up = (i & k) == 0;          // ascending block: the element that sorts first takes the lower position
                    if (up == b_first && (a >= 0 || bb >= 0)) { perm[i] = bb; perm[l] = a; }
                }
            }
        __syncthreads();
    };
    // candidates = boxes above the threshold: only they can suppress, and they all sort before every other box.  The exact order of
    // the others never matters to the scan, so only the K candidates are sorted; the complete order is still produced where it is
    // observable: for the class whose order is reported (order_out), and for negative thresholds.
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
    const bool full = !(0.0f <= thr) || (order_out && c == C - 1);
    if (full) {
        for (int i = tid; i < NP; i += 256) perm[i] = i < N ? i : -1;
        bitonic(NP);
        for (int p = tid; p < N; p += 256) place(p, perm[p]);
        if (!(0.0f <= thr)) K = N;               // negative threshold: every box is a candidate (the scan below tests the current score)
    } else {
        int np = 1;
        while (np < K) np <<= 1;
        for (int i = K + tid; i < np; i += 256) perm[i] = -1;
        int ci = before, ni = K + (lo - before);             // candidates keep box order in perm; the others fill positions K..N-1
        for (int i = lo; i < hi; ++i) {
            if (key[i] > thr) perm[ci++] = i;
            else place(ni++, i);
        }
        bitonic(np);
        for (int p = tid; p < K; p += 256) place(p, perm[p]);
    }
    __syncthreads();
    if (c == C - 1 && order_out)
        for (int p = tid; p < N; p += 256) order_out[base + p] = sidx[p];

    // ---- greedy scan.  Per group of 64 candidates: all four waves build the group's overlap rows (row p, word k = which of the boxes
    // 64k .. 64k+63 candidate p suppresses: one __ballot per word; independent of what has been removed so far), then ONE wave walks
    // the group in order, OR-ing the rows of the candidates that are still alive into the `removed` words (lane k owns chunk k).
    // IoU in the reference's operation order ((a1+a2)-inter, floor 1e-10, `>=`), FP contraction off.
    unsigned long long myword = 0ull;
    for (int g0 = 0; g0 < K && g0 + 1 < N; g0 += Y2_NMS_GROUP) {
        const int gend = min(g0 + Y2_NMS_GROUP, K);
        for (int p = g0 + wave; p < gend; p += 4) {
            const f32x4 bp = sbox[p];
            const float a1 = (bp[2] - bp[0]) * (bp[3] - bp[1]);
            for (int k = g0 >> 6; k < nchunks; ++k) {
                const int pos = (k << 6) + lane;
                bool hit = false;
                if (pos > p && pos < N) {
                    const f32x4 bq = sbox[pos];
                    const float a2 = (bq[2] - bq[0]) * (bq[3] - bq[1]);
                    const float w = fmaxf(fminf(bp[2], bq[2]) - fmaxf(bp[0], bq[0]), 0.0f);
                    const float h = fmaxf(fminf(bp[3], bq[3]) - fmaxf(bp[1], bq[1]), 0.0f);
                    const float inter = w * h;
                    const float iou = inter / fmaxf((a1 + a2) - inter, 1e-10f);
                    hit = iou >= thr_iou;             // utils/postprocess.py:49
                }
                const unsigned long long word = __ballot(hit);
                if (lane == 0) rows[(p - g0) * nchunks + k] = word;
            }
        }
        __syncthreads();
        if (wave == 0) {
            for (int p = g0; p < gend && p + 1 < N; ++p) {
                const unsigned long long wp = __shfl(myword, p >> 6, 64);
                const bool removed = (wp >> (p & 63)) & 1ull;
                const float cur = removed ? 0.0f : skey[p];
                if (cur <= thr) continue;                // utils/postprocess.py:46-47 (a suppressed box has score 0)
                if (lane >= (g0 >> 6) && lane < nchunks) myword |= rows[(p - g0) * nchunks + lane];
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
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
    int NP = 2;                       // (even: keeps the 8-byte words behind perm[] aligned)
    while (NP < N) NP <<= 1;
    const size_t nchunks = ((size_t)N + 63) / 64;
    const size_t lds = sizeof(float) * 3 * (size_t)N + 16 + sizeof(float) * 4 * (size_t)N + sizeof(int) * ((size_t)NP + 256);      // <= 129 KB at N = 4096
    // the overlap-bitmask rows of a candidate group (8 bytes x (64 x nchunks + 1)) are laid over the sort buffers key / perm / cnt
    // (4 N + 4 NP + 1024 bytes) once the sort is done: the overlay must fit
    if (8 * (64 * nchunks + 1) > 4 * (size_t)N + 4 * (size_t)NP + 1024) {
        yolo2_set_error("nms: bitmask rows (%zu bytes) do not fit the sort buffers they are laid over", 8 * (64 * nchunks + 1));
        return YOLO2_E_ARG;
    }
    if (lds > 64 * 1024) {
        // the attribute is per device: remember what each device of this process has been raised to (a second GPU never inherited the first one's limit)
        static std::atomic<size_t> lds_set[64];      // (atomic: two host threads may call yolo2_nms at once; zero-initialised)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
        if (dev < 0 || lds > lds_set[dev].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute((const void *)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                yolo2_set_error("nms: cannot reserve %zu bytes of LDS", lds);
                return YOLO2_E_LAUNCH;
            }
            if (dev >= 0) lds_set[dev].store(lds, std::memory_order_relaxed);
        }
    }
    nms_kernel<<<B * C, 256, lds, st>>>(conf, (const float *)ws, xy_min, xy_max, order_out, N, C, threshold, threshold_iou, NP);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
