// YOLO (v1) head on gfx950 and the three element-wise pieces only the v1 family needs: decode + Objectives forward and
// backward of reference model/yolo/__init__.py:37-100, leaky-ReLU backward for the un-normalised conv / fc layers
// (model/yolo/inference.py:27-61), slim.layers.dropout, slim.l2_regularizer.  All HBM/latency-bound vector kernels.
//
// Network output (one row per image, model/yolo/__init__.py:41-47): [cells*C class scores | cells*boxes*(iou, x, y, sqrt_w, sqrt_h)],
// all LINEAR (no sigmoid / exp / softmax as in v2): wh01 = base^2, wh01_sqrt = |base|.
// Loss: one lane per (image, cell, box), the boxes of a cell in a power-of-two lane group (responsible box = exact `iou == max`, :82-83,
// as in head.hip: FP contraction off); the class term belongs to the cell (masked by `mask`, not `mask_best`, :100) and is evaluated
// by the cell's first lane.
#include "common.h"
#pragma clang fp contract(off)

template <typename T>
__global__ __launch_bounds__(256) void yolo1_loss_kernel(
    const T *__restrict__ net, int ld, const float *__restrict__ mask, const float *__restrict__ prob, const float *__restrict__ coords,
    const float *__restrict__ off_min, const float *__restrict__ off_max, const float *__restrict__ areas, float w_best, float w_normal,
    float w_coords, float w_prob, T *__restrict__ dnet, float *__restrict__ partial, int B, int cell_h, int cell_w, int NB, int C, int LPC) {
    const int cells = cell_h * cell_w;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long cell_id = gid / LPC;               // flat (b, cell)
    const int a = (int)(gid % LPC);
    const bool act = cell_id < (long)B * cells && a < NB;
    const float cnt = (float)((long)B * cells * NB);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    float iou = -INFINITY, v[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, m = 0.f, tc[4] = {0.f, 0.f, 0.f, 0.f};
    const int b = act ? (int)(cell_id / cells) : 0, cell = act ? (int)(cell_id % cells) : 0;
    const T *row = net + (long)b * ld;
    const long rbase = (long)cells * C + ((long)cell * NB + a) * 5;
    if (act) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = (float)row[rbase + k];
        const float w = (v[3] * v[3]) * (float)cell_w, h = (v[4] * v[4]) * (float)cell_h;       // wh = wh01 * [cell_width, cell_height]
        const float hx = w / 2.0f, hy = h / 2.0f;
        const float pminx = v[1] - hx, pminy = v[2] - hy, pmaxx = v[1] + hx, pmaxy = v[2] + hy;
        m = mask[cell_id];
        const float ix = fmaxf(fminf(pmaxx, off_max[cell_id * 2]) - fmaxf(pminx, off_min[cell_id * 2]), 0.0f);
        const float iy = fmaxf(fminf(pmaxy, off_max[cell_id * 2 + 1]) - fmaxf(pminy, off_min[cell_id * 2 + 1]), 0.0f);
        const float inter = ix * iy;
        iou = inter / fmaxf((areas[cell_id] + w * h) - inter, 1e-10f);
#pragma unroll
        for (int k = 0; k < 4; ++k) tc[k] = coords[cell_id * 4 + k];
    }
    float best = iou;
    for (int o = LPC >> 1; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (act) {
        const float mb = m * (iou == best ? 1.0f : 0.0f);
        const float d_iou = v[0] - mb;
        s0 = mb * (d_iou * d_iou);
        s1 = (1.0f - mb) * (d_iou * d_iou);
        const float aw = fabsf(v[3]), ah = fabsf(v[4]);
        const float dc0 = v[1] - tc[0], dc1 = v[2] - tc[1], dc2 = aw - tc[2], dc3 = ah - tc[3];
        s2 = mb * (dc0 * dc0) + mb * (dc1 * dc1) + mb * (dc2 * dc2) + mb * (dc3 * dc3);
        T *drow = dnet ? dnet + (long)b * ld : nullptr;
        if (drow) {
            const float w_obj = w_best * mb + w_normal * (1.0f - mb);
            drow[rbase + 0] = (T)(2.0f * d_iou * w_obj / cnt);
            drow[rbase + 1] = (T)(2.0f * mb * dc0 * w_coords / cnt);
            drow[rbase + 2] = (T)(2.0f * mb * dc1 * w_coords / cnt);
            // d|x|/dx = sign(x) ([TF-sem] tf.abs gradient; 0 at 0)
            drow[rbase + 3] = (T)(2.0f * mb * dc2 * w_coords / cnt * (v[3] > 0.f ? 1.f : (v[3] < 0.f ? -1.f : 0.f)));
            drow[rbase + 4] = (T)(2.0f * mb * dc3 * w_coords / cnt * (v[4] > 0.f ? 1.f : (v[4] < 0.f ? -1.f : 0.f)));
        }
        if (a == 0) {        // class scores of the cell: masked by the cell's object mask
            const float *tp = prob + cell_id * C;
            float sp = 0.f;
            for (int k = 0; k < C; ++k) {
                const float e = (float)row[(long)cell * C + k] - tp[k];
                sp += e * e;
                if (drow) drow[(long)cell * C + k] = (T)(2.0f * m * e * w_prob / cnt);
            }
            s3 = m * sp;
            if (drow && cell == 0)
                for (int k = cells * (C + NB * 5); k < ld; ++k) drow[k] = (T)0.f;        // padding lanes
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    __shared__ float red[4][4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w][threadIdx.x];
        partial[(long)blockIdx.x * 4 + threadIdx.x] = t;
    }
}

__global__ void yolo1_loss_finalize_kernel(const float *__restrict__ partial, int nblocks, float cnt, float *__restrict__ objectives) {
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int i = lane; i < nblocks; i += 64) acc += (double)partial[(long)i * 4 + k];
    acc = wave_sum_d(acc);
    if (lane == 0) objectives[k] = (float)(acc / (double)cnt);
}

static int y1_lanes_per_cell(int n) {
    int l = 1;
    while (l < n) l <<= 1;
    return l;
}

extern "C" int yolo1_loss(const void *net, int ld, const float *mask, const float *prob, const float *coords, const float *off_min,
                          const float *off_max, const float *areas, const float *hparam, float *objectives, void *dnet, float *ws, int B,
                          int cell_h, int cell_w, int boxes_per_cell, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(net && mask && prob && coords && off_min && off_max && areas && hparam && objectives && ws);
    Y2_CHECK_ARG(B > 0 && cell_h > 0 && cell_w > 0 && boxes_per_cell > 0 && boxes_per_cell <= 64 && C > 0 && ld >= cell_h * cell_w * (C + boxes_per_cell * 5));
    hipStream_t st = (hipStream_t)stream;
    const int LPC = y1_lanes_per_cell(boxes_per_cell);
    const int nblocks = cdiv((long)B * cell_h * cell_w * LPC, 256);
    Y2_DISPATCH_DTYPE(dtype, yolo1_loss_kernel<T><<<nblocks, 256, 0, st>>>((const T *)net, ld, mask, prob, coords, off_min, off_max, areas, hparam[0], hparam[1],
                                                                           hparam[2], hparam[3], (T *)dnet, ws, B, cell_h, cell_w, boxes_per_cell, C, LPC));
    yolo1_loss_finalize_kernel<<<1, 256, 0, st>>>(ws, nblocks, (float)((long)B * cell_h * cell_w * boxes_per_cell), objectives);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// detection block (model/yolo/__init__.py:56-62): conf = iou * prob (the cell's scores for each of its boxes), corners in cell units
template <typename T>
__global__ __launch_bounds__(256) void yolo1_decode_kernel(const T *__restrict__ net, int ld, float *__restrict__ conf, float *__restrict__ xy_min,
                                                           float *__restrict__ xy_max, int *__restrict__ nan_flag, int B, int cell_h, int cell_w, int NB, int C) {
    const int cells = cell_h * cell_w;
    const long total = (long)B * cells * NB;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int a = (int)(gid % NB);
    const long cell_id = gid / NB;
    const int cell = (int)(cell_id % cells), b = (int)(cell_id / cells);
    const T *row = net + (long)b * ld;
    const long rbase = (long)cells * C + ((long)cell * NB + a) * 5;
    const float iou = (float)row[rbase], x = (float)row[rbase + 1], y = (float)row[rbase + 2], bw = (float)row[rbase + 3], bh = (float)row[rbase + 4];
    const float hx = (bw * bw) * (float)cell_w / 2.0f, hy = (bh * bh) * (float)cell_h / 2.0f;
    const float cx = (float)(cell % cell_w), cy = (float)(cell / cell_w);
    const float o[4] = {cx + (x - hx), cy + (y - hy), cx + (x + hx), cy + (y + hy)};
    xy_min[gid * 2] = o[0]; xy_min[gid * 2 + 1] = o[1];
    xy_max[gid * 2] = o[2]; xy_max[gid * 2 + 1] = o[3];
    bool bad = !(isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]));
    for (int k = 0; k < C; ++k) {
        const float v = iou * (float)row[(long)cell * C + k];
        conf[gid * C + k] = v;
        bad |= !isfinite(v);
    }
    if (bad && nan_flag) atomicOr(nan_flag, 1);
}

extern "C" int yolo1_head_decode(const void *net, int ld, float *conf, float *xy_min, float *xy_max, int *nan_flag, int B, int cell_h, int cell_w,
                                 int boxes_per_cell, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(net && conf && xy_min && xy_max);
    Y2_CHECK_ARG(B > 0 && cell_h > 0 && cell_w > 0 && boxes_per_cell > 0 && C > 0 && ld >= cell_h * cell_w * (C + boxes_per_cell * 5));
    const long total = (long)B * cell_h * cell_w * boxes_per_cell;
    Y2_DISPATCH_DTYPE(dtype, yolo1_decode_kernel<T><<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>((const T *)net, ld, conf, xy_min, xy_max, nan_flag, B, cell_h,
                                                                                                       cell_w, boxes_per_cell, C));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- leaky-ReLU backward from the OUTPUT (a = max(z, alpha z) has the sign of z for alpha > 0): dz = a >= 0 ? da : alpha * da
template <typename T>
__global__ void leaky_bwd_kernel(const T *__restrict__ A, const T *__restrict__ dA, T *__restrict__ dZ, long n8, float alpha) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const Vec16<T> a = ld16(A + i * Vec16<T>::N), g = ld16(dA + i * Vec16<T>::N);
        Vec16<T> o;
#pragma unroll
        for (int k = 0; k < Vec16<T>::N; ++k) o.set(k, a.get(k) >= 0.f ? g.get(k) : alpha * g.get(k));
        st16(dZ + i * Vec16<T>::N, o);
    }
}
extern "C" int yolo2_leaky_bwd(const void *A, const void *dA, void *dZ, long n, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(A && dA && dZ && n > 0);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(n % vec == 0);
    const long n8 = n / vec;
    const int grid = (int)(n8 / 256 + 1 < 2048 ? n8 / 256 + 1 : 2048);
    Y2_DISPATCH_DTYPE(dtype, leaky_bwd_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)A, (const T *)dA, (T *)dZ, n8, alpha));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- slim.layers.dropout (training): y = x * keep / keep_prob, keep ~ Bernoulli(keep_prob) from a counter-based hash of (seed, index);
// the byte mask is kept for the backward pass.  seed == 0: the mask is an INPUT (parity tests drive both sides with one mask).
__device__ __forceinline__ unsigned y1_hash(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (unsigned)(x >> 40);      // 24 uniform bits
}
template <typename T>
__global__ void dropout_kernel(const T *__restrict__ X, T *__restrict__ Y, unsigned char *__restrict__ mask, long n, float keep_prob, unsigned long long seed) {
    const float inv = 1.0f / keep_prob;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned char k = seed ? (unsigned char)((float)y1_hash(seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)i) * (1.0f / 16777216.0f) < keep_prob) : mask[i];
        if (seed) mask[i] = k;
        Y[i] = (T)(k ? (float)X[i] * inv : 0.f);
    }
}
extern "C" int yolo2_dropout(const void *X, void *Y, unsigned char *mask, long n, float keep_prob, unsigned long long seed, int dtype, void *stream) {
    Y2_CHECK_ARG(X && Y && mask && n > 0 && keep_prob > 0.f && keep_prob <= 1.f);
    const int grid = (int)(n / 256 + 1 < 2048 ? n / 256 + 1 : 2048);
    Y2_DISPATCH_DTYPE(dtype, dropout_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)X, (T *)Y, mask, n, keep_prob, seed));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_dropout_bwd(const void *dY, const unsigned char *mask, void *dX, long n, float keep_prob, int dtype, void *stream) {
    Y2_CHECK_ARG(dY && mask && dX && n > 0 && keep_prob > 0.f);
    const int grid = (int)(n / 256 + 1 < 2048 ? n / 256 + 1 : 2048);
    // backward = the same masking of the incoming gradient
    Y2_DISPATCH_DTYPE(dtype, dropout_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)dY, (T *)dX, const_cast<unsigned char *>(mask), n, keep_prob, 0ULL));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- slim.l2_regularizer(scale)(w) = scale * sum(w^2) / 2 ([TF-sem] tf.nn.l2_loss): adds scale * w to the gradient and the term to *loss (double)
__global__ void l2_reg_kernel(const float *__restrict__ w, float *__restrict__ g, long n, float scale, double *__restrict__ loss) {
    double acc = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i];
        g[i] += scale * x;
        acc += (double)x * (double)x;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss, acc * 0.5 * (double)scale);
}
extern "C" int yolo2_l2_regularizer(const float *w, float *g, long n, float scale, double *loss, void *stream) {
    Y2_CHECK_ARG(w && g && loss && n > 0);
    const int grid = (int)(n / 256 + 1 < 1024 ? n / 256 + 1 : 1024);
    l2_reg_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(w, g, n, scale, loss);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
