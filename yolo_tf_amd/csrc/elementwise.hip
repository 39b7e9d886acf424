// HBM-bound kernels of the YOLOv2 path (gfx950): filter layout prep, batch-norm statistics /
// apply / backward fused with leaky ReLU, 2x2 max pool, reorg (space-to-depth) and channel
// concat moves, image standardisation, bias gradient, optimizers.
// All activation traffic is 16-byte vectors (8 bf16 / 4 f32) with lanes running along the
// contiguous NHWC channel axis; per-channel parameters stay in registers because every thread
// keeps a fixed channel group while it strides over pixels.
// Map of the file (the order a training step meets them): image standardisation (image_sums / image_apply: per-workgroup f64 partials
// folded by the apply kernel); the consumers that FINISH the batch statistics in their own prologue (bn_leaky_fin_kernel: BN apply +
// leaky, optionally + 2x2 pool, + the raw output at the arg-max for the backward, + the un-pooled activation for a fan-out;
// bn_bwd_apply_fin_kernel: dgamma / dbeta from partial rows + the BN / leaky / pool backward); their two-launch forms (colsum /
// reduce_finalize / bn_finalize, bn_leaky(_pool), bn_bwd_reduce / apply) for shapes whose prologue would be too long, for inference and
// for sync_bn; max pool, reorg, channel moves, bias gradient (one launch for the few rows of a detection head); the optimizers, with
// Adam fused with the MFMA operand re-layout (adam_filter_prep_kernel); the bf16 wire casts of the data-parallel exchange.
#include "common.h"
#include <stdlib.h>
#include <stdarg.h>

// ------------------------------------------------------------------------------------------
// error string (thread local) + misc ABI
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void yolo2_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *yolo2_last_error(void) { return g_err; }

// CRC32C (Castagnoli) of a HOST buffer, slicing-by-8: the checksum of TFRecord / TensorBoard event / TF checkpoint files
// (utils/tfrecord.py, utils/events.py, tf_checkpoint.py); `crc` = value so far (0 to start).  ~1.5 GB/s, against ~1 MB/s in Python.
extern "C" uint32_t yolo2_crc32c(const void *data, size_t n, uint32_t crc) {
    static uint32_t table[8][256];
    static bool ready = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
        return true;
    }();
    (void)ready;
    const unsigned char *p = (const unsigned char *)data;
    crc = ~crc;
    while (n && ((uintptr_t)p & 7)) { crc = table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        __builtin_memcpy(&w, p, 8);
        w ^= crc;
        crc = table[7][w & 0xFF] ^ table[6][(w >> 8) & 0xFF] ^ table[5][(w >> 16) & 0xFF] ^ table[4][(w >> 24) & 0xFF] ^
              table[3][(w >> 32) & 0xFF] ^ table[2][(w >> 40) & 0xFF] ^ table[1][(w >> 48) & 0xFF] ^ table[0][(w >> 56) & 0xFF];
        p += 8;
        n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
extern "C" int yolo2_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------
// filter prep: HWIO f32 -> K-contiguous operand layouts (reference stores conv weights HWIO,
// parse_darknet_yolo2.py:95-97)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void filter_fwd_kernel(const float *__restrict__ Wt, T *__restrict__ F, int Cin, int ldcin, int Cout, int taps) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, n = n0 + tx;
        tile[i][tx] = (c < Cin && n < Cout) ? Wt[((long)tap * Cin + c) * Cout + n] : 0.f;
    }
    __syncthreads();
    const long Kf = (long)taps * ldcin;
    for (int i = ty; i < 32; i += 8) {
        int n = n0 + i, c = c0 + tx;
        if (n < Cout && c < ldcin) F[n * Kf + y2_filter_koff(tap, c, ldcin, taps)] = (T)tile[tx][i];
    }
}
template <typename T>
__global__ void filter_dgrad_kernel(const float *__restrict__ Wt, T *__restrict__ F, int Cin, int Cout, int ldcout, int taps) {
    const long total = (long)Cin * taps * ldcout;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n = (int)(i % ldcout);
        long r = i / ldcout;
        int tp = (int)(r % taps);
        int c = (int)(r / taps);
        float v = n < Cout ? Wt[((long)(taps - 1 - tp) * Cin + c) * Cout + n] : 0.f;
        F[c * (long)taps * ldcout + y2_filter_koff(tp, n, ldcout, taps)] = (T)v;
    }
}

extern "C" int yolo2_filter_prep(const float *W, void *Ffwd, void *Fdgr, int ksize, int Cin, int ldcin,
                                 int Cout, int ldcout, int dtype, void *stream) {
    Y2_CHECK_ARG(W && (Ffwd || Fdgr));
    Y2_CHECK_ARG(ksize == 1 || ksize == 3);
    Y2_CHECK_ARG(ldcin >= Cin && ldcout >= Cout);
    hipStream_t st = (hipStream_t)stream;
    const int taps = ksize * ksize;
    if (Ffwd) {
        dim3 grid(cdiv(Cout, 32), cdiv(ldcin, 32), taps), block(32, 8);
        Y2_DISPATCH_DTYPE(dtype, filter_fwd_kernel<T><<<grid, block, 0, st>>>(W, (T *)Ffwd, Cin, ldcin, Cout, taps));
    }
    if (Fdgr) {
        long total = (long)Cin * taps * ldcout;
        int grid = (int)(total / 256 + 1 < 4096 ? total / 256 + 1 : 4096);
        Y2_DISPATCH_DTYPE(dtype, filter_dgrad_kernel<T><<<grid, 256, 0, st>>>(W, (T *)Fdgr, Cin, Cout, ldcout, taps));
    }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// All layers in ONE launch (the per-layer calls were 43 launches of ~9 us each per training step): a device
// table of descriptors; work unit u of a layer = one 64 x 64 (c, n) tile of one tap, handling BOTH layouts
// from the same LDS tile (Ffwd needs the transpose, Fdgr is a re-strided copy).  16-byte loads and stores
// (a 32 x 32 tile with 2-byte stores ran at 2.5 TB/s: 215 us per training step).
template <typename T>
__device__ __forceinline__ void store8(T *dst, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)v[j];
        *reinterpret_cast<bf16x8 *>(dst) = o;
    } else {
        f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        reinterpret_cast<f32x4 *>(dst)[0] = a;
        reinterpret_cast<f32x4 *>(dst)[1] = b;
    }
}
// Both operand layouts of one TC x TN (c, n) tile of one tap, from its fresh f32 values in LDS: Ffwd rows n with c contiguous (the transpose),
// Fdgr rows c with n contiguous and the taps flipped.  8 elements (16 bytes of bf16) per lane and store.
template <typename T>
__device__ __forceinline__ void filter_tile_emit(const yolo2_filter_desc &d, const float (&tile)[YOLO2_FILTER_PREP_TILE][YOLO2_FILTER_PREP_TILE_N + 1], int tap, int taps,
                                                 int c0, int n0, int tid) {
    constexpr int TC = YOLO2_FILTER_PREP_TILE, TN = YOLO2_FILTER_PREP_TILE_N;
    T *Ff = (T *)d.Ffwd, *Fd = (T *)d.Fdgr;
    if (Ff) {       // rows n, c contiguous: TC / 8 lanes per row (the tile's odd pitch keeps the transposed reads conflict-free)
        constexpr int LPR = TC / 8, RPP = 256 / LPR;
        const int g8 = (tid % LPR) * 8, rr = tid / LPR;
        const long Kf = (long)taps * d.ldcin;
#pragma unroll
        for (int p = 0; p < TN / RPP; ++p) {
            const int nl = rr + p * RPP, nn = n0 + nl, c = c0 + g8;
            if (nn < d.cout && c < d.ldcin) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tile[g8 + j][nl];
                store8<T>(Ff + nn * Kf + y2_filter_koff(tap, c, d.ldcin, taps), v);
            }
        }
    }
    if (Fd) {       // rows c, n contiguous, taps flipped: TN / 8 lanes per row
        constexpr int LPR = TN / 8, RPP = 256 / LPR;
        const int g8 = (tid % LPR) * 8, rr = tid / LPR;
        const long Kd = (long)taps * d.ldcout;
        const int tp = taps - 1 - tap;
#pragma unroll
        for (int p = 0; p < TC / RPP; ++p) {
            const int cl = rr + p * RPP, c = c0 + cl, nn = n0 + g8;
            if (c < d.cin && nn < d.ldcout) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tile[cl][g8 + j];
                store8<T>(Fd + c * Kd + y2_filter_koff(tp, nn, d.ldcout, taps), v);
            }
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void filter_prep_batch_kernel(const yolo2_filter_desc *__restrict__ descs, int n) {
    constexpr int TC = YOLO2_FILTER_PREP_TILE, TN = YOLO2_FILTER_PREP_TILE_N;
    __shared__ float tile[TC][TN + 1];
    int li = 0;
    while (li + 1 < n && (int)blockIdx.x >= descs[li + 1].first_block) ++li;
    const yolo2_filter_desc d = descs[li];
    const int taps = d.ksize * d.ksize;
    const int ctiles = (d.ldcin + TC - 1) / TC, ntiles = (d.ldcout + TN - 1) / TN;
    int u = blockIdx.x - d.first_block;
    const int ntile = u % ntiles; u /= ntiles;
    const int ctile = u % ctiles;
    const int tap = u / ctiles;
    const int n0 = ntile * TN, c0 = ctile * TC;
    const int tid = threadIdx.x;
    const float *Wt = d.W + (long)tap * d.cin * d.cout;
    const bool vec_ok = (d.cout & 3) == 0 && (((uintptr_t)d.W) & 15) == 0;
    {   // load: TN / 4 lanes x float4 per row
        constexpr int LPR = TN / 4, RPP = 256 / LPR;
        const int col = (tid % LPR) * 4, r0 = tid / LPR;
#pragma unroll
        for (int p = 0; p < TC / RPP; ++p) {
            const int c = c0 + r0 + p * RPP, nn = n0 + col;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < d.cin) {
                if (vec_ok && nn + 3 < d.cout) {
                    const f32x4 t = *reinterpret_cast<const f32x4 *>(Wt + (long)c * d.cout + nn);
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nn + j < d.cout) v[j] = Wt[(long)c * d.cout + nn + j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r0 + p * RPP][col + j] = v[j];
        }
    }
    __syncthreads();
    filter_tile_emit<T>(d, tile, tap, taps, c0, n0, tid);
}

extern "C" int yolo2_filter_prep_blocks(int ksize, int ldcin, int ldcout) {
    if (ksize < 1 || ldcin < 1 || ldcout < 1) return 0;
    return ksize * ksize * cdiv(ldcin, YOLO2_FILTER_PREP_TILE) * cdiv(ldcout, YOLO2_FILTER_PREP_TILE_N);
}
extern "C" int yolo2_filter_prep_batch(const yolo2_filter_desc *descs_device, int n, int total_blocks, int dtype, void *stream) {
    Y2_CHECK_ARG(descs_device && n > 0 && total_blocks > 0);
    Y2_DISPATCH_DTYPE(dtype, filter_prep_batch_kernel<T><<<total_blocks, 256, 0, (hipStream_t)stream>>>(descs_device, n));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// column reductions over [M][C] (pixel stride ld): sum, centred sum of squares, BN backward sums
// ------------------------------------------------------------------------------------------
struct RowMap {  // fixed channel group per thread, rows strided
    int tpr, rpp, cg, rs;
    bool active;
    __device__ RowMap(int C, int vec) {
        tpr = C / vec;
        rpp = 256 / tpr;
        if (rpp < 1) rpp = 1;
        cg = threadIdx.x % tpr;
        rs = threadIdx.x / tpr;
        active = rs < rpp && threadIdx.x < tpr * rpp;
    }
};

// Column reductions are two-stage and atomic-free: every block reduces its row slots through LDS
// and stores one f32 partial per (quantity, channel) at part[(k*nb + block)*C + c]; a second kernel
// sums the nb partials per channel in f64 and applies the finalisation (mean/var, dgamma/dbeta,
// bias gradient).  (A first version used f64 atomics on 2*C addresses: 2048 blocks contending on
// 64 addresses made the reductions 40 % of the training step.)
// (round 6) With 4 .. 32 threads per row -- 32 .. 256 channels in bf16 -- the row slots of a wave meet on the VALU first (common.h y2_lane_group_sum: the
// lanes that share lane % tpr) and only the four waves' sums go through LDS.  The general path below leaves the whole sum to `tpr` threads, rpp serial
// LDS reads per value: with 32 channels that is 4 threads x 1024 dependent reads, ~15 us at the end of every launch (measured: conv0's BN-backward
// reduction took 33 us for 88 MB, its 128-channel sibling 19 us for 22 MB).
// One 8 KB scratch for every form of the block reduction below (a __shared__ array inside a function template is allocated once per instantiation:
// the four lane-group forms + the general path had grown the BN-backward reduction's workgroups to 48 KB of LDS)
__device__ __forceinline__ float *colsum_scratch() {
    __shared__ float buf[2048];
    return buf;
}
template <int N, int K, int G>
__device__ __forceinline__ void block_colsum_store_g(const float (&part)[K][N], int C, float *out, int nb) {
    float *const redw = colsum_scratch();      // [quantity][wave][channel group][value]: K x 4 x 32 x N floats <= 8 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float v = y2_lane_group_sum<G>(part[k][j]);
            if (lane < G) redw[((k * 4 + wave) * 32 + lane) * N + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < G) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < N; ++j)
                out[((long)k * nb + blockIdx.x) * C + threadIdx.x * N + j] =
                    (redw[((k * 4 + 0) * 32 + threadIdx.x) * N + j] + redw[((k * 4 + 1) * 32 + threadIdx.x) * N + j]) +
                    (redw[((k * 4 + 2) * 32 + threadIdx.x) * N + j] + redw[((k * 4 + 3) * 32 + threadIdx.x) * N + j]);
    }
}
template <int N, int K>
__device__ __forceinline__ void block_colsum_store(const float (&part)[K][N], const RowMap &rm, int C, float *out, int nb) {
    // (tpr * rpp == 256 for these: every thread is active)
    if (rm.tpr == 4) return block_colsum_store_g<N, K, 4>(part, C, out, nb);
    if (rm.tpr == 8) return block_colsum_store_g<N, K, 8>(part, C, out, nb);
    if (rm.tpr == 16) return block_colsum_store_g<N, K, 16>(part, C, out, nb);
    if (rm.tpr == 32) return block_colsum_store_g<N, K, 32>(part, C, out, nb);
    // general path (>= 64 threads per row: at most four row slots; odd channel counts): one quantity at a time through the same scratch
    float *const red = colsum_scratch();         // [256][N]
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k) __syncthreads();
#pragma unroll
        for (int j = 0; j < N; ++j) red[threadIdx.x * N + j] = rm.active ? part[k][j] : 0.f;
        __syncthreads();
        if (threadIdx.x < rm.tpr) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float s = 0.f;
                for (int r = 0; r < rm.rpp; ++r) s += red[(r * rm.tpr + threadIdx.x) * N + j];
                out[((long)k * nb + blockIdx.x) * C + threadIdx.x * N + j] = s;
            }
        }
    }
}

// K = 2: single-pass moments, SHIFTED by the channel's first sample s = X[0][c]: the partials are
// sum (x-s) and sum (x-s)^2, so the f64 finalisation's E[d^2] - E[d]^2 cancels against (mean-s)^2 ~ var
// instead of mean^2 (plain E[x^2]-mean^2 loses the variance when |mean| >> std, e.g. few samples per
// channel).  Block 0 also stores s as a third row of the partial buffer.   K = 1: plain column sum.
template <typename T, int K>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ X, int ld, long M, int C, float *__restrict__ part,
                                                     const float *__restrict__ shift_in = nullptr, int nb_rows = 0) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    float acc[K][N], sh[N];
#pragma unroll
    for (int j = 0; j < N; ++j) sh[j] = 0.f;
    if (K == 2 && rm.active && shift_in) {
#pragma unroll
        for (int j = 0; j < N; ++j) sh[j] = shift_in[rm.cg * N + j];
    } else if (K == 2 && rm.active) {
        Vec16<T> v0 = ld16(X + rm.cg * N);
#pragma unroll
        for (int j = 0; j < N; ++j) sh[j] = v0.get(j);
        if (blockIdx.x == 0 && rm.rs == 0)
#pragma unroll
            for (int j = 0; j < N; ++j) part[(long)2 * gridDim.x * C + rm.cg * N + j] = sh[j];
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < N; ++j) acc[k][j] = 0.f;
    if (rm.active) {
        const long step = (long)gridDim.x * rm.rpp;
        long r = (long)blockIdx.x * rm.rpp + rm.rs;
        for (; r + 3 * step < M; r += 4 * step) {          // 4 independent 16-byte loads in flight per lane
            Vec16<T> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld16(X + (r + u * step) * ld + rm.cg * N);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float x = v[u].get(j) - sh[j];
                    acc[0][j] += x;
                    if (K == 2) acc[K - 1][j] += x * x;
                }
        }
        for (; r < M; r += step) {
            Vec16<T> v = ld16(X + r * ld + rm.cg * N);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float x = v.get(j) - sh[j];
                acc[0][j] += x;
                if (K == 2) acc[K - 1][j] += x * x;
            }
        }
    }
    block_colsum_store<N, K>(acc, rm, C, part, nb_rows ? nb_rows : gridDim.x);
}

// FIN 0: (sum x, sum x^2) -> mean, biased var;  FIN 1: two sums -> two f32 outputs;  FIN 2: one sum -> o0
template <int FIN>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float *__restrict__ part, int nb, int C, long M,
                                                              float *__restrict__ o0, float *__restrict__ o1, int nout,
                                                              float *__restrict__ mm = nullptr, float *__restrict__ mv = nullptr,
                                                              float omd = 0.f) {
    // 16 columns x 16 row groups per block: every thread sums nb/16 partials with 4 independent f64 chains
    // (the first version walked up to 1024 partials serially per thread: 77 us of pure latency per call)
    constexpr int K = FIN == 2 ? 1 : 2;
    __shared__ double red[K][16][17];
    const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (c < C) {
            const float *p = part + (long)k * nb * C + c;
            int b = rg;
            for (; b + 48 < nb; b += 64) {
                s0 += (double)p[(long)b * C];
                s1 += (double)p[(long)(b + 16) * C];
                s2 += (double)p[(long)(b + 32) * C];
                s3 += (double)p[(long)(b + 48) * C];
            }
            for (; b < nb; b += 16) s0 += (double)p[(long)b * C];
        }
        red[k][rg][col] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (rg == 0 && c < nout) {
        double t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[k][r][col];
            t[k] = a;
        }
        if (FIN == 0) {
            double dm = t[0] / (double)M;                       // mean of (x - shift)
            double var = t[K - 1] / (double)M - dm * dm;
            double mean = (double)part[(long)2 * nb * C + c] + dm;
            const float fm = (float)mean, fv = (float)(var > 0.0 ? var : 0.0);
            o0[c] = fm;
            o1[c] = fv;
            if (mm) {   // fused assign_moving_average (UPDATE_OPS): moving -= (1-decay)*(moving-batch)
                mm[c] = mm[c] - (mm[c] - fm) * omd;
                mv[c] = mv[c] - (mv[c] - fv) * omd;
            }
        } else if (FIN == 1) {
            o0[c] = (float)t[0];
            o1[c] = (float)t[K - 1];
        } else {
            o0[c] = (float)t[0];
        }
    }
}

static int colsum_grid(long M, int C, int vec) {
    int tpr = C / vec, rpp = 256 / tpr;
    if (rpp < 1) rpp = 1;
    long g = (M + (long)rpp * 4 - 1) / ((long)rpp * 4);
    const int cap = 256;
    if (g > cap) g = cap;       // default 1 workgroup per CU (measured best: 64..1024 swept); keeps the finalisation short (workspace contract: <= 1024)
    if (g < 1) g = 1;
    return (int)g;
}

// Fallback producer of the convolution epilogue's partial format ([2][Y2_BN_PART_ROWS][C], zero on entry): used for the
// layers whose convolution cannot produce the sums itself (first-layer direct kernel, K-sliced grids).
int y2_colsum_into(const void *Y, int ld, long M, int C, const float *shift, float *part, int dtype, hipStream_t st, int *rows_used) {
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256 && ld == C);
    int nb = colsum_grid(M, C, vec);
    if (nb > Y2_BN_PART_ROWS) nb = Y2_BN_PART_ROWS;
    {   // no more rows than a consumer that finalises them in its prologue reads (fin_shape_ok below: rows x slice x 8 bytes <= 128 KB)
        const int tpr = C / vec, lpr = tpr < 16 ? tpr : 16;
        const long lim = (128L << 10) / ((long)lpr * vec * 8);
        if (nb > lim) nb = (int)lim;
    }
    if (rows_used) *rows_used = nb;
    Y2_DISPATCH_DTYPE(dtype, colsum_kernel<T, 2><<<nb, 256, 0, st>>>((const T *)Y, ld, M, C, part, shift, Y2_BN_PART_ROWS));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// partial rows -> batch mean / biased variance (+ moving-average update), and the rows are zeroed again for the next step
template <int FIN>      // 0: batch moments (+ moving averages); 1: plain column sums (plane 0 -> mean_out, plane 1 -> var_out)
__global__ __launch_bounds__(256) void bn_finalize_kernel(float *__restrict__ part, const float *__restrict__ shift, int C, long M,
                                                          float *__restrict__ mean_out, float *__restrict__ var_out,
                                                          float *__restrict__ mm, float *__restrict__ mv, float omd) {
    constexpr int NB = Y2_BN_PART_ROWS;
    __shared__ double red[2][16][17];
    const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (c < C) {
            float *p = part + (long)k * NB * C + c;
            float v[NB / 16];
#pragma unroll
            for (int u = 0; u < NB / 16; ++u) v[u] = p[(long)(rg + 16 * u) * C];
#pragma unroll
            for (int u = 0; u < NB / 16; ++u) p[(long)(rg + 16 * u) * C] = 0.f;
#pragma unroll
            for (int u = 0; u < NB / 16; u += 4) {
                s0 += (double)v[u];
                s1 += (double)v[u + 1];
                s2 += (double)v[u + 2];
                s3 += (double)v[u + 3];
            }
        }
        red[k][rg][col] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (rg == 0 && c < C) {
        double t[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += red[k][r][col];
            t[k] = a;
        }
        if (FIN == 1) {
            mean_out[c] = (float)t[0];
            var_out[c] = (float)t[1];
            return;
        }
        const double dm = t[0] / (double)M;
        const double var = t[1] / (double)M - dm * dm;
        const double mean = (double)shift[c] + dm;
        const float fm = (float)mean, fv = (float)(var > 0.0 ? var : 0.0);
        mean_out[c] = fm;
        var_out[c] = fv;
        if (mm) {
            mm[c] = mm[c] - (mm[c] - fm) * omd;
            mv[c] = mv[c] - (mv[c] - fv) * omd;
        }
    }
}
extern "C" int yolo2_bn_finalize(float *bn_part, const float *shift, long M, int C, float *mean, float *var, float *moving_mean,
                                 float *moving_var, double decay, void *stream) {
    Y2_CHECK_ARG(bn_part && shift && mean && var && M > 0 && C > 0);
    Y2_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr));
    bn_finalize_kernel<0><<<cdiv(C, 16), 256, 0, (hipStream_t)stream>>>(bn_part, shift, C, M, mean, var, moving_mean, moving_var, (float)(1.0 - decay));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

int y2_bn_part_to_grads(float *part, int C, float *dgamma, float *dbeta, hipStream_t st) {
    bn_finalize_kernel<1><<<cdiv(C, 16), 256, 0, st>>>(part, nullptr, C, 1, dgamma, dbeta, nullptr, nullptr, 0.f);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

static int bn_stats_impl(const void *Y, float *mean, float *var, float *mm, float *mv, double decay, double *ws, long M, int C, int dtype, void *stream);
extern "C" int yolo2_bn_stats(const void *Y, float *mean, float *var, double *ws, long M, int C, int dtype, void *stream) {
    return bn_stats_impl(Y, mean, var, nullptr, nullptr, 0.0, ws, M, C, dtype, stream);
}
extern "C" int yolo2_bn_stats_ema(const void *Y, float *mean, float *var, float *moving_mean, float *moving_var, double decay,
                                  double *ws, long M, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(moving_mean && moving_var);
    return bn_stats_impl(Y, mean, var, moving_mean, moving_var, decay, ws, M, C, dtype, stream);
}
static int bn_stats_impl(const void *Y, float *mean, float *var, float *mm, float *mv, double decay, double *ws, long M, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(Y && mean && var && ws && M > 0 && C > 0);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256);
    hipStream_t st = (hipStream_t)stream;
    const int nb = colsum_grid(M, C, vec);
    float *part = (float *)ws;
    Y2_DISPATCH_DTYPE(dtype, colsum_kernel<T, 2><<<nb, 256, 0, st>>>((const T *)Y, C, M, C, part));
    reduce_finalize_kernel<0><<<cdiv(C, 16), 256, 0, st>>>(part, nb, C, M, mean, var, C, mm, mv, (float)(1.0 - decay));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// wide rows (the fully connected layers of the YOLO v1 family: M = batch, thousands of columns): one lane per column, rows in order
template <typename T>
__global__ void colsum_wide_kernel(const T *__restrict__ dY, int ld, long M, int C, float *__restrict__ dbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double acc = 0.0;
    for (long r = 0; r < M; ++r) acc += (double)(float)dY[r * ld + c];
    dbias[c] = (float)acc;
}

// few rows (the detection head: 13x13 cells x batch): ONE launch, a workgroup per 16-byte channel group; every thread sums every 256th
// row in f32 (a handful of rows), the 256 partials meet in f64 through LDS.  The two-stage form costs a second 5 us launch here.
template <typename T>
__global__ __launch_bounds__(256) void colsum_direct_kernel(const T *__restrict__ dY, int ld, long M, int C, float *__restrict__ dbias) {
    constexpr int N = Vec16<T>::N;
    const T *col = dY + (long)blockIdx.x * N;
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    long r = threadIdx.x;
    for (; r + 768 < M; r += 1024) {       // four independent 16-byte loads in flight
        Vec16<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld16(col + (r + 256 * u) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] += v[u].get(j);
    }
    for (; r < M; r += 256) {
        const Vec16<T> v = ld16(col + r * ld);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += v.get(j);
    }
    __shared__ double red[256][N + 1];
#pragma unroll
    for (int j = 0; j < N; ++j) red[threadIdx.x][j] = (double)acc[j];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int j = 0; j < N; ++j) red[threadIdx.x][j] += red[threadIdx.x + s][j];
        __syncthreads();
    }
    const int c = blockIdx.x * N + threadIdx.x;
    if ((int)threadIdx.x < N && c < C) dbias[c] = (float)red[0][threadIdx.x];
}

extern "C" int yolo2_bias_grad(const void *dY, int ld, float *dbias, double *ws, long M, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(dY && dbias && ws && M > 0 && C > 0 && ld >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    hipStream_t st = (hipStream_t)stream;
    if (ld / vec > 256) {
        Y2_CHECK_ARG(M <= 4096);
        Y2_DISPATCH_DTYPE(dtype, colsum_wide_kernel<T><<<cdiv(C, 256), 256, 0, st>>>((const T *)dY, ld, M, C, dbias));
        Y2_CHECK_LAUNCH();
        return YOLO2_OK;
    }
    Y2_CHECK_ARG(ld % vec == 0);
    const long direct_rows = 8192;
    if (M <= direct_rows && ((uintptr_t)dY & 15) == 0) {
        Y2_DISPATCH_DTYPE(dtype, colsum_direct_kernel<T><<<cdiv(C, vec), 256, 0, st>>>((const T *)dY, ld, M, C, dbias));
        Y2_CHECK_LAUNCH();
        return YOLO2_OK;
    }
    // reduce over the padded width ld (padding lanes are zero by contract), report the first C
    const int nb = colsum_grid(M, ld, vec);
    float *part = (float *)ws;
    Y2_DISPATCH_DTYPE(dtype, colsum_kernel<T, 1><<<nb, 256, 0, st>>>((const T *)dY, ld, M, ld, part));
    reduce_finalize_kernel<2><<<cdiv(ld, 16), 256, 0, st>>>(part, nb, ld, M, dbias, nullptr, C);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

__global__ void bn_ema_kernel(float *mm, float *mv, const float *mean, const float *var, int C, float one_minus_decay) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        mm[c] = mm[c] - (mm[c] - mean[c]) * one_minus_decay;
        mv[c] = mv[c] - (mv[c] - var[c]) * one_minus_decay;
    }
}
extern "C" int yolo2_bn_ema(float *moving_mean, float *moving_var, const float *mean, const float *var, int C, double decay, void *stream) {
    Y2_CHECK_ARG(moving_mean && moving_var && mean && var && C > 0);
    bn_ema_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(moving_mean, moving_var, mean, var, C, (float)(1.0 - decay));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// BN apply + leaky (forward), BN+leaky backward
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_leaky_kernel(const T *__restrict__ Y, const float *__restrict__ mean, const float *__restrict__ var,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, T *__restrict__ A,
                                                       long M, int C, int lda, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    if (!rm.active) return;
    float mu[N], sc[N], bt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int c = rm.cg * N + j;
        mu[j] = mean[c];
        sc[j] = (1.0f / sqrtf(var[c] + eps)) * gamma[c];
        bt[j] = beta[c];
    }
    for (long r = (long)blockIdx.x * rm.rpp + rm.rs; r < M; r += (long)gridDim.x * rm.rpp) {
        Vec16<T> v = ld16(Y + r * C + rm.cg * N), o;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float z = (v.get(j) - mu[j]) * sc[j] + bt[j];
            o.set(j, fmaxf(z, alpha * z));
        }
        st16(A + r * lda + rm.cg * N, o);
    }
}

static int rowmap_grid(long M, int C, int vec, int rows_per_thread) {
    int tpr = C / vec, rpp = 256 / tpr;
    if (rpp < 1) rpp = 1;
    long g = (M + (long)rpp * rows_per_thread - 1) / ((long)rpp * rows_per_thread);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int yolo2_bn_leaky(const void *Y, const float *mean, const float *var, const float *gamma, const float *beta,
                              void *A, long M, int C, int lda, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(Y && mean && var && gamma && beta && A && M > 0 && C > 0 && lda >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256 && lda % vec == 0);
    int grid = rowmap_grid(M, C, vec, 4);
    Y2_DISPATCH_DTYPE(dtype, bn_leaky_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)Y, mean, var, gamma, beta, (T *)A, M, C, lda, eps, alpha));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T *__restrict__ dA, int ldda, const T *__restrict__ Y, const float *__restrict__ mean,
                                                            const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float *__restrict__ ws, long M, int C, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    float part[2][N];
    float mu[N], inv[N], ga[N], bt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        part[0][j] = part[1][j] = 0.f;
        int c = rm.cg * N + j;
        bool ok = rm.active;
        mu[j] = ok ? mean[c] : 0.f;
        inv[j] = ok ? 1.0f / sqrtf(var[c] + eps) : 0.f;
        ga[j] = ok ? gamma[c] : 0.f;
        bt[j] = ok ? beta[c] : 0.f;
    }
    if (rm.active) {
        const long step = (long)gridDim.x * rm.rpp;
        long r = (long)blockIdx.x * rm.rpp + rm.rs;
        auto accum = [&](const Vec16<T> &y, const Vec16<T> &d) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float xh = (y.get(j) - mu[j]) * inv[j];
                float z = (y.get(j) - mu[j]) * (inv[j] * ga[j]) + bt[j];
                float g = z >= 0.f ? d.get(j) : alpha * d.get(j);
                part[0][j] += g * xh;  // dgamma
                part[1][j] += g;       // dbeta
            }
        };
        for (; r + 3 * step < M; r += 4 * step) {          // 8 independent 16-byte loads in flight per lane
            Vec16<T> y[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                y[u] = ld16(Y + (r + u * step) * C + rm.cg * N);
                d[u] = ld16(dA + (r + u * step) * ldda + rm.cg * N);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(y[u], d[u]);
        }
        for (; r < M; r += step) {
            Vec16<T> y = ld16(Y + r * C + rm.cg * N), d = ld16(dA + r * ldda + rm.cg * N);
            accum(y, d);
        }
    }
    block_colsum_store<N, 2>(part, rm, C, ws, gridDim.x);
}
extern "C" int yolo2_bn_leaky_bwd_reduce(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma,
                                         const float *beta, float *dgamma, float *dbeta, double *ws, long M, int C, float eps, float alpha,
                                         int dtype, void *stream) {
    Y2_CHECK_ARG(dA && Y && mean && var && gamma && beta && dgamma && dbeta && ws && M > 0 && C > 0 && ldda >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256 && ldda % vec == 0);
    hipStream_t st = (hipStream_t)stream;
    const int nb = colsum_grid(M, C, vec);
    float *part = (float *)ws;
    Y2_DISPATCH_DTYPE(dtype, bn_bwd_reduce_kernel<T><<<nb, 256, 0, st>>>((const T *)dA, ldda, (const T *)Y, mean, var, gamma, beta, part, M, C, eps, alpha));
    reduce_finalize_kernel<1><<<cdiv(C, 16), 256, 0, st>>>(part, nb, C, M, dgamma, dbeta, C);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T *__restrict__ dA, int ldda, const T *__restrict__ Y, const float *__restrict__ mean,
                                                           const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ dgamma, const float *__restrict__ dbeta, T *__restrict__ dY,
                                                           long M, int C, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    if (!rm.active) return;
    const float invM = 1.0f / (float)M;
    float mu[N], inv[N], ga[N], bt[N], dgm[N], dbm[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int c = rm.cg * N + j;
        mu[j] = mean[c];
        inv[j] = 1.0f / sqrtf(var[c] + eps);
        ga[j] = gamma[c];
        bt[j] = beta[c];
        dgm[j] = dgamma[c] * invM;
        dbm[j] = dbeta[c] * invM;
    }
    for (long r = (long)blockIdx.x * rm.rpp + rm.rs; r < M; r += (long)gridDim.x * rm.rpp) {
        Vec16<T> y = ld16(Y + r * C + rm.cg * N), d = ld16(dA + r * ldda + rm.cg * N), o;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float xh = (y.get(j) - mu[j]) * inv[j];
            float z = (y.get(j) - mu[j]) * (inv[j] * ga[j]) + bt[j];
            float g = z >= 0.f ? d.get(j) : alpha * d.get(j);
            o.set(j, (ga[j] * inv[j]) * (g - dbm[j] - xh * dgm[j]));
        }
        st16(dY + r * C + rm.cg * N, o);
    }
}

extern "C" int yolo2_bn_leaky_bwd_apply(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma,
                                        const float *beta, const float *dgamma, const float *dbeta, void *dY, long M, int C, float eps,
                                        float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(dA && Y && mean && var && gamma && beta && dgamma && dbeta && dY && M > 0 && C > 0 && ldda >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256 && ldda % vec == 0);
    int grid = rowmap_grid(M, C, vec, 4);
    Y2_DISPATCH_DTYPE(dtype, bn_bwd_apply_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)dA, ldda, (const T *)Y, mean, var, gamma, beta, dgamma, dbeta, (T *)dY, M, C, eps, alpha));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// BN + leaky + 2x2/2 max pool in one pass, and its backward (layers whose only consumer is the pool: the
// full-resolution activation and its gradient are never materialised).
//   forward : P = maxpool(leaky(bn(Y))) on the values ROUNDED to T (= what the unfused pair stores and pools),
//             idx = position 0..3 (scan order: (0,0),(0,1),(1,0),(1,1)) of the first maximum, one byte per element
//   backward: dA = dP routed to idx (tf.nn.max_pool gradient, first-max as yolo2_maxpool_bwd), then the BN + leaky
//             backward of yolo2_bn_leaky_bwd_reduce/apply.  Only the arg-max position contributes to dgamma / dbeta.
// Traffic per layer in units of the conv output: forward 1.375 instead of 3.25, backward 3.75 instead of 7.25.
// ------------------------------------------------------------------------------------------
template <int N> struct IdxPack;
template <> struct IdxPack<8> { typedef unsigned long long type; };
template <> struct IdxPack<4> { typedef unsigned int type; };

struct PoolRow {   // pooled pixel r -> element offset of the window's top-left input pixel
    int OH, OW, H, W, C;
    __device__ PoolRow(int H_, int W_, int C_) : OH(H_ / 2), OW(W_ / 2), H(H_), W(W_), C(C_) {}
    __device__ long base(long r) const {
        const int ow = (int)(r % OW);
        const long t = r / OW;
        const int oh = (int)(t % OH);
        const long b = t / OH;
        return ((b * H + oh * 2) * W + ow * 2) * C;
    }
};

template <typename T>
__global__ __launch_bounds__(256) void bn_leaky_pool_kernel(const T *__restrict__ Y, const float *__restrict__ mean, const float *__restrict__ var,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, T *__restrict__ P,
                                                            unsigned char *__restrict__ idx, int B, int H, int W, int C, int ldp, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    if (!rm.active) return;
    float mu[N], sc[N], bt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int c = rm.cg * N + j;
        mu[j] = mean[c];
        sc[j] = (1.0f / sqrtf(var[c] + eps)) * gamma[c];
        bt[j] = beta[c];
    }
    const PoolRow pr(H, W, C);
    const long MP = (long)B * pr.OH * pr.OW;
    for (long r = (long)blockIdx.x * rm.rpp + rm.rs; r < MP; r += (long)gridDim.x * rm.rpp) {
        const T *src = Y + pr.base(r) + rm.cg * N;
        Vec16<T> v[4], o;
        v[0] = ld16(src);
        v[1] = ld16(src + C);
        v[2] = ld16(src + (long)W * C);
        v[3] = ld16(src + (long)W * C + C);
        typename IdxPack<N>::type pack = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = (v[k].get(j) - mu[j]) * sc[j] + bt[j];
                a[k] = (float)(T)fmaxf(z, alpha * z);
            }
            const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
            const int arg = a[0] == m ? 0 : a[1] == m ? 1 : a[2] == m ? 2 : 3;
            o.set(j, m);
            pack |= (typename IdxPack<N>::type)arg << (8 * j);
        }
        st16(P + r * ldp + rm.cg * N, o);
        if (idx) *reinterpret_cast<typename IdxPack<N>::type *>(idx + r * C + rm.cg * N) = pack;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_reduce_kernel(const T *__restrict__ dP, int lddp, const unsigned char *__restrict__ idx, const T *__restrict__ Y,
                                                                 const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float *__restrict__ ws, int B, int H, int W, int C, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    float part[2][N];
    float mu[N], inv[N], ga[N], bt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        part[0][j] = part[1][j] = 0.f;
        int c = rm.cg * N + j;
        bool ok = rm.active;
        mu[j] = ok ? mean[c] : 0.f;
        inv[j] = ok ? 1.0f / sqrtf(var[c] + eps) : 0.f;
        ga[j] = ok ? gamma[c] : 0.f;
        bt[j] = ok ? beta[c] : 0.f;
    }
    if (rm.active) {
        const PoolRow pr(H, W, C);
        const long MP = (long)B * pr.OH * pr.OW;
        for (long r = (long)blockIdx.x * rm.rpp + rm.rs; r < MP; r += (long)gridDim.x * rm.rpp) {
            const T *src = Y + pr.base(r) + rm.cg * N;
            Vec16<T> v[4];
            v[0] = ld16(src);
            v[1] = ld16(src + C);
            v[2] = ld16(src + (long)W * C);
            v[3] = ld16(src + (long)W * C + C);
            const Vec16<T> d = ld16(dP + r * lddp + rm.cg * N);
            const typename IdxPack<N>::type pack = *reinterpret_cast<const typename IdxPack<N>::type *>(idx + r * C + rm.cg * N);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int k = (int)((pack >> (8 * j)) & 3);
                const float y = k == 0 ? v[0].get(j) : k == 1 ? v[1].get(j) : k == 2 ? v[2].get(j) : v[3].get(j);
                const float xh = (y - mu[j]) * inv[j];
                const float z = (y - mu[j]) * (inv[j] * ga[j]) + bt[j];
                const float g = z >= 0.f ? d.get(j) : alpha * d.get(j);
                part[0][j] += g * xh;
                part[1][j] += g;
            }
        }
    }
    block_colsum_store<N, 2>(part, rm, C, ws, gridDim.x);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(const T *__restrict__ dP, int lddp, const unsigned char *__restrict__ idx, const T *__restrict__ Y,
                                                                const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                                T *__restrict__ dY, int B, int H, int W, int C, float eps, float alpha) {
    constexpr int N = Vec16<T>::N;
    RowMap rm(C, N);
    if (!rm.active) return;
    const float invM = 1.0f / (float)((long)B * H * W);
    float mu[N], inv[N], ga[N], bt[N], dgm[N], dbm[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int c = rm.cg * N + j;
        mu[j] = mean[c];
        inv[j] = 1.0f / sqrtf(var[c] + eps);
        ga[j] = gamma[c];
        bt[j] = beta[c];
        dgm[j] = dgamma[c] * invM;
        dbm[j] = dbeta[c] * invM;
    }
    const PoolRow pr(H, W, C);
    const long MP = (long)B * pr.OH * pr.OW;
    for (long r = (long)blockIdx.x * rm.rpp + rm.rs; r < MP; r += (long)gridDim.x * rm.rpp) {
        const long off = pr.base(r) + rm.cg * N;
        const long koff[4] = {0, C, (long)W * C, (long)W * C + C};
        Vec16<T> v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ld16(Y + off + koff[k]);
        const Vec16<T> d = ld16(dP + r * lddp + rm.cg * N);
        const typename IdxPack<N>::type pack = *reinterpret_cast<const typename IdxPack<N>::type *>(idx + r * C + rm.cg * N);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float da = (int)((pack >> (8 * j)) & 3) == k ? d.get(j) : 0.f;
                const float xh = (v[k].get(j) - mu[j]) * inv[j];
                const float z = (v[k].get(j) - mu[j]) * (inv[j] * ga[j]) + bt[j];
                const float g = z >= 0.f ? da : alpha * da;
                o.set(j, (ga[j] * inv[j]) * (g - dbm[j] - xh * dgm[j]));
            }
            st16(dY + off + koff[k], o);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Consumers that finalise the partial rows themselves (round 3): the 36 bn_finalize / 7 reduce_finalize launches of a training
// step were ~5 us each (a 16-workgroup kernel is all latency) plus a kernel boundary.  Here the kernel that NEEDS the batch
// moments (BN apply) or dgamma / dbeta (BN backward apply) sums the partial rows of its own channels in its prologue.
// Mapping: a workgroup owns ONE channel slice (blockIdx.y: up to 16 lanes x 16 bytes = 128 bf16 / 64 f32 channels) and strides
// over pixel rows (blockIdx.x), so its prologue reads rows x slice x 2 floats (L2-resident: the producer just wrote them) instead
// of rows x C x 2; the 256 threads split the rows, accumulate in f64, and meet in LDS.  The workgroups with blockIdx.x == 0 store
// mean / var (+ moving averages) or dgamma / dbeta for their slice.  Rows are read-only here: a buffer that needs to be zero for
// its next producer is cleared by the NEXT consumer kernel, which works on the other buffer of a pair (zero / zero_vec4 arguments).
// ------------------------------------------------------------------------------------------
struct SliceMap {   // 256 threads = rpb rows x lpr lanes; lane -> 16-byte channel group cg of slice blockIdx.y
    int lpr, rpb, lane, row, cg, cs, c0;
    __device__ SliceMap(int C, int vec) {
        const int tpr = C / vec;
        lpr = tpr < 16 ? tpr : 16;
        rpb = 256 / lpr;
        lane = threadIdx.x % lpr;
        row = threadIdx.x / lpr;
        cg = blockIdx.y * lpr + lane;
        cs = lpr * vec;                  // channels of the slice (<= 128)
        c0 = blockIdx.y * cs;
    }
};
#define Y2_SLICE_MAX 128

// thread c < cs: sums[k] = sum over rows of part[k * plane + row * C + c0 + c] (f64); ends with a barrier.  A thread owns four adjacent
// channels (one 16-byte load per row and plane) and every (256 / (cs / 4))-th row, so even a 128-channel slice has eight row groups
// working in parallel: the prologue is a dependent chain in front of the whole workgroup and its length is what the fold pays.
// the per-thread sums of the row groups that share a wave meet on the VALU (G = lanes per row: the lanes that share lane % G), then the four waves in LDS
template <int G>
__device__ __forceinline__ void slice_wave_sums(double (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = y2_lane_group_sum_f64<G>(v[j]);
}
__device__ __forceinline__ void slice_partial_sums(const float *__restrict__ part, int rows, long plane, int C, const SliceMap &sm, double (&sums)[2]) {
    // (round 6) 4 KB of scratch -- [wave][channel of the slice] per plane, one plane after the other -- and the results in registers of the threads that
    // use them (thread c < cs: sums[k] of channel c0 + c) instead of 16 + 2 KB: with < 8 KB of LDS these workgroups fit on a CU beside ANY
    // filter-gradient workgroup (140 .. 152 KB), which is where the side stream wants them
    __shared__ double red[4 * Y2_SLICE_MAX];
    const int q = sm.cs >> 2;                               // lanes per row (cs >= 8 for bf16, >= 4 for f32: q >= 1)
    const int l4 = threadIdx.x % q, g = threadIdx.x / q, ng = 256 / q;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool valu = q == 4 || q == 8 || q == 16 || q == 32;      // (q = 1, 2: tiny slices, the general path; 64 % q == 0 always)
    sums[0] = sums[1] = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float *p = part + (long)k * plane + sm.c0 + l4 * 4;
        // four independent 16-byte loads in flight per thread, the last group included (a row beyond the end loads row g again with weight 0): the
        // prologue is a chain of L2 latencies in front of the whole workgroup -- 44 rows over 8 row groups are two rounds instead of four
        double s[4] = {0.0, 0.0, 0.0, 0.0}, t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r = g; r < rows; r += 4 * ng) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4 *>(p + (long)(r + u * ng < rows ? r + u * ng : g) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = r + u * ng < rows;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double x = ok ? (double)v[u][j] : 0.0;
                    if (u & 1) t[j] += x; else s[j] += x;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += t[j];
        if (k) __syncthreads();                              // (plane 0's scratch has been read)
        if (valu) {
            if (q == 4) slice_wave_sums<4>(s); else if (q == 8) slice_wave_sums<8>(s); else if (q == 16) slice_wave_sums<16>(s); else slice_wave_sums<32>(s);
            if (lane < q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) red[wave * sm.cs + lane * 4 + j] = s[j];
            }
            __syncthreads();
            if (threadIdx.x < sm.cs) sums[k] = (red[threadIdx.x] + red[sm.cs + threadIdx.x]) + (red[2 * sm.cs + threadIdx.x] + red[3 * sm.cs + threadIdx.x]);
        } else {
            // general path (slices of 4 or 8 channels: one or two lanes per row, 128 .. 256 row groups): through the same scratch in rounds of
            // 4 * Y2_SLICE_MAX / cs row groups
            const int gmax = 4 * Y2_SLICE_MAX / sm.cs;
            double a = 0.0;
            for (int g0 = 0; g0 < ng; g0 += gmax) {
                if (g0) __syncthreads();
                if (g >= g0 && g < g0 + gmax) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) red[(g - g0) * sm.cs + l4 * 4 + j] = s[j];
                }
                __syncthreads();
                if (threadIdx.x < sm.cs)
                    for (int j = 0; j < gmax && g0 + j < ng; ++j) a += red[j * sm.cs + threadIdx.x];
            }
            if (threadIdx.x < sm.cs) sums[k] = a;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void grid_zero(float *__restrict__ zero, long zero_vec4) {
    if (!zero) return;
    const long nthreads = (long)gridDim.x * gridDim.y * 256;
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < zero_vec4; i += nthreads) reinterpret_cast<f32x4 *>(zero)[i] = z;
}

// forward: batch moments from the partial rows (same arithmetic as bn_finalize_kernel<0>) + BN apply + leaky (+ 2x2 max pool)
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_leaky_fin_kernel(const T *__restrict__ Y, const float *__restrict__ part, int rows, const float *__restrict__ shift,
                                                           long Mstat, float *__restrict__ mean_out, float *__restrict__ var_out, float *__restrict__ mm,
                                                           float *__restrict__ mv, float omd, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           T *__restrict__ A, unsigned char *__restrict__ idx, T *__restrict__ ymax, T *__restrict__ Afull,
                                                           int B, int H, int W, int C, int lda, float eps, float alpha, float *__restrict__ zero, long zero_vec4) {
    constexpr int N = Vec16<T>::N;
    const SliceMap sm(C, N);
    double sums[2];
    __shared__ float cst[3][Y2_SLICE_MAX];
    // the first pixel row's data is requested BEFORE the prologue: its HBM latency runs under the partial-row reduction
    const PoolRow pr(H, W, C);
    const long ML = POOL ? (long)B * pr.OH * pr.OW : (long)B * H * W;
    const long step = (long)gridDim.x * sm.rpb;
    long r = (long)blockIdx.x * sm.rpb + sm.row;
    Vec16<T> v[POOL ? 4 : 1];
    if (r < ML) {
        if (POOL) {
            const T *src = Y + pr.base(r) + sm.cg * N;
            v[0] = ld16(src);
            v[POOL ? 1 : 0] = ld16(src + C);
            v[POOL ? 2 : 0] = ld16(src + (long)W * C);
            v[POOL ? 3 : 0] = ld16(src + (long)W * C + C);
        } else v[0] = ld16(Y + r * C + sm.cg * N);
    }
    slice_partial_sums(part, rows, (long)Y2_BN_PART_ROWS * C, C, sm, sums);
    if (threadIdx.x < sm.cs) {
        const int c = sm.c0 + threadIdx.x;
        const double dm = sums[0] / (double)Mstat;
        const double var = sums[1] / (double)Mstat - dm * dm;
        const float fm = (float)((double)shift[c] + dm), fv = (float)(var > 0.0 ? var : 0.0);
        cst[0][threadIdx.x] = fm;
        cst[1][threadIdx.x] = (1.0f / sqrtf(fv + eps)) * gamma[c];
        cst[2][threadIdx.x] = beta[c];
        if (blockIdx.x == 0) {
            mean_out[c] = fm;
            var_out[c] = fv;
            if (mm) {
                mm[c] = mm[c] - (mm[c] - fm) * omd;
                mv[c] = mv[c] - (mv[c] - fv) * omd;
            }
        }
    }
    __syncthreads();
    float mu[N], sc[N], bt[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        mu[j] = cst[0][sm.lane * N + j];
        sc[j] = cst[1][sm.lane * N + j];
        bt[j] = cst[2][sm.lane * N + j];
    }
    grid_zero(zero, zero_vec4);
    while (r < ML) {
        const long rn = r + step;
        Vec16<T> vn[POOL ? 4 : 1];
        if (rn < ML) {       // next row in flight while this one is computed and stored
            if (POOL) {
                const T *src = Y + pr.base(rn) + sm.cg * N;
                vn[0] = ld16(src);
                vn[POOL ? 1 : 0] = ld16(src + C);
                vn[POOL ? 2 : 0] = ld16(src + (long)W * C);
                vn[POOL ? 3 : 0] = ld16(src + (long)W * C + C);
            } else vn[0] = ld16(Y + rn * C + sm.cg * N);
        }
        Vec16<T> o;
        if (!POOL) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float z = (v[0].get(j) - mu[j]) * sc[j] + bt[j];
                o.set(j, fmaxf(z, alpha * z));
            }
            st16(A + r * lda + sm.cg * N, o);
        } else {
            typename IdxPack<N>::type pack = 0;
            Vec16<T> ym, af[4];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float a[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = (v[POOL ? k : 0].get(j) - mu[j]) * sc[j] + bt[j];
                    a[k] = (float)(T)fmaxf(z, alpha * z);
                    af[k].set(j, a[k]);
                }
                const float m = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
                const int arg = a[0] == m ? 0 : a[1] == m ? 1 : a[2] == m ? 2 : 3;
                o.set(j, m);
                ym.set(j, arg == 0 ? v[0].get(j) : arg == 1 ? v[POOL ? 1 : 0].get(j) : arg == 2 ? v[POOL ? 2 : 0].get(j) : v[POOL ? 3 : 0].get(j));
                pack |= (typename IdxPack<N>::type)arg << (8 * j);
            }
            st16(A + r * lda + sm.cg * N, o);
            if (idx) *reinterpret_cast<typename IdxPack<N>::type *>(idx + r * C + sm.cg * N) = pack;
            if (Afull) {       // the activation has another reader besides the pool (Darknet-19's passthrough at the 26x26 stage): dense [B][H][W][C]
                T *dst = Afull + pr.base(r) + sm.cg * N;
                st16(dst, af[0]);
                st16(dst + C, af[1]);
                st16(dst + (long)W * C, af[2]);
                st16(dst + (long)W * C + C, af[3]);
            }
            if (ymax) st16(ymax + r * C + sm.cg * N, ym);       // the raw convolution output at the arg-max (the backward reduction reads this, not Y)
        }
#pragma unroll
        for (int k = 0; k < (POOL ? 4 : 1); ++k) v[k] = vn[k];
        r = rn;
    }
}

// backward: dgamma / dbeta from the partial rows (plain sums, as bn_finalize_kernel<1> / reduce_finalize_kernel<1>) + the apply pass
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_fin_kernel(const T *__restrict__ dA, int ldda, const unsigned char *__restrict__ idx, const T *__restrict__ Y,
                                                               const float *__restrict__ mean, const float *__restrict__ var, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, const float *__restrict__ part, int rows, long plane,
                                                               float *__restrict__ dgamma, float *__restrict__ dbeta, T *__restrict__ dY, int B, int H, int W, int C,
                                                               float eps, float alpha, float *__restrict__ zero, long zero_vec4) {
    constexpr int N = Vec16<T>::N;
    const SliceMap sm(C, N);
    double sums[2];
    __shared__ float cst[2][Y2_SLICE_MAX];
    const PoolRow pr(H, W, C);
    const long ML = POOL ? (long)B * pr.OH * pr.OW : (long)B * H * W;
    const long step = (long)gridDim.x * sm.rpb;
    long r = (long)blockIdx.x * sm.rpb + sm.row;
    const long koff[4] = {0, C, (long)W * C, (long)W * C + C};
    Vec16<T> v[POOL ? 4 : 1], d;
    typename IdxPack<N>::type pack = 0;
    auto fetch = [&](long row, Vec16<T> (&yv)[POOL ? 4 : 1], Vec16<T> &dv, typename IdxPack<N>::type &pk) {
        if (POOL) {
            const long off = pr.base(row) + sm.cg * N;
#pragma unroll
            for (int k = 0; k < (POOL ? 4 : 1); ++k) yv[k] = ld16(Y + off + koff[k]);
            pk = *reinterpret_cast<const typename IdxPack<N>::type *>(idx + row * C + sm.cg * N);
        } else yv[0] = ld16(Y + row * C + sm.cg * N);
        dv = ld16(dA + row * ldda + sm.cg * N);
    };
    if (r < ML) fetch(r, v, d, pack);        // in flight under the prologue
    slice_partial_sums(part, rows, plane, C, sm, sums);
    if (threadIdx.x < sm.cs) {
        const float dg = (float)sums[0], db = (float)sums[1];
        cst[0][threadIdx.x] = dg;
        cst[1][threadIdx.x] = db;
        if (blockIdx.x == 0) {
            dgamma[sm.c0 + threadIdx.x] = dg;
            dbeta[sm.c0 + threadIdx.x] = db;
        }
    }
    __syncthreads();
    const float invM = 1.0f / (float)((long)B * H * W);
    float mu[N], inv[N], ga[N], bt[N], dgm[N], dbm[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int c = sm.cg * N + j;
        mu[j] = mean[c];
        inv[j] = 1.0f / sqrtf(var[c] + eps);
        ga[j] = gamma[c];
        bt[j] = beta[c];
        dgm[j] = cst[0][sm.lane * N + j] * invM;
        dbm[j] = cst[1][sm.lane * N + j] * invM;
    }
    grid_zero(zero, zero_vec4);
    while (r < ML) {
        const long rn = r + step;
        Vec16<T> vn[POOL ? 4 : 1], dn;
        typename IdxPack<N>::type packn = 0;
        if (rn < ML) fetch(rn, vn, dn, packn);
        if (!POOL) {
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float xh = (v[0].get(j) - mu[j]) * inv[j];
                float z = (v[0].get(j) - mu[j]) * (inv[j] * ga[j]) + bt[j];
                float g = z >= 0.f ? d.get(j) : alpha * d.get(j);
                o.set(j, (ga[j] * inv[j]) * (g - dbm[j] - xh * dgm[j]));
            }
            st16(dY + r * C + sm.cg * N, o);
        } else {
            const long off = pr.base(r) + sm.cg * N;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float da = (int)((pack >> (8 * j)) & 3) == k ? d.get(j) : 0.f;
                    const float xh = (v[POOL ? k : 0].get(j) - mu[j]) * inv[j];
                    const float z = (v[POOL ? k : 0].get(j) - mu[j]) * (inv[j] * ga[j]) + bt[j];
                    const float g = z >= 0.f ? da : alpha * da;
                    o.set(j, (ga[j] * inv[j]) * (g - dbm[j] - xh * dgm[j]));
                }
                st16(dY + off + koff[k], o);
            }
        }
#pragma unroll
        for (int k = 0; k < (POOL ? 4 : 1); ++k) v[k] = vn[k];
        d = dn;
        pack = packn;
        r = rn;
    }
}

// ------------------------------------------------------------------------------------------
// max pool 2x2 SAME
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T *__restrict__ A, T *__restrict__ P, int B, int H, int W, int C, int stride) {
    constexpr int N = Vec16<T>::N;
    const int OH = stride == 2 ? H / 2 : H, OW = stride == 2 ? W / 2 : W;
    const int cgs = C / N;
    const long total = (long)B * OH * OW * cgs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        long p = i / cgs;
        int ow = (int)(p % OW);
        long q = p / OW;
        int oh = (int)(q % OH);
        int b = (int)(q / OH);
        const int h0 = oh * stride, w0 = ow * stride;
        Vec16<T> m = ld16(A + (((long)b * H + h0) * W + w0) * C + cg * N);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            int hh = h0 + (k >> 1), ww = w0 + (k & 1);
            if (hh < H && ww < W) {
                Vec16<T> v = ld16(A + (((long)b * H + hh) * W + ww) * C + cg * N);
#pragma unroll
                for (int j = 0; j < N; ++j) m.set(j, fmaxf(m.get(j), v.get(j)));
            }
        }
        st16(P + p * C + cg * N, m);
    }
}

// stride 2: one thread per pooled chunk writes all four input positions (full overwrite of dA)
template <typename T, bool ACC>      // ACC: dA += (a second writer of the tensor's gradient: passthrough fan-out), rounded to T like a separate add
__global__ void maxpool_bwd_s2_kernel(const T *__restrict__ A, const T *__restrict__ dP, T *dA, int B, int H, int W, int C) {
    constexpr int N = Vec16<T>::N;
    const int OH = H / 2, OW = W / 2, cgs = C / N;
    const long total = (long)B * OH * OW * cgs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        long p = i / cgs;
        int ow = (int)(p % OW);
        long q = p / OW;
        int oh = (int)(q % OH);
        int b = (int)(q / OH);
        Vec16<T> v[4], o[4];
        const long base = (((long)b * H + oh * 2) * W + ow * 2) * C + cg * N;
        v[0] = ld16(A + base);
        v[1] = ld16(A + base + C);
        v[2] = ld16(A + base + (long)W * C);
        v[3] = ld16(A + base + (long)W * C + C);
        Vec16<T> g = ld16(dP + p * C + cg * N);
        if (ACC) {
            o[0] = ld16(dA + base);
            o[1] = ld16(dA + base + C);
            o[2] = ld16(dA + base + (long)W * C);
            o[3] = ld16(dA + base + (long)W * C + C);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float m = fmaxf(fmaxf(v[0].get(j), v[1].get(j)), fmaxf(v[2].get(j), v[3].get(j)));
            int arg = v[0].get(j) == m ? 0 : v[1].get(j) == m ? 1 : v[2].get(j) == m ? 2 : 3;  // first max in scan order
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k].set(j, (ACC ? o[k].get(j) : 0.f) + (k == arg ? g.get(j) : 0.f));
        }
        st16(dA + base, o[0]);
        st16(dA + base + C, o[1]);
        st16(dA + base + (long)W * C, o[2]);
        st16(dA + base + (long)W * C + C, o[3]);
    }
}

// stride 1 (tiny model): one thread per INPUT chunk gathers from the <=4 windows containing it
template <typename T>
__global__ void maxpool_bwd_s1_kernel(const T *__restrict__ A, const T *__restrict__ dP, T *__restrict__ dA, int B, int H, int W, int C) {
    constexpr int N = Vec16<T>::N;
    const int cgs = C / N;
    const long total = (long)B * H * W * cgs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        long p = i / cgs;
        int w = (int)(p % W);
        long q = p / W;
        int h = (int)(q % H);
        int b = (int)(q / H);
        float acc[N];
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] = 0.f;
        for (int oh = h - 1; oh <= h; ++oh) {
            if (oh < 0) continue;
            for (int ow = w - 1; ow <= w; ++ow) {
                if (ow < 0) continue;
                // window (oh, ow) covers (oh..oh+1, ow..ow+1); position of (h, w) inside it:
                const int mypos = (h - oh) * 2 + (w - ow);
                Vec16<T> v[4];
                bool ok[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int hh = oh + (k >> 1), ww = ow + (k & 1);
                    ok[k] = hh < H && ww < W;
                    v[k] = ok[k] ? ld16(A + (((long)b * H + hh) * W + ww) * C + cg * N) : zero16<T>();
                }
                Vec16<T> g = ld16(dP + (((long)b * H + oh) * W + ow) * C + cg * N);
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float m = -INFINITY;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (ok[k]) m = fmaxf(m, v[k].get(j));
                    int arg = 3;
#pragma unroll
                    for (int k = 3; k >= 0; --k) if (ok[k] && v[k].get(j) == m) arg = k;
                    if (arg == mypos) acc[j] += g.get(j);
                }
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int j = 0; j < N; ++j) o.set(j, acc[j]);
        st16(dA + p * C + cg * N, o);
    }
}

static int ew_grid(long total) {
    long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int yolo2_maxpool_fwd(const void *A, void *P, int B, int H, int W, int C, int stride, int dtype, void *stream) {
    Y2_CHECK_ARG(A && P && B > 0 && H > 0 && W > 0 && C > 0);
    Y2_CHECK_ARG(stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0));
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0);
    long total = (long)B * (stride == 2 ? H / 2 : H) * (stride == 2 ? W / 2 : W) * (C / vec);
    Y2_DISPATCH_DTYPE(dtype, maxpool_fwd_kernel<T><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const T *)A, (T *)P, B, H, W, C, stride));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_maxpool_bwd(const void *A, const void *dP, void *dA, int B, int H, int W, int C, int stride, int dtype, void *stream) {
    Y2_CHECK_ARG(A && dP && dA && B > 0 && H > 0 && W > 0 && C > 0);
    Y2_CHECK_ARG(stride == 1 || (stride == 2 && H % 2 == 0 && W % 2 == 0));
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0);
    hipStream_t st = (hipStream_t)stream;
    if (stride == 2) {
        long total = (long)B * (H / 2) * (W / 2) * (C / vec);
        Y2_DISPATCH_DTYPE(dtype, maxpool_bwd_s2_kernel<T, false><<<ew_grid(total), 256, 0, st>>>((const T *)A, (const T *)dP, (T *)dA, B, H, W, C));
    } else {
        long total = (long)B * H * W * (C / vec);
        Y2_DISPATCH_DTYPE(dtype, maxpool_bwd_s1_kernel<T><<<ew_grid(total), 256, 0, st>>>((const T *)A, (const T *)dP, (T *)dA, B, H, W, C));
    }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// dA += the routed gradient (stride 2): yolo2_maxpool_bwd into a temporary + yolo2_add_inplace in one launch, same rounding
extern "C" int yolo2_maxpool_bwd_acc(const void *A, const void *dP, void *dA, int B, int H, int W, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(A && dP && dA && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0);
    long total = (long)B * (H / 2) * (W / 2) * (C / vec);
    Y2_DISPATCH_DTYPE(dtype, maxpool_bwd_s2_kernel<T, true><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const T *)A, (const T *)dP, (T *)dA, B, H, W, C));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// reorg (space-to-depth, model/yolo2/function.py:22-29) and channel-slice moves
// ------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ void reorg_kernel(const T *__restrict__ src, T *__restrict__ dst, int B, int H, int W, int C, int ld) {
    // forward: src = in [B,H,W,C], dst = out [B,H/2,W/2,ld];  backward: src = dout (stride ld), dst = din
    constexpr int N = Vec16<T>::N;
    const int OH = H / 2, OW = W / 2, cgs = C / N;
    const long total = (long)B * H * W * cgs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        long p = i / cgs;  // input pixel index (b, h, w)
        int w = (int)(p % W);
        long q = p / W;
        int h = (int)(q % H);
        int b = (int)(q / H);
        const long in_off = p * C + cg * N;
        const long out_off = (((long)b * OH + (h >> 1)) * OW + (w >> 1)) * ld + ((h & 1) * 2 + (w & 1)) * C + cg * N;
        if (!BWD) st16(dst + out_off, ld16(src + in_off));
        else st16(dst + in_off, ld16(src + out_off));
    }
}
extern "C" int yolo2_reorg(const void *in, void *out, int B, int H, int W, int C, int ldo, int dtype, void *stream) {
    Y2_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && ldo >= 4 * C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && ldo % vec == 0);
    long total = (long)B * H * W * (C / vec);
    Y2_DISPATCH_DTYPE(dtype, reorg_kernel<T, false><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const T *)in, (T *)out, B, H, W, C, ldo));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_reorg_bwd(const void *dout, int ldd, void *din, int B, int H, int W, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(dout && din && B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && ldd >= 4 * C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && ldd % vec == 0);
    long total = (long)B * H * W * (C / vec);
    Y2_DISPATCH_DTYPE(dtype, reorg_kernel<T, true><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const T *)dout, (T *)din, B, H, W, C, ldd));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

template <typename T>
__global__ void copy_channels_kernel(const T *__restrict__ src, int lds, T *__restrict__ dst, int ldd, long M, int C) {
    constexpr int N = Vec16<T>::N;
    const int cgs = C / N;
    const long total = M * cgs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % cgs);
        long r = i / cgs;
        st16(dst + r * ldd + cg * N, ld16(src + r * lds + cg * N));
    }
}
extern "C" int yolo2_copy_channels(const void *src, int lds, void *dst, int ldd, long M, int C, int dtype, void *stream) {
    Y2_CHECK_ARG(src && dst && M > 0 && C > 0 && lds >= C && ldd >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && lds % vec == 0 && ldd % vec == 0);
    Y2_DISPATCH_DTYPE(dtype, copy_channels_kernel<T><<<ew_grid(M * (C / vec)), 256, 0, (hipStream_t)stream>>>((const T *)src, lds, (T *)dst, ldd, M, C));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

template <typename T>
__global__ void add_inplace_kernel(T *__restrict__ dst, const T *__restrict__ src, long nvec) {
    constexpr int N = Vec16<T>::N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        Vec16<T> a = ld16(dst + i * N), b = ld16(src + i * N), o;
#pragma unroll
        for (int j = 0; j < N; ++j) o.set(j, a.get(j) + b.get(j));
        st16(dst + i * N, o);
    }
}
extern "C" int yolo2_add_inplace(void *dst, const void *src, long n, int dtype, void *stream) {
    Y2_CHECK_ARG(dst && src && n > 0);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(n % vec == 0);
    Y2_DISPATCH_DTYPE(dtype, add_inplace_kernel<T><<<ew_grid(n / vec), 256, 0, (hipStream_t)stream>>>((T *)dst, (const T *)src, n / vec));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// image prep (tf.image.per_image_standardization, train.py:103; utils/preprocess.py:23-25)
// ------------------------------------------------------------------------------------------
// Two launches, no memset, no atomics (round 3; was memset + atomic sums + apply): every workgroup of the first kernel stores its (sum,
// sum of squares) pair -- f64, 16-byte loads, four independent chains -- into ws[b][block][2]; the second kernel's workgroups each work
// on ONE image and fold that image's Y2_IMG_PARTS partial pairs in their prologue (the finalisation rides in the consumer, as for the
// batch-norm statistics).  The partial layout makes the result independent of scheduling: bit-reproducible run to run.
#define Y2_IMG_PARTS 64
__global__ __launch_bounds__(256) void image_sums_kernel(const float *__restrict__ img, double *__restrict__ ws, long n_per_image) {
    const int b = blockIdx.y;
    const float *p = img + (long)b * n_per_image;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if ((((uintptr_t)p) & 15) == 0) {         // 16-byte loads, 4 independent f64 chains per quantity
        const long n4 = n_per_image >> 2;
        const f32x4 *p4 = reinterpret_cast<const f32x4 *>(p);
        for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-byte loads in flight per lane (one per iteration left the
            f32x4 v[4];                                       // kernel latency-bound at 1 TB/s)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double d = (double)v[u][j];
                    s[j] += d;
                    q[j] += d * d;
                }
        }
        for (; i < n4; i += stride) {
            const f32x4 v = p4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = (double)v[j];
                s[j] += d;
                q[j] += d * d;
            }
        }
        i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x;
    }
    for (; i < n_per_image; i += stride) {
        const double d = (double)p[i];
        s[0] += d;
        q[0] += d * d;
    }
    __shared__ double red[2][4];
    const double st = wave_sum_d((s[0] + s[1]) + (s[2] + s[3]));
    const double qt = wave_sum_d((q[0] + q[1]) + (q[2] + q[3]));
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = st; red[1][threadIdx.x >> 6] = qt; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const double *r = red[threadIdx.x];
        ws[((long)b * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
    }
}
// grid = (blocks per image, B): the image index is blockIdx.y; parts = partial pairs per image left by image_sums_kernel (mode 0)
template <typename T>
__global__ __launch_bounds__(256) void image_apply_kernel(const float *__restrict__ img, T *__restrict__ out, const double *__restrict__ ws, long HW, int mode, int parts) {
    const int b = blockIdx.y;
    float sub = 0.f, den = 1.f;
    if (mode == 0) {
        __shared__ double red[2][4];
        double a = 0.0, c = 0.0;
        for (int k = threadIdx.x; k < parts; k += 256) { a += ws[((long)b * parts + k) * 2]; c += ws[((long)b * parts + k) * 2 + 1]; }
        a = wave_sum_d(a);
        c = wave_sum_d(c);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = c; }
        __syncthreads();
        const double sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), sq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        const double n = (double)HW * 3.0;
        const double mean = sum / n;
        double var = sq / n - mean * mean;
        if (var < 0) var = 0;
        sub = (float)mean;
        den = fmaxf((float)sqrt(var), (float)(1.0 / sqrt(n)));
    } else if (mode == 1) {
        den = 255.0f;
    }
    const float *pi = img + (long)b * HW * 3;
    T *po = out + (long)b * HW * 8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
        const float *p = pi + i * 3;
        float v0 = p[0], v1 = p[1], v2 = p[2];
        if (mode != 2) { v0 = (v0 - sub) / den; v1 = (v1 - sub) / den; v2 = (v2 - sub) / den; }
        Vec16<T> o[sizeof(T) == 2 ? 1 : 2];
        if constexpr (sizeof(T) == 2) {
            o[0].set(0, v0); o[0].set(1, v1); o[0].set(2, v2);
#pragma unroll
            for (int j = 3; j < 8; ++j) o[0].set(j, 0.f);
            st16(po + i * 8, o[0]);
        } else {
            o[0].set(0, v0); o[0].set(1, v1); o[0].set(2, v2); o[0].set(3, 0.f);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[1].set(j, 0.f);
            st16(po + i * 8, o[0]);
            st16(po + i * 8 + 4, o[1]);
        }
    }
}
extern "C" int yolo2_image_prep(const float *img, void *out, double *ws, int B, int HW, int mode, int dtype, void *stream) {
    Y2_CHECK_ARG(img && out && B > 0 && HW > 0 && mode >= 0 && mode <= 2 && ((uintptr_t)out & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    int parts = 0;
    if (mode == 0) {
        Y2_CHECK_ARG(ws);
        parts = Y2_IMG_PARTS;
        image_sums_kernel<<<dim3(parts, B), 256, 0, st>>>(img, ws, (long)HW * 3);
    }
    int gx = (int)(((long)HW + 1023) / 1024);        // ~4 pixels per thread
    if (gx > 256) gx = 256;
    if ((long)gx * B > 8192) gx = 8192 / B > 0 ? 8192 / B : 1;
    Y2_DISPATCH_DTYPE(dtype, image_apply_kernel<T><<<dim3(gx, B), 256, 0, st>>>(img, (T *)out, ws, (long)HW, mode, parts));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// optimizers, TF-1.0 Apply* semantics (train.py:70-80); g is scaled by gscale first (1/world
// for data-parallel gradient averaging)
// ------------------------------------------------------------------------------------------
#define OPT_LOOP(n) for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// the dominant optimizer: 28 B of HBM traffic per parameter.  16-byte accesses when the four arenas are 16-byte aligned
// (they are: the engine's arenas and every all-reduce bucket start on a multiple of 4 elements), scalar tail otherwise.
__device__ __forceinline__ void adam_one(float &w, float g, float &m, float &v, float alpha, float omb1, float omb2, float eps, float gs) {
    // no FMA contraction: the update is inlined into several kernels (adam_kernel, adam_filter_prep_kernel) whose results must agree bit
    // for bit, and separate multiplies / adds are what the oracle (and TF's Eigen expression) evaluates
#pragma clang fp contract(off)
    const float gi = g * gs;
    const float mi = m + (gi - m) * omb1;
    const float vi = v + (gi * gi - v) * omb2;
    m = mi;
    v = vi;
    w = w - (mi * alpha) / (sqrtf(vi) + eps);
}
__global__ __launch_bounds__(256) void adam_kernel(float *w, const float *g, float *m, float *v, long n, float alpha, float omb1, float omb2, float eps, float gs) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    long done = 0;
    if (((((uintptr_t)w) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
        const long n4 = n >> 2;
        f32x4 *w4 = reinterpret_cast<f32x4 *>(w), *m4 = reinterpret_cast<f32x4 *>(m), *v4 = reinterpret_cast<f32x4 *>(v);
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
        for (long k = i; k < n4; k += stride) {
            f32x4 wv = w4[k], mv = m4[k], vv = v4[k];
            const f32x4 gv = g4[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float wj = wv[j], mj = mv[j], vj = vv[j];
                adam_one(wj, gv[j], mj, vj, alpha, omb1, omb2, eps, gs);
                wv[j] = wj; mv[j] = mj; vv[j] = vj;
            }
            m4[k] = mv;
            v4[k] = vv;
            w4[k] = wv;
        }
        done = n4 << 2;
    }
    for (long k = done + i; k < n; k += stride) adam_one(w[k], g[k], m[k], v[k], alpha, omb1, omb2, eps, gs);
}
__global__ void momentum_kernel(float *w, const float *g, float *acc, long n, float lr, float mom, float gs) {
    OPT_LOOP(n) {
        float a = acc[i] * mom + g[i] * gs;
        acc[i] = a;
        w[i] = w[i] - lr * a;
    }
}
__global__ void sgd_kernel(float *w, const float *g, long n, float lr, float gs) {
    OPT_LOOP(n) w[i] = w[i] - lr * (g[i] * gs);
}
__global__ void rmsprop_kernel(float *w, const float *g, float *ms, float *mom, long n, float lr, float omd, float momentum, float eps, float gs) {
    OPT_LOOP(n) {
        float gi = g[i] * gs;
        float s = ms[i] + (gi * gi - ms[i]) * omd;
        float mo = mom[i] * momentum + lr * gi / sqrtf(s + eps);
        ms[i] = s;
        mom[i] = mo;
        w[i] = w[i] - mo;
    }
}
__global__ void adagrad_kernel(float *w, const float *g, float *acc, long n, float lr, float gs) {
    OPT_LOOP(n) {
        float gi = g[i] * gs;
        float a = acc[i] + gi * gi;
        acc[i] = a;
        w[i] = w[i] - lr * gi / sqrtf(a);
    }
}
__global__ void adadelta_kernel(float *w, const float *g, float *acc, float *accu, long n, float lr, float rho, float eps, float gs) {
    OPT_LOOP(n) {
        float gi = g[i] * gs;
        float a = acc[i] * rho + gi * gi * (1.0f - rho);
        float u = sqrtf(accu[i] + eps) / sqrtf(a + eps) * gi;
        accu[i] = accu[i] * rho + u * u * (1.0f - rho);
        acc[i] = a;
        w[i] = w[i] - lr * u;
    }
}
// [TF-sem] ApplyFtrl (tf.train.FtrlOptimizer, reference train.py:78): accum starts at initial_accumulator_value, linear at 0.
//   new_accum = accum + g^2;  linear += g - (new_accum^-p - accum^-p) / lr * w;   (p = learning_rate_power, sqrt when p = -0.5)
//   w = |linear| > l1 ? (l1 * sign(linear) - linear) / (new_accum^-p / lr + 2 * l2) : 0
__global__ void ftrl_kernel(float *w, const float *g, float *accum, float *linear, long n, float lr, float lr_power, float l1, float l2, float gs) {
    const bool half = lr_power == -0.5f;
    OPT_LOOP(n) {
        const float gi = g[i] * gs;
        const float a = accum[i], na = a + gi * gi;
        const float pa = half ? sqrtf(a) : powf(a, -lr_power), pna = half ? sqrtf(na) : powf(na, -lr_power);
        const float li = linear[i] + (gi - (pna - pa) / lr * w[i]);
        const float sgn = li > 0.f ? 1.f : (li < 0.f ? -1.f : 0.f);
        const float x = l1 * sgn - li;
        const float y = pna / lr + 2.0f * l2;
        w[i] = fabsf(li) > l1 ? x / y : 0.f;
        linear[i] = li;
        accum[i] = na;
    }
}
__global__ void scale_kernel(float *x, long n, float sc) {
    OPT_LOOP(n) x[i] = x[i] * sc;
}
// inference-time batch-norm folding: Wf[r, n] = W[r, n] * s[n], bias[n] = beta[n] - mean[n] * s[n], s = gamma / sqrt(var + eps)
__global__ void bn_fold_kernel(const float *__restrict__ W, const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean,
                               const float *__restrict__ var, float *__restrict__ Wf, float *__restrict__ bias, long rows, int C, float eps) {
    const long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i % C);
        const float sc = gamma[n] / sqrtf(var[n] + eps);
        Wf[i] = W[i] * sc;
        if (i < C) bias[n] = beta[n] - mean[n] * sc;
    }
}

extern "C" int yolo2_ftrl(float *w, const float *g, float *accum, float *linear, long n, float lr, float lr_power, float l1, float l2, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && accum && linear && n > 0 && lr > 0.f && lr_power <= 0.f);
    ftrl_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, accum, linear, n, lr, lr_power, l1, l2, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_scale(float *x, long n, float scale, void *stream) {
    Y2_CHECK_ARG(x && n > 0);
    scale_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(x, n, scale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
// up to Y2_ZR_MAX ranges per launch, passed by value (one launch instead of one hipMemsetAsync node per range: 5 launches per training step)
#define Y2_ZR_MAX 16
struct Y2ZeroRanges { long a[Y2_ZR_MAX], b[Y2_ZR_MAX]; int n; };
__global__ __launch_bounds__(256) void zero_ranges_kernel(float *__restrict__ x, const Y2ZeroRanges zr) {
    const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    for (int r = 0; r < zr.n; ++r) {
        const long a = zr.a[r], b = zr.b[r];
        const long a4 = (a + 3) & ~3L, b4 = b & ~3L;            // 16-byte body, scalar edges
        if (a4 <= b4) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (long i = a4 / 4 + tid; i < b4 / 4; i += stride) reinterpret_cast<f32x4 *>(x)[i] = z;
            for (long i = a + tid; i < a4; i += stride) x[i] = 0.f;
            for (long i = b4 + tid; i < b; i += stride) x[i] = 0.f;
        } else {
            for (long i = a + tid; i < b; i += stride) x[i] = 0.f;
        }
    }
}
extern "C" int yolo2_zero_ranges(float *x, const long *ranges_host, int nranges, void *stream) {
    Y2_CHECK_ARG(x && (nranges == 0 || ranges_host) && nranges >= 0 && ((uintptr_t)x & 15) == 0);
    for (int i0 = 0; i0 < nranges; i0 += Y2_ZR_MAX) {
        Y2ZeroRanges zr;
        zr.n = 0;
        long total = 0;
        for (int i = i0; i < nranges && i < i0 + Y2_ZR_MAX; ++i) {
            const long a = ranges_host[2 * i], b = ranges_host[2 * i + 1];
            Y2_CHECK_ARG(a >= 0 && b >= a);
            if (b > a) { zr.a[zr.n] = a; zr.b[zr.n] = b; ++zr.n; total += b - a; }
        }
        if (zr.n) zero_ranges_kernel<<<ew_grid(total / 4 + 1), 256, 0, (hipStream_t)stream>>>(x, zr);
    }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_bn_fold(const float *W, const float *gamma, const float *beta, const float *moving_mean, const float *moving_var, float *Wf,
                             float *bias, long rows, int C, float eps, void *stream) {
    Y2_CHECK_ARG(W && gamma && beta && moving_mean && moving_var && Wf && bias && rows > 0 && C > 0);
    bn_fold_kernel<<<ew_grid(rows * C), 256, 0, (hipStream_t)stream>>>(W, gamma, beta, moving_mean, moving_var, Wf, bias, rows, C, eps);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// empty kernel: bench.py calibrates what a HIP-event bracket adds to the kernel it brackets (dispatch latency between the start
// event's completion and the kernel's first wave) by bracketing this
__global__ void noop_kernel() {}
extern "C" int yolo2_debug_noop(void *stream) {
    noop_kernel<<<1, 64, 0, (hipStream_t)stream>>>();
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- workspace sizes (bytes) of the entries that take a caller-owned scratch buffer: the single source of truth for callers
extern "C" size_t yolo2_bn_workspace_bytes(int C) { return (size_t)1025 * (size_t)(C > 0 ? C : 0) * sizeof(double); }
extern "C" size_t yolo2_bias_grad_workspace_bytes(int ld) { return (size_t)512 * (size_t)(ld > 0 ? ld : 0) * sizeof(double); }
extern "C" size_t yolo2_image_prep_workspace_bytes(int B) { return (size_t)2 * Y2_IMG_PARTS * (size_t)(B > 0 ? B : 0) * sizeof(double); }
extern "C" size_t yolo2_clip_workspace_bytes(int nseg) { return (size_t)(nseg > 0 ? nseg : 0) * sizeof(double); }
extern "C" size_t yolo2_augment_workspace_bytes(int B) { return (size_t)3 * (size_t)(B > 0 ? B : 0) * sizeof(double); }
extern "C" size_t yolo2_nms_workspace_bytes(int B, int N, int C) { return (size_t)(B > 0 ? B : 0) * (size_t)(N > 0 ? N : 0) * (size_t)(C > 0 ? C : 0) * sizeof(int); }
extern "C" size_t yolo2_loss_workspace_bytes(int B, int cells, int A) {
    int lpc = 1;
    while (lpc < A) lpc *= 2;
    return (size_t)(4 * (((long)B * cells * lpc + 255) / 256) + 4) * sizeof(float);
}

extern "C" int yolo2_adam(float *w, const float *g, float *m, float *v, long n, float alpha, float beta1, float beta2, float eps, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && m && v && n > 0);
    adam_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, m, v, n, alpha, 1.0f - beta1, 1.0f - beta2, eps, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_momentum(float *w, const float *g, float *acc, long n, float lr, float momentum, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && acc && n > 0);
    momentum_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, acc, n, lr, momentum, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_sgd(float *w, const float *g, long n, float lr, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && n > 0);
    sgd_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, n, lr, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_rmsprop(float *w, const float *g, float *ms, float *mom, long n, float lr, float decay, float momentum, float eps, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && ms && mom && n > 0);
    rmsprop_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, ms, mom, n, lr, 1.0f - decay, momentum, eps, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_adagrad(float *w, const float *g, float *acc, long n, float lr, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && acc && n > 0);
    adagrad_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, acc, n, lr, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_adadelta(float *w, const float *g, float *acc, float *acc_update, long n, float lr, float rho, float eps, float gscale, void *stream) {
    Y2_CHECK_ARG(w && g && acc && acc_update && n > 0);
    adadelta_kernel<<<ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(w, g, acc, acc_update, n, lr, rho, eps, gscale);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// per-tensor clip_by_norm: one block row per segment (grid.y = segment), two passes
__global__ void seg_sumsq_kernel(const float *__restrict__ g, const long *__restrict__ seg_off, double *__restrict__ ws) {
    const int s = blockIdx.y;
    const long beg = seg_off[s], end = seg_off[s + 1];
    double acc = 0.0;
    for (long i = beg + blockIdx.x * (long)blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x) {
        double v = (double)g[i];
        acc += v * v;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(ws + s, acc);
}
__global__ void seg_scale_kernel(float *__restrict__ g, const long *__restrict__ seg_off, const double *__restrict__ ws, float clip) {
    const int s = blockIdx.y;
    const long beg = seg_off[s], end = seg_off[s + 1];
    const float norm = (float)sqrt(ws[s]);
    const float scale = clip / fmaxf(norm, clip);
    if (scale == 1.0f) return;
    for (long i = beg + blockIdx.x * (long)blockDim.x + threadIdx.x; i < end; i += (long)gridDim.x * blockDim.x) g[i] = g[i] * scale;
}
extern "C" int yolo2_clip_by_norm(float *g, const long *seg_off, int nseg, float clip, double *ws, void *stream) {
    Y2_CHECK_ARG(g && seg_off && ws && nseg > 0 && clip > 0.f);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double) * nseg, st) != hipSuccess) { yolo2_set_error("clip_by_norm: memset failed"); return YOLO2_E_LAUNCH; }
    dim3 grid(64, nseg);
    seg_sumsq_kernel<<<grid, 256, 0, st>>>(g, seg_off, ws);
    seg_scale_kernel<<<grid, 256, 0, st>>>(g, seg_off, ws, clip);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ------------------------------------------------------------------------------------------
// layout self-test for ds_read_b64_tr_b16 (used once on hardware to confirm the gather the
// filter-gradient kernel assumes)
// ------------------------------------------------------------------------------------------
__global__ void selftest_tr16_kernel(short *out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + threadIdx.x * 4));
    out[threadIdx.x * 4 + 0] = v[0];
    out[threadIdx.x * 4 + 1] = v[1];
    out[threadIdx.x * 4 + 2] = v[2];
    out[threadIdx.x * 4 + 3] = v[3];
}
extern "C" int yolo2_selftest_tr16(short *out, void *stream) {
    Y2_CHECK_ARG(out);
    selftest_tr16_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- fused BN + leaky + max pool entry points (kernels above the max-pool section)
static bool pool_args_ok(int B, int H, int W, int C, int dtype) {
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    return B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % vec == 0 && C / vec <= 256;
}
extern "C" int yolo2_bn_leaky_pool(const void *Y, const float *mean, const float *var, const float *gamma, const float *beta, void *P,
                                   unsigned char *idx, int B, int H, int W, int C, int ldp, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(Y && mean && var && gamma && beta && P && ldp >= C);
    Y2_CHECK_ARG(pool_args_ok(B, H, W, C, dtype) && ldp % (dtype == YOLO2_BF16 ? 8 : 4) == 0);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const long MP = (long)B * (H / 2) * (W / 2);
    int grid = rowmap_grid(MP, C, vec, 2);
    Y2_DISPATCH_DTYPE(dtype, bn_leaky_pool_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)Y, mean, var, gamma, beta, (T *)P, idx, B, H, W, C, ldp, eps, alpha));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
static int pool_bwd_reduce_impl(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                const float *gamma, const float *beta, float *dgamma, float *dbeta, double *ws, int *rows, int rows_limit, int B, int H, int W,
                                int C, float eps, float alpha, int dtype, void *stream);
extern "C" int yolo2_bn_leaky_pool_bwd_reduce(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                              const float *gamma, const float *beta, float *dgamma, float *dbeta, double *ws, int B, int H, int W,
                                              int C, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(dgamma && dbeta);
    return pool_bwd_reduce_impl(dP, lddp, idx, Y, mean, var, gamma, beta, dgamma, dbeta, ws, nullptr, 1024, B, H, W, C, eps, alpha, dtype, stream);
}
// reduction alone: partial rows [2][*rows][C] stay in ws (for yolo2_bn_leaky_pool_bwd_apply_fin); at most rows_limit of them
extern "C" int yolo2_bn_leaky_pool_bwd_reduce_part(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                                   const float *gamma, const float *beta, double *ws, int *rows, int rows_limit, int B, int H, int W, int C,
                                                   float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(rows && rows_limit >= 1);
    return pool_bwd_reduce_impl(dP, lddp, idx, Y, mean, var, gamma, beta, nullptr, nullptr, ws, rows, rows_limit, B, H, W, C, eps, alpha, dtype, stream);
}
static int pool_bwd_reduce_impl(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                const float *gamma, const float *beta, float *dgamma, float *dbeta, double *ws, int *rows, int rows_limit, int B, int H, int W,
                                int C, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(dP && idx && Y && mean && var && gamma && beta && ws && lddp >= C);
    Y2_CHECK_ARG(pool_args_ok(B, H, W, C, dtype));
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    hipStream_t st = (hipStream_t)stream;
    const long MP = (long)B * (H / 2) * (W / 2);
    int nb = colsum_grid(MP, C, vec);
    // the 416x416 / 208x208 stages (> 64 MB of conv output): one workgroup per CU is latency-bound at 3 TB/s (measured 80 -> 62 us
    // with four); smaller tensors keep the short finalisation
    const int big = 1024;
    if (big > nb && (long)B * H * W * C * (16 / vec) >= (64L << 20)) {
        const int tpr = C / vec, rpp = 256 / tpr < 1 ? 1 : 256 / tpr;
        long g = (MP + (long)rpp * 4 - 1) / ((long)rpp * 4);
        nb = (int)(g < big ? g : big);
        if (nb > 1024) nb = 1024;
    }
    if (nb > rows_limit) nb = rows_limit;
    float *part = (float *)ws;
    Y2_DISPATCH_DTYPE(dtype, bn_pool_bwd_reduce_kernel<T><<<nb, 256, 0, st>>>((const T *)dP, lddp, idx, (const T *)Y, mean, var, gamma, beta, part, B, H, W, C, eps, alpha));
    if (rows) *rows = nb;
    else reduce_finalize_kernel<1><<<cdiv(C, 16), 256, 0, st>>>(part, nb, C, (long)B * H * W, dgamma, dbeta, C);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_bn_leaky_pool_bwd_apply(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                             const float *gamma, const float *beta, const float *dgamma, const float *dbeta, void *dY, int B, int H,
                                             int W, int C, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(dP && idx && Y && mean && var && gamma && beta && dgamma && dbeta && dY && lddp >= C);
    Y2_CHECK_ARG(pool_args_ok(B, H, W, C, dtype));
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const long MP = (long)B * (H / 2) * (W / 2);
    int grid = rowmap_grid(MP, C, vec, 2);
    Y2_DISPATCH_DTYPE(dtype, bn_pool_bwd_apply_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)dP, lddp, idx, (const T *)Y, mean, var, gamma, beta, dgamma, dbeta, (T *)dY, B, H, W, C, eps, alpha));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- consumers with the finalisation in their prologue (kernels: bn_leaky_fin_kernel, bn_bwd_apply_fin_kernel)
static bool fin_shape_ok(int rows, int C, int dtype) {
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    if (rows < 1 || C < vec || C % vec) return false;
    const int tpr = C / vec;
    if (tpr & (tpr - 1)) return false;                       // lanes per row must divide 256
    const int lpr = tpr < 16 ? tpr : 16;
    return (long)rows * lpr * vec * 8 <= (128L << 10);      // the prologue of EVERY workgroup reads this much: beyond it a separate finalisation is cheaper
}
extern "C" int yolo2_bn_fin_supported(int rows, int C, int dtype) { return fin_shape_ok(rows, C, dtype) ? 1 : 0; }

static dim3 slice_grid(long loop_rows, int C, int vec, int rows_per_thread, int part_rows) {
    const int tpr = C / vec, lpr = tpr < 16 ? tpr : 16, rpb = 256 / lpr, slices = tpr / lpr;
    long gx = (loop_rows + (long)rpb * rows_per_thread - 1) / ((long)rpb * rows_per_thread);
    const long per = (long)part_rows * lpr * vec * 8;        // prologue bytes per workgroup
    long cap = (48L << 20) / (per * slices);                 // <= ~48 MB of L2 reads for all prologues together ...
    const long floor_ = (512 + slices - 1) / slices;         // ... but never fewer than two workgroups per CU
    if (cap < floor_) cap = floor_;
    if (cap > 4096 / slices) cap = 4096 / slices;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)slices);
}
#define Y2_CHECK_ZERO(zero, zero_floats) Y2_CHECK_ARG((zero_floats) >= 0 && (zero_floats) % 4 == 0 && ((zero) || (zero_floats) == 0) && ((uintptr_t)(zero) & 15) == 0)

extern "C" int yolo2_bn_leaky_fin(const void *Y, const float *bn_part, int rows, const float *shift, float *mean, float *var, float *moving_mean,
                                  float *moving_var, double decay, const float *gamma, const float *beta, void *A, long M, int C, int lda, float eps,
                                  float alpha, float *zero, long zero_floats, int dtype, void *stream) {
    Y2_CHECK_ARG(Y && bn_part && shift && mean && var && gamma && beta && A && M > 0 && C > 0 && lda >= C && M < (1L << 31));
    Y2_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr) && rows <= Y2_BN_PART_ROWS && fin_shape_ok(rows, C, dtype));
    Y2_CHECK_ARG(shift != moving_mean && shift != mean);      // every workgroup reads the shift; one per channel slice writes these (see the header)
    Y2_CHECK_ZERO(zero, zero_floats);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(lda % vec == 0);
    const dim3 grid = slice_grid(M, C, vec, 4, rows);
    Y2_DISPATCH_DTYPE(dtype, bn_leaky_fin_kernel<T, false><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)Y, bn_part, rows, shift, M, mean, var, moving_mean, moving_var,
                      (float)(1.0 - decay), gamma, beta, (T *)A, nullptr, nullptr, nullptr, 1, 1, (int)M, C, lda, eps, alpha, zero, zero_floats / 4));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

extern "C" int yolo2_bn_leaky_pool_fin(const void *Y, const float *bn_part, int rows, const float *shift, float *mean, float *var, float *moving_mean,
                                       float *moving_var, double decay, const float *gamma, const float *beta, void *P, unsigned char *idx, void *ymax,
                                       void *A_full, int B, int H, int W, int C, int ldp, float eps, float alpha, float *zero, long zero_floats, int dtype,
                                       void *stream) {
    Y2_CHECK_ARG(Y && bn_part && shift && mean && var && gamma && beta && P && ldp >= C);
    Y2_CHECK_ARG(pool_args_ok(B, H, W, C, dtype) && ldp % (dtype == YOLO2_BF16 ? 8 : 4) == 0);
    Y2_CHECK_ARG((moving_mean == nullptr) == (moving_var == nullptr) && rows <= Y2_BN_PART_ROWS && fin_shape_ok(rows, C, dtype));
    Y2_CHECK_ARG(shift != moving_mean && shift != mean);
    Y2_CHECK_ZERO(zero, zero_floats);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const dim3 grid = slice_grid((long)B * (H / 2) * (W / 2), C, vec, 2, rows);
    Y2_DISPATCH_DTYPE(dtype, bn_leaky_fin_kernel<T, true><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)Y, bn_part, rows, shift, (long)B * H * W, mean, var, moving_mean,
                      moving_var, (float)(1.0 - decay), gamma, beta, (T *)P, idx, (T *)ymax, (T *)A_full, B, H, W, C, ldp, eps, alpha, zero, zero_floats / 4));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

extern "C" int yolo2_bn_leaky_bwd_apply_fin(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma, const float *beta,
                                            const float *part, int rows, long plane_stride, float *dgamma, float *dbeta, void *dY, long M, int C, float eps,
                                            float alpha, float *zero, long zero_floats, int dtype, void *stream) {
    Y2_CHECK_ARG(dA && Y && mean && var && gamma && beta && part && dgamma && dbeta && dY && M > 0 && C > 0 && ldda >= C && M < (1L << 31));
    Y2_CHECK_ARG(plane_stride >= (long)rows * C && fin_shape_ok(rows, C, dtype));
    Y2_CHECK_ZERO(zero, zero_floats);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(ldda % vec == 0);
    const dim3 grid = slice_grid(M, C, vec, 4, rows);
    Y2_DISPATCH_DTYPE(dtype, bn_bwd_apply_fin_kernel<T, false><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)dA, ldda, nullptr, (const T *)Y, mean, var, gamma, beta, part, rows,
                      plane_stride, dgamma, dbeta, (T *)dY, 1, 1, (int)M, C, eps, alpha, zero, zero_floats / 4));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

extern "C" int yolo2_bn_leaky_pool_bwd_apply_fin(const void *dP, int lddp, const unsigned char *idx, const void *Y, const float *mean, const float *var,
                                                 const float *gamma, const float *beta, const float *part, int rows, long plane_stride, float *dgamma,
                                                 float *dbeta, void *dY, int B, int H, int W, int C, float eps, float alpha, float *zero, long zero_floats,
                                                 int dtype, void *stream) {
    Y2_CHECK_ARG(dP && idx && Y && mean && var && gamma && beta && part && dgamma && dbeta && dY && lddp >= C);
    Y2_CHECK_ARG(pool_args_ok(B, H, W, C, dtype) && plane_stride >= (long)rows * C && fin_shape_ok(rows, C, dtype));
    Y2_CHECK_ZERO(zero, zero_floats);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    const dim3 grid = slice_grid((long)B * (H / 2) * (W / 2), C, vec, 2, rows);
    Y2_DISPATCH_DTYPE(dtype, bn_bwd_apply_fin_kernel<T, true><<<grid, 256, 0, (hipStream_t)stream>>>((const T *)dP, lddp, idx, (const T *)Y, mean, var, gamma, beta, part, rows,
                      plane_stride, dgamma, dbeta, (T *)dY, B, H, W, C, eps, alpha, zero, zero_floats / 4));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// the reduction halves of yolo2_bn_leaky_bwd_reduce / yolo2_bn_leaky_pool_bwd_reduce alone: partial rows [2][*rows][C] left in ws for a *_fin consumer
extern "C" int yolo2_bn_leaky_bwd_reduce_part(const void *dA, int ldda, const void *Y, const float *mean, const float *var, const float *gamma,
                                              const float *beta, double *ws, int *rows, int rows_limit, long M, int C, float eps, float alpha, int dtype,
                                              void *stream) {
    Y2_CHECK_ARG(dA && Y && mean && var && gamma && beta && ws && rows && rows_limit >= 1 && M > 0 && C > 0 && ldda >= C);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(C % vec == 0 && C / vec <= 256 && ldda % vec == 0);
    int nb = colsum_grid(M, C, vec);
    // > 32 MB to read (the pooled 208x208 / 104x104 stages): one workgroup per CU is latency-bound, as for the pooled reduction above
    const int big = 1024;
    if (big > nb && M * C * (16 / vec) * 2 >= (32L << 20)) {
        const int tpr = C / vec, rpp = 256 / tpr < 1 ? 1 : 256 / tpr;
        const long g = (M + (long)rpp * 16 - 1) / ((long)rpp * 16);
        nb = (int)(g < big ? g : big);
        if (nb > 1024) nb = 1024;
    }
    if (nb > rows_limit) nb = rows_limit;
    Y2_DISPATCH_DTYPE(dtype, bn_bwd_reduce_kernel<T><<<nb, 256, 0, (hipStream_t)stream>>>((const T *)dA, ldda, (const T *)Y, mean, var, gamma, beta, (float *)ws, M, C, eps, alpha));
    Y2_CHECK_LAUNCH();
    *rows = nb;
    return YOLO2_OK;
}

// ---- gradient wire format of the data-parallel exchange (parallel.GradReducer, grad_dtype = bf16): the f32 gradient bucket is rounded
// to bf16 into a wire buffer, all-reduced there (half the xGMI bytes: 134 MB instead of 269 MB per step and rank), and widened back into
// the f32 arena the optimizer reads.  16 bytes per lane on the wide side.
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float *__restrict__ src, bf16 *__restrict__ dst, long n) {
    const long nv = n >> 3, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4 *>(src)[2 * i], b = reinterpret_cast<const f32x4 *>(src)[2 * i + 1];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = (bf16)a[j]; o[4 + j] = (bf16)b[j]; }
        reinterpret_cast<bf16x8 *>(dst)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nv << 3) + threadIdx.x] = (bf16)src[(nv << 3) + threadIdx.x];
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16 *__restrict__ src, float *__restrict__ dst, long n) {
    const long nv = n >> 3, stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        const bf16x8 v = reinterpret_cast<const bf16x8 *>(src)[i];
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = (float)v[j]; b[j] = (float)v[4 + j]; }
        reinterpret_cast<f32x4 *>(dst)[2 * i] = a;
        reinterpret_cast<f32x4 *>(dst)[2 * i + 1] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nv << 3) + threadIdx.x] = (float)src[(nv << 3) + threadIdx.x];
}
extern "C" int yolo2_cast_f32_bf16(const float *src, void *dst, long n, void *stream) {
    Y2_CHECK_ARG(src && dst && n >= 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    if (n == 0) return YOLO2_OK;
    cast_f32_bf16_kernel<<<ew_grid((n + 7) / 8), 256, 0, (hipStream_t)stream>>>(src, (bf16 *)dst, n);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
extern "C" int yolo2_cast_bf16_f32(const void *src, float *dst, long n, void *stream) {
    Y2_CHECK_ARG(src && dst && n >= 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
    if (n == 0) return YOLO2_OK;
    cast_bf16_f32_kernel<<<ew_grid((n + 7) / 8), 256, 0, (hipStream_t)stream>>>((const bf16 *)src, dst, n);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- test instrument: what an RCCL ring looks like to the dispatcher.  `workgroups` persistent 256-thread workgroups, each holding a
// whole CU (all 160 KiB of LDS), spin until *stop becomes non-zero or `max_us` microseconds have passed (bounded: a test can never hang
// the GPU on it).  *started counts the workgroups that are resident.  tests/test_streamk_occupied_gpu.py runs the stream-K convolutions
// beside it.
__global__ __launch_bounds__(256) void occupy_kernel(volatile int *stop, int *started, long max_ticks) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        __hip_atomic_fetch_add(started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long t0 = wall_clock64();
        while (__hip_atomic_load(const_cast<int *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
}
extern "C" int yolo2_debug_occupy(int workgroups, int *stop, int *started, int max_us, void *stream) {
    Y2_CHECK_ARG(workgroups > 0 && workgroups <= 256 && stop && started && max_us > 0 && max_us <= 2000000);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            yolo2_set_error("yolo2_debug_occupy: cannot raise the dynamic LDS limit");
            return YOLO2_E_LAUNCH;
        }
        attr_set = true;
    }
    occupy_kernel<<<workgroups, 256, 160 * 1024, (hipStream_t)stream>>>(stop, started, (long)max_us * 100);      // wall_clock64 ticks at 100 MHz
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---- Adam + operand layouts in one pass (round 3).  The Adam kernel streams every weight through registers anyway; the operand
// re-layout (filter_prep_batch_kernel: 268 MB re-read of the f32 masters + one more launch per step) rides on it: a workgroup takes one
// 64 x 64 (c, n) tile of one tap of one layer -- the re-layout's own decomposition, which covers every filter element exactly once --
// updates w / m / v in place (16-byte accesses, the same arithmetic as adam_kernel) and emits both bf16 / f32 operand layouts from the
// fresh values staged in LDS.  The parameters that are not convolution filters (gamma, beta, biases: ~22 k floats in 43 ranges) are
// updated by the extra workgroups at the end of the grid.  Results are bit-identical to yolo2_adam followed by yolo2_filter_prep_batch.
struct Y2AdamArgs { float *params; const float *grads; float *m, *v; float alpha, omb1, omb2, eps, gs; };
template <typename T>
__global__ __launch_bounds__(256) void adam_filter_prep_kernel(const yolo2_filter_desc *__restrict__ descs, int n, int conv_blocks,
                                                               const long *__restrict__ small, const Y2AdamArgs a) {
    if ((int)blockIdx.x >= conv_blocks) {      // a non-filter parameter range [small[2i], small[2i] + small[2i+1])
        const long *r = small + 2 * ((int)blockIdx.x - conv_blocks);
        for (long k = threadIdx.x; k < r[1]; k += 256) {
            const long o = r[0] + k;
            adam_one(a.params[o], a.grads[o], a.m[o], a.v[o], a.alpha, a.omb1, a.omb2, a.eps, a.gs);
        }
        return;
    }
    constexpr int TC = YOLO2_FILTER_PREP_TILE, TN = YOLO2_FILTER_PREP_TILE_N;
    __shared__ float tile[TC][TN + 1];
    int li = 0;
    while (li + 1 < n && (int)blockIdx.x >= descs[li + 1].first_block) ++li;
    const yolo2_filter_desc d = descs[li];
    const int taps = d.ksize * d.ksize;
    const int ctiles = (d.ldcin + TC - 1) / TC, ntiles = (d.ldcout + TN - 1) / TN;
    int u = blockIdx.x - d.first_block;
    const int ntile = u % ntiles; u /= ntiles;
    const int ctile = u % ctiles;
    const int tap = u / ctiles;
    const int n0 = ntile * TN, c0 = ctile * TC;
    const int tid = threadIdx.x;
    const long base = (d.W - a.params) + (long)tap * d.cin * d.cout;       // element offset of this tap's [cin][cout] plane in the arenas
    const bool vec_ok = (d.cout & 3) == 0 && ((base & 3) == 0);
    {
        // TN / 4 lanes x float4 per row: a 128-wide tile reads and writes w / m / v / g in 512-byte runs (round 6: with 64 x 64 tiles the four streams
        // moved in 256-byte runs at 5.5-5.7 TB/s where the linear adam_kernel reaches 6.9)
        constexpr int LPR = TN / 4, RPP = 256 / LPR;
        const int col = (tid % LPR) * 4, r0 = tid / LPR;
#pragma unroll
        for (int p = 0; p < TC / RPP; ++p) {
            const int c = c0 + r0 + p * RPP, nn = n0 + col;
            float w4[4] = {0.f, 0.f, 0.f, 0.f};
            if (c < d.cin) {
                const long o = base + (long)c * d.cout + nn;
                if (vec_ok && nn + 3 < d.cout) {
                    f32x4 wv = *reinterpret_cast<const f32x4 *>(a.params + o), mv = *reinterpret_cast<const f32x4 *>(a.m + o), vv = *reinterpret_cast<const f32x4 *>(a.v + o);
                    const f32x4 gv = *reinterpret_cast<const f32x4 *>(a.grads + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float wj = wv[j], mj = mv[j], vj = vv[j];
                        adam_one(wj, gv[j], mj, vj, a.alpha, a.omb1, a.omb2, a.eps, a.gs);
                        wv[j] = wj; mv[j] = mj; vv[j] = vj; w4[j] = wj;
                    }
                    *reinterpret_cast<f32x4 *>(a.m + o) = mv;
                    *reinterpret_cast<f32x4 *>(a.v + o) = vv;
                    *reinterpret_cast<f32x4 *>(a.params + o) = wv;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nn + j < d.cout) {
                            adam_one(a.params[o + j], a.grads[o + j], a.m[o + j], a.v[o + j], a.alpha, a.omb1, a.omb2, a.eps, a.gs);
                            w4[j] = a.params[o + j];
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[r0 + p * RPP][col + j] = w4[j];
        }
    }
    __syncthreads();
    filter_tile_emit<T>(d, tile, tap, taps, c0, n0, tid);
}
extern "C" int yolo2_adam_filter_prep(const yolo2_filter_desc *descs_device, int n, int total_blocks, const long *small_ranges_device, int n_small,
                                      float *params, const float *grads, float *m, float *v, float alpha, float beta1, float beta2, float eps,
                                      float gscale, int dtype, void *stream) {
    // (n == 0: only the non-filter ranges -- the last launch of a step whose filters were updated layer by layer during backward)
    Y2_CHECK_ARG(n >= 0 && total_blocks >= 0 && (n > 0) == (total_blocks > 0) && (descs_device || n == 0) && n_small >= 0 && total_blocks + n_small > 0 &&
                 (small_ranges_device || n_small == 0) && params && grads && m && v);
    Y2_CHECK_ARG(((((uintptr_t)params) | ((uintptr_t)grads) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0);
    const Y2AdamArgs a{params, grads, m, v, alpha, 1.0f - beta1, 1.0f - beta2, eps, gscale};
    Y2_DISPATCH_DTYPE(dtype, adam_filter_prep_kernel<T><<<total_blocks + n_small, 256, 0, (hipStream_t)stream>>>(descs_device, n, total_blocks, small_ranges_device, a));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
