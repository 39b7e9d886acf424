// NHWC implicit-GEMM convolution on MFMA (gfx950), forward and data-gradient.
//
// Replaces slim.layers.conv2d (reference model/yolo2/inference.py:37-48,73-118) and its
// tf.gradients input-gradient (train.py:127-129).
//
//   O[m, n] = bias[n] + sum_{tap, c} P[pix(m) + shift(tap), c] * F[n][tap*Cp + c]
//   GEMM view: M = B*H*W output pixels, N = Nf filters, K = ksize^2 * Cp.
//
// Tiling: 256 threads = 4 waves, block tile 128(M) x BN(N), K step = 64 bytes of channels of one
// tap (32 bf16 / 16 f32).  Both operands are staged global -> registers -> LDS (rows of 64 B +
// 16 B pad = 80 B: conflict-free ds_read_b128 for the 32x32 MFMA fragments) with a 2-deep LDS
// ring: the global loads of tile t+1 are issued before the MFMAs of tile t and written to the
// other LDS buffer after them, one barrier per K step.  Halo / image-border / channel-tail /
// M-tail elements are zero-filled in registers, so SAME padding costs nothing extra.
// A = pixels (rows), B = filters (cols): the 32x32 accumulator layout then puts 32 consecutive
// output channels of one pixel in 32 consecutive lanes -> contiguous 64 B (bf16) / 128 B (f32)
// store segments.
// blockIdx -> tile: filter tile fastest, so the blocks an XCD receives (bid % 8) keep re-using
// the same filter slab from that XCD's private L2 while sweeping M.
#include "common.h"
#include <stdlib.h>

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static constexpr int KSTEP = 16;
    typedef bf16x8 Frag;
    static __device__ __forceinline__ Frag load(const bf16 *row, int kk, int lane) {
        return *reinterpret_cast<const bf16x8 *>(row + kk * 16 + (lane >> 5) * 8);
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 2;
    typedef float Frag;
    static __device__ __forceinline__ Frag load(const float *row, int kk, int lane) {
        return row[kk * 2 + (lane >> 5)];
    }
    static __device__ __forceinline__ f32x16 mma(Frag a, Frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
};

// CH = 16-byte chunks per LDS row (K step = CH*16 bytes of channels); SPLITK: gridDim.y slices of the
// K loop accumulate f32 partial tiles into Oacc with hardware atomics (finished by splitk_finish_kernel)
template <typename T, int BN, int WGN, int CH, bool SPLITK>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const T *__restrict__ P, const T *__restrict__ F, const float *__restrict__ bias,
    T *__restrict__ O, float *__restrict__ Oacc, int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize, int M, int NT) {
    constexpr int BM = 128;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = CH * VEC;
    constexpr int LDS = BK + VEC;  // row stride: CH*16 + 16 bytes (80 / 144 B: conflict-free b128 fragment reads)
    constexpr int WGM = 4 / WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int RPP = 256 / CH;                 // rows covered per pass of the 256 threads
    constexpr int A_IT = BM / RPP;
    constexpr int B_IT = (BN + RPP - 1) / RPP;
    static_assert(TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) T smem[2][(BM + BN) * LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nt = blockIdx.x % NT, mt = blockIdx.x / NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int chunk = tid % CH, trow = tid / CH;
    const int Ktot = ksize * ksize * Cp;
    const int pad = ksize >> 1;

    int a_h[A_IT], a_w[A_IT];
    long a_off[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + trow + i * RPP;
        if (m < M) {
            int rem = m % (H * W);
            a_h[i] = rem / W;
            a_w[i] = rem - a_h[i] * W;
        } else {
            a_h[i] = -100000;
            a_w[i] = 0;
        }
        a_off[i] = (long)m * ldp;
    }
    long b_off[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int row = trow + i * RPP;
        int n = n0 + row;
        b_ok[i] = (row < BN) && (n < Nf);
        b_off[i] = (long)n * Ktot;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kpt = (Cp + BK - 1) / BK;  // K tiles per tap
    const int nk = ksize * ksize * kpt;

    Vec16<T> ra[A_IT], rb[B_IT];
    auto g_load = [&](int tap, int c0) {
        const int dh = tap / ksize - pad, dw = tap % ksize - pad;
        const int c = c0 + chunk * VEC;
        const bool cok = c < Cp;
        const long shift = (long)(dh * W + dw) * ldp + c;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int hh = a_h[i] + dh, ww = a_w[i] + dw;
            bool ok = cok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            ra[i] = ok ? ld16(P + a_off[i] + shift) : zero16<T>();
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            bool ok = cok && b_ok[i];
            rb[i] = ok ? ld16(F + b_off[i] + (long)tap * Cp + c) : zero16<T>();
        }
    };
    auto s_store = [&](int buf) {
        T *As = smem[buf];
        T *Bs = smem[buf] + BM * LDS;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st16(As + (trow + i * RPP) * LDS + chunk * VEC, ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = trow + i * RPP;
            if (row < BN) st16(Bs + row * LDS + chunk * VEC, rb[i]);
        }
    };

    int kt_beg = 0, kt_end = nk;
    if (SPLITK) {
        const int per = (nk + gridDim.y - 1) / gridDim.y;
        kt_beg = blockIdx.y * per;
        kt_end = min(nk, kt_beg + per);
        if (kt_beg >= kt_end) return;
    }
    int tap = kt_beg / kpt, c0 = (kt_beg - tap * kpt) * BK;
    g_load(tap, c0);
    s_store(0);
    __syncthreads();

    for (int kt = kt_beg; kt < kt_end; ++kt) {
        const int cur = (kt - kt_beg) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) {
            c0 += BK;
            if (c0 >= Cp) { c0 = 0; ++tap; }
            g_load(tap, c0);
        }
        const T *As = smem[cur] + (wm * TM * 32 + (lane & 31)) * LDS;
        const T *Bs = smem[cur] + (BM + wn * TN * 32 + (lane & 31)) * LDS;
#pragma unroll
        for (int kk = 0; kk < BK / Mma<T>::KSTEP; ++kk) {
            typename Mma<T>::Frag af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = Mma<T>::load(As + i * 32 * LDS, kk, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = Mma<T>::load(Bs + j * 32 * LDS, kk, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::mma(af[i], bf[j], acc[i][j]);
        }
        if (more) s_store(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (n >= Nf) continue;
        const float bv = (!SPLITK && bias) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < M) {
                    if (SPLITK) unsafeAtomicAdd(Oacc + (long)m * Nf + n, acc[i][j][r]);
                    else O[(long)m * ldo + n] = (T)(acc[i][j][r] + bv);
                }
            }
        }
    }
}

// f32 partial sums [M][Nf] -> O (dtype, pixel stride ldo) + bias
template <typename T>
__global__ void splitk_finish_kernel(const float *__restrict__ acc, const float *__restrict__ bias, T *__restrict__ O, long M, int Nf, int ldo) {
    const long total = M * Nf;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Nf;
        const int n = (int)(i - m * Nf);
        O[m * ldo + n] = (T)(acc[i] + (bias ? bias[n] : 0.f));
    }
}

// number of K slices: aim at >= 3 resident blocks per CU (register-limited occupancy of the 128-wide
// tile) when the M x N tile grid alone cannot fill 256 CUs, keeping >= 8 K tiles per slice
struct Tune { int ch; int target_blocks; };
static const Tune &tune() {   // tuning knobs (defaults = measured best); env overrides are for A/B runs only
    static Tune t = [] {
        Tune v{8, 512};
        if (const char *e = getenv("YOLO2_IGEMM_CH")) v.ch = atoi(e) == 4 ? 4 : 8;
        if (const char *e = getenv("YOLO2_KSPLIT_BLOCKS")) v.target_blocks = atoi(e);
        return v;
    }();
    return t;
}
static int choose_ksplit(int tiles, int nk, int target) {
    if (target <= 0 || tiles * 3 >= target * 2 || nk < 32) return 1;
    int ks = (target + tiles - 1) / tiles;
    int max_ks = nk / 8;
    if (ks > max_ks) ks = max_ks;
    return ks < 1 ? 1 : ks;
}

template <typename T>
static int launch_conv(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B, int H, int W,
                       int Cp, int ldp, int Nf, int ldo, int ksize, hipStream_t st) {
    const int M = B * H * W;
    const int MT = cdiv(M, 128);
    constexpr int VEC = 16 / sizeof(T);
    if (Nf > 64) {
        const int NT = cdiv(Nf, 128);
        const Tune &tu = tune();
        const int CHsel = (Cp >= 8 * VEC) ? tu.ch : 4;
        const int nk = ksize * ksize * cdiv(Cp, CHsel * VEC);
        int ks = ws ? choose_ksplit(MT * NT, nk, tu.target_blocks) : 1;
        if (ks > 1 && (size_t)M * Nf * sizeof(float) > ws_bytes) ks = 1;
        if (ks > 1) {
            if (hipMemsetAsync(ws, 0, (size_t)M * Nf * sizeof(float), st) != hipSuccess) return 1;
            dim3 grid(MT * NT, ks);
            if (CHsel == 8)
                conv_igemm_kernel<T, 128, 2, 8, true><<<grid, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, ws, H, W, Cp, ldp, Nf, ldo, ksize, M, NT);
            else
                conv_igemm_kernel<T, 128, 2, 4, true><<<grid, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, ws, H, W, Cp, ldp, Nf, ldo, ksize, M, NT);
            long total = (long)M * Nf;
            int g = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            splitk_finish_kernel<T><<<g, 256, 0, st>>>(ws, bias, (T *)O, M, Nf, ldo);
        } else if (CHsel == 8) {
            conv_igemm_kernel<T, 128, 2, 8, false><<<MT * NT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, nullptr, H, W, Cp, ldp, Nf, ldo, ksize, M, NT);
        } else {
            conv_igemm_kernel<T, 128, 2, 4, false><<<MT * NT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, nullptr, H, W, Cp, ldp, Nf, ldo, ksize, M, NT);
        }
    } else if (Nf > 32) {
        conv_igemm_kernel<T, 64, 1, 4, false><<<MT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, nullptr, H, W, Cp, ldp, Nf, ldo, ksize, M, 1);
    } else {
        conv_igemm_kernel<T, 32, 1, 4, false><<<MT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, nullptr, H, W, Cp, ldp, Nf, ldo, ksize, M, 1);
    }
    return 0;
}

static int conv2d_impl(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B, int H,
                       int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream, const char *fn) {
    if (!(P && F && O) || !(B > 0 && H > 0 && W > 0 && Cp > 0 && Nf > 0) || !(ksize == 1 || ksize == 3) || !(ldp >= Cp && ldo >= Nf)) {
        yolo2_set_error("%s: argument check failed: pointers / extents / ksize / strides", fn);
        return YOLO2_E_ARG;
    }
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    if ((long)B * H * W * (long)(ldp > ldo ? ldp : ldo) >= (1L << 31) || Cp % vec || ldp % vec || ((uintptr_t)P & 15) || ((uintptr_t)F & 15)) {
        yolo2_set_error("%s: argument check failed: size / alignment (channels must be a multiple of %d)", fn, vec);
        return YOLO2_E_ARG;
    }
    int rc = 0;
    Y2_DISPATCH_DTYPE(dtype, rc = launch_conv<T>(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, (hipStream_t)stream));
    if (rc) { yolo2_set_error("%s: workspace memset failed", fn); return YOLO2_E_LAUNCH; }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

extern "C" int yolo2_conv2d(const void *P, const void *F, const float *bias, void *O, int B, int H,
                            int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream) {
    return conv2d_impl(P, F, bias, O, nullptr, 0, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d");
}

extern "C" int yolo2_conv2d_ws(const void *P, const void *F, const float *bias, void *O, float *ws, size_t ws_bytes, int B,
                               int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream) {
    return conv2d_impl(P, F, bias, O, ws, ws_bytes, B, H, W, Cp, ldp, Nf, ldo, ksize, dtype, stream, "yolo2_conv2d_ws");
}
