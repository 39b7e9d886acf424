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

template <typename T, int BN, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(
    const T *__restrict__ P, const T *__restrict__ F, const float *__restrict__ bias,
    T *__restrict__ O, int H, int W, int Cp, int ldp, int Nf, int ldo, int ksize, int M, int NT) {
    constexpr int BM = 128;
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BK = 4 * VEC;
    constexpr int LDS = BK + VEC;  // row stride in elements (80 bytes)
    constexpr int WGM = 4 / WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int A_IT = BM * 4 / 256;
    constexpr int B_IT = (BN * 4 + 255) / 256;
    static_assert(TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) T smem[2][(BM + BN) * LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nt = blockIdx.x % NT, mt = blockIdx.x / NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int chunk = tid & 3;
    const int Ktot = ksize * ksize * Cp;
    const int pad = ksize >> 1;

    int a_h[A_IT], a_w[A_IT];
    long a_off[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int m = m0 + (tid >> 2) + i * 64;
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
        int row = (tid >> 2) + i * 64;
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
        for (int i = 0; i < A_IT; ++i) st16(As + ((tid >> 2) + i * 64) * LDS + chunk * VEC, ra[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = (tid >> 2) + i * 64;
            if (row < BN) st16(Bs + row * LDS + chunk * VEC, rb[i]);
        }
    };

    int tap = 0, c0 = 0;
    g_load(tap, c0);
    s_store(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
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
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < M) O[(long)m * ldo + n] = (T)(acc[i][j][r] + bv);
            }
        }
    }
}

template <typename T>
static int launch_conv(const void *P, const void *F, const float *bias, void *O, int B, int H, int W,
                       int Cp, int ldp, int Nf, int ldo, int ksize, hipStream_t st) {
    const int M = B * H * W;
    const int MT = cdiv(M, 128);
    if (Nf > 64) {
        const int NT = cdiv(Nf, 128);
        conv_igemm_kernel<T, 128, 2><<<MT * NT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, H, W, Cp, ldp, Nf, ldo, ksize, M, NT);
    } else if (Nf > 32) {
        conv_igemm_kernel<T, 64, 1><<<MT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, H, W, Cp, ldp, Nf, ldo, ksize, M, 1);
    } else {
        conv_igemm_kernel<T, 32, 1><<<MT, 256, 0, st>>>((const T *)P, (const T *)F, bias, (T *)O, H, W, Cp, ldp, Nf, ldo, ksize, M, 1);
    }
    return 0;
}

extern "C" int yolo2_conv2d(const void *P, const void *F, const float *bias, void *O, int B, int H,
                            int W, int Cp, int ldp, int Nf, int ldo, int ksize, int dtype, void *stream) {
    Y2_CHECK_ARG(P && F && O);
    Y2_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cp > 0 && Nf > 0);
    Y2_CHECK_ARG(ksize == 1 || ksize == 3);
    Y2_CHECK_ARG(ldp >= Cp && ldo >= Nf);
    Y2_CHECK_ARG((long)B * H * W * (long)(ldp > ldo ? ldp : ldo) < (1L << 31));
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(Cp % vec == 0 && ldp % vec == 0);
    Y2_CHECK_ARG(((uintptr_t)P & 15) == 0 && ((uintptr_t)F & 15) == 0);
    Y2_DISPATCH_DTYPE(dtype, launch_conv<T>(P, F, bias, O, B, H, W, Cp, ldp, Nf, ldo, ksize, (hipStream_t)stream));
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
