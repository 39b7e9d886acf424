// Filter gradient of the NHWC convolution on MFMA (gfx950).
//
// Replaces tf.gradients of slim.layers.conv2d w.r.t. its weights (reference train.py:127-129,
// call sites model/yolo2/inference.py:37-48,73-118):
//   dW[tap][c][n] += sum_{m in pixel range} X[pix(m) + shift(tap), c] * dY[m, n]
//   GEMM view per tap: rows = Cin, cols = Cout, reduction = B*H*W pixels.
//
// The reduction index (pixels) is the strided one in NHWC for BOTH operands, so the MFMA
// fragments need a transpose.  Tiles are staged pixel-major [pixel][channel] (exactly as they
// lie in HBM, 16-B coalesced loads) and the bf16 fragments are gathered with the gfx950
// hardware transpose read ds_read_b64_tr_b16: each 16-lane group reads a 4-pixel x 16-channel
// block and every lane receives 4 pixels of its own channel; two reads give the 8 reduction
// slots of v_mfma_f32_32x32x16_bf16.  Rows are padded by 64 B so the 4 pixel rows of a group
// land on disjoint banks.  The f32 parity path uses v_mfma_f32_32x32x2_f32, whose operands are
// one scalar per lane (plain ds_read_b32, lanes along channels, conflict-free).
// Grid: x = (tap, c-tile, n-tile), y = pixel-range split; partial tiles are accumulated into the
// zero-initialised f32 dW with hardware f32 atomics (lanes run along n -> coalesced).
#include "common.h"

template <typename T, int BC, int BNN, int VARIANT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const T *__restrict__ X, const T *__restrict__ dY, float *__restrict__ dW, int H, int W,
    int Cin, int ldx, int Cout, int ldy, int ksize, int M, int CT, int NT, int mchunk) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int BKP = sizeof(T) == 2 ? 32 : 16;  // pixels per reduction tile
    constexpr int PADE = 64 / sizeof(T);           // 64-byte row pad
    constexpr int LX = BC + PADE, LY = BNN + PADE;
    constexpr int CPRX = BC / VEC, CPRY = BNN / VEC;
    constexpr int RPX = 256 / CPRX, RPY = 256 / CPRY;  // rows covered per pass
    constexpr int XI = BKP / RPX, YI = BKP / RPY;
    static_assert(XI >= 1 && YI >= 1, "tile too narrow for this thread mapping");
    constexpr int TM = BC / 64, TN = BNN / 64;  // 2x2 waves, each (BC/2) x (BNN/2)
    constexpr int KSTEP = sizeof(T) == 2 ? 16 : 2;

    __shared__ __attribute__((aligned(16))) T smem[2][BKP * (LX + LY)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x;
    const int nt = bx % NT; bx /= NT;
    const int ct = bx % CT;
    const int tap = bx / CT;
    const int pad = ksize >> 1;
    const int dh = tap / ksize - pad, dw = tap % ksize - pad;
    const int c0 = ct * BC, n0 = nt * BNN;
    const int mbeg = blockIdx.y * mchunk;
    const int mend = min(M, mbeg + mchunk);
    if (mbeg >= mend) return;

    const int xc = (tid % CPRX) * VEC, xr = tid / CPRX;
    const int yc = (tid % CPRY) * VEC, yr = tid / CPRY;
    const bool xc_ok = c0 + xc < Cin, yc_ok = n0 + yc < Cout;
    // NB lanes beyond Cin/Cout inside the padded pixel stride read zeros (padding contract), but
    // lanes beyond ldx/ldy must not be touched:
    const bool xc_in = c0 + xc < ldx, yc_in = n0 + yc < ldy;

    int xh[XI], xw[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        int m = mbeg + xr + i * RPX;
        int rem = m % (H * W);
        xh[i] = rem / W;
        xw[i] = rem - xh[i] * W;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Vec16<T> rx[XI], ry[YI];
    auto g_load = [&](int mt0) {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int m = mt0 + xr + i * RPX;
            int hh = xh[i] + dh, ww = xw[i] + dw;
            bool ok = (xc_ok && xc_in) && m < mend && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            rx[i] = ok ? ld16(X + ((long)m + dh * W + dw) * ldx + c0 + xc) : zero16<T>();
            // advance this slot's (h, w) by BKP pixels for the next tile
            xw[i] += BKP;
            while (xw[i] >= W) { xw[i] -= W; xh[i] += 1; }
            while (xh[i] >= H) xh[i] -= H;
        }
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            int m = mt0 + yr + i * RPY;
            bool ok = (yc_ok && yc_in) && m < mend;
            ry[i] = ok ? ld16(dY + (long)m * ldy + n0 + yc) : zero16<T>();
        }
    };
    auto s_store = [&](int buf) {
        T *Xs = smem[buf];
        T *Ys = smem[buf] + BKP * LX;
#pragma unroll
        for (int i = 0; i < XI; ++i) st16(Xs + (xr + i * RPX) * LX + xc, rx[i]);
#pragma unroll
        for (int i = 0; i < YI; ++i) st16(Ys + (yr + i * RPY) * LY + yc, ry[i]);
    };

    const int nk = (mend - mbeg + BKP - 1) / BKP;
    g_load(mbeg);
    s_store(0);
    __syncthreads();

    const int g = lane >> 4, t = lane & 15;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) g_load(mbeg + (kt + 1) * BKP);
        const T *Xs = smem[cur];
        const T *Ys = smem[cur] + BKP * LX;
#pragma unroll
        for (int ks = 0; ks < BKP / KSTEP; ++ks) {
            if constexpr (sizeof(T) == 2) {
                bf16x8 af[TM], bf[TN];
                if constexpr (VARIANT == 0) {
                    // hardware transpose read: lane (g,t) supplies the address of pixel row (t>>2),
                    // channel quad (t&3) of its group's 4x16 block and receives 4 pixels of channel t
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int px = ks * 16 + 8 * (g >> 1) + 4 * r + (t >> 2);
                        const int co = 16 * (g & 1) + 4 * (t & 3);
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const bf16 *p = (const bf16 *)Xs + px * LX + (wm * TM + i) * 32 + co;
                            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                            bf16x4 b = __builtin_bit_cast(bf16x4, v);
                            af[i][4 * r + 0] = b[0]; af[i][4 * r + 1] = b[1]; af[i][4 * r + 2] = b[2]; af[i][4 * r + 3] = b[3];
                        }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const bf16 *p = (const bf16 *)Ys + px * LY + (wn * TN + j) * 32 + co;
                            s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                            bf16x4 b = __builtin_bit_cast(bf16x4, v);
                            bf[j][4 * r + 0] = b[0]; bf[j][4 * r + 1] = b[1]; bf[j][4 * r + 2] = b[2]; bf[j][4 * r + 3] = b[3];
                        }
                    }
                } else {
                    // reference gather (slow, layout-proof): 8 scalar LDS reads per fragment
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int px = ks * 16 + 8 * (lane >> 5) + e;
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[i][e] = ((const bf16 *)Xs)[px * LX + (wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
                        for (int j = 0; j < TN; ++j) bf[j][e] = ((const bf16 *)Ys)[px * LY + (wn * TN + j) * 32 + (lane & 31)];
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            } else {
                float af[TM], bf[TN];
                const int px = ks * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = ((const float *)Xs)[px * LX + (wm * TM + i) * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = ((const float *)Ys)[px * LY + (wn * TN + j) * 32 + (lane & 31)];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) s_store(cur ^ 1);
        __syncthreads();
    }

    // epilogue: rows = input channels c, cols = filters n; dW is HWIO [tap][Cin][Cout]
    float *out = dW + (long)tap * Cin * Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (n >= Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cb = c0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb + (r & 3) + 8 * (r >> 2);
                if (c < Cin) unsafeAtomicAdd(out + (long)c * Cout + n, acc[i][j][r]);
            }
        }
    }
}

static int g_wgrad_variant = 0;
extern "C" void yolo2_debug_set_wgrad_variant(int v) { g_wgrad_variant = v; }

template <typename T, int BC, int BNN>
static void launch_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin, int ldx,
                         int Cout, int ldy, int ksize, hipStream_t st) {
    const int M = B * H * W;
    const int CT = cdiv(Cin, BC), NT = cdiv(Cout, BNN);
    const int tiles = ksize * ksize * CT * NT;
    constexpr int BKP = sizeof(T) == 2 ? 32 : 16;
    // split the pixel range so the grid has >= ~4 blocks per CU, but keep >= 8 tiles per block
    int ks = cdiv(1024, tiles);
    int max_ks = cdiv(M, 8 * BKP);
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    int mchunk = cdiv(cdiv(M, ks), BKP) * BKP;
    ks = cdiv(M, mchunk);
    dim3 grid(tiles, ks);
    if (g_wgrad_variant == 0)
        conv_wgrad_kernel<T, BC, BNN, 0><<<grid, 256, 0, st>>>((const T *)X, (const T *)dY, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk);
    else
        conv_wgrad_kernel<T, BC, BNN, 1><<<grid, 256, 0, st>>>((const T *)X, (const T *)dY, dW, H, W, Cin, ldx, Cout, ldy, ksize, M, CT, NT, mchunk);
}

extern "C" int yolo2_conv2d_wgrad(const void *X, const void *dY, float *dW, int B, int H, int W, int Cin,
                                  int ldx, int Cout, int ldy, int ksize, int dtype, void *stream) {
    Y2_CHECK_ARG(X && dY && dW);
    Y2_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    Y2_CHECK_ARG(ksize == 1 || ksize == 3);
    Y2_CHECK_ARG(ldx >= Cin && ldy >= Cout);
    const int vec = dtype == YOLO2_BF16 ? 8 : 4;
    Y2_CHECK_ARG(ldx % vec == 0 && ldy % vec == 0);
    Y2_CHECK_ARG((long)B * H * W * (long)(ldx > ldy ? ldx : ldy) < (1L << 31));
    hipStream_t st = (hipStream_t)stream;
    const bool small = Cin <= 64 || Cout <= 64;
    if (small) {
        Y2_DISPATCH_DTYPE(dtype, launch_wgrad<T, 64, 64>(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize, st));
    } else {
        Y2_DISPATCH_DTYPE(dtype, launch_wgrad<T, 128, 128>(X, dY, dW, B, H, W, Cin, ldx, Cout, ldy, ksize, st));
    }
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
