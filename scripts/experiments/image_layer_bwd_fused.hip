// EXPERIMENT RECORD -- not built, not part of libyolo2hip.so.
// One-pass backward of the image layer (conv0 -> batch norm -> leaky -> 2x2 max pool) with a closed-form filter gradient.
// Exact (it was tested against an f64 evaluation in round 1) but measured 2.3x SLOWER than the three unfused kernels it
// replaces (0.48 ms vs 0.21 ms at batch 16: 316 VGPRs = one wave per SIMD, ~25 VALU per element with nothing to hide behind),
// so it was removed from the library, the header and the engine in round 2.  Kept here for the derivation.
// It compiled inside yolo_tf_amd/csrc/conv_first.hip (it uses that file's First<T> helpers and DMA idioms).
// ---------------------------------------------------------------------------------------------------
// Whole backward of the image layer in ONE pass (bf16): conv0 -> batch norm -> leaky -> 2x2 max pool.
//
// Nothing upstream needs dX, so the only results are dW (864 numbers), dgamma and dbeta.  The unfused chain reads the
// 177 MB conv output twice, writes a 177 MB dY and reads it back (reduce 82 us + apply 70 us + filter gradient 61 us at
// batch 16).  dY never has to exist:
//     dY = gamma*inv * (g - dbeta/M - xhat*dgamma/M),   g = leaky'(z) * route(dP, idx),   xhat = (y - mean)*inv
//     dW[tc,n] = sum_m xs[m,tc] dY[m,n]
//              = gamma_n inv_n * ( S1[tc,n] - dbeta_n/M * S2[tc] - dgamma_n/M * inv_n * (S3[tc,n] - mean_n S2[tc]) )
//     S1 = sum xs*g,  S3 = sum xs*y,  S2 = sum xs (over the valid output pixels),  dbeta = sum g,
//     dgamma = inv * (sum g*y - mean * dbeta)            (xs = the tap-shifted, zero-padded input, tc = (tap, channel))
// so one sweep over (x, y, dP, idx) accumulates S1, S3, S2 on the MFMA (three B operands: the g tile this wave just
// computed into LDS, the y tile, and a ones column) and the two column sums in registers; a one-block finalisation
// evaluates the closed form in f64.  Same wave-per-32-pixel-segment structure as the filter gradient above.
// STATUS: exact (tested against an f64 evaluation) but not yet a win -- nine accumulator tiles + per-channel constants = 316
// VGPRs = one wave per SIMD, and the g computation is ~25 VALU per element with nothing to hide it behind: 0.48 ms against
// 0.21 ms for the three unfused kernels at batch 16.  The engine keeps the unfused chain (YOLO2_FUSE_IMAGE_BWD=1 selects
// this kernel); the restructuring that should pay (wave-specialised accumulators, S2 from the forward pass, a deeper DMA
// ring) is next round's.
// ---------------------------------------------------------------------------------------------------
#define Y2_FB_S1 0
#define Y2_FB_S3 (72 * 32)
#define Y2_FB_S2 (2 * 72 * 32)
#define Y2_FB_SG (2 * 72 * 32 + 96)
#define Y2_FB_SGY (2 * 72 * 32 + 96 + 32)
#define Y2_FB_FLOATS (2 * 72 * 32 + 96 + 64)

__global__ __launch_bounds__(256) void conv_first_bwd_fused_kernel(const bf16 *__restrict__ X, unsigned x_bytes, const bf16 *__restrict__ Y, unsigned y_bytes,
                                                                   const bf16 *__restrict__ dP, unsigned dp_bytes, const unsigned char *__restrict__ idx,
                                                                   unsigned idx_bytes, const float *__restrict__ mean, const float *__restrict__ var,
                                                                   const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ scratch,
                                                                   int B, int H, int W, int units, float eps, float alpha) {
    typedef bf16 T;
    constexpr int PXB = First<T>::PXB, HALO = First<T>::HALO;       // 16 B per halo pixel, 3 KiB
    constexpr int YROWB = 64;                                       // 32 channels x bf16
    constexpr int OFF_Y = HALO, OFF_DP = HALO + 2048, OFF_IDX = HALO + 3072, DMA = HALO + 4096;   // DMA-filled part of a slot
    constexpr int SLOT = DMA;                                       // double buffered; the g tile (2 KiB) is single buffered
    constexpr int WAVE_LDS = 2 * SLOT + 2048;
    constexpr int LOADS = 3 + 2 + 1 + 1;
    constexpr int RED = 4 * 3 * 16 * 64 * 4;
    constexpr int SMEM = 4 * WAVE_LDS > RED ? 4 * WAVE_LDS : RED;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *my = smem + wave * WAVE_LDS;
    unsigned char *gs = my + 2 * SLOT;
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(X), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(Y), 0, y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16 *>(dP), 0, dp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcI = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(idx), 0, idx_bytes, 0x00020000);
    const int SW = (W + 31) / 32, OH = H / 2, OW = W / 2;

    auto stage = [&](unsigned char *dst, int b, int h, int w0) {
        stage_halo<T>(rsrcX, dst, b, h, w0, H, W, lane);
#pragma unroll
        for (int p = 0; p < 2; ++p) {                                // y tile: 32 pixels x 64 B
            const int chunk = p * 64 + lane, px = chunk >> 2;
            const unsigned voff = (w0 + px < W) ? (unsigned)((((long)b * H + h) * W + w0 + px) * YROWB + (chunk & 3) * 16) : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (__attribute__((address_space(3))) void *)(dst + OFF_Y + p * 1024), 16, voff, 0, 0, 0);
        }
        {                                                            // pooled gradient: 16 pooled pixels x 64 B
            const int ppx = lane >> 2, pw = (w0 >> 1) + ppx;
            const unsigned voff = (pw < OW) ? (unsigned)((((long)b * OH + (h >> 1)) * OW + pw) * YROWB + (lane & 3) * 16) : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcP, (__attribute__((address_space(3))) void *)(dst + OFF_DP), 16, voff, 0, 0, 0);
        }
        {                                                            // arg-max bytes: 16 pooled pixels x 32 B (lanes 0..31)
            const int ppx = lane >> 1, pw = (w0 >> 1) + ppx;
            const unsigned voff = (lane < 32 && pw < OW) ? (unsigned)((((long)b * OH + (h >> 1)) * OW + pw) * 32 + (lane & 1) * 16) : Y2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcI, (__attribute__((address_space(3))) void *)(dst + OFF_IDX), 16, voff, 0, 0, 0);
        }
    };

    // this lane's 16 channels in the g computation: pixel lane>>1 of the segment, channels 16*(lane&1) ..
    const int gpx = lane >> 1, gch = (lane & 1) * 16;
    float mu[16], sc[16], bt[16], sg[16], sgy[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mu[j] = mean[gch + j];
        sc[j] = (1.0f / sqrtf(var[gch + j] + eps)) * gamma[gch + j];
        bt[j] = beta[gch + j];
        sg[j] = sgy[j] = 0.f;
    }

    f32x16 accg[3], accy[3], acc1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[t][r] = accy[t][r] = acc1[t][r] = 0.f;

    const int stride = gridDim.x * 4;
    int u = blockIdx.x * 4 + wave;
    auto decode = [&](int uu, int &b, int &h, int &w0) { w0 = (uu % SW) * 32; int t = uu / SW; h = t % H; b = t / H; };
    int b = 0, h = 0, w0 = 0;
    if (u < units) {
        decode(u, b, h, w0);
        stage(my, b, h, w0);
    }
    const int g4 = lane >> 4, t16 = lane & 15;
    int st = 0;
    for (; u < units; u += stride) {
        const int un = u + stride;
        int nb = 0, nh = 0, nw0 = 0;
        if (un < units) {
            decode(un, nb, nh, nw0);
            stage(my + (st ^ 1) * SLOT, nb, nh, nw0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *hs = my + st * SLOT;
        const unsigned char *ys = hs + OFF_Y;
        // ---- g tile: route the pooled gradient to the arg-max position, apply leaky' on the recomputed pre-activation
        {
            const bool valid = w0 + gpx < W;
            const int k = ((h & 1) << 1) | (gpx & 1);                 // this pixel's position inside its 2x2 window
            const bf16x8 y0 = *reinterpret_cast<const bf16x8 *>(ys + gpx * YROWB + gch * 2);
            const bf16x8 y1 = *reinterpret_cast<const bf16x8 *>(ys + gpx * YROWB + gch * 2 + 16);
            const bf16x8 d0 = *reinterpret_cast<const bf16x8 *>(hs + OFF_DP + (gpx >> 1) * YROWB + gch * 2);
            const bf16x8 d1 = *reinterpret_cast<const bf16x8 *>(hs + OFF_DP + (gpx >> 1) * YROWB + gch * 2 + 16);
            const unsigned long long i0 = *reinterpret_cast<const unsigned long long *>(hs + OFF_IDX + (gpx >> 1) * 32 + gch);
            const unsigned long long i1 = *reinterpret_cast<const unsigned long long *>(hs + OFF_IDX + (gpx >> 1) * 32 + gch + 8);
            bf16x8 o0, o1;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float yv = (float)(j < 8 ? y0[j] : y1[j - 8]);
                const float dv = (float)(j < 8 ? d0[j] : d1[j - 8]);
                const int ix = (int)(((j < 8 ? i0 : i1) >> (8 * (j & 7))) & 3);
                const float da = (valid && ix == k) ? dv : 0.f;
                const float z = (yv - mu[j]) * sc[j] + bt[j];
                const bf16 gq = (bf16)(z >= 0.f ? da : alpha * da);
                const float gf = (float)gq;                           // the rounded value is what the MFMA sums: keep the column sums consistent
                sg[j] += gf;
                sgy[j] += gf * yv;
                if (j < 8) o0[j] = gq; else o1[j - 8] = gq;
            }
            *reinterpret_cast<bf16x8 *>(gs + gpx * YROWB + gch * 2) = o0;
            *reinterpret_cast<bf16x8 *>(gs + gpx * YROWB + gch * 2 + 16) = o1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own LDS writes precede the transposed reads below
        // ---- S1 += xs^T g,  S3 += xs^T y,  S2 += xs^T 1  (reduction over the segment's 32 pixels)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bg, by, b1, afrag[3];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int px = ks * 16 + 8 * (g4 >> 1) + 4 * r + (t16 >> 2);     // pixel row this lane addresses
                const int coff = (16 * (g4 & 1) + 4 * (t16 & 3)) * 2;
                {
                    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(gs + px * YROWB + coff));
                    bf16x4 q = __builtin_bit_cast(bf16x4, v);
                    bg[4 * r] = q[0]; bg[4 * r + 1] = q[1]; bg[4 * r + 2] = q[2]; bg[4 * r + 3] = q[3];
                    v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ys + px * YROWB + coff));
                    q = __builtin_bit_cast(bf16x4, v);
                    by[4 * r] = q[0]; by[4 * r + 1] = q[1]; by[4 * r + 2] = q[2]; by[4 * r + 3] = q[3];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)                            // element e of read r = pixel ks*16 + 8*(lane>>5) + 4r + e
                    b1[4 * r + e] = (w0 + ks * 16 + 8 * (lane >> 5) + 4 * r + e < W) ? (bf16)1.0f : (bf16)0.0f;
#pragma unroll
                for (int rt = 0; rt < 3; ++rt) {
                    const int row = 32 * rt + 16 * (g4 & 1) + 4 * (t16 & 3);
                    int tap = row >> 3;
                    if (tap > 8) tap = 8;
                    const unsigned char *p = hs + (tap / 3) * First<T>::HROWB + (px + tap % 3) * PXB + (row & 7) * 2;
                    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
                    bf16x4 q = __builtin_bit_cast(bf16x4, v);
                    afrag[rt][4 * r] = q[0]; afrag[rt][4 * r + 1] = q[1]; afrag[rt][4 * r + 2] = q[2]; afrag[rt][4 * r + 3] = q[3];
                }
            }
#pragma unroll
            for (int rt = 0; rt < 3; ++rt) {
                accg[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rt], bg, accg[rt], 0, 0, 0);
                accy[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rt], by, accy[rt], 0, 0, 0);
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag[rt], b1, acc1[rt], 0, 0, 0);
            }
        }
        b = nb; h = nh; w0 = nw0;
        st ^= 1;
    }

    // column sums: lanes with the same channel half (lane & 1) hold different pixels
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) {
            sg[j] += __shfl_xor(sg[j], o, 64);
            sgy[j] += __shfl_xor(sgy[j], o, 64);
        }
    }
    if (lane < 2) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            unsafeAtomicAdd(scratch + Y2_FB_SG + gch + j, sg[j]);
            unsafeAtomicAdd(scratch + Y2_FB_SGY + gch + j, sgy[j]);
        }
    }
    // combine the 4 waves through LDS, one accumulator set at a time (the staging slots are dead now)
    float *red = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int set = 0; set < 3; ++set) {
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[((wave * 3 + rt) * 16 + r) * 64 + lane] = set == 0 ? accg[rt][r] : set == 1 ? accy[rt][r] : acc1[rt][r];
        __syncthreads();
        if (wave == 0) {
            const int n = lane & 31;
#pragma unroll
            for (int rt = 0; rt < 3; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);    // (tap, channel)
                    if (row < 72 && (set < 2 || n == 0)) {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) v += red[((w * 3 + rt) * 16 + r) * 64 + lane];
                        if (set == 0) unsafeAtomicAdd(scratch + Y2_FB_S1 + row * 32 + n, v);
                        else if (set == 1) unsafeAtomicAdd(scratch + Y2_FB_S3 + row * 32 + n, v);
                        else unsafeAtomicAdd(scratch + Y2_FB_S2 + row, v);
                    }
                }
        }
    }
}

// closed form in f64; leaves the scratch zero for the next step
__global__ __launch_bounds__(256) void conv_first_bwd_finalize_kernel(float *__restrict__ scratch, const float *__restrict__ mean, const float *__restrict__ var,
                                                                      const float *__restrict__ gamma, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                      float *__restrict__ dW, int Cin, double M, float eps) {
    __shared__ double s2[72], db[32], dg[32], inv[32];
    const int tid = threadIdx.x;
    if (tid < 72) s2[tid] = (double)scratch[Y2_FB_S2 + tid];
    if (tid < 32) {
        const double iv = 1.0 / sqrt((double)var[tid] + (double)eps);
        const double sgv = (double)scratch[Y2_FB_SG + tid], sgyv = (double)scratch[Y2_FB_SGY + tid];
        inv[tid] = iv;
        db[tid] = sgv;
        dg[tid] = iv * (sgyv - (double)mean[tid] * sgv);
        dbeta[tid] = (float)sgv;
        dgamma[tid] = (float)dg[tid];
    }
    __syncthreads();
    for (int i = tid; i < 72 * 32; i += 256) {
        const int row = i >> 5, n = i & 31, tap = row >> 3, c = row & 7;
        if (c < Cin) {
            const double S1 = (double)scratch[Y2_FB_S1 + i], S3 = (double)scratch[Y2_FB_S3 + i];
            const double v = (double)gamma[n] * inv[n] * (S1 - db[n] / M * s2[row] - dg[n] / M * inv[n] * (S3 - (double)mean[n] * s2[row]));
            dW[((long)tap * Cin + c) * 32 + n] = (float)v;
        }
    }
    __syncthreads();
    for (int i = tid; i < Y2_FB_FLOATS; i += 256) scratch[i] = 0.f;
}

int y2_first_layer_bwd_fused(const void *X, const void *Y, const void *dP, const unsigned char *idx, const float *mean, const float *var,
                             const float *gamma, const float *beta, float *dgamma, float *dbeta, float *dW, float *scratch, int B, int H, int W,
                             int Cin, float eps, float alpha, hipStream_t st) {
    const int units = B * H * ((W + 31) / 32);
    const int grid = units / 4 + 1 < 512 ? units / 4 + 1 : 512;
    conv_first_bwd_fused_kernel<<<grid, 256, 0, st>>>((const bf16 *)X, (unsigned)((size_t)B * H * W * 8 * 2), (const bf16 *)Y, (unsigned)((size_t)B * H * W * 32 * 2),
                                                      (const bf16 *)dP, (unsigned)((size_t)B * (H / 2) * (W / 2) * 32 * 2), idx,
                                                      (unsigned)((size_t)B * (H / 2) * (W / 2) * 32), mean, var, gamma, beta, scratch, B, H, W, units, eps, alpha);
    conv_first_bwd_finalize_kernel<<<1, 256, 0, st>>>(scratch, mean, var, gamma, dgamma, dbeta, dW, Cin, (double)B * H * W, eps);
    return 0;
}


// C ABI: the whole backward of the image layer (conv -> batch norm -> leaky -> 2x2 max pool) in one pass; see include/yolo2_hip.h
extern "C" int yolo2_image_layer_bwd(const void *X, const void *Y, const void *dP, const unsigned char *idx, const float *mean, const float *var,
                                     const float *gamma, const float *beta, float *dgamma, float *dbeta, float *dW, float *scratch, int B, int H,
                                     int W, int Cin, float eps, float alpha, int dtype, void *stream) {
    Y2_CHECK_ARG(X && Y && dP && idx && mean && var && gamma && beta && dgamma && dbeta && dW && scratch);
    Y2_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin > 0 && Cin <= 8);
    Y2_CHECK_ARG(dtype == YOLO2_BF16);                                    // the f32 parity mode keeps the unfused chain
    Y2_CHECK_ARG((size_t)B * H * W * 32 * 2 < (1ull << 31));              // 32-bit DMA offsets
    y2_first_layer_bwd_fused(X, Y, dP, idx, mean, var, gamma, beta, dgamma, dbeta, dW, scratch, B, H, W, Cin, eps, alpha, (hipStream_t)stream);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

