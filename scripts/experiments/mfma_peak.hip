// Sustained v_mfma_f32_32x32x16_bf16 rate on this GPU: how far below the 2.5 PFLOP/s datasheet peak does a pure MFMA loop run?
// build: hipcc --offload-arch=gfx950 -O3 scripts/experiments/mfma_peak.hip -o /tmp/mfma_peak ; run: /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(float *out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *out;
    hipMalloc(&out, 1 << 24);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int cus = p.multiProcessorCount;
    for (int waves = 4; waves <= 16; waves *= 2)
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000, nacc = 4;
            const int wgs = cus * (waves > 8 ? waves / 8 : 1), threads = waves > 8 ? 512 : waves * 64;
            hipEventRecord(a);
            mfma_loop<4><<<wgs, threads>>>(out, iters);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double flop = (double)wgs * (threads / 64) * iters * nacc * 32 * 32 * 16 * 2;
            printf("%2d waves/CU: %.3f ms  %.0f TFLOP/s  (%.0f MHz-equivalent at 1024 flop/clk/SIMD)\n", waves, ms, flop / ms * 1e-9,
                   flop / ms * 1e-3 / (cus * 4 * 1024.0) * 1e-3);
        }
    return 0;
}
