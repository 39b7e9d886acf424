// How fast can N workgroups add private f32 partial tiles into ONE shared tile?  (sizing the filter-gradient reduction: conv_wgrad.hip)
//   ./atomic_rate  -> per (tile floats, workgroups): us for unsafeAtomicAdd (hardware f32 atomics, lanes along consecutive addresses),
//                     and for the two-pass alternative (plain stores of every partial + one reduce pass)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(512) void add_tile(float *dst, int tile_floats, int reps) {
    const float v = 1.0f + (float)(threadIdx.x & 3);
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < tile_floats; i += 512) unsafeAtomicAdd(dst + i, v);
}
__global__ __launch_bounds__(512) void store_tile(float *ws, int tile_floats) {
    float *p = ws + (size_t)blockIdx.x * tile_floats;
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (int i = threadIdx.x * 4; i < tile_floats; i += 2048) *reinterpret_cast<float4 *>(p + i) = v;
}
__global__ __launch_bounds__(256) void reduce_tiles(const float *ws, float *dst, int tile_floats, int n) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= tile_floats) return;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n; ++k) {
        const float4 v = *reinterpret_cast<const float4 *>(ws + (size_t)k * tile_floats + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4 *>(dst + i) = a;
}
int main() {
    const int tiles[] = {4096, 8192, 24576, 73728, 147456, 294912};
    const int wgs[] = {256, 512, 1024};
    float *dst, *ws;
    hipMalloc(&dst, 294912 * 4 * 2);
    hipMalloc(&ws, (size_t)1024 * 294912 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int t : tiles)
        for (int g : wgs) {
            if ((size_t)t * g > (size_t)300e6) continue;
            float ms_at = 0, ms_2p = 0;
            for (int it = 0; it < 3; ++it) {
                hipMemset(dst, 0, t * 4);
                hipDeviceSynchronize();
                hipEventRecord(a);
                add_tile<<<g, 512>>>(dst, t, 1);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms_at, a, b);
                hipEventRecord(a);
                store_tile<<<g, 512>>>(ws, t);
                reduce_tiles<<<(t / 4 + 255) / 256, 256>>>(ws, dst + 294912, t, g);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms_2p, a, b);
            }
            printf("tile %7d floats x %4d workgroups = %6.1f M adds: atomics %7.1f us (%.1f G adds/s)   store+reduce %7.1f us\n", t, g, (double)t * g / 1e6,
                   ms_at * 1e3, (double)t * g / (ms_at * 1e-3) / 1e9, ms_2p * 1e3);
        }
    return 0;
}
