// A C++ host that runs the data-parallel half of a training step through the two C ABIs only (include/yolo2_hip.h, include/yolo2_comm.h):
// no Python, no torch.  One process per GPU (this program forks them): rendezvous, broadcast of rank 0's replica, then per "step" a
// gradient arena is reduced bucket by bucket on a communication stream -- f32 on the wire, then bf16 on the wire -- while the compute
// stream consumes the buckets with the optimizer kernel as they arrive (gscale = 1 / world folds the averaging into the update),
// exactly the schedule of yolo_tf_amd/parallel.GradReducer.  Every result is checked against the closed form.
//
//   hipcc -O2 -std=c++17 examples/dp_host.cpp -Iinclude -Lyolo_tf_amd/csrc -lyolo2hip -lyolo2comm -Wl,-rpath,$PWD/yolo_tf_amd/csrc -o dp_host
//   ./dp_host [world]          (world <= visible GPUs; default 1)
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "yolo2_comm.h"
#include "yolo2_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define COMM_OK(x) do { if ((x) != 0) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, yolo2_comm_last_error()); return 3; } } while (0)
#define Y2_OK(x) do { if ((x) != 0) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, yolo2_last_error()); return 4; } } while (0)

static float bf16_round(float x) {      // round-to-nearest-even, like the cast kernels
    unsigned u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

static int run_rank(int rank, int world, const char *id_path) {
    unsigned char id[YOLO2_COMM_ID_BYTES];
    if (rank == 0) {
        COMM_OK(yolo2_comm_unique_id(id));
        std::string tmp = std::string(id_path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        fwrite(id, 1, sizeof(id), f);
        fclose(f);
        rename(tmp.c_str(), id_path);        // atomic publish
    } else {
        FILE *f = nullptr;
        for (int i = 0; i < 3000 && !(f = fopen(id_path, "rb")); ++i) usleep(10000);
        if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no rendezvous id\n", rank); return 1; }
        fclose(f);
    }
    yolo2_comm *comm = nullptr;
    COMM_OK(yolo2_comm_init(&comm, id, rank, world, rank));
    if (yolo2_comm_rank(comm) != rank || yolo2_comm_world(comm) != world) return 5;

    const long N = (long)3 << 20;                      // a 12 MiB "arena" in 3 buckets with ragged bounds
    const long bounds[4] = {0, (1 << 20) + 4096, (2 << 20) + 8, N};
    float *w, *g, *m, *v;
    void *wire;
    int *scratch;
    HIP_OK(hipMalloc(&w, N * 4)); HIP_OK(hipMalloc(&g, N * 4)); HIP_OK(hipMalloc(&m, N * 4)); HIP_OK(hipMalloc(&v, N * 4));
    HIP_OK(hipMalloc(&wire, N * 2)); HIP_OK(hipMalloc(&scratch, 4));
    hipStream_t compute, commst;
    HIP_OK(hipStreamCreate(&compute));
    HIP_OK(hipStreamCreateWithPriority(&commst, hipStreamNonBlocking, -1));
    std::vector<float> h(N), hw(N);

    // ---- the replicas start identical: rank 0's parameters win
    for (long i = 0; i < N; ++i) hw[i] = (float)(rank + 1) * 0.001f * (float)(i % 977);
    HIP_OK(hipMemcpy(w, hw.data(), N * 4, hipMemcpyHostToDevice));
    COMM_OK(yolo2_comm_broadcast(comm, w, N * 4, 0, commst));
    HIP_OK(hipStreamSynchronize(commst));
    HIP_OK(hipMemcpy(hw.data(), w, N * 4, hipMemcpyDeviceToHost));
    for (long i = 0; i < N; i += 4099) if (hw[i] != 0.001f * (float)(i % 977)) { fprintf(stderr, "rank %d: broadcast mismatch at %ld\n", rank, i); return 6; }

    for (int wire_bf16 = 0; wire_bf16 < 2; ++wire_bf16) {
        // ---- this rank's gradients (what backward would have left in the arena), zero optimizer slots
        for (long i = 0; i < N; ++i) h[i] = (float)((i * 7 + rank * 13) % 251 - 125) / 64.0f;       // exactly representable in bf16
        HIP_OK(hipMemcpyAsync(g, h.data(), N * 4, hipMemcpyHostToDevice, compute));
        HIP_OK(hipMemsetAsync(m, 0, N * 4, compute)); HIP_OK(hipMemsetAsync(v, 0, N * 4, compute));
        // ---- buckets in the order backward completes them (front of the arena first); the optimizer consumes each as it lands
        hipEvent_t ready[3], landed[3];
        for (int b = 0; b < 3; ++b) {
            const long s = bounds[b], n = bounds[b + 1] - bounds[b];
            HIP_OK(hipEventCreateWithFlags(&ready[b], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&landed[b], hipEventDisableTiming));
            if (wire_bf16) Y2_OK(yolo2_cast_f32_bf16(g + s, (char *)wire + s * 2, n, compute));
            HIP_OK(hipEventRecord(ready[b], compute));
            HIP_OK(hipStreamWaitEvent(commst, ready[b], 0));
            if (wire_bf16) COMM_OK(yolo2_comm_allreduce_bucket(comm, (char *)wire + s * 2, n, YOLO2_COMM_BF16, commst));
            else COMM_OK(yolo2_comm_allreduce_bucket(comm, g + s, n, YOLO2_COMM_F32, commst));
            HIP_OK(hipEventRecord(landed[b], commst));
        }
        const float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, alpha = lr * std::sqrt(1.0f - b2) / (1.0f - b1);
        for (int b = 0; b < 3; ++b) {
            const long s = bounds[b], n = bounds[b + 1] - bounds[b];
            HIP_OK(hipStreamWaitEvent(compute, landed[b], 0));
            if (wire_bf16) Y2_OK(yolo2_cast_bf16_f32((char *)wire + s * 2, g + s, n, compute));
            Y2_OK(yolo2_adam(w + s, g + s, m + s, v + s, n, alpha, b1, b2, eps, 1.0f / (float)world, compute));
        }
        HIP_OK(hipStreamSynchronize(compute));
        // ---- closed form: summed gradient, then one Adam step from zero slots on the averaged gradient
        std::vector<float> gs(N), wn(N);
        HIP_OK(hipMemcpy(gs.data(), g, N * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(wn.data(), w, N * 4, hipMemcpyDeviceToHost));
        double worst_g = 0, worst_w = 0;
        for (long i = 0; i < N; i += 1021) {
            double sum = 0;
            for (int r = 0; r < world; ++r) sum += (double)((i * 7 + r * 13) % 251 - 125) / 64.0;
            const double want_g = wire_bf16 ? (double)bf16_round((float)sum) : sum;
            // (bf16 wire: every term is bf16-exact; a ring rounds the running sum at every hop, so allow two bf16 ulps of the result)
            worst_g = fmax(worst_g, fabs(gs[i] - want_g) / (wire_bf16 ? fmax(fabs(sum), 1.0) : 1.0));
            const double ga = (double)gs[i] / world, m1 = (1 - b1) * ga, v1 = (1 - b2) * ga * ga;       // the optimizer on what actually arrived
            const double want_w = (double)hw[i] - alpha * m1 / (std::sqrt(v1) + eps);
            worst_w = fmax(worst_w, fabs(wn[i] - want_w));
        }
        if (worst_g > (wire_bf16 ? (world > 1 ? 1.6e-2 : 0.0) : 1e-5) || worst_w > 2e-6) {
            fprintf(stderr, "rank %d: wire %s: gradient error %.3g, weight error %.3g\n", rank, wire_bf16 ? "bf16" : "f32", worst_g, worst_w);
            return 7;
        }
        hw = wn;
        for (int b = 0; b < 3; ++b) { (void)hipEventDestroy(ready[b]); (void)hipEventDestroy(landed[b]); }
    }
    {   // ---- optimizer sharding ([mi355x] shard_optimizer): reduce-scatter every bucket, Adam on THIS rank's shard only, all-gather the updated
        //      parameter shards -- all on the communication stream; the remainder of a bucket (< world * 64 elements) stays replicated
        for (long i = 0; i < N; ++i) h[i] = (float)((i * 7 + rank * 13) % 251 - 125) / 64.0f;
        HIP_OK(hipMemcpyAsync(g, h.data(), N * 4, hipMemcpyHostToDevice, compute));
        HIP_OK(hipMemsetAsync(m, 0, N * 4, compute)); HIP_OK(hipMemsetAsync(v, 0, N * 4, compute));
        const float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, alpha = lr * std::sqrt(1.0f - b2) / (1.0f - b1);
        hipEvent_t ready;
        HIP_OK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
        for (int b = 0; b < 3; ++b) {
            const long s = bounds[b], n = bounds[b + 1] - bounds[b], L = n / ((long)world * 64) * 64, rem = n - (long)world * L;
            HIP_OK(hipEventRecord(ready, compute));
            HIP_OK(hipStreamWaitEvent(commst, ready, 0));
            if (L) COMM_OK(yolo2_comm_reduce_scatter_bucket(comm, g + s, L, YOLO2_COMM_F32, commst));
            if (rem) COMM_OK(yolo2_comm_allreduce_bucket(comm, g + s + world * L, rem, YOLO2_COMM_F32, commst));
            const long o = s + (long)rank * L;
            if (L) Y2_OK(yolo2_adam(w + o, g + o, m + o, v + o, L, alpha, b1, b2, eps, 1.0f / (float)world, commst));
            if (rem) Y2_OK(yolo2_adam(w + s + world * L, g + s + world * L, m + s + world * L, v + s + world * L, rem, alpha, b1, b2, eps, 1.0f / (float)world, commst));
            if (L) COMM_OK(yolo2_comm_allgather(comm, w + s, L * 4, commst));
        }
        HIP_OK(hipStreamSynchronize(commst));
        std::vector<float> wn(N);
        HIP_OK(hipMemcpy(wn.data(), w, N * 4, hipMemcpyDeviceToHost));
        double worst_w = 0;
        for (long i = 0; i < N; i += 509) {
            double sum = 0;
            for (int r = 0; r < world; ++r) sum += (double)((i * 7 + r * 13) % 251 - 125) / 64.0;
            const double ga = sum / world, m1 = (1 - b1) * ga, v1 = (1 - b2) * ga * ga;
            worst_w = fmax(worst_w, fabs(wn[i] - ((double)hw[i] - alpha * m1 / (std::sqrt(v1) + eps))));
        }
        if (worst_w > 2e-6) { fprintf(stderr, "rank %d: sharded update: weight error %.3g\n", rank, worst_w); return 10; }
        hw = wn;
        (void)hipEventDestroy(ready);
    }
    // ---- a decision all ranks take together (train.py's agree()): only the last rank saw a problem
    int verdict = -1;
    COMM_OK(yolo2_comm_agree_max(comm, rank == world - 1 ? 1 : 0, &verdict, scratch, commst));
    if (verdict != 1) { fprintf(stderr, "rank %d: agree_max gave %d\n", rank, verdict); return 8; }
    // ---- argument errors come back as codes with a message, not as crashes
    if (yolo2_comm_allreduce_bucket(comm, nullptr, 4, YOLO2_COMM_F32, commst) != YOLO2_COMM_E_ARG || !strstr(yolo2_comm_last_error(), "allreduce_bucket")) return 9;
    HIP_OK(hipDeviceSynchronize());
    COMM_OK(yolo2_comm_destroy(comm));
    Y2_OK(yolo2_shutdown());
    printf("rank %d of %d: broadcast, 3-bucket all-reduce (f32 and bf16 wire) + overlapped Adam, agree: OK\n", rank, world);
    return 0;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    if (world < 1) { fprintf(stderr, "usage: dp_host [world >= 1]\n"); return 64; }
    char id_path[64];
    snprintf(id_path, sizeof(id_path), "/tmp/yolo2_dp_host_%d.id", (int)getpid());
    std::vector<pid_t> kids;
    for (int r = 1; r < world; ++r) {        // fork BEFORE any HIP call: a HIP context does not survive fork()
        pid_t p = fork();
        if (p == 0) return run_rank(r, world, id_path);
        kids.push_back(p);
    }
    int rc = run_rank(0, world, id_path);
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 10;
    }
    unlink(id_path);
    return rc;
}
