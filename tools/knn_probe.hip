// Phase-level timing of knn_mfma_kernel at the second EdgeConv's shape (B=32, N=M=1024, D=64, k=20+1):
// includes knn.hip with FX3D_PROBE so that thread 0 of every block stores cycle-counter stamps.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFX3D_PROBE -I include -I flux3d.jl_amd/csrc \
//         tools/knn_probe.hip flux3d.jl_amd/csrc/runtime.hip -o tools/knn_probe
#define FX3D_KNN_ONE_TU  // the three units of the k-NN family in one translation unit: one g_kprobe, no cross-unit shims
#include "../flux3d.jl_amd/csrc/knn_d3.hip"
#include "../flux3d.jl_amd/csrc/knn_mfma.hip"
#include "../flux3d.jl_amd/csrc/knn.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void probe_empty_kernel(int *p) {
    extern __shared__ int sm[];
    if (p && threadIdx.x == 9999) p[0] = sm[0];
}

int main(int argc, char **argv) {
    const int B = 32, N = 1024, D = argc > 1 ? atoi(argv[1]) : 64, K = getenv("K") ? atoi(getenv("K")) : 20;
    std::vector<float> hx((size_t)D * N * B);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return getenv("UNIT") ? ((s >> 8) * (1.0f / 16777216.0f)) : ((s >> 8) * (1.0f / 16777216.0f)) * 4.0f - 2.0f; };
    for (auto &v : hx) v = rnd();
    if (getenv("CLOUD")) { FILE *f = fopen(getenv("CLOUD"), "rb"); size_t got = fread(hx.data(), 4, hx.size(), f); fclose(f); printf("loaded %zu floats\n", got); }
    float *x, *dst = nullptr; int32_t *idx;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&idx, (size_t)K * N * B * 4);
    if (getenv("DIST")) hipMalloc(&dst, (size_t)K * N * B * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    void *ws = nullptr; size_t wsb = 0;
    if (getenv("WS")) { fx3d_knn_workspace_bytes(N, N, B, D, K, 1, &wsb); if (wsb) hipMalloc(&ws, wsb); printf("pre-pass workspace %zu bytes\n", wsb); }
#define fx3d_knn(x_, N_, y_, M_, B_, D_, K_, dr_, idx_, dst_, s_) fx3d_knn_ws(x_, N_, y_, M_, B_, D_, K_, dr_, idx_, dst_, ws, wsb, s_)
    for (int it = 0; it < 3; ++it) fx3d_knn(x, N, x, N, B, D, K, 1, idx, dst, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) fx3d_knn(x, N, x, N, B, D, K, 1, idx, dst, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("avg per call %.2f us\n", ms * 100);
    {
        // isolated launches (a dependent pipeline sees these, not the back-to-back rate) + an empty kernel of the
        // D = 3 kernel's geometry (256 x 512 threads, 136 KiB LDS) as the launch floor
        float iso = 0, emp = 0, t;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&probe_empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
        for (int it = 0; it < 10; ++it) {
            hipDeviceSynchronize();
            hipEventRecord(e0); fx3d_knn(x, N, x, N, B, D, K, 1, idx, dst, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1); iso += t;
            hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(probe_empty_kernel, dim3(256), dim3(512), 136 * 1024, 0, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1); emp += t;
        }
        printf("isolated launch: %.2f us; empty kernel of the same geometry: %.2f us\n", iso * 100, emp * 100);
    }
    std::vector<unsigned long long> pr(4096 * 32);
    hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_kprobe), pr.size() * 8);
    const int nb = D == 3 ? (K + 1 > 32 ? 1024 : 512) : 256;
    std::vector<double> d(32, 0.0);
    std::vector<int> c(32, 0);
    for (int b = 0; b < nb; ++b) {
        const unsigned long long *q = &pr[b * 32];
        unsigned long long prev = q[0];
        for (int k = 1; k < 32; ++k) { if (!q[k]) continue; d[k] += (double)(q[k] - prev); c[k]++; prev = q[k]; }
    }
    for (int k = 1; k < 32; ++k) if (c[k]) printf("  mark %2d: avg +%9.1f ticks (n=%d)\n", k, d[k] / c[k], c[k]);
    if (getenv("RAW"))  // stamps of a few blocks relative to their first one (marks need not be in index order)
        for (int b = 0; b < 3; ++b) {
            printf("block %d:", b);
            for (int k = 1; k < 32; ++k) if (pr[b * 32 + k]) printf(" %d:%llu", k, pr[b * 32 + k] - pr[b * 32]);
            printf("\n");
        }
    unsigned long long tmin = ~0ull, tmax = 0;
    double bsum = 0, bmax = 0, smax = 0;
    int nbk = 0;
    for (int b = 0; b < nb; ++b) {
        const unsigned long long *q = &pr[b * 32];
        if (!q[0]) continue;
        unsigned long long last = 0;
        for (int k = 1; k < 32; ++k) last = std::max(last, q[k]);
        tmin = std::min(tmin, q[0]);
        tmax = std::max(tmax, last);
        bsum += (double)(last - q[0]);
        bmax = std::max(bmax, (double)(last - q[0]));
        ++nbk;
    }
    for (int b = 0; b < nb; ++b) if (pr[b * 32]) smax = std::max(smax, (double)(pr[b * 32] - tmin));
    if (D == 3)
        for (int b = 0; b < nb; ++b) {
            const unsigned long long *q = &pr[b * 32];
            if (q[11] > q[10]) printf("block %d: tail after the output stage (tie re-rank / leftovers) %llu ticks\n", b, q[11] - q[10]);
        }
    {   // the slowest block (a one-round launch lasts as long as it does): its stamps next to the average block's
        int worst = 0;
        double wt = 0;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long *q = &pr[b * 32];
            if (!q[0]) continue;
            unsigned long long last = 0;
            for (int k = 1; k < 32; ++k) last = std::max(last, q[k]);
            if ((double)(last - q[0]) > wt) { wt = (double)(last - q[0]); worst = b; }
        }
        auto timeline = [&](int b, const char *what) {  // a block's stamps in TIME order
            std::vector<std::pair<unsigned long long, int>> ev;
            for (int k = 1; k < 32; ++k) if (pr[b * 32 + k]) ev.push_back({pr[b * 32 + k], k});
            std::sort(ev.begin(), ev.end());
            printf("%s %d:", what, b);
            unsigned long long prev = pr[b * 32];
            for (auto &e : ev) { printf(" %d:+%llu", e.second, e.first - prev); prev = e.first; }
            printf("\n");
        };
        timeline(worst, "slowest block");
        timeline(0, "block");
        timeline(100, "block");
        std::vector<double> tot;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long *q = &pr[b * 32];
            if (!q[0]) continue;
            unsigned long long last = 0;
            for (int k = 1; k < 32; ++k) last = std::max(last, q[k]);
            tot.push_back((double)(last - q[0]));
        }
        std::sort(tot.begin(), tot.end());
        if (!tot.empty()) printf("block durations: p50 %.0f p90 %.0f p99 %.0f max %.0f\n", tot[tot.size() / 2], tot[tot.size() * 9 / 10], tot[tot.size() * 99 / 100], tot.back());
    }
    printf("blocks %d: thread-0 time avg %.0f max %.0f ticks; latest block start +%.0f; first start -> last end %.0f ticks\n", nbk,
           bsum / std::max(nbk, 1), bmax, smax, (double)(tmax - tmin));
    printf("queries %llu slow %llu unusable %llu list-overflow %llu n>cap %llu n<kk %llu sum(n) %llu (x13 launches)\n", pr[4095 * 32], pr[4095 * 32 + 1],
           pr[4095 * 32 + 2], pr[4095 * 32 + 3], pr[4095 * 32 + 4], pr[4095 * 32 + 5], pr[4095 * 32 + 6]);
    printf("count-mismatch %llu bad %llu\n", pr[4095 * 32 + 7], pr[4095 * 32 + 8]);
    return 0;
}
