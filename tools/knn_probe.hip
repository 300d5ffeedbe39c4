// Phase-level timing of knn_mfma_kernel at the second EdgeConv's shape (B=32, N=M=1024, D=64, k=20+1):
// includes knn.hip with FX3D_PROBE so that thread 0 of every block stores cycle-counter stamps.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFX3D_PROBE -I include -I flux3d.jl_amd/csrc \
//         tools/knn_probe.hip flux3d.jl_amd/csrc/runtime.hip -o tools/knn_probe
#include "../flux3d.jl_amd/csrc/knn.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

int main(int argc, char **argv) {
    const int B = 32, N = 1024, D = argc > 1 ? atoi(argv[1]) : 64, K = 20;
    std::vector<float> hx((size_t)D * N * B);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return getenv("UNIT") ? ((s >> 8) * (1.0f / 16777216.0f)) : ((s >> 8) * (1.0f / 16777216.0f)) * 4.0f - 2.0f; };
    for (auto &v : hx) v = rnd();
    if (getenv("CLOUD")) { FILE *f = fopen(getenv("CLOUD"), "rb"); size_t got = fread(hx.data(), 4, hx.size(), f); fclose(f); printf("loaded %zu floats\n", got); }
    float *x, *dst = nullptr; int32_t *idx;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&idx, (size_t)K * N * B * 4);
    if (getenv("DIST")) hipMalloc(&dst, (size_t)K * N * B * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int it = 0; it < 3; ++it) fx3d_knn(x, N, x, N, B, D, K, 1, idx, dst, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) fx3d_knn(x, N, x, N, B, D, K, 1, idx, dst, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("avg per call %.2f us\n", ms * 100);
    std::vector<unsigned long long> pr(4096 * 32);
    hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_kprobe), pr.size() * 8);
    const int nb = D == 3 ? 512 : 256;
    std::vector<double> d(32, 0.0);
    std::vector<int> c(32, 0);
    for (int b = 0; b < nb; ++b) {
        const unsigned long long *q = &pr[b * 32];
        unsigned long long prev = q[0];
        for (int k = 1; k < 32; ++k) { if (!q[k]) continue; d[k] += (double)(q[k] - prev); c[k]++; prev = q[k]; }
    }
    for (int k = 1; k < 32; ++k) if (c[k]) printf("  mark %2d: avg +%9.1f ticks (n=%d)\n", k, d[k] / c[k], c[k]);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int b = 0; b < nb; ++b) { tmin = std::min(tmin, pr[b * 32]); tmax = std::max(tmax, std::max(pr[b * 32 + 25], pr[b * 32 + 9])); }
    printf("queries %llu slow %llu unusable %llu list-overflow %llu n>cap %llu n<kk %llu sum(n) %llu (x13 launches)\n", pr[4095 * 32], pr[4095 * 32 + 1],
           pr[4095 * 32 + 2], pr[4095 * 32 + 3], pr[4095 * 32 + 4], pr[4095 * 32 + 5], pr[4095 * 32 + 6]);
    printf("count-mismatch %llu bad %llu\n", pr[4095 * 32 + 7], pr[4095 * 32 + 8]);
    printf("first start -> last end: %llu ticks\n", tmax - tmin);
    return 0;
}
