// Phase-level timing of the nn1 kernels at BASELINE config 2 (B=32, N=M=4096): includes chamfer.hip
// with FX3D_PROBE so the kernel stores s_memtime stamps per block.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DFX3D_PROBE -I include -I flux3d.jl_amd/csrc \
//         tools/nn1_probe.hip flux3d.jl_amd/csrc/runtime.hip -o tools/nn1_probe
#ifndef FX3D_CHAMFER_SRC  // (A/B builds point this at another revision of the kernel source)
#define FX3D_CHAMFER_SRC "../flux3d.jl_amd/csrc/chamfer.hip"
#endif
#include FX3D_CHAMFER_SRC

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    // shape: C2 (B = 32, N = M = 4096) unless PB / PN / PM say otherwise (C3's chamfer launch: PB=8 PN=5000 PM=5000 PNB=240)
    const int B = getenv("PB") ? atoi(getenv("PB")) : 32, N = getenv("PN") ? atoi(getenv("PN")) : 4096, M = getenv("PM") ? atoi(getenv("PM")) : 4096;
    std::vector<float> hx((size_t)3 * N * B), hy((size_t)3 * M * B);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (auto &v : hx) v = rnd();
    for (auto &v : hy) v = rnd();
    if (getenv("PFILE_X") && getenv("PFILE_Y")) {  // raw Float32 (3, N, B) / (3, M, B) files (e.g. bench.py's surface-sampled clouds)
        FILE *fa = fopen(getenv("PFILE_X"), "rb"), *fb = fopen(getenv("PFILE_Y"), "rb");
        if (!fa || !fb || fread(hx.data(), 4, hx.size(), fa) != hx.size() || fread(hy.data(), 4, hy.size(), fb) != hy.size()) { printf("cannot read the cloud files\n"); return 1; }
        fclose(fa); fclose(fb);
    }
    if (getenv("PNORMAL")) {  // standard normal coordinates (sum of 12 uniforms)
        for (auto &v : hx) { float a = 0; for (int i = 0; i < 12; ++i) a += rnd(); v = a - 6.0f; }
        for (auto &v : hy) { float a = 0; for (int i = 0; i < 12; ++i) a += rnd(); v = a - 6.0f; }
    }
    if (getenv("POUTLIER")) {  // tools/nn1_distribution_time.py's "outlier": a tiny bulk and one point 1e4 away per cloud
        for (auto &v : hx) v *= 1e-2f;
        for (auto &v : hy) v *= 1e-2f;
        for (int b = 0; b < B; ++b) for (int d = 0; d < 3; ++d) { hx[(size_t)b * N * 3 + d] = 1e4f; hy[(size_t)b * M * 3 + d] = 1e4f; }
    }
    if (getenv("PSHIFT")) for (auto &v : hx) v += (float)atof(getenv("PSHIFT"));
    if (getenv("PLINE")) {  // the reference harness's input (benchmarks/metrics.jl:11-15): p_i = (i, i, i) / n, the same cloud on both sides
        for (int b = 0; b < B; ++b) {
            for (int i = 0; i < N; ++i) for (int d = 0; d < 3; ++d) hx[((size_t)b * N + i) * 3 + d] = (float)(i + 1) / (float)N;
            for (int i = 0; i < M; ++i) for (int d = 0; d < 3; ++d) hy[((size_t)b * M + i) * 3 + d] = (float)(i + 1) / (float)M;
        }
    }
    if (getenv("PLATTICE")) {  // a 16^3 lattice (exact ties everywhere); PLATTICE=same: y = x
        for (auto &v : hx) v = (float)((int)(rnd() * 16) % 16) * 0.0625f;
        for (auto &v : hy) v = (float)((int)(rnd() * 16) % 16) * 0.0625f;
        if (!strcmp(getenv("PLATTICE"), "same")) hy = hx;
    }
    if (getenv("PCLUSTERS")) {  // 40 cluster centres ~ N(0, 3^2) per cloud, jitter PSIGMA (1e-3); PCLUSTERS=shared: y around x's centres; =same: y = x
        const float sigma = getenv("PSIGMA") ? (float)atof(getenv("PSIGMA")) : 1e-3f;
        auto gauss = [&]() { float a = 0; for (int i = 0; i < 12; ++i) a += rnd(); return a - 6.0f; };
        const bool shared = !strcmp(getenv("PCLUSTERS"), "shared"), same = !strcmp(getenv("PCLUSTERS"), "same");
        for (int b = 0; b < B; ++b) {
            float cx[40][3], cy[40][3];
            for (int c = 0; c < 40; ++c) for (int d = 0; d < 3; ++d) { cx[c][d] = 3 * gauss(); cy[c][d] = shared ? cx[c][d] : 3 * gauss(); }
            for (int i = 0; i < N; ++i) { const int c = (int)(rnd() * 40) % 40; for (int d = 0; d < 3; ++d) hx[((size_t)b * N + i) * 3 + d] = cx[c][d] + sigma * gauss(); }
            for (int i = 0; i < M; ++i) { const int c = (int)(rnd() * 40) % 40; for (int d = 0; d < 3; ++d) hy[((size_t)b * M + i) * 3 + d] = cy[c][d] + sigma * gauss(); }
        }
        if (same) hy = hx;
    }
    float *x, *y; double *part; float *loss;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&y, hy.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
    size_t wsb; fx3d_chamfer_workspace_bytes(N, M, B, 3, &wsb);
    void *ws; hipMalloc(&ws, wsb); hipMalloc(&loss, 4);
    float hl = 0;
    for (int it = 0; it < 5; ++it) fx3d_chamfer_fwd(x, N, y, M, B, 3, 1.f, 1.f, loss, &hl, nullptr, nullptr, ws, wsb, nullptr);
    const int warm = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 20;  // (warm: launches before the timed ones -- steady clocks)
    for (int it = 0; it < warm; ++it) fx3d_chamfer_fwd(x, N, y, M, B, 3, 1.f, 1.f, loss, nullptr, nullptr, nullptr, ws, wsb, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rnd = 0; rnd < (argc > 1 ? 3 : 1); ++rnd) {
        hipEventRecord(e0);
        for (int it = 0; it < reps; ++it) fx3d_chamfer_fwd(x, N, y, M, B, 3, 1.f, 1.f, loss, nullptr, nullptr, nullptr, ws, wsb, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("loss %.8f  avg per call %.3f us (%d calls)\n", hl, ms * 1000 / reps, reps);
    }
    std::vector<unsigned long long> pr(4096 * 16);
    hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_probe), pr.size() * 8);
    // s_memtime runs at 100 MHz on gfx9 (constant), report in ns*10 -> convert: ticks * 10 ns
    const char *names[] = {"start", "bbox done", "c0 image staged", "c0 main loop done", "c0 exact done", "-",
                           "c1 image staged", "c1 main loop done", "c1 exact done", "-", "-", "merge", "end"};
    int nb = getenv("PNB") ? atoi(getenv("PNB")) : 256;
    std::vector<double> d(13, 0.0);
    unsigned long long t0min = ~0ull, tmax = 0;
    for (int b = 0; b < nb; ++b) {
        const unsigned long long *q = &pr[b * 16];
        if (!q[12]) continue;
        t0min = std::min(t0min, q[0]); tmax = std::max(tmax, q[12]);
        unsigned long long prev = q[0];
        for (int k = 1; k <= 12; ++k) { if (!q[k]) continue; d[k] += (double)(q[k] - prev); prev = q[k]; }
    }
    printf("kernel span (first block start -> last block end): %llu ticks\n", tmax - t0min);
    for (int k = 1; k <= 12; ++k) printf("  %-20s avg %10.1f ticks\n", names[k], d[k] / nb);
    printf("accumulated over all launches: wave-passes %llu, with slow queries %llu, retried %llu, with run items %llu; slow queries %llu, lanes with an unusable filter %llu\n", pr[4095 * 16], pr[4095 * 16 + 1], pr[4095 * 16 + 2], pr[4095 * 16 + 3], pr[4095 * 16 + 4], pr[4095 * 16 + 5]);
    printf("(pruned launches) lane tiles run in phase 0: %llu, in phase 1: %llu -- per wave-pass of %d: %.1f + %.1f\n", pr[4095 * 16 + 6], pr[4095 * 16 + 7],
           (M + 63) / 64, pr[4095 * 16] ? (double)pr[4095 * 16 + 6] / pr[4095 * 16] : 0.0, pr[4095 * 16] ? (double)pr[4095 * 16 + 7] / pr[4095 * 16] : 0.0);
    if (getenv("RAW")) {
        std::vector<unsigned long long> p2(256 * 16);
        hipMemcpyFromSymbol(p2.data(), HIP_SYMBOL(g_probe2), p2.size() * 8);
        for (int b = 0; b < 3; ++b) {
            printf("block %d sort-phase stamps (cycles from the kernel's start):", b);
            for (int k = 0; k <= 8; ++k) if (p2[b * 16 + k]) printf(" %d:%llu", k, p2[b * 16 + k] - pr[b * 16]);
            printf("\n");
        }
    }
    if (getenv("RAW")) {
        std::vector<unsigned long long> p3(16 * 16 * 8);
        hipMemcpyFromSymbol(p3.data(), HIP_SYMBOL(g_probe3), p3.size() * 8);
        for (int b = 0; b < 2; ++b) {
            printf("block %d per wave (cycles from the kernel's start): image staged | pass 0 loop done | pass 0 exact done | pass 1 loop done | pass 1 exact done\n", b);
            for (int w = 0; w < 16; ++w) {
                printf("  wave %2d (SIMD %d):", w, w & 3);
                for (int k = 0; k < 5; ++k) printf(" %7llu", p3[(b * 16 + w) * 8 + k] - pr[b * 16]);
                printf("\n");
            }
        }
    }
    if (getenv("RAW"))  // stamps of a few blocks relative to their first one (marks need not be in index order)
        for (int b = 0; b < 3; ++b) {
            printf("block %d:", b);
            for (int k = 1; k <= 12; ++k) if (pr[b * 16 + k]) printf(" %d:%llu", k, pr[b * 16 + k] - pr[b * 16]);
            printf("\n");
        }
    {   // wall clock (constant rate, shared by all blocks): block durations and the launch's span
        int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);  // kHz
        unsigned long long a = ~0ull, e = 0; double dsum = 0, dmax = 0, late = 0;
        for (int b = 0; b < nb; ++b) { a = std::min(a, pr[b * 16 + 13]); e = std::max(e, pr[b * 16 + 14]); }
        for (int b = 0; b < nb; ++b) { const double d = (double)(pr[b * 16 + 14] - pr[b * 16 + 13]); dsum += d; dmax = std::max(dmax, d); late = std::max(late, (double)(pr[b * 16 + 13] - a)); }
        printf("wall clock %d kHz: block duration avg %.2f us max %.2f us; first start -> last end %.2f us; latest start +%.2f us\n", wrate,
               dsum / nb * 1e3 / wrate, dmax * 1e3 / wrate, (double)(e - a) * 1e3 / wrate, late * 1e3 / wrate);
    }
    {   // the end of a block: mark 11 (passes done) -> mark 9 (ticket returned) -> mark 12 (end), ordinary blocks vs the last arriver
        double a = 0, b2 = 0; int n = 0;
        for (int b = 0; b < nb; ++b) {
            const unsigned long long *q = &pr[b * 16];
            if (q[15]) printf("last arriver: block %d, passes done -> ticket %llu cycles, ticket -> end %llu cycles\n", b, q[9] - q[11], q[12] - q[9]);
            if (q[15] && b < 256 && getenv("RAW")) {
                std::vector<unsigned long long> p2(256 * 16);
                hipMemcpyFromSymbol(p2.data(), HIP_SYMBOL(g_probe2), p2.size() * 8);
                printf("  tail detail (cycles from passes done): stores issued %lld, drained %lld, barrier %lld, counter back %lld, tile known %lld, rows loaded %lld, merged %lld\n",
                       (long long)(p2[b * 16 + 10] - q[11]), (long long)(p2[b * 16 + 11] - q[11]), (long long)(p2[b * 16 + 12] - q[11]), (long long)(p2[b * 16 + 13] - q[11]),
                       (long long)(q[6] - q[11]), (long long)(p2[b * 16 + 14] - q[11]), (long long)(q[7] - q[11]));
            }
            if (q[15] && q[6] > q[11] && q[8] > q[7])  // (candidate-split runs: the tile's last subset)
                printf("  its tail from passes done (thread 0): barrier + tile counter %llu, merge %llu, block sum %llu, finalisation to the ticket %llu cycles\n",
                       q[6] - q[11], q[7] - q[6], q[8] - q[7], q[9] - q[8]);
            else { a += (double)(q[9] - q[11]); b2 += (double)(q[12] - q[9]); ++n; }
        }
        printf("other blocks: passes done -> ticket %.0f cycles, ticket -> end %.0f cycles\n", a / n, b2 / n);
    }
    {   // block durations (wall clock): quantiles, per XCD, per direction
        int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
        std::vector<double> du;
        double xs[8] = {0}, ds[2] = {0};
        for (int b = 0; b < nb; ++b) {
            const double d = (double)(pr[b * 16 + 14] - pr[b * 16 + 13]) * 1e3 / wrate;
            du.push_back(d); xs[b & 7] += d / (nb / 8);
            const int c = ((b >> 3) / 4) * 8 + (b & 7);  // (C2 plan: 4 blocks per cloud and direction)
            ds[c >= B ? 1 : 0] += d / (nb / 2);
        }
        std::sort(du.begin(), du.end());
        printf("block duration us: p0 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f p100 %.2f\n", du[0], du[nb / 10], du[nb / 2], du[nb * 9 / 10], du[nb * 99 / 100], du[nb - 1]);
        printf("per XCD mean us:"); for (int x = 0; x < 8; ++x) printf(" %.2f", xs[x]); printf("  per direction: %.2f %.2f\n", ds[0], ds[1]);
    }
    // per XCD (blocks L with equal L % 8 share a clock): span from the first start to the last end, and the starts
    for (int x = 0; x < 8; ++x) {
        unsigned long long a = ~0ull, e = 0, latest = 0;
        for (int b = x; b < nb; b += 8) { a = std::min(a, pr[b * 16]); e = std::max(e, pr[b * 16 + 12]); latest = std::max(latest, pr[b * 16]); }
        printf("xcd %d: first start -> last end %llu ticks, latest start +%llu\n", x, e - a, latest - a);
    }
    // block start/end distribution
    std::vector<unsigned long long> st, en;
    for (int b = 0; b < nb; ++b) { st.push_back(pr[b * 16] - t0min); en.push_back(pr[b * 16 + 12] - t0min); }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    printf("block start ticks: p0 %llu p25 %llu p50 %llu p75 %llu p100 %llu\n", st[0], st[nb / 4], st[nb / 2], st[3 * nb / 4], st[nb - 1]);
    printf("block end   ticks: p0 %llu p25 %llu p50 %llu p75 %llu p100 %llu\n", en[0], en[nb / 4], en[nb / 2], en[3 * nb / 4], en[nb - 1]);
    return 0;
}
