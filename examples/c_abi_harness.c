/* The reference's metric harness (benchmarks/metrics.jl:17-63: chamfer_distance forward and forward + gradient at n = 2^6 .. 2^14
 * points, one cloud pair, A == B collinear points) timed from plain C against the shared library -- the call floor an FFI caller
 * (Julia's @ccall, cgo, JNI ...) sees, without Python's ctypes in the path.  Per size: microseconds per call of
 *   forward   : fx3d_chamfer_fwd, loss left on the device (back-to-back calls on one stream, one synchronisation at the end),
 *   value+grad: fx3d_chamfer_fwd_bwd (forward with indices + the adjoint queued by one call),
 *   blocking  : fx3d_chamfer_fwd with the loss copied to the host (the stream is synchronised every call).
 *
 *   gcc -std=c99 -O2 -I include examples/c_abi_harness.c -o c_abi_harness \
 *       -L flux3d.jl_amd/lib -lflux3d_hip -Wl,-rpath,$PWD/flux3d.jl_amd/lib && ./c_abi_harness
 */
#define _POSIX_C_SOURCE 199309L
#include "flux3d_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define CHECK(call)                                                           \
    do {                                                                      \
        fx3d_status rc_ = (call);                                             \
        if (rc_ != FX3D_OK) {                                                 \
            char msg_[512];                                                   \
            fx3d_last_error(msg_, sizeof msg_);                               \
            fprintf(stderr, "%s failed (%d): %s\n", #call, (int)rc_, msg_);   \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}

int main(void) {
    int32_t ndev = 0;
    if (fx3d_device_count(&ndev) != FX3D_OK || ndev == 0) {
        printf("no MI355X visible: nothing to run (the library has no CPU fallback)\n");
        return 0;
    }
    enum { D = 3, B = 1, REPS = 2000 };
    printf("{\"what\": \"benchmarks/metrics.jl sizes from plain C (us per call, %d calls)\", \"rows\": [", REPS);
    for (int e = 6; e <= 14; e += 2) {
        const int n = 1 << e;
        float *h = (float *)malloc(sizeof(float) * D * n);
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < D; ++d) h[i * D + d] = (float)(i + 1) / (float)n; /* benchmarks/metrics.jl:11-15 */
        void *x = NULL, *y = NULL, *ws = NULL, *loss_dev = NULL, *gx = NULL, *gy = NULL;
        size_t wsb = 0, wsb2 = 0;
        CHECK(fx3d_malloc(&x, sizeof(float) * D * n));
        CHECK(fx3d_malloc(&y, sizeof(float) * D * n));
        CHECK(fx3d_malloc(&gx, sizeof(float) * D * n));
        CHECK(fx3d_malloc(&gy, sizeof(float) * D * n));
        CHECK(fx3d_malloc(&loss_dev, sizeof(float)));
        CHECK(fx3d_memcpy_h2d(x, h, sizeof(float) * D * n, NULL));
        CHECK(fx3d_memcpy_h2d(y, h, sizeof(float) * D * n, NULL));
        CHECK(fx3d_chamfer_workspace_bytes(n, n, B, D, &wsb));
        CHECK(fx3d_chamfer_fwd_bwd_workspace_bytes(n, n, B, D, &wsb2));
        if (wsb2 > wsb) wsb = wsb2;
        CHECK(fx3d_malloc(&ws, wsb));
        float loss = 0.0f;
        for (int i = 0; i < 20; ++i)
            CHECK(fx3d_chamfer_fwd((const float *)x, n, (const float *)y, n, B, D, 1.0f, 1.0f, (float *)loss_dev, NULL, NULL, NULL, ws, wsb, NULL));
        CHECK(fx3d_stream_sync(NULL));
        double t0 = now_us();
        for (int i = 0; i < REPS; ++i)
            CHECK(fx3d_chamfer_fwd((const float *)x, n, (const float *)y, n, B, D, 1.0f, 1.0f, (float *)loss_dev, NULL, NULL, NULL, ws, wsb, NULL));
        CHECK(fx3d_stream_sync(NULL));
        const double fwd = (now_us() - t0) / REPS;
        for (int i = 0; i < 20; ++i)
            CHECK(fx3d_chamfer_fwd_bwd((const float *)x, n, (const float *)y, n, B, D, 1.0f, 1.0f, 1.0f, B, (float *)loss_dev, NULL, (float *)gx,
                                       (float *)gy, NULL, NULL, ws, wsb, NULL));
        CHECK(fx3d_stream_sync(NULL));
        t0 = now_us();
        for (int i = 0; i < REPS; ++i)
            CHECK(fx3d_chamfer_fwd_bwd((const float *)x, n, (const float *)y, n, B, D, 1.0f, 1.0f, 1.0f, B, (float *)loss_dev, NULL, (float *)gx,
                                       (float *)gy, NULL, NULL, ws, wsb, NULL));
        CHECK(fx3d_stream_sync(NULL));
        const double tot = (now_us() - t0) / REPS;
        t0 = now_us();
        for (int i = 0; i < REPS / 4; ++i)
            CHECK(fx3d_chamfer_fwd((const float *)x, n, (const float *)y, n, B, D, 1.0f, 1.0f, (float *)loss_dev, &loss, NULL, NULL, ws, wsb, NULL));
        const double blk = (now_us() - t0) / (REPS / 4);
        printf("%s{\"n\": %d, \"forward_us\": %.2f, \"value_and_grad_us\": %.2f, \"forward_blocking_us\": %.2f, \"loss\": %.3g}", e > 6 ? ", " : "", n, fwd,
               tot, blk, loss);
        fx3d_free(ws); fx3d_free(loss_dev); fx3d_free(gy); fx3d_free(gx); fx3d_free(y); fx3d_free(x);
        free(h);
    }
    printf("]}\n");
    return 0;
}
