/* The drop-in boundary from plain C (what the Julia shim's @ccall lines, or any other FFI, bind): two synthetic
 * clouds -> device -> chamfer_distance forward -> loss on the host.  No torch, no Python.
 *
 *   gcc -std=c99 -I include examples/c_abi_example.c -o c_abi_example \
 *       -L flux3d.jl_amd/lib -lflux3d_hip -Wl,-rpath,$PWD/flux3d.jl_amd/lib && ./c_abi_example
 */
#include "flux3d_hip.h"

#include <stdio.h>
#include <stdlib.h>

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

int main(void) {
    printf("%s\n", fx3d_version());
    int32_t ndev = 0;
    if (fx3d_device_count(&ndev) != FX3D_OK || ndev == 0) {
        printf("no MI355X visible: nothing to run (the library has no CPU fallback)\n");
        return 0;
    }
    enum { D = 3, N = 1024, M = 1024, B = 2 };
    float *hx = (float *)malloc(sizeof(float) * D * N * B), *hy = (float *)malloc(sizeof(float) * D * M * B);
    unsigned s = 12345u; /* (D,N,B) column-major == contiguous xyz triples, batch slowest */
    for (int i = 0; i < D * N * B; ++i) { s = s * 1664525u + 1013904223u; hx[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
    for (int i = 0; i < D * M * B; ++i) { s = s * 1664525u + 1013904223u; hy[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }

    void *x = NULL, *y = NULL, *ws = NULL, *loss_dev = NULL;
    size_t wsb = 0;
    CHECK(fx3d_malloc(&x, sizeof(float) * D * N * B));
    CHECK(fx3d_malloc(&y, sizeof(float) * D * M * B));
    CHECK(fx3d_malloc(&loss_dev, sizeof(float)));
    CHECK(fx3d_memcpy_h2d(x, hx, sizeof(float) * D * N * B, NULL));
    CHECK(fx3d_memcpy_h2d(y, hy, sizeof(float) * D * M * B, NULL));
    CHECK(fx3d_chamfer_workspace_bytes(N, M, B, D, &wsb));
    CHECK(fx3d_malloc(&ws, wsb));
    float loss = 0.0f;
    CHECK(fx3d_chamfer_fwd((const float *)x, N, (const float *)y, M, B, D, 1.0f, 1.0f, (float *)loss_dev, &loss, NULL, NULL,
                           ws, wsb, NULL));
    printf("chamfer_distance(A, B) = %.8f  (B=%d clouds of %d x %d points)\n", loss, B, N, M);
    fx3d_free(ws); fx3d_free(loss_dev); fx3d_free(y); fx3d_free(x);
    free(hx); free(hy);
    return 0;
}
