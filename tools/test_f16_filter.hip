// Layout + numerics check for the 16-bit split filter: t~(i,j) = n~_i + sum_d qm~_d c~_d evaluated by ONE
// v_mfma_f32_32x32x16_f16 with 2-way fp16 splits.  Prints the max error vs float64 in scaled units.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split2(float v, _Float16 &h, _Float16 &l) { h = (_Float16)v; l = (_Float16)(v - (float)h); }
__device__ inline void split3(float v, _Float16 &a, _Float16 &b, _Float16 &c) {
    a = (_Float16)v; float r = v - (float)a; b = (_Float16)r; c = (_Float16)(r - (float)b);
}

// cand: 32 x 3 (scaled, |.|<=1), qry: 32 x 3 (qm~ = -2 q~), out: 32x32 (row=cand, col=query)
__global__ void k(const float *cand, const float *qry, float *out) {
    const int l = threadIdx.x, row = l & 31, h = l >> 5;
    _Float16 chx, clx, chy, cly, chz, clz, n1, n2, n3;
    const float cx = cand[row * 3], cy = cand[row * 3 + 1], cz = cand[row * 3 + 2];
    split2(cx, chx, clx); split2(cy, chy, cly); split2(cz, chz, clz);
    split3(cx * cx + cy * cy + cz * cz, n1, n2, n3);
    _Float16 qhx, qlx, qhy, qly, qhz, qlz;
    split2(qry[row * 3], qhx, qlx); split2(qry[row * 3 + 1], qhy, qly); split2(qry[row * 3 + 2], qhz, qlz);
    const _Float16 one = (_Float16)1.0f, zero = (_Float16)0.0f;
    h8 a, b;
    if (h == 0) {
        a = h8{chx, chx, clx, chy, chy, cly, chz, chz};
        b = h8{qhx, qlx, qhx, qhy, qly, qhy, qhz, qlz};
    } else {
        a = h8{clz, n1, n2, n3, clx, cly, clz, zero};
        b = h8{qhz, one, one, one, qlx, qly, qlz, zero};
    }
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[rr * 32 + (l & 31)] = d[r];
    }
}

int main() {
    std::vector<float> c(96), q(96), o(1024);
    float *dc, *dq, *dout;
    hipMalloc(&dc, 384); hipMalloc(&dq, 384); hipMalloc(&dout, 4096);
    unsigned s = 7;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    double worst = 0, worst_rel = 0, worst_model = 0, worst_cross = 0, worst_cross_far = 0;
    for (int trial = 0; trial < 5600; ++trial) {
        // magnitudes of the robust scaling (|c~| < 2^7, |qm~| < 2^8 for queries inside the cloud) down to a bulk 10^-5 of it
        const float csv[8] = {1.0f, 1e-2f, 0.3f, 1e-4f, 127.0f, 30.0f, 3.0f, 1e-3f};
        // (round 4: + the magnitudes of a per-query-scaled far query, sq S in [2^13, 2^14): |qm~_d| up to ~10^4)
        const float qsv[9] = {2.0f, 8.0f, 0.5f, 250.0f, 60.0f, 1e-2f, 2e-3f, 3000.0f, 9000.0f};
        const float cs = csv[trial % 8], qs = qsv[trial % 9];
        for (auto &v : c) v = (2 * rnd() - 1) * cs;
        for (auto &v : q) v = (2 * rnd() - 1) * qs;
        hipMemcpy(dc, c.data(), 384, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), 384, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dc, dq, dout);
        hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            const float cxx = c[i * 3], cyy = c[i * 3 + 1], czz = c[i * 3 + 2];
            const double n = (double)(cxx * cxx + cyy * cyy + czz * czz);  // fp32 norm as the kernel computes it
            double ref = n, mag = fabs(n);
            for (int d = 0; d < 3; ++d) { ref += (double)q[j * 3 + d] * c[i * 3 + d]; mag += fabs((double)q[j * 3 + d] * c[i * 3 + d]); }
            const double err = fabs((double)o[i * 32 + j] - ref);
            const double qsum = fabs(q[j * 3]) + fabs(q[j * 3 + 1]) + fabs(q[j * 3 + 2]);
            const double bound = 3.0 + qsum;  // scaled-units denominator: (3 + sum|qm~|)
            if (err / bound > worst) worst = err / bound;
            if (mag > 0 && err / mag > worst_rel) worst_rel = err / mag;
            const double qn = 0.25 * ((double)q[j * 3] * q[j * 3] + (double)q[j * 3 + 1] * q[j * 3 + 1] + (double)q[j * 3 + 2] * q[j * 3 + 2]);
            // the kernels' model: err <= beta (n_c + |q~|^2) + floor, floor = 2^-25 (S + 2) (subnormal fp16 pieces)
            const double excess = err - 0x1p-25 * (qsum + 2.0);
            if (excess > 0 && excess / (n + qn) > worst_model) worst_model = excess / (n + qn);
            // the refined model (round 4, far queries): err <= b1 n_c + b2 |q~| |c~| + floor -- the cross term is what a query far
            // outside the cloud (|q~| >> |c~|) really pays; (n_c + |q~|^2) over-charges it by |q~| / |c~|
            const double cross = n + sqrt(qn) * sqrt(n);
            if (excess > 0 && cross > 0 && excess / cross > worst_cross) worst_cross = excess / cross;
            if (excess > 0 && cross > 0 && qn > 64.0 * n && excess / cross > worst_cross_far) worst_cross_far = excess / cross;
        }
    }
    printf("max |err| / (3 + sum|qm~|) = %.3e  (2^%.2f)   max err/sum|terms| = %.3e\n", worst, log2(worst), worst_rel);
    printf("max (|err| - 2^-25 (S + 2)) / (n_c + |q~|^2) = %.3e  (2^%.2f);  the kernels use beta = 2^-18\n", worst_model, log2(worst_model));
    printf("max (|err| - 2^-25 (S + 2)) / (n_c + |q~| |c~|) = %.3e  (2^%.2f) over all pairs, %.3e (2^%.2f) over far pairs (|q~| > 8 |c~|)\n",
           worst_cross, log2(worst_cross), worst_cross_far, log2(worst_cross_far));
    return 0;
}
