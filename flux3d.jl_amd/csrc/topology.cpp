// Host-side mesh topology: unique edge list and the uniform Laplacian in CSR.
//
// The reference builds both on the host, once per mesh, and caches them for the lifetime of the
// TriMesh (src/rep/mesh.jl:87-97, 907-1002): faces never change under the optimisation loops that
// use this path (examples/fit_mesh.jl:106-116), so this is integer set-up work, not hot-path
// work.  The results are uploaded once and reused by fx3d_edge_loss / fx3d_laplacian_loss.
#include <algorithm>
#include <vector>

#include "fx3d_common.h"

using namespace fx3d;

extern "C" {

// _compute_edges_packed, src/rep/mesh.jl:907-955.
fx3d_status fx3d_build_edges_packed(const int64_t *faces, int64_t F, int64_t V, int32_t index_base,
                                    int64_t *edges_out, int64_t *faces_to_edges, int64_t *E_out) {
    FX3D_REQUIRE(faces && edges_out && E_out, "fx3d_build_edges_packed: null pointer");
    FX3D_REQUIRE(F > 0 && V > 0, "fx3d_build_edges_packed: bad sizes F=%lld V=%lld", (long long)F, (long long)V);
    FX3D_REQUIRE(index_base == 0 || index_base == 1, "fx3d_build_edges_packed: index_base must be 0 or 1");
    // integer hash (V+1)*v0 + v1 with v0 <= v1, 1-based ids as in :928-929
    const uint64_t Vh = (uint64_t)V + 1;
    std::vector<uint64_t> h((size_t)3 * F);
    for (int64_t f = 0; f < F; ++f) {
        int64_t v[3];
        for (int t = 0; t < 3; ++t) {
            v[t] = faces[3 * f + t] - index_base;
            FX3D_REQUIRE(v[t] >= 0 && v[t] < V, "fx3d_build_edges_packed: face %lld has vertex id %lld outside 1:%lld",
                         (long long)f, (long long)faces[3 * f + t], (long long)V);
        }
        const int64_t pr[3][2] = {{v[0], v[1]}, {v[1], v[2]}, {v[2], v[0]}};  // e12, e23, e31
        for (int e = 0; e < 3; ++e) {
            const uint64_t lo = (uint64_t)std::min(pr[e][0], pr[e][1]) + 1;
            const uint64_t hi = (uint64_t)std::max(pr[e][0], pr[e][1]) + 1;
            h[(size_t)e * F + f] = Vh * lo + hi;
        }
    }
    std::vector<uint64_t> u(h);
    std::sort(u.begin(), u.end());                       // sort!  (:932)
    u.erase(std::unique(u.begin(), u.end()), u.end());   // unique! (:933)
    const int64_t E = (int64_t)u.size();
    for (int64_t e = 0; e < E; ++e) {  // (E,2) column-major
        edges_out[e] = (int64_t)(u[e] / Vh) - 1 + index_base;
        edges_out[E + e] = (int64_t)(u[e] % Vh) - 1 + index_base;
    }
    if (faces_to_edges) {  // (F,3) column-major, columns (e23, e31, e12)  (:946)
        const int col_of[3] = {2, 0, 1};
        for (int e = 0; e < 3; ++e)
            for (int64_t f = 0; f < F; ++f) {
                const int64_t pos = std::lower_bound(u.begin(), u.end(), h[(size_t)e * F + f]) - u.begin();
                faces_to_edges[(size_t)col_of[e] * F + f] = pos + index_base;
            }
    }
    *E_out = E;
    return FX3D_OK;
}

// _compute_laplacian_packed, src/rep/mesh.jl:957-1002, emitted as CSR with ascending columns.
// Duplicate (i,j) entries are summed, as SparseArrays.sparse does (only degenerate faces with a
// repeated vertex produce them).
fx3d_status fx3d_build_laplacian_csr(const int64_t *edges, int64_t E, int64_t V, int32_t index_base,
                                     int32_t *rowptr, int32_t *colind, float *vals,
                                     int64_t *nnz_out) {
    FX3D_REQUIRE(edges && rowptr && colind && vals && nnz_out, "fx3d_build_laplacian_csr: null pointer");
    FX3D_REQUIRE(E >= 0 && V > 0 && 2 * E + V < (1ll << 31), "fx3d_build_laplacian_csr: bad sizes");
    std::vector<int64_t> deg((size_t)V, 0);
    for (int64_t e = 0; e < E; ++e) {
        const int64_t i = edges[e] - index_base, j = edges[E + e] - index_base;
        FX3D_REQUIRE(i >= 0 && i < V && j >= 0 && j < V, "fx3d_build_laplacian_csr: edge %lld out of range", (long long)e);
        deg[(size_t)i]++;  // A = sparse([e1;e2],[e2;e1],1): row sums (:976-985)
        deg[(size_t)j]++;
    }
    struct Trip { int64_t r, c; float v; int64_t ord; };
    std::vector<Trip> t;
    t.reserve((size_t)(2 * E + V));
    auto inv = [&](int64_t i) { return deg[(size_t)i] > 0 ? (float)(1.0 / (double)deg[(size_t)i]) : (float)deg[(size_t)i]; };
    int64_t n = 0;
    for (int64_t e = 0; e < E; ++e) {  // (e1,e2,deg1)
        const int64_t i = edges[e] - index_base, j = edges[E + e] - index_base;
        t.push_back({i, j, inv(i), n++});
    }
    for (int64_t e = 0; e < E; ++e) {  // (e2,e1,deg2)
        const int64_t i = edges[e] - index_base, j = edges[E + e] - index_base;
        t.push_back({j, i, inv(j), n++});
    }
    for (int64_t i = 0; i < V; ++i) t.push_back({i, i, -1.0f, n++});  // diag (:989-997)
    std::sort(t.begin(), t.end(), [](const Trip &a, const Trip &b) {
        if (a.r != b.r) return a.r < b.r;
        if (a.c != b.c) return a.c < b.c;
        return a.ord < b.ord;
    });
    std::fill(rowptr, rowptr + V + 1, 0);
    int64_t nnz = 0;
    for (size_t k = 0; k < t.size(); ++k) {
        if (k > 0 && t[k].r == t[k - 1].r && t[k].c == t[k - 1].c) {
            vals[nnz - 1] = vals[nnz - 1] + t[k].v;
        } else {
            colind[nnz] = (int32_t)t[k].c;
            vals[nnz] = t[k].v;
            rowptr[t[k].r + 1]++;
            ++nnz;
        }
    }
    for (int64_t i = 0; i < V; ++i) rowptr[i + 1] += rowptr[i];
    *nnz_out = nnz;
    return FX3D_OK;
}

// Vertex -> (face, corner) table of a padded batch, for the ordered sampling adjoint (sample_gather.h): a CSR over the
// vertices of every mesh, entries face * 4 + corner in ascending order.  Host arrays: faces_padded (3,Fmax,B) int32 0-based
// mesh-local (entries of padding faces are ignored), faces_len (B); vf_rowptr (Vmax+1,B), vf_ent (3*Fmax,B) (the entries of
// mesh b fill the first vf_rowptr[Vmax,b] slots of its column).  Like the edge list and the Laplacian it is built once per
// mesh topology (src/rep/mesh.jl:87-97: faces never change under the loops that differentiate through sample_points).
fx3d_status fx3d_build_vertex_faces(const int32_t *faces_padded, const int32_t *faces_len, int32_t Vmax, int32_t Fmax,
                                    int32_t B, int32_t *vf_rowptr, int32_t *vf_ent) {
    FX3D_REQUIRE(faces_padded && faces_len && vf_rowptr && vf_ent, "fx3d_build_vertex_faces: null pointer");
    FX3D_REQUIRE(Vmax > 0 && Fmax > 0 && B > 0 && Fmax < (1 << 29), "fx3d_build_vertex_faces: bad sizes");
    for (int32_t b = 0; b < B; ++b) {
        const int32_t *fb = faces_padded + (size_t)b * Fmax * 3;
        int32_t *rp = vf_rowptr + (size_t)b * (Vmax + 1), *en = vf_ent + (size_t)b * 3 * Fmax;
        const int32_t L = faces_len[b];
        FX3D_REQUIRE(L >= 0 && L <= Fmax, "fx3d_build_vertex_faces: faces_len[%d] = %d outside 0:%d", b, L, Fmax);
        std::fill(rp, rp + Vmax + 1, 0);
        for (int32_t f = 0; f < L; ++f)
            for (int t = 0; t < 3; ++t) {
                const int32_t v = fb[3 * (size_t)f + t];
                FX3D_REQUIRE(v >= 0 && v < Vmax, "fx3d_build_vertex_faces: mesh %d face %d has vertex id %d outside 0:%d", b, f, v, Vmax - 1);
                rp[v + 1]++;
            }
        for (int32_t v = 0; v < Vmax; ++v) rp[v + 1] += rp[v];
        std::vector<int32_t> cur(rp, rp + Vmax);
        for (int32_t f = 0; f < L; ++f)          // ascending (face, corner): a stable bucket fill keeps that order per vertex
            for (int t = 0; t < 3; ++t) en[cur[fb[3 * (size_t)f + t]]++] = f * 4 + t;
    }
    return FX3D_OK;
}

}  // extern "C"
