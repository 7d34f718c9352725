// C ABI of the host-side mesh operations (csrc/remesh_host.h): sculpt_mesh_subdivide / _decimate / _remesh_botsch.
// HOST pointers in, an opaque result object out (the output size is not known beforehand).  No GPU work here: the reference
// runs this step on the CPU through gpytoolbox (StableFast/sf3d/models/mesh.py:175-237); see remesh_host.h for the algorithms.
#include <exception>
#include <memory>
#include <new>

#include "common.h"
#include "remesh_host.h"

struct sculpt_host_mesh {
    std::vector<double> V;
    std::vector<int32_t> F;
};

using namespace sculpt;

// No C++ exception may cross the C ABI (it would terminate the host process -- Blender): out of memory, std::length_error from
// a vector that cannot grow, std::system_error from a thread that cannot be created, ... all become status codes.
#define SCULPT_MESH_CATCH(who)                                                     \
    catch (const std::bad_alloc &) { SC_REQUIRE(false, who ": out of memory"); }    \
    catch (const std::exception &e) { SC_REQUIRE(false, who ": %s", e.what()); }    \
    catch (...) { SC_REQUIRE(false, who ": unknown C++ exception"); }

namespace {
int check_in(const double *V, size_t nv, const int32_t *F, size_t nf, sculpt_host_mesh_t **out, const char *who) {
    SC_REQUIRE(out, "%s: null result pointer", who);
    *out = nullptr;
    SC_REQUIRE((V || nv == 0) && (F || nf == 0), "%s: null input", who);
    SC_REQUIRE(nv < (size_t)1 << 31 && nf < (size_t)1 << 31, "%s: mesh too large for int32 indices", who);
    return 0;
}
}  // namespace

extern "C" int sculpt_mesh_subdivide(const double *V, size_t nv, const int32_t *F, size_t nf, int iters, sculpt_host_mesh_t **out) {
    if (int rc = check_in(V, nv, F, nf, out, "mesh_subdivide")) return rc;
    SC_REQUIRE(iters >= 0 && iters <= 12, "mesh_subdivide: iters=%d out of range", iters);
    for (size_t i = 0; i < 3 * nf; ++i) SC_REQUIRE(F[i] >= 0 && (size_t)F[i] < nv, "mesh_subdivide: face index out of range");
    // 4^iters faces: refuse what cannot be indexed
    double faces = (double)nf;
    for (int i = 0; i < iters; ++i) faces *= 4;
    SC_REQUIRE(faces < 1.5e9, "mesh_subdivide: %zu faces x 4^%d does not fit int32 indices", nf, iters);
    try {
        std::unique_ptr<sculpt_host_mesh> m(new sculpt_host_mesh);
        m->V.assign(V, V + 3 * nv);
        m->F.assign(F, F + 3 * nf);
        for (int i = 0; i < iters; ++i) remesh::subdivide_once(m->V, m->F);
        *out = m.release();
    }
    SCULPT_MESH_CATCH("mesh_subdivide")
    return 0;
}

extern "C" int sculpt_mesh_decimate(const double *V, size_t nv, const int32_t *F, size_t nf, size_t target_faces,
                                    sculpt_host_mesh_t **out) {
    if (int rc = check_in(V, nv, F, nf, out, "mesh_decimate")) return rc;
    try {
        remesh::Mesh M;
        const std::string err = M.build(V, nv, F, nf);
        SC_REQUIRE(err.empty(), "mesh_decimate: %s", err.c_str());
        remesh::decimate(M, target_faces);
        std::unique_ptr<sculpt_host_mesh> m(new sculpt_host_mesh);
        M.compact(m->V, m->F);
        *out = m.release();
    }
    SCULPT_MESH_CATCH("mesh_decimate")
    return 0;
}

extern "C" int sculpt_mesh_remesh_botsch(const double *V, size_t nv, const int32_t *F, size_t nf, int iters, double h, int project,
                                         sculpt_host_mesh_t **out) {
    if (int rc = check_in(V, nv, F, nf, out, "mesh_remesh_botsch")) return rc;
    SC_REQUIRE(iters >= 0 && iters <= 1000, "mesh_remesh_botsch: iters=%d out of range", iters);
    SC_REQUIRE(h == h, "mesh_remesh_botsch: h is NaN");
    try {
        remesh::Mesh M;
        const std::string err = M.build(V, nv, F, nf);
        SC_REQUIRE(err.empty(), "mesh_remesh_botsch: %s", err.c_str());
        remesh::remesh_botsch(M, iters, h, project != 0);
        std::unique_ptr<sculpt_host_mesh> m(new sculpt_host_mesh);
        M.compact(m->V, m->F);
        *out = m.release();
    }
    SCULPT_MESH_CATCH("mesh_remesh_botsch")
    return 0;
}

extern "C" size_t sculpt_mesh_num_vertices(const sculpt_host_mesh_t *m) { return m ? m->V.size() / 3 : 0; }
extern "C" size_t sculpt_mesh_num_faces(const sculpt_host_mesh_t *m) { return m ? m->F.size() / 3 : 0; }

extern "C" int sculpt_mesh_read(const sculpt_host_mesh_t *m, double *V, int32_t *F) {
    SC_REQUIRE(m, "mesh_read: null mesh");
    SC_REQUIRE((V || m->V.empty()) && (F || m->F.empty()), "mesh_read: null output");
    if (!m->V.empty()) memcpy(V, m->V.data(), m->V.size() * sizeof(double));
    if (!m->F.empty()) memcpy(F, m->F.data(), m->F.size() * sizeof(int32_t));
    return 0;
}

extern "C" void sculpt_mesh_free(sculpt_host_mesh_t *m) { delete m; }
