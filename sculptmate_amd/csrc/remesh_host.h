// Triangle-mesh subdivision, decimation and isotropic remeshing on the HOST (plain C++17, no HIP: this header also builds
// with g++ -fsanitize=address, tests/native/asan_remesh.cpp).
//
// Replaces the three gpytoolbox calls of the reference's Mesh.triangle_remesh
// (/root/reference/StableFast/sf3d/models/mesh.py:175-237):
//   gpytoolbox.subdivide(v, f, iters)        -> subdivide():      midpoint ("upsample") 1 -> 4 subdivision
//   gpytoolbox.decimate(v, f, face_ratio)    -> decimate():       libigl's default decimation -- shortest edge first,
//                                                                 collapse to the midpoint, link condition, boundary kept
//                                                                 manifold through a virtual vertex at infinity
//   gpytoolbox.remesh_botsch(v, f, i, h)     -> remesh_botsch():  Botsch & Kobbelt 2004, "A remeshing approach to
//                                                                 multiresolution modeling": per iteration split edges
//                                                                 longer than 4/3 h, collapse edges shorter than 4/5 h,
//                                                                 flip edges towards valence 6 (4 on the boundary),
//                                                                 tangential relaxation, projection onto the input surface
// gpytoolbox (and libigl under it) are NOT in the reference tree and not installed in the build image: PARITY UNPINNED.
// The algorithms are restated from their publications; the tests check invariants (manifoldness, Euler characteristic,
// orientation, target counts, edge-length band, distance to the input surface), not vertex-for-vertex equality.
//
// One dynamic-mesh structure serves all three: faces with tombstones + per-vertex incident-face lists; every local
// operation (collapse, split, flip) costs O(valence).  Everything is sequential and deterministic.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace sculpt {
namespace remesh {

struct V3 {
    double x, y, z;
};
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }

// int list with 8 inline slots (the typical valence is 6: no heap allocation for most vertices)
class IntList {
  public:
    IntList() : n_(0), cap_(8), heap_(nullptr) {}
    IntList(const IntList &o) : n_(0), cap_(8), heap_(nullptr) { assign(o); }
    IntList &operator=(const IntList &o) {
        if (this != &o) assign(o);
        return *this;
    }
    ~IntList() { delete[] heap_; }
    int size() const { return n_; }
    const int *begin() const { return data(); }
    const int *end() const { return data() + n_; }
    int operator[](int i) const { return data()[i]; }
    void clear() { n_ = 0; }
    void push(int v) {
        if (n_ == cap_) grow();
        data()[n_++] = v;
    }
    void remove(int v) {  // first occurrence; order is not kept
        int *d = data();
        for (int i = 0; i < n_; ++i)
            if (d[i] == v) {
                d[i] = d[--n_];
                return;
            }
    }

  private:
    int *data() { return heap_ ? heap_ : inl_; }
    const int *data() const { return heap_ ? heap_ : inl_; }
    void grow() {
        int *h = new int[2 * cap_];
        std::memcpy(h, data(), sizeof(int) * n_);
        delete[] heap_;
        heap_ = h;
        cap_ *= 2;
    }
    void assign(const IntList &o) {
        n_ = 0;
        for (int v : o) push(v);
    }
    int n_, cap_;
    int inl_[8];
    int *heap_;
};

struct Mesh {
    std::vector<V3> P;
    std::vector<std::array<int, 3>> F;
    std::vector<uint8_t> valive, falive;
    std::vector<IntList> vf;  // faces incident to a vertex
    std::vector<uint8_t> bnd;  // vertex lies on the boundary (or on a non-manifold edge): computed by build(), kept by the operations
    size_t faces_alive = 0;

    // ---- construction / output -------------------------------------------------------------------------------------
    // returns "" or an error text.  Faces with a repeated or out-of-range index are an error (the reference hands over
    // marching-tetrahedra output, which has neither).
    std::string build(const double *V, size_t nv, const int32_t *Fi, size_t nf) {
        P.resize(nv);
        for (size_t i = 0; i < nv; ++i) P[i] = {V[3 * i], V[3 * i + 1], V[3 * i + 2]};
        for (size_t i = 0; i < nv; ++i)
            if (!(std::isfinite(P[i].x) && std::isfinite(P[i].y) && std::isfinite(P[i].z))) return "non-finite vertex position";
        F.resize(nf);
        valive.assign(nv, 1);
        falive.assign(nf, 1);
        vf.assign(nv, IntList());
        for (size_t f = 0; f < nf; ++f) {
            const int a = Fi[3 * f], b = Fi[3 * f + 1], c = Fi[3 * f + 2];
            if (a < 0 || b < 0 || c < 0 || (size_t)a >= nv || (size_t)b >= nv || (size_t)c >= nv) return "face index out of range";
            if (a == b || b == c || a == c) return "degenerate face (repeated vertex index)";
            F[f] = {a, b, c};
            vf[a].push((int)f);
            vf[b].push((int)f);
            vf[c].push((int)f);
        }
        faces_alive = nf;
        bnd.resize(nv);
        for (size_t v = 0; v < nv; ++v) bnd[v] = scan_boundary_vertex((int)v);
        return "";
    }
    // live vertices that are still referenced and live faces, both in their original relative order
    void compact(std::vector<double> &Vo, std::vector<int32_t> &Fo) const {
        std::vector<int> id(P.size(), -1);
        int n = 0;
        for (size_t v = 0; v < P.size(); ++v)
            if (valive[v] && vf[v].size() > 0) id[v] = n++;
        Vo.resize(3 * (size_t)n);
        for (size_t v = 0; v < P.size(); ++v)
            if (id[v] >= 0) {
                Vo[3 * (size_t)id[v]] = P[v].x;
                Vo[3 * (size_t)id[v] + 1] = P[v].y;
                Vo[3 * (size_t)id[v] + 2] = P[v].z;
            }
        Fo.clear();
        Fo.reserve(3 * faces_alive);
        for (size_t f = 0; f < F.size(); ++f)
            if (falive[f])
                for (int k = 0; k < 3; ++k) Fo.push_back(id[F[f][k]]);
    }

    // ---- local queries ---------------------------------------------------------------------------------------------
    static int index_in(const std::array<int, 3> &f, int v) { return f[0] == v ? 0 : (f[1] == v ? 1 : (f[2] == v ? 2 : -1)); }
    // faces containing both u and v: count (may exceed 2 on a non-manifold edge; the first two are stored)
    int edge_faces(int u, int v, int out[2]) const {
        int n = 0;
        for (int f : vf[u])
            if (index_in(F[f], v) >= 0) {
                if (n < 2) out[n] = f;
                ++n;
            }
        return n;
    }
    int third(int f, int u, int v) const {
        const auto &t = F[f];
        return t[0] != u && t[0] != v ? t[0] : (t[1] != u && t[1] != v ? t[1] : t[2]);
    }
    bool directed(int f, int u, int v) const {  // does f run u -> v ?
        const int k = index_in(F[f], u);
        return F[f][(k + 1) % 3] == v;
    }
    void neighbours(int u, std::vector<int> &out) const {
        out.clear();
        for (int f : vf[u])
            for (int k = 0; k < 3; ++k) {
                const int w = F[f][k];
                if (w != u && std::find(out.begin(), out.end(), w) == out.end()) out.push_back(w);
            }
    }
    // a vertex is on the boundary (or on a non-manifold edge) when one of its edges does not have exactly two faces.
    // Collapses, splits and flips keep this property computable locally: a collapse merges the flags (the link condition
    // never lets a hole close), a split vertex inherits its edge's, a flip touches interior edges only.
    bool is_boundary_vertex(int u) const { return bnd[u] != 0; }
    bool scan_boundary_vertex(int u) const {
        // every neighbour must appear in exactly two incident faces
        int nb[64], cnt[64], n = 0;
        for (int f : vf[u])
            for (int k = 0; k < 3; ++k) {
                const int w = F[f][k];
                if (w == u) continue;
                int i = 0;
                while (i < n && nb[i] != w) ++i;
                if (i == n) {
                    if (n == 64) return true;  // absurd valence: treat as a feature
                    nb[n] = w;
                    cnt[n++] = 0;
                }
                ++cnt[i];
            }
        for (int i = 0; i < n; ++i)
            if (cnt[i] != 2) return true;
        return n == 0;
    }
    V3 face_normal_raw(int f) const { return cross(P[F[f][1]] - P[F[f][0]], P[F[f][2]] - P[F[f][0]]); }
    double edge_len(int u, int v) const { return norm(P[u] - P[v]); }

    // ---- local operations ------------------------------------------------------------------------------------------
    // Link condition for collapsing edge (u, v) (Dey et al. 1999; libigl's edge_collapse_is_valid with the boundary
    // closed by a virtual vertex at infinity): the common neighbours of u and v are exactly the apexes of the faces on
    // the edge -- with infinity counting as a common neighbour of two boundary vertices and as the second apex of a
    // boundary edge.
    bool can_collapse(int u, int v, std::vector<int> &nu, std::vector<int> &nv) const {
        int ef[2];
        const int nef = edge_faces(u, v, ef);
        if (nef != 1 && nef != 2) return false;
        neighbours(u, nu);
        neighbours(v, nv);
        int common = 0;
        for (int w : nu)
            if (std::find(nv.begin(), nv.end(), w) != nv.end()) ++common;
        if (common != nef) return false;
        if (nef == 2 && is_boundary_vertex(u) && is_boundary_vertex(v)) return false;  // infinity would be a third common neighbour
        if (nef == 2) {
            // the links must not share an EDGE either: faces (u, a, b) and (v, a, b) over the two apexes (a tetrahedron, or a
            // tetrahedral "ear") would collapse into a two-faced pillow
            const int a = third(ef[0], u, v), b = third(ef[1], u, v);
            bool at_u = false, at_v = false;
            for (int f : vf[a])
                if (index_in(F[f], b) >= 0) {
                    at_u = at_u || index_in(F[f], u) >= 0;
                    at_v = at_v || index_in(F[f], v) >= 0;
                }
            if (at_u && at_v) return false;
        } else {
            // boundary edge: its second "face" is (u, v, infinity).  The same shared-link-edge rule with infinity as the second
            // apex: if (u, a) and (v, a) are boundary edges too, the faces (u, a, inf) and (v, a, inf) exist -- an isolated
            // triangle (or a fan tip), whose collapse would make the component disappear (libigl refuses it the same way)
            const int a = third(ef[0], u, v);
            int tmp[2];
            if (edge_faces(u, a, tmp) == 1 && edge_faces(v, a, tmp) == 1) return false;
        }
        return true;
    }
    // remove u, keep v at position p.  Caller has checked can_collapse.
    void collapse(int u, int v, V3 p) {
        int ef[2];
        const int nef = std::min(2, edge_faces(u, v, ef));
        for (int i = 0; i < nef; ++i) {
            const int f = ef[i];
            for (int k = 0; k < 3; ++k) vf[F[f][k]].remove(f);
            falive[f] = 0;
            --faces_alive;
        }
        for (int f : vf[u]) {
            F[f][index_in(F[f], u)] = v;
            vf[v].push(f);
        }
        vf[u].clear();
        valive[u] = 0;
        bnd[v] = bnd[v] | bnd[u];
        P[v] = p;
    }
    // insert a vertex at p on edge (u, v); returns its index
    int split(int u, int v, V3 p) {
        int ef[2];
        const int nef = std::min(2, edge_faces(u, v, ef));
        const int m = (int)P.size();
        P.push_back(p);
        valive.push_back(1);
        vf.emplace_back();
        bnd.push_back(nef != 2);
        for (int i = 0; i < nef; ++i) {
            const int f = ef[i];
            const int w = third(f, u, v);
            std::array<int, 3> g = F[f];  // second half: u replaced by m
            g[index_in(g, u)] = m;
            F[f][index_in(F[f], v)] = m;  // first half: v replaced by m
            const int f2 = (int)F.size();
            F.push_back(g);
            falive.push_back(1);
            ++faces_alive;
            vf[v].remove(f);
            vf[v].push(f2);
            vf[w].push(f2);
            vf[m].push(f);
            vf[m].push(f2);
        }
        return m;
    }
    // flip the edge (u, v) shared by exactly two consistently oriented faces into (a, b); false if not possible
    bool flip(int u, int v) {
        int ef[2];
        if (edge_faces(u, v, ef) != 2) return false;
        int f1 = ef[0], f2 = ef[1];
        if (!directed(f1, u, v)) std::swap(f1, f2);
        if (!directed(f1, u, v) || !directed(f2, v, u)) return false;  // inconsistent orientation
        const int a = third(f1, u, v), b = third(f2, u, v);
        if (a == b) return false;
        int tmp[2];
        if (edge_faces(a, b, tmp) != 0) return false;  // the new edge exists already
        F[f1] = {u, b, a};
        F[f2] = {b, v, a};
        vf[v].remove(f1);
        vf[u].remove(f2);
        vf[b].push(f1);
        vf[a].push(f2);
        return true;
    }

    // unique undirected edges (u < v) of the live faces
    void edges(std::vector<std::pair<int, int>> &E) const {
        E.clear();
        E.reserve(faces_alive * 3 / 2 + 16);
        for (size_t u = 0; u < P.size(); ++u) {
            if (!valive[u]) continue;
            const size_t first = E.size();
            for (int f : vf[u])
                for (int k = 0; k < 3; ++k) {
                    const int w = F[f][k];
                    if ((size_t)w <= u) continue;
                    bool seen = false;
                    for (size_t i = first; i < E.size(); ++i)
                        if (E[i].second == w) {
                            seen = true;
                            break;
                        }
                    if (!seen) E.emplace_back((int)u, w);
                }
        }
    }
    double mean_edge_length() const {
        // gpytoolbox averages over HALF-edges (every face contributes its three sides)
        double s = 0;
        size_t n = 0;
        for (size_t f = 0; f < F.size(); ++f)
            if (falive[f])
                for (int k = 0; k < 3; ++k) {
                    s += edge_len(F[f][k], F[f][(k + 1) % 3]);
                    ++n;
                }
        return n ? s / (double)n : 0.0;
    }
};

// ---- subdivide ---------------------------------------------------------------------------------------------------------
// One round of midpoint subdivision: a new vertex per edge (appended after the old ones, in order of first appearance over
// the faces), each face (a, b, c) -> (a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca); orientation kept.
static inline void subdivide_once(std::vector<double> &V, std::vector<int32_t> &F) {
    const size_t nf = F.size() / 3;
    std::unordered_map<uint64_t, int32_t> mid;
    mid.reserve(nf * 2);
    auto midpoint = [&](int32_t a, int32_t b) -> int32_t {
        const uint64_t key = ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b);
        auto it = mid.find(key);
        if (it != mid.end()) return it->second;
        const int32_t m = (int32_t)(V.size() / 3);
        for (int k = 0; k < 3; ++k) V.push_back(0.5 * (V[3 * (size_t)a + k] + V[3 * (size_t)b + k]));
        mid.emplace(key, m);
        return m;
    };
    std::vector<int32_t> G;
    G.reserve(F.size() * 4);
    for (size_t f = 0; f < nf; ++f) {
        const int32_t a = F[3 * f], b = F[3 * f + 1], c = F[3 * f + 2];
        const int32_t ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
        const int32_t t[12] = {a, ab, ca, b, bc, ab, c, ca, bc, ab, bc, ca};
        G.insert(G.end(), t, t + 12);
    }
    F.swap(G);
}

// ---- decimate ----------------------------------------------------------------------------------------------------------
// Shortest edge first, collapse to the midpoint, until at most `target_faces` faces are left or no edge can be collapsed
// (libigl decimate with its default cost / placement and max_faces stopping condition).  An edge whose collapse would break
// the link condition is skipped, like libigl's "cost = infinity, try the next".
static inline void decimate(Mesh &M, size_t target_faces) {
    struct Item {
        double cost;
        int u, v;
        uint32_t su, sv;  // vertex stamps at insertion: the entry is stale once either endpoint changed
        bool operator<(const Item &o) const {  // min-heap through std::priority_queue; ties by indices for determinism
            if (cost != o.cost) return cost > o.cost;
            if (u != o.u) return u > o.u;
            return v > o.v;
        }
    };
    std::vector<uint32_t> stamp(M.P.size(), 0);
    std::vector<int> nu, nv;
    std::vector<std::pair<int, int>> E;
    // An edge that fails the link condition leaves the heap; it can become collapsible again when two of its common
    // neighbours merge elsewhere, which re-inserts only the merged vertex's own edges: when the heap runs dry above the
    // target, it is rebuilt from the current edges as long as the previous pass still collapsed something.
    bool progress = true;
    while (M.faces_alive > target_faces && progress) {
        progress = false;
        M.edges(E);
        std::vector<Item> items;
        items.reserve(E.size());
        for (auto &e : E) items.push_back({M.edge_len(e.first, e.second), e.first, e.second, stamp[e.first], stamp[e.second]});
        std::priority_queue<Item> heap(std::less<Item>(), std::move(items));
        while (M.faces_alive > target_faces && !heap.empty()) {
            const Item it = heap.top();
            heap.pop();
            if (!M.valive[it.u] || !M.valive[it.v] || stamp[it.u] != it.su || stamp[it.v] != it.sv) continue;
            if (!M.can_collapse(it.u, it.v, nu, nv)) continue;
            M.collapse(it.u, it.v, 0.5 * (M.P[it.u] + M.P[it.v]));
            progress = true;
            const int v = it.v;
            ++stamp[v];  // every older entry of an edge at v is stale now (its length changed)
            M.neighbours(v, nv);
            for (int w : nv) {
                const int a = std::min(v, w), b = std::max(v, w);
                heap.push({M.edge_len(a, b), a, b, stamp[a], stamp[b]});
            }
        }
    }
}

// ---- closest point on a triangle soup (uniform grid) ---------------------------------------------------------------------
static inline V3 closest_on_triangle(V3 p, V3 a, V3 b, V3 c) {
    // Ericson, Real-Time Collision Detection, 5.1.5
    const V3 ab = b - a, ac = c - a, ap = p - a;
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) return a;
    const V3 bp = p - b;
    const double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) return b;
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) return a + (d1 / (d1 - d3)) * ab;
    const V3 cp = p - c;
    const double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return c;
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) return a + (d2 / (d2 - d6)) * ac;
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return b + ((d4 - d3) / ((d4 - d3) + (d5 - d6))) * (c - b);
    const double den = 1.0 / (va + vb + vc);
    return a + (vb * den) * ab + (vc * den) * ac;
}

class SurfaceGrid {
  public:
    void build(const std::vector<V3> &P, const std::vector<std::array<int, 3>> &F) {
        P_ = P;
        F_ = F;
        lo_ = {1e300, 1e300, 1e300};
        V3 hi = {-1e300, -1e300, -1e300};
        for (auto &f : F_)
            for (int k = 0; k < 3; ++k) {
                const V3 p = P_[f[k]];
                lo_ = {std::min(lo_.x, p.x), std::min(lo_.y, p.y), std::min(lo_.z, p.z)};
                hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
            }
        if (F_.empty()) {
            n_[0] = n_[1] = n_[2] = 1;
            cell_ = 1;
            start_.assign(2, 0);
            return;
        }
        const V3 ext = hi - lo_;
        const double longest = std::max(ext.x, std::max(ext.y, ext.z));
        // cells about as wide as a triangle -- sqrt(#faces / 2) along the longest side of a surface -- but no more than
        // ~4 cells per face in total (the dense cell table is 4 bytes per cell)
        const double per_side = std::max(1.0, std::min(std::sqrt((double)F_.size() / 2.0), std::cbrt(4.0 * (double)F_.size())));
        cell_ = longest > 0 ? longest / per_side : 1.0;
        const double e[3] = {ext.x, ext.y, ext.z};
        for (int k = 0; k < 3; ++k) n_[k] = std::max(1, std::min(1024, (int)std::floor(e[k] / cell_) + 1));
        const size_t nc = (size_t)n_[0] * n_[1] * n_[2];
        std::vector<uint32_t> count(nc + 1, 0);
        auto range = [&](const std::array<int, 3> &f, int lo[3], int hi2[3]) {
            for (int k = 0; k < 3; ++k) {
                double mn = 1e300, mx = -1e300;
                for (int j = 0; j < 3; ++j) {
                    const V3 p = P_[f[j]];
                    const double c = k == 0 ? p.x : (k == 1 ? p.y : p.z);
                    mn = std::min(mn, c);
                    mx = std::max(mx, c);
                }
                const double o = k == 0 ? lo_.x : (k == 1 ? lo_.y : lo_.z);
                lo[k] = std::max(0, std::min(n_[k] - 1, (int)std::floor((mn - o) / cell_)));
                hi2[k] = std::max(0, std::min(n_[k] - 1, (int)std::floor((mx - o) / cell_)));
            }
        };
        for (auto &f : F_) {
            int a[3], b[3];
            range(f, a, b);
            for (int z = a[2]; z <= b[2]; ++z)
                for (int y = a[1]; y <= b[1]; ++y)
                    for (int x = a[0]; x <= b[0]; ++x) ++count[((size_t)z * n_[1] + y) * n_[0] + x + 1];
        }
        for (size_t i = 0; i < nc; ++i) count[i + 1] += count[i];
        start_ = count;
        items_.resize(start_[nc]);
        std::vector<uint32_t> fill(start_.begin(), start_.end() - 1);
        for (size_t fi = 0; fi < F_.size(); ++fi) {
            int a[3], b[3];
            range(F_[fi], a, b);
            for (int z = a[2]; z <= b[2]; ++z)
                for (int y = a[1]; y <= b[1]; ++y)
                    for (int x = a[0]; x <= b[0]; ++x) items_[fill[((size_t)z * n_[1] + y) * n_[0] + x]++] = (uint32_t)fi;
        }
    }
    // closest point of the surface to p
    V3 closest(V3 p) const {
        if (F_.empty()) return p;
        const double q[3] = {(p.x - lo_.x) / cell_, (p.y - lo_.y) / cell_, (p.z - lo_.z) / cell_};
        int c[3];
        for (int k = 0; k < 3; ++k) c[k] = std::max(0, std::min(n_[k] - 1, (int)std::floor(q[k])));
        double best = std::numeric_limits<double>::infinity();
        V3 bp = p;
        const int rmax = std::max(n_[0], std::max(n_[1], n_[2]));
        for (int r = 0; r <= rmax; ++r) {
            // shell of Chebyshev radius r around cell c
            for (int z = c[2] - r; z <= c[2] + r; ++z) {
                if (z < 0 || z >= n_[2]) continue;
                for (int y = c[1] - r; y <= c[1] + r; ++y) {
                    if (y < 0 || y >= n_[1]) continue;
                    const bool inner = std::abs(z - c[2]) != r && std::abs(y - c[1]) != r;
                    const int step = inner ? std::max(1, 2 * r) : 1;
                    for (int x = c[0] - r; x <= c[0] + r; x += step) {
                        if (x < 0 || x >= n_[0]) continue;
                        const size_t ci = ((size_t)z * n_[1] + y) * n_[0] + x;
                        for (uint32_t i = start_[ci]; i < start_[ci + 1]; ++i) {
                            const auto &f = F_[items_[i]];
                            const V3 cp = closest_on_triangle(p, P_[f[0]], P_[f[1]], P_[f[2]]);
                            const double d = dot(cp - p, cp - p);
                            if (d < best) {
                                best = d;
                                bp = cp;
                            }
                        }
                    }
                }
            }
            // everything not yet visited lies outside the block of cells [c - r, c + r]: at least as far away as the nearest
            // face of that block (p outside the grid on some axis: the block is clamped there and nothing lies beyond)
            double reach = std::numeric_limits<double>::infinity();
            for (int k = 0; k < 3; ++k) {
                if (c[k] - r > 0) reach = std::min(reach, q[k] - (c[k] - r));
                if (c[k] + r < n_[k] - 1) reach = std::min(reach, (c[k] + r + 1) - q[k]);
            }
            if (reach == std::numeric_limits<double>::infinity()) break;  // the block covers the whole grid
            reach = std::max(0.0, reach) * cell_;
            if (best <= reach * reach) break;
        }
        return bp;
    }

  private:
    std::vector<V3> P_;
    std::vector<std::array<int, 3>> F_;
    V3 lo_{0, 0, 0};
    double cell_ = 1;
    int n_[3] = {1, 1, 1};
    std::vector<uint32_t> start_, items_;
};

// ---- Botsch-Kobbelt isotropic remeshing ---------------------------------------------------------------------------------
struct RemeshStats {
    size_t splits = 0, collapses = 0, flips = 0;
};

static inline void split_long_edges(Mesh &M, double high, RemeshStats &st) {
    // repeated sweeps: halving an edge longer than 2 * high leaves halves that are still too long
    std::vector<std::pair<int, int>> E;
    for (int sweep = 0; sweep < 16; ++sweep) {
        M.edges(E);
        size_t n = 0;
        for (auto &e : E) {
            int ef[2];
            const int nef = M.edge_faces(e.first, e.second, ef);
            if (nef != 1 && nef != 2) continue;
            if (M.edge_len(e.first, e.second) > high) {
                M.split(e.first, e.second, 0.5 * (M.P[e.first] + M.P[e.second]));
                ++n;
            }
        }
        st.splits += n;
        if (!n) break;
    }
}

static inline void collapse_short_edges(Mesh &M, double low, double high, RemeshStats &st) {
    std::vector<std::pair<int, int>> E;
    M.edges(E);
    std::vector<int> nu, nv;
    for (auto &e : E) {
        int u = e.first, v = e.second;
        if (!M.valive[u] || !M.valive[v]) continue;
        int ef[2];
        const int nef = M.edge_faces(u, v, ef);
        if (nef == 0) continue;  // the edge went away with an earlier collapse
        if (M.edge_len(u, v) >= low) continue;
        const bool bu = M.is_boundary_vertex(u), bv = M.is_boundary_vertex(v);
        // boundary vertices are features: they are never moved or removed (an edge between two of them stays)
        if (bu && bv) continue;
        if (bu) std::swap(u, v);  // remove the interior endpoint
        if (!M.can_collapse(u, v, nu, nv)) continue;
        const bool keep_v = bu || bv;
        const V3 p = keep_v ? M.P[v] : 0.5 * (M.P[u] + M.P[v]);
        // no edge of the merged vertex may become longer than `high` (Botsch-Kobbelt: otherwise split and collapse undo
        // each other forever) ...
        bool ok = true;
        for (int w : nu)
            if (w != v && norm(M.P[w] - p) > high) ok = false;
        for (int w : nv)
            if (w != u && norm(M.P[w] - p) > high) ok = false;
        if (!ok) continue;
        // ... and no face around the merged vertex may turn over
        auto turns_over = [&](int x, int other) {
            for (int f : M.vf[x]) {
                if (Mesh::index_in(M.F[f], other) >= 0) continue;  // dies with the edge
                V3 q[3];
                for (int k = 0; k < 3; ++k) q[k] = M.F[f][k] == x ? p : M.P[M.F[f][k]];
                const V3 before = M.face_normal_raw(f), after = cross(q[1] - q[0], q[2] - q[0]);
                if (dot(before, after) <= 0) return true;
            }
            return false;
        };
        if (turns_over(u, v) || turns_over(v, u)) continue;
        M.collapse(u, v, p);
        ++st.collapses;
    }
}

static inline void equalize_valences(Mesh &M, RemeshStats &st) {
    std::vector<std::pair<int, int>> E;
    M.edges(E);
    auto valence = [&](int x) {
        // number of edges = faces for an interior vertex, faces + 1 on the boundary
        return M.vf[x].size() + (M.is_boundary_vertex(x) ? 1 : 0);
    };
    auto target = [&](int x) { return M.is_boundary_vertex(x) ? 4 : 6; };
    for (auto &e : E) {
        const int u = e.first, v = e.second;
        int ef[2];
        if (M.edge_faces(u, v, ef) != 2) continue;
        int f1 = ef[0], f2 = ef[1];
        if (!M.directed(f1, u, v)) std::swap(f1, f2);
        if (!M.directed(f1, u, v) || !M.directed(f2, v, u)) continue;
        const int a = M.third(f1, u, v), b = M.third(f2, u, v);
        if (a == b) continue;
        auto sq = [](int x) { return x * x; };
        const int before = sq(valence(u) - target(u)) + sq(valence(v) - target(v)) + sq(valence(a) - target(a)) + sq(valence(b) - target(b));
        const int after = sq(valence(u) - 1 - target(u)) + sq(valence(v) - 1 - target(v)) + sq(valence(a) + 1 - target(a)) +
                          sq(valence(b) + 1 - target(b));
        if (after >= before) continue;
        // geometric guard: the flipped pair must not fold over (both new normals on the side of the old ones) and the old
        // pair must not be a sharp crease (a flip across a crease cuts the corner off the surface)
        const V3 n1 = M.face_normal_raw(f1), n2 = M.face_normal_raw(f2);
        const double l1 = norm(n1), l2 = norm(n2);
        if (l1 == 0 || l2 == 0 || dot(n1, n2) < 0.5 * l1 * l2) continue;
        const V3 m1 = cross(M.P[b] - M.P[u], M.P[a] - M.P[u]), m2 = cross(M.P[v] - M.P[b], M.P[a] - M.P[b]);
        const V3 avg = (1.0 / l1) * n1 + (1.0 / l2) * n2;
        if (dot(m1, avg) <= 0 || dot(m2, avg) <= 0) continue;
        if (M.flip(u, v)) ++st.flips;
    }
}

// fn(begin, end) over [0, n) on up to 32 host threads (contiguous chunks; the result does not depend on the thread count as
// long as fn writes only its own indices)
template <class Fn>
static inline void parallel_ranges(size_t n, Fn fn) {
    size_t nt = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
    nt = std::min(nt, n / 4096 + 1);
    if (nt <= 1) {
        fn((size_t)0, n);
        return;
    }
    // A worker must not let an exception escape (std::terminate), and a thread that cannot be created (process thread limit)
    // must not lose its range: failures are collected and the ranges that did not run are done on the calling thread.
    const size_t chunk = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    std::vector<char> failed(nt, 0);
    size_t started = 0;
    for (size_t t = 0; t < nt; ++t) {
        const size_t a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) { started = t + 1; continue; }
        try {
            th.emplace_back([=, &failed]() {
                try { fn(a, b); } catch (...) { failed[t] = 1; }
            });
            started = t + 1;
        } catch (...) {  // std::system_error: no more threads
            break;
        }
    }
    for (auto &x : th) x.join();
    for (size_t t = 0; t < nt; ++t) {
        const size_t a = t * chunk, b = std::min(n, a + chunk);
        if (a < b && (t >= started || failed[t])) fn(a, b);  // serial fallback (rethrows on this thread if it fails again)
    }
}

static inline void tangential_relaxation(Mesh &M, const SurfaceGrid *surface) {
    const size_t nv = M.P.size();
    std::vector<V3> N(nv, V3{0, 0, 0});
    for (size_t f = 0; f < M.F.size(); ++f)
        if (M.falive[f]) {
            const V3 n = M.face_normal_raw((int)f);  // length = 2 * area: area-weighted vertex normals
            for (int k = 0; k < 3; ++k) N[M.F[f][k]] = N[M.F[f][k]] + n;
        }
    std::vector<V3> Q(M.P);
    // Jacobi step: every vertex reads the old positions and writes only its own new one
    parallel_ranges(nv, [&](size_t lo, size_t hi) {
    std::vector<int> nb;
    for (size_t u = lo; u < hi; ++u) {
        if (!M.valive[u] || M.vf[u].size() == 0 || M.is_boundary_vertex((int)u)) continue;
        M.neighbours((int)u, nb);
        V3 c{0, 0, 0};
        for (int w : nb) c = c + M.P[w];
        c = (1.0 / (double)nb.size()) * c;
        const double ln = norm(N[u]);
        if (ln == 0) continue;
        const V3 n = (1.0 / ln) * N[u];
        // p' = q + n n^T (p - q): the centroid, moved back into the vertex's tangent plane
        V3 p = c + dot(n, M.P[u] - c) * n;
        // a move must not turn a face of the 1-ring over (a sliver next to a crease can)
        bool ok = true;
        for (int f : M.vf[u]) {
            V3 q[3];
            for (int k = 0; k < 3; ++k) q[k] = M.F[f][k] == (int)u ? p : M.P[M.F[f][k]];
            if (dot(M.face_normal_raw(f), cross(q[1] - q[0], q[2] - q[0])) <= 0) ok = false;
        }
        if (!ok) continue;
        Q[u] = surface ? surface->closest(p) : p;
    }
    });
    // Second pass: the fold-over test above saw the OLD neighbours and the position BEFORE the projection; with every vertex
    // moving at once (Jacobi) and then being pulled onto the surface a face can still turn over.  Moves that flip a face of
    // their 1-ring against the old orientation are taken back, most displaced first would be finer -- taking back every
    // vertex of an inverted face is enough and order-independent.
    std::vector<char> undo(nv, 0);
    bool any = false;
    for (size_t f = 0; f < M.F.size(); ++f)
        if (M.falive[f]) {
            const V3 a = Q[M.F[f][0]], b = Q[M.F[f][1]], c = Q[M.F[f][2]];
            if (dot(M.face_normal_raw((int)f), cross(b - a, c - a)) <= 0)
                for (int k = 0; k < 3; ++k) { undo[M.F[f][k]] = 1; any = true; }
        }
    if (any)
        for (size_t u = 0; u < nv; ++u)
            if (undo[u]) Q[u] = M.P[u];
    M.P.swap(Q);
}

// `h` <= 0: the mean (half-)edge length of the input, like gpytoolbox's default.
static inline RemeshStats remesh_botsch(Mesh &M, int iters, double h, bool project) {
    RemeshStats st;
    if (h <= 0) h = M.mean_edge_length();
    if (!(h > 0) || M.faces_alive == 0) return st;
    SurfaceGrid grid;
    if (project) {
        std::vector<std::array<int, 3>> F0;
        F0.reserve(M.faces_alive);
        for (size_t f = 0; f < M.F.size(); ++f)
            if (M.falive[f]) F0.push_back(M.F[f]);
        grid.build(M.P, F0);
    }
    const double high = 4.0 / 3.0 * h, low = 4.0 / 5.0 * h;
    for (int it = 0; it < iters; ++it) {
        split_long_edges(M, high, st);
        collapse_short_edges(M, low, high, st);
        equalize_valences(M, st);
        tangential_relaxation(M, project ? &grid : nullptr);
    }
    return st;
}

}  // namespace remesh
}  // namespace sculpt
