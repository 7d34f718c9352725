// AddressSanitizer / UBSan driver for the host-side mesh operations (sculptmate_amd/csrc/remesh_host.h, the code behind the
// host-pointer entry points sculpt_mesh_subdivide / _decimate / _remesh_botsch).  TEST INFRASTRUCTURE: built and run by
// tests/test_sanitizers.py with g++ -fsanitize=address,undefined (GPU ASan is not available on the pool; this code has no GPU part).
#include <cstdio>
#include <cstdlib>

#include "../../sculptmate_amd/csrc/remesh_host.h"

using namespace sculpt::remesh;

static void icosahedron(std::vector<double> &V, std::vector<int32_t> &F) {
    const double t = (1 + std::sqrt(5.0)) / 2;
    const double v[12][3] = {{-1, t, 0}, {1, t, 0}, {-1, -t, 0}, {1, -t, 0}, {0, -1, t}, {0, 1, t}, {0, -1, -t}, {0, 1, -t}, {t, 0, -1}, {t, 0, 1}, {-t, 0, -1}, {-t, 0, 1}};
    const int f[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                          {3, 9, 4}, {3, 4, 2},  {3, 2, 6}, {3, 6, 8},  {3, 8, 9},   {4, 9, 5}, {2, 4, 11}, {6, 2, 10},  {8, 6, 7},  {9, 8, 1}};
    V.assign(&v[0][0], &v[0][0] + 36);
    F.assign(&f[0][0], &f[0][0] + 60);
}

static void check_closed(const std::vector<double> &V, const std::vector<int32_t> &F, const char *what) {
    // every directed edge exactly once and its opposite present
    std::unordered_map<uint64_t, int> dir;
    const size_t nf = F.size() / 3;
    for (size_t f = 0; f < nf; ++f)
        for (int k = 0; k < 3; ++k) {
            const uint64_t a = (uint32_t)F[3 * f + k], b = (uint32_t)F[3 * f + (k + 1) % 3];
            if (a >= V.size() / 3 || b >= V.size() / 3 || a == b) { std::printf("%s: bad face\n", what); std::exit(1); }
            if (++dir[(a << 32) | b] != 1) { std::printf("%s: directed edge twice\n", what); std::exit(1); }
        }
    for (auto &kv : dir)
        if (!dir.count((kv.first << 32) | (kv.first >> 32))) { std::printf("%s: open edge\n", what); std::exit(1); }
    if ((long)(V.size() / 3) - (long)(dir.size() / 2) + (long)nf != 2) { std::printf("%s: euler characteristic\n", what); std::exit(1); }
}

int main() {
    std::vector<double> V;
    std::vector<int32_t> F;
    icosahedron(V, F);
    for (int i = 0; i < 4; ++i) subdivide_once(V, F);  // 5120 faces
    for (size_t i = 0; i < V.size() / 3; ++i) {          // a bumpy sphere with uneven edges
        const double n = std::sqrt(V[3 * i] * V[3 * i] + V[3 * i + 1] * V[3 * i + 1] + V[3 * i + 2] * V[3 * i + 2]);
        const double s = (1 + 0.2 * std::sin(5 * V[3 * i]) * std::cos(3 * V[3 * i + 1])) / n;
        for (int k = 0; k < 3; ++k) V[3 * i + k] *= s;
    }
    check_closed(V, F, "input");
    {
        Mesh M;
        if (!M.build(V.data(), V.size() / 3, F.data(), F.size() / 3).empty()) return 1;
        decimate(M, 600);
        std::vector<double> Vo;
        std::vector<int32_t> Fo;
        M.compact(Vo, Fo);
        check_closed(Vo, Fo, "decimate");
        if (Fo.size() / 3 > 600) { std::printf("decimate: %zu faces\n", Fo.size() / 3); return 1; }
        Mesh R;
        if (!R.build(Vo.data(), Vo.size() / 3, Fo.data(), Fo.size() / 3).empty()) return 1;
        const RemeshStats st = remesh_botsch(R, 6, -1.0, true);
        std::vector<double> Vr;
        std::vector<int32_t> Fr;
        R.compact(Vr, Fr);
        check_closed(Vr, Fr, "remesh");
        std::printf("decimate %zu -> %zu faces; remesh -> %zu faces (%zu splits, %zu collapses, %zu flips)\n", F.size() / 3, Fo.size() / 3,
                    Fr.size() / 3, st.splits, st.collapses, st.flips);
        // a much finer target: many splits per edge (vector growth while iterating)
        Mesh S;
        if (!S.build(Vo.data(), Vo.size() / 3, Fo.data(), Fo.size() / 3).empty()) return 1;
        remesh_botsch(S, 3, 0.25 * S.mean_edge_length(), true);
        S.compact(Vr, Fr);
        check_closed(Vr, Fr, "remesh fine");
        // and a much coarser one: collapses down to a few faces
        Mesh T;
        if (!T.build(Vo.data(), Vo.size() / 3, Fo.data(), Fo.size() / 3).empty()) return 1;
        remesh_botsch(T, 4, 6.0 * T.mean_edge_length(), false);
        T.compact(Vr, Fr);
        check_closed(Vr, Fr, "remesh coarse");
    }
    {  // decimate to nothing: ends at a tetrahedron
        Mesh M;
        if (!M.build(V.data(), V.size() / 3, F.data(), F.size() / 3).empty()) return 1;
        decimate(M, 0);
        if (M.faces_alive != 4) { std::printf("decimate(0): %zu faces\n", M.faces_alive); return 1; }
    }
    {  // open strip (boundary), bad input, empty input
        std::vector<double> P;
        std::vector<int32_t> T;
        const int n = 12;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < 3; ++j) { P.push_back(i * 0.37); P.push_back(j * 0.5); P.push_back(0.05 * i * j); }
        for (int i = 0; i + 1 < n; ++i)
            for (int j = 0; j < 2; ++j) {
                const int a = i * 3 + j, b = (i + 1) * 3 + j, c = (i + 1) * 3 + j + 1, d = i * 3 + j + 1;
                const int t[6] = {a, b, c, a, c, d};
                T.insert(T.end(), t, t + 6);
            }
        Mesh M;
        if (!M.build(P.data(), P.size() / 3, T.data(), T.size() / 3).empty()) return 1;
        decimate(M, 10);
        remesh_botsch(M, 5, -1.0, true);
        std::vector<double> Vo;
        std::vector<int32_t> Fo;
        M.compact(Vo, Fo);
        if (Fo.empty()) return 1;
        T[4] = 1000;
        Mesh B;
        if (B.build(P.data(), P.size() / 3, T.data(), T.size() / 3).empty()) { std::printf("bad index accepted\n"); return 1; }
        Mesh E;
        if (!E.build(nullptr, 0, nullptr, 0).empty()) return 1;
        decimate(E, 0);
        remesh_botsch(E, 3, -1.0, true);
        E.compact(Vo, Fo);
        if (!Vo.empty() || !Fo.empty()) return 1;
    }
    std::printf("asan_remesh ok\n");
    return 0;
}
