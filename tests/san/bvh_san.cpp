// Host-side tree code under AddressSanitizer / UBSan: builds, refits and widens trees over random scenes of every
// object kind (degenerate ones included) and checks the structural invariants the device traversal relies on.
// Compiled and run by tests/test_host_sanitizers.py together with pathtracer_amd/csrc/pt_bvh.cpp; no GPU involved.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../pathtracer_amd/csrc/pt_bvh.h"

namespace {

struct Scene {
    std::vector<uint8_t> type;
    std::vector<float> param;
    std::vector<uint32_t> mat;
    std::vector<uint8_t> mat_type;
    std::vector<float> mat_param;
    pt_scene_desc desc{};
    void finish()
    {
        mat_type = {PT_LAMBERTIAN};
        mat_param = {0.5f, 0.5f, 0.5f, 0.0f};
        desc.n_prims = (uint32_t)type.size();
        desc.prim_type = type.data();
        desc.prim_param = param.data();
        desc.prim_material = mat.data();
        desc.n_materials = 1;
        desc.mat_type = mat_type.data();
        desc.mat_param = mat_param.data();
        desc.camera = pt_camera{};
        desc.camera.origin[2] = 5.0f;
        desc.background[0] = desc.background[1] = desc.background[2] = 1.0f;
    }
};

void add(Scene &s, uint8_t type, std::mt19937 &rng, float spread, bool degenerate)
{
    std::uniform_real_distribution<float> pos(-spread, spread), size(0.01f, 0.1f * spread + 0.02f);
    float p[6] = {0, 0, 0, 0, 0, 0};
    switch (type) {
    case PT_SPHERE:
        p[0] = pos(rng); p[1] = pos(rng); p[2] = pos(rng);
        p[3] = degenerate ? 0.0f : size(rng);
        break;
    case PT_XY_RECT: case PT_XZ_RECT: case PT_YZ_RECT: {
        const float a = pos(rng), b = pos(rng);
        p[0] = a; p[1] = degenerate ? a : a + size(rng); // zero-area rect
        p[2] = b; p[3] = b + size(rng);
        p[4] = pos(rng);
        break;
    }
    case PT_CONE:
        p[0] = pos(rng); p[1] = pos(rng); p[2] = pos(rng);
        p[3] = size(rng); p[4] = degenerate ? 0.0f : size(rng);
        p[5] = (p[4] / p[3]) * (p[4] / p[3]);
        break;
    default:
        p[0] = pos(rng); p[1] = pos(rng); p[2] = pos(rng);
        p[3] = degenerate ? -size(rng) : size(rng); // a cylinder that extends downwards
        p[4] = size(rng);
        break;
    }
    s.type.push_back(type);
    s.param.insert(s.param.end(), p, p + 6);
    s.mat.push_back(0);
}

int fail(const char *what, uint32_t n, uint32_t leaf)
{
    fprintf(stderr, "bvh_san: %s (n = %u, leaf_max = %u)\n", what, n, leaf);
    return 1;
}

// every object once in the leaf order, every leaf within leaf_max, child boxes ordered lo <= hi, references in range
int check_binary(const pt::BvhBuild &b, uint32_t n, uint32_t leaf_max)
{
    if (b.order.size() != n) return fail("order size", n, leaf_max);
    std::vector<uint8_t> seen(n, 0), slot_seen(n, 0);
    for (uint32_t id : b.order) {
        if (id >= n || seen[id]) return fail("order is not a permutation", n, leaf_max);
        seen[id] = 1;
    }
    if (b.nodes.empty()) return fail("no nodes", n, leaf_max);
    std::vector<int32_t> todo = {0};
    size_t visited = 0;
    while (!todo.empty()) {
        const int32_t at = todo.back();
        todo.pop_back();
        if (at < 0 || (size_t)at >= b.nodes.size()) return fail("node reference out of range", n, leaf_max);
        if (++visited > b.nodes.size()) return fail("cycle in the tree", n, leaf_max);
        const pt::HostNode &nd = b.nodes[(size_t)at];
        for (int c = 0; c < 2; ++c) {
            const bool absent = !(nd.b[0][c] < INFINITY);
            if (absent) continue;
            for (int a = 0; a < 3; ++a)
                if (!(nd.b[a][c] <= nd.b[a][2 + c])) return fail("child box with lo > hi", n, leaf_max);
            if (nd.c[c] >= 0) {
                todo.push_back(nd.c[c]);
            } else {
                const uint32_t ref = (uint32_t)~nd.c[c], first = ref >> 3, count = (ref & 7u) + 1u;
                if (count > leaf_max || first + count > n) return fail("leaf out of range", n, leaf_max);
                for (uint32_t k = 0; k < count; ++k) {
                    if (slot_seen[first + k]) return fail("leaf slot referenced twice", n, leaf_max);
                    slot_seen[first + k] = 1;
                }
            }
        }
    }
    for (uint32_t k = 0; k < n; ++k)
        if (!slot_seen[k]) return fail("leaf slot never referenced", n, leaf_max);
    return 0;
}

// the wide tree references every leaf slot once as well, and its decoded planes are ordered
int check_wide(const pt::WideBuild &w, uint32_t n)
{
    std::vector<uint8_t> slot_seen(n, 0);
    std::vector<int32_t> todo = {0};
    size_t visited = 0;
    while (!todo.empty()) {
        const int32_t at = todo.back();
        todo.pop_back();
        if (at < 0 || (size_t)at >= w.nodes.size()) return fail("wide node reference out of range", n, 0);
        if (++visited > w.nodes.size()) return fail("cycle in the wide tree", n, 0);
        const pt::WideNode &nd = w.nodes[(size_t)at];
        if (nd.n_children > 4) return fail("wide node with more than four children", n, 0);
        for (int k = 0; k < 4; ++k) {
            if (nd.c[k] == pt::kWideAbsent) continue;
            for (int a = 0; a < 3; ++a)
                if (nd.qlo[a][k] > nd.qhi[a][k]) return fail("wide child with qlo > qhi", n, 0);
            if (nd.c[k] >= 0) {
                todo.push_back(nd.c[k]);
            } else {
                const uint32_t ref = (uint32_t)~nd.c[k], first = ref >> 3, count = (ref & 7u) + 1u;
                if (first + count > n) return fail("wide leaf out of range", n, 0);
                for (uint32_t j = 0; j < count; ++j) {
                    if (slot_seen[first + j]) return fail("wide leaf slot referenced twice", n, 0);
                    slot_seen[first + j] = 1;
                }
            }
        }
    }
    for (uint32_t k = 0; k < n; ++k)
        if (!slot_seen[k]) return fail("wide leaf slot never referenced", n, 0);
    return 0;
}

} // namespace

int main()
{
    std::mt19937 rng(20240607u);
    const uint32_t sizes[] = {1, 2, 3, 5, 8, 9, 64, 485, 3000};
    int cases = 0;
    for (uint32_t n : sizes) {
        for (uint32_t leaf_max : {1u, 2u, 4u, 8u}) {
            for (float spread : {1.0f, 300.0f}) {
                Scene s;
                for (uint32_t k = 0; k < n; ++k) add(s, (uint8_t)(rng() % PT_N_PRIM_TYPES), rng, spread, rng() % 16 == 0);
                if (n >= 5) { // coincident objects: splits that cannot separate them
                    for (int a = 0; a < 6; ++a) s.param[6 * 1 + a] = s.param[6 * 0 + a];
                    s.type[1] = s.type[0];
                }
                s.finish();
                pt::BvhBuild b;
                std::string err = pt::build_bvh(s.desc, leaf_max, b);
                if (!err.empty()) { fprintf(stderr, "bvh_san: build_bvh: %s\n", err.c_str()); return 1; }
                if (check_binary(b, n, leaf_max)) return 1;
                pt::WideBuild w;
                pt::make_wide(b, w);
                if (check_wide(w, n)) return 1;
                // the objects move and change size; same topology, new boxes
                std::uniform_real_distribution<float> jitter(-0.5f, 0.5f);
                for (uint32_t k = 0; k < n; ++k) {
                    const bool rect = s.type[k] >= PT_XY_RECT && s.type[k] <= PT_YZ_RECT;
                    if (rect) s.param[6 * k + 4] += jitter(rng);
                    else for (int a = 0; a < 3; ++a) s.param[6 * k + a] += jitter(rng);
                }
                err = pt::refit_bvh(s.desc, b);
                if (!err.empty()) { fprintf(stderr, "bvh_san: refit_bvh: %s\n", err.c_str()); return 1; }
                if (check_binary(b, n, leaf_max)) return 1;
                pt::make_wide(b, w);
                if (check_wide(w, n)) return 1;
                ++cases;
            }
        }
    }
    // descriptions the builder must refuse without touching memory it should not
    {
        Scene s;
        s.finish(); // no objects
        pt::BvhBuild b;
        const std::string err = pt::build_bvh(s.desc, 4, b);
        printf("empty scene: %s\n", err.empty() ? "accepted" : err.c_str());
        Scene t;
        add(t, PT_SPHERE, rng, 1.0f, false);
        t.type[0] = 17; // no such object kind
        t.finish();
        const std::string err2 = pt::build_bvh(t.desc, 4, b);
        if (err2.empty()) return fail("unknown object kind accepted", 1, 4);
        Scene u;
        add(u, PT_SPHERE, rng, 1.0f, false);
        u.param[0] = NAN;
        u.finish();
        const std::string err3 = pt::build_bvh(u.desc, 4, b);
        printf("NaN centre: %s\n", err3.empty() ? "accepted" : err3.c_str());
    }
    printf("ok %d cases\n", cases);
    return 0;
}
