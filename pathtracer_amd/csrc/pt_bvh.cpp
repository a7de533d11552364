// pt_bvh.cpp -- binned-SAH BVH builder (host).  The reference declares aabb / bounding_box()
// (aabb.h:27-101 and every hittable) but never builds a hierarchy from them: its world is one
// linear hittable_list scan (hittable_list.h:92-98).  This builder is what turns those volumes
// into the flattened tree the gfx950 traversal kernel walks.
#include "pt_bvh.h"
#include <cstring>

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace pt {

namespace {

constexpr int kBins = 16;
constexpr uint32_t kMedianDepth = 40; // beyond this depth split at the object median (bounds the stack)
constexpr float kInf = std::numeric_limits<float>::infinity();

inline void box_empty(Box &b)
{
    for (int a = 0; a < 3; ++a) { b.lo[a] = kInf; b.hi[a] = -kInf; }
}
inline void box_grow(Box &b, const Box &o)
{
    for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); }
}
inline float box_half_area(const Box &b)
{
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
}

struct Builder {
    const std::vector<Box> &boxes;
    std::vector<float> cx[3];
    std::vector<uint32_t> &order;
    std::vector<HostNode> &nodes;
    uint32_t leaf_max;
    uint32_t max_depth = 0;

    Builder(const std::vector<Box> &b, std::vector<uint32_t> &o, std::vector<HostNode> &n, uint32_t lm)
        : boxes(b), order(o), nodes(n), leaf_max(lm)
    {
        for (int a = 0; a < 3; ++a) {
            cx[a].resize(b.size());
            for (size_t i = 0; i < b.size(); ++i) cx[a][i] = 0.5f * b[i].lo[a] + 0.5f * b[i].hi[a];
        }
    }

    Box range_box(uint32_t first, uint32_t count) const
    {
        Box r;
        box_empty(r);
        for (uint32_t i = first; i < first + count; ++i) box_grow(r, boxes[order[i]]);
        return r;
    }

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

    // Chooses where to cut [first, first+count); returns the size of the left part (0 < nl < count).
    uint32_t split(uint32_t first, uint32_t count, uint32_t depth)
    {
        float clo[3] = {kInf, kInf, kInf}, chi[3] = {-kInf, -kInf, -kInf};
        for (uint32_t i = first; i < first + count; ++i)
            for (int a = 0; a < 3; ++a) {
                float c = cx[a][order[i]];
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
        int best_axis = -1, best_bin = -1;
        float best_cost = kInf;
        if (depth < kMedianDepth) {
            for (int a = 0; a < 3; ++a) {
                float ext = chi[a] - clo[a];
                if (!(ext > 0) || !std::isfinite(ext)) continue;
                float scale = (float)kBins / ext;
                Box bb[kBins];
                uint32_t bn[kBins] = {0};
                for (int b = 0; b < kBins; ++b) box_empty(bb[b]);
                for (uint32_t i = first; i < first + count; ++i) {
                    uint32_t id = order[i];
                    int b = std::min(kBins - 1, std::max(0, (int)((cx[a][id] - clo[a]) * scale)));
                    bn[b]++;
                    box_grow(bb[b], boxes[id]);
                }
                float right_area[kBins];
                uint32_t right_n[kBins];
                Box acc;
                box_empty(acc);
                uint32_t n = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    box_grow(acc, bb[b]);
                    n += bn[b];
                    right_area[b] = box_half_area(acc);
                    right_n[b] = n;
                }
                box_empty(acc);
                n = 0;
                for (int b = 0; b < kBins - 1; ++b) { // cut between bin b and b+1
                    box_grow(acc, bb[b]);
                    n += bn[b];
                    if (n == 0 || right_n[b + 1] == 0) continue;
                    float cost = box_half_area(acc) * (float)n + right_area[b + 1] * (float)right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
        }
        if (best_axis >= 0) {
            float ext = chi[best_axis] - clo[best_axis];
            float scale = (float)kBins / ext, lo = clo[best_axis];
            const std::vector<float> &c = cx[best_axis];
            int bb = best_bin;
            auto mid = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t id) {
                int b = std::min(kBins - 1, std::max(0, (int)((c[id] - lo) * scale)));
                return b <= bb;
            });
            uint32_t nl = (uint32_t)(mid - (order.begin() + first));
            if (nl > 0 && nl < count) return nl;
        }
        // object median along the widest centroid axis (also the fallback for coincident centroids)
        int axis = 0;
        float w = -1.0f;
        for (int a = 0; a < 3; ++a) {
            float e = chi[a] - clo[a];
            if (e > w) { w = e; axis = a; }
        }
        uint32_t nl = count / 2;
        const std::vector<float> &c = cx[axis];
        std::nth_element(order.begin() + first, order.begin() + first + nl, order.begin() + first + count,
                         [&](uint32_t x, uint32_t y) { return c[x] < c[y] || (c[x] == c[y] && x < y); });
        return nl;
    }

    // Builds the subtree over [first, first+count) (count > leaf_max) and returns its node index.
    int32_t build_inner(uint32_t first, uint32_t count, uint32_t depth)
    {
        max_depth = std::max(max_depth, depth + 1);
        uint32_t nl = split(first, count, depth);
        uint32_t nr = count - nl;
        int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box bl = range_box(first, nl), br = range_box(first + nl, nr);
        int32_t c0 = nl > leaf_max ? build_inner(first, nl, depth + 1) : leaf_ref(first, nl);
        int32_t c1 = nr > leaf_max ? build_inner(first + nl, nr, depth + 1) : leaf_ref(first + nl, nr);
        HostNode &n = nodes[me];
        for (int a = 0; a < 3; ++a) {
            n.b[a][0] = bl.lo[a]; n.b[a][2] = bl.hi[a];
            n.b[a][1] = br.lo[a]; n.b[a][3] = br.hi[a];
        }
        n.c[0] = c0;
        n.c[1] = c1;
        n.pad[0] = n.pad[1] = 0;
        return me;
    }
};

} // namespace

bool prim_box(uint8_t type, const float *q, float kappa, float extra, Box &b)
{
    float extent = 0.0f; // quadrics: largest distance between the object's reference point and its surface
    const float pad_rect = 0.0001f; // aarect.h:48,147,232
    switch (type) {
    case PT_SPHERE: { // sphere.h:112-117 (|radius|: a negative radius would invert the reference's box)
        float r = std::fabs(q[3]);
        for (int a = 0; a < 3; ++a) { b.lo[a] = q[a] - r; b.hi[a] = q[a] + r; }
        extent = r;
        break;
    }
    case PT_XY_RECT: // aarect.h:45-50
        b.lo[0] = q[0]; b.hi[0] = q[1]; b.lo[1] = q[2]; b.hi[1] = q[3]; b.lo[2] = q[4] - pad_rect; b.hi[2] = q[4] + pad_rect;
        break;
    case PT_XZ_RECT: // aarect.h:144-149
        b.lo[0] = q[0]; b.hi[0] = q[1]; b.lo[2] = q[2]; b.hi[2] = q[3]; b.lo[1] = q[4] - pad_rect; b.hi[1] = q[4] + pad_rect;
        break;
    case PT_YZ_RECT: // aarect.h:229-234
        b.lo[1] = q[0]; b.hi[1] = q[1]; b.lo[2] = q[2]; b.hi[2] = q[3]; b.lo[0] = q[4] - pad_rect; b.hi[0] = q[4] + pad_rect;
        break;
    case PT_CONE:     // cone.h:142-147
    case PT_CYLINDER: // cylinder.h:133-138
    {
        float r = std::fabs(q[4]), h = q[3];
        b.lo[0] = q[0] - r; b.hi[0] = q[0] + r;
        b.lo[2] = q[2] - r; b.hi[2] = q[2] + r;
        b.lo[1] = q[1]; b.hi[1] = q[1] + h;
        extent = r + std::fabs(h);
        break;
    }
    default: return false;
    }
    float m = 0.0f;
    for (int a = 0; a < 3; ++a) {
        if (b.lo[a] > b.hi[a]) std::swap(b.lo[a], b.hi[a]);
        if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) return false;
        m = std::max(m, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    // The reference computes hit points in fp32 from coordinates of magnitude ~m, so an accepted
    // hit can sit a few ulp(m) outside the exact volume; 2^-17 * m is hundreds of ulps.
    float pad = m * (1.0f / 131072.0f) + 1.01f * kappa * extent / (1.0f - kappa) + extra + 1e-30f;
    for (int a = 0; a < 3; ++a) { b.lo[a] -= pad; b.hi[a] += pad; }
    return true;
}

// The ray cone of the scene and every object's padded box (what a build and a refit share).
static std::string cone_and_boxes(const pt_scene_desc &d, ConeParams &cp, std::vector<Box> &boxes)
{
    if (d.n_prims == 0) return "scene has no objects";
    if (d.n_prims >= (1u << 28)) return "too many objects (limit 2^28 - 1)";
    cp = ConeParams();
    boxes.resize(d.n_prims);
    // exact volumes first: the scene's extent decides which objects can carry their own padding
    Box root;
    box_empty(root);
    bool any_quadric = false, any_cone = false;
    float cone_slope = 1.0f;
    for (uint32_t i = 0; i < d.n_prims; ++i) {
        const float *q = d.prim_param + (size_t)i * 6;
        if (!prim_box(d.prim_type[i], q, 0.0f, 0.0f, boxes[i]))
            return "object " + std::to_string(i) + " has an unknown tag or non-finite parameters";
        box_grow(root, boxes[i]);
        if (d.prim_type[i] == PT_SPHERE || d.prim_type[i] == PT_CYLINDER) any_quadric = true;
        if (d.prim_type[i] == PT_CONE) {
            any_cone = true;
            cone_slope = std::max(cone_slope, std::sqrt(std::fabs(q[5]))); // tangens = (radius/height)^2, cone.h:31
        }
    }
    const float k_quad = 1.0f / 1024.0f, k_round = 1.0f / 1048576.0f;
    std::vector<float> extra(d.n_prims, 0.0f);
    if (any_cone) {
        cp.kappa_max = k_quad * cone_slope;
        if (!(cp.kappa_max <= 1.0f / 16.0f)) cp.kappa_max = 1.0f / 16.0f;
        cp.kappa_min = cp.kappa_max;
    } else if (!any_quadric) {
        cp.kappa_max = cp.kappa_min = k_round;
    } else {
        cp.kappa_max = k_quad;
        cp.kappa_min = k_round;
        cp.adaptive = 1;
        Box ext = root; // ray origins of a render: the camera and points on the objects
        for (int a = 0; a < 3; ++a) {
            ext.lo[a] = std::min(ext.lo[a], d.camera.origin[a]);
            ext.hi[a] = std::max(ext.hi[a], d.camera.origin[a]);
        }
        double diag2 = 0;
        for (int a = 0; a < 3; ++a) diag2 += ((double)ext.hi[a] - ext.lo[a]) * ((double)ext.hi[a] - ext.lo[a]);
        const double reach = 1.05 * std::sqrt(diag2) + 1e-3; // no origin in `ext` (grown below) is farther from any object
        Box dyn;
        box_empty(dyn);
        for (uint32_t i = 0; i < d.n_prims; ++i) {
            if (d.prim_type[i] != PT_SPHERE && d.prim_type[i] != PT_CYLINDER) continue;
            const float *q = d.prim_param + (size_t)i * 6;
            const double r = std::fabs(d.prim_type[i] == PT_SPHERE ? q[3] : q[4]);
            const double eta = (double)k_quad * reach;
            const double f = eta * eta / (std::sqrt(r * r + eta * eta) + r);
            if (f <= r / 256.0) {
                extra[i] = (float)(1.01 * f);
            } else {
                cp.r_min = std::min(cp.r_min, (float)r);
                box_grow(dyn, boxes[i]);
            }
        }
        for (int a = 0; a < 3; ++a) {
            const float grow = (float)(0.01 * reach) + std::fabs(d.camera.lens_radius);
            cp.elo[a] = ext.lo[a] - grow;
            cp.ehi[a] = ext.hi[a] + grow;
            if (dyn.lo[a] <= dyn.hi[a]) {
                cp.qc[a] = 0.5f * dyn.lo[a] + 0.5f * dyn.hi[a];
                cp.qh += (0.5f * (dyn.hi[a] - dyn.lo[a])) * 1.000001f + 1e-30f;
            }
        }
    }
    for (uint32_t i = 0; i < d.n_prims; ++i) // an object that carries its own padding needs no share of the cone's
        prim_box(d.prim_type[i], d.prim_param + (size_t)i * 6, extra[i] > 0.0f ? 0.0f : cp.kappa_max, extra[i], boxes[i]);
    return "";
}

std::string build_bvh(const pt_scene_desc &d, uint32_t leaf_max, BvhBuild &out)
{
    if (leaf_max < 1 || leaf_max > 8) return "leaf_max must be 1..8";
    std::vector<Box> boxes;
    const std::string e = cone_and_boxes(d, out.cone, boxes);
    if (!e.empty()) return e;
    out.order.resize(d.n_prims);
    std::iota(out.order.begin(), out.order.end(), 0u);
    out.nodes.clear();
    out.nodes.reserve(d.n_prims / 2 + 2);
    Builder b(boxes, out.order, out.nodes, leaf_max);
    if (d.n_prims > leaf_max) {
        b.build_inner(0, d.n_prims, 0);
    } else { // one leaf: a root whose second child is absent
        HostNode n{};
        Box bl = b.range_box(0, d.n_prims);
        for (int a = 0; a < 3; ++a) {
            n.b[a][0] = bl.lo[a]; n.b[a][2] = bl.hi[a];
            n.b[a][1] = kInf; n.b[a][3] = kInf; // a point at infinity: no ray reaches it
        }
        n.c[0] = Builder::leaf_ref(0, d.n_prims);
        n.c[1] = Builder::leaf_ref(0, 1);
        out.nodes.push_back(n);
        b.max_depth = 1;
    }
    out.max_depth = b.max_depth;
    return "";
}

std::string refit_bvh(const pt_scene_desc &d, BvhBuild &bvh)
{
    if (bvh.order.size() != d.n_prims || bvh.nodes.empty()) return "refit needs the tree of a scene with the same number of objects";
    std::vector<Box> boxes;
    const std::string e = cone_and_boxes(d, bvh.cone, boxes);
    if (!e.empty()) return e;
    // Children are created after their parent (build_inner), so a reverse sweep sees every child's box
    // before the node that holds it.
    std::vector<Box> own(bvh.nodes.size());
    for (size_t k = bvh.nodes.size(); k-- > 0;) {
        HostNode &n = bvh.nodes[k];
        Box me;
        box_empty(me);
        for (int child = 0; child < 2; ++child) {
            if (!(n.b[0][child] < kInf)) continue; // absent child (single-leaf scenes): stays a point at infinity
            Box cb;
            box_empty(cb);
            const int32_t ref = n.c[child];
            if (ref >= 0) {
                cb = own[(size_t)ref];
            } else {
                const uint32_t r = (uint32_t)~ref, first = r >> 3, count = (r & 7u) + 1u;
                for (uint32_t s = first; s < first + count; ++s) box_grow(cb, boxes[bvh.order[s]]);
            }
            for (int a = 0; a < 3; ++a) { n.b[a][child] = cb.lo[a]; n.b[a][2 + child] = cb.hi[a]; }
            box_grow(me, cb);
        }
        own[k] = me;
    }
    return "";
}


namespace {

struct WideChild {
    Box box;
    int32_t ref;
};

// step 2^(e-127) and origin for one axis of a wide node whose children span [lo, hi]
void quantisation_grid(float lo, float hi, uint8_t &e_out, float &origin)
{
    const double ext = (double)hi - (double)lo;
    int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
    e = std::max(e, -126);
    for (;; ++e) {
        const double s = std::ldexp(1.0, e);
        const double k = std::floor((double)lo / s);
        if (std::fabs(k) > 8388608.0 || std::fabs((double)hi / s) > 8388608.0) continue; // origin + q * s must stay exact in fp32
        if (((double)hi - k * s) / s > 255.0) continue;
        e_out = (uint8_t)(e + 127);
        origin = (float)(k * s);
        return;
    }
}

struct WideMaker {
    const BvhBuild &bvh;
    WideBuild &out;

    static double area(const Box &b)
    {
        const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    }
    void children_of(int32_t node, std::vector<WideChild> &kids) const
    {
        const HostNode &n = bvh.nodes[(size_t)node];
        for (int c = 0; c < 2; ++c) {
            if (!(n.b[0][c] < kInf)) continue; // absent
            WideChild k;
            for (int a = 0; a < 3; ++a) { k.box.lo[a] = n.b[a][c]; k.box.hi[a] = n.b[a][2 + c]; }
            k.ref = n.c[c];
            kids.push_back(k);
        }
    }
    // emits the wide node that stands for binary node `node`; returns its index
    int32_t emit(int32_t node, uint32_t depth)
    {
        out.max_depth = std::max(out.max_depth, depth);
        std::vector<WideChild> kids;
        children_of(node, kids);
        while (kids.size() < 4) { // pull up the inner child with the largest surface
            int pick = -1;
            double best = -1.0;
            for (size_t k = 0; k < kids.size(); ++k)
                if (kids[k].ref >= 0 && area(kids[k].box) > best) { best = area(kids[k].box); pick = (int)k; }
            if (pick < 0) break;
            const int32_t inner = kids[(size_t)pick].ref;
            kids.erase(kids.begin() + pick);
            children_of(inner, kids);
        }
        const int32_t me = (int32_t)out.nodes.size();
        out.nodes.emplace_back();
        WideNode w{};
        Box u;
        box_empty(u);
        for (const WideChild &k : kids) box_grow(u, k.box);
        float step[3];
        for (int a = 0; a < 3; ++a) {
            quantisation_grid(u.lo[a], u.hi[a], w.exp[a], w.origin[a]);
            step[a] = std::ldexp(1.0f, (int)w.exp[a] - 127);
        }
        w.n_children = (uint8_t)kids.size();
        for (int k = 0; k < 4; ++k) {
            w.c[k] = kWideAbsent;
            for (int a = 0; a < 3; ++a) { w.qlo[a][k] = 255; w.qhi[a][k] = 0; }
        }
        for (size_t k = 0; k < kids.size(); ++k) {
            for (int a = 0; a < 3; ++a) {
                int ql = (int)std::floor(((double)kids[k].box.lo[a] - w.origin[a]) / step[a]);
                int qh = (int)std::ceil(((double)kids[k].box.hi[a] - w.origin[a]) / step[a]);
                ql = std::min(std::max(ql, 0), 255);
                qh = std::min(std::max(qh, 0), 255);
                // the decoded planes, as the device computes them, must enclose the child's box
                while (ql > 0 && w.origin[a] + (float)ql * step[a] > kids[k].box.lo[a]) --ql;
                while (qh < 255 && w.origin[a] + (float)qh * step[a] < kids[k].box.hi[a]) ++qh;
                w.qlo[a][k] = (uint8_t)ql;
                w.qhi[a][k] = (uint8_t)qh;
            }
        }
        for (size_t k = 0; k < kids.size(); ++k) w.c[k] = kids[k].ref >= 0 ? emit(kids[k].ref, depth + 1) : kids[k].ref;
        out.nodes[(size_t)me] = w;
        return me;
    }
};

} // namespace

namespace {
struct Wide32Maker {
    const BvhBuild &bvh;
    std::vector<float> &out;
    uint32_t max_depth = 0;
    int32_t emit(int32_t node, uint32_t depth)
    {
        max_depth = std::max(max_depth, depth);
        auto children_of = [&](int32_t nd, std::vector<WideChild> &list) {
            const HostNode &n = bvh.nodes[(size_t)nd];
            for (int c = 0; c < 2; ++c) {
                if (!(n.b[0][c] < kInf)) continue; // absent
                WideChild k;
                for (int a = 0; a < 3; ++a) { k.box.lo[a] = n.b[a][c]; k.box.hi[a] = n.b[a][2 + c]; }
                k.ref = n.c[c];
                list.push_back(k);
            }
        };
        std::vector<WideChild> kids;
        children_of(node, kids);
        while (kids.size() < 4) { // pull up the inner child with the largest surface
            int pick = -1;
            double best = -1.0;
            for (size_t k = 0; k < kids.size(); ++k)
                if (kids[k].ref >= 0 && WideMaker::area(kids[k].box) > best) { best = WideMaker::area(kids[k].box); pick = (int)k; }
            if (pick < 0) break;
            const int32_t inner = kids[(size_t)pick].ref;
            kids.erase(kids.begin() + pick);
            children_of(inner, kids);
        }
        const size_t me = out.size() / 28;
        out.resize(out.size() + 28, kInf);
        int32_t refs[4] = {kWideAbsent, kWideAbsent, kWideAbsent, kWideAbsent};
        for (size_t k = 0; k < kids.size(); ++k) {
            for (int a = 0; a < 3; ++a) {
                out[me * 28 + (size_t)a * 8 + k] = kids[k].box.lo[a];
                out[me * 28 + (size_t)a * 8 + 4 + k] = kids[k].box.hi[a];
            }
        }
        for (size_t k = 0; k < kids.size(); ++k) refs[k] = kids[k].ref >= 0 ? emit(kids[k].ref, depth + 1) : kids[k].ref;
        memcpy(&out[me * 28 + 24], refs, 16);
        return (int32_t)me;
    }
};
} // namespace

uint32_t make_wide_fp32(const BvhBuild &bvh, std::vector<float> &out)
{
    out.clear();
    Wide32Maker m{bvh, out};
    m.emit(0, 1);
    return m.max_depth;
}

void make_wide(const BvhBuild &bvh, WideBuild &out)
{
    out.nodes.clear();
    out.nodes.reserve(bvh.nodes.size() / 2 + 2);
    out.max_depth = 0;
    WideMaker m{bvh, out};
    m.emit(0, 1);
}

} // namespace pt
