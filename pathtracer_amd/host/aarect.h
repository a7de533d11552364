// aarect.h -- xy_rect / xz_rect / yz_rect (aarect.h:14-292 in the reference): an axis-aligned
// rectangle in the plane <axis> = k, bounded on the two other axes.
#ifndef PT_HOST_AARECT_H
#define PT_HOST_AARECT_H
#include <fstream>
#include <vector>

#include "hittable.h"
#include "sphere.h"

namespace pt_host {
struct RectText { float a0 = 0, a1 = 1, b0 = 0, b1 = 1, k = 0.2f; int material = 0; };
inline RectText read_rect(std::ifstream &is, const char *an, const char *bn)
{
    RectText t;
    const std::string a0 = std::string(an) + "0", a1 = std::string(an) + "1", b0 = std::string(bn) + "0", b1 = std::string(bn) + "1";
    read_block(is, [&](const std::string &key, const std::string &v) {
        if (key == a0) t.a0 = std::stof(v);
        else if (key == a1) t.a1 = std::stof(v);
        else if (key == b0) t.b0 = std::stof(v);
        else if (key == b1) t.b1 = std::stof(v);
        else if (key == "k") t.k = std::stof(v);
        else if (key == "material") t.material = std::stoi(v);
    });
    return t;
}
} // namespace pt_host

class xy_rect : public hittable {
public:
    float x0 = 0, x1 = 0, y0 = 0, y1 = 0, k = 0;
    xy_rect() {}
    xy_rect(float _x0, float _x1, float _y0, float _y1, float _k, material *mat) : x0(_x0), x1(_x1), y0(_y0), y1(_y1), k(_k) { material_ptr = mat; }
    void create_hittable_on_gpu() override { const float p[6] = {x0, x1, y0, y1, k, 0.0f}; pt_register(PT_XY_RECT, p); }
    bool bounding_box(float, float, aabb &b) const override { b = aabb(point3(x0, y0, k - 0.0001f), point3(x1, y1, k + 0.0001f)); return true; }
    static xy_rect *load_from_file(std::ifstream &f, std::vector<material *> &materials)
    {
        auto t = pt_host::read_rect(f, "x", "y");
        return new xy_rect(t.a0, t.a1, t.b0, t.b1, t.k, pt_host::pick_material(materials, t.material));
    }
};

class xz_rect : public hittable {
public:
    float x0 = 0, x1 = 0, z0 = 0, z1 = 0, k = 0;
    xz_rect() {}
    xz_rect(float _x0, float _x1, float _z0, float _z1, float _k, material *mat) : x0(_x0), x1(_x1), z0(_z0), z1(_z1), k(_k) { material_ptr = mat; }
    void create_hittable_on_gpu() override { const float p[6] = {x0, x1, z0, z1, k, 0.0f}; pt_register(PT_XZ_RECT, p); }
    bool bounding_box(float, float, aabb &b) const override { b = aabb(point3(x0, k - 0.0001f, z0), point3(x1, k + 0.0001f, z1)); return true; }
    static xz_rect *load_from_file(std::ifstream &f, std::vector<material *> &materials)
    {
        auto t = pt_host::read_rect(f, "x", "z");
        return new xz_rect(t.a0, t.a1, t.b0, t.b1, t.k, pt_host::pick_material(materials, t.material));
    }
};

class yz_rect : public hittable {
public:
    float y0 = 0, y1 = 0, z0 = 0, z1 = 0, k = 0;
    yz_rect() {}
    yz_rect(float _y0, float _y1, float _z0, float _z1, float _k, material *mat) : y0(_y0), y1(_y1), z0(_z0), z1(_z1), k(_k) { material_ptr = mat; }
    void create_hittable_on_gpu() override { const float p[6] = {y0, y1, z0, z1, k, 0.0f}; pt_register(PT_YZ_RECT, p); }
    bool bounding_box(float, float, aabb &b) const override { b = aabb(point3(k - 0.0001f, y0, z0), point3(k + 0.0001f, y1, z1)); return true; }
    static yz_rect *load_from_file(std::ifstream &f, std::vector<material *> &materials)
    {
        auto t = pt_host::read_rect(f, "y", "z");
        return new yz_rect(t.a0, t.a1, t.b0, t.b1, t.k, pt_host::pick_material(materials, t.material));
    }
};
#endif
