// cylinder.h -- cylinder2(center of the bottom, HEIGHT, RADIUS, material): note the argument order
// (cylinder.h:28 in the reference).  Open tube, no caps.
#ifndef PT_HOST_CYLINDER_H
#define PT_HOST_CYLINDER_H
#include <fstream>
#include <vector>

#include "cone.h"
#include "hittable.h"

class cylinder2 : public hittable {
public:
    point3 center;
    float height = 0.0f, radius = 0.0f;
    cylinder2() {}
    cylinder2(point3 center, float height, float radius, material *m) : center(center), height(height), radius(radius) { material_ptr = m; }
    void create_hittable_on_gpu() override
    {
        const float p[6] = {center.x(), center.y(), center.z(), height, radius, 0.0f};
        pt_register(PT_CYLINDER, p);
    }
    bool bounding_box(float, float, aabb &b) const override
    {
        b = aabb(center - vec3(radius, 0, radius), center + vec3(radius, height, radius));
        return true;
    }
    static cylinder2 *init_from_file(std::ifstream &is, std::vector<material *> &materials)
    {
        auto t = pt_host::read_round(is);
        return new cylinder2(t.center, t.height, t.radius, pt_host::pick_material(materials, t.material));
    }
};
#endif
