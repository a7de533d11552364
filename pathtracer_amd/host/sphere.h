// sphere.h -- sphere(center, radius, material) (sphere.h:11-83 in the reference).
#ifndef PT_HOST_SPHERE_H
#define PT_HOST_SPHERE_H
#include <fstream>
#include <vector>

#include "hittable.h"
#include "material.h"

namespace pt_host {
inline material *pick_material(std::vector<material *> &materials, int index)
{
    // the reference indexes the vector unchecked (sphere.h:71)
    if (index < 0 || index >= (int)materials.size()) throw scene_error("object names material " + std::to_string(index) + ", which the Materials section does not define");
    return materials[index];
}
} // namespace pt_host

class sphere : public hittable {
public:
    point3 center;
    float radius = 0.0f;
    sphere() {}
    sphere(point3 cen, float r, material *m) : center(cen), radius(r) { material_ptr = m; }
    void create_hittable_on_gpu() override
    {
        const float p[6] = {center.x(), center.y(), center.z(), radius, 0.0f, 0.0f};
        pt_register(PT_SPHERE, p);
    }
    bool bounding_box(float, float, aabb &box) const override
    {
        box = aabb(center - vec3(radius, radius, radius), center + vec3(radius, radius, radius));
        return true;
    }
    static sphere *init_from_file(std::ifstream &is, std::vector<material *> &materials)
    {
        point3 cen;
        float r = 1.0f;
        int mi = 0;
        pt_host::read_block(is, [&](const std::string &k, const std::string &v) {
            if (k == "center") cen.load_from_string(v);
            else if (k == "radius") r = std::stof(v);
            else if (k == "material") mi = std::stoi(v);
        });
        return new sphere(cen, r, pt_host::pick_material(materials, mi));
    }
};
#endif
