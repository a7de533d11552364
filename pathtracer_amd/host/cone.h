// cone.h -- cone(center of the base, height, base radius, material) (cone.h:11-99 in the reference).
// No base cap; tangens = (radius/height)^2 is fixed at construction (cone.h:30).
#ifndef PT_HOST_CONE_H
#define PT_HOST_CONE_H
#include <fstream>
#include <vector>

#include "hittable.h"
#include "sphere.h"

namespace pt_host {
struct RoundText { point3 center; float radius = 1.0f, height = 10.0f; int material = 0; };
inline RoundText read_round(std::ifstream &is)
{
    RoundText t;
    read_block(is, [&](const std::string &k, const std::string &v) {
        if (k == "center") {
            point3 p;
            p.load_from_string(v); // parsed in any case, so malformed text is still an error
            if (options().honor_cone_cylinder_center) t.center = p; // else dropped, as in the reference
        } else if (k == "radius") t.radius = std::stof(v);
        else if (k == "height") t.height = std::stof(v);
        else if (k == "material") t.material = std::stoi(v);
    });
    return t;
}
} // namespace pt_host

class cone : public hittable {
public:
    point3 center;
    float height = 0.0f, radius = 0.0f, tangens = 0.0f;
    cone() {}
    cone(point3 center, float height, float radius, material *m) : center(center), height(height), radius(radius)
    {
        material_ptr = m;
        tangens = (radius / height) * (radius / height);
    }
    void create_hittable_on_gpu() override
    {
        // the device twin is rebuilt from (center, height, radius) and recomputes tangens (cone.h:161-165)
        const float p[6] = {center.x(), center.y(), center.z(), height, radius, (radius / height) * (radius / height)};
        pt_register(PT_CONE, p);
    }
    bool bounding_box(float, float, aabb &b) const override
    {
        b = aabb(center - vec3(radius, 0, radius), center + vec3(radius, height, radius));
        return true;
    }
    static cone *init_from_file(std::ifstream &is, std::vector<material *> &materials)
    {
        auto t = pt_host::read_round(is);
        return new cone(t.center, t.height, t.radius, pt_host::pick_material(materials, t.material));
    }
};
#endif
