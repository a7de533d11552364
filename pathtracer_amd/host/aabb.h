// aabb.h -- axis-aligned box, usable on the host (the reference's is device-only, aabb.h:27-101,
// and is never used by it; here the boxes feed the BVH builder inside libpathtracer_amd).
#ifndef PT_HOST_AABB_H
#define PT_HOST_AABB_H
#include <utility>

#include "ray.h"

inline float ffmin(float a, float b) { return a < b ? a : b; }
inline float ffmax(float a, float b) { return a > b ? a : b; }

class aabb {
public:
    point3 minimum, maximum;
    aabb() {}
    aabb(const point3 &a, const point3 &b) : minimum(a), maximum(b) {}
    point3 min() const { return minimum; }
    point3 max() const { return maximum; }
    // slab test, same comparisons as aabb.h:78-91
    bool hit(const ray &r, float t_min, float t_max) const
    {
        for (int a = 0; a < 3; ++a) {
            float inv = 1.0f / r.direction()[a];
            float t0 = (minimum[a] - r.origin()[a]) * inv, t1 = (maximum[a] - r.origin()[a]) * inv;
            if (inv < 0.0f) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        return true;
    }
};

inline aabb surrounding_box(aabb a, aabb b)
{
    return aabb(point3(ffmin(a.min().x(), b.min().x()), ffmin(a.min().y(), b.min().y()), ffmin(a.min().z(), b.min().z())),
                point3(ffmax(a.max().x(), b.max().x()), ffmax(a.max().y(), b.max().y()), ffmax(a.max().z(), b.max().z())));
}
#endif
