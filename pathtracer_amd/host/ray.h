// ray.h -- origin + direction (ray.h:10-60 in the reference).  Directions are never normalised.
#ifndef PT_HOST_RAY_H
#define PT_HOST_RAY_H
#include "vec3.h"

class ray {
public:
    point3 orig;
    vec3 dir;
    ray() {}
    ray(const point3 &origin, const vec3 &direction) : orig(origin), dir(direction) {}
    point3 origin() const { return orig; }
    vec3 direction() const { return dir; }
    point3 at(float t) const { return orig + t * dir; }
};
using Ray = ray;
#endif
