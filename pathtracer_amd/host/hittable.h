// hittable.h -- base class of scene objects (hittable.h:41-90 in the reference) and hit_record.
// hit()/scatter() exist only on the device in the reference, so host code never calls them; the
// GPU side is libpathtracer_amd.  create_hittable_on_gpu() snapshots the object into the session.
#ifndef PT_HOST_HITTABLE_H
#define PT_HOST_HITTABLE_H
#include "aabb.h"
#include "pt_host.h"
#include "ray.h"

class material;
class hittable;

struct hit_record { // hittable.h:15-33; the material member is named `mat` (the reference's shadows its type)
    point3 p;
    vec3 normal;
    material *mat = nullptr;
    float t = 0.0f;
    bool front_face = false;
    void set_face_normal(const ray &r, const vec3 &outward_normal)
    {
        front_face = dot(r.direction(), outward_normal) < 0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

class hittable {
public:
    virtual ~hittable() {}
    virtual bool bounding_box(float time0, float time1, aabb &output_box) const = 0;
    virtual void create_hittable_on_gpu() = 0;

    material *material_ptr = nullptr;
    hittable **d_this = nullptr; // opaque, non-null once created; *d_this is this object
    int pt_slot = -1;            // the session's snapshot of this object

protected:
    void pt_register(uint8_t tag, const float p[6]); // defined in material.h (needs the material's index)
    hittable *self_ = nullptr;
};
#endif
