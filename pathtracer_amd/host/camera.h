// camera.h -- thin-lens camera with the reference's constructor and public basis fields
// (camera.h:14-150 there).  The basis is computed here on the host, in the reference's operation
// order, and handed to the GPU as is.
#ifndef PT_HOST_CAMERA_H
#define PT_HOST_CAMERA_H
#include "commons.h"

class camera {
public:
    point3 origin, lower_left_corner;
    vec3 horizontal, vertical;
    vec3 u, v, w;
    float lens_radius = 0.0f;
    camera **d_this = nullptr;

    camera() {}
    camera(point3 lookfrom, point3 lookat, vec3 vup, const float vfov /* vertical, degrees */, const float aspect_ratio,
           const float aperture, const float focus_dist)
    {
        const float theta = vfov * PI / 180.0f;
        const float h = tanf(theta / 2.0f);
        const float viewport_height = 2.0f * h;
        const float viewport_width = aspect_ratio * viewport_height;
        w = unit_vector(lookfrom - lookat);
        u = unit_vector(cross(vup, w));
        v = cross(w, u);
        origin = lookfrom;
        horizontal = focus_dist * viewport_width * u;
        vertical = focus_dist * viewport_height * v;
        lower_left_corner = origin - horizontal / 2 - vertical / 2 - focus_dist * w;
        lens_radius = aperture / 2;
    }
    ~camera()
    {
        if (d_this) pt_host::Session::get().drop_camera(d_this);
    }
    void create_camera_on_gpu()
    {
        pt_camera c;
        auto put = [](float *dst, const vec3 &s) { dst[0] = s.x(); dst[1] = s.y(); dst[2] = s.z(); };
        put(c.origin, origin); put(c.lower_left_corner, lower_left_corner); put(c.horizontal, horizontal); put(c.vertical, vertical);
        put(c.u, u); put(c.v, v); put(c.w, w);
        c.lens_radius = lens_radius;
        self_ = this;
        d_this = &self_;
        pt_host::Session::get().set_camera(d_this, c);
    }

private:
    camera *self_ = nullptr;
};
#endif
