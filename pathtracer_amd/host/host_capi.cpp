// host_capi.cpp -- C entry points over the scene API headers, for tests: build a scene the way a
// reference user would (constructors, create_*_on_gpu, scene.txt) and hand back what would be uploaded.
#include <cstring>
#include <fstream>
#include <string>

#include "render.h"
#include "scene.h"

namespace {
std::string g_err;
scene *g_scene = nullptr;
pt_host::Flat g_flat;
pt_camera g_cam;

int finish()
{
    g_flat = pt_host::Session::get().flatten(g_scene->world->d_this);
    g_cam = pt_host::Session::get().camera(g_scene->cam->d_this);
    return 0;
}
} // namespace

extern "C" {

const char *pth_error() { return g_err.c_str(); }

void pth_close()
{
    delete g_scene;
    g_scene = nullptr;
    pt_host::Session::get().reset();
}

// which: 0 = scene(), 1 = scene(int); returns 0 or -1 (pth_error() has the text)
int pth_open_builtin(int which)
{
    pth_close();
    try {
        g_scene = which ? new scene(which) : new scene();
        return finish();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// a scene.txt: the Image header is read like main does; width/height/spp come back through the pointers
int pth_open_file(const char *path, int honor_center, int *width, int *height, int *spp)
{
    pth_close();
    pt_host::options().honor_cone_cylinder_center = honor_center != 0;
    try {
        std::ifstream file(path);
        if (!file.is_open()) throw pt_host::scene_error(std::string("cannot open ") + path);
        std::string line;
        int w = 1200, h = 902, s = 80;
        do {
            if (!std::getline(file, line)) throw pt_host::scene_error("scene file has no Scene section");
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line == "Image") continue;
            auto kv = get_key_value(line);
            if (kv[0] == "width") w = std::stoi(kv[1]);
            else if (kv[0] == "height") h = std::stoi(kv[1]);
            else if (kv[0] == "samples_per_pixel") s = std::stoi(kv[1]);
        } while (line != "Scene" && line != "END");
        if (line != "Scene") throw pt_host::scene_error("Wrong file format");
        g_scene = new scene(file, (float)w / (float)h);
        if (width) *width = w;
        if (height) *height = h;
        if (spp) *spp = s;
        return finish();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

int pth_counts(int *n_prims, int *n_materials)
{
    if (!g_scene) return -1;
    *n_prims = (int)g_flat.prim_type.size();
    *n_materials = (int)g_flat.mat_type.size();
    return 0;
}

int pth_copy(uint8_t *prim_type, float *prim_param, uint32_t *prim_material, uint8_t *mat_type, float *mat_param, pt_camera *cam, float *background)
{
    if (!g_scene) return -1;
    memcpy(prim_type, g_flat.prim_type.data(), g_flat.prim_type.size());
    memcpy(prim_param, g_flat.prim_param.data(), g_flat.prim_param.size() * 4);
    memcpy(prim_material, g_flat.prim_material.data(), g_flat.prim_material.size() * 4);
    memcpy(mat_type, g_flat.mat_type.data(), g_flat.mat_type.size());
    memcpy(mat_param, g_flat.mat_param.data(), g_flat.mat_param.size() * 4);
    *cam = g_cam;
    background[0] = g_scene->background_color.x(); background[1] = g_scene->background_color.y(); background[2] = g_scene->background_color.z();
    return 0;
}

// renders the open scene through pt_host::render (needs a GPU); fb is width*height*3 floats, gamma-corrected, bottom-up
int pth_render(int width, int height, int spp, int max_depth, float *fb, unsigned long long *rays)
{
    if (!g_scene) return -1;
    try {
        pt_host::options().max_depth = max_depth;
        std::vector<vec3> buf((size_t)width * height);
        pt_stats st;
        pt_host::render(buf.data(), width, height, spp, g_scene->background_color, g_scene->cam->d_this, g_scene->world->d_this, &st);
        for (size_t p = 0; p < buf.size(); ++p) memcpy(fb + 3 * p, buf[p].e, 12);
        if (rays) *rays = st.rays;
        return 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// Two scenes alive at once: the second one's world must flatten to what it flattens to alone (its own materials, numbered
// from 0).  Returns 1 if so, 0 if not, -1 on an error.
int pth_two_scenes_alive()
{
    pth_close();
    try {
        scene *alone = new scene(1);
        const pt_host::Flat ref = pt_host::Session::get().flatten(alone->world->d_this);
        delete alone;
        pt_host::Session::get().reset();
        scene *first = new scene();
        scene *second = new scene(1);
        const pt_host::Flat got = pt_host::Session::get().flatten(second->world->d_this);
        const pt_host::Flat one = pt_host::Session::get().flatten(first->world->d_this);
        const bool same = got.mat_type == ref.mat_type && got.mat_param == ref.mat_param && got.prim_material == ref.prim_material &&
                          got.prim_type == ref.prim_type && got.prim_param == ref.prim_param;
        (void)one;
        delete second;
        delete first;
        pt_host::Session::get().reset();
        return same ? 1 : 0;
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}

// ordering contract: an object created before its material must be refused, not crash
int pth_ordering_violation()
{
    pt_host::Session::get().reset();
    lambertian m(color(1, 1, 1));
    sphere s(point3(0, 0, 0), 1, &m);
    try { s.create_hittable_on_gpu(); } catch (const pt_host::scene_error &) { return 1; }
    return 0;
}
}
