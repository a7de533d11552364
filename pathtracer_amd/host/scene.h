// scene.h -- the three scene constructors of the reference (scene.h:15-316 there): the default
// scene, the Cornell-style box, and the scene.txt loader.  Section order of the file:
//   Image{...} (read by main) -> Scene -> options until "Camera" -> camera keys until "Materials"
//   -> material blocks until "Objects" -> object blocks until "END".
// Unlike the reference, a file that ends early, a malformed number or an undefined material index
// raises pt_host::scene_error instead of looping forever or indexing out of bounds.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "aarect.h"
#include "camera.h"
#include "commons.h"
#include "cone.h"
#include "cylinder.h"
#include "hittable_list.h"
#include "material.h"
#include "sphere.h"

class scene {
public:
    std::vector<hittable *> hitables; // (sic) the reference's spelling
    std::vector<material *> materials;
    hittable_list *world = nullptr;
    color background_color;
    camera *cam = nullptr;

    // ground sphere, one sphere, one emissive rectangle; 16:9 camera at (26,3,6)
    scene()
    {
        auto ground_material = new lambertian(color(0.5, 0.5, 0.5));
        own(ground_material)->create_material_on_gpu();
        hitables.push_back(new sphere(point3(0, -1000, 0), 1000, ground_material));
        auto material2 = new lambertian(color(0.4f, 0.2f, 0.1f));
        own(material2)->create_material_on_gpu();
        hitables.push_back(new sphere(point3(0, 2, 0), 2.0f, material2));
        auto difflight = new diffuse_light(color(4, 4, 4));
        own(difflight)->create_material_on_gpu();
        hitables.push_back(new xy_rect(3, 5, 1, 3, -2, difflight));
        finish_world();
        background_color = color(0, 0, 0);
        auto dist_to_focus = 10.0;
        auto aperture = 0.1;
        cam = new camera(point3(26, 3, 6), point3(0, 2, 0), vec3(0, 1, 0), 20.0f, 16.0f / 9.0f, aperture, dist_to_focus);
        cam->create_camera_on_gpu();
    }

    // five rectangular walls, a cylinder (height 200, radius 64) and a rectangular light; the argument is unused
    scene(int)
    {
        auto add = [&](material *m, hittable *h) { own(m)->create_material_on_gpu(); hitables.push_back(h); };
        material *m;
        m = new lambertian(color(.65f, .05f, .05f)); add(m, new yz_rect(0, 555, 0, 555, 0, m));
        m = new lambertian(color(.73f, .73f, .73f)); add(m, new xz_rect(0, 555, 0, 555, 0, m));
        m = new lambertian(color(.73f, .73f, .73f)); add(m, new xz_rect(0, 555, 0, 555, 555, m));
        m = new lambertian(color(.73f, .73f, .73f)); add(m, new xy_rect(0, 555, 0, 555, 555, m));
        m = new lambertian(color(.12f, .45f, .15f)); add(m, new cylinder2(point3(348, 0, 200), 200, 64, m));
        m = new lambertian(color(.12f, .45f, .15f)); add(m, new yz_rect(0, 555, 0, 555, 555, m));
        m = new diffuse_light(color(15, 15, 15));    add(m, new xz_rect(213, 343, 227, 332, 554, m));
        finish_world();
        background_color = color(0, 0, 0);
        auto dist_to_focus = 10.0;
        auto aperture = 0.1;
        cam = new camera(point3(278, 278, -800), point3(278, 278, 0), vec3(0, 1, 0), 40.0f, 1.0f, aperture, dist_to_focus);
        cam->create_camera_on_gpu();
    }

    // the rest of scene.txt, positioned just after its "Scene" line
    scene(std::ifstream &file, float aspect_ratio)
    {
        std::string line;
        try {
            try {
                loadOptions(file, line);
                loadCamera(file, line, aspect_ratio);
                loadMaterials(file, line);
                loadObjects(file, line);
            } catch (const std::invalid_argument &e) {
                throw pt_host::scene_error(std::string("scene file: malformed number near '") + line + "'");
            } catch (const std::out_of_range &e) {
                throw pt_host::scene_error(std::string("scene file: number out of range near '") + line + "'");
            }
            if (hitables.empty()) throw pt_host::scene_error("scene file: the Objects section is empty");
            for (material *m : materials) m->create_material_on_gpu();
            finish_world();
            cam->create_camera_on_gpu();
        } catch (...) { // a constructor that throws runs no destructor: what the file gave so far is freed here
            release();
            throw;
        }
    }

    void loadOptions(std::ifstream &file, std::string &line)
    {
        do {
            next_line(file, line, "Camera");
            auto kv = get_key_value(line);
            if (kv[0] == "background_color") {
                color bc;
                bc.load_from_string(kv[1]);
                background_color = bc;
            }
        } while (line != "Camera");
    }

    void loadCamera(std::ifstream &file, std::string &line, float aspect_ratio)
    {
        point3 lookfrom, lookat;
        vec3 vup = vec3(0, 1, 0);
        float vfov = 20, dist_to_focus = 10, aperture = 0.1f;
        do {
            next_line(file, line, "Materials");
            auto kv = get_key_value(line);
            if (kv[0] == "lookfrom") lookfrom.load_from_string(kv[1]);
            else if (kv[0] == "lookat") lookat.load_from_string(kv[1]);
            else if (kv[0] == "vfov") vfov = std::stof(kv[1]);
            else if (kv[0] == "distance_to_focus") dist_to_focus = std::stof(kv[1]);
            else if (kv[0] == "aperture") aperture = std::stof(kv[1]);
            else if (kv[0] == "vup") vup.load_from_string(kv[1]);
        } while (line != "Materials");
        cam = new camera(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, dist_to_focus);
    }

    void loadMaterials(std::ifstream &file, std::string &line)
    {
        do {
            next_line(file, line, "Objects");
            if (line == "Lambertian") materials.push_back(lambertian::load_from_file(file));
            else if (line == "Dielectric") materials.push_back(dielectric::load_from_file(file));
            else if (line == "Metal") materials.push_back(metal::load_from_file(file));
            else if (line == "Diffuse_light") materials.push_back(diffuse_light::load_from_file(file));
        } while (line != "Objects");
    }

    void loadObjects(std::ifstream &file, std::string &line)
    {
        do {
            next_line(file, line, "END");
            if (line == "Sphere") hitables.push_back(sphere::init_from_file(file, materials));
            else if (line == "Cylinder") hitables.push_back(cylinder2::init_from_file(file, materials));
            else if (line == "Cone") hitables.push_back(cone::init_from_file(file, materials));
            else if (line == "Xy_rect") hitables.push_back(xy_rect::load_from_file(file, materials));
            else if (line == "Xz_rect") hitables.push_back(xz_rect::load_from_file(file, materials));
            else if (line == "Yz_rect") hitables.push_back(yz_rect::load_from_file(file, materials));
        } while (line != "END");
    }

    ~scene() { release(); }

private:
    void release()
    {
        for (hittable *h : hitables) delete h;
        for (material *m : materials) delete m;
        hitables.clear();
        materials.clear();
        if (world) {
            delete[] world->list;
            delete world;
            world = nullptr;
        }
        delete cam;
        cam = nullptr;
    }
    material *own(material *m)
    {
        materials.push_back(m);
        return m;
    }
    // objects -> GPU snapshots -> world list, in the reference's order (scene.h:37-45)
    void finish_world()
    {
        hittable **list = new hittable *[hitables.size()];
        for (size_t i = 0; i < hitables.size(); ++i) list[i] = hitables[i];
        world = new hittable_list(list, (int)hitables.size()); // owned from here on, whatever a create call below throws
        for (size_t i = 0; i < hitables.size(); ++i) list[i]->create_hittable_on_gpu();
        world->create_hittable_on_gpu();
        world->copy_list_to_gpu();
    }
    static void next_line(std::ifstream &file, std::string &line, const char *expected_section)
    {
        if (!std::getline(file, line))
            throw pt_host::scene_error(std::string("scene file ended before the '") + expected_section + "' line");
        if (!line.empty() && line.back() == '\r') line.pop_back();
    }
};
