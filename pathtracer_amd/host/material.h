// material.h -- lambertian, metal, dielectric, diffuse_light with the reference's constructors,
// public fields and file loaders (material.h:23-340 there).  create_material_on_gpu() snapshots
// the parameters into the session and makes d_this non-null.
#ifndef PT_HOST_MATERIAL_H
#define PT_HOST_MATERIAL_H
#include <fstream>
#include <functional>
#include <string>

#include "commons.h"
#include "hittable.h"

class material {
public:
    virtual ~material() {}
    virtual void create_material_on_gpu() = 0;
    material **d_this = nullptr;
    int pt_index = -1;

protected:
    void pt_created(int index)
    {
        pt_index = index;
        self_ = this;
        d_this = &self_;
    }
    material *self_ = nullptr;
};

inline void hittable::pt_register(uint8_t tag, const float p[6])
{
    if (!material_ptr || !material_ptr->d_this)
        throw pt_host::scene_error("create_hittable_on_gpu(): create the object's material on the GPU first");
    pt_slot = pt_host::Session::get().add_object(tag, p, material_ptr->pt_index);
    self_ = this;
    d_this = &self_;
}

namespace pt_host {
// Reads one "Name { key = value ... }" block body: lines up to, not including, the next line whose
// first character is an upper-case letter (the next block or section name).  The reference does this
// with tellg/seekg (e.g. material.h:111-120) and spins forever at end of file; this throws instead.
inline void read_block(std::ifstream &is, const std::function<void(const std::string &, const std::string &)> &on_key)
{
    std::string line;
    for (;;) {
        std::streampos pos = is.tellg();
        if (!std::getline(is, line)) throw scene_error("scene file ended inside a block (no END / next section)");
        auto kv = get_key_value(line);
        on_key(kv[0], kv[1]);
        if (!line.empty() && isupper((unsigned char)line[0])) {
            is.clear();
            is.seekg(pos);
            return;
        }
    }
}
} // namespace pt_host

class lambertian : public material {
public:
    color albedo;
    lambertian(const color a) : albedo(a) {}
    void create_material_on_gpu() override { pt_created(pt_host::Session::get().add_material(PT_LAMBERTIAN, albedo.x(), albedo.y(), albedo.z(), 0.0f)); }
    static lambertian *load_from_file(std::ifstream &file)
    {
        color a;
        pt_host::read_block(file, [&](const std::string &k, const std::string &v) { if (k == "albedo") a.load_from_string(v); });
        return new lambertian(a);
    }
};

class metal : public material {
public:
    color albedo;
    float fuzz;
    metal(const color &a, float f) : albedo(a), fuzz(f > 1.0f ? 1.0f : f) {}
    void create_material_on_gpu() override { pt_created(pt_host::Session::get().add_material(PT_METAL, albedo.x(), albedo.y(), albedo.z(), fuzz)); }
    static metal *load_from_file(std::ifstream &is)
    {
        color a;
        float f = 0.5f;
        pt_host::read_block(is, [&](const std::string &k, const std::string &v) {
            if (k == "albedo") a.load_from_string(v);
            else if (k == "fuzz") f = std::stof(v);
        });
        return new metal(a, f);
    }
};

class dielectric : public material {
public:
    float ir;
    dielectric(float index_of_refraction) : ir(index_of_refraction) {}
    void create_material_on_gpu() override { pt_created(pt_host::Session::get().add_material(PT_DIELECTRIC, ir, 0.0f, 0.0f, 0.0f)); }
    static dielectric *load_from_file(std::ifstream &is)
    {
        float ior = 1.0f;
        pt_host::read_block(is, [&](const std::string &k, const std::string &v) { if (k == "ior") ior = std::stof(v); });
        return new dielectric(ior);
    }
};

class diffuse_light : public material {
public:
    color emit;
    diffuse_light(color c) : emit(c) {}
    void create_material_on_gpu() override { pt_created(pt_host::Session::get().add_material(PT_DIFFUSE_LIGHT, emit.x(), emit.y(), emit.z(), 0.0f)); }
    static diffuse_light *load_from_file(std::ifstream &is)
    {
        color e;
        pt_host::read_block(is, [&](const std::string &k, const std::string &v) { if (k == "emit") e.load_from_string(v); });
        return new diffuse_light(e);
    }
};
#endif
