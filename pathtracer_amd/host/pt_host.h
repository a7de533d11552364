// pt_host.h -- the host-side session behind the scene API headers.
//
// In the reference every host object has a device twin made by a <<<1,1>>> kernel doing a
// device-side `new` at create_*_on_gpu() time (sphere.h:31-34, material.h:125-128, ...).  Here the
// same calls take a SNAPSHOT of the object's parameters into this registry (same moment, same
// ordering contract: materials before the objects that use them, objects before
// hittable_list::copy_list_to_gpu).  A world and a camera become device data only when they are
// first rendered: pt_host::render() flattens the snapshot, uploads it through the C ABI
// (pt_scene_upload) and runs the wavefront integrator (pt_render).
#ifndef PT_HOST_SESSION_H
#define PT_HOST_SESSION_H
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "vec3.h"

namespace pt_host {

struct scene_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Options {
    int max_depth = 5;         // ray_color's depth argument in the reference (main.cu:93)
    uint64_t seed = 1984;      // main.cu:34
    int device = 0;
    // > 0: render on the GPUs 0 .. gpus-1 of this node (pt_group: the samples of every pixel cut into one range per GPU,
    // the accumulation buffers summed with one RCCL all-reduce).  The reference renders on one device (main.cu:181-186).
    int gpus = 0;
    // The reference's file loaders parse a cone's / cylinder's `center` and then drop it, so those
    // objects always sit at the origin (cone.h:58-61,74; cylinder.h:57-60,73).  false reproduces that.
    bool honor_cone_cylinder_center = false;
};

struct Flat { // what pt_scene_desc points into
    std::vector<uint8_t> prim_type, mat_type;
    std::vector<float> prim_param, mat_param;
    std::vector<uint32_t> prim_material;
};

class Session {
public:
    static Session &get()
    {
        static Session s;
        return s;
    }
    Options options;

    // create_material_on_gpu(): snapshot; returns the material's index
    int add_material(uint8_t tag, float a, float b, float c, float d)
    {
        mat_tag_.push_back(tag);
        const float p[4] = {a, b, c, d};
        mat_par_.insert(mat_par_.end(), p, p + 4);
        return (int)mat_tag_.size() - 1;
    }
    // create_hittable_on_gpu(): snapshot; returns the object's slot
    int add_object(uint8_t tag, const float p[6], int material_index)
    {
        if (material_index < 0 || material_index >= (int)mat_tag_.size())
            throw scene_error("create_hittable_on_gpu(): the object's material has not been created on the GPU yet");
        obj_tag_.push_back(tag);
        obj_par_.insert(obj_par_.end(), p, p + 6);
        obj_mat_.push_back((uint32_t)material_index);
        return (int)obj_tag_.size() - 1;
    }
    // hittable_list::copy_list_to_gpu(): the world is the listed slots, in list order
    void set_world(const void *handle, const std::vector<int> &slots)
    {
        for (int s : slots)
            if (s < 0 || s >= (int)obj_tag_.size()) throw scene_error("copy_list_to_gpu(): a listed object has not been created on the GPU yet");
        worlds_[handle] = slots;
        uploaded_world_ = nullptr;
    }
    void drop_world(const void *handle)
    {
        worlds_.erase(handle);
        if (uploaded_world_ == handle) uploaded_world_ = nullptr;
    }
    void set_camera(const void *handle, const pt_camera &c)
    {
        cameras_[handle] = c;
        uploaded_camera_ = nullptr;
    }
    void drop_camera(const void *handle)
    {
        cameras_.erase(handle);
        if (uploaded_camera_ == handle) uploaded_camera_ = nullptr;
    }

    // The flattened arrays of a world, as pt_scene_upload would receive them (no GPU involved).
    Flat flatten(const void *world_handle) const
    {
        auto it = worlds_.find(world_handle);
        if (it == worlds_.end()) throw scene_error("unknown world handle: call create_hittable_on_gpu() and copy_list_to_gpu() first");
        Flat f;
        // The world's materials: the run of the registry from the first to the last one its objects use, in creation order
        // (a scene creates its materials together, scene.h:216-237), renumbered from 0.  A second scene alive in the same
        // process therefore neither sees nor uploads the first one's materials.
        uint32_t lo = UINT32_MAX, hi = 0;
        for (int s : it->second) {
            lo = obj_mat_[s] < lo ? obj_mat_[s] : lo;
            hi = obj_mat_[s] > hi ? obj_mat_[s] : hi;
        }
        if (lo <= hi) {
            f.mat_type.assign(mat_tag_.begin() + lo, mat_tag_.begin() + hi + 1);
            f.mat_param.assign(mat_par_.begin() + 4 * (size_t)lo, mat_par_.begin() + 4 * ((size_t)hi + 1));
        }
        for (int s : it->second) {
            f.prim_type.push_back(obj_tag_[s]);
            f.prim_param.insert(f.prim_param.end(), obj_par_.begin() + 6 * (size_t)s, obj_par_.begin() + 6 * (size_t)s + 6);
            f.prim_material.push_back(obj_mat_[s] - lo);
        }
        return f;
    }
    const pt_camera &camera(const void *handle) const
    {
        auto it = cameras_.find(handle);
        if (it == cameras_.end()) throw scene_error("unknown camera handle: call create_camera_on_gpu() first");
        return it->second;
    }

    // Linear sums of `spp` samples per pixel (pt_render); uploads the world/camera pair if it changed.  The sums arrive in
    // a page-locked buffer the session keeps (the device writes it at the link's rate; the reference reads its managed fb
    // back by page migration, main.cu:167-170, :192-205): valid until the next call.
    const float *render_sum(const void *world_handle, const void *camera_handle, const float background[3], int width, int height, int spp,
                            pt_stats *stats = nullptr)
    {
        ensure_context();
        if (uploaded_world_ != world_handle || uploaded_camera_ != camera_handle || std::memcmp(uploaded_bg_, background, 12) != 0) {
            Flat f = flatten(world_handle);
            pt_scene_desc d{};
            d.n_prims = (uint32_t)f.prim_type.size();
            d.prim_type = f.prim_type.data();
            d.prim_param = f.prim_param.data();
            d.prim_material = f.prim_material.data();
            d.n_materials = (uint32_t)f.mat_type.size();
            d.mat_type = f.mat_type.data();
            d.mat_param = f.mat_param.data();
            d.camera = camera(camera_handle);
            std::memcpy(d.background, background, 12);
            if (grp_) check_group(pt_group_scene_upload(grp_, &d), "pt_group_scene_upload");
            else check(pt_scene_upload(ctx_, &d), "pt_scene_upload");
            uploaded_world_ = world_handle;
            uploaded_camera_ = camera_handle;
            std::memcpy(uploaded_bg_, background, 12);
        }
        pt_render_params p{};
        p.width = width;
        p.height = height;
        p.spp = (uint32_t)spp;
        p.max_depth = options.max_depth;
        p.seed = options.seed;
        const size_t bytes = (size_t)width * height * 3 * sizeof(float);
        if (bytes > film_bytes_) {
            pt_host_free(film_);
            film_bytes_ = 0;
            film_ = static_cast<float *>(pt_host_alloc(bytes));
            if (!film_) throw scene_error("pt_host_alloc: no page-locked memory for the film");
            film_bytes_ = bytes;
        }
        if (grp_) check_group(pt_group_render(grp_, &p, film_, stats), "pt_group_render");
        else check(pt_render(ctx_, &p, film_, stats), "pt_render");
        return film_;
    }

    // Forgets every snapshot (the scene on the GPU stays until the next upload).
    void reset()
    {
        mat_tag_.clear(); obj_tag_.clear(); mat_par_.clear(); obj_par_.clear(); obj_mat_.clear();
        worlds_.clear(); cameras_.clear();
        uploaded_world_ = uploaded_camera_ = nullptr;
    }

    ~Session()
    {
        pt_host_free(film_);
        if (ctx_) pt_context_destroy(ctx_);
        if (grp_) pt_group_destroy(grp_);
    }

private:
    Session() {}
    void ensure_context()
    {
        if (ctx_ || grp_) return;
        if (options.gpus > 0) {
            std::vector<int> ids(options.gpus);
            for (int i = 0; i < options.gpus; ++i) ids[i] = i;
            if (pt_group_create(options.gpus, ids.data(), &grp_) != PT_OK)
                throw scene_error("pt_group_create failed: fewer usable GPUs than asked for, or RCCL is not loadable (this renderer has no CPU fallback)");
        } else if (pt_context_create(options.device, &ctx_) != PT_OK) {
            throw scene_error("pt_context_create failed: no usable GPU (this renderer has no CPU fallback)");
        }
    }
    void check_group(pt_status st, const char *what)
    {
        if (st != PT_OK) throw scene_error(std::string(what) + ": " + pt_group_last_error(grp_));
    }
    void check(pt_status st, const char *what)
    {
        if (st != PT_OK) throw scene_error(std::string(what) + ": " + pt_last_error(ctx_));
    }
    std::vector<uint8_t> mat_tag_, obj_tag_;
    std::vector<float> mat_par_, obj_par_;
    std::vector<uint32_t> obj_mat_;
    std::map<const void *, std::vector<int>> worlds_;
    std::map<const void *, pt_camera> cameras_;
    pt_context *ctx_ = nullptr;
    pt_group *grp_ = nullptr;
    float *film_ = nullptr; // page-locked (pt_host_alloc)
    size_t film_bytes_ = 0;
    const void *uploaded_world_ = nullptr, *uploaded_camera_ = nullptr;
    float uploaded_bg_[3] = {0, 0, 0};
};

inline Options &options() { return Session::get().options; }

} // namespace pt_host
#endif
