// hittable_list.h -- the world: an array of hittable* in list order (hittable_list.h:14-85 in the
// reference).  The order matters: at equal distance the later entry wins, and a steep cone's answer
// depends on what earlier entries hit; the GPU path reproduces the linear scan's result.
#ifndef PT_HOST_HITTABLE_LIST_H
#define PT_HOST_HITTABLE_LIST_H
#include <vector>

#include "hittable.h"

class hittable_list : public hittable {
public:
    hittable **list = nullptr;
    hittable **d_list = nullptr; // non-null once created
    int list_size = 0;

    hittable_list() {}
    hittable_list(hittable **list, int list_size) : list(list), list_size(list_size) {}
    ~hittable_list() override
    {
        if (d_this) pt_host::Session::get().drop_world(d_this);
    }

    void create_hittable_on_gpu() override
    {
        self_ = this;
        d_this = &self_;
        d_list = list;
    }
    // every list[i] must have been created on the GPU already (hittable_list.h:50-56)
    void copy_list_to_gpu()
    {
        if (list_size > 0 && d_list != nullptr && d_this != nullptr) {
            std::vector<int> slots;
            for (int i = 0; i < list_size; ++i) {
                if (!list[i] || !list[i]->d_this) throw pt_host::scene_error("copy_list_to_gpu(): list entry " + std::to_string(i) + " was not created on the GPU");
                slots.push_back(list[i]->pt_slot);
            }
            pt_host::Session::get().set_world(d_this, slots);
        }
    }
    bool bounding_box(float t0, float t1, aabb &box) const override
    {
        if (list_size < 1) return false;
        aabb tmp;
        for (int i = 0; i < list_size; ++i) {
            if (!list[i]->bounding_box(t0, t1, tmp)) return false;
            box = i == 0 ? tmp : surrounding_box(box, tmp);
        }
        return true;
    }
};
#endif
