// pt_group.cpp -- one image on several GPUs of a node, in one process: the samples of every pixel are cut into one
// contiguous range per GPU, each GPU renders all pixels of its range into its own linear accumulation buffer
// (pt_render_device, one host thread per GPU), and the buffers are summed with ONE RCCL all-reduce over xGMI before the
// non-linear finish (mean, sqrt gamma: main.cu:95-98) is applied by the caller.  The reference has nothing of the kind:
// it renders on one device (main.cu:181-186).
//
// Built on the public C ABI only (a group is a set of ordinary contexts).  RCCL is loaded with dlopen when the first
// group is created, so that programs that never use a group do not depend on it, and a process that already holds an
// RCCL (PyTorch's) shares that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pt_api.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("RCCL is not loadable: ") + dlerror();
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("RCCL lacks ") + n;
            return p;
        };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

} // namespace

struct pt_group {
    std::vector<int> devices;
    std::vector<pt_context *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<float *> d_sum;
    size_t sum_floats = 0;
    bool has_scene = false;
    std::string err;
};

static pt_status gfail(pt_group *g, pt_status st, const std::string &msg)
{
    if (g) g->err = msg;
    return st;
}

extern "C" const char *pt_group_last_error(const pt_group *g) { return g ? g->err.c_str() : "null group"; }
extern "C" int pt_group_size(const pt_group *g) { return g ? (int)g->devices.size() : 0; }

extern "C" void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    Rccl &r = rccl();
    for (size_t i = 0; i < g->devices.size(); ++i) {
        hipSetDevice(g->devices[i]);
        if (i < g->streams.size() && g->streams[i]) hipStreamSynchronize(g->streams[i]);
        if (i < g->comms.size() && g->comms[i] && r.CommDestroy) r.CommDestroy(g->comms[i]);
        if (i < g->d_sum.size()) hipFree(g->d_sum[i]);
        if (i < g->streams.size() && g->streams[i]) hipStreamDestroy(g->streams[i]);
        if (i < g->ctx.size()) pt_context_destroy(g->ctx[i]);
    }
    delete g;
}

extern "C" pt_status pt_group_create(int n_devices, const int *device_ids, pt_group **out)
{
    if (!out) return PT_ERR_INVALID;
    *out = nullptr;
    if (n_devices <= 0 || n_devices > 64 || !device_ids) return PT_ERR_INVALID;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (device_ids[i] == device_ids[j]) return PT_ERR_INVALID; // one rank per GPU
    Rccl &r = rccl();
    if (!r.error.empty()) return PT_ERR_DEVICE;
    pt_group *g = new pt_group();
    g->devices.assign(device_ids, device_ids + n_devices);
    g->ctx.assign(n_devices, nullptr);
    g->streams.assign(n_devices, nullptr);
    g->comms.assign(n_devices, nullptr);
    g->d_sum.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        if (pt_context_create(device_ids[i], &g->ctx[i]) != PT_OK || hipSetDevice(device_ids[i]) != hipSuccess ||
            hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking) != hipSuccess) {
            pt_group_destroy(g);
            return PT_ERR_DEVICE;
        }
    }
    if (r.CommInitAll(g->comms.data(), n_devices, device_ids) != ncclSuccess) {
        for (auto &c : g->comms) c = nullptr;
        pt_group_destroy(g);
        return PT_ERR_DEVICE;
    }
    *out = g;
    return PT_OK;
}

extern "C" pt_status pt_group_configure(pt_group *g, const pt_tuning *t)
{
    if (!g || !t) return PT_ERR_INVALID;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const pt_status st = pt_context_configure(g->ctx[i], t);
        if (st != PT_OK) return gfail(g, st, pt_last_error(g->ctx[i]));
    }
    return PT_OK;
}

// The scene and its tree are replicated: every GPU traces whole paths of its own samples.
extern "C" pt_status pt_group_scene_upload(pt_group *g, const pt_scene_desc *desc)
{
    if (!g) return PT_ERR_INVALID;
    const size_t n = g->ctx.size();
    std::vector<pt_status> st(n, PT_OK);
    std::vector<std::thread> th;
    for (size_t i = 0; i < n; ++i) th.emplace_back([&, i] { st[i] = pt_scene_upload(g->ctx[i], desc); });
    for (auto &t : th) t.join();
    for (size_t i = 0; i < n; ++i)
        if (st[i] != PT_OK) return gfail(g, st[i], "GPU " + std::to_string(g->devices[i]) + ": " + pt_last_error(g->ctx[i]));
    g->has_scene = true;
    return PT_OK;
}

extern "C" pt_status pt_group_render(pt_group *g, const pt_render_params *pr, float *rgb_sum_host, pt_stats *stats)
{
    if (!g) return PT_ERR_INVALID;
    if (!g->has_scene) return gfail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    if (!pr || !rgb_sum_host) return gfail(g, PT_ERR_INVALID, "null parameters or output buffer");
    if (pr->width <= 0 || pr->height <= 0 || pr->spp == 0) return gfail(g, PT_ERR_INVALID, "width, height and spp must be positive");
    const size_t G = g->ctx.size();
    if (pr->spp < G) return gfail(g, PT_ERR_INVALID, std::to_string(pr->spp) + " samples cannot be split over " + std::to_string(G) + " GPUs: every GPU needs at least one");
    Rccl &r = rccl();
    const size_t n_floats = (size_t)pr->width * pr->height * 3;
    if (n_floats != g->sum_floats) {
        for (size_t i = 0; i < G; ++i) {
            hipSetDevice(g->devices[i]);
            hipFree(g->d_sum[i]);
            g->d_sum[i] = nullptr;
            if (hipMalloc(&g->d_sum[i], n_floats * sizeof(float)) != hipSuccess) {
                g->sum_floats = 0;
                return gfail(g, PT_ERR_DEVICE, "accumulation buffer allocation failed on GPU " + std::to_string(g->devices[i]));
            }
        }
        g->sum_floats = n_floats;
    }
    // GPU i of G renders the absolute samples [offset + i spp / G, offset + (i + 1) spp / G): contiguous ranges that
    // differ by at most one sample.  The random stream is keyed by the absolute sample index, so the image does not
    // depend on G except for the fp32 association of the per-pixel sum.
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<pt_status> st(G, PT_OK);
    std::vector<pt_stats> part(G);
    std::vector<std::thread> th;
    const uint32_t base = pr->spp / (uint32_t)G, extra = pr->spp % (uint32_t)G;
    for (size_t i = 0; i < G; ++i) {
        th.emplace_back([&, i] {
            pt_render_params p = *pr;
            p.spp = base + (i < extra ? 1u : 0u);
            p.sample_offset = pr->sample_offset + (uint32_t)i * base + std::min<uint32_t>((uint32_t)i, extra);
            if (hipSetDevice(g->devices[i]) != hipSuccess || hipMemsetAsync(g->d_sum[i], 0, n_floats * sizeof(float), g->streams[i]) != hipSuccess) {
                st[i] = PT_ERR_DEVICE;
                return;
            }
            st[i] = pt_render_device(g->ctx[i], &p, g->d_sum[i], g->streams[i], &part[i]);
        });
    }
    for (auto &t : th) t.join();
    for (size_t i = 0; i < G; ++i)
        if (st[i] != PT_OK) return gfail(g, st[i], "GPU " + std::to_string(g->devices[i]) + ": " + pt_last_error(g->ctx[i]));
    // one all-reduce of the linear sums (width x height x 3 floats per GPU), in place, every rank on its own stream
    ncclResult_t nr = r.GroupStart();
    for (size_t i = 0; i < G && nr == ncclSuccess; ++i)
        nr = r.AllReduce(g->d_sum[i], g->d_sum[i], n_floats, ncclFloat, ncclSum, g->comms[i], g->streams[i]);
    const ncclResult_t ne = r.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return gfail(g, PT_ERR_DEVICE, std::string("ncclAllReduce: ") + r.GetErrorString(nr));
    hipError_t he = hipSetDevice(g->devices[0]);
    if (he == hipSuccess) he = hipMemcpyAsync(rgb_sum_host, g->d_sum[0], n_floats * sizeof(float), hipMemcpyDeviceToHost, g->streams[0]);
    for (size_t i = 0; i < G && he == hipSuccess; ++i) {
        he = hipSetDevice(g->devices[i]);
        if (he == hipSuccess) he = hipStreamSynchronize(g->streams[i]);
    }
    if (he != hipSuccess) return gfail(g, PT_ERR_DEVICE, std::string("pt_group_render: ") + hipGetErrorString(he));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (size_t i = 0; i < G; ++i) {
            stats->rays += part[i].rays;
            stats->primary_rays += part[i].primary_rays;
            stats->box_tests += part[i].box_tests;
            stats->prim_tests += part[i].prim_tests;
            stats->exact_fallbacks += part[i].exact_fallbacks;
            stats->exact_replays += part[i].exact_replays;
            stats->launches_intersect += part[i].launches_intersect;
            stats->passes = std::max(stats->passes, part[i].passes);
            stats->ms_paths = std::max(stats->ms_paths, part[i].ms_paths);
        }
        stats->kernel = part[0].kernel;
        // the whole call on the host's clock: the slowest GPU's render, the all-reduce and the copy to the host
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PT_OK;
}
