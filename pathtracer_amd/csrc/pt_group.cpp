// pt_group.cpp -- one image on several GPUs of a node, in one process: the samples of every pixel are cut into one
// contiguous range per GPU, each GPU renders all pixels of its range into its own linear accumulation buffer
// (pt_render_device, called by a worker thread per GPU that lives as long as the group), and the buffers are summed
// with ONE RCCL all-reduce over xGMI before the
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
#include <condition_variable>
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

// A worker thread per GPU, alive from pt_group_create to pt_group_destroy: a render hands every worker one job and
// waits for all of them, so a call costs two condition-variable round trips instead of G thread creations.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    pt_render_params params{};
    pt_status status = PT_OK;
    pt_stats stats{};
};

struct pt_group {
    std::vector<int> devices;
    std::vector<pt_context *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    std::vector<float *> d_sum;
    std::vector<Worker *> workers;
    size_t sum_floats = 0;
    float *h_stage = nullptr; // pinned staging for the copy of GPU 0's sum into a pageable host buffer
    size_t cap_stage = 0;
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

// GPU `rank` of `n_ranks` renders the absolute samples [*offset, *offset + *count): [sample_offset, sample_offset + spp)
// cut into n_ranks contiguous ranges that differ by at most one sample (the first spp % n_ranks ranges hold one more).
extern "C" pt_status pt_group_shard(uint32_t spp, uint32_t sample_offset, int n_ranks, int rank, uint32_t *offset, uint32_t *count)
{
    if (!offset || !count || n_ranks <= 0 || rank < 0 || rank >= n_ranks || spp < (uint32_t)n_ranks) return PT_ERR_INVALID;
    const uint32_t base = spp / (uint32_t)n_ranks, extra = spp % (uint32_t)n_ranks;
    *offset = sample_offset + (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, extra);
    *count = base + ((uint32_t)rank < extra ? 1u : 0u);
    return PT_OK;
}

static void worker_main(pt_group *g, size_t i)
{
    Worker &w = *g->workers[i];
    hipSetDevice(g->devices[i]); // per thread, once
    for (;;) {
        std::unique_lock<std::mutex> lk(w.m);
        w.cv.wait(lk, [&] { return w.has_job || w.quit; });
        if (w.quit) return;
        w.has_job = false;
        lk.unlock();
        pt_status st = PT_OK;
        if (hipMemsetAsync(g->d_sum[i], 0, g->sum_floats * sizeof(float), g->streams[i]) != hipSuccess) st = PT_ERR_DEVICE;
        else st = pt_render_device(g->ctx[i], &w.params, g->d_sum[i], g->streams[i], &w.stats);
        lk.lock();
        w.status = st;
        w.done = true;
        lk.unlock();
        w.cv.notify_all();
    }
}

extern "C" void pt_group_destroy(pt_group *g)
{
    if (!g) return;
    Rccl &r = rccl();
    for (Worker *w : g->workers) {
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lk(w->m);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (size_t i = 0; i < g->devices.size(); ++i) {
        hipSetDevice(g->devices[i]);
        if (i < g->streams.size() && g->streams[i]) hipStreamSynchronize(g->streams[i]);
        if (i < g->comms.size() && g->comms[i] && r.CommDestroy) r.CommDestroy(g->comms[i]);
        if (i < g->d_sum.size()) hipFree(g->d_sum[i]);
        if (i < g->streams.size() && g->streams[i]) hipStreamDestroy(g->streams[i]);
        if (i < g->ctx.size()) pt_context_destroy(g->ctx[i]);
    }
    if (g->h_stage) hipHostFree(g->h_stage);
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
    g->workers.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        g->workers[i] = new Worker();
        g->workers[i]->th = std::thread(worker_main, g, (size_t)i);
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

// Renders every GPU's range and sums the buffers; on return every GPU's d_sum holds the whole image's linear sums and
// every stream is idle.
static pt_status group_render_on_devices(pt_group *g, const pt_render_params *pr, pt_stats *stats)
{
    if (!g->has_scene) return gfail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    if (!pr) return gfail(g, PT_ERR_INVALID, "null parameters");
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
    // The random stream is keyed by the absolute sample index, so the image does not depend on G except for the fp32
    // association of the per-pixel sum.
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < G; ++i) {
        Worker &w = *g->workers[i];
        {
            std::lock_guard<std::mutex> lk(w.m);
            w.params = *pr;
            pt_group_shard(pr->spp, pr->sample_offset, (int)G, (int)i, &w.params.sample_offset, &w.params.spp);
            w.done = false;
            w.has_job = true;
        }
        w.cv.notify_all();
    }
    for (size_t i = 0; i < G; ++i) {
        Worker &w = *g->workers[i];
        std::unique_lock<std::mutex> lk(w.m);
        w.cv.wait(lk, [&] { return w.done; });
    }
    for (size_t i = 0; i < G; ++i)
        if (g->workers[i]->status != PT_OK)
            return gfail(g, g->workers[i]->status, "GPU " + std::to_string(g->devices[i]) + ": " + pt_last_error(g->ctx[i]));
    // one all-reduce of the linear sums (width x height x 3 floats per GPU), in place, every rank on its own stream
    ncclResult_t nr = r.GroupStart();
    for (size_t i = 0; i < G && nr == ncclSuccess; ++i)
        nr = r.AllReduce(g->d_sum[i], g->d_sum[i], n_floats, ncclFloat, ncclSum, g->comms[i], g->streams[i]);
    const ncclResult_t ne = r.GroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return gfail(g, PT_ERR_DEVICE, std::string("ncclAllReduce: ") + r.GetErrorString(nr));
    hipError_t he = hipSuccess;
    for (size_t i = 0; i < G && he == hipSuccess; ++i) {
        he = hipSetDevice(g->devices[i]);
        if (he == hipSuccess) he = hipStreamSynchronize(g->streams[i]);
    }
    if (he != hipSuccess) return gfail(g, PT_ERR_DEVICE, std::string("pt_group_render: ") + hipGetErrorString(he));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (size_t i = 0; i < G; ++i) {
            const pt_stats &part = g->workers[i]->stats;
            stats->rays += part.rays;
            stats->primary_rays += part.primary_rays;
            stats->box_tests += part.box_tests;
            stats->prim_tests += part.prim_tests;
            stats->exact_fallbacks += part.exact_fallbacks;
            stats->exact_replays += part.exact_replays;
            stats->launches_intersect += part.launches_intersect;
            stats->passes = std::max(stats->passes, part.passes);
            stats->ms_paths = std::max(stats->ms_paths, part.ms_paths);
        }
        stats->kernel = g->workers[0]->stats.kernel;
        // the whole call on the host's clock: the slowest GPU's render and the all-reduce
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return PT_OK;
}

extern "C" pt_status pt_group_render_device(pt_group *g, const pt_render_params *pr, const float **d_rgb_sum, pt_stats *stats)
{
    if (!g) return PT_ERR_INVALID;
    if (!d_rgb_sum) return gfail(g, PT_ERR_INVALID, "null output pointer");
    *d_rgb_sum = nullptr;
    const pt_status st = group_render_on_devices(g, pr, stats);
    if (st == PT_OK) *d_rgb_sum = g->d_sum[0];
    return st;
}

extern "C" pt_status pt_group_render(pt_group *g, const pt_render_params *pr, float *rgb_sum_host, pt_stats *stats)
{
    if (!g) return PT_ERR_INVALID;
    if (!rgb_sum_host) return gfail(g, PT_ERR_INVALID, "null parameters or output buffer");
    const auto t0 = std::chrono::steady_clock::now();
    const pt_status st = group_render_on_devices(g, pr, stats);
    if (st != PT_OK) return st;
    // GPU 0's copy of the sum to the host: directly into a pinned buffer (pt_host_alloc), otherwise through the group's
    // pinned staging buffer (a copy into pageable memory runs at a fifth of the link's rate)
    const size_t bytes = g->sum_floats * sizeof(float);
    hipError_t he = hipSetDevice(g->devices[0]);
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    const bool pinned = hipPointerGetAttributes(&at, rgb_sum_host) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (he == hipSuccess && pinned) {
        he = hipMemcpyAsync(rgb_sum_host, g->d_sum[0], bytes, hipMemcpyDeviceToHost, g->streams[0]);
        if (he == hipSuccess) he = hipStreamSynchronize(g->streams[0]);
    } else if (he == hipSuccess) {
        if (g->cap_stage < bytes) {
            if (g->h_stage) hipHostFree(g->h_stage);
            g->h_stage = nullptr; g->cap_stage = 0;
            he = hipHostMalloc(reinterpret_cast<void **>(&g->h_stage), bytes, hipHostMallocDefault);
            if (he == hipSuccess) g->cap_stage = bytes;
        }
        // halves: the host copies the first out of the staging buffer while the second arrives
        const size_t half = (bytes / 2) & ~(size_t)255;
        char *hs = reinterpret_cast<char *>(g->h_stage), *out = reinterpret_cast<char *>(rgb_sum_host);
        const char *src = reinterpret_cast<const char *>(g->d_sum[0]);
        if (he == hipSuccess) he = hipMemcpyAsync(hs, src, half, hipMemcpyDeviceToHost, g->streams[0]);
        if (he == hipSuccess) he = hipStreamSynchronize(g->streams[0]);
        if (he == hipSuccess) he = hipMemcpyAsync(hs + half, src + half, bytes - half, hipMemcpyDeviceToHost, g->streams[0]);
        if (he == hipSuccess) memcpy(out, hs, half);
        if (he == hipSuccess) he = hipStreamSynchronize(g->streams[0]);
        if (he == hipSuccess) memcpy(out + half, hs + half, bytes - half);
    }
    if (he != hipSuccess) return gfail(g, PT_ERR_DEVICE, std::string("pt_group_render: ") + hipGetErrorString(he));
    if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}
