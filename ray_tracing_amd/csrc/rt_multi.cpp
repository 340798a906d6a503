// rt_multi.cpp -- one image on several GPUs of one node from ONE host process, no Python: rt_render_multi_to_host and
// rt_sample_range (include/rt_amd.h).
//
// The reference splits an image by SAMPLE (one task per sample, src/main.cpp:543-557) and sums the per-sample images
// afterwards (src/main.cpp:720-726).  Here device g of G renders the samples rt_sample_range gives it into its own
// float4[nx*ny] sums (rt_render on its own stream, all devices at once), the sums are added onto the first device with
// ONE RCCL reduce over xGMI (ncclReduce, float, sum, 4*nx*ny elements: 16 MiB at 1024 x 1024), and that device resolves
// (rt_resolve).  Paths are keyed by (pixel, sample), so the set of paths does not depend on G.
// RCCL is opened at run time (dlopen): the library does not depend on it unless this entry point is used.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_error.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *problem = nullptr;
};

// the process's RCCL (whatever copy is loaded already -- a Python host has torch's -- else librccl.so.1 from the loader's path)
const Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) {
            r.problem = "librccl.so.1 cannot be opened (dlopen)";
            return;
        }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.Reduce = (decltype(r.Reduce))dlsym(r.lib, "ncclReduce");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString)
            r.problem = "librccl lacks one of ncclCommInitAll / ncclCommDestroy / ncclReduce / ncclGroupStart / ncclGroupEnd / ncclGetErrorString";
    });
    return r;
}

struct Rank { // what one device holds for the call
    int device = -1;
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    float *d_accum = nullptr;
    ncclComm_t comm = nullptr;
};

void release(std::vector<Rank> &ranks, uint32_t *d_image)
{
    const Rccl &R = rccl();
    for (Rank &k : ranks) {
        if (k.device < 0) continue;
        (void)hipSetDevice(k.device);
        if (k.comm && R.CommDestroy) (void)R.CommDestroy(k.comm);
        if (k.d_accum) (void)hipFree(k.d_accum);
        if (k.stream) (void)hipStreamDestroy(k.stream);
        if (k.scene) rt_scene_destroy(k.scene);
    }
    if (d_image && !ranks.empty() && ranks[0].device >= 0) {
        (void)hipSetDevice(ranks[0].device);
        (void)hipFree(d_image);
    }
}

} // namespace

#define RTM_HIP(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            release(ranks, d_image);                                                                  \
            return rt_fail(RT_ERR_NO_DEVICE, "rt_render_multi_to_host: %s: %s", #call, hipGetErrorString(e_)); \
        }                                                                                             \
    } while (0)
#define RTM_NCCL(call)                                                                                \
    do {                                                                                              \
        ncclResult_t n_ = (call);                                                                     \
        if (n_ != ncclSuccess) {                                                                      \
            const char *msg_ = R.GetErrorString(n_);                                                  \
            release(ranks, d_image);                                                                  \
            return rt_fail(RT_ERR_NO_DEVICE, "rt_render_multi_to_host: %s: %s", #call, msg_);           \
        }                                                                                             \
    } while (0)
#define RTM_RT(call)                                                                                  \
    do {                                                                                              \
        int s_ = (call);                                                                              \
        if (s_ != RT_OK) {                                                                            \
            release(ranks, d_image);                                                                  \
            return s_; /* rt_last_error() holds the callee's message */                               \
        }                                                                                             \
    } while (0)

extern "C" int rt_sample_range(int32_t rank, int32_t world, int32_t sample_begin, int32_t sample_end, int32_t *begin, int32_t *end)
{
    if (!begin || !end || world < 1 || rank < 0 || rank >= world || sample_begin < 0 || sample_end < sample_begin)
        return rt_fail(RT_ERR_INVALID, "rt_sample_range: rank %d of %d, samples [%d,%d)", rank, world, sample_begin, sample_end);
    const int64_t n = (int64_t)sample_end - sample_begin;
    *begin = sample_begin + (int32_t)(rank * n / world);
    *end = sample_begin + (int32_t)((rank + 1) * n / world);
    return RT_OK;
}

extern "C" int rt_render_multi_to_host(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, const rt_render_params *p,
                                       uint32_t flags, float *h_accum, uint32_t *h_rgba8)
{
    if (!desc || !devices || !p || !h_accum) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: null argument");
    if (p->struct_size != sizeof(rt_render_params)) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: bad rt_render_params");
    if (n_devices < 1 || n_devices > 64) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: %d devices", n_devices);
    if (flags & ~RT_MULTI_REDUCE_ALWAYS) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: unknown flags 0x%x", flags);
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: device %d is listed twice", devices[i]);
    const bool reduce = n_devices > 1 || (flags & RT_MULTI_REDUCE_ALWAYS);
    const Rccl &R = rccl();
    if (reduce && R.problem) return rt_fail(RT_ERR_UNSUPPORTED, "rt_render_multi_to_host: %s", R.problem);

    const size_t npix = (size_t)p->nx * p->ny, bytes = npix * 16;
    std::vector<Rank> ranks(n_devices);
    uint32_t *d_image = nullptr;
    // one scene, one stream and one accumulator per device (the scene tables are replicated: 2.6 KiB for the Cornell box)
    for (int i = 0; i < n_devices; ++i) {
        Rank &k = ranks[i];
        RTM_RT(rt_scene_create(desc, devices[i], &k.scene));
        k.device = devices[i];
        RTM_HIP(hipSetDevice(k.device));
        RTM_HIP(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
        RTM_HIP(hipMalloc(&k.d_accum, bytes));
    }
    if (reduce) {
        std::vector<ncclComm_t> comms(n_devices);
        std::vector<int> devs(devices, devices + n_devices);
        RTM_NCCL(R.CommInitAll(comms.data(), n_devices, devs.data()));
        for (int i = 0; i < n_devices; ++i) ranks[i].comm = comms[i];
    }
    // every device renders its range; the launches return at once, so the devices run side by side
    for (int i = 0; i < n_devices; ++i) {
        rt_render_params q = *p;
        q.flags &= ~RT_RENDER_ACCUMULATE;
        RTM_RT(rt_sample_range(i, n_devices, p->sample_begin, p->sample_end, &q.sample_begin, &q.sample_end));
        RTM_RT(rt_render(ranks[i].scene, &q, ranks[i].d_accum, ranks[i].stream));
    }
    // one sum-reduce onto the first device, in stream order behind each device's render
    if (reduce) {
        RTM_NCCL(R.GroupStart());
        for (int i = 0; i < n_devices; ++i) {
            (void)hipSetDevice(ranks[i].device);
            ncclResult_t n_ = R.Reduce(ranks[i].d_accum, ranks[i].d_accum, 4 * npix, ncclFloat, ncclSum, 0, ranks[i].comm, ranks[i].stream);
            if (n_ != ncclSuccess) {
                (void)R.GroupEnd();
                const char *msg_ = R.GetErrorString(n_);
                release(ranks, d_image);
                return rt_fail(RT_ERR_NO_DEVICE, "rt_render_multi_to_host: ncclReduce: %s", msg_);
            }
        }
        RTM_NCCL(R.GroupEnd());
    }
    // the first device resolves (src/main.cpp:708-743) and hands both images to the host
    RTM_HIP(hipSetDevice(ranks[0].device));
    if (h_rgba8) {
        RTM_HIP(hipMalloc(&d_image, npix * 4));
        RTM_RT(rt_resolve(ranks[0].d_accum, p->nx, p->ny, d_image, ranks[0].stream));
    }
    for (int i = n_devices - 1; i >= 0; --i) { // the first device last: its stream holds the reduce's result and the resolve
        RTM_HIP(hipSetDevice(ranks[i].device));
        RTM_HIP(hipStreamSynchronize(ranks[i].stream));
    }
    RTM_HIP(hipMemcpy(h_accum, ranks[0].d_accum, bytes, hipMemcpyDeviceToHost));
    if (h_rgba8) RTM_HIP(hipMemcpy(h_rgba8, d_image, npix * 4, hipMemcpyDeviceToHost));
    release(ranks, d_image);
    return RT_OK;
}
