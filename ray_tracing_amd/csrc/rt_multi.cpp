// rt_multi.cpp -- images on several GPUs of one node from ONE host process, no Python: rt_multi_create /
// rt_multi_render_to_host / rt_multi_destroy (a context built once), rt_render_multi_to_host (the one-shot form) and
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
#include <new>
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

struct Rank { // what one device holds for the life of a context
    int device = -1;
    rt_scene *scene = nullptr;
    hipStream_t stream = nullptr;
    float *d_accum = nullptr;
    ncclComm_t comm = nullptr;
};

} // namespace

// A multi-GPU context: built once (scenes, streams, ONE communicator), used for any number of images -- the reference
// builds its thread pool once and renders with it (src/main.cpp:523-561).  Accumulators and the 8-bit image grow to the
// largest image asked for and are kept.
struct rt_multi {
    std::vector<Rank> ranks;
    bool reduce = false;
    size_t cap_pixels = 0;       // pixels the accumulators (and d_image) hold
    uint32_t *d_image = nullptr; // on ranks[0].device
    std::mutex mutex;            // one image at a time per context
};

namespace {

// Everything the context holds goes, in an order that leaves nothing queued behind a destroyed object: first every
// stream is drained (a failed call may have left renders or the reduce in flight; errors are ignored, the point is to
// wait), then communicators, buffers, streams, scenes.
void release(rt_multi *m)
{
    if (!m) return;
    const Rccl &R = rccl();
    for (Rank &k : m->ranks) {
        if (k.device < 0 || !k.stream) continue;
        (void)hipSetDevice(k.device);
        (void)hipStreamSynchronize(k.stream);
    }
    for (Rank &k : m->ranks) {
        if (k.device < 0) continue;
        (void)hipSetDevice(k.device);
        if (k.comm && R.CommDestroy) (void)R.CommDestroy(k.comm);
        if (k.d_accum) (void)hipFree(k.d_accum);
        if (k.stream) (void)hipStreamDestroy(k.stream);
        if (k.scene) rt_scene_destroy(k.scene);
    }
    if (m->d_image && !m->ranks.empty() && m->ranks[0].device >= 0) {
        (void)hipSetDevice(m->ranks[0].device);
        (void)hipFree(m->d_image);
    }
    delete m;
}

// a failed image leaves work queued on the streams: wait for it, so that the context can be used (or destroyed) afterwards
void drain(rt_multi *m)
{
    for (Rank &k : m->ranks) {
        if (k.device < 0 || !k.stream) continue;
        (void)hipSetDevice(k.device);
        (void)hipStreamSynchronize(k.stream);
    }
}

} // namespace

extern "C" int rt_sample_range(int32_t rank, int32_t world, int32_t sample_begin, int32_t sample_end, int32_t *begin, int32_t *end)
{
    if (!begin || !end || world < 1 || rank < 0 || rank >= world || sample_begin < 0 || sample_end < sample_begin)
        return rt_fail(RT_ERR_INVALID, "rt_sample_range: rank %d of %d, samples [%d,%d)", rank, world, sample_begin, sample_end);
    const int64_t n = (int64_t)sample_end - sample_begin;
    *begin = sample_begin + (int32_t)(rank * n / world);
    *end = sample_begin + (int32_t)((rank + 1) * n / world);
    return RT_OK;
}

#define RTM_FAIL_CREATE(status, ...)      \
    do {                                  \
        const int st_ = rt_fail(status, __VA_ARGS__); \
        release(m);                       \
        return st_;                       \
    } while (0)

extern "C" int rt_multi_create(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags, rt_multi **out)
{
    if (!out) return rt_fail(RT_ERR_INVALID, "rt_multi_create: null output");
    *out = nullptr;
    if (!desc || !devices) return rt_fail(RT_ERR_INVALID, "rt_multi_create: null argument");
    if (n_devices < 1 || n_devices > 64) return rt_fail(RT_ERR_INVALID, "rt_multi_create: %d devices", n_devices);
    if (flags & ~RT_MULTI_REDUCE_ALWAYS) return rt_fail(RT_ERR_INVALID, "rt_multi_create: unknown flags 0x%x", flags);
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return rt_fail(RT_ERR_INVALID, "rt_multi_create: device %d is listed twice", devices[i]);
    const bool reduce = n_devices > 1 || (flags & RT_MULTI_REDUCE_ALWAYS);
    const Rccl &R = rccl();
    if (reduce && R.problem) return rt_fail(RT_ERR_UNSUPPORTED, "rt_multi_create: %s", R.problem);

    rt_multi *m = new (std::nothrow) rt_multi;
    if (!m) return rt_fail(RT_ERR_NOMEM, "rt_multi_create: out of host memory");
    m->reduce = reduce;
    m->ranks.resize(n_devices);
    // one scene and one stream per device (the scene tables are replicated: 2.6 KiB for the Cornell box)
    for (int i = 0; i < n_devices; ++i) {
        Rank &k = m->ranks[i];
        const int st = rt_scene_create(desc, devices[i], &k.scene);
        if (st != RT_OK) { // rt_last_error() holds the callee's message
            release(m);
            return st;
        }
        k.device = devices[i];
        hipError_t e = hipSetDevice(k.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking);
        if (e != hipSuccess) RTM_FAIL_CREATE(RT_ERR_NO_DEVICE, "rt_multi_create: device %d: %s", k.device, hipGetErrorString(e));
    }
    if (reduce) {
        std::vector<ncclComm_t> comms(n_devices);
        std::vector<int> devs(devices, devices + n_devices);
        const ncclResult_t n = R.CommInitAll(comms.data(), n_devices, devs.data());
        if (n != ncclSuccess) RTM_FAIL_CREATE(RT_ERR_NO_DEVICE, "rt_multi_create: ncclCommInitAll: %s", R.GetErrorString(n));
        for (int i = 0; i < n_devices; ++i) m->ranks[i].comm = comms[i];
    }
    *out = m;
    return RT_OK;
}

extern "C" void rt_multi_destroy(rt_multi *m) { release(m); }

#define RTM_HIP(call)                                                                                 \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            drain(m);                                                                                 \
            return rt_fail(RT_ERR_NO_DEVICE, "rt_multi_render_to_host: %s: %s", #call, hipGetErrorString(e_)); \
        }                                                                                             \
    } while (0)
#define RTM_NCCL(call)                                                                                \
    do {                                                                                              \
        ncclResult_t n_ = (call);                                                                     \
        if (n_ != ncclSuccess) {                                                                      \
            drain(m);                                                                                 \
            return rt_fail(RT_ERR_NO_DEVICE, "rt_multi_render_to_host: %s: %s", #call, R.GetErrorString(n_)); \
        }                                                                                             \
    } while (0)
#define RTM_RT(call)                                                                                  \
    do {                                                                                              \
        int s_ = (call);                                                                              \
        if (s_ != RT_OK) {                                                                            \
            drain(m);                                                                                 \
            return s_; /* rt_last_error() holds the callee's message */                               \
        }                                                                                             \
    } while (0)

extern "C" int rt_multi_render_to_host(rt_multi *m, const rt_render_params *p, float *h_accum, uint32_t *h_rgba8)
{
    if (!m || !p || (!h_accum && !h_rgba8)) return rt_fail(RT_ERR_INVALID, "rt_multi_render_to_host: null argument");
    if (p->struct_size != sizeof(rt_render_params)) return rt_fail(RT_ERR_INVALID, "rt_multi_render_to_host: bad rt_render_params");
    if (p->nx < 1 || p->ny < 1 || p->nx > 65536 || p->ny > 65536) return rt_fail(RT_ERR_INVALID, "rt_multi_render_to_host: image size %dx%d", p->nx, p->ny);
    std::lock_guard<std::mutex> lock(m->mutex);
    const Rccl &R = rccl();
    const int n_devices = (int)m->ranks.size();
    const size_t npix = (size_t)p->nx * p->ny, bytes = npix * 16;
    // the accumulators and the 8-bit image grow to the largest image seen and are kept
    if (npix > m->cap_pixels) {
        m->cap_pixels = 0; // until every buffer below exists: a failure half-way must not leave the old capacity standing
        for (Rank &k : m->ranks) {
            RTM_HIP(hipSetDevice(k.device));
            if (k.d_accum) (void)hipFree(k.d_accum);
            k.d_accum = nullptr;
            RTM_HIP(hipMalloc(&k.d_accum, bytes));
        }
        RTM_HIP(hipSetDevice(m->ranks[0].device));
        if (m->d_image) (void)hipFree(m->d_image);
        m->d_image = nullptr;
        RTM_HIP(hipMalloc(&m->d_image, npix * 4));
        m->cap_pixels = npix;
    }
    // every device renders its range; the launches return at once, so the devices run side by side
    for (int i = 0; i < n_devices; ++i) {
        rt_render_params q = *p;
        q.flags &= ~RT_RENDER_ACCUMULATE;
        RTM_RT(rt_sample_range(i, n_devices, p->sample_begin, p->sample_end, &q.sample_begin, &q.sample_end));
        RTM_RT(rt_render(m->ranks[i].scene, &q, m->ranks[i].d_accum, m->ranks[i].stream));
    }
    // one sum-reduce onto the first device, in stream order behind each device's render
    if (m->reduce) {
        RTM_NCCL(R.GroupStart());
        for (int i = 0; i < n_devices; ++i) {
            Rank &k = m->ranks[i];
            (void)hipSetDevice(k.device);
            const ncclResult_t n_ = R.Reduce(k.d_accum, k.d_accum, 4 * npix, ncclFloat, ncclSum, 0, k.comm, k.stream);
            if (n_ != ncclSuccess) {
                (void)R.GroupEnd();
                drain(m);
                return rt_fail(RT_ERR_NO_DEVICE, "rt_multi_render_to_host: ncclReduce: %s", R.GetErrorString(n_));
            }
        }
        RTM_NCCL(R.GroupEnd());
    }
    // the first device resolves (src/main.cpp:708-743) and hands the image(s) to the host
    Rank &first = m->ranks[0];
    RTM_HIP(hipSetDevice(first.device));
    if (h_rgba8) RTM_RT(rt_resolve(first.d_accum, p->nx, p->ny, m->d_image, first.stream));
    for (int i = n_devices - 1; i >= 0; --i) { // the first device last: its stream holds the reduce's result and the resolve
        RTM_HIP(hipSetDevice(m->ranks[i].device));
        RTM_HIP(hipStreamSynchronize(m->ranks[i].stream));
    }
    if (h_accum) RTM_HIP(hipMemcpy(h_accum, first.d_accum, bytes, hipMemcpyDeviceToHost));
    if (h_rgba8) RTM_HIP(hipMemcpy(h_rgba8, m->d_image, npix * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

extern "C" int rt_multi_last_render_ms(rt_multi *m, float *ms_max)
{
    if (!m || !ms_max) return rt_fail(RT_ERR_INVALID, "rt_multi_last_render_ms: null argument");
    std::lock_guard<std::mutex> lock(m->mutex);
    float worst = 0.0f;
    for (Rank &k : m->ranks) {
        float ms = 0.0f;
        const int st = rt_last_render_ms(k.scene, &ms);
        if (st != RT_OK) return st;
        worst = ms > worst ? ms : worst;
    }
    *ms_max = worst;
    return RT_OK;
}

extern "C" int rt_multi_set_option(rt_multi *m, int option, int value)
{
    if (!m) return rt_fail(RT_ERR_INVALID, "rt_multi_set_option: null context");
    std::lock_guard<std::mutex> lock(m->mutex);
    for (Rank &k : m->ranks) {
        const int st = rt_scene_set_option(k.scene, option, value);
        if (st != RT_OK) return st;
    }
    return RT_OK;
}

// the one-shot form: a context for one image
extern "C" int rt_render_multi_to_host(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, const rt_render_params *p,
                                       uint32_t flags, float *h_accum, uint32_t *h_rgba8)
{
    if (!desc || !devices || !p || !h_accum) return rt_fail(RT_ERR_INVALID, "rt_render_multi_to_host: null argument");
    rt_multi *m = nullptr;
    int st = rt_multi_create(desc, devices, n_devices, flags, &m);
    if (st != RT_OK) return st;
    st = rt_multi_render_to_host(m, p, h_accum, h_rgba8);
    rt_multi_destroy(m);
    return st;
}
