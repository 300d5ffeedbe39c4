// Device / stream / memory plumbing of the C ABI (include/flux3d_hip.h, first block).
// Replaces what the reference gets from CUDA.jl + Flux's gpu/cpu functor walkers
// (src/Flux3D.jl:52-61, src/rep/pcloud.jl:57, src/rep/mesh.jl:189-190).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "fx3d_common.h"

namespace fx3d {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- variant switches ---------------------------------------------------------------------------
namespace {
struct OptDef { const char *name, *env; int dflt; bool valued; };  // valued: an integer, not an on/off switch
// (order = enum Opt)
const OptDef kOptDefs[OPT_COUNT] = {
    {"nn1_variant", "FX3D_NN1_VARIANT", 3, true},
    {"nn1_nosplit", "FX3D_NN1_NOSPLIT", 0, false},
    {"bwd_global_atomics", "FX3D_BWD_GLOBAL_ATOMICS", 0, false},
    {"knn_no_mfma", "FX3D_KNN_NO_MFMA", 0, false},
    {"knn_no_prepass", "FX3D_KNN_NO_PREPASS", 0, false},
    {"knn_slices", "FX3D_KNN_SLICES", 0, true},
    {"edgeconv_unfused", "FX3D_EDGECONV_UNFUSED", 0, false},
    {"lap_bwd_scatter", "FX3D_LAP_BWD_SCATTER", 0, false},
    {"cdf_multiblock_from", "FX3D_CDF_MULTIBLOCK_FROM", 0, true},
    {"nn1_tiny_mpairs", "FX3D_NN1_TINY_MPAIRS", 24, true},
    {"mesh_max_blocks", "FX3D_MESH_MAX_BLOCKS", 0, true},
    {"nn1_prune", "FX3D_NN1_PRUNE", 1, true},
};
std::atomic<int> g_opt[OPT_COUNT];
std::once_flag g_opt_once;
void opt_init() {
    std::call_once(g_opt_once, [] {
        for (int i = 0; i < OPT_COUNT; ++i) {
            const char *e = getenv(kOptDefs[i].env);  // the environment seeds the defaults, once (read-once: INTEGRATION.md)
            int v = kOptDefs[i].dflt;
            if (e && *e) {  // an EMPTY value is "unset" (scripts neutralise a switch with FX3D_X=): the default stands
                char *end = nullptr;
                const long n = strtol(e, &end, 10);
                while (end && (*end == ' ' || *end == '\t')) ++end;
                const bool numeric = end != e && !(end && *end);
                if (numeric) v = (int)n;
                else if (!kOptDefs[i].valued) v = 1;  // on/off switch set to a word (FX3D_KNN_NO_MFMA=yes): on
                else fprintf(stderr, "flux3d_hip: %s=\"%s\" is not an integer; keeping the default %d\n", kOptDefs[i].env, e, v);
            }
            g_opt[i].store(v, std::memory_order_relaxed);
        }
    });
}
int opt_index(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, kOptDefs[i].name) || !strcmp(name, kOptDefs[i].env)) return i;
    return -1;
}
}  // namespace

int device_cus() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

int opt(Opt o) {
    opt_init();
    return g_opt[o].load(std::memory_order_relaxed);
}

// ---- per-kernel event timing -------------------------------------------------------------------
namespace {
struct ProfRec { const char *name; hipEvent_t e0, e1; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::atomic<int> g_prof_on{0};     // 0 = off, n = bracket every n-th launch of a kernel
std::atomic<unsigned> g_prof_seq{0};
constexpr size_t kProfCap = 8192;
}  // namespace

bool profile_on() {
    const int n = g_prof_on.load(std::memory_order_relaxed);
    if (n <= 0) return false;
    return n == 1 || (g_prof_seq.fetch_add(1, std::memory_order_relaxed) % (unsigned)n) == 0;
}

void profile_mark(const char *name, hipStream_t st, bool begin) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (begin) {
        if (g_prof.size() >= kProfCap) return;
        ProfRec r{name, nullptr, nullptr};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        g_prof.push_back(r);
    } else {
        for (size_t i = g_prof.size(); i-- > 0;)
            if (g_prof[i].name == name) { (void)hipEventRecord(g_prof[i].e1, st); break; }
    }
}

static void profile_clear() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &r : g_prof) {
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
}

}  // namespace fx3d

using namespace fx3d;

extern "C" {

const char *fx3d_version(void) { return "flux3d_hip 0.1.0 (gfx950)"; }

fx3d_status fx3d_set_option(const char *name, int32_t value) {
    const int i = fx3d::opt_index(name);
    if (i < 0) { fx3d::set_error("fx3d_set_option: unknown option %s", name ? name : "(null)"); return FX3D_ERR_INVALID_ARG; }
    fx3d::opt_init();
    fx3d::g_opt[i].store(value, std::memory_order_relaxed);
    return FX3D_OK;
}

fx3d_status fx3d_get_option(const char *name, int32_t *value) {
    const int i = fx3d::opt_index(name);
    if (i < 0 || !value) { fx3d::set_error("fx3d_get_option: unknown option %s", name ? name : "(null)"); return FX3D_ERR_INVALID_ARG; }
    *value = fx3d::opt(static_cast<fx3d::Opt>(i));
    return FX3D_OK;
}

int32_t fx3d_option_count(void) { return fx3d::OPT_COUNT; }

const char *fx3d_option_name(int32_t index) {
    return index >= 0 && index < fx3d::OPT_COUNT ? fx3d::kOptDefs[index].name : nullptr;
}

size_t fx3d_last_error(char *buf, size_t n) {
    size_t len = strlen(g_err);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return len;
}

fx3d_status fx3d_profile_enable(int32_t on) {
    profile_clear();
    g_prof_seq.store(0);
    g_prof_on.store(on > 0 ? on : 0);
    return FX3D_OK;
}

fx3d_status fx3d_profile_kernel_stats(const char *name, double *avg_ms, double *min_ms,
                                      double *max_ms, int64_t *count) {
    FX3D_REQUIRE(name && count, "fx3d_profile_kernel_stats: null argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double sum = 0.0, mn = 1e30, mx = 0.0;
    int64_t n = 0;
    for (auto &r : g_prof) {
        if (strcmp(r.name, name) != 0) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        sum += ms; mn = ms < mn ? ms : mn; mx = ms > mx ? ms : mx; ++n;
    }
    *count = n;
    if (avg_ms) *avg_ms = n ? sum / (double)n : 0.0;
    if (min_ms) *min_ms = n ? mn : 0.0;
    if (max_ms) *max_ms = n ? mx : 0.0;
    return FX3D_OK;
}

fx3d_status fx3d_device_count(int32_t *n) {
    FX3D_REQUIRE(n, "fx3d_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return hip_fail(e, "hipGetDeviceCount", __FILE__, __LINE__);
    }
    *n = c;
    return FX3D_OK;
}

fx3d_status fx3d_set_device(int32_t dev) {
    FX3D_HIP(hipSetDevice(dev));
    return FX3D_OK;
}

fx3d_status fx3d_get_device(int32_t *dev) {
    FX3D_REQUIRE(dev, "fx3d_get_device: null output");
    int d = 0;
    FX3D_HIP(hipGetDevice(&d));
    *dev = d;
    return FX3D_OK;
}

fx3d_status fx3d_device_name(int32_t dev, char *buf, size_t n) {
    FX3D_REQUIRE(buf && n, "fx3d_device_name: null buffer");
    hipDeviceProp_t p;
    FX3D_HIP(hipGetDeviceProperties(&p, dev));
    snprintf(buf, n, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return FX3D_OK;
}

fx3d_status fx3d_device_identity(int32_t dev, char *pci_bus_id, size_t n, uint8_t *uuid16) {
    FX3D_REQUIRE(pci_bus_id && n >= 16 && uuid16, "fx3d_device_identity: null / short buffer");
    FX3D_HIP(hipDeviceGetPCIBusId(pci_bus_id, (int)n, dev));
    hipDeviceProp_t p;
    FX3D_HIP(hipGetDeviceProperties(&p, dev));
    memcpy(uuid16, p.uuid.bytes, 16);
    return FX3D_OK;
}

fx3d_status fx3d_device_sync(void) {
    FX3D_HIP(hipDeviceSynchronize());
    return FX3D_OK;
}

fx3d_status fx3d_malloc(void **p, size_t bytes) {
    FX3D_REQUIRE(p, "fx3d_malloc: null output");
    *p = nullptr;
    if (bytes == 0) return FX3D_OK;
    FX3D_HIP(hipMalloc(p, bytes));
    return FX3D_OK;
}

fx3d_status fx3d_free(void *p) {
    if (p) FX3D_HIP(hipFree(p));
    return FX3D_OK;
}

fx3d_status fx3d_memcpy_h2d(void *dst, const void *src, size_t bytes, fx3d_stream_t s) {
    if (bytes == 0) return FX3D_OK;
    FX3D_REQUIRE(dst && src, "fx3d_memcpy_h2d: null pointer");
    FX3D_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(s)));
    FX3D_HIP(hipStreamSynchronize(as_stream(s)));  // pageable host memory: keep it blocking
    return FX3D_OK;
}

fx3d_status fx3d_memcpy_d2h(void *dst, const void *src, size_t bytes, fx3d_stream_t s) {
    if (bytes == 0) return FX3D_OK;
    FX3D_REQUIRE(dst && src, "fx3d_memcpy_d2h: null pointer");
    FX3D_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(s)));
    FX3D_HIP(hipStreamSynchronize(as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_memcpy_d2d(void *dst, const void *src, size_t bytes, fx3d_stream_t s) {
    if (bytes == 0) return FX3D_OK;
    FX3D_REQUIRE(dst && src, "fx3d_memcpy_d2d: null pointer");
    FX3D_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_memset(void *dst, int32_t byte, size_t bytes, fx3d_stream_t s) {
    if (bytes == 0) return FX3D_OK;
    FX3D_REQUIRE(dst, "fx3d_memset: null pointer");
    FX3D_HIP(hipMemsetAsync(dst, byte, bytes, as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_stream_create(fx3d_stream_t *s) {
    FX3D_REQUIRE(s, "fx3d_stream_create: null output");
    hipStream_t st;
    FX3D_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *s = st;
    return FX3D_OK;
}

fx3d_status fx3d_stream_destroy(fx3d_stream_t s) {
    if (s) FX3D_HIP(hipStreamDestroy(as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_stream_sync(fx3d_stream_t s) {
    FX3D_HIP(hipStreamSynchronize(as_stream(s)));
    return FX3D_OK;
}

// ---- stream capture: a launch-bound loop body (the fit_mesh iteration: ~30 small kernels, memsets and copies)
//      is recorded once and replayed as one hipGraph launch ---------------------------------------------------------
fx3d_status fx3d_graph_begin_capture(fx3d_stream_t s) {
    FX3D_REQUIRE(s, "fx3d_graph_begin_capture: capture needs a created stream (not the default stream)");
    FX3D_HIP(hipStreamBeginCapture(as_stream(s), hipStreamCaptureModeRelaxed));
    return FX3D_OK;
}

fx3d_status fx3d_graph_end_capture(fx3d_stream_t s, fx3d_graph_t *g) {
    FX3D_REQUIRE(s && g, "fx3d_graph_end_capture: bad argument");
    hipGraph_t graph = nullptr;
    FX3D_HIP(hipStreamEndCapture(as_stream(s), &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        set_error("fx3d_graph_end_capture: hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return FX3D_ERR_HIP;
    }
    *g = exec;
    return FX3D_OK;
}

fx3d_status fx3d_graph_launch(fx3d_graph_t g, fx3d_stream_t s) {
    FX3D_REQUIRE(g, "fx3d_graph_launch: null graph");
    FX3D_HIP(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(g), as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_graph_destroy(fx3d_graph_t g) {
    if (g) FX3D_HIP(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(g)));
    return FX3D_OK;
}

namespace {
__global__ void counter_add_kernel(unsigned long long *ctr, unsigned long long inc) { *ctr += inc; }
}  // namespace

fx3d_status fx3d_counter_add(uint64_t *ctr, uint64_t inc, fx3d_stream_t s) {
    FX3D_REQUIRE(ctr, "fx3d_counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(s), reinterpret_cast<unsigned long long *>(ctr),
                       (unsigned long long)inc);
    FX3D_LAUNCH_CHECK();
    return FX3D_OK;
}

fx3d_status fx3d_event_create(fx3d_event_t *e) {
    FX3D_REQUIRE(e, "fx3d_event_create: null output");
    hipEvent_t ev;
    FX3D_HIP(hipEventCreate(&ev));
    *e = ev;
    return FX3D_OK;
}

// An event that only ORDERS work between streams of one device (fx3d_stream_wait_event) or tells the host that device work
// is done: no timestamps, no system-scope fence at the record (kernel boundaries keep their device-scope release/acquire).
// Measured on the overlapped sharded evaluation (two such events per step): 62.5 -> 59.9 us per step.
fx3d_status fx3d_event_create_sync(fx3d_event_t *e) {
    FX3D_REQUIRE(e, "fx3d_event_create_sync: null output");
    hipEvent_t ev;
    FX3D_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    *e = ev;
    return FX3D_OK;
}

fx3d_status fx3d_event_destroy(fx3d_event_t e) {
    if (e) FX3D_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(e)));
    return FX3D_OK;
}

fx3d_status fx3d_event_record(fx3d_event_t e, fx3d_stream_t s) {
    FX3D_REQUIRE(e, "fx3d_event_record: null event");
    FX3D_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(e), as_stream(s)));
    return FX3D_OK;
}

fx3d_status fx3d_event_sync(fx3d_event_t e) {
    FX3D_REQUIRE(e, "fx3d_event_sync: null event");
    FX3D_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(e)));
    return FX3D_OK;
}

fx3d_status fx3d_stream_wait_event(fx3d_stream_t s, fx3d_event_t e) {
    FX3D_REQUIRE(e, "fx3d_stream_wait_event: null event");
    FX3D_HIP(hipStreamWaitEvent(as_stream(s), reinterpret_cast<hipEvent_t>(e), 0));
    return FX3D_OK;
}

fx3d_status fx3d_event_elapsed_ms(fx3d_event_t a, fx3d_event_t b, float *ms) {
    FX3D_REQUIRE(a && b && ms, "fx3d_event_elapsed_ms: null argument");
    FX3D_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(a), reinterpret_cast<hipEvent_t>(b)));
    return FX3D_OK;
}

}  // extern "C"

namespace fx3d {
fx3d_status ensure_dynamic_lds(const void *kernel, int bytes, const char *name) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    FX3D_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return FX3D_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%s, %d bytes of dynamic LDS) failed: %s", name, bytes, hipGetErrorString(e));
        return FX3D_ERR_HIP;
    }
    done.insert({kernel, dev});
    return FX3D_OK;
}
}  // namespace fx3d
