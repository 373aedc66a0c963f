// libpt_hip.so — C ABI over the wavefront kernels (see include/pt_hip.h).
// Host side: validation and device repack of the uploaded scene, queue allocation, the stage
// schedule of one render call (all launches enqueued on one stream, no host round trip per
// bounce), statistics.
#include "pt_hip.h"
#include "pt_kernels.hpp"
#include "pt_build.hpp"
#include "pt_repack.hpp"
#include "pt_instances.hpp"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <functional>
#include <memory>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct HostNode { float mn[3], mx[3]; int32_t leftFirst, count; };
struct HostVertex { float pos[3], pad0, nrm[3], pad1; };
struct HostTriangle { HostVertex v[3]; int32_t material; int32_t pad[3]; };
struct HostMaterial { float color[4], emissive[4]; int32_t type; float index, roughness; int32_t pad; };
struct HostRenderState { int32_t width, height, pad[2]; float sky[4]; };
struct HostCamera { float p1[3], _0, p2[3], _1, p3[3], _2, pos[3], _3, right[3], _4, up[3], _5; };
struct HostFrameState { int32_t frames, samples; float focal; int32_t pad; HostCamera cam; };
static_assert(sizeof(HostNode) == 32 && sizeof(HostTriangle) == 112 && sizeof(HostMaterial) == 48, "host layouts");
static_assert(sizeof(HostRenderState) == 32 && sizeof(HostFrameState) == 112, "host layouts");

std::string g_err;   // errors without a context

}  // namespace

// ---- fixed device staging of pt_trace_closest_noalloc -----------------------------------------
// Rays in and hits out of one call, per ray: origin + direction (24 B), triangle (4 B), distance (4 B), (u, v) (8 B).
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process ends.
#define PT_NOALLOC_BYTES ((size_t)PT_NOALLOC_MAX_RAYS * 40)
static __device__ __align__(256) unsigned char scratch[PT_NOALLOC_BYTES];
namespace {
// `scratch` is only a host-side handle for hipGetSymbolAddress; the device address of each GPU's copy is kept here
// (resolved once per device by pt_create, with that device current) and is what kernels and copies receive.
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

hipError_t resolveScratch(int device, void** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct Group;

struct pt_ctx {
    pt_config cfg;
    Group* group = nullptr;       // device_count > 1: this context only fans calls out to one member context per GPU
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t stream_connect = nullptr;   // the connect stage of a bounce runs here, beside extend of the next bounce
    hipEvent_t ev_shaded = nullptr, ev_connected = nullptr;
    // `stream` is the context's ordering point: accumulation, read-outs, queries, whatever a caller queues after pt_stream.  A render
    // call traces on a stream of its own (this one for the context's first set of queues) and hands its planes over through two events.
    hipStream_t stream_render = nullptr;
    hipEvent_t ev_done0 = nullptr, ev_resolved0 = nullptr;
    int num_cu = 256;
    // scene
    DevScene sc{};
    void *d_pairs = nullptr, *d_quads = nullptr, *d_big = nullptr, *d_hot = nullptr, *d_cold = nullptr, *d_mats = nullptr, *d_lights = nullptr;
    void *d_qquads = nullptr, *d_gate = nullptr;   // the quantised quad records and the triangles with their leaves' boxes
    bool has_scene = false;
    // image / queues
    int width = 0, height = 0;
    uint32_t groups = 0, local_groups = 0, slots = 0, slots_padded = 0;
    uint32_t capacity = 0, batch = 0;   // queue capacity in paths, samples per batch
    Queues q{};
    std::vector<void*> qmem;
    float4* d_sum = nullptr;
    float4* d_image = nullptr;
    Ctl* d_ctl = nullptr;
    HostRenderState rs{};
    bool has_rs = false;
    uint32_t samples_accumulated = 0;
    // timing
    struct Span { hipEvent_t a, b; int stage; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool render_pending = false;
    pt_stats host_stats{};
    // idle lanes a traversal wave collects before it fetches new rays: camera rays of a tile finish together, so waiting for
    // most of the wave costs little and saves refills; scattered rays finish at very different times
    int refill_primary = 48, refill_bounce = 40;
    // A tree that does not fit L2 (1M small triangles and up) means camera rays that part ways inside it: such scenes gain 3-4 % from
    // waves that fetch camera rays earlier (at 32 idle lanes) and from camera rays queued group by group; scenes whose tree stays in
    // L2 lose 1-4 % with either.  An explicit PT_OPT_REFILL_PRIMARY wins.
    bool large_tree = false, refill_primary_set = false, refill_bounce_set = false;
    // Trees beyond L2 are walked through the 64-byte quantised records (half the bytes per step, the exact leaf boxes applied at the
    // triangles); trees that stay in L2 are bound by instruction issue and keep the exact 128-byte records.  quant_mode: 0 never,
    // 1 whenever the records exist, 2 for large trees (PT_OPT_QUANTIZED).
    bool quant_ok = false;
    // the traversal addresses an exact quad record with a 32-bit byte offset: beyond 4 GB of them (about 100M triangles) only the
    // quantised records are walked, whatever PT_OPT_QUANTIZED says
    bool exact_ok = true;
    int quant_mode = 2;
    // The scene as instances of meshes (pt_build_scene_instanced): walked only when the host asks (PT_OPT_INSTANCED), because hits
    // the flattened walk drops for missing their leaf's box are found by this one.
    bool inst_ok = false;
    int inst_mode = 0, fit_inst = 4;
    void *d_iquads = nullptr, *d_ibig = nullptr, *d_itris = nullptr, *d_iinst = nullptr, *d_built_of = nullptr;
    uint64_t local_pixels = 0;
    int grid_extend = 0, grid_connect = 0, grid_shade = 0, grid_shade_small = 0, fit_extend = 1, fit_connect = 1;
    bool traverse_blocks_set = false;   // PT_OPT_TRAVERSE_BLOCKS_PER_CU was given: it then holds for every call
    void* scratch_dev = nullptr;   // this device's copy of the fixed staging array (pt_trace_closest_noalloc)
    // A second set of streams, events, control block and queues.  Consecutive render calls of moderate size alternate between the
    // context's first set and this one: every bounce of a frame ends in a tail of a few long rays during which most of the chip idles,
    // and two frames in flight fill each other's tails (5 % on back-to-back 1080p x 16 spp frames, more on small ones).  Only the
    // traversal and shading of a call run on a set's streams; its planes are added to the accumulator on the context's stream, in call
    // order, so everything queued on that stream — read-outs, resets, a caller's own work after pt_stream — is ordered after every
    // render call made before it.
    struct Lane {
        hipStream_t stream = nullptr;
        hipEvent_t ev_done = nullptr, ev_resolved = nullptr;
        Ctl* d_ctl = nullptr;
        Queues q{};
        std::vector<void*> qmem;
        uint32_t capacity = 0;
    } lane1;
    bool lane1_next = false;   // which set the next eligible render call takes
    int pipeline_calls = 1;    // PT_OPT_PIPELINE_CALLS
};

// What a render call runs on.  `hop`: the traversal stream is not the context's stream, so the batch's end is handed over by ev_done;
// ev_resolved (recorded on the context's stream after the planes were added) is what the set's next batch waits for.
struct LaneRef {
    hipStream_t stream, stream_connect;
    hipEvent_t ev_shaded, ev_connected, ev_done, ev_resolved;
    Ctl* d_ctl;
    Queues* q;
    std::vector<void*>* qmem;
    uint32_t* capacity;
    bool hop;
    // the call shares the GPU with another frame's kernels: its traversal grids leave room on every CU and its shade stage runs
    // in small blocks that fit into that room (renderBatches)
    bool shared = false;
};
// the first set on its own traversal stream (calls that may overlap with their neighbours)
static LaneRef ownLane(pt_ctx* c) {
    return LaneRef{c->stream_render, nullptr, nullptr, nullptr, c->ev_done0, c->ev_resolved0, c->d_ctl, &c->q, &c->qmem, &c->capacity, true};
}
// the first set on the context's stream itself: after everything queued before, before everything queued after (per-stage timing,
// very large calls, the debugging entry points)
static LaneRef inlineLane(pt_ctx* c) {
    return LaneRef{c->stream, c->stream_connect, c->ev_shaded, c->ev_connected, nullptr, c->ev_resolved0, c->d_ctl, &c->q, &c->qmem, &c->capacity, false};
}
static LaneRef secondLane(pt_ctx* c) {
    pt_ctx::Lane& l = c->lane1;
    return LaneRef{l.stream, nullptr, nullptr, nullptr, l.ev_done, l.ev_resolved, l.d_ctl, &l.q, &l.qmem, &l.capacity, true};
}

#define FAIL(ctx, code, ...) do { char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__); if (ctx) (ctx)->err = _b; else g_err = _b; return (code); } while (0)
#define HIPCHECK(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { \
    FAIL(ctx, (_e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } } while (0)

namespace {

void freeQueues(pt_ctx* c) {
    for (void* p : c->qmem) (void)hipFree(p);
    c->qmem.clear();
    c->q = Queues{};
    for (void* p : c->lane1.qmem) (void)hipFree(p);
    c->lane1.qmem.clear();
    c->lane1.q = Queues{};
    c->lane1.capacity = 0;
    if (c->d_sum) { (void)hipFree(c->d_sum); c->d_sum = nullptr; }
    if (c->d_image) { (void)hipFree(c->d_image); c->d_image = nullptr; }
    c->capacity = 0;
}

void freeScene(pt_ctx* c) {
    void** ps[] = {&c->d_pairs, &c->d_quads, &c->d_big, &c->d_hot, &c->d_cold, &c->d_mats, &c->d_lights, &c->d_qquads, &c->d_gate,
                   &c->d_iquads, &c->d_ibig, &c->d_itris, &c->d_iinst, &c->d_built_of};
    for (void** p : ps) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
    c->has_scene = false;
    c->quant_ok = false;
    c->inst_ok = false;
    c->sc = DevScene{};
}

#define PT_BYTES_PER_PATH 176u   // eleven 16-byte queue columns (allocQueues)
#ifndef PT_PIPELINE_MAX_PATHS
#define PT_PIPELINE_MAX_PATHS (64ull << 20)   // batches of more paths than this are traced on the context's stream, one at a time
#endif
int allocQueues(pt_ctx* c, const LaneRef& L, uint32_t capacity) {
    std::vector<void*>& qmem = *L.qmem;
    Queues& q = *L.q;
    for (void* p : qmem) (void)hipFree(p);
    qmem.clear();
    // nothing may keep pointing at the freed columns if an allocation below fails
    q = Queues{};
    *L.capacity = 0;
    float4** cols[] = {&q.o[0], &q.o[1], &q.d[0], &q.d[1], &q.T[0], &q.T[1], &q.hit, &q.so, &q.sd, &q.sc, &q.plane};
    for (float4** col : cols) {
        void* p = nullptr;
        const hipError_t e = ptMalloc(&p, (size_t)capacity * sizeof(float4));
        if (e != hipSuccess) {
            for (void* m : qmem) (void)hipFree(m);
            qmem.clear();
            q = Queues{};
            FAIL(c, (e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "ray queues of %u paths: %s", capacity, hipGetErrorString(e));
        }
        qmem.push_back(p);
        *col = (float4*)p;
    }
    *L.capacity = capacity;
    return PT_OK;
}

uint32_t chooseBatch(const pt_ctx* c, uint32_t spp) {
    const uint32_t slots = std::max(1u, c->slots);
    spp = std::max(1u, spp);
    // queue indices are 32-bit: never more than 2^30 paths in flight
    const uint32_t cap = std::max(1u, (1u << 30) / slots);
    if (c->cfg.samples_per_batch != 0) return std::max(1u, std::min(std::min((uint32_t)c->cfg.samples_per_batch, cap), spp));   // an explicit size is taken literally
    // As many paths per launch as memory reasonably allows: every bounce of a batch ends in a tail of a few long rays and the late
    // bounces are small launches, so a frame traced as one batch of 265M paths is 12-17 % faster than the same frame in batches of
    // 33M (3840x2160 x 32 spp; the 20M-triangle scene at 8 spp; the 1M soup at 64 spp).  A path costs 176 bytes of queues; the
    // target is 256M paths (45 GB of the 288), less if a quarter of the device memory that is free would not hold it.
    uint64_t target = 256ull << 20;
    if (std::min<uint64_t>(target, (uint64_t)slots * spp) > c->capacity) {   // (queues this context holds already need no second look)
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
            const uint64_t have = (uint64_t)freeB + (uint64_t)c->capacity * PT_BYTES_PER_PATH;   // the present queues would be replaced
            target = std::min<uint64_t>(target, std::max<uint64_t>(c->capacity, have / 4 / PT_BYTES_PER_PATH));
        }
    }
    uint32_t want = (uint32_t)std::max<uint64_t>(1, target / slots);
    want = std::min(want, 1024u);   // tile-sharded ranks own few slots and batch many samples
    const uint32_t most = std::max(1u, std::min(std::min(want, cap), spp));
    // equal batches: 128 samples at 123 per batch would leave a second batch of 5 that cannot fill the chip
    const uint32_t batches = (spp + most - 1u) / most;
    return (spp + batches - 1u) / batches;
}

hipEvent_t getEvent(pt_ctx* c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct StageTimer {
    pt_ctx* c; int stage; hipStream_t on; hipEvent_t a = nullptr, b = nullptr;
    StageTimer(pt_ctx* ctx, int s, hipStream_t stream) : c(ctx), stage(s), on(stream) {
        if (c->cfg.stage_timing == 1 || (c->cfg.stage_timing == 2 && (s == 1 || s == 4))) { a = getEvent(c); b = getEvent(c); if (a) (void)hipEventRecord(a, on); }
    }
    ~StageTimer() {
        if (a && b) { (void)hipEventRecord(b, on); c->spans.push_back({a, b, stage}); }
    }
};

void collectSpans(pt_ctx* c) {
    for (auto& s : c->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            // stage 4 = the extend launch of bounce 0 (camera rays): part of ms_extend, and kept apart as well
            float* dst[] = {&c->host_stats.ms_generate, &c->host_stats.ms_extend, &c->host_stats.ms_shade, &c->host_stats.ms_connect, &c->host_stats.ms_extend};
            *dst[s.stage] += ms;
            if (s.stage == 4) c->host_stats.ms_extend_primary += ms;
        }
        c->pool.push_back(s.a);
        c->pool.push_back(s.b);
    }
    c->spans.clear();
}

int launchGrid(uint32_t n) { return (int)((n + PT_BLOCK - 1) / PT_BLOCK); }

}  // namespace

namespace {
// Nothing may unwind through the C boundary: host allocations of an entry point (std::vector, new) that fail become PT_E_OOM.
template <class F>
int guarded(pt_ctx* c, const char* what, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        char b[128];
        snprintf(b, sizeof(b), "%s: out of host memory", what);
        if (c) c->err = b; else g_err = b;
        return PT_E_OOM;
    } catch (const std::exception& e) {
        char b[256];
        snprintf(b, sizeof(b), "%s: %s", what, e.what());
        if (c) c->err = b; else g_err = b;
        return PT_E_HIP;
    }
}
}  // namespace

// ---- several GPUs behind one context ---------------------------------------------------------------------------
// A context created with device_count > 1 owns one ordinary single-GPU context per device ("member", tile_rank = its
// index): the scene is replicated, member i renders the pixel groups g with g % N == i, and pt_read_image gathers the
// members' accumulators to member 0 once per image.  Every member has a host thread of its own, so blocking work
// (uploads, builds, allocation) and the enqueueing of a frame's launches run on all GPUs at once.
namespace {

// RCCL, loaded when the first multi-GPU context is created: a single-GPU host never maps the library.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) commInitAll = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr;
    decltype(&ncclGroupEnd) groupEnd = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGetErrorString) errorString = nullptr;
    std::string problem;
    bool load() {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { problem = std::string("cannot load librccl: ") + dlerror(); return false; }
        commInitAll = (decltype(commInitAll))dlsym(handle, "ncclCommInitAll");
        commDestroy = (decltype(commDestroy))dlsym(handle, "ncclCommDestroy");
        groupStart = (decltype(groupStart))dlsym(handle, "ncclGroupStart");
        groupEnd = (decltype(groupEnd))dlsym(handle, "ncclGroupEnd");
        send = (decltype(send))dlsym(handle, "ncclSend");
        recv = (decltype(recv))dlsym(handle, "ncclRecv");
        errorString = (decltype(errorString))dlsym(handle, "ncclGetErrorString");
        if (!commInitAll || !commDestroy || !groupStart || !groupEnd || !send || !recv || !errorString) {
            problem = "librccl lacks one of ncclCommInitAll/ncclCommDestroy/ncclGroupStart/ncclGroupEnd/ncclSend/ncclRecv/ncclGetErrorString";
            handle = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// One host thread that runs the jobs posted to it, one at a time.
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<void()> job) {
        { std::lock_guard<std::mutex> l(m_); job_ = std::move(job); pending_ = true; done_ = false; }
        cv_.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return done_; });
    }
private:
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return pending_ || stop_; });
                if (!pending_ && stop_) return;
                job = std::move(job_);
                pending_ = false;
            }
            job();
            { std::lock_guard<std::mutex> l(m_); done_ = true; }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool pending_ = false, done_ = true, stop_ = false;
    std::thread th_;
};

}  // namespace

struct Group {
    std::vector<pt_ctx*> members;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<ncclComm_t> comms;        // one per member when the tiles travel by RCCL; empty: device-to-device copies
    std::vector<hipEvent_t> ready;        // per member: its accumulator has been copied out (copy path)
    float4* d_gather = nullptr;           // on member 0's GPU: N accumulators of slots_padded float4 each
    size_t gather_slots = 0;

    // Runs f(member) for every member on that member's thread and waits for all; the first failure is the result and
    // its message becomes the owner's.  Nothing thrown by f leaves its thread.
    template <class F>
    int all(pt_ctx* owner, F f) {
        std::vector<int> rc(members.size(), PT_OK);
        for (size_t i = 0; i < members.size(); i++) {
            pt_ctx* m = members[i];
            int* out = &rc[i];
            workers[i]->post([m, out, &f] { *out = guarded(m, "multi-GPU call", [&]() -> int { return f(m); }); });
        }
        for (auto& w : workers) w->wait();
        for (size_t i = 0; i < members.size(); i++)
            if (rc[i] != PT_OK) {
                char b[640];
                snprintf(b, sizeof(b), "device %d (member %zu of %zu): %s", members[i]->cfg.device, i, members.size(), members[i]->err.c_str());
                owner->err = b;
                return rc[i];
            }
        return PT_OK;
    }
};

extern "C" {

int pt_default_config(pt_config* cfg) {
    if (!cfg) return PT_E_INVALID_ARG;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(pt_config);
    cfg->tile_world = 1;
    cfg->device_count = 1;
    return PT_OK;
}

const char* pt_last_error(pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static void destroyGroup(pt_ctx* owner);
static int impl_create_single(const pt_config* cfg, pt_ctx** out);

// A context over several GPUs: one member context per device, a host thread each, and the communicators of the gather.
static int impl_create_group(const pt_config* cfg, pt_ctx** out) {
    const int n = cfg->device_count < 1 ? 1 : cfg->device_count;
    const bool sameDevice = (cfg->group_flags & PT_GROUP_SAME_DEVICE) != 0;
    if (cfg->tile_rank != 0 || cfg->tile_world != 1) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: device_count %d with tile rank %d of %d (a multi-GPU context shards by itself)", n, cfg->tile_rank, cfg->tile_world);
    if (n > 64) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: device_count %d", n);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) FAIL((pt_ctx*)nullptr, PT_E_NO_DEVICE, "pt_create: no HIP device");
    if (cfg->device < 0 || cfg->device + (sameDevice ? 1 : n) > ndev) FAIL((pt_ctx*)nullptr, PT_E_NO_DEVICE, "pt_create: devices %d..%d of %d", cfg->device, cfg->device + n - 1, ndev);
    pt_ctx* owner = new pt_ctx();
    owner->cfg = *cfg;
    owner->group = new Group();
    Group* g = owner->group;
    for (int i = 0; i < n; i++) {
        pt_config mc = *cfg;
        mc.device = sameDevice ? cfg->device : cfg->device + i;
        mc.device_count = 1;
        mc.group_flags = 0;
        mc.tile_rank = i;
        mc.tile_world = n;
        pt_ctx* m = nullptr;
        const int rc = impl_create_single(&mc, &m);
        if (rc != PT_OK) { destroyGroup(owner); return rc; }   // g_err holds the message
        g->members.push_back(m);
        g->workers.emplace_back(new Worker());
    }
    const bool rccl = (!sameDevice && n > 1) || (cfg->group_flags & PT_GROUP_FORCE_RCCL);
    if (rccl) {
        if (sameDevice && n > 1) { destroyGroup(owner); FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: RCCL needs one GPU per member; PT_GROUP_SAME_DEVICE and PT_GROUP_FORCE_RCCL only combine with device_count 1"); }
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!g_rccl.load()) { destroyGroup(owner); FAIL((pt_ctx*)nullptr, PT_E_RCCL, "pt_create: %s", g_rccl.problem.c_str()); }
        std::vector<int> devs;
        for (pt_ctx* m : g->members) devs.push_back(m->cfg.device);
        g->comms.assign(n, nullptr);
        const ncclResult_t r = g_rccl.commInitAll(g->comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            g->comms.clear();
            destroyGroup(owner);
            FAIL((pt_ctx*)nullptr, PT_E_RCCL, "pt_create: ncclCommInitAll over %d GPUs: %s", n, g_rccl.errorString(r));
        }
    } else {
        for (pt_ctx* m : g->members) {
            hipEvent_t e = nullptr;
            if (hipSetDevice(m->cfg.device) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                destroyGroup(owner);
                FAIL((pt_ctx*)nullptr, PT_E_HIP, "pt_create: no event for member on device %d", m->cfg.device);
            }
            g->ready.push_back(e);
        }
    }
    *out = owner;
    return PT_OK;
}

static void destroyGroup(pt_ctx* owner) {
    Group* g = owner->group;
    if (g) {
        for (pt_ctx* m : g->members) { (void)hipSetDevice(m->cfg.device); if (m->stream) (void)hipStreamSynchronize(m->stream); }
        for (ncclComm_t cm : g->comms) if (cm) (void)g_rccl.commDestroy(cm);
        for (hipEvent_t e : g->ready) (void)hipEventDestroy(e);
        if (g->d_gather && !g->members.empty()) { (void)hipSetDevice(g->members[0]->cfg.device); (void)hipFree(g->d_gather); }
        g->workers.clear();   // joins the threads
        for (pt_ctx* m : g->members) pt_destroy(m);
        delete g;
    }
    delete owner;
}

static int impl_create(const pt_config* cfg, pt_ctx** out) {
    if (!cfg || !out) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(pt_config)) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: struct_size %d, expected %zu", cfg->struct_size, sizeof(pt_config));
    if (cfg->device_count < 0) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: device_count %d", cfg->device_count);
    if (cfg->device_count > 1 || (cfg->group_flags & PT_GROUP_FORCE_RCCL)) return impl_create_group(cfg, out);
    return impl_create_single(cfg, out);
}

static int impl_create_single(const pt_config* cfg, pt_ctx** out) {
    if (cfg->tile_world < 1 || cfg->tile_rank < 0 || cfg->tile_rank >= cfg->tile_world) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: tile rank %d of %d", cfg->tile_rank, cfg->tile_world);
    if (cfg->nee < 0 || cfg->nee > 2) FAIL((pt_ctx*)nullptr, PT_E_INVALID_ARG, "pt_create: nee %d", cfg->nee);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) FAIL((pt_ctx*)nullptr, PT_E_NO_DEVICE, "pt_create: no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) FAIL((pt_ctx*)nullptr, PT_E_NO_DEVICE, "pt_create: device %d of %d", cfg->device, ndev);
    if (cfg->device >= kMaxDevices) FAIL((pt_ctx*)nullptr, PT_E_NO_DEVICE, "pt_create: device ordinal %d is beyond the %d this library keeps state for", cfg->device, kMaxDevices);
    pt_ctx* c = new pt_ctx();
    c->cfg = *cfg;
    hipError_t e = hipSetDevice(cfg->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_connect, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream_render, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done0, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_resolved0, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_shaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_connected, hipEventDisableTiming);
    if (e == hipSuccess) e = ptMalloc((void**)&c->d_ctl, sizeof(Ctl));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_ctl, 0, sizeof(Ctl), c->stream);   // (a plain hipMemset of device memory need not be through when it returns)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_end);
    if (e == hipSuccess) e = resolveScratch(cfg->device, &c->scratch_dev);   // with cfg->device current
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, cfg->device);
    if (e != hipSuccess) {
        g_err = std::string("pt_create: ") + hipGetErrorString(e);
        delete c;
        return PT_E_HIP;
    }
    c->num_cu = prop.multiProcessorCount;
    int perCu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_extend<false, PT_WALK_EXACT>, PT_BLOCK, 0) != hipSuccess || perCu < 1) perCu = 4;
    int perCuQ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCuQ, k_extend<false, PT_WALK_QUANT>, PT_BLOCK, 0) != hipSuccess || perCuQ < 1) perCuQ = 4;
    perCu = std::min(perCu, perCuQ);   // one grid size for all forms of the kernel
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCuQ, k_extend<false, PT_WALK_INST>, PT_BLOCK, 0) != hipSuccess || perCuQ < 1) perCuQ = 4;
    c->fit_inst = perCuQ;
    c->fit_extend = perCu;
    perCu = std::min(perCu, 7);   // more resident waves than this thrash the caches on incoherent rays
    c->grid_extend = c->num_cu * perCu;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_connect<false, PT_WALK_EXACT>, PT_BLOCK, 0) != hipSuccess || perCu < 1) perCu = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCuQ, k_connect<false, PT_WALK_QUANT>, PT_BLOCK, 0) != hipSuccess || perCuQ < 1) perCuQ = 4;
    perCu = std::min(perCu, perCuQ);
    c->fit_connect = perCu;
    perCu = std::min(perCu, 7);
    c->grid_connect = c->num_cu * perCu;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_shade<PT_SHADE_BLOCK>, PT_SHADE_BLOCK, 0) != hipSuccess || perCu < 1) perCu = 2;
    c->grid_shade = c->num_cu * perCu;   // one resident round: the kernel strides over the queue
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_shade<PT_SHADE_BLOCK_SMALL>, PT_SHADE_BLOCK_SMALL, 0) != hipSuccess || perCu < 1) perCu = 6;
    c->grid_shade_small = c->num_cu * perCu;
    *out = c;
    return PT_OK;
}

static void freeSecondLane(pt_ctx* c);
int pt_destroy(pt_ctx* c) {
    if (!c) return PT_OK;
    if (c->group) { destroyGroup(c); return PT_OK; }
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream_render) { (void)hipStreamSynchronize(c->stream_render); (void)hipStreamDestroy(c->stream_render); }
    if (c->ev_done0) (void)hipEventDestroy(c->ev_done0);
    if (c->ev_resolved0) (void)hipEventDestroy(c->ev_resolved0);
    freeSecondLane(c);
    freeQueues(c);
    freeScene(c);
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    for (auto& s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->stream_connect) { (void)hipStreamSynchronize(c->stream_connect); (void)hipStreamDestroy(c->stream_connect); }
    if (c->ev_shaded) (void)hipEventDestroy(c->ev_shaded);
    if (c->ev_connected) (void)hipEventDestroy(c->ev_connected);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PT_OK;
}

int pt_set_option(pt_ctx* c, int32_t key, int32_t value) {
    if (!c) return PT_E_INVALID_ARG;
    if (c->group) {
        for (pt_ctx* m : c->group->members) {
            const int rc = pt_set_option(m, key, value);
            if (rc != PT_OK) { c->err = m->err; return rc; }
        }
        return PT_OK;
    }
    switch (key) {
        case PT_OPT_NEE: if (value < 0 || value > 2) FAIL(c, PT_E_INVALID_ARG, "nee %d", value); c->cfg.nee = value; break;
        case PT_OPT_RNG_MODE: if (value != 0 && value != 1) FAIL(c, PT_E_INVALID_ARG, "rng_mode %d", value); c->cfg.rng_mode = value; break;
        case PT_OPT_TRAVERSAL_STATS: c->cfg.traversal_stats = value ? 1 : 0; break;
        case PT_OPT_STAGE_TIMING: if (value < 0 || value > 2) FAIL(c, PT_E_INVALID_ARG, "stage_timing %d", value); c->cfg.stage_timing = value; break;
        case PT_OPT_TRAVERSE_BLOCKS_PER_CU:
            if (value < 0 || value > 16) FAIL(c, PT_E_INVALID_ARG, "traverse blocks per CU %d", value);
            c->grid_extend = c->num_cu * std::min(value ? value : 7, c->fit_extend);
            c->grid_connect = c->num_cu * std::min(value ? value : 7, c->fit_connect);
            c->traverse_blocks_set = true;
            break;
        case PT_OPT_REFILL_PRIMARY: if (value < 1 || value > 64) FAIL(c, PT_E_INVALID_ARG, "refill threshold %d", value); c->refill_primary = value; c->refill_primary_set = true; break;
        case PT_OPT_REFILL_BOUNCE: if (value < 1 || value > 64) FAIL(c, PT_E_INVALID_ARG, "refill threshold %d", value); c->refill_bounce = value; c->refill_bounce_set = true; break;
        case PT_OPT_QUANTIZED: if (value < 0 || value > 2) FAIL(c, PT_E_INVALID_ARG, "quantized %d", value); c->quant_mode = value; break;
        case PT_OPT_INSTANCED: if (value != 0 && value != 1) FAIL(c, PT_E_INVALID_ARG, "instanced %d", value); c->inst_mode = value; break;
        case PT_OPT_PIPELINE_CALLS: if (value != 0 && value != 1) FAIL(c, PT_E_INVALID_ARG, "pipeline_calls %d", value); c->pipeline_calls = value; break;
        case PT_OPT_SAMPLES_PER_BATCH: if (value < 0) FAIL(c, PT_E_INVALID_ARG, "samples_per_batch %d", value); c->cfg.samples_per_batch = value; break;
        default: FAIL(c, PT_E_INVALID_ARG, "unknown option %d", key);
    }
    return PT_OK;
}

static int impl_resize(pt_ctx* c, int32_t width, int32_t height) {
    if (!c) return PT_E_INVALID_ARG;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28)) FAIL(c, PT_E_INVALID_ARG, "pt_resize: %d x %d", width, height);
    if (c->group) {
        Group* g = c->group;
        int rc = g->all(c, [&](pt_ctx* m) { return impl_resize(m, width, height); });
        if (rc != PT_OK) return rc;
        pt_ctx* first = g->members[0];
        HIPCHECK(c, hipSetDevice(first->cfg.device));
        if (g->d_gather) { (void)hipFree(g->d_gather); g->d_gather = nullptr; }
        g->gather_slots = (size_t)std::max(1u, first->slots_padded);
        HIPCHECK(c, ptMalloc((void**)&g->d_gather, g->gather_slots * g->members.size() * sizeof(float4)));
        c->width = width;
        c->height = height;
        return PT_OK;
    }
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    freeQueues(c);
    c->width = width;
    c->height = height;
    c->groups = (uint32_t)((width + 7) / 8) * (uint32_t)((height + 7) / 8);
    const uint32_t world = (uint32_t)c->cfg.tile_world, rank = (uint32_t)c->cfg.tile_rank;
    c->local_groups = (c->groups + world - 1 - rank) / world;
    c->slots = c->local_groups * 64u;
    c->slots_padded = ((c->groups + world - 1) / world) * 64u;   // the same on every rank
    // pixels inside the image among this rank's groups (edge groups of a ragged image are partly empty)
    c->local_pixels = 0;
    const uint32_t gw = (uint32_t)((width + 7) / 8);
    for (uint32_t g = rank; g < c->groups; g += world) {
        const uint32_t gx = g % gw, gy = g / gw;
        c->local_pixels += (uint64_t)std::min(8u, (uint32_t)width - gx * 8u) * std::min(8u, (uint32_t)height - gy * 8u);
    }
    HIPCHECK(c, ptMalloc((void**)&c->d_sum, (size_t)std::max(1u, c->slots_padded) * sizeof(float4)));
    HIPCHECK(c, ptMalloc((void**)&c->d_image, (size_t)width * height * sizeof(float4)));
    HIPCHECK(c, hipMemsetAsync(c->d_sum, 0, (size_t)std::max(1u, c->slots_padded) * sizeof(float4), c->stream));
    c->samples_accumulated = 0;
    c->has_rs = false;
    return PT_OK;
}

int pt_set_render_state(pt_ctx* c, const void* rs32) {
    if (!c || !rs32) return PT_E_INVALID_ARG;
    if (c->group) {
        for (pt_ctx* m : c->group->members) {
            const int rc = pt_set_render_state(m, rs32);
            if (rc != PT_OK) { c->err = m->err; return rc; }
        }
        return PT_OK;
    }
    HostRenderState rs;
    memcpy(&rs, rs32, sizeof(rs));
    if (rs.width != c->width || rs.height != c->height) FAIL(c, PT_E_INVALID_ARG, "render state is %d x %d, pt_resize was %d x %d", rs.width, rs.height, c->width, c->height);
    c->rs = rs;
    c->has_rs = true;
    return PT_OK;
}

int pt_reset_accumulation(pt_ctx* c) {
    if (!c) return PT_E_INVALID_ARG;
    if (c->group) {
        for (pt_ctx* m : c->group->members) {
            const int rc = pt_reset_accumulation(m);
            if (rc != PT_OK) { c->err = m->err; return rc; }
        }
        return PT_OK;
    }
    if (!c->d_sum) FAIL(c, PT_E_STATE, "pt_reset_accumulation before pt_resize");
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipMemsetAsync(c->d_sum, 0, (size_t)std::max(1u, c->slots_padded) * sizeof(float4), c->stream));
    c->samples_accumulated = 0;
    return PT_OK;
}

// The wide pass over the queue, then the two-wide pass over the rays it set aside (normally none).
// whether the traversal kernels walk the quantised records of the installed scene
static bool useQuant(const pt_ctx* c) { return c->quant_ok && (c->quant_mode == 1 || (c->quant_mode == 2 && c->large_tree) || !c->exact_ok); }

// which records the traversal kernels walk for the installed scene
static int walkMode(const pt_ctx* c) {
    if (c->inst_ok && c->inst_mode) return PT_WALK_INST;
    return useQuant(c) ? PT_WALK_QUANT : PT_WALK_EXACT;
}
// Idle lanes a wave collects before it fetches bounce or shadow rays.  Over the quantised records a step costs the memory pipeline
// four loads instead of seven, and waves that refill at 24 idle lanes (16 to 32 measure alike) are 5 % faster than at 40.
static uint32_t bounceRefill(const pt_ctx* c) { return (uint32_t)((walkMode(c) == PT_WALK_QUANT && !c->refill_bounce_set) ? 24 : c->refill_bounce); }
// whether the scene counts as a tree beyond L2 for the queue order and the camera rays' refill threshold (walked as instances it does not)
static bool largeWalk(const pt_ctx* c) { return c->large_tree && walkMode(c) != PT_WALK_INST; }
#define PT_LAUNCH_WALK(KERNEL, FLAG, MODE, GRID, BLOCK, STREAM, ...)                                                   \
    do {                                                                                                                \
        if ((MODE) == PT_WALK_INST) {                                                                                   \
            if (FLAG) hipLaunchKernelGGL((KERNEL<true, PT_WALK_INST>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);            \
            else hipLaunchKernelGGL((KERNEL<false, PT_WALK_INST>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);                \
        } else if ((MODE) == PT_WALK_QUANT) {                                                                           \
            if (FLAG) hipLaunchKernelGGL((KERNEL<true, PT_WALK_QUANT>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);           \
            else hipLaunchKernelGGL((KERNEL<false, PT_WALK_QUANT>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);               \
        } else {                                                                                                        \
            if (FLAG) hipLaunchKernelGGL((KERNEL<true, PT_WALK_EXACT>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);           \
            else hipLaunchKernelGGL((KERNEL<false, PT_WALK_EXACT>), GRID, BLOCK, 0, STREAM, __VA_ARGS__);               \
        }                                                                                                               \
    } while (0)

// The persistent grids fill the chip: as many blocks as fit (the instanced kernels need more registers, fewer fit).  A frame that
// shares the GPU with another one launches four blocks per CU instead: two frames' traversal grids together still fill every CU,
// and beside one of them there is room (registers for two more waves per SIMD) for the other frame's shade and generate blocks, so
// the stage that is bound by memory runs under the stage that is bound by instruction issue (1080p x 16 spp frames back to back:
// 7.30 -> 6.92 ms; five blocks leave room for one wave per SIMD only and measure no gain, three cost the traversal too much).
#define PT_SHARED_BLOCKS_PER_CU 4
static int traverseGrid(const pt_ctx* c, const LaneRef& L, int grid) {
    if (walkMode(c) == PT_WALK_INST) grid = std::min(grid, c->num_cu * c->fit_inst);
    if (L.shared && !c->traverse_blocks_set) grid = std::min(grid, c->num_cu * PT_SHARED_BLOCKS_PER_CU);
    return grid;
}
static void launchExtend(pt_ctx* c, const LaneRef& L, bool stats, int cur, uint32_t refillMin) {
    PT_LAUNCH_WALK(k_extend, stats, walkMode(c), dim3(traverseGrid(c, L, c->grid_extend)), dim3(PT_BLOCK), L.stream, c->sc, *L.q, L.d_ctl, cur, refillMin);
}
// extend of the queue `cur` and, in the same grid, connect of the shadow rays the bounce before left (k_connect_extend; the walk over
// instances keeps the stages apart and has no such kernel)
static void launchConnectExtend(pt_ctx* c, const LaneRef& L, bool stats, int cur, uint32_t refillMin) {
    const dim3 grid(traverseGrid(c, L, c->grid_extend)), block(PT_BLOCK);
    const uint32_t refillShadow = bounceRefill(c);
    if (walkMode(c) == PT_WALK_QUANT) {
        if (stats) hipLaunchKernelGGL((k_connect_extend<true, PT_WALK_QUANT>), grid, block, 0, L.stream, c->sc, *L.q, L.d_ctl, cur, refillMin, refillShadow);
        else hipLaunchKernelGGL((k_connect_extend<false, PT_WALK_QUANT>), grid, block, 0, L.stream, c->sc, *L.q, L.d_ctl, cur, refillMin, refillShadow);
    } else {
        if (stats) hipLaunchKernelGGL((k_connect_extend<true, PT_WALK_EXACT>), grid, block, 0, L.stream, c->sc, *L.q, L.d_ctl, cur, refillMin, refillShadow);
        else hipLaunchKernelGGL((k_connect_extend<false, PT_WALK_EXACT>), grid, block, 0, L.stream, c->sc, *L.q, L.d_ctl, cur, refillMin, refillShadow);
    }
}
static void launchConnect(pt_ctx* c, const LaneRef& L, bool stats, int cur, hipStream_t on) {
    PT_LAUNCH_WALK(k_connect, stats, walkMode(c), dim3(traverseGrid(c, L, c->grid_connect)), dim3(PT_BLOCK), on, c->sc, *L.q, L.d_ctl, cur, bounceRefill(c));
}
static void launchTrace(pt_ctx* c, bool anyhit, const TraceSource& src, int n) {
    PT_LAUNCH_WALK(k_trace, anyhit, walkMode(c), dim3(launchGrid((uint32_t)n)), dim3(PT_BLOCK), c->stream, c->sc, src, n, c->d_ctl);
}

// The second set of streams, events and control block, made when the first render call takes it.
static int ensureSecondLane(pt_ctx* c) {
    pt_ctx::Lane& l = c->lane1;
    if (l.stream) return PT_OK;
    hipError_t e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_resolved, hipEventDisableTiming);
    if (e == hipSuccess) e = ptMalloc((void**)&l.d_ctl, sizeof(Ctl));
    if (e == hipSuccess) e = hipMemsetAsync(l.d_ctl, 0, sizeof(Ctl), l.stream);   // on the stream whose kernels use it: ordered before them
    if (e != hipSuccess) {
        if (l.d_ctl) { (void)hipFree(l.d_ctl); l.d_ctl = nullptr; }
        for (hipEvent_t* ev : {&l.ev_done, &l.ev_resolved}) { if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; } }
        if (l.stream) { (void)hipStreamDestroy(l.stream); l.stream = nullptr; }
        FAIL(c, (e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "second set of streams: %s", hipGetErrorString(e));
    }
    return PT_OK;
}

static void freeSecondLane(pt_ctx* c) {
    pt_ctx::Lane& l = c->lane1;
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    for (void* p : l.qmem) (void)hipFree(p);
    l.qmem.clear();
    l.q = Queues{};
    l.capacity = 0;
    if (l.d_ctl) { (void)hipFree(l.d_ctl); l.d_ctl = nullptr; }
    for (hipEvent_t* ev : {&l.ev_done, &l.ev_resolved}) { if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; } }
    if (l.stream) { (void)hipStreamDestroy(l.stream); l.stream = nullptr; }
}

// The batches of one call, batch k on lanes[k % n_lanes]: with two sets of queues neighbouring batches are in flight together, and their
// planes still enter the accumulator in batch order (on the context's stream), so the image does not depend on the number of sets.
static int renderBatches(pt_ctx* c, const LaneRef* lanes, uint32_t n_lanes, uint32_t batch, const HostFrameState& fs, int32_t spp, int32_t max_bounces, bool generate_only) {
    RenderParams rp{};
    rp.width = c->width;
    rp.height = c->height;
    memcpy(rp.sky, c->rs.sky, 16);
    memcpy(rp.p1, fs.cam.p1, 12); memcpy(rp.p2, fs.cam.p2, 12); memcpy(rp.p3, fs.cam.p3, 12);
    memcpy(rp.pos, fs.cam.pos, 12); memcpy(rp.right, fs.cam.right, 12); memcpy(rp.up, fs.cam.up, 12);
    rp.slots = c->slots;
    rp.rank = c->cfg.tile_rank;
    rp.world = c->cfg.tile_world;
    rp.rng_mode = c->cfg.rng_mode;
    rp.nee = c->cfg.nee;
    const bool stats = c->cfg.traversal_stats != 0;

    if (generate_only) batch = 1u;
    const uint32_t need = c->slots * batch;
    const uint32_t n_batches = ((uint32_t)spp + batch - 1u) / batch;
    for (uint32_t k = 0; k < std::min(n_lanes, n_batches); k++) {
        const LaneRef& L = lanes[k];
        if (need <= *L.capacity) continue;
        // the set's queues are replaced: whatever still reads them (its own streams; the accumulation of its planes, which runs on
        // the context's stream) has to be through
        HIPCHECK(c, hipStreamSynchronize(L.stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream_render));   // (the first set is used from two streams)
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        int rc = allocQueues(c, L, need);
        if (rc != PT_OK) return rc;
    }
    if (c->slots == 0) return PT_OK;

    uint32_t turn = 0;
    for (uint32_t s0 = 0; s0 < (uint32_t)spp; s0 += batch, turn++) {
        const LaneRef& L = lanes[turn % n_lanes];
        const Queues& q = *L.q;
        const uint32_t ns = std::min(batch, (uint32_t)spp - s0);
        const uint32_t total = c->slots * ns;
        rp.frames0 = (uint32_t)fs.frames + s0;
        rp.samples = ns;
        const bool dense = (c->width % 8 == 0) && (c->height % 8 == 0);
        // dense batches: every slot is a path, and shade of bounce 0 stores each path's first value itself
        rp.plane_fresh = (dense && !generate_only) ? 1 : 0;
        rp.group_major = (!generate_only && largeWalk(c)) ? 1 : 0;
        // a set's planes are read by an accumulation that runs on the context's stream: the batch before this one on this set (of
        // this call or of an earlier one) must have been added before its planes, queues and counters are written again
        HIPCHECK(c, hipStreamWaitEvent(L.stream, L.ev_resolved, 0));
        if (!rp.plane_fresh) HIPCHECK(c, hipMemsetAsync(q.plane, 0, (size_t)total * sizeof(float4), L.stream));
        hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, L.stream, L.d_ctl, dense ? total : 0u, 0ull);
        {
            StageTimer t(c, 0, L.stream);
            if (dense) hipLaunchKernelGGL(k_generate<true>, dim3(launchGrid(total)), dim3(PT_BLOCK), 0, L.stream, rp, q, L.d_ctl);
            else hipLaunchKernelGGL(k_generate<false>, dim3(launchGrid(total)), dim3(PT_BLOCK), 0, L.stream, rp, q, L.d_ctl);
            c->host_stats.generate_launches++;
        }
        if (generate_only) break;
        int cur = 0;
        // connect(b) only meets shade(b) (its rays) and shade(b + 1) (the planes, the shadow queue, the cursors): it runs on a
        // stream of its own beside extend(b + 1), whose tail it fills.  Per-stage timing keeps the stages apart on one stream, and
        // so does a call that overlaps with its neighbour (it was given no connect stream): the other frame fills the tails then,
        // and a fifth stream would share a hardware queue (four per process) with the context's stream, whose waits would hold its
        // kernels back (measured: 1 % from two frames in flight with a connect stream per set, 4.4 % without).
        const bool beside = c->cfg.stage_timing != 1 && L.stream_connect != nullptr;
        bool connectPending = false;
        // A frame that shares the GPU traces the shadow rays of every bounce but the last inside the next bounce's extend launch
        // (k_connect_extend): three launches fewer in its chain, and no connect launch that ends on a few long shadow rays (headline
        // +0.6 %, the 8-spp soup frame +2 %), and so does every call over the quantised records (132M- and 265M-path calls on the
        // two large trees: +2 % and +1 % against connect beside the next extend on a stream of its own; the same on the bundled-like
        // 4K scene: -0.9 %, so lone calls over the exact records keep the stages apart).  With any stage timing on, every stage has
        // launches of its own, so that the events around a launch bracket one kind of ray.
        const bool fuse = c->cfg.stage_timing == 0 && walkMode(c) != PT_WALK_INST && (L.shared || walkMode(c) == PT_WALK_QUANT);
        bool connectDeferred = false;
        // a way out of this loop with a connect still running on its stream (a failed call below) waits for it: everything else
        // in the library orders itself after the context's main stream only
        struct Join {
            hipStream_t on; bool* pending;
            ~Join() { if (*pending && on) (void)hipStreamSynchronize(on); }
        } join{L.stream_connect, &connectPending};
        for (int b = 0; b < max_bounces; b++) {
            rp.bounce = b;
            rp.last_bounce = (b == max_bounces - 1) ? 1 : 0;
            {
                StageTimer t(c, b == 0 ? 4 : 1, L.stream);
                const uint32_t refill = (uint32_t)(b == 0 ? ((largeWalk(c) && !c->refill_primary_set) ? 32 : c->refill_primary) : bounceRefill(c));
                if (connectDeferred) launchConnectExtend(c, L, stats, cur, refill);
                else launchExtend(c, L, stats, cur, refill);
                connectDeferred = false;
                c->host_stats.extend_launches++;
            }
            if (connectPending) { HIPCHECK(c, hipStreamWaitEvent(L.stream, L.ev_connected, 0)); connectPending = false; }
            {
                StageTimer t(c, 2, L.stream);
                if (L.shared) hipLaunchKernelGGL(k_shade<PT_SHADE_BLOCK_SMALL>, dim3(c->grid_shade_small), dim3(PT_SHADE_BLOCK_SMALL), 0, L.stream, c->sc, rp, q, L.d_ctl, cur);
                else hipLaunchKernelGGL(k_shade<PT_SHADE_BLOCK>, dim3(c->grid_shade), dim3(PT_SHADE_BLOCK), 0, L.stream, c->sc, rp, q, L.d_ctl, cur);
                c->host_stats.shade_launches++;
            }
            if (rp.nee && c->sc.n_lights > 0 && fuse && b + 1 < max_bounces) {
                connectDeferred = true;
                c->host_stats.connect_launches++;
            } else if (rp.nee && c->sc.n_lights > 0) {
                StageTimer t(c, 3, L.stream);
                hipStream_t on = L.stream;
                if (beside) {
                    HIPCHECK(c, hipEventRecord(L.ev_shaded, L.stream));
                    HIPCHECK(c, hipStreamWaitEvent(L.stream_connect, L.ev_shaded, 0));
                    on = L.stream_connect;
                }
                launchConnect(c, L, stats, cur, on);
                if (beside) { HIPCHECK(c, hipEventRecord(L.ev_connected, L.stream_connect)); connectPending = true; }
                c->host_stats.connect_launches++;
            }
            cur ^= 1;
        }
        if (connectPending) { HIPCHECK(c, hipStreamWaitEvent(L.stream, L.ev_connected, 0)); connectPending = false; }
        hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, L.stream, L.d_ctl, 0u, (unsigned long long)c->local_pixels * ns);
        // the batch's planes go into the accumulator on the context's stream: calls add up in the order they were made
        if (L.hop) {
            HIPCHECK(c, hipEventRecord(L.ev_done, L.stream));
            HIPCHECK(c, hipStreamWaitEvent(c->stream, L.ev_done, 0));
        }
        hipLaunchKernelGGL(k_resolve, dim3(launchGrid(c->slots)), dim3(PT_BLOCK), 0, c->stream, c->d_sum, (const float4*)q.plane, c->slots, ns);
        HIPCHECK(c, hipEventRecord(L.ev_resolved, c->stream));
    }
    HIPCHECK(c, hipGetLastError());
    return PT_OK;
}

static int impl_render(pt_ctx* c, const void* frame_state112, int32_t spp, int32_t max_bounces) {
    if (!c || !frame_state112) return PT_E_INVALID_ARG;
    // every member enqueues its own tiles' launches from its own host thread
    if (c->group) return c->group->all(c, [&](pt_ctx* m) { return impl_render(m, frame_state112, spp, max_bounces); });
    if (spp < 1 || max_bounces < 1 || max_bounces > 64) FAIL(c, PT_E_INVALID_ARG, "pt_render: spp %d, max_bounces %d", spp, max_bounces);
    if (!c->has_scene) FAIL(c, PT_E_STATE, "pt_render before pt_upload_scene");
    if (!c->d_sum) FAIL(c, PT_E_STATE, "pt_render before pt_resize");
    if (!c->has_rs) FAIL(c, PT_E_STATE, "pt_render before pt_set_render_state");
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HostFrameState fs;
    memcpy(&fs, frame_state112, sizeof(fs));
    // FrameState.Samples == 0 is how the host restarts the picture (Game.cs:140,157-165): the running mean of
    // shade.glsl:89-96 then discards the old value, so the accumulator starts over here too
    if (fs.samples == 0 && c->samples_accumulated != 0) {
        HIPCHECK(c, hipMemsetAsync(c->d_sum, 0, (size_t)std::max(1u, c->slots_padded) * sizeof(float4), c->stream));
        c->samples_accumulated = 0;
    }
    HIPCHECK(c, hipEventRecord(c->ev_begin, c->stream));
    // Batches of up to 64M paths alternate between the two sets of streams and queues (11 GB each at most) — the batches of one call
    // and consecutive calls of one batch each alike: two are in flight together, fill each other's tails, and the shade stage of one
    // runs under the traversal of the other.  Larger batches are long enough for their tails not to matter and take the context's
    // own set (halving them to have two in flight measured +1 % on a 265M-path call of 4 bounces and -4 % on one of 8 bounces).
    // With per-stage timing the stages of a call have the GPU to themselves, on the context's stream.
    uint32_t batch = chooseBatch(c, (uint32_t)spp);
    // Over the exact records (traversal bound by instruction issue, shade by memory) a larger call is worth more as several batches
    // of up to 64M paths that share the GPU two at a time than as one batch with the GPU to itself: 3840x2160 x 32 spp as four batches
    // of 8 samples 59.1 -> 57.6 ms (as two of 16: 58.6, as eight of 4: 58.7).  Over the quantised records it is not (the 1M soup's
    // 64 spp as eight batches of 8: 127 -> 138 ms; the 20M-triangle scene's 32 spp as four: 63 -> 68 ms): those keep the one batch.
    if (c->cfg.samples_per_batch == 0 && c->pipeline_calls != 0 && c->cfg.stage_timing != 1 && walkMode(c) == PT_WALK_EXACT &&
        (uint64_t)c->slots * batch > PT_PIPELINE_MAX_PATHS && c->slots != 0 && c->slots <= PT_PIPELINE_MAX_PATHS) {
        const uint32_t most = (uint32_t)(PT_PIPELINE_MAX_PATHS / c->slots);
        const uint32_t batches = ((uint32_t)spp + most - 1u) / most;
        batch = ((uint32_t)spp + batches - 1u) / batches;
    }
    const bool eligible = c->pipeline_calls != 0 && c->cfg.stage_timing != 1 && (uint64_t)c->slots * batch <= PT_PIPELINE_MAX_PATHS;
    LaneRef lanes[2] = {inlineLane(c), inlineLane(c)};
    uint32_t n_lanes = 1;
    if (eligible) {
        // both sets are made ready by the first call that could use them: the second set's queues are not allocated in the middle
        // of a run of frames
        const int rs = ensureSecondLane(c);
        if (rs != PT_OK) return rs;
        const uint32_t need = c->slots * batch;
        bool both = true;
        for (const LaneRef& each : {ownLane(c), secondLane(c)}) {
            if (need <= *each.capacity) continue;
            HIPCHECK(c, hipStreamSynchronize(each.stream));
            HIPCHECK(c, hipStreamSynchronize(c->stream));
            const int ra = allocQueues(c, each, need);
            if (ra == PT_OK) continue;
            if (each.d_ctl == c->d_ctl) return ra;
            // no room for a second set of queues: the context goes on with one, as if the option were off
            c->err.clear();
            (void)hipGetLastError();   // (the failed allocation's error is not this call's)
            c->pipeline_calls = 0;
            both = false;
        }
        if (both) {
            const bool second_first = c->lane1_next;
            lanes[0] = second_first ? secondLane(c) : ownLane(c);
            lanes[1] = second_first ? ownLane(c) : secondLane(c);
            n_lanes = 2;
            const uint32_t n_batches = ((uint32_t)spp + batch - 1u) / batch;
            hipEvent_t const other = second_first ? c->ev_done0 : c->lane1.ev_done;
            if (n_batches >= 2u) {
                lanes[0].shared = lanes[1].shared = true;
            } else if (other == nullptr || hipEventQuery(other) != hipErrorNotReady) {
                // a lone batch that finds the other set idle (a host that waits for every small frame) has nobody to fill its tails:
                // it gets the connect stream, as a call on the context's stream does
                lanes[0].stream_connect = c->stream_connect;
                lanes[0].ev_shaded = c->ev_shaded;
                lanes[0].ev_connected = c->ev_connected;
            } else {
                lanes[0].shared = true;
            }
            if (n_batches & 1u) c->lane1_next = !c->lane1_next;   // the next call starts on the set this one did not end on
        }
    }
    int rc = renderBatches(c, lanes, n_lanes, batch, fs, spp, max_bounces, false);
    if (rc != PT_OK) {
        for (uint32_t k = 0; k < n_lanes; k++)
            if (lanes[k].hop) (void)hipStreamSynchronize(lanes[k].stream);   // nothing of a failed call stays behind on a set's stream
        return rc;
    }
    HIPCHECK(c, hipEventRecord(c->ev_end, c->stream));
    c->render_pending = true;
    c->samples_accumulated += (uint32_t)spp;
    return PT_OK;
}

int pt_synchronize(pt_ctx* c) {
    if (!c) return PT_E_INVALID_ARG;
    if (c->group) {
        for (pt_ctx* m : c->group->members) {
            const int rc = pt_synchronize(m);
            if (rc != PT_OK) { c->err = m->err; return rc; }
        }
        return PT_OK;
    }
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));   // (which has waited for every batch traced on the second set)
    if (c->render_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->host_stats.ms_render = ms;
        c->render_pending = false;
    }
    collectSpans(c);
    return PT_OK;
}

int pt_stream(pt_ctx* c, void** stream) {
    if (!c || !stream) return PT_E_INVALID_ARG;
    if (c->group) FAIL(c, PT_E_STATE, "pt_stream on a multi-GPU context: it has one stream per GPU and orders them itself");
    *stream = (void*)c->stream;
    return PT_OK;
}

int pt_accum_device(pt_ctx* c, void** device_ptr, uint64_t* n_slots_padded, uint32_t* samples) {
    if (!c) return PT_E_INVALID_ARG;
    if (c->group) FAIL(c, PT_E_STATE, "pt_accum_device on a multi-GPU context: pt_read_image gathers the tiles itself");
    if (!c->d_sum) FAIL(c, PT_E_STATE, "pt_accum_device before pt_resize");
    if (device_ptr) *device_ptr = c->d_sum;
    if (n_slots_padded) *n_slots_padded = c->slots_padded;
    if (samples) *samples = c->samples_accumulated;
    return PT_OK;
}

static int impl_assemble_tiles(pt_ctx* c, const void* gathered_device, int32_t world, uint32_t samples, float* rgba_host) {
    if (!c || !gathered_device || !rgba_host) return PT_E_INVALID_ARG;
    if (c->group) FAIL(c, PT_E_STATE, "pt_assemble_tiles on a multi-GPU context: pt_read_image gathers the tiles itself");
    if (world != c->cfg.tile_world) FAIL(c, PT_E_INVALID_ARG, "pt_assemble_tiles: world %d, context was created for %d", world, c->cfg.tile_world);
    if (!c->d_image) FAIL(c, PT_E_STATE, "pt_assemble_tiles before pt_resize");
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    hipLaunchKernelGGL(k_assemble, dim3(launchGrid(c->groups * 64u)), dim3(PT_BLOCK), 0, c->stream, (const float4*)gathered_device, c->slots_padded, world,
                       c->width, c->height, (float)samples, c->d_image);
    HIPCHECK(c, hipGetLastError());
    HIPCHECK(c, hipMemcpyAsync(rgba_host, c->d_image, (size_t)c->width * c->height * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    return pt_synchronize(c);
}

// The once-per-image exchange of a multi-GPU context: every member's accumulator (slots_padded float4) to member 0's GPU,
// then the de-interleave there.  RCCL: one grouped ncclSend per member on its own stream (so it follows that member's
// kernels) and the matching ncclRecv on member 0's stream — a direct gather over the point-to-point xGMI links, no ring.
static int groupReadImage(pt_ctx* c, float* rgba) {
    Group* g = c->group;
    pt_ctx* first = g->members[0];
    if (!g->d_gather) FAIL(c, PT_E_STATE, "pt_read_image before pt_resize");
    const size_t n = g->members.size(), stride = g->gather_slots, bytes = stride * sizeof(float4);
    HIPCHECK(c, hipSetDevice(first->cfg.device));
    if (!g->comms.empty()) {
        ncclResult_t r = g_rccl.groupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; i++) {
            if (i == 0 && n > 1) continue;   // member 0's own tiles are a local copy (below); a lone member sends to itself
            r = g_rccl.send(g->members[i]->d_sum, stride * 4, ncclFloat, 0, g->comms[i], g->members[i]->stream);
            if (r == ncclSuccess) r = g_rccl.recv(g->d_gather + i * stride, stride * 4, ncclFloat, (int)i, g->comms[0], first->stream);
        }
        const ncclResult_t e = g_rccl.groupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) FAIL(c, PT_E_RCCL, "tile gather: %s", g_rccl.errorString(r));
        if (n > 1) HIPCHECK(c, hipMemcpyAsync(g->d_gather, first->d_sum, bytes, hipMemcpyDeviceToDevice, first->stream));
    } else {
        for (size_t i = 0; i < n; i++) {
            pt_ctx* m = g->members[i];
            HIPCHECK(c, hipMemcpyAsync(g->d_gather + i * stride, m->d_sum, bytes, hipMemcpyDeviceToDevice, m->stream));
            if (i > 0) {
                HIPCHECK(c, hipEventRecord(g->ready[i], m->stream));
                HIPCHECK(c, hipStreamWaitEvent(first->stream, g->ready[i], 0));
            }
        }
    }
    const int rc = impl_assemble_tiles(first, g->d_gather, (int32_t)n, std::max(1u, first->samples_accumulated), rgba);
    if (rc != PT_OK) { c->err = first->err; return rc; }
    return pt_synchronize(c);
}

static int impl_read_image(pt_ctx* c, float* rgba) {
    if (!c || !rgba) return PT_E_INVALID_ARG;
    if (c->group) return groupReadImage(c, rgba);
    if (!c->d_sum) FAIL(c, PT_E_STATE, "pt_read_image before pt_resize");
    if (c->cfg.tile_world != 1) FAIL(c, PT_E_STATE, "pt_read_image on a tile-sharded context: gather with pt_accum_device + pt_assemble_tiles");
    return pt_assemble_tiles(c, c->d_sum, 1, std::max(1u, c->samples_accumulated), rgba);
}

static int traceCommon(pt_ctx* c, bool anyhit, const float* o_d, const float* tmax, int32_t n, int32_t* tri, float* t, float* uv, uint8_t* occ) {
    if (!c || n < 0 || (n && !o_d)) return PT_E_INVALID_ARG;
    if (c->group) {   // single queries run on the first GPU
        pt_ctx* m = c->group->members[0];
        const int rc = traceCommon(m, anyhit, o_d, tmax, n, tri, t, uv, occ);
        if (rc != PT_OK) c->err = m->err;
        return rc;
    }
    if (!c->has_scene) FAIL(c, PT_E_STATE, "trace before pt_upload_scene");
    if (n == 0) return PT_OK;
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    float *d_rays = nullptr, *d_tmax = nullptr, *d_t = nullptr, *d_uv = nullptr;
    int32_t* d_tri = nullptr;
    uint8_t* d_occ = nullptr;
    int rc = PT_OK;
    auto cleanup = [&]() { (void)hipFree(d_rays); (void)hipFree(d_tmax); (void)hipFree(d_t); (void)hipFree(d_uv); (void)hipFree(d_tri); (void)hipFree(d_occ); };
#define TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); FAIL(c, (_e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "%s failed: %s", #call, hipGetErrorString(_e)); } } while (0)
    TRY(ptMalloc((void**)&d_rays, (size_t)n * 24));
    TRY(hipMemcpyAsync(d_rays, o_d, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, c->stream, c->d_ctl, 0u, 0ull);
    if (anyhit) {
        TRY(ptMalloc((void**)&d_tmax, (size_t)n * 4));
        TRY(ptMalloc((void**)&d_occ, (size_t)n));
        TRY(hipMemcpyAsync(d_tmax, tmax, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        TraceSource src{d_rays, d_tmax, nullptr, nullptr, nullptr, d_occ};
        launchTrace(c, true, src, n);
        TRY(hipGetLastError());
        TRY(hipMemcpyAsync(occ, d_occ, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    } else {
        TRY(ptMalloc((void**)&d_tri, (size_t)n * 4));
        TRY(ptMalloc((void**)&d_t, (size_t)n * 4));
        if (uv) TRY(ptMalloc((void**)&d_uv, (size_t)n * 8));
        TraceSource src{d_rays, nullptr, d_tri, d_t, d_uv, nullptr};
        launchTrace(c, false, src, n);
        TRY(hipGetLastError());
        if (tri) TRY(hipMemcpyAsync(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (t) TRY(hipMemcpyAsync(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (uv) TRY(hipMemcpyAsync(uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    }
    TRY(hipStreamSynchronize(c->stream));
#undef TRY
    cleanup();
    return rc;
}

static int impl_trace_closest(pt_ctx* c, const float* o_d, int32_t n, int32_t* tri, float* t, float* uv) {
    return traceCommon(c, false, o_d, nullptr, n, tri, t, uv, nullptr);
}

static int impl_trace_occluded(pt_ctx* c, const float* o_d, const float* tmax, int32_t n, uint8_t* occluded) {
    if (n > 0 && (!tmax || !occluded)) return PT_E_INVALID_ARG;
    return traceCommon(c, true, o_d, tmax, n, nullptr, nullptr, nullptr, occluded);
}

// pt_trace_closest without a single allocation: the rays are staged in, and the hits out of, this device's copy of the
// fixed `scratch` array (always through c->scratch_dev, the address hipGetSymbolAddress gave for this device).  Every ray
// of [0, n) is stored by the kernel, so nothing in the array has to start at zero; what a call leaves behind is simply
// overwritten by the next.  Calls that overlap on one GPU share the array and race.
int pt_trace_closest_noalloc(pt_ctx* c, const float* o_d, int32_t n, int32_t* tri, float* t, float* uv) {
    if (!c || n < 0 || (n && !o_d)) return PT_E_INVALID_ARG;
    if (c->group) {
        pt_ctx* m = c->group->members[0];
        const int rc = pt_trace_closest_noalloc(m, o_d, n, tri, t, uv);
        if (rc != PT_OK) c->err = m->err;
        return rc;
    }
    if (n > PT_NOALLOC_MAX_RAYS) FAIL(c, PT_E_INVALID_ARG, "pt_trace_closest_noalloc: %d rays, the fixed staging array holds at most %d per call", n, PT_NOALLOC_MAX_RAYS);
    if (!c->has_scene) FAIL(c, PT_E_STATE, "trace before pt_upload_scene");
    if (!c->scratch_dev) FAIL(c, PT_E_STATE, "pt_trace_closest_noalloc: the staging array was not resolved on device %d", c->cfg.device);
    if (n == 0) return PT_OK;
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    unsigned char* const base = (unsigned char*)c->scratch_dev;
    float* const d_rays = (float*)base;
    int32_t* const d_tri = (int32_t*)(base + (size_t)PT_NOALLOC_MAX_RAYS * 24);
    float* const d_t = (float*)(base + (size_t)PT_NOALLOC_MAX_RAYS * 28);
    float* const d_uv = (float*)(base + (size_t)PT_NOALLOC_MAX_RAYS * 32);
    HIPCHECK(c, hipMemcpyAsync(d_rays, o_d, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, c->stream, c->d_ctl, 0u, 0ull);
    TraceSource src{d_rays, nullptr, d_tri, d_t, uv ? d_uv : nullptr, nullptr};
    launchTrace(c, false, src, n);
    HIPCHECK(c, hipGetLastError());
    if (tri) HIPCHECK(c, hipMemcpyAsync(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (t) HIPCHECK(c, hipMemcpyAsync(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (uv) HIPCHECK(c, hipMemcpyAsync(uv, d_uv, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

static int impl_generate_rays(pt_ctx* c, const void* frame_state112, int32_t n, float* o_d, uint32_t* pixel) {
    if (!c || !frame_state112 || !o_d || !pixel || n < 0) return PT_E_INVALID_ARG;
    if (c->group) FAIL(c, PT_E_STATE, "pt_generate_rays on a multi-GPU context: thread indices are per GPU; use a single-GPU context");
    if (!c->d_sum) FAIL(c, PT_E_STATE, "pt_generate_rays before pt_resize");
    if (!c->has_rs) FAIL(c, PT_E_STATE, "pt_generate_rays before pt_set_render_state");
    if ((uint32_t)n > c->slots) FAIL(c, PT_E_INVALID_ARG, "pt_generate_rays: n %d > local slots %u", n, c->slots);
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HostFrameState fs;
    memcpy(&fs, frame_state112, sizeof(fs));
    const LaneRef only = inlineLane(c);
    int rc = renderBatches(c, &only, 1, 1, fs, 1, 1, true);
    if (rc != PT_OK) return rc;
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    Ctl ctl;
    HIPCHECK(c, hipMemcpy(&ctl, c->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    const uint32_t cnt = ctl.count[0];
    std::vector<float4> o(cnt), d(cnt);
    HIPCHECK(c, hipMemcpy(o.data(), c->q.o[0], (size_t)cnt * 16, hipMemcpyDeviceToHost));
    HIPCHECK(c, hipMemcpy(d.data(), c->q.d[0], (size_t)cnt * 16, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { pixel[i] = 0xFFFFFFFFu; for (int k = 0; k < 6; k++) o_d[6 * i + k] = 0.0f; }
    const uint32_t gcx = (uint32_t)(c->width + 7) / 8;
    auto compact = [](uint32_t x) { x &= 0x55555555; x = (x ^ (x >> 1)) & 0x33333333; x = (x ^ (x >> 2)) & 0x0f0f0f0f; x = (x ^ (x >> 4)) & 0x00ff00ff; x = (x ^ (x >> 8)) & 0x0000ffff; return x; };
    for (uint32_t j = 0; j < cnt; j++) {
        uint32_t birth;
        memcpy(&birth, &o[j].w, 4);
        if (birth >= (uint32_t)n) continue;
        const uint32_t group = (birth >> 6) * (uint32_t)c->cfg.tile_world + (uint32_t)c->cfg.tile_rank, lid = birth & 63u;
        const uint32_t sx = (group % gcx) * 8 + compact(lid), sy = (group / gcx) * 8 + compact(lid >> 1);
        pixel[birth] = sx + sy * (uint32_t)c->width;
        o_d[6 * birth] = o[j].x; o_d[6 * birth + 1] = o[j].y; o_d[6 * birth + 2] = o[j].z;
        o_d[6 * birth + 3] = d[j].x; o_d[6 * birth + 4] = d[j].y; o_d[6 * birth + 5] = d[j].z;
    }
    // the debug batch must not count as rendered samples
    HIPCHECK(c, hipMemsetAsync(c->d_ctl, 0, offsetof(Ctl, paths), c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

static int impl_bench_extend(pt_ctx* c, const float* o_d, int32_t n, int32_t iters, float* avg_ms, int32_t* tri, float* t) {
    if (!c || !o_d || n <= 0 || iters <= 0 || !avg_ms) return PT_E_INVALID_ARG;
    if (c->group) {
        pt_ctx* m = c->group->members[0];
        const int rc = impl_bench_extend(m, o_d, n, iters, avg_ms, tri, t);
        if (rc != PT_OK) c->err = m->err;
        return rc;
    }
    if (!c->has_scene) FAIL(c, PT_E_STATE, "pt_bench_extend before pt_upload_scene");
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if ((uint32_t)n > c->capacity) { int rc = allocQueues(c, inlineLane(c), (uint32_t)n); if (rc != PT_OK) return rc; }
    std::vector<float4> o((size_t)n), d((size_t)n);
    for (int i = 0; i < n; i++) {
        o[i] = make_float4(o_d[6 * i], o_d[6 * i + 1], o_d[6 * i + 2], 0.0f);
        d[i] = make_float4(o_d[6 * i + 3], o_d[6 * i + 4], o_d[6 * i + 5], 0.0f);
    }
    HIPCHECK(c, hipMemcpy(c->q.o[0], o.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    HIPCHECK(c, hipMemcpy(c->q.d[0], d.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    hipEvent_t a = getEvent(c), b = getEvent(c);
    if (!a || !b) FAIL(c, PT_E_HIP, "pt_bench_extend: no events");
    float total = 0.0f;
    const bool stats = c->cfg.traversal_stats != 0;
    for (int it = 0; it < iters + 1; it++) {   // the first launch warms caches and is not counted
        hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, c->stream, c->d_ctl, (uint32_t)n, 0ull);
        HIPCHECK(c, hipEventRecord(a, c->stream));
        launchExtend(c, inlineLane(c), stats, 0, bounceRefill(c));
        HIPCHECK(c, hipEventRecord(b, c->stream));
        HIPCHECK(c, hipStreamSynchronize(c->stream));
        float ms = 0.0f;
        HIPCHECK(c, hipEventElapsedTime(&ms, a, b));
        if (it > 0) total += ms;
    }
    c->pool.push_back(a);
    c->pool.push_back(b);
    *avg_ms = total / (float)iters;
    if (tri || t) {
        std::vector<float4> h((size_t)n);
        HIPCHECK(c, hipMemcpy(h.data(), c->q.hit, (size_t)n * 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) {
            if (t) t[i] = h[i].x;
            if (tri) memcpy(&tri[i], &h[i].y, 4);
        }
    }
    hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, c->stream, c->d_ctl, 0u, 0ull);
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    return PT_OK;
}

static int impl_build_bvh(pt_ctx* c, const void* triangles_in, int32_t n_tri, void* triangles_out, void* nodes_out, int32_t node_capacity, int32_t* n_nodes,
                 float* build_ms) {
    if (!c || n_tri < 0 || (n_tri && (!triangles_in || !triangles_out)) || !nodes_out || !n_nodes) return PT_E_INVALID_ARG;
    if (c->group) {
        pt_ctx* m = c->group->members[0];
        const int rc = impl_build_bvh(m, triangles_in, n_tri, triangles_out, nodes_out, node_capacity, n_nodes, build_ms);
        if (rc != PT_OK) c->err = m->err;
        return rc;
    }
    const int32_t need = n_tri ? 2 * n_tri - 1 : 1;
    if (node_capacity < need) FAIL(c, PT_E_INVALID_ARG, "pt_build_bvh: room for %d nodes, %d triangles may need %d", node_capacity, n_tri, need);
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    ptbuild::Tri112 *d_in = nullptr, *d_out = nullptr;
    ptbuild::Node32* d_nodes = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_nodes); };
#define TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); FAIL(c, (_e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "%s failed: %s", #call, hipGetErrorString(_e)); } } while (0)
    if (n_tri) {
        TRY(ptMalloc((void**)&d_in, (size_t)n_tri * 112));
        TRY(ptMalloc((void**)&d_out, (size_t)n_tri * 112));
        TRY(hipMemcpyAsync(d_in, triangles_in, (size_t)n_tri * 112, hipMemcpyHostToDevice, c->stream));
    }
    TRY(ptMalloc((void**)&d_nodes, (size_t)need * 32));
    hipEvent_t a = getEvent(c), b = getEvent(c);
    if (a) (void)hipEventRecord(a, c->stream);
    uint32_t nn = 0, levels = 0;
    TRY(ptbuild::build(c->stream, d_in, (uint32_t)n_tri, d_nodes, d_out, &nn, &levels));
    if (b) (void)hipEventRecord(b, c->stream);
    if (n_tri) TRY(hipMemcpyAsync(triangles_out, d_out, (size_t)n_tri * 112, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(nodes_out, d_nodes, (size_t)nn * 32, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
#undef TRY
    if (build_ms) {
        *build_ms = 0.0f;
        if (a && b) (void)hipEventElapsedTime(build_ms, a, b);
    }
    if (a) c->pool.push_back(a);
    if (b) c->pool.push_back(b);
    cleanup();
    *n_nodes = (int32_t)nn;
    return PT_OK;
}

// Buffers of one scene installation (pt_upload_scene, pt_build_scene).  Everything is made in buffers of the call; the context's
// scene is replaced only when all of it exists, and the destructor releases whatever is still held on any way out (error returns
// and exceptions alike) after waiting for the stream, so nothing queued can outlive the buffers or this object.
struct SceneWork {
    pt_ctx* c;
    ptbuild::Tri112 *d_in = nullptr, *d_sorted = nullptr;
    ptbuild::Node32* d_nodes = nullptr;
    ptbuild::BNode* bn = nullptr;
    uint32_t *d_flagA = nullptr, *d_idxA = nullptr, *d_flagB = nullptr, *d_idxB = nullptr, *d_bad = nullptr;
    uint32_t *d_parent = nullptr, *d_refs = nullptr, *d_anc[2] = {nullptr, nullptr}, *d_dep[2] = {nullptr, nullptr};
    ptrepack::UploadCheck* d_check = nullptr;
    void* tmp = nullptr;
    size_t tmpBytes = 0;
    void *pairs = nullptr, *quads = nullptr, *big = nullptr, *hot = nullptr, *cold = nullptr, *mats = nullptr, *lights = nullptr;   // the new scene
    void *qquads = nullptr, *gate = nullptr;
    void *iquads = nullptr, *ibig = nullptr, *itris = nullptr, *iinst = nullptr, *built_of = nullptr;   // the scene as instances (optional)
    bool instanced = false;
    uint32_t itopEntry = 0;
    float itopMin[3] = {0, 0, 0}, itopMax[3] = {0, 0, 0}, instReach[3] = {-1, -1, -1};
    std::vector<void*> scratch;   // device buffers of the instanced build, released with this object
    uint32_t* d_qbad = nullptr;
    uint32_t qbad = 0;         // a box that does not fit the grid (non-finite): the scene keeps to the exact records
    hipEvent_t ea = nullptr, eb = nullptr;
    float lightSum = 0.0f;     // written by copies queued on the stream
    uint32_t rbTri[3] = {0, 0, 0}, rbNode[4] = {0, 0, 0, 0};   // small read-backs that size allocations: kept here, not on a caller's stack,
                                                              // because a copy may still be pending when an error return unwinds
    ptbuild::BNode root{};
    uint32_t nLights = 0, nQuads = 0;
    explicit SceneWork(pt_ctx* ctx) : c(ctx) {}
    SceneWork(const SceneWork&) = delete;
    SceneWork& operator=(const SceneWork&) = delete;
    ~SceneWork() {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        void* all[] = {d_in, d_sorted, d_nodes, bn, d_flagA, d_idxA, d_flagB, d_idxB, d_bad, d_parent, d_refs, d_anc[0], d_anc[1], d_dep[0], d_dep[1], d_check, tmp,
                       pairs, quads, big, hot, cold, mats, lights, qquads, gate, d_qbad, iquads, ibig, itris, iinst, built_of};
        for (void* p : all) if (p) (void)hipFree(p);
        for (void* p : scratch) if (p) (void)hipFree(p);
        if (ea) c->pool.push_back(ea);
        if (eb) c->pool.push_back(eb);
    }
};
#define TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { FAIL(c, (_e == hipErrorOutOfMemory ? PT_E_OOM : PT_E_HIP), "%s failed: %s", #call, hipGetErrorString(_e)); } } while (0)

// The device scene — hot / cold triangles, light table, pair and quad records, big-leaf rows — from triangles in tree order
// (w.d_sorted) and the tree as BNode records whose children sit side by side (w.bn, nn of them): queued on the context's stream,
// except for two small read-backs that size the allocations.
static int repackOnDevice(pt_ctx* c, SceneWork& w, uint32_t n, uint32_t nn, const void* materials, int32_t n_mat) {
    const size_t flagsLen = std::max<size_t>(std::max<size_t>(n, nn), 1);
    TRY(ptMalloc((void**)&w.d_flagA, flagsLen * 4));
    TRY(ptMalloc((void**)&w.d_idxA, flagsLen * 4));
    TRY(ptMalloc((void**)&w.d_flagB, flagsLen * 4));
    TRY(ptMalloc((void**)&w.d_idxB, flagsLen * 4));
    TRY(ptMalloc((void**)&w.d_bad, 4));
    TRY(hipMemsetAsync(w.d_bad, 0, 4, c->stream));
    if (n_mat) {
        TRY(ptMalloc(&w.mats, (size_t)n_mat * 48));
        TRY(hipMemcpyAsync(w.mats, materials, (size_t)n_mat * 48, hipMemcpyHostToDevice, c->stream));
    }
    // ---- triangles: hot / cold records, the light table
    if (n) {
        TRY(ptMalloc(&w.hot, (size_t)n * 48));
        TRY(ptMalloc(&w.cold, (size_t)n * 48));
        hipLaunchKernelGGL(ptrepack::k_triangles, dim3(launchGrid(n)), dim3(256), 0, c->stream, w.d_sorted, n, (const float4*)w.mats, n_mat,
                           (float4*)w.hot, (float4*)w.cold, w.d_flagA, w.d_bad);
        TRY(ptrepack::exclusiveScan(c->stream, w.d_flagA, w.d_idxA, n, &w.tmp, &w.tmpBytes));
        TRY(hipMemcpyAsync(&w.rbTri[0], w.d_flagA + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&w.rbTri[1], w.d_idxA + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&w.rbTri[2], w.d_bad, 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipStreamSynchronize(c->stream));
        if (w.rbTri[2]) FAIL(c, PT_E_INVALID_ARG, "a triangle has a material index outside [0, %d)", n_mat);
        w.nLights = w.rbTri[0] + w.rbTri[1];
        if (w.nLights) {
            // rows in triangle order; the running sum rounds as a CPU loop over the triangles would (d_bad is free again: it receives the total)
            TRY(ptMalloc(&w.lights, (size_t)w.nLights * 3 * sizeof(float4)));
            float* const areas = (float*)w.d_flagB;   // free until the node flags are formed
            hipLaunchKernelGGL(ptrepack::k_light_rows, dim3(launchGrid(n)), dim3(256), 0, c->stream, w.d_sorted, n, w.d_flagA, w.d_idxA, (float4*)w.lights, areas);
            hipLaunchKernelGGL(ptrepack::k_light_sums, dim3(1), dim3(64), 0, c->stream, areas, (float4*)w.lights, w.nLights, (float*)w.d_bad);
            TRY(hipMemcpyAsync(&w.lightSum, w.d_bad, 4, hipMemcpyDeviceToHost, c->stream));   // read after the final synchronisation
        }
    }
    // ---- nodes: pair records, quad records, big-leaf rows
    hipLaunchKernelGGL(ptrepack::k_flags, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.bn, nn, w.d_flagA, w.d_flagB);
    TRY(ptrepack::exclusiveScan(c->stream, w.d_flagA, w.d_idxA, nn, &w.tmp, &w.tmpBytes));
    TRY(ptrepack::exclusiveScan(c->stream, w.d_flagB, w.d_idxB, nn, &w.tmp, &w.tmpBytes));
    uint32_t* const v = w.rbNode;
    TRY(hipMemcpyAsync(&v[0], w.d_flagA + (nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(&v[1], w.d_idxA + (nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(&v[2], w.d_flagB + (nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(&v[3], w.d_idxB + (nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    const uint32_t nQuads = v[0] + v[1], nBig = v[2] + v[3];
    w.nQuads = nQuads;
    if (nBig >= PT_LEAF_MAX_FIRST) FAIL(c, PT_E_BAD_BVH, "%u leaves of more than %d triangles", nBig, PT_LEAF_MAX_COUNT);
    if (nn > 1) TRY(ptMalloc(&w.pairs, (size_t)(nn - 1) * 32));
    if (nQuads) TRY(ptMalloc(&w.quads, (size_t)nQuads * 128));
    if (nBig) TRY(ptMalloc(&w.big, (size_t)nBig * 8));
    hipLaunchKernelGGL(ptrepack::k_records, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.bn, nn, w.d_idxA, w.d_idxB, w.d_flagB, (float4*)w.pairs,
                       (float4*)w.quads, (uint2*)w.big);
    // the same records in 64 bytes on the scene's grid, and the triangles with their leaves' boxes beside them
    TRY(ptMalloc((void**)&w.d_qbad, 4));
    TRY(hipMemsetAsync(w.d_qbad, 0, 4, c->stream));
    if (nQuads) {
        TRY(ptMalloc(&w.qquads, (size_t)nQuads * 64));
        hipLaunchKernelGGL(ptrepack::k_qrecords, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.bn, nn, w.d_idxA, w.d_idxB, (uint4*)w.qquads, w.d_qbad);
    }
    if (n) {
        TRY(ptMalloc(&w.gate, (size_t)n * 64));
        hipLaunchKernelGGL(ptrepack::k_tri_gate, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.bn, nn, (const float4*)w.hot, (float4*)w.gate);
    }
    TRY(hipMemcpyAsync(&w.qbad, w.d_qbad, 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipMemcpyAsync(&w.root, w.bn, sizeof(w.root), hipMemcpyDeviceToHost, c->stream));
    return PT_OK;
}

// The new scene replaces the old one (after the stream has finished everything repackOnDevice queued).
static void adoptScene(pt_ctx* c, SceneWork& w, uint32_t n, uint32_t nn) {
    freeScene(c);
    c->d_pairs = w.pairs; c->d_quads = w.quads; c->d_big = w.big; c->d_hot = w.hot; c->d_cold = w.cold; c->d_mats = w.mats; c->d_lights = w.lights;
    c->d_qquads = w.qquads; c->d_gate = w.gate;
    c->d_iquads = w.iquads; c->d_ibig = w.ibig; c->d_itris = w.itris; c->d_iinst = w.iinst; c->d_built_of = w.built_of;
    w.iquads = w.ibig = w.itris = w.iinst = w.built_of = nullptr;
    w.pairs = w.quads = w.big = w.hot = w.cold = w.mats = w.lights = w.qquads = w.gate = nullptr;
    DevScene& sc = c->sc;
    sc.pairs = (const float4*)c->d_pairs;
    sc.quads = (const float4*)c->d_quads;
    sc.big = (const uint2*)c->d_big;
    sc.tri_hot = (const float4*)c->d_hot;
    sc.tri_cold = (const float4*)c->d_cold;
    sc.mats = (const float4*)c->d_mats;
    sc.lights = (const float4*)c->d_lights;
    sc.qquads = (const uint4*)c->d_qquads;
    sc.tri_gate = (const float4*)c->d_gate;
    sc.n_nodes = n ? (int32_t)nn : 0;
    sc.n_tris = (int32_t)n;
    sc.n_lights = (int32_t)w.nLights;
    sc.light_total = w.lightSum;
    c->large_tree = (size_t)w.nQuads * 128 > ((size_t)32 << 20);
    for (int k = 0; k < 3; k++) { sc.root_min[k] = w.root.mn[k]; sc.root_max[k] = w.root.mx[k]; }
    sc.root_left = w.root.left >= 0 ? w.root.left : w.root.first;
    sc.root_count = w.root.left >= 0 ? 0 : w.root.count;
    ptrepack::QGrid grid;
    ptrepack::gridOf(w.root.mn, w.root.mx, &grid);   // the expression k_qrecords evaluated on the device
    for (int k = 0; k < 3; k++) { sc.grid_min[k] = grid.mn[k]; sc.grid_cell[k] = grid.cell[k]; sc.grid_mid[k] = grid.mid[k]; sc.grid_reach[k] = grid.reach[k]; }
    c->quant_ok = n > 0 && w.qbad == 0 && (w.nQuads == 0 || c->d_qquads != nullptr) && c->d_gate != nullptr;
    c->exact_ok = (size_t)w.nQuads * 128 < ((size_t)1 << 32);
    sc.iquads = (const float4*)c->d_iquads;
    sc.ibig = (const uint2*)c->d_ibig;
    sc.itris = (const float4*)c->d_itris;
    sc.iinst = (const float4*)c->d_iinst;
    sc.built_of = (const uint32_t*)c->d_built_of;
    sc.itop_entry = w.itopEntry;
    for (int k = 0; k < 3; k++) { sc.itop_min[k] = w.itopMin[k]; sc.itop_max[k] = w.itopMax[k]; sc.inst_reach[k] = w.instReach[k]; }
    c->inst_ok = w.instanced;
    c->has_scene = true;
}


// ---- the scene as instances of meshes, beside the flattened scene (pt_build_scene_instanced)
struct InstancedInput {
    const void* mesh_tris; const int32_t* mesh_first; const int32_t* mesh_count; int32_t n_meshes;
    const pt_instance* inst; int32_t n_inst;
};

// order: built position -> position in the host's list, of the flattened build (device).  w.d_in still holds the host's list.
static int buildInstances(pt_ctx* c, SceneWork& w, uint32_t n, const InstancedInput& in, const uint32_t* order) {
    const uint32_t nm = (uint32_t)in.n_meshes, ni = (uint32_t)in.n_inst;
    // ---- the host's description must tile the flattened list: instance after instance, each as long as its mesh
    uint32_t meshTris = 0;
    for (uint32_t m = 0; m < nm; m++) {
        if (in.mesh_first[m] < 0 || in.mesh_count[m] < 1) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: mesh %u has range [%d, +%d)", m, in.mesh_first[m], in.mesh_count[m]);
        meshTris = std::max(meshTris, (uint32_t)in.mesh_first[m] + (uint32_t)in.mesh_count[m]);
    }
    std::vector<uint32_t> byFirst(ni), istart(ni + 1, 0);
    for (uint32_t i = 0; i < ni; i++) {
        byFirst[i] = i;
        if (in.inst[i].mesh < 0 || (uint32_t)in.inst[i].mesh >= nm) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: instance %u names mesh %d of %u", i, in.inst[i].mesh, nm);
        istart[i + 1] = istart[i] + (uint32_t)in.mesh_count[in.inst[i].mesh];
    }
    std::sort(byFirst.begin(), byFirst.end(), [&](uint32_t a, uint32_t b) { return in.inst[a].first_triangle < in.inst[b].first_triangle; });
    long long cursor = 0;
    for (uint32_t k = 0; k < ni; k++) {
        const pt_instance& it = in.inst[byFirst[k]];
        if (it.first_triangle != cursor) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: the instances do not tile the triangle list (instance %u starts at %d, expected %lld)", byFirst[k], it.first_triangle, cursor);
        cursor += in.mesh_count[it.mesh];
    }
    if (cursor != (long long)n) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: the instances cover %lld triangles, the list has %u", cursor, n);

    auto dev = [&](size_t bytes) -> void* { void* p = nullptr; if (ptMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; w.scratch.push_back(p); return p; };
#define NEED(ptr) do { if (!(ptr)) FAIL(c, PT_E_OOM, "pt_build_scene_instanced: out of device memory"); } while (0)
    ptbuild::Tri112* d_mesh = (ptbuild::Tri112*)dev((size_t)meshTris * 112);
    ptinst::HostInstance* d_inst = (ptinst::HostInstance*)dev((size_t)ni * 72);
    int32_t* d_mfirst = (int32_t*)dev((size_t)nm * 4);
    int32_t* d_mcount = (int32_t*)dev((size_t)nm * 4);
    uint32_t* d_istart = (uint32_t*)dev((size_t)(ni + 1) * 4);
    uint32_t* d_small = (uint32_t*)dev((size_t)(nm + 4) * 4);          // [0] mismatches, [4..] per-mesh stretch (float bits)
    ptinst::InstWork* d_work = (ptinst::InstWork*)dev((size_t)ni * sizeof(ptinst::InstWork));
    NEED(d_mesh); NEED(d_inst); NEED(d_mfirst); NEED(d_mcount); NEED(d_istart); NEED(d_small); NEED(d_work);
    TRY(hipMemcpyAsync(d_mesh, in.mesh_tris, (size_t)meshTris * 112, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_inst, in.inst, (size_t)ni * 72, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_mfirst, in.mesh_first, (size_t)nm * 4, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_mcount, in.mesh_count, (size_t)nm * 4, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(d_istart, istart.data(), (size_t)(ni + 1) * 4, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemsetAsync(d_small, 0, (size_t)(nm + 4) * 4, c->stream));
    TRY(ptMalloc(&w.built_of, (size_t)n * 4));
    hipLaunchKernelGGL(ptinst::k_invert, dim3(launchGrid(n)), dim3(256), 0, c->stream, order, n, (uint32_t*)w.built_of);
    hipLaunchKernelGGL(ptinst::k_verify, dim3(launchGrid(istart[ni])), dim3(256), 0, c->stream, w.d_in, d_mesh, d_inst, d_mfirst, d_mcount, d_istart, ni, istart[ni], d_small);
    hipLaunchKernelGGL(ptinst::k_instance_boxes, dim3(ni), dim3(64), 0, c->stream, w.d_in, d_inst, d_mcount, ni, d_work, d_small + 4);
    std::vector<uint32_t> small(nm + 4);
    TRY(hipMemcpyAsync(small.data(), d_small, (size_t)(nm + 4) * 4, hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));   // also completes the copy of w.root that repackOnDevice queued
    if (small[0]) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: %u triangles of the list are not their mesh triangle under the instance matrix (Vector3.Transform), bit for bit", small[0]);

    // ---- how far the boxes must be widened (see rayOffInstances in pt_device.hpp)
    ptrepack::QGrid grid;
    ptrepack::gridOf(w.root.mn, w.root.mx, &grid);
    float widest = 0.0f;
    for (int a = 0; a < 3; a++) {
        widest = std::max(widest, w.root.mx[a] - w.root.mn[a]);
        w.instReach[a] = grid.reach[a] < 0.0f ? -1.0f : grid.reach[a] / 8.0f;   // two (effective) extents around the middle
    }
    const float farthest = 1.7320508f * 2.5f * widest;   // from a ray origin within reach to any point of the scene
    std::vector<float> padObj(nm);
    for (uint32_t m = 0; m < nm; m++) {
        float stretch;
        memcpy(&stretch, &small[4 + m], 4);
        if (!(stretch < 1.0e30f)) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: an instance of mesh %u has a singular matrix", m);
        padObj[m] = 16.0f * 5.9604645e-8f * (farthest * stretch + 2.0f);   // + the mesh's own size below
    }

    // ---- one tree per mesh
    struct MeshBuild { ptbuild::BNode* bn = nullptr; uint32_t* order = nullptr; ptbuild::Tri112* sorted = nullptr; uint32_t *flagA = nullptr, *idxA = nullptr, *flagB = nullptr, *idxB = nullptr;
                       uint32_t nn = 0, levels = 0, quads = 0, bigs = 0, quadBase = 0, triBase = 0, bigBase = 0; ptbuild::BNode root{}; };
    auto buildTree = [&](const ptbuild::Tri112* tris, uint32_t count, uint32_t triBase, MeshBuild& b) -> int {
        b.sorted = (ptbuild::Tri112*)dev((size_t)count * 112);
        ptbuild::Node32* nodesTmp = (ptbuild::Node32*)dev((size_t)(2 * count) * 32);
        NEED(b.sorted); NEED(nodesTmp);
        TRY(ptbuild::build(c->stream, tris, count, nodesTmp, b.sorted, &b.nn, &b.levels, &b.bn, &b.order));
        w.scratch.push_back(b.bn);
        w.scratch.push_back(b.order);
        b.flagA = (uint32_t*)dev((size_t)b.nn * 4); b.idxA = (uint32_t*)dev((size_t)b.nn * 4);
        b.flagB = (uint32_t*)dev((size_t)b.nn * 4); b.idxB = (uint32_t*)dev((size_t)b.nn * 4);
        NEED(b.flagA); NEED(b.idxA); NEED(b.flagB); NEED(b.idxB);
        b.triBase = triBase;
        hipLaunchKernelGGL(ptinst::k_iflags, dim3(launchGrid(b.nn)), dim3(256), 0, c->stream, b.bn, b.nn, triBase, b.flagA, b.flagB);
        TRY(ptrepack::exclusiveScan(c->stream, b.flagA, b.idxA, b.nn, &w.tmp, &w.tmpBytes));
        TRY(ptrepack::exclusiveScan(c->stream, b.flagB, b.idxB, b.nn, &w.tmp, &w.tmpBytes));
        uint32_t v[4] = {0, 0, 0, 0};
        TRY(hipMemcpyAsync(&v[0], b.flagA + (b.nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&v[1], b.idxA + (b.nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&v[2], b.flagB + (b.nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&v[3], b.idxB + (b.nn - 1), 4, hipMemcpyDeviceToHost, c->stream));
        TRY(hipMemcpyAsync(&b.root, b.bn, sizeof(b.root), hipMemcpyDeviceToHost, c->stream));
        TRY(hipStreamSynchronize(c->stream));   // v and b.root live until here
        b.quads = v[0] + v[1];
        b.bigs = v[2] + v[3];
        return PT_OK;
    };
    std::vector<MeshBuild> meshes(nm);
    uint32_t triBase = 0, deepest = 0;
    for (uint32_t m = 0; m < nm; m++) {
        const int rc = buildTree(d_mesh + in.mesh_first[m], (uint32_t)in.mesh_count[m], triBase, meshes[m]);
        if (rc != PT_OK) return rc;
        triBase += (uint32_t)in.mesh_count[m];
        deepest = std::max(deepest, meshes[m].levels);
        float size = 0.0f;
        for (int a = 0; a < 3; a++) size = std::max(size, std::max(std::fabs(meshes[m].root.mn[a]), std::fabs(meshes[m].root.mx[a])));
        padObj[m] += 16.0f * 5.9604645e-8f * 2.0f * size;
    }
    // ---- the tree over the instances' widened world boxes
    float* d_pad = (float*)dev((size_t)nm * 4);
    ptbuild::Tri112* d_prims = (ptbuild::Tri112*)dev((size_t)ni * 112);
    NEED(d_pad); NEED(d_prims);
    TRY(hipMemcpyAsync(d_pad, padObj.data(), (size_t)nm * 4, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ptinst::k_instance_prims, dim3(launchGrid(ni)), dim3(256), 0, c->stream, d_inst, d_work, d_pad, ni, d_prims);
    MeshBuild top;
    {
        const int rc = buildTree(d_prims, ni, 0u, top);
        if (rc != PT_OK) return rc;
    }
    if (top.bigs) FAIL(c, PT_E_BAD_BVH, "pt_build_scene_instanced: %u leaves of the tree over the instances hold more than %d instances (coinciding instances?)", top.bigs, PT_LEAF_MAX_COUNT);
    if (3 * (((int)top.levels + 1) / 2) + 2 + 3 * (((int)deepest + 1) / 2) > PT_STACK_TOTAL)
        FAIL(c, PT_E_BAD_BVH, "pt_build_scene_instanced: tree depths %u + %u exceed the traversal stack (%d entries)", top.levels, deepest, PT_STACK_TOTAL);
    // ---- the shared arrays: the top-level tree's records first
    uint32_t quadTotal = top.quads, bigTotal = 0;
    for (MeshBuild& b : meshes) { b.quadBase = quadTotal; b.bigBase = bigTotal; quadTotal += b.quads; bigTotal += b.bigs; }
    if (triBase >= PT_LEAF_MAX_FIRST || quadTotal >= PT_LEAF_MAX_FIRST) FAIL(c, PT_E_BAD_BVH, "pt_build_scene_instanced: %u mesh triangles, %u records", triBase, quadTotal);
    TRY(ptMalloc(&w.iquads, std::max<size_t>((size_t)quadTotal, 1) * 128));
    TRY(ptMalloc(&w.ibig, std::max<size_t>((size_t)bigTotal, 1) * 8));
    TRY(ptMalloc(&w.itris, (size_t)triBase * 48));
    TRY(ptMalloc(&w.iinst, (size_t)ni * 128));
    hipLaunchKernelGGL(ptinst::k_irecords, dim3(launchGrid(top.nn)), dim3(256), 0, c->stream, top.bn, top.nn, top.idxA, top.idxB, top.flagB, 0u, 0u, 0u, 0.0f,
                       (float4*)w.iquads, (uint2*)w.ibig);
    std::vector<ptinst::MeshInfo> info(nm);
    for (uint32_t m = 0; m < nm; m++) {
        MeshBuild& b = meshes[m];
        hipLaunchKernelGGL(ptinst::k_irecords, dim3(launchGrid(b.nn)), dim3(256), 0, c->stream, b.bn, b.nn, b.idxA, b.idxB, b.flagB, b.quadBase, b.triBase, b.bigBase, padObj[m],
                           (float4*)w.iquads, (uint2*)w.ibig);
        hipLaunchKernelGGL(ptinst::k_itris, dim3(launchGrid((uint32_t)in.mesh_count[m])), dim3(256), 0, c->stream, b.sorted, b.order, (uint32_t)in.mesh_count[m], b.triBase, (float4*)w.itris);
        if (b.root.left >= 0) info[m].root_entry = b.quadBase + 1u;
        else if (b.root.count <= PT_LEAF_MAX_COUNT + 1 && b.triBase + (uint32_t)b.root.first < PT_LEAF_MAX_FIRST)
            info[m].root_entry = PT_LEAF_FLAG | ((uint32_t)(b.root.count - 1) << 27) | (b.triBase + (uint32_t)b.root.first);
        else info[m].root_entry = PT_LEAF_FLAG | (15u << 27) | b.bigBase;   // the root is the tree's only (big) leaf: row 0 of its rows
        for (int a = 0; a < 3; a++) { info[m].mn[a] = b.root.mn[a] - padObj[m]; info[m].mx[a] = b.root.mx[a] + padObj[m]; }
    }
    ptinst::MeshInfo* d_info = (ptinst::MeshInfo*)dev((size_t)nm * sizeof(ptinst::MeshInfo));
    NEED(d_info);
    TRY(hipMemcpyAsync(d_info, info.data(), (size_t)nm * sizeof(ptinst::MeshInfo), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ptinst::k_iinst, dim3(launchGrid(ni)), dim3(256), 0, c->stream, d_inst, d_work, top.order, d_info, ni, (float4*)w.iinst);
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(c->stream));   // info (host) is read by the copy above
    w.itopEntry = top.root.left >= 0 ? 1u : (PT_LEAF_FLAG | ((uint32_t)(top.root.count - 1) << 27) | (uint32_t)top.root.first);
    for (int a = 0; a < 3; a++) { w.itopMin[a] = top.root.mn[a]; w.itopMax[a] = top.root.mx[a]; }
    w.instanced = true;
#undef NEED
    return PT_OK;
}

static int impl_build_scene(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles_in, int32_t n_tri, void* triangles_out, void* nodes_out,
                   int32_t node_capacity, int32_t* n_nodes, float* build_ms, const InstancedInput* instanced = nullptr) {
    if (!c || n_mat < 0 || n_tri < 0 || (n_mat && !materials) || (n_tri && !triangles_in)) return PT_E_INVALID_ARG;
    if (c->group) {
        // every GPU builds its own replica (the build is deterministic: the trees are identical); the first hands the arrays back
        pt_ctx* first = c->group->members[0];
        return c->group->all(c, [&](pt_ctx* m) {
            const bool lead = m == first;
            return impl_build_scene(m, materials, n_mat, triangles_in, n_tri, lead ? triangles_out : nullptr, lead ? nodes_out : nullptr, node_capacity,
                                    lead ? n_nodes : nullptr, lead ? build_ms : nullptr, instanced);
        });
    }
    const int32_t need = n_tri ? 2 * n_tri - 1 : 1;
    if (nodes_out && node_capacity < need) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene: room for %d nodes, %d triangles may need %d", node_capacity, n_tri, need);
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    const uint32_t n = (uint32_t)n_tri;
    SceneWork w(c);
    if (n) {
        TRY(ptMalloc((void**)&w.d_in, (size_t)n * 112));
        TRY(ptMalloc((void**)&w.d_sorted, (size_t)n * 112));
        TRY(hipMemcpyAsync(w.d_in, triangles_in, (size_t)n * 112, hipMemcpyHostToDevice, c->stream));
    }
    TRY(ptMalloc((void**)&w.d_nodes, (size_t)need * 32));
    w.ea = getEvent(c);
    w.eb = getEvent(c);
    if (w.ea) (void)hipEventRecord(w.ea, c->stream);
    uint32_t nn = 0, levels = 0;
    uint32_t* order = nullptr;
    TRY(ptbuild::build(c->stream, w.d_in, n, w.d_nodes, w.d_sorted, &nn, &levels, &w.bn, instanced ? &order : nullptr));
    if (order) w.scratch.push_back(order);
    if (!instanced) {
        (void)hipFree(w.d_in);
        w.d_in = nullptr;
    }
    if ((int)levels > PT_STACK_TOTAL || 3 * (((int)levels + 1) / 2) > PT_STACK_TOTAL)
        FAIL(c, PT_E_BAD_BVH, "BVH depth %u exceeds the traversal stack (%d entries)", levels, PT_STACK_TOTAL);
    const int rc = repackOnDevice(c, w, n, nn, materials, n_mat);
    if (rc != PT_OK) return rc;
    if (instanced) {
        if (n < 3) FAIL(c, PT_E_INVALID_ARG, "pt_build_scene_instanced: %u triangles", n);
        const int ri = buildInstances(c, w, n, *instanced, order);
        if (ri != PT_OK) return ri;
    }
    if (w.eb) (void)hipEventRecord(w.eb, c->stream);
    if (triangles_out && n) TRY(hipMemcpyAsync(triangles_out, w.d_sorted, (size_t)n * 112, hipMemcpyDeviceToHost, c->stream));
    if (nodes_out) TRY(hipMemcpyAsync(nodes_out, w.d_nodes, (size_t)nn * 32, hipMemcpyDeviceToHost, c->stream));
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(c->stream));
    if ((size_t)w.nQuads * 128 >= ((size_t)1 << 32) && w.qbad != 0)
        FAIL(c, PT_E_INVALID_ARG, "%u quad records exceed 4 GB and the scene's extent does not admit the quantised records", w.nQuads);
    if (build_ms) {
        *build_ms = 0.0f;
        if (w.ea && w.eb) (void)hipEventElapsedTime(build_ms, w.ea, w.eb);
    }
    adoptScene(c, w, n, nn);
    if (n_nodes) *n_nodes = (int32_t)nn;
    return PT_OK;
}

// pt_upload_scene: the host's arrays as Scene.CopyToDevice hands them over (Scene.cs:176-183).  They are copied to the GPU as they
// are; the GPU checks the tree (nothing a kernel indexes with may be out of range: child indices, leaf ranges, every pair
// referenced once, the leaves' triangle count, the depth), finds every node's depth by pointer jumping over the parent links,
// and derives the device scene with the kernels pt_build_scene uses.  The host formats the message for the first offending node.
static int uploadOne(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles, int32_t n_tri, const void* nodes, int32_t n_nodes) {
    HIPCHECK(c, hipSetDevice(c->cfg.device));
    HIPCHECK(c, hipStreamSynchronize(c->stream));
    if (n_tri == 0) n_nodes = 0;   // Accelerator.Build on an empty scene leaves a childless root (Accelerator.cs:120-125)
    const uint32_t n = (uint32_t)n_tri, nn = (uint32_t)n_nodes;
    SceneWork w(c);
    if (nn == 0) {
        if (n_mat) {
            TRY(ptMalloc(&w.mats, (size_t)n_mat * 48));
            TRY(hipMemcpyAsync(w.mats, materials, (size_t)n_mat * 48, hipMemcpyHostToDevice, c->stream));
            TRY(hipStreamSynchronize(c->stream));
        }
        w.root.left = -1;
        adoptScene(c, w, 0, 0);
        return PT_OK;
    }
    if ((nn & 1u) == 0u) FAIL(c, PT_E_BAD_BVH, "node count %d is even: nodes are a root plus sibling pairs", n_nodes);
    const HostNode* hn = (const HostNode*)nodes;
    TRY(ptMalloc((void**)&w.d_sorted, (size_t)n * 112));
    TRY(ptMalloc((void**)&w.d_nodes, (size_t)nn * 32));
    TRY(ptMalloc((void**)&w.bn, (size_t)nn * sizeof(ptbuild::BNode)));
    TRY(ptMalloc((void**)&w.d_parent, (size_t)nn * 4));
    TRY(ptMalloc((void**)&w.d_refs, (size_t)(nn / 2 + 1) * 4));
    for (int k = 0; k < 2; k++) { TRY(ptMalloc((void**)&w.d_anc[k], (size_t)nn * 4)); TRY(ptMalloc((void**)&w.d_dep[k], (size_t)nn * 4)); }
    TRY(ptMalloc((void**)&w.d_check, sizeof(ptrepack::UploadCheck)));
    TRY(hipMemcpyAsync(w.d_nodes, nodes, (size_t)nn * 32, hipMemcpyHostToDevice, c->stream));
    TRY(hipMemcpyAsync(w.d_sorted, triangles, (size_t)n * 112, hipMemcpyHostToDevice, c->stream));
    ptrepack::UploadCheck chk{};
    chk.first_error = ~0ull;
    TRY(hipMemcpyAsync(w.d_check, &chk, sizeof(chk), hipMemcpyHostToDevice, c->stream));
    TRY(hipMemsetAsync(w.d_refs, 0, (size_t)(nn / 2 + 1) * 4, c->stream));
    TRY(hipMemsetAsync(w.d_parent, 0, (size_t)nn * 4, c->stream));
    hipLaunchKernelGGL(ptrepack::k_check_nodes, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.d_nodes, nn, n, w.d_parent, w.d_refs, w.d_check);
    if (nn > 1) hipLaunchKernelGGL(ptrepack::k_check_refs, dim3(launchGrid(nn / 2)), dim3(256), 0, c->stream, w.d_refs, nn, w.d_check);
    TRY(hipMemcpyAsync(&chk, w.d_check, sizeof(chk), hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    if (chk.first_error != ~0ull) {
        const int i = (int)(chk.first_error >> 3);
        const HostNode& nd = hn[i];
        switch ((int)(chk.first_error & 7ull)) {
            case ptrepack::UPLOAD_LEAF_RANGE: FAIL(c, PT_E_BAD_BVH, "leaf %d references triangles [%d,%d) of %d", i, nd.leftFirst, nd.leftFirst + nd.count, n_tri);
            case ptrepack::UPLOAD_CHILD_INDEX: FAIL(c, PT_E_BAD_BVH, "interior node %d has children at %d (need odd, > parent, < %d)", i, nd.leftFirst, n_nodes - 1);
            case ptrepack::UPLOAD_UNREFERENCED: FAIL(c, PT_E_BAD_BVH, "node %d is not referenced by an earlier node", i);
            default: FAIL(c, PT_E_BAD_BVH, "node pair %d is referenced twice", i);
        }
    }
    if (chk.leaf_tris != (unsigned long long)n) FAIL(c, PT_E_BAD_BVH, "leaves cover %llu triangles, %d uploaded", chk.leaf_tris, n_tri);
    // depth of every node: seven doublings over the parent links reach any depth the traversal stack allows
    hipLaunchKernelGGL(ptrepack::k_depth_init, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.d_parent, nn, w.d_anc[0], w.d_dep[0]);
    int cur = 0;
    for (int round = 0; round < 7; round++) {
        hipLaunchKernelGGL(ptrepack::k_depth_jump, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.d_anc[cur], w.d_dep[cur], w.d_anc[cur ^ 1], w.d_dep[cur ^ 1], nn);
        cur ^= 1;
    }
    hipLaunchKernelGGL(ptrepack::k_convert, dim3(launchGrid(nn)), dim3(256), 0, c->stream, w.d_nodes, w.d_anc[cur], w.d_dep[cur], nn, w.bn, w.d_check);
    TRY(hipMemcpyAsync(&chk, w.d_check, sizeof(chk), hipMemcpyDeviceToHost, c->stream));
    TRY(hipStreamSynchronize(c->stream));
    const int depth = (int)chk.max_level + 1;
    // the wide traversal pushes up to three entries on each of its (depth + 1) / 2 levels
    if (chk.unresolved || depth > PT_STACK_TOTAL || 3 * ((depth + 1) / 2) > PT_STACK_TOTAL)
        FAIL(c, PT_E_BAD_BVH, "BVH depth %s%d exceeds the traversal stack (%d entries)", chk.unresolved ? "> " : "", chk.unresolved ? 127 : depth, PT_STACK_TOTAL);
    for (void** p : {(void**)&w.d_parent, (void**)&w.d_refs, (void**)&w.d_anc[0], (void**)&w.d_anc[1], (void**)&w.d_dep[0], (void**)&w.d_dep[1], (void**)&w.d_nodes}) {
        (void)hipFree(*p);
        *p = nullptr;
    }
    const int rc = repackOnDevice(c, w, n, nn, materials, n_mat);
    if (rc != PT_OK) return rc;
    TRY(hipGetLastError());
    TRY(hipStreamSynchronize(c->stream));
    if ((size_t)w.nQuads * 128 >= ((size_t)1 << 32) && w.qbad != 0)
        FAIL(c, PT_E_INVALID_ARG, "%u quad records exceed 4 GB and the scene's extent does not admit the quantised records", w.nQuads);
    adoptScene(c, w, n, nn);
    return PT_OK;
}
#undef TRY

static int impl_upload_scene(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles, int32_t n_tri, const void* nodes, int32_t n_nodes) {
    if (!c) return PT_E_INVALID_ARG;
    if (n_mat < 0 || n_tri < 0 || n_nodes < 0 || (n_mat && !materials) || (n_tri && !triangles) || (n_nodes && !nodes)) FAIL(c, PT_E_INVALID_ARG, "pt_upload_scene: bad counts or null arrays");
    if (n_tri > 0 && n_nodes == 0) FAIL(c, PT_E_BAD_BVH, "%d triangles but no nodes", n_tri);
    // several GPUs: each takes its own copy of the arrays and derives its own replica, all at once
    if (c->group) return c->group->all(c, [&](pt_ctx* m) { return uploadOne(m, materials, n_mat, triangles, n_tri, nodes, n_nodes); });
    return uploadOne(c, materials, n_mat, triangles, n_tri, nodes, n_nodes);
}

// The statistics in the library's own (current) struct; pt_get_stats copies as much of it as the caller's struct holds.
static int gatherStats(pt_ctx* c, pt_stats* out);

int pt_get_stats(pt_ctx* c, pt_stats* out) {
    if (!c || !out) return PT_E_INVALID_ARG;
    const uint32_t room = out->struct_size;
    if (room < 16u) FAIL(c, PT_E_INVALID_ARG, "pt_get_stats: pt_stats.struct_size is %u; set it to sizeof(pt_stats) before the call", room);
    pt_stats full{};
    const int rc = gatherStats(c, &full);
    if (rc != PT_OK) return rc;
    const uint32_t n = std::min<uint32_t>(room, (uint32_t)sizeof(pt_stats));
    full.struct_size = n;
    memcpy(out, &full, n);
    return PT_OK;
}

static int gatherStats(pt_ctx* c, pt_stats* out) {
    if (c->group) {
        // ray and step counts add up over the GPUs; times are the slowest GPU's; launch counts are per GPU
        pt_stats total{};
        for (size_t i = 0; i < c->group->members.size(); i++) {
            pt_stats s;
            const int rc = gatherStats(c->group->members[i], &s);
            if (rc != PT_OK) { c->err = c->group->members[i]->err; return rc; }
            if (i == 0) { total = s; continue; }
            total.paths += s.paths; total.extend_rays += s.extend_rays; total.shadow_rays += s.shadow_rays;
            total.extend_steps += s.extend_steps; total.extend_tris += s.extend_tris; total.shadow_steps += s.shadow_steps; total.shadow_tris += s.shadow_tris;
            total.stack_overflows += s.stack_overflows; total.sched_interior += s.sched_interior; total.sched_leaf += s.sched_leaf;
            total.sched_refill += s.sched_refill; total.special_rays += s.special_rays;
            total.extend_unique_steps += s.extend_unique_steps; total.extend_unique_tris += s.extend_unique_tris; total.shadow_occluded += s.shadow_occluded;
            total.ms_generate = std::max(total.ms_generate, s.ms_generate); total.ms_extend = std::max(total.ms_extend, s.ms_extend);
            total.ms_shade = std::max(total.ms_shade, s.ms_shade); total.ms_connect = std::max(total.ms_connect, s.ms_connect);
            total.ms_render = std::max(total.ms_render, s.ms_render); total.ms_extend_primary = std::max(total.ms_extend_primary, s.ms_extend_primary);
        }
        *out = total;
        return PT_OK;
    }
    int rc = pt_synchronize(c);
    if (rc != PT_OK) return rc;
    Ctl ctl;
    HIPCHECK(c, hipMemcpy(&ctl, c->d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    if (c->lane1.d_ctl) {   // calls traced on the second set counted there
        Ctl two;
        HIPCHECK(c, hipMemcpy(&two, c->lane1.d_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        ctl.paths += two.paths; ctl.extend_rays += two.extend_rays; ctl.shadow_rays += two.shadow_rays; ctl.extend_steps += two.extend_steps;
        ctl.extend_tris += two.extend_tris; ctl.shadow_steps += two.shadow_steps; ctl.shadow_tris += two.shadow_tris; ctl.stack_overflows += two.stack_overflows;
        ctl.sched_interior += two.sched_interior; ctl.sched_leaf += two.sched_leaf; ctl.sched_refill += two.sched_refill; ctl.special_rays += two.special_rays;
        ctl.unique_steps += two.unique_steps; ctl.unique_tris += two.unique_tris; ctl.shadow_occluded += two.shadow_occluded;
    }
    pt_stats s = c->host_stats;
    s.paths = ctl.paths;
    s.extend_rays = ctl.extend_rays;
    s.shadow_rays = ctl.shadow_rays;
    s.extend_steps = ctl.extend_steps;
    s.extend_tris = ctl.extend_tris;
    s.shadow_steps = ctl.shadow_steps;
    s.shadow_tris = ctl.shadow_tris;
    s.stack_overflows = ctl.stack_overflows;
    s.sched_interior = ctl.sched_interior;
    s.sched_leaf = ctl.sched_leaf;
    s.sched_refill = ctl.sched_refill;
    s.special_rays = ctl.special_rays;
    s.extend_unique_steps = ctl.unique_steps;
    s.extend_unique_tris = ctl.unique_tris;
    s.shadow_occluded = ctl.shadow_occluded;
    s.device_allocations = ptAllocCount().load(std::memory_order_relaxed);
    *out = s;
    return PT_OK;
}

int pt_reset_stats(pt_ctx* c) {
    if (!c) return PT_E_INVALID_ARG;
    if (c->group) {
        for (pt_ctx* m : c->group->members) {
            const int rc = pt_reset_stats(m);
            if (rc != PT_OK) { c->err = m->err; return rc; }
        }
        return PT_OK;
    }
    int rc = pt_synchronize(c);
    if (rc != PT_OK) return rc;
    HIPCHECK(c, hipMemsetAsync(c->d_ctl, 0, sizeof(Ctl), c->stream));
    if (c->lane1.d_ctl) HIPCHECK(c, hipMemsetAsync(c->lane1.d_ctl, 0, sizeof(Ctl), c->stream));
    HIPCHECK(c, hipStreamSynchronize(c->stream));   // (the control blocks are used from other streams next)
    c->host_stats = pt_stats{};
    return PT_OK;
}

// ---- the exported entry points that allocate on the host, each behind the exception barrier
int pt_create(const pt_config* cfg, pt_ctx** out) { return guarded((pt_ctx*)nullptr, "pt_create", [&]() -> int { return impl_create(cfg, out); }); }
int pt_upload_scene(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles, int32_t n_tri, const void* nodes, int32_t n_nodes) { return guarded(c, "pt_upload_scene", [&]() -> int { return impl_upload_scene(c, materials, n_mat, triangles, n_tri, nodes, n_nodes); }); }
int pt_resize(pt_ctx* c, int32_t width, int32_t height) { return guarded(c, "pt_resize", [&]() -> int { return impl_resize(c, width, height); }); }
int pt_render(pt_ctx* c, const void* frame_state112, int32_t spp, int32_t max_bounces) { return guarded(c, "pt_render", [&]() -> int { return impl_render(c, frame_state112, spp, max_bounces); }); }
int pt_read_image(pt_ctx* c, float* rgba) { return guarded(c, "pt_read_image", [&]() -> int { return impl_read_image(c, rgba); }); }
int pt_assemble_tiles(pt_ctx* c, const void* gathered_device, int32_t world, uint32_t samples, float* rgba_host) { return guarded(c, "pt_assemble_tiles", [&]() -> int { return impl_assemble_tiles(c, gathered_device, world, samples, rgba_host); }); }
int pt_trace_closest(pt_ctx* c, const float* o_d, int32_t n, int32_t* tri, float* t, float* uv) { return guarded(c, "pt_trace_closest", [&]() -> int { return impl_trace_closest(c, o_d, n, tri, t, uv); }); }
int pt_trace_occluded(pt_ctx* c, const float* o_d, const float* tmax, int32_t n, uint8_t* occluded) { return guarded(c, "pt_trace_occluded", [&]() -> int { return impl_trace_occluded(c, o_d, tmax, n, occluded); }); }
int pt_generate_rays(pt_ctx* c, const void* frame_state112, int32_t n, float* o_d, uint32_t* pixel) { return guarded(c, "pt_generate_rays", [&]() -> int { return impl_generate_rays(c, frame_state112, n, o_d, pixel); }); }
int pt_bench_extend(pt_ctx* c, const float* o_d, int32_t n, int32_t iters, float* avg_ms, int32_t* tri, float* t) { return guarded(c, "pt_bench_extend", [&]() -> int { return impl_bench_extend(c, o_d, n, iters, avg_ms, tri, t); }); }
int pt_build_bvh(pt_ctx* c, const void* triangles_in, int32_t n_tri, void* triangles_out, void* nodes_out, int32_t node_capacity, int32_t* n_nodes, float* build_ms) { return guarded(c, "pt_build_bvh", [&]() -> int { return impl_build_bvh(c, triangles_in, n_tri, triangles_out, nodes_out, node_capacity, n_nodes, build_ms); }); }
int pt_build_scene_instanced(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles_in, int32_t n_tri, const void* mesh_triangles, const int32_t* mesh_first,
                             const int32_t* mesh_count, int32_t n_meshes, const pt_instance* instances, int32_t n_instances, float* build_ms) {
    if (!c) return PT_E_INVALID_ARG;
    if (!mesh_triangles || !mesh_first || !mesh_count || n_meshes < 1 || !instances || n_instances < 1) {
        c->err = "pt_build_scene_instanced: null or empty mesh / instance arrays";
        return PT_E_INVALID_ARG;
    }
    return guarded(c, "pt_build_scene_instanced", [&]() -> int {
        const InstancedInput in{mesh_triangles, mesh_first, mesh_count, n_meshes, instances, n_instances};
        return impl_build_scene(c, materials, n_mat, triangles_in, n_tri, nullptr, nullptr, 0, nullptr, build_ms, &in);
    });
}
int pt_build_scene(pt_ctx* c, const void* materials, int32_t n_mat, const void* triangles_in, int32_t n_tri, void* triangles_out, void* nodes_out, int32_t node_capacity, int32_t* n_nodes, float* build_ms) { return guarded(c, "pt_build_scene", [&]() -> int { return impl_build_scene(c, materials, n_mat, triangles_in, n_tri, triangles_out, nodes_out, node_capacity, n_nodes, build_ms); }); }

}  // extern "C"
