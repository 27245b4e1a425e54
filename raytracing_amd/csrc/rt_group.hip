// Several GPUs behind the C ABI (include/raytracing_amd.h, "several GPUs"): the samples of every pixel are split over
// the members of a group by global sample index, every member renders its share with its own rt_context, and the
// fixed-point sums are added by one RCCL all-reduce per frame on the streams the kernels ran on.  Host code only.
//
// The reference's counterpart is prose: "copy the code N times with different seeds onto N cores, then average the
// images" ("Ray Tracing In a Weekend/README.md":2538-2542).
//
// RCCL is bound at run time (dlopen of librccl.so.1, the ROCm 7 soname; a process that already holds a copy -- PyTorch
// ships one -- gets that copy), so the single-GPU entry points neither load nor need it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "raytracing_amd.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <class F>
bool bind(void* h, const char* name, F& fn, std::string& err) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    if (!fn) err = std::string("librccl lacks ") + name;
    return fn != nullptr;
}

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& R = g_rccl;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
        }
        if (!R.handle) {
            const char* e = dlerror();
            R.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown dlopen error");
            return;
        }
        const bool ok = bind(R.handle, "ncclGetUniqueId", R.GetUniqueId, R.error) && bind(R.handle, "ncclCommInitRank", R.CommInitRank, R.error) &&
                        bind(R.handle, "ncclCommInitAll", R.CommInitAll, R.error) && bind(R.handle, "ncclCommDestroy", R.CommDestroy, R.error) &&
                        bind(R.handle, "ncclAllReduce", R.AllReduce, R.error) && bind(R.handle, "ncclGroupStart", R.GroupStart, R.error) &&
                        bind(R.handle, "ncclGroupEnd", R.GroupEnd, R.error) && bind(R.handle, "ncclGetErrorString", R.GetErrorString, R.error);
        if (!ok) R.handle = nullptr;
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

// The buffer of a slot holds the frame's 3 W H sums and, behind them, one more word that travels through the same
// all-reduce: the number of members, over all processes, that could not enqueue their share of the frame.  Every rank
// therefore learns from its own copy whether the summed frame is whole.
struct Slot {
    hipStream_t stream = nullptr;
    rt_context* ctx = nullptr;
    int64_t* d_accum = nullptr;
    size_t elems = 0;           // capacity in sums (the buffer holds elems + 1 words)
    int64_t* h_failed = nullptr;  // pinned: the frame's failure word, copied behind the all-reduce
    int width = 0, height = 0;  // of the frame the slot holds
};
struct Member {
    int device = -1;
    ncclComm_t comm = nullptr;
    std::vector<Slot> slots;
    // Collectives of one communicator must run in the order they were issued, on every rank.  The slots of a member
    // share its communicator but have streams of their own, so the order is made explicit: every all-reduce records
    // this event behind itself, and the next one -- whichever slot's stream it is on -- waits for it first.
    hipEvent_t last_collective = nullptr;
    bool collective_pending = false;
    bool fail_next = false;  // validation: rt_group_set_fail_next
};

thread_local std::string g_group_create_detail;

}  // namespace

struct rt_group {
    int n_ranks = 0;     // members over all processes
    int first_rank = 0;  // rank of local member 0; local member i is rank first_rank + i
    std::vector<Member> members;
    int frame_slots = 1;
    std::string detail;
};

namespace {

int gfail(rt_group* g, int code, const std::string& what) {
    if (g) g->detail = what;
    else g_group_create_detail = what;
    return code;
}
int gfail_hip(rt_group* g, hipError_t e, const char* where) { return gfail(g, RT_ERR_HIP, std::string(where) + ": " + hipGetErrorString(e)); }
int gfail_nccl(rt_group* g, ncclResult_t r, const char* where) {
    const Rccl* R = rccl();
    return gfail(g, RT_ERR_COMM, std::string(where) + ": " + (R ? R->GetErrorString(r) : "RCCL not loaded"));
}
// failure of a member's context: carry its detail over
int gfail_ctx(rt_group* g, int code, rt_context* ctx, const char* where) {
    return gfail(g, code, std::string(where) + ": " + rt_last_error_detail(ctx));
}
#define G_HIP(g, expr)                                           \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return gfail_hip((g), e_, #expr);  \
    } while (0)
#define G_NCCL(g, expr)                                            \
    do {                                                           \
        ncclResult_t r_ = (expr);                                  \
        if (r_ != ncclSuccess) return gfail_nccl((g), r_, #expr);  \
    } while (0)

// Streams, contexts (and later buffers) of the members named in g->members[*].device.
int make_members(rt_group* g) {
    for (Member& m : g->members) {
        G_HIP(nullptr, hipSetDevice(m.device));
        m.slots.resize((size_t)g->frame_slots);
        G_HIP(nullptr, hipEventCreateWithFlags(&m.last_collective, hipEventDisableTiming));
        for (Slot& s : m.slots) {
            G_HIP(nullptr, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            G_HIP(nullptr, hipHostMalloc((void**)&s.h_failed, sizeof(int64_t), hipHostMallocDefault));
            *s.h_failed = 0;
            const int rc = rt_context_create(m.device, (void*)s.stream, &s.ctx);
            if (rc != RT_OK) return gfail(nullptr, rc, std::string("rt_context_create: ") + rt_last_error_detail(nullptr));
        }
    }
    return RT_OK;
}

bool slot_ok(const rt_group* g, int slot) { return slot >= 0 && slot < g->frame_slots; }

// Room for a frame of `elems` sums in the slot's buffer of every local member.  Waits for the slot's stream before a
// buffer is replaced.
int reserve_slot(rt_group* g, int slot, size_t elems) {
    for (Member& m : g->members) {
        Slot& s = m.slots[(size_t)slot];
        if (s.elems >= elems) continue;
        G_HIP(g, hipSetDevice(m.device));
        G_HIP(g, hipStreamSynchronize(s.stream));
        (void)hipFree(s.d_accum);
        s.d_accum = nullptr;
        s.elems = 0;
        s.width = s.height = 0;
        G_HIP(g, hipMalloc((void**)&s.d_accum, (elems + 1) * sizeof(int64_t)));
        s.elems = elems;
    }
    return RT_OK;
}

}  // namespace

extern "C" {

void rt_split_samples(int sample_count, int index, int n, int* begin, int* count) {
    int b = 0, c = 0;
    if (n >= 1 && index >= 0 && index < n && sample_count >= 0) {
        const int base = sample_count / n, extra = sample_count % n;
        c = base + (index < extra ? 1 : 0);
        b = index * base + (index < extra ? index : extra);
    }
    if (begin) *begin = b;
    if (count) *count = c;
}

const char* rt_group_last_error_detail(const rt_group* g) { return g ? g->detail.c_str() : g_group_create_detail.c_str(); }

int rt_comm_id_create(rt_comm_id* out) {
    static_assert(sizeof(ncclUniqueId) == RT_COMM_ID_BYTES, "rt_comm_id carries an ncclUniqueId");
    if (!out) return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "out is null");
    const Rccl* R = rccl();
    if (!R) return gfail(nullptr, RT_ERR_COMM, g_rccl.error);
    ncclUniqueId id;
    G_NCCL(nullptr, R->GetUniqueId(&id));
    std::memcpy(out->bytes, &id, sizeof(id));
    return RT_OK;
}

void rt_group_destroy(rt_group* g) {
    if (!g) return;
    const Rccl* R = rccl();
    for (Member& m : g->members) {
        (void)hipSetDevice(m.device);
        for (Slot& s : m.slots)
            if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    for (Member& m : g->members) {
        (void)hipSetDevice(m.device);
        if (m.comm && R) (void)R->CommDestroy(m.comm);
        for (Slot& s : m.slots) {
            if (s.ctx) rt_context_destroy(s.ctx);  // waits for the stream, which the context does not own
            (void)hipFree(s.d_accum);
            if (s.h_failed) (void)hipHostFree(s.h_failed);
            if (s.stream) (void)hipStreamDestroy(s.stream);
        }
        if (m.last_collective) (void)hipEventDestroy(m.last_collective);
    }
    delete g;
}

static int group_alloc(const int* devices, int n_local, int n_ranks, int first_rank, int frame_slots, rt_group** out) {
    if (!out) return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (!devices || n_local < 1 || n_ranks < n_local || first_rank < 0 || first_rank + n_local > n_ranks)
        return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "bad member list or rank");
    if (frame_slots < 1 || frame_slots > 8) return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "frame_slots must be 1..8");
    const int visible = rt_get_device_count();
    for (int i = 0; i < n_local; ++i) {
        if (devices[i] < 0 || devices[i] >= visible)
            return gfail(nullptr, RT_ERR_NO_DEVICE, "device ordinal " + std::to_string(devices[i]) + " of " + std::to_string(visible));
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "a device is named twice (RCCL wants one rank per GPU)");
    }
    if (!rccl()) return gfail(nullptr, RT_ERR_COMM, g_rccl.error);
    rt_group* g = new (std::nothrow) rt_group();
    if (!g) return gfail(nullptr, RT_ERR_OUT_OF_MEMORY, "group");
    g->n_ranks = n_ranks;
    g->first_rank = first_rank;
    g->frame_slots = frame_slots;
    g->members.resize((size_t)n_local);
    for (int i = 0; i < n_local; ++i) g->members[(size_t)i].device = devices[i];
    const int rc = make_members(g);
    if (rc != RT_OK) {
        rt_group_destroy(g);
        return rc;
    }
    *out = g;
    return RT_OK;
}

int rt_group_create(const int* device_ordinals, int n, int frame_slots, rt_group** out) {
    int rc = group_alloc(device_ordinals, n, n, 0, frame_slots, out);
    if (rc != RT_OK) return rc;
    rt_group* g = *out;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    const ncclResult_t r = rccl()->CommInitAll(comms.data(), n, device_ordinals);
    if (r != ncclSuccess) {
        rc = gfail_nccl(nullptr, r, "ncclCommInitAll");
        rt_group_destroy(g);
        *out = nullptr;
        return rc;
    }
    for (int i = 0; i < n; ++i) g->members[(size_t)i].comm = comms[(size_t)i];
    return RT_OK;
}

int rt_group_create_rank(int device_ordinal, int rank, int n_ranks, const rt_comm_id* id, int frame_slots, rt_group** out) {
    if (!id) return gfail(nullptr, RT_ERR_INVALID_ARGUMENT, "id is null");
    int rc = group_alloc(&device_ordinal, 1, n_ranks, rank, frame_slots, out);
    if (rc != RT_OK) return rc;
    rt_group* g = *out;
    ncclUniqueId nid;
    std::memcpy(&nid, id->bytes, sizeof(nid));
    hipError_t e = hipSetDevice(device_ordinal);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = rccl()->CommInitRank(&g->members[0].comm, n_ranks, nid, rank);
    if (e != hipSuccess || r != ncclSuccess) {
        rc = e != hipSuccess ? gfail_hip(nullptr, e, "hipSetDevice") : gfail_nccl(nullptr, r, "ncclCommInitRank");
        rt_group_destroy(g);
        *out = nullptr;
        return rc;
    }
    return RT_OK;
}

int rt_group_size(const rt_group* g) { return g ? g->n_ranks : 0; }
int rt_group_local_count(const rt_group* g) { return g ? (int)g->members.size() : 0; }
int rt_group_frame_slots(const rt_group* g) { return g ? g->frame_slots : 0; }

rt_context* rt_group_context(rt_group* g, int local_index, int slot) {
    if (!g || local_index < 0 || local_index >= (int)g->members.size() || !slot_ok(g, slot)) return nullptr;
    return g->members[(size_t)local_index].slots[(size_t)slot].ctx;
}

int rt_group_scene_upload(rt_group* g, const rt_scene_desc* scene) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    for (Member& m : g->members)
        for (Slot& s : m.slots) {
            const int rc = rt_scene_upload(s.ctx, scene);
            if (rc != RT_OK) return gfail_ctx(g, rc, s.ctx, "rt_scene_upload");
        }
    return RT_OK;
}

int rt_group_reserve(rt_group* g, int width, int height) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (width < 2 || height < 2 || (int64_t)width * height > (int64_t)1 << 28) return gfail(g, RT_ERR_INVALID_ARGUMENT, "bad frame size");
    for (int slot = 0; slot < g->frame_slots; ++slot) {
        const int rc = reserve_slot(g, slot, (size_t)3 * width * height);
        if (rc != RT_OK) return rc;
    }
    return RT_OK;
}

int rt_group_set_fail_next(rt_group* g, int local_index) {
    if (!g || local_index < 0 || local_index >= (int)g->members.size()) return RT_ERR_INVALID_ARGUMENT;
    g->members[(size_t)local_index].fail_next = true;
    return RT_OK;
}

// A frame in three steps, so that a failure never leaves a peer waiting in the all-reduce:
//   1. everything that can be refused is checked, and every buffer grown, for ALL local members before anything is
//      enqueued: a call that returns an error from this step has enqueued nothing and joined nothing, and because every
//      process of the group makes the same call with the same arguments on the same world, it fails on all of them;
//   2. every member zeroes its buffer and enqueues its kernel.  A member whose enqueue fails now (a HIP error) is
//      remembered, its share is missing, and the frame's failure word in its buffer is set;
//   3. EVERY member joins the all-reduce regardless, the failure word travels with the sums, and each slot copies its
//      word to the host behind the collective.  rt_group_wait reports a frame whose word is not zero on every rank.
int rt_group_render_async(rt_group* g, const rt_camera_desc* cam, const rt_render_params* p, int slot) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || !p || !slot_ok(g, slot)) return gfail(g, RT_ERR_INVALID_ARGUMENT, "null camera or params, or bad slot");
    if (p->width < 2 || p->height < 2 || (int64_t)p->width * p->height > (int64_t)1 << 28 || p->sample_count < 0)
        return gfail(g, RT_ERR_INVALID_ARGUMENT, "bad frame size or sample count");
    const Rccl* R = rccl();
    const size_t elems = (size_t)3 * p->width * p->height;
    // ---- step 1
    std::vector<rt_render_params> share(g->members.size(), *p);
    for (size_t i = 0; i < g->members.size(); ++i) {
        int begin = 0, count = 0;
        rt_split_samples(p->sample_count, g->first_rank + (int)i, g->n_ranks, &begin, &count);
        share[i].sample_begin = p->sample_begin + begin;
        share[i].sample_count = count;
        rt_context* ctx = g->members[i].slots[(size_t)slot].ctx;
        // (the whole frame's range is checked as well: a member with fewer samples must not pass where another fails)
        int rc = rt_render_check(ctx, cam, p);
        if (rc == RT_OK) rc = rt_render_check(ctx, cam, &share[i]);
        if (rc != RT_OK) return gfail_ctx(g, rc, ctx, "rt_render_check");
    }
    {
        const int rc = reserve_slot(g, slot, elems);
        if (rc != RT_OK) return rc;
    }
    // ---- step 2
    int first_rc = RT_OK;
    std::string first_detail;
    auto remember = [&](int rc, const std::string& what) {
        if (first_rc == RT_OK) {
            first_rc = rc;
            first_detail = what;
        }
    };
    for (size_t i = 0; i < g->members.size(); ++i) {
        Member& m = g->members[i];
        Slot& s = m.slots[(size_t)slot];
        s.width = s.height = 0;  // the slot holds no frame until this one is enqueued in full
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipMemsetAsync(s.d_accum, 0, (elems + 1) * sizeof(int64_t), s.stream);
        int rc = e == hipSuccess ? RT_OK : RT_ERR_HIP;
        if (rc != RT_OK) remember(rc, std::string("zeroing the frame: ") + hipGetErrorString(e));
        if (rc == RT_OK && m.fail_next) {
            m.fail_next = false;
            rc = RT_ERR_HIP;
            remember(rc, "member " + std::to_string(g->first_rank + (int)i) + ": failure requested by rt_group_set_fail_next");
        } else if (rc == RT_OK) {
            rc = rt_render_device(s.ctx, cam, &share[i], s.d_accum);  // a member with no samples enqueues nothing
            if (rc != RT_OK) remember(rc, std::string("rt_render_device: ") + rt_last_error_detail(s.ctx));
        }
        if (rc != RT_OK)  // the low byte of the failure word becomes 1 (behind the zeroing, on the same stream)
            (void)hipMemsetAsync((unsigned char*)(s.d_accum + elems), 1, 1, s.stream);
    }
    // ---- step 3: the one exchange of the frame -- every member's buffer becomes the sum, on the stream its kernel ran on
    for (Member& m : g->members) {
        if (!m.collective_pending) continue;
        (void)hipSetDevice(m.device);
        const hipError_t e = hipStreamWaitEvent(m.slots[(size_t)slot].stream, m.last_collective, 0);
        if (e != hipSuccess) remember(RT_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e));
    }
    ncclResult_t first_bad = R->GroupStart();
    for (Member& m : g->members) {
        Slot& s = m.slots[(size_t)slot];
        const ncclResult_t r = R->AllReduce(s.d_accum, s.d_accum, elems + 1, ncclInt64, ncclSum, m.comm, s.stream);
        if (r != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
    }
    const ncclResult_t end = R->GroupEnd();
    if (first_bad == ncclSuccess) first_bad = end;
    for (Member& m : g->members) {
        Slot& s = m.slots[(size_t)slot];
        (void)hipSetDevice(m.device);
        hipError_t e = hipEventRecord(m.last_collective, s.stream);
        m.collective_pending = e == hipSuccess;
        if (e == hipSuccess) e = hipMemcpyAsync(s.h_failed, s.d_accum + elems, sizeof(int64_t), hipMemcpyDeviceToHost, s.stream);
        if (e != hipSuccess) remember(RT_ERR_HIP, std::string("behind the all-reduce: ") + hipGetErrorString(e));
    }
    if (first_bad != ncclSuccess) return gfail_nccl(g, first_bad, "ncclAllReduce (the communicator is not usable any more)");
    if (first_rc != RT_OK) return gfail(g, first_rc, first_detail + " (the collective was joined; the frame is incomplete on every rank)");
    for (Member& m : g->members) {
        m.slots[(size_t)slot].width = p->width;
        m.slots[(size_t)slot].height = p->height;
    }
    return RT_OK;
}

int rt_group_wait(rt_group* g, int slot) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (!slot_ok(g, slot)) return gfail(g, RT_ERR_INVALID_ARGUMENT, "bad slot");
    int64_t failed = 0;
    for (Member& m : g->members) {
        Slot& s = m.slots[(size_t)slot];
        G_HIP(g, hipSetDevice(m.device));
        G_HIP(g, hipStreamSynchronize(s.stream));
        if (*s.h_failed != 0) {  // (of the last frame this slot was given; the member that failed reports it as well)
            failed = *s.h_failed;
            s.width = s.height = 0;  // the slot holds no frame
        }
    }
    if (failed != 0)
        return gfail(g, RT_ERR_COMM, std::to_string(failed) + " member(s) of the group could not render their share: the summed frame is incomplete");
    return RT_OK;
}

int rt_group_read(rt_group* g, int slot, double* accum_rgb_out) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (!slot_ok(g, slot) || !accum_rgb_out) return gfail(g, RT_ERR_INVALID_ARGUMENT, "bad slot or null output");
    Slot& s = g->members[0].slots[(size_t)slot];
    if (!s.d_accum || s.width == 0) return gfail(g, RT_ERR_INVALID_ARGUMENT, "the slot holds no frame");
    const int rc = rt_read_sums(s.ctx, s.d_accum, s.width, s.height, accum_rgb_out);
    if (rc != RT_OK) return gfail_ctx(g, rc, s.ctx, "rt_read_sums");
    return RT_OK;
}

const int64_t* rt_group_device_sums(rt_group* g, int local_index, int slot) {
    if (!g || local_index < 0 || local_index >= (int)g->members.size() || !slot_ok(g, slot)) return nullptr;
    return g->members[(size_t)local_index].slots[(size_t)slot].d_accum;
}

int rt_group_take_stats(rt_group* g, rt_stats* out) {
    if (!g || !out) return RT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    for (Member& m : g->members)
        for (Slot& s : m.slots) {
            rt_stats one;
            const int rc = rt_context_take_stats(s.ctx, &one);
            if (rc != RT_OK) return gfail_ctx(g, rc, s.ctx, "rt_context_take_stats");
            out->samples += one.samples;
            out->rays += one.rays;
            out->nan_samples += one.nan_samples;
            out->overflow_rays += one.overflow_rays;
            out->kernel_launches += one.kernel_launches;
            out->kernel_ms += one.kernel_ms;
        }
    return RT_OK;
}

int rt_group_render(rt_group* g, const rt_camera_desc* cam, const rt_render_params* p, double* accum_rgb_out, rt_stats* stats) {
    if (!g) return RT_ERR_INVALID_ARGUMENT;
    if (!accum_rgb_out) return gfail(g, RT_ERR_INVALID_ARGUMENT, "null output");
    int rc = rt_group_render_async(g, cam, p, 0);
    if (rc != RT_OK) return rc;
    if ((rc = rt_group_wait(g, 0)) != RT_OK) return rc;
    if ((rc = rt_group_read(g, 0, accum_rgb_out)) != RT_OK) return rc;
    if (stats) rc = rt_group_take_stats(g, stats);
    return rc;
}

}  // extern "C"
