// Error reporting, version and the event-based kernel profiler of libresdepth_hip.so.
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "rd_common.h"

namespace rd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return RD_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return RD_ERR_HIP;
}

// ---- tuning / diagnosis knobs -----------------------------------------------------------
struct TuneEntry {
    const char* name;
    int value;
};
static TuneEntry g_tune[TUNE_COUNT] = {
    {"mfma_f32", 0},   {"nt_tile", -1},      {"nt_halo", -1},    {"nt_skew", 1},     {"tn_tile", -1},     {"tn_blocks", 512}, {"tn_split", -1},
    {"wg_strip", -1},  {"wg_minblocks", 768}, {"wg_blocks", 512}, {"wg_occ", 2}, {"convt_patch", -1}, {"edge_conv", -1}, {"rows_blocks", 512}, {"last_blocks", 2048},
    {"nt_splitk", -1}, {"nt_epi", -1}, {"mfma_products", 3}, {"d2h_blocks", 0},
};
static int tune_index(const char* name, size_t len) {
    for (int i = 0; i < TUNE_COUNT; ++i)
        if (strlen(g_tune[i].name) == len && !strncmp(g_tune[i].name, name, len)) return i;
    return -1;
}
static const int g_tune_env = [] {      // RD_TUNE="name=value,..." and the RD_MFMA=f32 mode switch, read once at load time
    if (const char* m = getenv("RD_MFMA"))
    {
        if (!strcmp(m, "f32")) g_tune[TUNE_MFMA_F32].value = 1;
        if (!strcmp(m, "split2h")) g_tune[TUNE_MFMA_PRODUCTS].value = 3;
        if (!strcmp(m, "split3")) g_tune[TUNE_MFMA_PRODUCTS].value = 6;
    }
    const char* e = getenv("RD_TUNE");
    while (e && *e) {
        const char* eq = strchr(e, '=');
        if (!eq) break;
        const int i = tune_index(e, (size_t)(eq - e));
        if (i >= 0) g_tune[i].value = atoi(eq + 1);
        e = strchr(eq, ',');
        if (e) ++e;
    }
    return 0;
}();
int tune(int key) { return g_tune[key].value; }

// ---- magnitude slots of the next call (rd_quant_next) ---------------------------------------------------------------
static thread_local QuantArgs t_quant = {nullptr, nullptr, nullptr, nullptr, 0};
QuantArgs quant_take_img() {
    const QuantArgs q = t_quant;
    t_quant = {nullptr, nullptr, nullptr, nullptr, 0};
    return q;
}
QuantArgs quant_take() {
    QuantArgs q = t_quant;
    t_quant = {nullptr, nullptr, nullptr, nullptr, 0};
    if (q.img_stride) q = {nullptr, nullptr, nullptr, nullptr, 0};
    return q;
}

// ---- launch plans (rd_plan_*) ----------------------------------------------------------------------------------------------
// One recording at a time per process (the forward runs on the caller's thread, the backward on an autograd worker: the
// recorder is keyed by STREAM, not by thread).  A plan = a list of operations in host enqueue order: kernel launches with their
// argument values, event records / waits between the plan's two streams, and segment ends (where the host acts between
// replays: a collective, a read-back).
struct PlanOp {
    int kind;                 // 0 launch | 1 event record | 2 event wait
    int role;                 // stream of the op: 0 = main, 1 = side
    const void* fn;
    dim3 grid, block;
    unsigned shmem;
    int nargs;
    size_t arg0;              // index of the first argument offset in Plan::arg_off
    int ev;                   // event index (kinds 1, 2)
};
struct Plan {
    std::vector<PlanOp> ops;
    std::vector<size_t> arg_off;      // byte offsets into blob
    std::vector<char> blob;           // argument values, each at its natural alignment
    std::vector<hipEvent_t> events;
    std::vector<size_t> seg_end;      // ops index one past each segment
    hipStream_t rec[2] = {nullptr, nullptr};
    int n_events = 0, n_launches = 0, dev = -1;
    bool poisoned = false;
    std::string why;
};
static std::mutex g_plan_mu;
static Plan* g_rec = nullptr;             // the plan being recorded (nullptr: none)
static volatile int g_recording = 0;

bool plan_recording() { return g_recording != 0; }

static int plan_role(const Plan* p, hipStream_t s) { return s == p->rec[0] ? 0 : s == p->rec[1] ? 1 : -1; }

void plan_note_launch(const void* fn, dim3 grid, dim3 block, size_t shmem, hipStream_t s, void** args, const size_t* sizes,
                      const size_t* aligns, int nargs) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    Plan* p = g_rec;
    if (!p) return;
    const int role = plan_role(p, s);
    if (role < 0) return;                 // another stream's work (a prefetcher, another model): not part of this plan
    if (nargs > 64) {
        p->poisoned = true;
        p->why = "a kernel with more than 64 arguments";
        return;
    }
    PlanOp op = {};
    op.kind = 0; op.role = role; op.fn = fn; op.grid = grid; op.block = block; op.shmem = (unsigned)shmem; op.nargs = nargs;
    op.arg0 = p->arg_off.size();
    for (int i = 0; i < nargs; ++i) {
        size_t off = (p->blob.size() + aligns[i] - 1) / aligns[i] * aligns[i];
        p->blob.resize(off + sizes[i]);
        memcpy(p->blob.data() + off, args[i], sizes[i]);
        p->arg_off.push_back(off);
    }
    p->ops.push_back(op);
    p->n_launches++;
}

void plan_poison(hipStream_t s, const char* why) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_rec && plan_role(g_rec, s) >= 0 && !g_rec->poisoned) {
        g_rec->poisoned = true;
        g_rec->why = why;
    }
}

// ---- split-K scratch (rd_set_splitk_workspace) --------------------------------------------------------------------
// One registration per (device, HIP stream): [64 KB of tile tickets, zeroed here once; the kernels leave them zero]
// [partial-sum slabs].  The device is part of the key because a stream HANDLE does not name a device: the null stream (torch's
// default stream) is handle 0 on every device of a process.
struct SkEntry {
    int dev;
    hipStream_t s;
    char* ws;
    size_t bytes;
};
static std::mutex g_sk_mu;
static std::vector<SkEntry> g_sk;
constexpr size_t kSkTicketBytes = 64 << 10;

bool splitk_workspace(hipStream_t s, unsigned** tickets, int* n_tickets, float** slab, size_t* slab_bytes) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;      // the launch that follows goes to the current device
    std::lock_guard<std::mutex> lk(g_sk_mu);
    for (const SkEntry& e : g_sk)
        if (e.dev == dev && e.s == s) {
            *tickets = reinterpret_cast<unsigned*>(e.ws);
            *n_tickets = (int)(kSkTicketBytes / sizeof(unsigned));
            *slab = reinterpret_cast<float*>(e.ws + kSkTicketBytes);
            *slab_bytes = e.bytes - kSkTicketBytes;
            return true;
        }
    return false;
}

// ---- profiler ---------------------------------------------------------------------------
struct ProfRec {
    int cls;
    hipEvent_t e0, e1;
    double flops, bytes;
};
struct ProfClass {
    std::string name;
    long long launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
static std::mutex g_mu;
static volatile int g_level = 0;
static std::vector<ProfRec> g_recs;
static std::vector<ProfClass> g_cls;
static std::vector<hipEvent_t> g_free_events;
static thread_local int t_open = -1;

int prof_level() { return g_level; }

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void prof_begin(hipStream_t s, const char* cls, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    int ci = -1;
    for (size_t i = 0; i < g_cls.size(); ++i)
        if (g_cls[i].name == cls) ci = (int)i;
    if (ci < 0) {
        if (g_cls.size() >= RD_PROF_MAX_CLASSES) return;
        g_cls.push_back(ProfClass());
        g_cls.back().name = cls;
        ci = (int)g_cls.size() - 1;
    }
    ProfRec r;
    r.cls = ci;
    r.e0 = get_event();
    r.e1 = get_event();
    r.flops = flops;
    r.bytes = bytes;
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    t_open = (int)g_recs.size() - 1;
}

void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (t_open < 0 || t_open >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[t_open].e1, s);
    t_open = -1;
}

static void drain_locked() {
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            ProfClass& c = g_cls[r.cls];
            c.launches += 1;
            c.ms += ms;
            c.flops += r.flops;
            c.bytes += r.bytes;
        }
        g_free_events.push_back(r.e0);
        g_free_events.push_back(r.e1);
    }
    g_recs.clear();
}

}  // namespace rd

extern "C" {

int rd_version(void) { return 107; }      // 101 (r04): rd_set_splitk_workspace keyed by (device, stream); 102 (r05): rd_host_register; 103: rd_mfma_products; 104: rd_adam_step_dev; 105 (r06): rd_quant_next, rd_amax, packed operands carry both split forms; 106: rd_plan_*; 107: rd_quant_next_img (per-image magnitude slots)

const char* rd_last_error_string(void) { return rd::g_err; }

// ---- launch plans -----------------------------------------------------------------------------------------------------------
int rd_plan_begin(rd_stream_t main_stream, rd_stream_t side_stream) {
    using namespace rd;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_rec) {
        set_error("rd_plan_begin: a plan is already being recorded");
        return RD_ERR_ARG;
    }
    Plan* p = new Plan();
    p->rec[0] = (hipStream_t)main_stream;
    p->rec[1] = (hipStream_t)side_stream;
    (void)hipGetDevice(&p->dev);
    g_rec = p;
    g_recording = 1;
    return RD_OK;
}

int rd_plan_event_record(rd_stream_t stream) {
    using namespace rd;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!g_rec) return -1;
    const int role = plan_role(g_rec, (hipStream_t)stream);
    if (role < 0) return -1;
    PlanOp op = {};
    op.kind = 1; op.role = role; op.ev = g_rec->n_events++;
    g_rec->ops.push_back(op);
    return op.ev;
}

int rd_plan_event_wait(rd_stream_t stream, int ev) {
    using namespace rd;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!g_rec) return RD_OK;
    const int role = plan_role(g_rec, (hipStream_t)stream);
    if (role < 0 || ev < 0) {
        // a wait on something the plan did not record (an event of a stream outside it): the replay could not reproduce it
        if (role >= 0 && !g_rec->poisoned) {
            g_rec->poisoned = true;
            g_rec->why = "a plan stream waited for an event recorded outside the plan";
        }
        return RD_OK;
    }
    PlanOp op = {};
    op.kind = 2; op.role = role; op.ev = ev;
    g_rec->ops.push_back(op);
    return RD_OK;
}

int rd_plan_segment(void) {
    using namespace rd;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (!g_rec) return -1;
    g_rec->seg_end.push_back(g_rec->ops.size());
    return (int)g_rec->seg_end.size() - 1;
}

int rd_plan_poison(const char* why) {
    using namespace rd;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    if (g_rec && !g_rec->poisoned) {
        g_rec->poisoned = true;
        g_rec->why = why ? why : "(no reason given)";
    }
    return RD_OK;
}

void* rd_plan_end(int* n_launches, int* n_segments) {
    using namespace rd;
    Plan* p;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        p = g_rec;
        g_rec = nullptr;
        g_recording = 0;
    }
    if (!p) {
        set_error("rd_plan_end: no plan is being recorded");
        return nullptr;
    }
    if (p->poisoned) {
        set_error("rd_plan_end: the recording cannot be replayed: %s", p->why.c_str());
        delete p;
        return nullptr;
    }
    if (p->seg_end.empty() || p->seg_end.back() != p->ops.size()) p->seg_end.push_back(p->ops.size());
    p->events.resize(p->n_events);
    // stream-to-stream ordering on ONE device: no timing, and no system-scope fence when the event fires (RD_PLAN_SYSFENCE=1 keeps it)
    static const int sysfence = getenv("RD_PLAN_SYSFENCE") ? atoi(getenv("RD_PLAN_SYSFENCE")) : 0;
    for (int i = 0; i < p->n_events; ++i)
        if (hipEventCreateWithFlags(&p->events[i], hipEventDisableTiming | (sysfence ? 0 : hipEventDisableSystemFence)) != hipSuccess) {
            set_error("rd_plan_end: hipEventCreate failed");
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(p->events[j]);
            delete p;
            return nullptr;
        }
    if (n_launches) *n_launches = p->n_launches;
    if (n_segments) *n_segments = (int)p->seg_end.size();
    return p;
}

int rd_plan_replay(void* plan, int segment, rd_stream_t main_stream, rd_stream_t side_stream) {
    using namespace rd;
    Plan* p = (Plan*)plan;
    RD_REQUIRE(p && segment >= 0 && segment < (int)p->seg_end.size(), "rd_plan_replay: bad plan / segment %d", segment);
    hipStream_t st[2] = {(hipStream_t)main_stream, (hipStream_t)side_stream};
    // diagnosis: 1 = synchronise after every launch | 2 = everything on the main stream | 3 / 4 = synchronise after SIDE / main launches
    static const int dbg = getenv("RD_PLAN_DEBUG") ? atoi(getenv("RD_PLAN_DEBUG")) : 0;
    if (dbg == 2) st[1] = st[0];
    static const int dlo = getenv("RD_PLAN_SYNC_LO") ? atoi(getenv("RD_PLAN_SYNC_LO")) : -1;       // 5: synchronise after ops [lo, hi)
    static const int dhi = getenv("RD_PLAN_SYNC_HI") ? atoi(getenv("RD_PLAN_SYNC_HI")) : -1;
    const size_t a = segment ? p->seg_end[segment - 1] : 0, b = p->seg_end[segment];
    void* argv[64];
    // A launch followed by an event record on its stream: the event rides on the kernel's own completion signal
    // (hipExtLaunchKernel's stop event) instead of a marker packet of its own, which the NEXT kernel of the stream would wait
    // for -- ~6 us in front of every data-gradient launch of the backward (RD_PLAN_EXT=0: separate records, for A/B runs)
    static const int ext = getenv("RD_PLAN_EXT") ? atoi(getenv("RD_PLAN_EXT")) : 1;
    for (size_t i = a; i < b; ++i) {
        const PlanOp& op = p->ops[i];
        if (op.kind == 0) {
            for (int k = 0; k < op.nargs; ++k) argv[k] = p->blob.data() + p->arg_off[op.arg0 + k];
            hipError_t e;
            if (ext && i + 1 < b && p->ops[i + 1].kind == 1 && p->ops[i + 1].role == op.role) {
                e = hipExtLaunchKernel(op.fn, op.grid, op.block, argv, op.shmem, st[op.role], nullptr, p->events[p->ops[i + 1].ev], 0);
                ++i;                                   // the record is done
            } else {
                e = hipLaunchKernel(op.fn, op.grid, op.block, argv, op.shmem, st[op.role]);
            }
            if (e != hipSuccess) return check_hip(e, "rd_plan_replay: launch");
            if ((dbg == 1 || (dbg == 3 && op.role == 1) || (dbg == 4 && op.role == 0) || (dbg == 5 && (int)i >= dlo && (int)i < dhi)) &&
                hipDeviceSynchronize() != hipSuccess)
                return check_hip(hipGetLastError(), "rd_plan_replay: debug sync");
        } else if (op.kind == 1) {
            if (int e = check_hip(hipEventRecord(p->events[op.ev], st[op.role]), "rd_plan_replay: event record")) return e;
        } else {
            if (int e = check_hip(hipStreamWaitEvent(st[op.role], p->events[op.ev], 0), "rd_plan_replay: event wait")) return e;
        }
    }
    return RD_OK;
}

// diagnosis: the plan's operations in enqueue order, one line each, to stderr
int rd_plan_dump(void* plan) {
    rd::Plan* p = (rd::Plan*)plan;
    if (!p) return RD_ERR_ARG;
    size_t seg = 0;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const rd::PlanOp& op = p->ops[i];
        while (seg < p->seg_end.size() && p->seg_end[seg] <= i) fprintf(stderr, "  ---- end of segment %zu\n", seg++);
        if (op.kind == 0) {
            const char* nm = hipKernelNameRefByPtr(op.fn, nullptr);
            fprintf(stderr, "%4zu %s launch %-60.60s grid %u block %u\n", i, op.role ? "SIDE" : "main", nm ? nm : "?", op.grid.x, op.block.x);
        } else {
            fprintf(stderr, "%4zu %s %s event %d\n", i, op.role ? "SIDE" : "main", op.kind == 1 ? "RECORD" : "WAIT  ", op.ev);
        }
    }
    return RD_OK;
}

int rd_plan_free(void* plan) {
    rd::Plan* p = (rd::Plan*)plan;
    if (!p) return RD_OK;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    delete p;
    return RD_OK;
}

// zero a 16-byte aligned range with a kernel of the library (a launch a plan can hold; hipMemsetAsync is not)
__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
int rd_zero(void* p, size_t bytes, rd_stream_t s) {
    using namespace rd;
    RD_REQUIRE(p && bytes % 16 == 0 && ((size_t)p & 15) == 0, "rd_zero: a 16-byte aligned range of whole 16-byte units");
    if (!bytes) return RD_OK;
    const size_t n16 = bytes / 16;
    RD_LAUNCH(zero_kernel, dim3((unsigned)(n16 < 256 * 256 ? (n16 + 255) / 256 : 256)), dim3(256), 0, (hipStream_t)s, (uint4*)p, n16);
    return check_hip(hipGetLastError(), "rd_zero");
}

// up to eight device-to-device copies in ONE launch (the five tensors of a batch into the static input buffers of a captured /
// planned iteration: five hipMemcpyAsync blit kernels were 55 us at the head of every step, serialized before the first convolution)
struct CopySegs {
    const uint4* src[8];
    uint4* dst[8];
    unsigned long long start[9];      // prefix sums of the segment lengths in 16-byte units
    int n;
};
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs s) {
    const unsigned long long total = s.start[s.n];
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (unsigned long long)gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) k += (j < s.n && i >= s.start[j]) ? 1 : 0;
        s.dst[k][i - s.start[k]] = s.src[k][i - s.start[k]];
    }
}
int rd_copy_segments(void* const* dst, const void* const* src, const size_t* bytes, int n, rd_stream_t st) {
    using namespace rd;
    RD_REQUIRE(dst && src && bytes && n >= 1 && n <= 8, "rd_copy_segments: 1..8 segments");
    CopySegs s = {};
    s.n = n;
    for (int k = 0; k < n; ++k) {
        RD_REQUIRE(dst[k] && src[k] && bytes[k] % 16 == 0 && (((size_t)dst[k] | (size_t)src[k]) & 15) == 0,
                   "rd_copy_segments: segment %d is not a 16-byte aligned range of whole 16-byte units", k);
        s.src[k] = (const uint4*)src[k];
        s.dst[k] = (uint4*)dst[k];
        s.start[k + 1] = s.start[k] + bytes[k] / 16;
    }
    const unsigned long long total = s.start[n];
    if (!total) return RD_OK;
    const unsigned grid = (unsigned)(total < 2048ull * 256 ? (total + 255) / 256 : 2048);
    RD_LAUNCH(copy_segments_kernel, dim3(grid), dim3(256), 0, (hipStream_t)st, s);
    return check_hip(hipGetLastError(), "rd_copy_segments");
}

int rd_quant_next(const unsigned* a_amax, const unsigned* b_amax, unsigned* out_amax, unsigned* out2_amax) {
    rd::t_quant = {a_amax, b_amax, out_amax, out2_amax, 0};
    return RD_OK;
}

int rd_quant_next_img(const unsigned* a_amax, const unsigned* b_amax, unsigned* out_amax, unsigned* out2_amax, int img_stride_words) {
    RD_REQUIRE(img_stride_words >= RD_AMAX_SLOT_BYTES / 4 && img_stride_words % 32 == 0,
               "rd_quant_next_img: the per-image stride must be a whole number of 128-byte lines and hold a slot (got %d words)", img_stride_words);
    rd::t_quant = {a_amax, b_amax, out_amax, out2_amax, img_stride_words};
    return RD_OK;
}

int rd_tune_set(const char* name, int value) {
    const int i = name ? rd::tune_index(name, strlen(name)) : -1;
    if (i < 0) {
        rd::set_error("rd_tune_set: unknown knob '%s'", name ? name : "(null)");
        return RD_ERR_ARG;
    }
    rd::g_tune[i].value = value;
    return RD_OK;
}

int rd_tune_get(const char* name, int* value) {
    const int i = name ? rd::tune_index(name, strlen(name)) : -1;
    if (i < 0 || !value) {
        rd::set_error("rd_tune_get: unknown knob '%s'", name ? name : "(null)");
        return RD_ERR_ARG;
    }
    *value = rd::g_tune[i].value;
    return RD_OK;
}

int rd_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(rd::g_mu);
    rd::g_level = on < 0 ? 0 : (on > 2 ? 2 : on);
    return RD_OK;
}

int rd_prof_reset(void) {
    std::lock_guard<std::mutex> lk(rd::g_mu);
    rd::drain_locked();
    rd::g_cls.clear();
    return RD_OK;
}

int rd_prof_collect(rd_prof_entry* out, int max_entries) {
    std::lock_guard<std::mutex> lk(rd::g_mu);
    rd::drain_locked();
    int n = 0;
    for (auto& c : rd::g_cls) {
        if (n >= max_entries) break;
        memset(&out[n], 0, sizeof(rd_prof_entry));
        strncpy(out[n].name, c.name.c_str(), sizeof(out[n].name) - 1);
        out[n].launches = c.launches;
        out[n].ms = c.ms;
        out[n].flops = c.flops;
        out[n].bytes = c.bytes;
        ++n;
    }
    return n;
}

}  // extern "C"

extern "C" int rd_set_splitk_workspace(void* ws, size_t bytes, rd_stream_t stream) {
    using namespace rd;
    hipStream_t s = (hipStream_t)stream;
    int dev = -1;
    if (int e = check_hip(hipGetDevice(&dev), "rd_set_splitk_workspace")) return e;
    if (ws) {
        if (bytes < kSkTicketBytes + (1u << 20) || ((size_t)ws & 255)) {
            set_error("rd_set_splitk_workspace: need a 256-byte aligned buffer of at least %zu bytes", kSkTicketBytes + (1u << 20));
            return RD_ERR_ARG;
        }
        // the scratch has to live on the device the launches of (current device, stream) go to: tickets and slabs are
        // accessed with agent-scope atomics, which another device's memory does not honour
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, ws) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != dev) {
            (void)hipGetLastError();
            set_error("rd_set_splitk_workspace: ws is not device memory of the current device %d", dev);
            return RD_ERR_ARG;
        }
    }
    std::lock_guard<std::mutex> lk(g_sk_mu);
    for (size_t i = 0; i < g_sk.size(); ++i)
        if (g_sk[i].dev == dev && g_sk[i].s == s) {
            g_sk.erase(g_sk.begin() + i);
            break;
        }
    if (!ws) return RD_OK;                           // un-register
    plan_poison(s, "rd_set_splitk_workspace inside a recording (register the scratch before)");
    if (int e = check_hip(hipMemsetAsync(ws, 0, kSkTicketBytes, s), "rd_set_splitk_workspace")) return e;
    g_sk.push_back({dev, s, (char*)ws, bytes});
    return RD_OK;
}

// ---- host side of the sweep's raster read-back (include/resdepth_hip.h) ---------------------------------------------
extern "C" int rd_host_register(void* p, size_t bytes) {
    if (!p || !bytes) {
        rd::set_error("rd_host_register: null range");
        return RD_ERR_ARG;
    }
    return rd::check_hip(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped), "rd_host_register");
}

extern "C" int rd_host_unregister(void* p) { return rd::check_hip(hipHostUnregister(p), "rd_host_unregister"); }

// Device -> page-locked host memory with a SMALL grid -- an r05 experiment kept behind the knob `d2h_blocks` (default 0 = off).
// hipMemcpyAsync to pinned memory runs as the runtime's blit kernel on this stack (`__amd_rocclr_copyBuffer` in the kernel trace,
// nothing in the memory-copy trace): a 33 MB raster stripe at the PCIe rate (0.62 ms), and for 45 % of that time no kernel of the
// sweep runs beside it (scripts/gaps_infer.sh) -- streamed under the sweep the read-back costs about half of what the serial copy
// did, not nothing.  Hypothesis: a PCIe-bound copy needs a few dozen waves, not the chip's wave slots.  Measured (cfg-G, 8192^2,
// interleaved on one box, ms per sweep): hipMemcpyAsync 393.2 / 393.9; this kernel with 16 / 48 / 128 workgroups 398.3-399.1 /
// 400.5-401.6 / 401.1-401.2 -- 1.3-2 % SLOWER (posted stores from a few waves do not reach the PCIe rate, and the longer the copy
// runs the longer it shares HBM and the fabric with the sweep).  The runtime's copy stays the default.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_to_host_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}

extern "C" int rd_copy_to_host_async(void* dst_host, const void* src_dev, size_t bytes, rd_stream_t stream) {
    if (!bytes) return RD_OK;
    if (!dst_host || !src_dev) {
        rd::set_error("rd_copy_to_host_async: null pointer");
        return RD_ERR_ARG;
    }
    rd::plan_poison((hipStream_t)stream, "rd_copy_to_host_async inside a recording");
    const int blocks = rd::tune(rd::TUNE_D2H_BLOCKS);
    void* dmap = nullptr;
    if (blocks > 0 && bytes % 16 == 0 && ((size_t)dst_host % 16) == 0 && ((size_t)src_dev % 16) == 0 &&
        hipHostGetDevicePointer(&dmap, dst_host, 0) == hipSuccess && dmap) {
        RD_LAUNCH(copy_to_host_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const u32x4*>(src_dev), reinterpret_cast<u32x4*>(dmap), bytes / 16);
        return rd::check_hip(hipGetLastError(), "rd_copy_to_host_async");
    }
    (void)hipGetLastError();          // not mapped host memory (pageable destination): the runtime's staged copy
    return rd::check_hip(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "rd_copy_to_host_async");
}
