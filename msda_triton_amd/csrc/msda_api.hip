// msda_api.hip — library-level pieces of the C ABI: version, last-error text, A/B options.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <hip/hip_runtime.h>

#include <atomic>
#include <thread>
#include <vector>
#include <map>
#include <mutex>
#include <string>

#include "../../include/msda_hip.h"

namespace msda {

static std::atomic<int> g_xcd_map{1};
static std::atomic<int> g_value_path{0};
static std::atomic<int> g_wg_target{1 << 30};
static std::atomic<int> g_cell_slices{0};
static std::atomic<int> g_debug{0};
static std::atomic<int> g_small_ns{0};
static std::atomic<int> g_q_round{0};
static std::atomic<int> g_level_cells{0};
static std::atomic<int> g_overlap{-1};
static std::atomic<int> g_gather_win{0};
static std::atomic<int> g_place_path{0};
static std::atomic<int> g_profile{0};
static std::atomic<int> g_records_in_grads{1};
static std::atomic<int> g_strict{0};
static std::atomic<int> g_lds_levels{1};
static std::atomic<int> g_lds_budget{-1};
static std::atomic<int> g_unit_fwd{1};
static std::atomic<int> g_lds_over{1};
static std::atomic<int> g_lds_planes{0};
static std::atomic<int> g_linear_slots{320};
static std::atomic<int> g_touch{1};
static std::atomic<int> g_unit_waves{1};
static std::atomic<int> g_ws_passes{1};
static std::atomic<int> g_lds_stagger{0};  // (measured 0 / 4 / 12 / 24 at c2 @ 10k: 0 is fastest — the work counter desynchronises the waves by itself)

// One side stream + two events per (host thread, device), created on first use and kept for the life of the thread.
// Per THREAD, because the fork (record on the user's stream, wait on the side stream) and the join are two calls
// each: with a stream / event pair shared by all callers, two threads issuing backwards on one device (autograd's
// worker threads, different user streams) could interleave record(A) record(B) wait(.) and make the sample kernel wait
// for the wrong work.  Inside one thread everything is in program order, whatever streams the caller alternates.
static const std::thread::id g_loader_thread = std::this_thread::get_id();  // the thread that loaded the library
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool tried = false;
    // a host thread that ends gives its stream and events back (thread pools, per-call worker threads); at process
    // exit the runtime may already be gone: errors are swallowed
    ~SideStream()
    {
        if (std::this_thread::get_id() == g_loader_thread) return;  // process exit: leave it to the runtime's teardown
        if (join != nullptr) (void)hipEventDestroy(join);
        if (fork != nullptr) (void)hipEventDestroy(fork);
        if (stream != nullptr) (void)hipStreamDestroy(stream);
        (void)hipGetLastError();
    }
};
static thread_local SideStream t_side[64];

static SideStream *side_for_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    SideStream &s = t_side[dev];
    if (!s.tried) {
        s.tried = true;
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            s.stream = nullptr;
        }
    }
    return s.stream ? &s : nullptr;
}

hipStream_t side_stream_fork(hipStream_t user)
{
    SideStream *s = side_for_current_device();
    if (s == nullptr) return nullptr;
    if (hipEventRecord(s->fork, user) != hipSuccess || hipStreamWaitEvent(s->stream, s->fork, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return s->stream;
}

int side_stream_join(hipStream_t user)
{
    SideStream *s = side_for_current_device();
    if (s == nullptr) return (int)hipErrorInvalidValue;
    hipError_t e = hipEventRecord(s->join, s->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(user, s->join, 0);
    return (int)e;
}
static thread_local char g_err[256] = "";

int option_xcd_map() { return g_xcd_map.load(std::memory_order_relaxed); }
int option_value_path() { return g_value_path.load(std::memory_order_relaxed); }
int option_wg_target() { return g_wg_target.load(std::memory_order_relaxed); }
int option_cell_slices() { return g_cell_slices.load(std::memory_order_relaxed); }
int option_debug() { return g_debug.load(std::memory_order_relaxed); }
int option_small_ns() { return g_small_ns.load(std::memory_order_relaxed); }
int option_q_round() { return g_q_round.load(std::memory_order_relaxed); }
int option_level_cells() { return g_level_cells.load(std::memory_order_relaxed); }  // process-wide promise (0: unknown)
int option_overlap() { return g_overlap.load(std::memory_order_relaxed); }
int option_gather_win() { return g_gather_win.load(std::memory_order_relaxed); }
int option_place_path() { return g_place_path.load(std::memory_order_relaxed); }
int option_profile() { return g_profile.load(std::memory_order_relaxed); }
int option_records_in_grads() { return g_records_in_grads.load(std::memory_order_relaxed); }
int option_strict() { return g_strict.load(std::memory_order_relaxed); }
int option_lds_levels() { return g_lds_levels.load(std::memory_order_relaxed); }
int option_lds_stagger() { return g_lds_stagger.load(std::memory_order_relaxed); }
int option_lds_over() { return g_lds_over.load(std::memory_order_relaxed); }
int option_lds_planes() { return g_lds_planes.load(std::memory_order_relaxed); }
int option_linear_slots() { return g_linear_slots.load(std::memory_order_relaxed); }
int option_unit_fwd() { return g_unit_fwd.load(std::memory_order_relaxed); }
int option_touch() { return g_touch.load(std::memory_order_relaxed); }
int option_unit_waves() { return g_unit_waves.load(std::memory_order_relaxed); }
int option_ws_passes() { return g_ws_passes.load(std::memory_order_relaxed); }
int option_lds_budget() { return g_lds_budget.load(std::memory_order_relaxed); }  // dev knob: cap on the level bytes (-1: none)
// CUs of the current device, asked once per device (the LDS-level gather variants size their grid by it)
int device_cu_count()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;  // MI355X
    }
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// ---- measurement only: device time of every kernel the library launches on this thread (option "profile") ----
struct ProfileRec {
    const char *name;
    hipEvent_t a, b;
    bool ended;  // profile_end has recorded `b` (only then may msda_profile_read consume the record)
};
// (process-wide: autograd launches the backward from its own thread.  Records are heap objects and the token handed to
// the launch is the record itself, so a read that runs between a launch's begin and end can neither redirect nor drop it;
// every access to the list is under the mutex — ADVICE r04)
static std::vector<ProfileRec *> t_profile;
static std::mutex t_profile_mutex;

void *profile_begin(const char *name, hipStream_t stream)
{
    if (!option_profile()) return nullptr;
    const std::lock_guard<std::mutex> lock(t_profile_mutex);
    if (t_profile.size() >= 65536) return nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipEventCreate(&b) != hipSuccess || hipEventRecord(a, stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipEventDestroy(a);
        if (b != nullptr) (void)hipEventDestroy(b);
        return nullptr;
    }
    ProfileRec *r = new ProfileRec{name, a, b, false};
    t_profile.push_back(r);
    return r;
}

void profile_end(void *token, hipStream_t stream)
{
    if (token == nullptr) return;
    ProfileRec *r = static_cast<ProfileRec *>(token);
    const std::lock_guard<std::mutex> lock(t_profile_mutex);
    (void)hipEventRecord(r->b, stream);
    r->ended = true;
}

// ---- measurement only: what the last launches were (msda_last_launch_info) — process-wide, like the profile records:
//      autograd issues the backward from its own thread ----
static std::atomic<int> g_info[8];
void note_launch(int slot, int value)
{
    if (slot >= 0 && slot < 8) g_info[slot].store(value, std::memory_order_relaxed);
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace msda

extern "C" int msda_abi_version(void) { return MSDA_ABI_VERSION; }

// the layout lives in a device header (msda_value_sorted.hpp); msda_f32.hip exposes its size formula
extern "C" int64_t msda_bwd_workspace_bytes_impl(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, int, int64_t, int);

extern "C" int64_t msda_bwd_workspace_bytes(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L,
                                            int64_t P, int elem_size, int value_elem_size, int64_t max_level_cells,
                                            int flags)
{
    if (B < 0 || I < 0 || H < 0 || D < 0 || Q < 0 || L < 0 || P < 0) return 0;
    return msda_bwd_workspace_bytes_impl(B, I, H, D, Q, L, P, elem_size, (flags & MSDA_WS_RECORDS_IN_GRADS) ? 1 : 0,
                                         value_elem_size > 0 ? value_elem_size : elem_size, max_level_cells,
                                         ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) : msda::option_ws_passes());
}

extern "C" int64_t msda_bwd_fused_workspace_bytes(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L,
                                                  int64_t P, int elem_size, int value_elem_size, int64_t max_level_cells,
                                                  int flags)
{
    if (B < 0 || I < 0 || H < 0 || D < 0 || Q < 0 || L < 0 || P < 0 || elem_size <= 0) return 0;
    // the derived sampling points + attention weights (3 elements per sample, rounded up to 256 bytes), then
    // the sorted pipeline's own workspace (msda_launch.hpp: fused_mat_bytes)
    (void)value_elem_size;
    const int64_t mat = (B * Q * H * L * P * 3 * (int64_t)elem_size + 255) / 256 * 256;
    return mat + msda_bwd_workspace_bytes_impl(B, I, H, D, Q, L, P, elem_size, 0, 0, max_level_cells,
                                               ((flags >> 8) & 0xff) ? ((flags >> 8) & 0xff) : msda::option_ws_passes());
}

extern "C" int msda_bwd_supported_impl(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int);

extern "C" int msda_bwd_supported(int64_t B, int64_t I, int64_t H, int64_t D, int64_t Q, int64_t L, int64_t P, int elem_size)
{
    if (B < 0 || I < 0 || H < 0 || D < 0 || Q < 0 || L < 0 || P < 0 || elem_size <= 0) return 0;
    return msda_bwd_supported_impl(B, I, H, D, Q, L, P, elem_size);
}

extern "C" int64_t msda_fused_lp_limit_impl(int64_t, int);

extern "C" int64_t msda_fused_lp_limit(int64_t D, int elem_size) { return msda_fused_lp_limit_impl(D, elem_size); }

extern "C" const char *msda_last_error(void) { return msda::g_err; }

extern "C" int msda_last_launch_info(const char *key)
{
    static const char *const kKeys[8] = {"fwd_variant", "fwd_lds_level_bytes", "fwd_lds_planes", "fwd_workgroups",
                                         "sample_variant", "sample_lds_level_bytes", "value_path", "value_passes"};
    for (int i = 0; key != nullptr && i < 8; ++i)
        if (strcmp(key, kKeys[i]) == 0) return msda::g_info[i].load(std::memory_order_relaxed);
    msda::set_error("unknown launch-info key '%s'", key ? key : "(null)");
    return MSDA_ERR_BAD_ARG;
}

// "name launches total_us\n" per kernel launched on this thread since the last read while option "profile" was 1;
// (any thread: the records are process-wide) waits for the recorded events, then forgets them.  Returns the number of characters written (without the NUL).
extern "C" int msda_profile_read(char *buf, int cap)
{
    std::map<std::string, std::pair<int, double>> acc;
    const std::lock_guard<std::mutex> lock(msda::t_profile_mutex);
    std::vector<msda::ProfileRec *> open;  // begun, not ended yet: they stay for the next read
    for (msda::ProfileRec *r : msda::t_profile) {
        if (!r->ended) {
            open.push_back(r);
            continue;
        }
        float ms = 0.f;
        if (hipEventSynchronize(r->b) == hipSuccess && hipEventElapsedTime(&ms, r->a, r->b) == hipSuccess) {
            auto &e = acc[r->name];
            e.first += 1;
            e.second += (double)ms * 1e3;
        }
        (void)hipEventDestroy(r->a);
        (void)hipEventDestroy(r->b);
        delete r;
    }
    (void)hipGetLastError();
    msda::t_profile.swap(open);
    std::string out;
    for (const auto &kv : acc) {
        char line[160];
        snprintf(line, sizeof(line), "%s %d %.3f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        out += line;
    }
    if (buf == nullptr || cap <= 0) return (int)out.size();
    const int n = (int)out.size() < cap - 1 ? (int)out.size() : cap - 1;
    memcpy(buf, out.data(), (size_t)n);
    buf[n] = 0;
    return n;
}

// One table for set and get.  dev: an experiment knob, reachable only in builds with -DMSDA_DEV (include/msda_hip.h).
namespace msda {
struct OptionEntry {
    const char *key;
    std::atomic<int> *slot;
    int lo, hi;  // accepted range (values outside are clamped to `fallback` when clamp, rejected otherwise)
    bool dev;
};
static const OptionEntry kOptions[] = {
    {"xcd_map", &g_xcd_map, 0, 2, false},
    {"value_path", &g_value_path, 0, 3, false},
    {"level_cells", &g_level_cells, 0, 0x7fffffff, false},
    {"q_round", &g_q_round, 0, 0x7fffffff, false},
    {"small_ns", &g_small_ns, 0, 16, false},
    {"overlap", &g_overlap, -1, 1, false},
    {"place_path", &g_place_path, 0, 3, false},
    {"strict", &g_strict, 0, 1, false},
    {"records_in_grads", &g_records_in_grads, 0, 1, false},
    {"profile", &g_profile, 0, 1, false},
    {"lds_levels", &g_lds_levels, 0, 2, false},
    {"unit_fwd", &g_unit_fwd, 0, 2, false},
    {"debug", &g_debug, (int)0x80000000, 0x7fffffff, true},
    {"gather_win", &g_gather_win, 0, 4096, true},
    {"cell_slices", &g_cell_slices, 0, 64, true},
    {"wg_target", &g_wg_target, 1, 0x7fffffff, true},
    {"lds_budget", &g_lds_budget, -1, 0x7fffffff, true},
    {"lds_stagger", &g_lds_stagger, 0, 4096, true},
    {"lds_over", &g_lds_over, 1, 8, true},
    {"lds_planes", &g_lds_planes, 0, 2, false},
    {"linear_slots", &g_linear_slots, 1, 1 << 30, false},
    {"touch", &g_touch, 0, 2, false},
    {"unit_waves", &g_unit_waves, 1, 2, false},
    {"ws_passes", &g_ws_passes, 1, 128, false},
};
static const OptionEntry *find_option(const char *key)
{
    if (key == nullptr) return nullptr;
    for (const OptionEntry &o : kOptions) {
        if (strcmp(key, o.key) != 0) continue;
#ifndef MSDA_DEV
        if (o.dev) return nullptr;
#endif
        return &o;
    }
    return nullptr;
}
}  // namespace msda

extern "C" int msda_set_option(const char *key, int value)
{
    const msda::OptionEntry *o = msda::find_option(key);
    if (o == nullptr) {
        msda::set_error("unknown option '%s'", key ? key : "(null)");
        return MSDA_ERR_BAD_ARG;
    }
    if (value < o->lo || value > o->hi || (o->slot == &msda::g_value_path && value == 1) ||
        (o->slot == &msda::g_gather_win && value != 0 && value < 8)) {
        msda::set_error("option '%s': value %d out of range [%d, %d]", key, value, o->lo, o->hi);
        return MSDA_ERR_BAD_ARG;
    }
    o->slot->store(value, std::memory_order_relaxed);
    return 0;
}

extern "C" int msda_get_option(const char *key)
{
    const msda::OptionEntry *o = msda::find_option(key);
    if (o == nullptr) {
        msda::set_error("unknown option '%s'", key ? key : "(null)");
        return MSDA_ERR_BAD_ARG;
    }
    return o->slot->load(std::memory_order_relaxed);
}
