// C ABI of libfrosting_rasterizer.so (include/frosting_rasterizer.h): host-side
// orchestration of the forward / backward kernel sequence on the caller's HIP
// stream.  Mirrors the reference's Rasterizer::forward / backward control flow
// (rasterizer_impl.cu:198-336, :340-434) -- one blocking 48-byte read-back for
// num_rendered, everything else asynchronous.
#include "../../include/frosting_rasterizer.h"
#include "kernels.h"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <thread>

namespace {

thread_local char g_err[512] = "";

// spin-wait hint of the mailbox polls (the host side is not tied to x86)
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

std::atomic<int> g_exact_blend{-1};
std::atomic<int> g_profile{0};
std::atomic<int> g_profile_stage{-1};  // -1: every stage; k: only stage k gets events (each record costs ~3 us of stream time)
// Options below shape the FORWARD only.  What a backward needs to know about the forward that filled its
// buffers travels with those buffers: the carve of every field the backward reads depends on (P, W, H, R)
// alone, and the binning mode is stamped into the image chunk's counters (Counters::tight_binning).
std::atomic<int> g_global_bins{0};    // test hook: force the large-image (global-atomic) binning path
std::atomic<int> g_ablate{0};         // TIMING EXPERIMENTS ONLY: kernels skip parts of their work (results are wrong)
std::atomic<int> g_async_sh{0};       // SH colours on a side stream beside the binning stages (0: inside preprocess)
std::atomic<int> g_bwd_batch{3};      // tuning: instances per reduction step of the backward blend (2 | 3)
std::atomic<int> g_bwd_seg_log{0};    // 0: the backward blend's segment length by the frame's instance count (frg_common.h) | 8 .. 10: pinned
std::atomic<int> g_tight_binning{0};  // drop (Gaussian, tile) instances that cannot reach alpha >= 1/255 in the tile
// TIMING EXPERIMENTS ONLY (results are those of the previous frame's lists / slots): bit 0 launches the forward blend
// beside the sort, bit 1 the per-Gaussian backward beside the backward blend -- an upper bound on what overlapping
// a VALU-bound with an LDS- or HBM-bound stage can give before any dependency-respecting pipeline is built
std::atomic<int> g_probe{0};
// TEST HOOK (FROSTING_EXPERIMENTS=1): pretend every forward posted "no heavy waves", so that the backward skips the 16-wave
// launch of the per-Gaussian backward whatever the geometry holds -- the plain kernel's safety net must then do that work
std::atomic<int> g_assume_no_heavy{0};

// Optional per-stage GPU timing (frg_set_option("profile", 1)): hipEvents are
// recorded on the caller's stream between the kernels of one forward / backward;
// frg_stage_times() synchronises and returns the elapsed milliseconds.
enum { ST_PREPROCESS = 0, ST_SCAN, ST_SCATTER, ST_SORT, ST_BLEND_FWD, ST_BLEND_BWD, ST_PREPROCESS_BWD, ST_SH_COLOR, ST_COUNT };
// Event pairs are kept for the last ST_SLOTS launches of every stage and only read (and
// synchronised on) by frg_stage_times(), so timing a run of steps does not serialise them.
constexpr int ST_SLOTS = 64;
struct StageTimers {
    hipEvent_t ev[ST_COUNT][ST_SLOTS][2];
    unsigned launches[ST_COUNT];
    bool init = false;
    bool ensure()
    {
        if (init) return true;
        for (int i = 0; i < ST_COUNT; i++) {
            for (int k = 0; k < ST_SLOTS; k++) {
                if (hipEventCreate(&ev[i][k][0]) != hipSuccess || hipEventCreate(&ev[i][k][1]) != hipSuccess) return false;
            }
            launches[i] = 0;
        }
        init = true;
        return true;
    }
};
// One timer set per DEVICE, shared by every host thread (torch runs autograd backward on its engine
// thread: thread-local timers would never show the backward stages to the thread that asks for them).
constexpr int kMaxDevices = 16;
std::mutex g_timers_mu;
StageTimers g_timers[kMaxDevices];

struct StageScope {
    int id; hipStream_t s; bool on; int slot; int dev;
    StageScope(int id_, hipStream_t s_) : id(id_), s(s_), on(g_profile.load() != 0), slot(0), dev(-1)
    {
        const int only = g_profile_stage.load();
        if (only >= 0 && only != id) on = false;
        if (!on) return;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { on = false; return; }
        std::lock_guard<std::mutex> lk(g_timers_mu);
        StageTimers& t = g_timers[dev];
        if (!t.ensure()) { on = false; return; }
        slot = (int)(t.launches[id] % ST_SLOTS);
        (void)hipEventRecord(t.ev[id][slot][0], s);
    }
    ~StageScope()
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_timers_mu);
        StageTimers& t = g_timers[dev];
        (void)hipEventRecord(t.ev[id][slot][1], s);
        t.launches[id]++;
    }
};

int exact_blend()
{
    int v = g_exact_blend.load();
    if (v < 0) {
        const char* e = getenv("FROSTING_EXACT_BLEND");
        v = (e && e[0] == '1') ? 1 : 0;
        g_exact_blend.store(v);
    }
    return v;
}

// One pinned landing pad per host thread for the counters read-back.
frg::Counters* pinned_counters()
{
    thread_local frg::Counters* p = nullptr;
    if (!p) {
        if (hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(frg::Counters), hipHostMallocDefault) != hipSuccess) p = nullptr;
    }
    return p;
}

// Deferred-counters forward: the counters of each outstanding forward land in a pinned slot
// behind an event; frg_forward_finish() waits on that event only (not on the whole stream).
struct PendingCounters {
    frg::Counters* host = nullptr;     // pinned
    hipEvent_t ev = nullptr;           // counters have landed in `host`
    hipEvent_t scanned = nullptr;      // scan finished on the caller's stream
    hipStream_t copy_stream = nullptr; // carries the 48-byte read-back off the caller's stream
    const void* key = nullptr;      // image buffer of the forward; cleared by frg_forward_finish
    const void* last_key = nullptr; // ... kept: the next forward on the same buffer orders itself after the read-back
    int device = -1;
};
constexpr int kPendingSlots = 8;
struct PendingRing {
    PendingCounters slot[kPendingSlots];
    int next = 0;
    uint32_t last_class_count[FRG_SORT_CLASSES] = {0, 0, 0, 0, 0};
    bool have_hint = false;
    PendingCounters* acquire(const void* key, bool* reused)
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        PendingCounters* p = nullptr;
        for (auto& c : slot) if (c.last_key == key && c.ev && c.device == dev) p = &c;   // the same image buffer again
        *reused = p != nullptr;
        if (!p) { p = &slot[next]; next = (next + 1) % kPendingSlots; }
        if (!p->host && hipHostMalloc(reinterpret_cast<void**>(&p->host), sizeof(frg::Counters), hipHostMallocDefault) != hipSuccess) return nullptr;
        if (p->ev && p->device != dev) {
            (void)hipEventDestroy(p->ev); (void)hipEventDestroy(p->scanned); (void)hipStreamDestroy(p->copy_stream);
            p->ev = nullptr; p->scanned = nullptr; p->copy_stream = nullptr;
        }
        if (!p->ev && hipEventCreateWithFlags(&p->ev, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (!p->scanned && hipEventCreateWithFlags(&p->scanned, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (!p->copy_stream && hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        p->device = dev;
        p->key = key;
        p->last_key = key;
        return p;
    }
    PendingCounters* find(const void* key)
    {
        for (auto& c : slot) if (c.key == key && c.ev) return &c;
        return nullptr;
    }
};
thread_local PendingRing g_pending;

// The host thread's mailbox (frg::Mailbox, frg_common.h): pinned, mapped, written by the scan workgroups.
struct HostMail {
    frg::Mailbox* host = nullptr;
    uint32_t seq = 0;
    bool long_lists = false;     // the previous forward of this thread had tile lists beyond the LDS sort
    int last_P = 0; uint32_t last_seq = 0;      // the forward whose scatter post may still be in the mailbox
    // the previous forward of this thread, on a model of the same size, saw less than three quarters of it
    bool sparse_view(int P) const
    {
        if (!host || P != last_P || (uint32_t)(__atomic_load_n(&host->heavy_post, __ATOMIC_ACQUIRE) >> 32) != last_seq) return false;
        return (uint64_t)host->visible * 4u < (uint64_t)P * 3u;
    }
    bool failed = false;         // a post never arrived although the stream had drained: stay with the copy + synchronise
    frg::Mailbox* get()
    {
        if (!host && !failed) {
            if (hipHostMalloc(reinterpret_cast<void**>(&host), sizeof(frg::Mailbox), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
                host = nullptr; failed = true;
                (void)hipGetLastError();
            } else memset(host, 0, sizeof(frg::Mailbox));
        }
        return host;
    }
};
thread_local HostMail g_mail;
std::atomic<int> g_use_mailbox{1};
std::atomic<int> g_sh_no_dir{0};           // option "sh_dir_in_backward"
std::atomic<int> g_fwd_unroll8{1};         // option "fwd_unroll8": 0 never | 1 frames of a few long lists (the host's rule) | 2 always
std::atomic<int> g_fused_small{1};         // option "fused_small": frames of at most 2^20 instances sort their short lists inside the forward blend
std::atomic<int> g_sparse_sh{1};            // option "sparse_sh": the SH pass over the visible Gaussians only, where a view sees a part of the model
std::atomic<int> g_bwd_heavy_first{1};
std::atomic<int> g_clear_image_state{0};   // 1: the memset in front of every forward, needed or not

// What the host remembers about the forward that last filled a geometry buffer (process-wide: autograd runs the backward
// on another thread than the forward): which mailbox post is its scatter's, how many instances it rendered, whether it was
// told that no backward follows.  Scheduling hints and early refusals only -- keyed by ADDRESS, a note can be stale (a buffer
// copied to an address an earlier forward used), so nothing that decides a gradient bit hangs on one: the blend arithmetic
// and "nothing kept" are stamped into the image chunk by the forward's blend kernel (Counters::fwd_flags) and read there
// (backward_impl).  A ring of kFwdNotes entries; a forgotten forward's backward launches both forms of the per-Gaussian
// backward (as if nothing had been posted).  The pinned mailboxes are never freed.
struct FwdNote { const void* geom = nullptr; const frg::Mailbox* mail = nullptr; uint32_t seq = 0; int exact = -1; int rendered = -1; bool fwd_only = false; };
constexpr int kFwdNotes = 1024;
std::mutex g_heavy_mu;
FwdNote g_fwd_notes[kFwdNotes];
unsigned g_fwd_next = 0;
// a forward starts on `geom`: whatever an earlier forward posted about this buffer is void now
void note_forward(const void* geom, int exact, bool fwd_only)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (auto& n : g_fwd_notes) if (n.geom == geom) { n.mail = nullptr; n.seq = 0; n.exact = exact; n.rendered = -1; n.fwd_only = fwd_only; return; }
    g_fwd_notes[g_fwd_next++ % kFwdNotes] = FwdNote{geom, nullptr, 0, exact, -1, fwd_only};
}
// the forward that last filled `geom` was told that no backward would follow (frg_forward_args::forward_only): 1 | 0, or
// -1 when that forward is not remembered
int forward_was_forward_only(const void* geom)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (const auto& n : g_fwd_notes) if (n.geom == geom) return n.fwd_only ? 1 : 0;
    return -1;
}
// the blocking forward on `geom` rendered R instances (a deferred forward does not know)
void note_rendered(const void* geom, int R)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (auto& n : g_fwd_notes) if (n.geom == geom) { n.rendered = R; return; }
}
// -> the instance count of the forward that last filled `geom`, or -1 when it is not remembered
int forward_rendered(const void* geom)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (const auto& n : g_fwd_notes) if (n.geom == geom) return n.rendered;
    return -1;
}
void note_heavy_post(const void* geom, const frg::Mailbox* mail, uint32_t seq)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (auto& n : g_fwd_notes) if (n.geom == geom) { n.mail = mail; n.seq = seq; return; }
}
// -> the number of heavy waves of the forward that last filled `geom`, or -1 when unknown (no post, not arrived yet,
// the mailbox already belongs to a later forward)
int heavy_waves_posted(const void* geom)
{
    FwdNote n;
    {
        std::lock_guard<std::mutex> lk(g_heavy_mu);
        for (const auto& x : g_fwd_notes) if (x.geom == geom) n = x;
    }
    if (!n.mail || !g_use_mailbox.load(std::memory_order_relaxed)) return -1;
    // The forward returned when the scan stage was through; the scatter posts as its first act.  A backward called
    // straight away may be a few microseconds early: it waits that long (the GPU has the rest of the forward ahead of
    // it, the host nothing better to do), but not for a scatter stuck behind other work.
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
        const unsigned long long post = __atomic_load_n(&n.mail->heavy_post, __ATOMIC_ACQUIRE);    // (sequence << 32) | count, one word
        const uint32_t cur = (uint32_t)(post >> 32);
        if (cur == n.seq) return (int)((uint32_t)post > 0x7fffffffu ? 0x7fffffffu : (uint32_t)post);
        if ((int32_t)(cur - n.seq) > 0) return -1;      // the mailbox already carries a later forward's post
        if ((spin & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(40)) return -1;
        cpu_relax();
    }
}

// Two-call backward (frg_backward_args::phase): phase 2 reads the nine per-Gaussian sums phase 1 left in the workspace.
// What phase 1 was called with is remembered per workspace pointer; a phase 2 that does not match (an arena that grew or
// was reused between the calls, another frame's buffers) is refused instead of producing garbage gradients.
struct PhaseNote { const void* workspace = nullptr; const void* geom = nullptr; const void* image = nullptr; int P = 0, R = 0; };
constexpr int kPhaseNotes = 16;
PhaseNote g_phase_notes[kPhaseNotes];
unsigned g_phase_next = 0;
void note_phase1(const void* workspace, const void* geom, const void* image, int P, int R)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (auto& n : g_phase_notes) if (n.workspace == workspace) { n = PhaseNote{workspace, geom, image, P, R}; return; }
    g_phase_notes[g_phase_next++ % kPhaseNotes] = PhaseNote{workspace, geom, image, P, R};
}
bool phase1_matches(const void* workspace, const void* geom, const void* image, int P, int R)
{
    std::lock_guard<std::mutex> lk(g_heavy_mu);
    for (auto& n : g_phase_notes)
        if (n.workspace == workspace) {
            const bool ok = n.geom == geom && n.image == image && n.P == P && n.R == R;
            n = PhaseNote{};      // the sums are consumed once
            return ok;
        }
    return false;
}

// Spin until the kernel's post arrives.  false: the stream failed, or it drained without the post becoming visible.
bool mailbox_wait(const uint32_t* flag, uint32_t seq, hipStream_t stream)
{
    for (unsigned spin = 1;; spin++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return true;
        if ((spin & 0x7ffu) == 0) {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq;
            if (q != hipErrorNotReady) return false;
        }
        // a long wait (a 3 M-Gaussian preprocess is ~0.25 ms ahead of the first post): after the first ~10 us give the core
        // to whoever else wants it (data loaders, the other ranks' host threads on a busy node) between looks
        if (spin > 4096 && (spin & 63u) == 0) std::this_thread::yield();
        cpu_relax();
    }
}

// Side stream of the deferred SH colour kernel (one per host thread and device, like the sort's).
struct ShSide {
    hipStream_t stream = nullptr;
    hipEvent_t geo_done = nullptr, sh_done = nullptr;
    int device = -1;
    bool ensure()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev == device) return true;
        if (stream) (void)hipStreamDestroy(stream);
        if (geo_done) (void)hipEventDestroy(geo_done);
        if (sh_done) (void)hipEventDestroy(sh_done);
        stream = nullptr; geo_done = nullptr; sh_done = nullptr; device = -1;
        // lowest priority: the colour kernel floods every CU with streaming waves; the small latency-bound kernels
        // of the binning stages on the caller's stream must win the arbitration
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, least) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&geo_done, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&sh_done, hipEventDisableTiming) != hipSuccess) return false;
        device = dev;
        return true;
    }
};
thread_local ShSide g_sh_side;

struct ProbeSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ensure()
    {
        if (stream) return true;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) return false;
        return true;
    }
};
thread_local ProbeSide g_probe_side;
// side stream of the per-Gaussian backward's 16-wave launch (one per host thread, re-created when the thread's
// current device changes)
struct BwdSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
    bool ensure()
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev == device) return true;
        if (stream) (void)hipStreamDestroy(stream);
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        stream = nullptr; fork = nullptr; join = nullptr; device = -1;
        // HIGHEST priority: the 16-wave workgroups need a whole CU's LDS each; both launches become ready when the blend
        // backward ends, and unless the dispatcher places these first they wait until the plain kernel has drained
        // (rocprofv3, round 3: 274 us "duration" for a launch whose workgroups found an empty list)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) return false;
        device = dev;
        return true;
    }
};
thread_local BwdSide g_bwd_side;

#define FRG_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail(FRG_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// debug mode: synchronise after every stage so a faulting kernel is attributed
// (the reference's CHECK_CUDA, auxiliary.h:166-173)
#define FRG_STAGE(call, name)                                                                               \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ == hipSuccess && debug) e_ = hipStreamSynchronize(stream);                                   \
        if (e_ != hipSuccess) return fail(FRG_EHIP, "stage '%s' failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

// Modes of one forward: each is the per-call value of frg_forward_args when given, else the process-wide option.
struct FwdModes {
    int exact, tight, async_sh;
    int fwd_only = 0;     // frg_forward_args::forward_only (per call only: there is no process-wide form)
    static int pick(int field, int max_value, int fallback) { return field >= 1 && field <= max_value + 1 ? field - 1 : fallback; }
};
int exact_blend();
FwdModes default_modes();

frg::ViewParams make_view(int P, int D, int M, int width, int height, float tan_fovx, float tan_fovy, float scale_modifier, int tight)
{
    (void)P;
    frg::ViewParams vp;
    vp.tan_fovx = tan_fovx; vp.tan_fovy = tan_fovy;
    vp.focal_y = height / (2.0f * tan_fovy);  // rasterizer_impl.cu:222-223
    vp.focal_x = width / (2.0f * tan_fovx);
    vp.scale_modifier = scale_modifier;
    vp.W = width; vp.H = height;
    vp.gx = (width + FRG_TILE - 1) / FRG_TILE; vp.gy = (height + FRG_TILE - 1) / FRG_TILE;
    vp.D = D; vp.M = M;
    vp.tight = tight;
    vp.sparse_sh = 0;
    vp.sh_no_dir = 0;
    return vp;
}

FwdModes default_modes() { return FwdModes{exact_blend(), g_tight_binning.load(), g_async_sh.load(), 0}; }

}  // namespace

extern "C" {

int frg_version(void) { return 2; }
const char* frg_last_error(void) { return g_err; }

int frg_set_option(const char* name, int value)
{
    if (name && strcmp(name, "exact_blend") == 0) {
        int old = exact_blend();
        g_exact_blend.store(value ? 1 : 0);
        return old;
    }
    if (name && strcmp(name, "profile") == 0) return g_profile.exchange(value ? 1 : 0);
    if (name && strcmp(name, "profile_stage") == 0) return g_profile_stage.exchange(value < 0 || value >= ST_COUNT ? -1 : value);
    if (name && strcmp(name, "global_bins") == 0) return g_global_bins.exchange(value ? 1 : 0);
    if (name && strcmp(name, "tight_binning") == 0) return g_tight_binning.exchange(value ? 1 : 0);
    if (name && strcmp(name, "bwd_batch") == 0) return g_bwd_batch.exchange(value == 2 ? 2 : 3);
    if (name && strcmp(name, "bwd_seg_log") == 0) return g_bwd_seg_log.exchange(value >= FRG_BWD_SEG_LOG_MIN && value <= FRG_BWD_SEG_LOG_MAX ? value : 0);
    if (name && strcmp(name, "counter_mailbox") == 0) return g_use_mailbox.exchange(value ? 1 : 0);
    if (name && strcmp(name, "sparse_sh") == 0) return g_sparse_sh.exchange(value ? 1 : 0);
    if (name && strcmp(name, "fwd_prefetch") == 0) { const int old = frg::g_fwd_prefetch; frg::g_fwd_prefetch = value ? 1 : 0; return old; }
    if (name && strcmp(name, "bwd_heavy_first") == 0) return g_bwd_heavy_first.exchange(value ? 1 : 0);
    if (name && strcmp(name, "bwd_waves") == 0) { const int old = frg::g_bwd_waves; frg::g_bwd_waves = value < 0 ? 0 : value; return old; }
    if (name && strcmp(name, "fwd_order") == 0) { const int old = frg::g_fwd_order; frg::g_fwd_order = value ? 1 : 0; return old; }
    if (name && strcmp(name, "sh_dir_in_backward") == 0) return g_sh_no_dir.exchange(value ? 1 : 0);
    if (name && strcmp(name, "clear_image_state") == 0) return g_clear_image_state.exchange(value ? 1 : 0);
    if (name && strcmp(name, "sort_heavy_on_caller") == 0) { const int old = frg::g_sort_heavy_on_caller; frg::g_sort_heavy_on_caller = value ? 1 : 0; return old; }
    // timing-experiment knobs: "ablate" and "probe" make kernels skip work or ignore dependencies (WRONG results), so a
    // stray call must not be able to switch them on -- they exist only in processes started with FROSTING_EXPERIMENTS=1
    if (name && (strcmp(name, "ablate") == 0 || strcmp(name, "probe") == 0 || strcmp(name, "rows_grid") == 0 || strcmp(name, "assume_no_heavy") == 0)) {
        static const bool experiments = [] { const char* e = getenv("FROSTING_EXPERIMENTS"); return e && e[0] == '1'; }();
        if (!experiments)
            return fail(FRG_EINVAL, "option '%s' is a timing experiment (results are wrong by design): start the process with "
                                    "FROSTING_EXPERIMENTS=1 to use it", name);
        if (strcmp(name, "ablate") == 0) return g_ablate.exchange(value);
        if (strcmp(name, "probe") == 0) return g_probe.exchange(value);
        if (strcmp(name, "assume_no_heavy") == 0) return g_assume_no_heavy.exchange(value ? 1 : 0);
        const int old = frg::g_rows_grid; frg::g_rows_grid = value <= 0 ? 0 : value < 8 ? 8 : value; return old;
    }
    if (name && strcmp(name, "async_sh") == 0) return g_async_sh.exchange(value < 0 || value > 3 ? 1 : value);
    if (name && strcmp(name, "fused_small") == 0) return g_fused_small.exchange(value ? 1 : 0);
    if (name && strcmp(name, "fwd_unroll8") == 0) return g_fwd_unroll8.exchange(value < 0 || value > 2 ? 1 : value);
    // tuning: blocks of 64 Gaussians per tile of the combine pass (3, 6, 12 or 24; 0 = chosen from the number of views); same results
    if (name && strcmp(name, "combine_blocks") == 0) { const int old = frg::g_combine_blocks; frg::g_combine_blocks = value < 0 ? 0 : value; return old; }
    return fail(FRG_EINVAL, "unknown option '%s'", name ? name : "(null)");
}

int frg_stage_times(float* ms, int n)
{
    if (!ms || n < ST_COUNT) return fail(FRG_EINVAL, "need room for %d stages", (int)ST_COUNT);
    for (int i = 0; i < n; i++) ms[i] = -1.0f;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return ST_COUNT;
    std::lock_guard<std::mutex> lk(g_timers_mu);
    StageTimers& tm = g_timers[dev];
    if (!tm.init) return ST_COUNT;
    for (int i = 0; i < ST_COUNT; i++) {
        const unsigned cnt = tm.launches[i] < (unsigned)ST_SLOTS ? tm.launches[i] : (unsigned)ST_SLOTS;
        double sum = 0.0;
        unsigned good = 0;
        for (unsigned k = 0; k < cnt; k++) {
            if (hipEventSynchronize(tm.ev[i][k][1]) != hipSuccess) continue;
            float t = -1.0f;
            if (hipEventElapsedTime(&t, tm.ev[i][k][0], tm.ev[i][k][1]) == hipSuccess) { sum += t; good++; }
        }
        if (good) ms[i] = (float)(sum / good);
        tm.launches[i] = 0;
    }
    return ST_COUNT;
}

int frg_get_option(const char* name)
{
    if (name && strcmp(name, "exact_blend") == 0) return exact_blend();
    if (name && strcmp(name, "profile") == 0) return g_profile.load();
    if (name && strcmp(name, "profile_stage") == 0) return g_profile_stage.load();
    if (name && strcmp(name, "global_bins") == 0) return g_global_bins.load();
    if (name && strcmp(name, "tight_binning") == 0) return g_tight_binning.load();
    if (name && strcmp(name, "bwd_batch") == 0) return g_bwd_batch.load();
    if (name && strcmp(name, "bwd_seg_log") == 0) return g_bwd_seg_log.load();
    if (name && strcmp(name, "bwd_waves") == 0) return frg::g_bwd_waves;
    if (name && strcmp(name, "fwd_order") == 0) return frg::g_fwd_order;
    if (name && strcmp(name, "sh_dir_in_backward") == 0) return g_sh_no_dir.load();
    if (name && strcmp(name, "counter_mailbox") == 0) return g_use_mailbox.load();
    if (name && strcmp(name, "sparse_sh") == 0) return g_sparse_sh.load();
    if (name && strcmp(name, "fwd_prefetch") == 0) return frg::g_fwd_prefetch;
    if (name && strcmp(name, "bwd_heavy_first") == 0) return g_bwd_heavy_first.load();
    if (name && strcmp(name, "clear_image_state") == 0) return g_clear_image_state.load();
    if (name && strcmp(name, "sort_heavy_on_caller") == 0) return frg::g_sort_heavy_on_caller;
    if (name && strcmp(name, "async_sh") == 0) return g_async_sh.load();
    if (name && strcmp(name, "fused_small") == 0) return g_fused_small.load();
    if (name && strcmp(name, "fwd_unroll8") == 0) return g_fwd_unroll8.load();
    if (name && strcmp(name, "combine_blocks") == 0) return frg::g_combine_blocks;
    return fail(FRG_EINVAL, "unknown option '%s'", name ? name : "(null)");
}

size_t frg_geometry_bytes(int P) { return frg::GeomState::carve(nullptr, P).bytes; }
size_t frg_image_bytes(int width, int height) { return frg::ImageState::carve(nullptr, width, height, g_global_bins.load() != 0).bytes; }
size_t frg_binning_bytes(int R, int max_tile_count) { return frg::BinningState::carve(nullptr, R, max_tile_count, g_bwd_seg_log.load()).bytes; }
// slots (36 B per instance) ...
static size_t slots_bytes(int R) { return frg::align_up((size_t)(R > 0 ? R : 1) * FRG_SLOT_STRIDE * sizeof(float), 256); }
// ... + the nine per-Gaussian sums a two-call backward (frg_backward_args::phase) keeps between its calls.  (The backward
// blend's work items are listed by the forward, in its own chunks: frg_common.h, BinningState::bwd_full, ImageState::bwd_last.)
static size_t sums_bytes(int P) { return frg::align_up((size_t)(P > 0 ? P : 1) * FRG_SLOT_FLOATS * sizeof(float), 256); }
// ... + one bit per Gaussian, "its sums are not all zero", left by phase 1 for the slot-sum exchange (whole 256-Gaussian workgroups)
static size_t live_mask_words(int P) { return (size_t)(P > 0 ? P : 1) / 64 + 8; }
// ... + the pack's scratch: one row count per group of 256 such words
static size_t live_mask_bytes(int P) { return frg::align_up(live_mask_words(P) * 8, 256); }
// ... + the three view-direction terms per Gaussian that phase 1 leaves for the slot-sum packets
static size_t dir_terms_bytes(int P) { return frg::align_up((size_t)(P > 0 ? P : 1) * 3 * sizeof(float), 256); }
static size_t pack_scratch_bytes(int P) { return frg::align_up((live_mask_words(P) / 256 + 8) * 4, 256); }
size_t frg_backward_workspace_bytes(int P, int R)
{
    return slots_bytes(R) + sums_bytes(P) + live_mask_bytes(P) + pack_scratch_bytes(P) + dir_terms_bytes(P);
}

int frg_geometry_layout_n(int P, long long* out, int n)
{
    frg::GeomState s = frg::GeomState::carve(nullptr, P);
    const long long v[6] = {(long long)(size_t)s.xydr, (long long)(size_t)s.conic_opacity, (long long)(size_t)s.rgb_clamped,
                            (long long)(size_t)s.tiles_touched, (long long)(size_t)s.point_offsets,
                            (long long)(16 * FRG_REC)};   // [5]: byte stride between consecutive Gaussians' float4 of out[0..2]
    for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return 6;
}
void frg_geometry_layout(int P, long long* out) { (void)frg_geometry_layout_n(P, out, 5); }   // the five values of version 1 callers
void frg_image_layout(int width, int height, long long* out)
{
    frg::ImageState s = frg::ImageState::carve(nullptr, width, height, g_global_bins.load() != 0);
    out[0] = (long long)(size_t)s.final_T; out[1] = (long long)(size_t)s.n_contrib; out[2] = (long long)(size_t)s.ranges;
    out[3] = (long long)(size_t)s.tile_count;
}
void frg_binning_layout(int R, int max_tile_count, long long* out)
{
    frg::BinningState s = frg::BinningState::carve(nullptr, R, max_tile_count);
    out[0] = (long long)(size_t)s.point_list; out[1] = (long long)(size_t)s.pairs;
}

int frg_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     unsigned char* present, void* hip_stream)
{
    (void)projmatrix;  // the reference's frustum test only uses the view matrix (auxiliary.h:154)
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!means3D || !viewmatrix || !present) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)hip_stream));
    return FRG_OK;
}

// capacity == 0: the reference's flow, one blocking read-back of the counters between scan and
// scatter.  capacity > 0: no host synchronisation at all -- the binning buffer is sized for
// `capacity` instances up front, launches that depend on the counters use device-side values,
// and the counters travel to a pinned slot that frg_forward_finish() inspects later.
static int forward_impl(frg_alloc_fn geometry_alloc, frg_alloc_fn binning_alloc, frg_alloc_fn image_alloc, void* user,
                        int P, int D, int M, const float* background, int width, int height,
                        const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                        const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                        const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                        float tan_fovx, float tan_fovy, int prefiltered,
                        float* out_color, int* radii, int debug, void* hip_stream, int capacity,
                        const unsigned char* keep_mask = nullptr, const frg::RawInputs* raw = nullptr, const FwdModes* modes = nullptr)
{
    hipStream_t stream = (hipStream_t)hip_stream;
    const FwdModes md = modes ? *modes : default_modes();
    const int exact = md.exact;
    const int seg_forced = g_bwd_seg_log.load();      // (read once: the size asked of the callback and the carve must agree)
    auto bin_bytes = [seg_forced](int R_, int longest) { return frg::BinningState::carve(nullptr, R_, longest, seg_forced).bytes; };
    if (P < 0 || width <= 0 || height <= 0) return fail(FRG_EINVAL, "bad sizes P=%d W=%d H=%d", P, width, height);
    if (!out_color) return fail(FRG_EINVAL, "out_color is null");
    if (P == 0) {  // rasterize_points.cu:68,81: zero image, background not applied
        FRG_HIP(hipMemsetAsync(out_color, 0, (size_t)3 * width * height * sizeof(float), stream));
        return 0;
    }
    const frg::RawInputs rw = raw ? *raw : frg::RawInputs{};
    if (!viewmatrix || !projmatrix || !cam_pos || !background) return fail(FRG_EINVAL, "null required pointer");
    if ((means3D == nullptr) == (rw.shell_logits == nullptr))
        return fail(FRG_EINVAL, "provide exactly one of means3D / shell_logits");
    if (rw.shell_logits && (!rw.shell_verts || !rw.shell_cells))
        return fail(FRG_EINVAL, "shell_logits needs shell_cell_verts and shell_cells");
    if ((opacities == nullptr) == (rw.raw_opacity == nullptr))
        return fail(FRG_EINVAL, "provide exactly one of opacities / raw_opacities");
    if ((shs == nullptr) == (colors_precomp == nullptr))
        return fail(FRG_EINVAL, "provide exactly one of shs / colors_precomp");
    if ((rw.raw_scale == nullptr) != (rw.raw_rot == nullptr))
        return fail(FRG_EINVAL, "raw_scales and raw_rotations come together");
    const bool have_sr = (scales && rotations) || rw.raw_scale;
    if ((scales || rotations) && rw.raw_scale) return fail(FRG_EINVAL, "provide (scales, rotations) or their raw forms, not both");
    if (((scales == nullptr) != (rotations == nullptr)) || have_sr == (cov3D_precomp != nullptr))
        return fail(FRG_EINVAL, "provide exactly one of (scales, rotations) / cov3D_precomp");
    if (shs && (D < 0 || D > 3 || M < (D + 1) * (D + 1)))
        return fail(FRG_EINVAL, "SH degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
    if (!geometry_alloc || !binning_alloc || !image_alloc) return fail(FRG_EINVAL, "null allocation callback");

    frg::ViewParams vp = make_view(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier, md.tight);
    // Will this view see only a part of the model?  With an occlusion mask: yes.  Otherwise: what the previous forward of
    // this thread saw (posted by its scatter) -- the SH pass then streams the rows of the visible Gaussians only.
    vp.sparse_sh = g_sparse_sh.load(std::memory_order_relaxed) && (keep_mask != nullptr || g_mail.sparse_view(P));
    vp.sh_no_dir = (g_sh_no_dir.load(std::memory_order_relaxed) || md.fwd_only) ? 1 : 0;
    const int T = vp.gx * vp.gy;

    char* geom_chunk = geometry_alloc(user, frg_geometry_bytes(P));
    char* img_chunk = image_alloc(user, frg_image_bytes(width, height));
    if (!geom_chunk || !img_chunk) return fail(FRG_EALLOC, "allocation callback returned null");
    note_forward(geom_chunk, exact, md.fwd_only != 0);
    const frg::GeomState g = frg::GeomState::carve(geom_chunk, P);
    const frg::ImageState img = frg::ImageState::carve(img_chunk, width, height, g_global_bins.load() != 0);
    if (!radii) radii = g.internal_radii;   // rasterizer_impl.cu:228-231

    PendingCounters* pend = nullptr;
    if (capacity > 0) {
        bool reused = false;
        pend = g_pending.acquire(img_chunk, &reused);
        if (!pend) return fail(FRG_EHIP, "pinned counter slot / event creation failed");
        // the previous deferred forward on this image buffer reads its counters back on a side stream:
        // that copy must have happened before the counters are cleared again
        if (reused) FRG_HIP(hipStreamWaitEvent(stream, pend->ev, 0));
    }
    // LDS-bins paths: nothing of the image chunk needs clearing in front of the forward -- colsum_kernel zeroes the
    // scatter cursors and the blend's depth marks on its way, every counter is written unconditionally; only the flag
    // of the prefiltered assertion is set-only.  Global bins (more tiles than the LDS holds): the per-tile counts are
    // accumulated with atomics, the whole region is cleared.
    if (!img.lds_bins || g_clear_image_state.load(std::memory_order_relaxed)) FRG_HIP(hipMemsetAsync(img_chunk + img.zero_begin, 0, img.zero_bytes, stream));
    else if (prefiltered || capacity > 0) FRG_HIP(hipMemsetAsync(&img.counters->filtered, 0, sizeof(uint32_t), stream));   // (deferred: frg_forward_finish is told `prefiltered` again)

    frg::FwdInputs in{means3D, scales, rotations, opacities, shs, cov3D_precomp, colors_precomp, viewmatrix, projmatrix, cam_pos};
    in.keep_mask = keep_mask;
    in.raw = rw;
    // SH colours: nothing before the blend needs them, and the stages in between (scan, scatter, sort) leave the
    // HBM nearly idle -- the colour kernel (the largest single stream of the forward, 192 B per visible Gaussian)
    // runs beside them on a side stream; the blend joins it.
    const int sh_mode = (shs != nullptr && !rw.shell_logits) ? md.async_sh : 0;   // 0 inside preprocess | side stream forked after: 1 preprocess, 2 scan, 3 scatter
    const bool defer_sh = sh_mode != 0 && g_sh_side.ensure();
    bool sh_forked = false;
    // an error return between the fork and the join must not leave the side kernel running on the caller's inputs
    struct ShJoin {
        bool armed; hipStream_t side;
        ~ShJoin() { if (armed) (void)hipStreamSynchronize(side); }
    } sh_join{false, g_sh_side.stream};
    auto fork_sh = [&](int at) -> int {
        if (!defer_sh || sh_forked || (at < sh_mode && at < 3)) return FRG_OK;
        sh_forked = true;
        sh_join.armed = true;
        FRG_HIP(hipEventRecord(g_sh_side.geo_done, stream));
        FRG_HIP(hipStreamWaitEvent(g_sh_side.stream, g_sh_side.geo_done, 0));
        {
            StageScope sc_(ST_SH_COLOR, g_sh_side.stream);
            FRG_HIP(frg::launch_sh_color(P, vp, in, radii, g, g_sh_side.stream));
        }
        FRG_HIP(hipEventRecord(g_sh_side.sh_done, g_sh_side.stream));
        if (debug) FRG_HIP(hipStreamSynchronize(g_sh_side.stream));
        return FRG_OK;
    };
    { StageScope sc_(ST_PREPROCESS, stream); FRG_STAGE(frg::launch_preprocess_fwd(P, vp, in, radii, g, img, prefiltered, defer_sh, stream), "preprocess"); }
    { const int rc_ = fork_sh(1); if (rc_ < 0) return rc_; }
    // blocking form: the scan workgroups post the counters into this thread's pinned mailbox (frg_common.h) and the
    // host polls it, instead of a copy kernel + stream synchronisation behind the scan
    frg::Mailbox* mail = (capacity == 0 && !debug && g_use_mailbox.load(std::memory_order_relaxed)) ? g_mail.get() : nullptr;
    uint32_t mail_seq = 0;
    if (mail) { if (++g_mail.seq == 0) g_mail.seq = 1; mail_seq = g_mail.seq; }
    { StageScope sc_(ST_SCAN, stream); FRG_STAGE(frg::launch_scan(P, vp, g, img, (uint32_t)capacity, stream, mail, mail_seq), "scan"); }
    { const int rc_ = fork_sh(2); if (rc_ < 0) return rc_; }

    int index_bits = 1;
    while (index_bits < 32 && (1u << index_bits) < (uint32_t)P) index_bits++;
    int R = capacity;
    if (capacity > 0) {
        // deferred counters: everything below is enqueued without knowing R on the host; the
        // 48-byte read-back rides a side stream so that no later kernel queues behind it
        FRG_HIP(hipEventRecord(pend->scanned, stream));
        FRG_HIP(hipStreamWaitEvent(pend->copy_stream, pend->scanned, 0));
        FRG_HIP(hipMemcpyAsync(pend->host, img.counters, sizeof(frg::Counters), hipMemcpyDeviceToHost, pend->copy_stream));
        FRG_HIP(hipEventRecord(pend->ev, pend->copy_stream));
        char* bin_chunk = binning_alloc(user, bin_bytes(capacity, FRG_SORT_LDS_CAP + 1));
        if (!bin_chunk) return fail(FRG_EALLOC, "binning allocation callback returned null");
        const frg::BinningState b = frg::BinningState::carve(bin_chunk, capacity, FRG_SORT_LDS_CAP + 1, seg_forced);
        FRG_STAGE(frg::launch_sort_plan(T, nullptr, img.counters->class_count, img.class_tiles, img.ranges, b.big_plan, (uint32_t)capacity, stream), "sort plan");
        { StageScope sc_(ST_SCATTER, stream); FRG_STAGE(frg::launch_scatter(P, vp, radii, g, img, b, stream, g_ablate.load()), "scatter"); }
        { const int rc_ = fork_sh(3); if (rc_ < 0) return rc_; }
        { StageScope sc_(ST_SORT, stream); FRG_STAGE(frg::launch_tile_sort(T, nullptr, g_pending.have_hint ? g_pending.last_class_count : nullptr, img.counters->class_count, img.class_tiles, img.ranges, b.pairs, b.pairs_tmp, b.big_hist, b.big_plan, (uint32_t)capacity, 0, index_bits, b.point_list, stream), "sort"); }
        if (defer_sh) { FRG_HIP(hipStreamWaitEvent(stream, g_sh_side.sh_done, 0)); sh_join.armed = false; }
        StageScope sc_(ST_BLEND_FWD, stream);
        if (exact)
            FRG_STAGE(frg::launch_blend_fwd_exact(vp, g, img, b, background, out_color, stream), "blend");
        else
            FRG_STAGE(frg::launch_blend_fwd_fast(vp, g, img, b, background, out_color, stream), "blend");
        return R;
    }

    // the single host synchronisation of the op (rasterizer_impl.cu:280-281)
    frg::Counters c;
    frg::BinningState b;
    bool have_counters = false, early = false;
    if (mail) {
        // Stage 1: the instance count (posted by the chunk scan, ~40 us before the scan stage ends at C3).  The binning
        // buffer is sized for every sort path -- the longest tile list is not known yet -- and the scatter is enqueued
        // while the reorder still runs.  Stage 2: the tile scan's counters (the sort's grids).
        if (mailbox_wait(&mail->seq_r, mail_seq, stream)) {
            const uint32_t r = mail->num_rendered;
            if (r > 0x7fffffffu) return fail(FRG_EINVAL, "num_rendered overflows int32");
            if (r > 0) {
                R = (int)r;
                char* bin_chunk = binning_alloc(user, bin_bytes(R, FRG_SORT_LDS_CAP + 1));
                if (!bin_chunk) return fail(FRG_EALLOC, "binning allocation callback returned null");
                b = frg::BinningState::carve(bin_chunk, R, FRG_SORT_LDS_CAP + 1, seg_forced);
                if (g_mail.long_lists)
                    FRG_STAGE(frg::launch_sort_plan(T, nullptr, img.counters->class_count, img.class_tiles, img.ranges, b.big_plan, (uint32_t)R, stream, 1), "sort plan");
                { StageScope sc_(ST_SCATTER, stream); FRG_STAGE(frg::launch_scatter(P, vp, radii, g, img, b, stream, g_ablate.load(), mail, mail_seq), "scatter"); }
                { const int rc_ = fork_sh(3); if (rc_ < 0) return rc_; }
                note_heavy_post(geom_chunk, mail, mail_seq);
                g_mail.last_P = P; g_mail.last_seq = mail_seq;
                early = true;
            }
            if (mailbox_wait(&mail->seq_c, mail_seq, stream)) { c = mail->c; have_counters = true; }
        }
        if (!have_counters) g_mail.failed = true, g_mail.host = nullptr;    // (the pinned block is left to the process)
    }
    if (!have_counters) {
        frg::Counters* host = pinned_counters();
        if (!host) return fail(FRG_EHIP, "hipHostMalloc failed");
        FRG_HIP(hipMemcpyAsync(host, img.counters, sizeof(frg::Counters), hipMemcpyDeviceToHost, stream));
        FRG_HIP(hipStreamSynchronize(stream));
        c = *host;
    }
    if (prefiltered && c.filtered)
        return fail(FRG_EFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
    if (c.num_rendered > 0x7fffffffu) return fail(FRG_EINVAL, "num_rendered overflows int32");
    R = (int)c.num_rendered;
    const int max_tile = (int)c.max_tile_count;
    note_rendered(geom_chunk, R);

    if (!early) {
        char* bin_chunk = binning_alloc(user, bin_bytes(R, max_tile));
        if (!bin_chunk) return fail(FRG_EALLOC, "binning allocation callback returned null");
        b = frg::BinningState::carve(bin_chunk, R, max_tile, seg_forced);
    }
    const bool forked_plan = early && g_mail.long_lists;
    if (mail) g_mail.long_lists = c.class_count[4] > 0;

    // small frames (at most 2^20 instances: C2 has 350 000 in 2 500 lists of 140): the lists of up to 512 entries are sorted by the
    // forward blend's own workgroups (blend_impl.h FUSED) -- one launch and the point_list round trip less in a step that is a
    // chain of short launches; the longest-first tile order (class lists) is what the fused form walks
    const bool fused_small = g_fused_small.load() != 0 && R > 0 && R <= (1 << 20) && frg::g_fwd_order != 0 && !(g_probe.load() & 1);
    const bool probe_fwd = (g_probe.load() & 1) && R > 0 && !exact && g_probe_side.ensure();
    if (R > 0) {
        FRG_STAGE(frg::launch_sort_plan(T, c.class_count, img.counters->class_count, img.class_tiles, img.ranges, b.big_plan, (uint32_t)R, stream, forked_plan ? 2 : 0), "sort plan");
        if (!early) {
            { StageScope sc_(ST_SCATTER, stream); FRG_STAGE(frg::launch_scatter(P, vp, radii, g, img, b, stream, g_ablate.load()), "scatter"); }
            { const int rc_ = fork_sh(3); if (rc_ < 0) return rc_; }
        }
        if (probe_fwd) {
            FRG_HIP(hipEventRecord(g_probe_side.fork, stream));
            FRG_HIP(hipStreamWaitEvent(g_probe_side.stream, g_probe_side.fork, 0));
            FRG_HIP(frg::launch_blend_fwd_fast(vp, g, img, b, background, out_color, g_probe_side.stream));
            FRG_HIP(hipEventRecord(g_probe_side.join, g_probe_side.stream));
        }
        { StageScope sc_(ST_SORT, stream); FRG_STAGE(frg::launch_tile_sort(T, c.class_count, nullptr, img.counters->class_count, img.class_tiles, img.ranges, b.pairs, b.pairs_tmp, b.big_hist, b.big_plan, (uint32_t)R, max_tile, index_bits, b.point_list, stream, fused_small), "sort"); }
        if (probe_fwd) { FRG_HIP(hipStreamWaitEvent(stream, g_probe_side.join, 0)); return R; }
    } else {
        // point_offsets must still be defined for backward
        FRG_STAGE(frg::launch_scatter(P, vp, radii, g, img, b, stream), "scatter");
        { const int rc_ = fork_sh(3); if (rc_ < 0) return rc_; }
    }
    if (defer_sh) { FRG_HIP(hipStreamWaitEvent(stream, g_sh_side.sh_done, 0)); sh_join.armed = false; }
    {
        StageScope sc_(ST_BLEND_FWD, stream);
        // A frame that does not fill the GPU (fewer than 1280 instances per tile of the image on average) and whose longest list
        // is several times its mean list -- the limb of a shell seen from outside (C4: 792 per tile, 6 267 against a mean of
        // 1 366 over the active tiles) -- walks eight entries per trip: its launch is stall-bound inside the waves of its long
        // tiles (0.240 -> 0.222 ms).  A full frame is bound by instruction issue and keeps four, uniform (C3: 0.203 -> 0.243 with
        // eight) or clustered (the skew scene, 2 670 per tile: 0.186 -> 0.218).  Same bits either way.
        bool long_lists = false;
        if (R > 0 && g_fwd_unroll8.load() != 0) {
            uint32_t active = 0;
            for (int k = 0; k < FRG_SORT_CLASSES; k++) active += c.class_count[k];
            long_lists = g_fwd_unroll8.load() == 2 ||
                         ((double)max_tile >= 3.5 * (double)R / (double)(active ? active : 1u) && (double)R < 1280.0 * (double)T);
        }
        if (exact)
            FRG_STAGE(frg::launch_blend_fwd_exact(vp, g, img, b, background, out_color, stream, md.fwd_only != 0, fused_small, long_lists), "blend");
        else
            FRG_STAGE(frg::launch_blend_fwd_fast(vp, g, img, b, background, out_color, stream, md.fwd_only != 0, fused_small, long_lists), "blend");
    }
    return R;
}

int frg_forward(frg_alloc_fn geometry_alloc, frg_alloc_fn binning_alloc, frg_alloc_fn image_alloc, void* user,
                int P, int D, int M, const float* background, int width, int height,
                const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                float tan_fovx, float tan_fovy, int prefiltered,
                float* out_color, int* radii, int debug, void* hip_stream)
{
    return forward_impl(geometry_alloc, binning_alloc, image_alloc, user, P, D, M, background, width, height, means3D, shs,
                        colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                        cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, debug, hip_stream, 0);
}

int frg_forward_deferred(frg_alloc_fn geometry_alloc, frg_alloc_fn binning_alloc, frg_alloc_fn image_alloc, void* user,
                         int P, int D, int M, const float* background, int width, int height,
                         const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                         const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                         const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                         float tan_fovx, float tan_fovy, int prefiltered,
                         float* out_color, int* radii, int instance_capacity, void* hip_stream)
{
    if (instance_capacity <= 0) return fail(FRG_EINVAL, "instance_capacity must be positive");
    return forward_impl(geometry_alloc, binning_alloc, image_alloc, user, P, D, M, background, width, height, means3D, shs,
                        colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix,
                        cam_pos, tan_fovx, tan_fovy, prefiltered, out_color, radii, 0, hip_stream, instance_capacity);
}

int frg_forward_ex(const frg_forward_args* a)
{
    // four generations of the struct: up to keep_mask (version 1 callers), with the raw-parameter fields, with the
    // per-call modes, with forward_only
    const size_t v1 = offsetof(frg_forward_args, raw_opacities), v2 = offsetof(frg_forward_args, exact_blend),
                 v3 = offsetof(frg_forward_args, forward_only);
    if (!a || (a->struct_size != sizeof(frg_forward_args) && a->struct_size != v1 && a->struct_size != v2 && a->struct_size != v3))
        return fail(FRG_EINVAL, "frg_forward_args: struct_size %zu, this library expects %zu (or %zu, %zu, %zu)", a ? a->struct_size : (size_t)0,
                    sizeof(frg_forward_args), v3, v2, v1);
    if (a->instance_capacity < 0) return fail(FRG_EINVAL, "instance_capacity < 0");
    frg::RawInputs rw;
    if (a->struct_size >= v2) {
        rw.raw_opacity = a->raw_opacities; rw.raw_scale = a->raw_scales; rw.raw_rot = a->raw_rotations;
        rw.shell_logits = a->shell_logits; rw.shell_verts = a->shell_cell_verts; rw.shell_cells = a->shell_cells;
    }
    FwdModes md = default_modes();
    if (a->struct_size == sizeof(frg_forward_args)) {
        if (a->forward_only < 0 || a->forward_only > 1) return fail(FRG_EINVAL, "frg_forward_args: forward_only must be 0 or 1");
        if (a->forward_only && a->instance_capacity > 0)
            return fail(FRG_EINVAL, "frg_forward_args: forward_only with deferred counters (instance_capacity > 0) is not offered");
        md.fwd_only = a->forward_only;
    }
    if (a->struct_size >= v3) {
        if (a->exact_blend < 0 || a->exact_blend > 2 || a->tight_binning < 0 || a->tight_binning > 2 || a->async_sh < 0 ||
            a->async_sh > 4 || a->shell_bary_mode < 0 || a->shell_bary_mode > 1)
            return fail(FRG_EINVAL, "frg_forward_args: mode out of range (exact_blend %d, tight_binning %d, async_sh %d, shell_bary_mode %d)",
                        a->exact_blend, a->tight_binning, a->async_sh, a->shell_bary_mode);
        md.exact = FwdModes::pick(a->exact_blend, 1, md.exact);
        md.tight = FwdModes::pick(a->tight_binning, 1, md.tight);
        md.async_sh = FwdModes::pick(a->async_sh, 3, md.async_sh);
        rw.bary_mode = a->shell_bary_mode;
    }
    return forward_impl(a->geometry_alloc, a->binning_alloc, a->image_alloc, a->user, a->P, a->D, a->M, a->background,
                        a->width, a->height, a->means3D, a->shs, a->colors_precomp, a->opacities, a->scales,
                        a->scale_modifier, a->rotations, a->cov3D_precomp, a->viewmatrix, a->projmatrix, a->cam_pos,
                        a->tan_fovx, a->tan_fovy, a->prefiltered, a->out_color, a->radii,
                        a->instance_capacity > 0 ? 0 : a->debug, a->hip_stream, a->instance_capacity, a->keep_mask, &rw, &md);
}

int frg_forward_finish(const char* image_buffer, int prefiltered, int* num_rendered)
{
    PendingCounters* p = g_pending.find(image_buffer);
    if (!p) return fail(FRG_EINVAL, "no deferred forward is pending for this image buffer on this thread");
    FRG_HIP(hipEventSynchronize(p->ev));
    const frg::Counters c = *p->host;
    p->key = nullptr;
    if (num_rendered) *num_rendered = (int)(c.num_rendered > 0x7fffffffu ? 0x7fffffffu : c.num_rendered);
    if (c.num_rendered > 0x7fffffffu) return fail(FRG_EINVAL, "num_rendered overflows int32");
    if (c.overflow)
        return fail(FRG_ECAPACITY, "%u instances exceed the instance capacity of the deferred forward: nothing was "
                                   "rasterized, repeat the view with a larger capacity", c.num_rendered);
    for (int k = 0; k < FRG_SORT_CLASSES; k++) g_pending.last_class_count[k] = c.class_count[k];
    g_pending.have_hint = true;
    if (prefiltered && c.filtered)
        return fail(FRG_EFILTER, "Point is filtered although prefiltered is set. This shouldn't happen!");
    return FRG_OK;
}

}  // extern "C"

static int backward_impl(int P, int D, int M, int R, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 char* workspace, size_t workspace_bytes, int debug, void* hip_stream,
                 const frg::RawInputs& rw, float* dL_dshell_logits, float* dL_dshell_verts, int exact_mode = 0, int phase = 0,
                 unsigned char* row_live = nullptr, int range_first = 0, int range_count = 0)
{
    hipStream_t stream = (hipStream_t)hip_stream;
    if (P < 0 || R < 0 || width <= 0 || height <= 0) return fail(FRG_EINVAL, "bad sizes");
    if (P == 0) return FRG_OK;
    if (!geom_buffer || !binning_buffer || !image_buffer || !dL_dpix || !background || !viewmatrix || !projmatrix || !campos)
        return fail(FRG_EINVAL, "null required pointer");
    if ((means3D == nullptr) == (rw.shell_logits == nullptr)) return fail(FRG_EINVAL, "provide exactly one of means3D / shell_logits");
    if (rw.shell_logits && (!rw.shell_verts || !rw.shell_cells || !dL_dshell_logits))
        return fail(FRG_EINVAL, "shell_logits needs shell_cell_verts, shell_cells and dL_dshell_logits");
    if (!dL_dmean2D || !dL_dopacity || !dL_dmean3D) return fail(FRG_EINVAL, "null gradient output");
    // intermediates of the chain may be left out when the caller has no use for them: dL_dcolor when the SH rows are
    // written (it is then only the factor of dL_dsh), dL_dcov3D when the covariance comes from scales / rotations
    if (!dL_dcolor && !(shs && dL_dsh)) return fail(FRG_EINVAL, "dL_dcolor may only be NULL when shs and dL_dsh are given");
    if (!dL_dcov3D && cov3D_precomp) return fail(FRG_EINVAL, "dL_dcov3D may only be NULL without cov3D_precomp");
    if ((rw.raw_scale == nullptr) != (rw.raw_rot == nullptr)) return fail(FRG_EINVAL, "raw_scales and raw_rotations come together");
    if (((scales && (!dL_dscale || !dL_drot || !rotations))) || (rw.raw_scale && (!dL_dscale || !dL_drot)))
        return fail(FRG_EINVAL, "null gradient output for a provided input");
    if (workspace_bytes < frg_backward_workspace_bytes(P, R) || !workspace)
        return fail(FRG_EALLOC, "workspace too small: need %zu bytes", frg_backward_workspace_bytes(P, R));
    // R sizes the slots and the backward blend's item list: fewer than the forward rendered would overrun them.  (More is
    // fine -- a deferred forward's capacity: where the forward's checkpoints lie in the binning chunk is taken from what the
    // forward stamped, Counters::carved_R, not from R.)
    // The arithmetic of this backward's blend pass (the backward recomputes its forward's alpha, T and contributor tests: the
    // same arithmetic keeps them consistent) and whether that forward kept anything for a backward were stamped into the
    // image chunk by the forward's blend kernel (Counters::fwd_flags): they travel with the buffers.  The host's notes are
    // keyed by ADDRESS -- a buffer copied to an address some earlier forward used would inherit that forward's note -- so
    // they decide nothing a stamp can contradict:
    //   * arithmetic: what the caller states (frg_backward_args::exact_blend; the Python layer carries it in its autograd
    //     ctx), else BOTH instantiations are launched and each leaves at once unless the stamp names it (exact = -1 below);
    //   * forward_only: a note that says "an ordinary forward" lets the call through (a forward_only stamp then still leaves
    //     the kernels without work); anything else -- no note, or a note that says forward_only -- is settled by reading the
    //     stamp back, one blocking 4-byte copy, which also tells the arithmetic.
    //   * R: a note that says the forward rendered MORE instances than this call's R (slots and item lists would overrun) is
    //     checked against the stamped count the same way before the call is refused.
    int exact = exact_mode == 0 ? -1 : FwdModes::pick(exact_mode, 1, 0);
    const int noted_rendered = forward_rendered(geom_buffer);
    if ((forward_was_forward_only(geom_buffer) != 0 || (noted_rendered >= 0 && R < noted_rendered)) && phase != 2) {
        frg::Counters* host = pinned_counters();
        if (!host) return fail(FRG_EHIP, "hipHostMalloc failed");
        const frg::ImageState img0 = frg::ImageState::carve(image_buffer, width, height, false);
        FRG_HIP(hipMemcpyAsync(host, img0.counters, sizeof(frg::Counters), hipMemcpyDeviceToHost, stream));
        FRG_HIP(hipStreamSynchronize(stream));
        const uint32_t flags = host->fwd_flags;
        if (!(flags & FRG_FWD_STAMPED))
            return fail(FRG_EINVAL, "the image buffer carries no forward's stamp: these are not the buffers of a completed frg_forward");
        if (flags & FRG_FWD_ONLY)
            return fail(FRG_EINVAL, "the forward that filled these buffers was called with forward_only = 1: it kept nothing for a backward");
        if ((uint32_t)R < host->num_rendered)
            return fail(FRG_EINVAL, "R = %d, but the forward that filled this geometry buffer rendered %u instances", R, host->num_rendered);
        if (exact < 0) exact = (flags & FRG_FWD_EXACT) ? 1 : 0;
    }
    (void)colors_precomp;  // forward copied precomputed colours into the geometry state

    // Nothing here depends on the process-wide binning options: every field of the three chunks that the
    // backward reads is carved from (P, W, H, R) alone (the option-dependent matrices of the binning stage
    // come last in the image chunk), and the kernels take the forward's binning mode from the counters it
    // stamped.  "exact_blend" only selects the arithmetic of this backward's own blend pass.
    frg::ViewParams vp = make_view(P, D, M, width, height, tan_fovx, tan_fovy, scale_modifier, 0);
    const frg::GeomState g = frg::GeomState::carve(geom_buffer, P);
    const frg::ImageState img = frg::ImageState::carve(image_buffer, width, height, false);
    const frg::BinningState b = frg::BinningState::carve(binning_buffer, R, 0);
    float* slots = reinterpret_cast<float*>(workspace);
    float* sums = reinterpret_cast<float*>(workspace + slots_bytes(R));
    unsigned long long* live_masks = phase == 1 ? reinterpret_cast<unsigned long long*>(workspace + slots_bytes(R) + sums_bytes(P)) : nullptr;
    float* dir_terms = reinterpret_cast<float*>(workspace + slots_bytes(R) + sums_bytes(P) + live_mask_bytes(P) + pack_scratch_bytes(P));
    if (phase < 0 || phase > 2) return fail(FRG_EINVAL, "frg_backward_args: phase %d (0 whole | 1 blend + slot sums | 2 the rest)", phase);
    if (!radii) radii = g.internal_radii;   // rasterizer_impl.cu:375-377

    frg::FwdInputs in{means3D, scales, rotations, nullptr, shs, cov3D_precomp, colors_precomp, viewmatrix, projmatrix, campos};
    in.raw = rw;
    frg::BwdOutputs out{dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot};
    out.dL_dshell_logits = dL_dshell_logits;
    out.dL_dshell_verts = dL_dshell_verts;
    out.row_live = row_live;
    const int pbw_flags = phase == 1 ? FRG_PBW_SUMS_ONLY : phase == 2 ? FRG_PBW_FROM_SUMS : 0;
    if (phase == 2) {     // the sums are in the workspace: one launch, no slot reduction, hence no 16-wave form either
        if (!phase1_matches(workspace, geom_buffer, image_buffer, P, R))
            return fail(FRG_EINVAL, "backward phase 2 without a matching phase 1 on this workspace (same P, R, geometry and image buffers)");
        StageScope sc_(ST_PREPROCESS_BWD, stream);
        FRG_STAGE(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags | FRG_PBW_NO_HEAVY_LAUNCH, false, stream, sums), "preprocess_bwd (phase 2)");
        return FRG_OK;
    }
    if (phase == 1) note_phase1(workspace, geom_buffer, image_buffer, P, R);
    const bool ranged = range_count > 0;
    if (ranged) {
        if (phase != 1) return fail(FRG_EINVAL, "frg_backward_args: a range is offered with phase 1 only (phase %d)", phase);
        if (range_first < 0 || range_first % 256 != 0 || (long long)range_first + range_count > P)
            return fail(FRG_EINVAL, "frg_backward_args: range [%d, +%d) of %d Gaussians (range_first: a multiple of 256)", range_first, range_count, P);
    }
    const bool probe_bwd = (g_probe.load() & 2) && g_probe_side.ensure();
    if (probe_bwd) {   // timing experiment: the per-Gaussian backward beside the blend (it reads the previous frame's slots)
        FRG_HIP(hipEventRecord(g_probe_side.fork, stream));
        FRG_HIP(hipStreamWaitEvent(g_probe_side.stream, g_probe_side.fork, 0));
        FRG_HIP(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags, false, g_probe_side.stream, sums));
        FRG_HIP(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags, true, g_probe_side.stream, sums));
        FRG_HIP(hipEventRecord(g_probe_side.join, g_probe_side.stream));
    }
    if (!ranged || range_first == 0) {
        StageScope sc_(ST_BLEND_BWD, stream);
        if (exact != 0)
            FRG_STAGE(frg::launch_blend_bwd_exact(vp, g, img, b, background, dL_dpix, slots, (uint32_t)R, g_bwd_batch.load(), stream, exact < 0), "blend_bwd");
        if (exact <= 0)
            FRG_STAGE(frg::launch_blend_bwd_fast(vp, g, img, b, background, dL_dpix, slots, (uint32_t)R, g_bwd_batch.load(), stream, exact < 0), "blend_bwd");
    }
    if (probe_bwd) { FRG_HIP(hipStreamWaitEvent(stream, g_probe_side.join, 0)); return FRG_OK; }
    if (ranged) {      // phase 1 in pieces: the plain kernel over this range, whatever its waves own (no 16-wave side launch)
        StageScope sc_(ST_PREPROCESS_BWD, stream);
        FRG_STAGE(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags | FRG_PBW_NO_HEAVY_LAUNCH, false, stream, sums,
                                             live_masks, dir_terms, range_first, range_count), "preprocess_bwd (phase 1, range)");
        return FRG_OK;
    }
    {
        // the 16-wave form for the Gaussians that own thousands of slots runs on a side stream beside the plain kernel
        // (usually its workgroups find an empty list and leave)
        StageScope sc_(ST_PREPROCESS_BWD, stream);
        // the forward's scatter posted how many waves of Gaussians need the 16-wave form (Mailbox::heavy): none, usually
        const int heavy = g_assume_no_heavy.load(std::memory_order_relaxed) ? 0 : debug ? -1 : heavy_waves_posted(geom_buffer);
        const bool skip_heavy = heavy == 0;
        const bool side = !skip_heavy && !debug && g_bwd_side.ensure();
        // Known to exist: the few 16-wave workgroups go on the CALLER's stream and start at once on an empty GPU, the
        // plain kernel follows on the side stream a cross-queue hop later and fills the rest -- behind the plain kernel's
        // 47 k waves the 1024-thread workgroups waited for a whole free CU and ran mostly after it (clustered scene:
        // 0.41 -> 0.33 ms).  Unknown (no post): the 16-wave form on the high-priority side stream, as before.
        const bool heavy_first = side && heavy > 0 && g_bwd_heavy_first.load(std::memory_order_relaxed);
        hipStream_t hs = side ? g_bwd_side.stream : stream;
        hipStream_t s_heavy = heavy_first ? stream : hs, s_plain = heavy_first ? hs : stream;
        if (side) { FRG_HIP(hipEventRecord(g_bwd_side.fork, stream)); FRG_HIP(hipStreamWaitEvent(hs, g_bwd_side.fork, 0)); }
        if (!skip_heavy)
            FRG_STAGE(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags, true, s_heavy, sums, live_masks, dir_terms), "preprocess_bwd (long runs)");
        FRG_STAGE(frg::launch_preprocess_bwd(P, vp, in, radii, g, img, slots, out, g_ablate.load(), pbw_flags | (skip_heavy ? FRG_PBW_NO_HEAVY_LAUNCH : 0), false, s_plain, sums, live_masks, dir_terms), "preprocess_bwd");
        if (side) { FRG_HIP(hipEventRecord(g_bwd_side.join, hs)); FRG_HIP(hipStreamWaitEvent(stream, g_bwd_side.join, 0)); }
    }
    return FRG_OK;
}

extern "C" {

int frg_backward(int P, int D, int M, int R, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos,
                 float tan_fovx, float tan_fovy, const int* radii,
                 char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                 float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 char* workspace, size_t workspace_bytes, int debug, void* hip_stream)
{
    return backward_impl(P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                         cov3D_precomp, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, radii, geom_buffer, binning_buffer,
                         image_buffer, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                         dL_dscale, dL_drot, workspace, workspace_bytes, debug, hip_stream, frg::RawInputs{}, nullptr, nullptr);
}

int frg_backward_ex(const frg_backward_args* a)
{
    // five generations of the struct: up to shell_*, + exact_blend / shell_bary_mode, + phase, + row_live, + range_first / range_count
    const size_t b1 = offsetof(frg_backward_args, exact_blend), b2 = offsetof(frg_backward_args, phase), b3 = offsetof(frg_backward_args, row_live),
                 b4 = offsetof(frg_backward_args, range_first);
    if (!a || (a->struct_size != sizeof(frg_backward_args) && a->struct_size != b1 && a->struct_size != b2 && a->struct_size != b3 && a->struct_size != b4))
        return fail(FRG_EINVAL, "frg_backward_args: struct_size %zu, this library expects %zu (or %zu, %zu, %zu, %zu)", a ? a->struct_size : (size_t)0,
                    sizeof(frg_backward_args), b4, b3, b2, b1);
    frg::RawInputs rw;
    rw.raw_opacity = a->raw_opacities; rw.raw_scale = a->raw_scales; rw.raw_rot = a->raw_rotations;
    rw.shell_logits = a->shell_logits; rw.shell_verts = a->shell_cell_verts; rw.shell_cells = a->shell_cells;
    int exact_mode = 0;
    const int phase = a->struct_size >= b3 ? a->phase : 0;
    unsigned char* row_live = a->struct_size >= b4 ? a->row_live : nullptr;
    const bool has_range = a->struct_size == sizeof(frg_backward_args);
    if (a->struct_size >= b2) {
        if (a->exact_blend < 0 || a->exact_blend > 2 || a->shell_bary_mode < 0 || a->shell_bary_mode > 1)
            return fail(FRG_EINVAL, "frg_backward_args: mode out of range (exact_blend %d, shell_bary_mode %d)", a->exact_blend, a->shell_bary_mode);
        exact_mode = a->exact_blend;
        rw.bary_mode = a->shell_bary_mode;
    }
    return backward_impl(a->P, a->D, a->M, a->R, a->background, a->width, a->height, a->means3D, a->shs, a->colors_precomp,
                         a->scales, a->scale_modifier, a->rotations, a->cov3D_precomp, a->viewmatrix, a->projmatrix, a->campos,
                         a->tan_fovx, a->tan_fovy, a->radii, a->geom_buffer, a->binning_buffer, a->image_buffer, a->dL_dpix,
                         a->dL_dmean2D, a->dL_dconic, a->dL_dopacity, a->dL_dcolor, a->dL_dmean3D, a->dL_dcov3D, a->dL_dsh,
                         a->dL_dscale, a->dL_drot, a->workspace, a->workspace_bytes, a->debug, a->hip_stream, rw,
                         a->dL_dshell_logits, a->dL_dshell_cell_verts, exact_mode, phase, row_live, has_range ? a->range_first : 0,
                         has_range ? a->range_count : 0);
}

int frg_sh_color_grad(int P, const char* geom_buffer, const int* radii, const float* dL_dcolors,
                      float* out_drgb, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!geom_buffer || !radii || !dL_dcolors || !out_drgb) return fail(FRG_EINVAL, "null pointer");
    const frg::GeomState g = frg::GeomState::carve(const_cast<char*>(geom_buffer), P);
    FRG_HIP(frg::launch_sh_color_grad(P, g, radii, dL_dcolors, out_drgb, (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_sh_grad_from_views(int P, int D, int M, int n_views, const float* means3D,
                           const float* campos, long long campos_stride,
                           const float* drgb, long long view_stride, float* dL_dsh, void* hip_stream)
{
    if (P < 0 || n_views < 0 || D < 0 || D > 3) return fail(FRG_EINVAL, "bad sizes P=%d views=%d D=%d", P, n_views, D);
    if (M < (D + 1) * (D + 1)) return fail(FRG_EINVAL, "degree %d needs %d coefficients, got M=%d", D, (D + 1) * (D + 1), M);
    if (P == 0) return FRG_OK;
    if (!means3D || !dL_dsh || (n_views > 0 && (!campos || !drgb))) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_sh_grad_from_views(P, D, M, n_views, means3D, campos, campos_stride, drgb, view_stride, dL_dsh,
                                           (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_pack_grad_rows(int P, const float* dL_dmeans3D, const float* dL_dscales, const float* dL_drotations, const float* dL_dopacity,
                       const float* drgb, float* rows, long long capacity_rows, unsigned int* count, void* hip_stream)
{
    if (P < 0 || capacity_rows < 0 || capacity_rows > 0xffffffffLL) return fail(FRG_EINVAL, "bad sizes P=%d capacity=%lld", P, capacity_rows);
    if (!count) return fail(FRG_EINVAL, "null pointer");
    if (P == 0) { FRG_HIP(hipMemsetAsync(count, 0, sizeof(unsigned int), (hipStream_t)hip_stream)); return FRG_OK; }
    if (!dL_dmeans3D || !dL_dscales || !dL_drotations || !dL_dopacity || !drgb || (!rows && capacity_rows > 0)) return fail(FRG_EINVAL, "null pointer");
    if (reinterpret_cast<uintptr_t>(rows) % 16 != 0) return fail(FRG_EINVAL, "rows must be 16-byte aligned");
    FRG_HIP(frg::launch_pack_grad_rows(P, dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacity, drgb, rows, (unsigned int)capacity_rows, count,
                                       (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_scatter_grad_rows(long long n_rows, int P, const float* rows, float* dL_dmeans3D, float* dL_dscales, float* dL_drotations,
                          float* dL_dopacity, float* drgb_dense, void* hip_stream)
{
    if (P < 0 || n_rows < 0 || n_rows > 0xffffffffLL) return fail(FRG_EINVAL, "bad sizes P=%d rows=%lld", P, n_rows);
    if (n_rows == 0 || P == 0) return FRG_OK;
    if (!rows || !dL_dmeans3D || !dL_dscales || !dL_drotations || !dL_dopacity) return fail(FRG_EINVAL, "null pointer");
    if (reinterpret_cast<uintptr_t>(rows) % 16 != 0) return fail(FRG_EINVAL, "rows must be 16-byte aligned");
    FRG_HIP(frg::launch_scatter_grad_rows((unsigned int)n_rows, P, rows, dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacity, drgb_dense,
                                          (hipStream_t)hip_stream));
    return FRG_OK;
}

size_t frg_sum_packet_bytes(int n_gaussians, long long capacity_rows)
{
    if (n_gaussians < 0 || capacity_rows < 0) return 0;
    return frg::sum_packet_bytes((size_t)n_gaussians, (size_t)capacity_rows);
}

int frg_pack_sum_rows(int P, int R, int first, int count, char* workspace, size_t workspace_bytes, const float* drgb_masked,
                      const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy,
                      int width, int height, float scale_modifier, int D, void* packet, size_t packet_bytes, long long capacity_rows,
                      void* hip_stream)
{
    if (P < 0 || R < 0 || first < 0 || count < 0 || (long long)first + count > P || first % 64 != 0)
        return fail(FRG_EINVAL, "bad range: P=%d first=%d (a multiple of 64) count=%d", P, first, count);
    if (capacity_rows < 0 || capacity_rows > 0x7fffffffLL) return fail(FRG_EINVAL, "capacity_rows %lld", capacity_rows);
    if (!workspace || workspace_bytes < frg_backward_workspace_bytes(P, R)) return fail(FRG_EALLOC, "not the workspace of a backward with P=%d R=%d", P, R);
    if (!packet || packet_bytes < frg_sum_packet_bytes(count, capacity_rows) || reinterpret_cast<uintptr_t>(packet) % 16 != 0)
        return fail(FRG_EALLOC, "packet: need %zu bytes, 16-byte aligned", frg_sum_packet_bytes(count, capacity_rows));
    if (!drgb_masked || !viewmatrix || !projmatrix || !campos) return fail(FRG_EINVAL, "null pointer");
    if (width <= 0 || height <= 0 || D < 0 || D > 3) return fail(FRG_EINVAL, "bad view: %dx%d degree %d", width, height, D);
    const float* sums = reinterpret_cast<const float*>(workspace + slots_bytes(R));
    const unsigned long long* masks = reinterpret_cast<const unsigned long long*>(workspace + slots_bytes(R) + sums_bytes(P));
    uint32_t* group_tot = reinterpret_cast<uint32_t*>(workspace + slots_bytes(R) + sums_bytes(P) + live_mask_bytes(P));
    const float* dir_terms = reinterpret_cast<const float*>(workspace + slots_bytes(R) + sums_bytes(P) + live_mask_bytes(P) + pack_scratch_bytes(P));
    const frg::SumCamera cam{tan_fovx, tan_fovy, scale_modifier, width, height, D};
    FRG_HIP(frg::launch_pack_sum_rows(first, count, (uint32_t)capacity_rows, masks, sums, dir_terms, drgb_masked, cam, viewmatrix, projmatrix, campos, packet,
                                      group_tot, (hipStream_t)hip_stream));
    return FRG_OK;
}

size_t frg_combine_workspace_bytes(int n_views, long long capacity_rows)
{
    if (n_views < 0 || capacity_rows < 0) return 0;
    return frg::combine_workspace_bytes(n_views, (size_t)capacity_rows);
}

int frg_backward_combine(const frg_combine_args* a)
{
    if (!a || a->struct_size != sizeof(frg_combine_args))
        return fail(FRG_EINVAL, "frg_combine_args: struct_size %zu, this library expects %zu", a ? a->struct_size : (size_t)0, sizeof(frg_combine_args));
    if (a->P < 0 || a->first < 0 || a->count < 0 || (long long)a->first + a->count > a->P || a->first % 64 != 0)
        return fail(FRG_EINVAL, "bad range: P=%d first=%d (a multiple of 64) count=%d", a->P, a->first, a->count);
    if (a->n_views < 1 || a->n_views > 16) return fail(FRG_EINVAL, "1..16 views expected, got %d", a->n_views);
    if (a->M != 16) return fail(FRG_EINVAL, "the combine pass takes SH rows of 16 coefficients (M = %d)", a->M);
    if (a->capacity_rows >= 65535LL * 256) return fail(FRG_EINVAL, "capacity_rows %lld: a packet holds fewer than 2^24 rows (cut the Gaussians into more ranges)", a->capacity_rows);
    if (a->count == 0) return FRG_OK;
    if (!a->packets || a->packet_stride_bytes % 16 != 0 || a->packet_stride_bytes < frg_sum_packet_bytes(a->count, a->capacity_rows))
        return fail(FRG_EINVAL, "packets: stride %zu, a packet of %d Gaussians and %lld rows has %zu bytes", a->packet_stride_bytes, a->count,
                    a->capacity_rows, frg_sum_packet_bytes(a->count, a->capacity_rows));
    if ((a->means3D == nullptr) || !a->shs) return fail(FRG_EINVAL, "means3D and shs are required (shell-bound centres are not offered here)");
    if (!a->workspace || a->workspace_bytes < frg_combine_workspace_bytes(a->n_views, a->capacity_rows) || reinterpret_cast<uintptr_t>(a->workspace) % 16 != 0)
        return fail(FRG_EALLOC, "workspace: need %zu bytes, 16-byte aligned", frg_combine_workspace_bytes(a->n_views, a->capacity_rows));
    if ((a->opacities == nullptr) == (a->raw_opacities == nullptr)) return fail(FRG_EINVAL, "provide exactly one of opacities / raw_opacities");
    const bool raw_sr = a->raw_scales && a->raw_rotations;
    if (raw_sr == (a->scales && a->rotations) || (a->raw_scales == nullptr) != (a->raw_rotations == nullptr) || (a->scales == nullptr) != (a->rotations == nullptr))
        return fail(FRG_EINVAL, "provide (scales, rotations) or (raw_scales, raw_rotations)");
    if (!a->dL_dmean3D || !a->dL_dscale || !a->dL_drot || !a->dL_dopacity || !a->dL_dsh) return fail(FRG_EINVAL, "null gradient output");
    if ((reinterpret_cast<uintptr_t>(a->shs) | reinterpret_cast<uintptr_t>(a->dL_dsh) | reinterpret_cast<uintptr_t>(a->dL_drot) |
         reinterpret_cast<uintptr_t>(a->rotations) | reinterpret_cast<uintptr_t>(a->packets)) % 16 != 0)
        return fail(FRG_EINVAL, "shs, rotations, dL_dsh, dL_drot and the packets must be 16-byte aligned");
    frg::FwdInputs in{a->means3D, a->scales, a->rotations, a->opacities, a->shs, nullptr, nullptr, nullptr, nullptr, nullptr};
    in.raw.raw_opacity = a->raw_opacities; in.raw.raw_scale = a->raw_scales; in.raw.raw_rot = a->raw_rotations;
    frg::BwdOutputs out{nullptr, nullptr, a->dL_dopacity, nullptr, a->dL_dmean3D, nullptr, a->dL_dsh, a->dL_dscale, a->dL_drot};
    FRG_HIP(frg::launch_backward_combine(a->first, a->count, a->n_views, a->packets, a->packet_stride_bytes, (uint32_t)a->capacity_rows, in, out,
                                         a->status, a->status_seq, a->row_live, a->workspace, (hipStream_t)a->hip_stream));
    return FRG_OK;
}

static thread_local frg::AdamRows g_adam_rows;     // set by frg_adam_step_rows around its call of frg_adam_step
static thread_local bool g_adam_shard = false;     // frg_adam_step_shard: segment ends before the first element (negative) are expected

int frg_adam_step(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                  const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                  const int* segment_head, const float* segment_head_lrs, int n_segments,
                  double beta1, double beta2, double eps, int step, float grad_scale, void* hip_stream)
{
    if (n < 0 || step < 1) return fail(FRG_EINVAL, "bad sizes n=%lld step=%d", n, step);
    if (n_segments < 1 || n_segments > FRG_ADAM_MAX_SEGMENTS || !segment_ends || !segment_lrs)
        return fail(FRG_EINVAL, "1..%d segments expected, got %d", FRG_ADAM_MAX_SEGMENTS, n_segments);
    for (int k = 1; k < n_segments; k++)
        if (segment_ends[k] < segment_ends[k - 1]) return fail(FRG_EINVAL, "segment ends must not decrease");
    if ((segment_ends[0] < 0 && !g_adam_shard) || segment_ends[n_segments - 1] != n) return fail(FRG_EINVAL, "the last segment must end at n");
    if (n == 0) return FRG_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq) return fail(FRG_EINVAL, "null pointer");
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 != 0)
        return fail(FRG_EINVAL, "the four arrays must be 16-byte aligned");
    if ((n + 3) / 4 / 256 + 1 > 0x7fffffffLL) return fail(FRG_EINVAL, "n too large for one launch");
    // Python-float arithmetic of torch/optim/adam.py: doubles, rounded to float where a tensor op takes them
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    frg::AdamSegments seg;
    seg.count = n_segments;
    for (int k = 0; k < FRG_ADAM_MAX_SEGMENTS; k++) {
        seg.end[k] = k < n_segments ? segment_ends[k] : n;
        seg.step_size[k] = k < n_segments ? (float)((double)segment_lrs[k] / bc1) : 0.0f;
        const bool sub = k < n_segments && segment_period && segment_head && segment_head_lrs && segment_period[k] > 0;
        if (sub && (segment_head[k] < 0 || segment_head[k] > segment_period[k]))
            return fail(FRG_EINVAL, "segment %d: head %d outside its period %d", k, segment_head[k], segment_period[k]);
        seg.period[k] = sub ? segment_period[k] : 0;
        seg.head[k] = sub ? segment_head[k] : 0;
        seg.head_step_size[k] = sub ? (float)((double)segment_head_lrs[k] / bc1) : 0.0f;
    }
    const float w1 = (float)(1.0 - beta1);      // betas arrive as doubles: 1 - beta is formed before rounding to float,
    const float omb2 = (float)(1.0 - beta2);    // as the Python floats of torch/optim/adam.py are
    const float inv_bc2_sqrt = 1.0f / (float)std::sqrt(bc2);   // ATen divides by a scalar as a multiplication by its float reciprocal
    FRG_HIP(frg::launch_adam_step(n, params, grads, exp_avg, exp_avg_sq, seg, w1, (float)beta2, omb2, inv_bc2_sqrt, (float)eps, grad_scale,
                                  (hipStream_t)hip_stream, g_adam_rows.live ? &g_adam_rows : nullptr));
    return FRG_OK;
}

int frg_adam_step_shard(long long n, long long first, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                        const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                        const int* segment_head, const float* segment_head_lrs, int n_segments,
                        double beta1, double beta2, double eps, int step, float grad_scale, void* hip_stream)
{
    // elements [first, first + n) of the flat layout, the four arrays pointing at element `first`: the segment table is
    // shifted by `first` behind a zero-length segment that ends at -first -- the kernel takes an element's segment as the last
    // one whose predecessor ends at or before it and its offset from that end, so negative ends give every element of the
    // shard its true segment and its true phase inside it (the DC / rest split of the SH rows)
    if (n < 0 || first < 0 || first % 4 != 0) return fail(FRG_EINVAL, "bad shard: n=%lld first=%lld (a multiple of 4 elements)", n, first);
    if (n_segments < 1 || n_segments + 1 > FRG_ADAM_MAX_SEGMENTS || !segment_ends || !segment_lrs)
        return fail(FRG_EINVAL, "a sharded step takes 1..%d segments, got %d", FRG_ADAM_MAX_SEGMENTS - 1, n_segments);
    if (first + n > segment_ends[n_segments - 1]) return fail(FRG_EINVAL, "the shard ends behind the last segment");
    long long ends[FRG_ADAM_MAX_SEGMENTS];
    float lrs[FRG_ADAM_MAX_SEGMENTS], hlrs[FRG_ADAM_MAX_SEGMENTS];
    int per[FRG_ADAM_MAX_SEGMENTS], head[FRG_ADAM_MAX_SEGMENTS];
    ends[0] = -first; lrs[0] = 0.0f; hlrs[0] = 0.0f; per[0] = 0; head[0] = 0;
    for (int k = 0; k < n_segments; k++) {
        ends[k + 1] = segment_ends[k] - first;
        lrs[k + 1] = segment_lrs[k];
        per[k + 1] = segment_period ? segment_period[k] : 0;
        head[k + 1] = segment_head ? segment_head[k] : 0;
        hlrs[k + 1] = segment_head_lrs ? segment_head_lrs[k] : 0.0f;
    }
    ends[n_segments] = n;       // the shard ends inside (or at the end of) the last segment it reaches; later ones are cut off
    for (int k = 1; k <= n_segments; k++) if (ends[k] > n) ends[k] = n;
    g_adam_shard = true;
    const int rc = frg_adam_step(n, params, grads, exp_avg, exp_avg_sq, ends, lrs, per, head, hlrs, n_segments + 1, beta1, beta2, eps, step,
                                 grad_scale, hip_stream);
    g_adam_shard = false;
    return rc;
}

int frg_adam_step_rows(long long n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                       const long long* segment_ends, const float* segment_lrs, const int* segment_period,
                       const int* segment_head, const float* segment_head_lrs, int n_segments,
                       double beta1, double beta2, double eps, int step, float grad_scale,
                       const unsigned char* row_live, int P, const int* segment_width, void* hip_stream)
{
    if (!row_live) return frg_adam_step(n, params, grads, exp_avg, exp_avg_sq, segment_ends, segment_lrs, segment_period, segment_head,
                                        segment_head_lrs, n_segments, beta1, beta2, eps, step, grad_scale, hip_stream);
    if (P < 0 || !segment_width) return fail(FRG_EINVAL, "row_live needs P >= 0 and segment_width");
    if (n_segments < 1 || n_segments > FRG_ADAM_MAX_SEGMENTS) return fail(FRG_EINVAL, "1..%d segments expected, got %d", FRG_ADAM_MAX_SEGMENTS, n_segments);
    for (int k = 0; k < n_segments; k++) {
        const long long begin = k ? segment_ends[k - 1] : 0;
        if (segment_width[k] < 0 || (long long)segment_width[k] * P > segment_ends[k] - begin)
            return fail(FRG_EINVAL, "segment %d: %d elements per Gaussian x %d Gaussians exceed its %lld elements", k, segment_width[k], P, segment_ends[k] - begin);
        if (segment_width[k] > 0 && segment_ends[k] - begin > 0xffffffffLL)
            return fail(FRG_EINVAL, "segment %d: a per-Gaussian segment of a masked step holds at most 2^32 elements", k);
        // the kernel takes four consecutive elements per thread and lets ONE mask look-up stand for all four when the rows of
        // their segment are a multiple of four elements long: true only if the segment starts on a multiple of four elements
        if (segment_width[k] > 0 && segment_width[k] % 4 == 0 && begin % 4 != 0)
            return fail(FRG_EINVAL, "segment %d: rows of %d elements must begin on a multiple of 4 elements (begins at %lld)", k, segment_width[k], begin);
    }
    g_adam_rows.live = row_live; g_adam_rows.P = P;
    for (int k = 0; k < FRG_ADAM_MAX_SEGMENTS; k++) {
        g_adam_rows.width[k] = k < n_segments ? segment_width[k] : 0;
        g_adam_rows.magic[k] = g_adam_rows.width[k] > 1 ? (unsigned int)(0x100000000ull / (unsigned long long)g_adam_rows.width[k]) : 0u;
    }
    const int rc = frg_adam_step(n, params, grads, exp_avg, exp_avg_sq, segment_ends, segment_lrs, segment_period, segment_head,
                                 segment_head_lrs, n_segments, beta1, beta2, eps, step, grad_scale, hip_stream);
    g_adam_rows.live = nullptr;
    return rc;
}

size_t frg_photometric_workspace_bytes(int channels, int width, int height)
{
    if (channels <= 0 || width <= 0 || height <= 0) return 0;
    return frg::photometric_workspace_bytes(channels, width, height);
}

int frg_photometric_loss(int channels, int width, int height, const float* image, const float* target,
                         const float* window11, float lambda_dssim, float* loss, float* dL_dimage,
                         char* workspace, size_t workspace_bytes, void* hip_stream)
{
    if (channels <= 0 || width <= 0 || height <= 0) return fail(FRG_EINVAL, "bad sizes C=%d W=%d H=%d", channels, width, height);
    if (!image || !target || !window11 || !loss) return fail(FRG_EINVAL, "null pointer");
    if (!workspace || workspace_bytes < frg_photometric_workspace_bytes(channels, width, height))
        return fail(FRG_EALLOC, "workspace too small: need %zu bytes", frg_photometric_workspace_bytes(channels, width, height));
    if ((long long)channels * ((width + 15) / 16) * ((height + 15) / 16) > 0x7fffffffLL) return fail(FRG_EINVAL, "image too large");
    FRG_HIP(frg::launch_photometric(channels, width, height, image, target, window11, lambda_dssim, loss, dL_dimage, workspace,
                                    (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_activate(int P, const float* raw_opacity, const float* raw_scale, const float* raw_rot,
                 float* opacity, float* scale, float* rot, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!raw_opacity || !raw_scale || !raw_rot || !opacity || !scale || !rot) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_activate(P, raw_opacity, raw_scale, raw_rot, opacity, scale, rot, (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_activate_backward(int P, const float* opacity, const float* scale, const float* raw_rot,
                          float* g_opacity, float* g_scale, float* g_rot, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!opacity || !scale || !raw_rot || !g_opacity || !g_scale || !g_rot) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_activate_bwd(P, opacity, scale, raw_rot, g_opacity, g_scale, g_rot, (hipStream_t)hip_stream));
    return FRG_OK;
}

size_t frg_knn_workspace_bytes(int P) { return P > 0 ? frg::knn_workspace_bytes(P) : 0; }

int frg_knn_mean_dist2(int P, const float* points, float* mean_dist2, char* workspace, size_t workspace_bytes, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!points || !mean_dist2) return fail(FRG_EINVAL, "null pointer");
    if (!workspace || workspace_bytes < frg_knn_workspace_bytes(P))
        return fail(FRG_EALLOC, "workspace too small: need %zu bytes", frg_knn_workspace_bytes(P));
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) return fail(FRG_EINVAL, "workspace must be 256-byte aligned");
    FRG_HIP(frg::launch_knn(P, points, mean_dist2, workspace, (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_shell_points(int P, const float* bary_logits, const float* cell_verts, const long long* point_cell_indices,
                     float* points, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!bary_logits || !cell_verts || !point_cell_indices || !points) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_shell_points(P, bary_logits, cell_verts, point_cell_indices, points, (hipStream_t)hip_stream));
    return FRG_OK;
}

int frg_shell_points_backward(int P, const float* bary_logits, const float* cell_verts, const long long* point_cell_indices,
                              const float* dL_dpoints, float* dL_dlogits, void* hip_stream)
{
    if (P < 0) return fail(FRG_EINVAL, "P < 0");
    if (P == 0) return FRG_OK;
    if (!bary_logits || !cell_verts || !point_cell_indices || !dL_dpoints || !dL_dlogits) return fail(FRG_EINVAL, "null pointer");
    FRG_HIP(frg::launch_shell_points_bwd(P, bary_logits, cell_verts, point_cell_indices, dL_dpoints, dL_dlogits,
                                         (hipStream_t)hip_stream));
    return FRG_OK;
}

}  // extern "C"
