// Error reporting + small runtime utilities of the C-ABI library (see include/lvsr_hip.h).
#include "common.h"
#include "persist.h"
#include "graph_cache.h"
#include <list>
#include <mutex>
#include <unordered_map>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void lvsr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int lvsr_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        lvsr_set_error("%s: %s", what, hipGetErrorString(e));
        return LVSR_ERR_HIP;
    }
    return LVSR_OK;
}

// ---- tuning knobs ---------------------------------------------------------------------------------
#include "lvsr_hip.h"
#include <atomic>
static std::atomic<int> g_knobs[LVSR_KNOB_COUNT];
int lvsr_knob(int knob) { return (knob >= 0 && knob < LVSR_KNOB_COUNT) ? g_knobs[knob].load(std::memory_order_relaxed) : 0; }
int lvsr_max_cluster_wgs() {
    const int forced = lvsr_knob(LVSR_KNOB_MAX_CLUSTER_WGS);
    if (forced > 0) return forced;
    static std::atomic<int> cached{0};
    int v = cached.load(std::memory_order_relaxed);
    if (v == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        v = cus;
        cached.store(v, std::memory_order_relaxed);
    }
    const int reserve = lvsr_knob(LVSR_KNOB_CLUSTER_RESERVE);
    return reserve > 0 && reserve < v ? v - reserve : v;
}

// ---- graph cache (LRU, bounded) ----------------------------------------------------------------
namespace {
struct Entry { hipGraphExec_t exec; std::list<std::string>::iterator it; };
std::mutex g_mu;
std::unordered_map<std::string, Entry> g_graphs;
std::list<std::string> g_lru;
const size_t kMaxGraphs = 256;
}  // namespace

hipGraphExec_t lvsr_graph_lookup(const GraphKey& key, bool* known_bad) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto f = g_graphs.find(key.bytes);
    if (f == g_graphs.end()) { *known_bad = false; return nullptr; }
    g_lru.erase(f->second.it);
    g_lru.push_front(key.bytes);
    f->second.it = g_lru.begin();
    *known_bad = (f->second.exec == nullptr);
    return f->second.exec;
}

void lvsr_graph_store(const GraphKey& key, hipGraphExec_t exec) {
    std::lock_guard<std::mutex> lk(g_mu);
    while (g_graphs.size() >= kMaxGraphs) {
        auto victim = g_graphs.find(g_lru.back());
        if (victim->second.exec) (void)hipGraphExecDestroy(victim->second.exec);
        g_graphs.erase(victim);
        g_lru.pop_back();
    }
    g_lru.push_front(key.bytes);
    g_graphs[key.bytes] = Entry{exec, g_lru.begin()};
}

bool lvsr_stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st == hipStreamCaptureStatusActive;
}

static thread_local std::string g_region_key;
static thread_local int g_suppress = 0;
bool lvsr_graphs_suppressed() { return g_suppress != 0; }

extern "C" {
// ---- graph regions: several library calls (and whatever else the host enqueues on the stream) as ONE cached graph --
// begin: 1 = a cached graph for `key` was launched, the caller skips its enqueue code; 0 = capture started, the caller
// enqueues as usual and then calls lvsr_region_end; 2 = capture impossible (e.g. legacy stream), enqueue eagerly, no end.
int lvsr_region_begin(void* stream, const char* key, long long key_bytes) {
    LVSR_REQUIRE(key && key_bytes > 0, "lvsr_region_begin: empty key");
    hipStream_t s = (hipStream_t)stream;
    GraphKey k("region:");
    k.add(key, (size_t)key_bytes);
    bool bad = false;
    hipGraphExec_t exec = lvsr_graph_lookup(k, &bad);
    if (exec) {
        if (hipGraphLaunch(exec, s) != hipSuccess) {
            lvsr_set_error("lvsr_region_begin: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
            return LVSR_ERR_HIP;
        }
        return 1;
    }
    if (bad || lvsr_stream_is_capturing(s)) return 2;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        lvsr_graph_store(k, nullptr);
        return 2;
    }
    g_region_key = k.bytes;
    return 0;
}

// end: instantiate, cache under the key of the matching begin and launch.  keep = 0 discards the capture instead (the
// caller noticed something that must not be replayed, e.g. a device allocation inside the region) and marks the key as
// not capturable; the caller then enqueues again eagerly.
int lvsr_region_end(void* stream, int keep) {
    hipStream_t s = (hipStream_t)stream;
    LVSR_REQUIRE(!g_region_key.empty(), "lvsr_region_end without lvsr_region_begin");
    GraphKey k("");
    k.bytes = g_region_key;
    g_region_key.clear();
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (e == hipSuccess && graph && keep) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (!keep) {
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        lvsr_graph_store(k, nullptr);
        return LVSR_OK;
    }
    if (e != hipSuccess || !exec) {
        lvsr_set_error("lvsr_region_end: capture failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        lvsr_graph_store(k, nullptr);
        return LVSR_ERR_HIP;
    }
    lvsr_graph_store(k, exec);
    if (hipGraphLaunch(exec, s) != hipSuccess) {
        lvsr_set_error("lvsr_region_end: hipGraphLaunch failed: %s", hipGetErrorString(hipGetLastError()));
        return LVSR_ERR_HIP;
    }
    return LVSR_OK;
}

// While on, entry points called with use_graph = 1 launch eagerly and cache nothing (used for the one eager pass that
// precedes the capture of a region: its inner time-loop graphs would never be replayed).
void lvsr_graph_suppress(int on) { g_suppress = on; }

// Drop every cached graph (call when workspaces are freed / pointers may be recycled).
void lvsr_graph_clear(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    g_graphs.clear();
    g_lru.clear();
}
int lvsr_graph_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (auto& kv : g_graphs) n += kv.second.exec != nullptr;
    return n;
}
const char* lvsr_last_error(void) { return g_err; }
int lvsr_set_knob(int knob, int value) {
    LVSR_REQUIRE(knob >= 0 && knob < LVSR_KNOB_COUNT, "lvsr_set_knob: unknown knob %d", knob);
#ifndef LVSR_PROBES
    LVSR_REQUIRE(knob != LVSR_KNOB_PERSIST_FLAGS || (value & PF_WRONG_RESULT_BITS) == 0,
                 "lvsr_set_knob: persist_flags %d contains timing-ablation bits (1, 8, 16, 128: wrong results); they exist only in the "
                 "probe build of the library (csrc/build.py --probes)", value);
#endif
    g_knobs[knob].store(value, std::memory_order_relaxed);
    return LVSR_OK;
}
int lvsr_get_knob(int knob) { return lvsr_knob(knob); }
int lvsr_abi_version(void) { return 1; }
}
