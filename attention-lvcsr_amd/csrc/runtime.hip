// Error reporting + small runtime utilities of the C-ABI library (see include/lvsr_hip.h).
#include "common.h"
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

// ---- graph cache (LRU, bounded) ----------------------------------------------------------------
namespace {
struct Entry { hipGraphExec_t exec; std::list<std::string>::iterator it; };
std::mutex g_mu;
std::unordered_map<std::string, Entry> g_graphs;
std::list<std::string> g_lru;
const size_t kMaxGraphs = 96;
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

extern "C" {
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
int lvsr_abi_version(void) { return 1; }
}
