// hipGraph cache for launch-bound time loops.  A recurrent layer is thousands of tiny dependent
// kernels; eager launches are host-bound (~3.5 us each) while graph replay paces them on the device
// (~1.5 us per dependent boundary).  Graphs bake in pointers and sizes, so they are keyed by the
// full argument block; the host side keeps its workspaces stable per shape so replays hit.
#pragma once
#include "common.h"
#include "lvsr_hip.h"
#include <string.h>
#include <string>
#include <vector>

struct GraphKey {
    std::string bytes;
    // the tuning knobs select kernel variants, so their values belong to every key: a graph captured under one setting is never
    // replayed under another
    explicit GraphKey(const char* tag) : bytes(tag) {
        for (int k = 0; k < LVSR_KNOB_COUNT; ++k) { const int v = lvsr_knob(k); add(&v, sizeof(v)); }
    }
    void add(const void* p, size_t n) { bytes.append((const char*)p, n); }
};

// Returns the cached executable graph for `key` or nullptr; see runtime.hip.
hipGraphExec_t lvsr_graph_lookup(const GraphKey& key, bool* known_bad);
void lvsr_graph_store(const GraphKey& key, hipGraphExec_t exec);   // exec == nullptr marks "cannot capture"

bool lvsr_stream_is_capturing(hipStream_t s);
bool lvsr_graphs_suppressed();

template <class F>
int lvsr_run_graph(hipStream_t s, int use_graph, const GraphKey& key, F&& enqueue, const char* what) {
    // inside an lvsr_region_begin/end capture just record the launches; lvsr_graph_suppress(1): eager launches only
    if (!use_graph || lvsr_graphs_suppressed() || lvsr_stream_is_capturing(s)) {
        enqueue();
        return lvsr_check_launch(what);
    }
    bool bad = false;
    hipGraphExec_t exec = lvsr_graph_lookup(key, &bad);
    if (!exec && !bad) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) {
            (void)hipGetLastError();
            lvsr_graph_store(key, nullptr);
        } else {
            enqueue();
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(s, &graph);
            if (e == hipSuccess && graph) {
                e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
            }
            if (e != hipSuccess || !exec) {
                (void)hipGetLastError();
                exec = nullptr;
                lvsr_graph_store(key, nullptr);
            } else {
                lvsr_graph_store(key, exec);
            }
        }
    }
    if (exec) {
        if (hipGraphLaunch(exec, s) != hipSuccess) {
            lvsr_set_error("%s: hipGraphLaunch failed: %s", what, hipGetErrorString(hipGetLastError()));
            return LVSR_ERR_HIP;
        }
        return LVSR_OK;
    }
    enqueue();
    return lvsr_check_launch(what);
}
