// hipGraph assembled node by node from the launch schedule's data dependencies.
//
// Stream capture turns a schedule into a graph only along the streams it was enqueued on, and capturing on three or more streams
// crashes hipStreamEndCapture on ROCm 7.2 for the DLA / HRNet schedules (two work).  Here every launch of a plan is captured ALONE
// (one single-kernel capture on a private stream -> a child graph) and added to a parent graph with explicit edges to the launches
// it depends on (RAW / WAR / WAW, computed by the host from the launch records).  The instantiated graph then carries the full DAG:
// HRNet's four resolution branches, the IDAUp projections and the six head branches are independent paths the runtime may overlap.
#include <vector>
#include "common.h"

struct cp_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    std::vector<hipGraphNode_t> nodes;
    std::vector<hipGraph_t> children;
    bool capturing = false;
};

extern "C" int cp_graph_create(cp_graph** out)
{
    CP_CHECK_ARG(out, "graph_create: null pointer");
    cp_graph* g = new cp_graph();
    if (hipGraphCreate(&g->graph, 0) != hipSuccess || hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking) != hipSuccess) {
        delete g;
        cp_set_error("graph_create: hipGraphCreate / hipStreamCreate failed");
        return 2;
    }
    *out = g;
    return 0;
}

// the stream the NEXT launch has to be enqueued on (between cp_graph_begin_node and cp_graph_end_node)
extern "C" void* cp_graph_stream(cp_graph* g) { return g ? (void*)g->cap : nullptr; }

extern "C" int cp_graph_begin_node(cp_graph* g)
{
    CP_CHECK_ARG(g && !g->capturing && !g->exec, "graph_begin_node: bad state");
    hipError_t e = hipStreamBeginCapture(g->cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { cp_set_error("graph_begin_node: %s", hipGetErrorString(e)); return 2; }
    g->capturing = true;
    return 0;
}

// ends the capture of one launch and adds it behind the nodes `deps` (ids returned by earlier calls); *node_id = its id
extern "C" int cp_graph_end_node(cp_graph* g, const int* deps, int ndeps, int* node_id)
{
    CP_CHECK_ARG(g && g->capturing && (ndeps == 0 || deps), "graph_end_node: bad state");
    hipGraph_t child = nullptr;
    hipError_t e = hipStreamEndCapture(g->cap, &child);
    g->capturing = false;
    if (e != hipSuccess || !child) { cp_set_error("graph_end_node: end capture: %s", hipGetErrorString(e)); return 2; }
    std::vector<hipGraphNode_t> d;
    for (int i = 0; i < ndeps; ++i) {
        CP_CHECK_ARG(deps[i] >= 0 && deps[i] < (int)g->nodes.size(), "graph_end_node: dependency %d out of range", deps[i]);
        d.push_back(g->nodes[deps[i]]);
    }
    hipGraphNode_t node = nullptr;
    // a single-kernel capture is re-added as a plain KERNEL node of the parent (same parameters; the captured graph stays alive as
    // their owner): child-graph nodes replay measurably slower (hrnet B=8: 9.6 vs 7.0 ms per step)
    size_t n = 0;
    hipGraphNode_t only = nullptr;
    hipGraphNodeType ty = hipGraphNodeTypeEmpty;
    if (hipGraphGetNodes(child, nullptr, &n) == hipSuccess && n == 1) {
        n = 1;
        if (hipGraphGetNodes(child, &only, &n) != hipSuccess || hipGraphNodeGetType(only, &ty) != hipSuccess) ty = hipGraphNodeTypeEmpty;
    }
    hipKernelNodeParams kp;
    if (ty == hipGraphNodeTypeKernel && hipGraphKernelNodeGetParams(only, &kp) == hipSuccess)
        e = hipGraphAddKernelNode(&node, g->graph, d.empty() ? nullptr : d.data(), d.size(), &kp);
    else
        e = hipGraphAddChildGraphNode(&node, g->graph, d.empty() ? nullptr : d.data(), d.size(), child);
    if (e != hipSuccess) { cp_set_error("graph_end_node: add node: %s", hipGetErrorString(e)); (void)hipGraphDestroy(child); return 2; }
    g->children.push_back(child);
    g->nodes.push_back(node);
    if (node_id) *node_id = (int)g->nodes.size() - 1;
    return 0;
}

extern "C" int cp_graph_instantiate(cp_graph* g)
{
    CP_CHECK_ARG(g && !g->capturing && !g->exec, "graph_instantiate: bad state");
    hipError_t e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { cp_set_error("graph_instantiate: %s", hipGetErrorString(e)); g->exec = nullptr; return 2; }
    return 0;
}

extern "C" int cp_graph_launch(cp_graph* g, void* stream)
{
    CP_CHECK_ARG(g && g->exec, "graph_launch: not instantiated");
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) { cp_set_error("graph_launch: %s", hipGetErrorString(e)); return 2; }
    return 0;
}

extern "C" int cp_graph_destroy(cp_graph* g)
{
    if (!g) return 0;
    if (g->capturing) { hipGraph_t tmp = nullptr; (void)hipStreamEndCapture(g->cap, &tmp); if (tmp) (void)hipGraphDestroy(tmp); }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    for (hipGraph_t c : g->children) (void)hipGraphDestroy(c);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->cap) (void)hipStreamDestroy(g->cap);
    delete g;
    return 0;
}
