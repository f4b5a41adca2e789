// A frame step is a fixed list of kernel launches whose arguments never change between frames (every
// varying quantity - offsets, masks, tokens - lives in device memory).  The list is captured once
// into a hipGraph and replayed, which is what replaces the reference's CUDAGraphed wrappers
// (utils/compile.py:190-280; compression.py:224-229; lm.py:630-632).  MMI_NO_GRAPH=1 runs the same
// list eagerly, mirroring the reference's NO_CUDA_GRAPH switch (utils/compile.py:169-175).
#pragma once
#include "mmi_common.h"
#include <stdlib.h>

// Launch-list recorder: while a program runs for the first time (eagerly, or under stream capture) every MMI_LAUNCH is
// logged as "site<TAB>kernel", where the site is the label of the op that issued it ("L.ffn_in", "dep.out_proj", ...).
// profiles/ joins that list with a rocprofv3 kernel trace by position inside the step (scripts/rocpd_sites.py), which is how
// the per-site durations of kernels that share one name (k_gemm_xp serves six GEMM shapes) are recomputed.
void mmi_record_begin(std::vector<std::string>* log);
void mmi_record_site(const char* site);
void mmi_record_bytes(long bytes);      // the NEXT launch streams this many weight bytes: third column of its line
void mmi_record_end();

struct MmiProgram {
    std::vector<std::function<int(hipStream_t)>> ops;
    std::vector<std::string> sites;             // one label per op
    std::vector<long> op_bytes;                 // weight bytes the op streams (GEMMs), 0 otherwise
    std::function<void(size_t, bool, hipStream_t)> tap;   // profiling: called before (true) and after (false) every eagerly run op
    std::vector<std::string> launch_log;        // "site\tkernel" per launch, in launch order (filled by the first run)
    std::string site_ = "-";                    // label given to the ops added from now on
    bool logged = false;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // the same list captured as two graphs, ops [0, cut) and [cut, end), for callers that put something between the halves
    // (run_split: an event record that another stream waits on)
    hipGraph_t graph_a = nullptr, graph_b = nullptr;
    hipGraphExec_t exec_a = nullptr, exec_b = nullptr;
    size_t cut_ = 0;

    void site(const std::string& label) { site_ = label; }
    void add(std::function<int(hipStream_t)> f, long bytes = 0) { ops.push_back(std::move(f)); sites.push_back(site_); op_bytes.push_back(bytes); }

    int run_eager(hipStream_t s) {
        const bool rec = !logged;
        if (rec) mmi_record_begin(&launch_log);
        int rc = MMI_OK;
        for (size_t i = 0; i < ops.size() && !rc; ++i) {
            if (rec) mmi_record_site(sites[i].c_str());
            if (tap) tap(i, true, s);
            rc = ops[i](s);
            if (tap) tap(i, false, s);
        }
        if (rec) { mmi_record_end(); logged = true; }
        return rc;
    }

    // ops [i0, i1) launched eagerly (the step cut into segments around per-step host callbacks)
    int run_range(hipStream_t s, size_t i0, size_t i1) {
        int rc = MMI_OK;
        for (size_t i = i0; i < i1 && i < ops.size() && !rc; ++i) rc = ops[i](s);
        return rc;
    }

    int run(hipStream_t s, bool use_graph, hipStream_t capture_stream) {
        if (!use_graph) return run_eager(s);
        if (!exec) {
            MMI_HIP_CHECK(hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal));
            int rc = run_eager(capture_stream);
            hipError_t e = hipStreamEndCapture(capture_stream, &graph);
            if (rc) return rc;
            if (e != hipSuccess) return mmi_fail(MMI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            MMI_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        }
        MMI_HIP_CHECK(hipGraphLaunch(exec, s));
        return MMI_OK;
    }

    int capture_range(hipStream_t capture_stream, size_t i0, size_t i1, hipGraph_t* g, hipGraphExec_t* e) {
        MMI_HIP_CHECK(hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal));
        int rc = run_range(capture_stream, i0, i1);
        hipError_t err = hipStreamEndCapture(capture_stream, g);
        if (rc || err != hipSuccess) {          // an op failed while capturing: the half-built graph is not kept
            if (err == hipSuccess && *g) hipGraphDestroy(*g);
            *g = nullptr;
            if (rc) return rc;
            return mmi_fail(MMI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(err));
        }
        err = hipGraphInstantiate(e, *g, nullptr, nullptr, 0);
        if (err != hipSuccess) {
            hipGraphDestroy(*g);
            *g = nullptr;
            *e = nullptr;
            return mmi_fail(MMI_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(err));
        }
        return MMI_OK;
    }

    // ops [0, cut) - between(s) - ops [cut, end): as two graph launches (or eagerly), `between` enqueued on s in the middle
    int run_split(hipStream_t s, bool use_graph, hipStream_t capture_stream, size_t cut, const std::function<int(hipStream_t)>& between) {
        int rc;
        if (!use_graph || !logged) {            // the first run of a program is the eager one that records the launch list
            const bool rec = !logged;
            if (rec) mmi_record_begin(&launch_log);
            rc = MMI_OK;
            for (size_t i = 0; i < ops.size() && !rc; ++i) {
                if (i == cut && (rc = between(s))) break;
                if (rec) mmi_record_site(sites[i].c_str());
                if (tap) tap(i, true, s);
                rc = ops[i](s);
                if (tap) tap(i, false, s);
            }
            if (rec) { mmi_record_end(); logged = true; }
            return rc;
        }
        if (!exec_a || cut_ != cut) {
            if (exec_a) { hipGraphExecDestroy(exec_a); hipGraphDestroy(graph_a); exec_a = nullptr; }
            if (exec_b) { hipGraphExecDestroy(exec_b); hipGraphDestroy(graph_b); exec_b = nullptr; }
            if ((rc = capture_range(capture_stream, 0, cut, &graph_a, &exec_a))) return rc;
            if ((rc = capture_range(capture_stream, cut, ops.size(), &graph_b, &exec_b))) return rc;
            cut_ = cut;
        }
        MMI_HIP_CHECK(hipGraphLaunch(exec_a, s));
        if ((rc = between(s))) return rc;
        MMI_HIP_CHECK(hipGraphLaunch(exec_b, s));
        return MMI_OK;
    }

    void clear() {
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        if (exec_a) { hipGraphExecDestroy(exec_a); hipGraphDestroy(graph_a); }
        if (exec_b) { hipGraphExecDestroy(exec_b); hipGraphDestroy(graph_b); }
        exec_a = exec_b = nullptr;
        graph_a = graph_b = nullptr;
        exec = nullptr;
        graph = nullptr;
        ops.clear();
        sites.clear();
        op_bytes.clear();
        tap = nullptr;
        launch_log.clear();
        logged = false;
        site_ = "-";
    }
};

static inline bool mmi_graphs_enabled() {
    const char* e = getenv("MMI_NO_GRAPH");
    return !(e && e[0] && e[0] != '0');
}
