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
    // A list may exist in up to NV variants: the op lambdas may read `variant` when they are launched / captured and issue
    // different launches for it.  Each variant is captured into its own graph(s) the first time it runs; the caller picks one per
    // step from what it knows on the HOST (lm_engine.hip: how deep the KV rings can be decides whether the decode attention
    // needs its merge launch).  Programs that never touch `variant` have one.
    static constexpr int NV = 2;
    std::vector<std::function<int(hipStream_t)>> ops;
    std::vector<std::string> sites;             // one label per op
    std::vector<long> op_bytes;                 // weight bytes the op streams (GEMMs), 0 otherwise
    std::function<void(size_t, bool, hipStream_t)> tap;   // profiling: called before (true) and after (false) every eagerly run op
    std::vector<std::string> launch_logs[NV];   // "site\tkernel" per launch, in launch order (filled by the variant's first run)
    std::string site_ = "-";                    // label given to the ops added from now on
    bool logged_[NV] = {false, false};
    int variant = 0;
    hipGraph_t graph[NV] = {nullptr, nullptr};
    hipGraphExec_t exec[NV] = {nullptr, nullptr};
    // the same list captured as two graphs, ops [0, cut) and [cut, end), for callers that put something between the halves
    // (run_split: an event record that another stream waits on)
    hipGraph_t graph_a[NV] = {nullptr, nullptr}, graph_b[NV] = {nullptr, nullptr};
    hipGraphExec_t exec_a[NV] = {nullptr, nullptr}, exec_b[NV] = {nullptr, nullptr};
    size_t cut_[NV] = {0, 0};

    void site(const std::string& label) { site_ = label; }
    void add(std::function<int(hipStream_t)> f, long bytes = 0) {
        // MMI_SKIP_SITES="enc.tr,dec.res" (timing experiments ONLY - the outputs are then garbage): ops whose site label starts with
        // one of the prefixes are left out of the list.  How round 6 found which part of the codec stretches the depth-transformer
        // phase under the duplex pipeline (profiles/r06_logs/pipeline_skip_sites.txt)
        if (const char* e = getenv("MMI_SKIP_SITES")) {
            std::string all(e);
            size_t i = 0;
            while (i <= all.size()) {
                const size_t j = all.find(',', i);
                const std::string pre = all.substr(i, j == std::string::npos ? std::string::npos : j - i);
                if (!pre.empty() && site_.compare(0, pre.size(), pre) == 0) return;
                if (j == std::string::npos) break;
                i = j + 1;
            }
        }
        ops.push_back(std::move(f)); sites.push_back(site_); op_bytes.push_back(bytes);
    }
    bool logged() const { return logged_[variant]; }
    const std::vector<std::string>& launch_log() const { return launch_logs[variant]; }

    int run_eager(hipStream_t s) {
        const bool rec = !logged_[variant];
        if (rec) mmi_record_begin(&launch_logs[variant]);
        int rc = MMI_OK;
        for (size_t i = 0; i < ops.size() && !rc; ++i) {
            if (rec) mmi_record_site(sites[i].c_str());
            if (tap) tap(i, true, s);
            rc = ops[i](s);
            if (tap) tap(i, false, s);
        }
        if (rec) { mmi_record_end(); logged_[variant] = true; }
        return rc;
    }

    // ops [i0, i1) launched eagerly (the step cut into segments around per-step host callbacks)
    int run_range(hipStream_t s, size_t i0, size_t i1) {
        int rc = MMI_OK;
        for (size_t i = i0; i < i1 && i < ops.size() && !rc; ++i) rc = ops[i](s);
        return rc;
    }

    // the CURRENT variant's whole list captured into graph[v] / exec[v] (nothing executes; the first capture also records the launch list)
    int capture_full(hipStream_t capture_stream) {
        const int v = variant;
        if (exec[v]) return MMI_OK;
        MMI_HIP_CHECK(hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal));
        int rc = run_eager(capture_stream);
        hipError_t e = hipStreamEndCapture(capture_stream, &graph[v]);
        if (rc || e != hipSuccess) {
            if (e == hipSuccess && graph[v]) hipGraphDestroy(graph[v]);
            graph[v] = nullptr;
            if (rc) return rc;
            return mmi_fail(MMI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        }
        e = hipGraphInstantiate(&exec[v], graph[v], nullptr, nullptr, 0);
        if (e != hipSuccess) {
            hipGraphDestroy(graph[v]);
            graph[v] = nullptr;
            exec[v] = nullptr;
            return mmi_fail(MMI_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        }
        return MMI_OK;
    }

    int run(hipStream_t s, bool use_graph, hipStream_t capture_stream) {
        if (!use_graph) return run_eager(s);
        const int v = variant;
        if (!exec[v]) {
            int rc = capture_full(capture_stream);
            if (rc) return rc;
        }
        MMI_HIP_CHECK(hipGraphLaunch(exec[v], s));
        return MMI_OK;
    }

    // Variant v made ready AHEAD of the step that first needs it (a stream whose rings grow past the short-ring program
    // would otherwise capture and instantiate the other program in the middle of a live session - a one-off latency spike in a
    // real-time stream): its graph(s) captured and instantiated now, nothing launched.  split: the two-graph form run_split uses.
    int precapture(int v, hipStream_t capture_stream, bool split, size_t cut) {
        const int keep = variant;
        variant = v;
        int rc = MMI_OK;
        if (!split || !logged_[v]) rc = capture_full(capture_stream);       // (also records the variant's launch list)
        if (!rc && split && (!exec_a[v] || cut_[v] != cut)) {
            drop_split(v);
            if (!(rc = capture_range(capture_stream, 0, cut, &graph_a[v], &exec_a[v])) &&
                !(rc = capture_range(capture_stream, cut, ops.size(), &graph_b[v], &exec_b[v])))
                cut_[v] = cut;
        }
        variant = keep;
        return rc;
    }
    bool ready(int v, bool split) const { return split ? (exec_a[v] && exec_b[v]) : exec[v] != nullptr; }

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
        const int v = variant;
        if (!use_graph || !logged_[v]) {        // the first run of a variant is the eager one that records the launch list
            const bool rec = !logged_[v];
            if (rec) mmi_record_begin(&launch_logs[v]);
            rc = MMI_OK;
            for (size_t i = 0; i < ops.size() && !rc; ++i) {
                if (i == cut && (rc = between(s))) break;
                    if (rec) mmi_record_site(sites[i].c_str());
                if (tap) tap(i, true, s);
                rc = ops[i](s);
                if (tap) tap(i, false, s);
            }
            if (rec) { mmi_record_end(); logged_[v] = true; }
            return rc;
        }
        if (!exec_a[v] || cut_[v] != cut) {
            drop_split(v);
            if ((rc = capture_range(capture_stream, 0, cut, &graph_a[v], &exec_a[v]))) return rc;
            if ((rc = capture_range(capture_stream, cut, ops.size(), &graph_b[v], &exec_b[v]))) return rc;
            cut_[v] = cut;
        }
        MMI_HIP_CHECK(hipGraphLaunch(exec_a[v], s));
        if ((rc = between(s))) return rc;
        MMI_HIP_CHECK(hipGraphLaunch(exec_b[v], s));
        return MMI_OK;
    }

    void drop_split(int v) {
        if (exec_a[v]) { hipGraphExecDestroy(exec_a[v]); hipGraphDestroy(graph_a[v]); }
        if (exec_b[v]) { hipGraphExecDestroy(exec_b[v]); hipGraphDestroy(graph_b[v]); }
        exec_a[v] = exec_b[v] = nullptr;
        graph_a[v] = graph_b[v] = nullptr;
    }

    void clear() {
        for (int v = 0; v < NV; ++v) {
            if (exec[v]) hipGraphExecDestroy(exec[v]);
            if (graph[v]) hipGraphDestroy(graph[v]);
            exec[v] = nullptr;
            graph[v] = nullptr;
            drop_split(v);
            launch_logs[v].clear();
            logged_[v] = false;
        }
        variant = 0;
        ops.clear();
        sites.clear();
        op_bytes.clear();
        tap = nullptr;
        site_ = "-";
    }
};

static inline bool mmi_graphs_enabled() {
    const char* e = getenv("MMI_NO_GRAPH");
    return !(e && e[0] && e[0] != '0');
}
