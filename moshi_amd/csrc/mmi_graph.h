// A frame step is a fixed list of kernel launches whose arguments never change between frames (every
// varying quantity - offsets, masks, tokens - lives in device memory).  The list is captured once
// into a hipGraph and replayed, which is what replaces the reference's CUDAGraphed wrappers
// (utils/compile.py:190-280; compression.py:224-229; lm.py:630-632).  MMI_NO_GRAPH=1 runs the same
// list eagerly, mirroring the reference's NO_CUDA_GRAPH switch (utils/compile.py:169-175).
#pragma once
#include "mmi_common.h"
#include <stdlib.h>

struct MmiProgram {
    std::vector<std::function<int(hipStream_t)>> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;

    void add(std::function<int(hipStream_t)> f) { ops.push_back(std::move(f)); }

    int run_eager(hipStream_t s) {
        for (auto& op : ops) {
            int rc = op(s);
            if (rc) return rc;
        }
        return MMI_OK;
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

    void clear() {
        if (exec) hipGraphExecDestroy(exec);
        if (graph) hipGraphDestroy(graph);
        exec = nullptr;
        graph = nullptr;
        ops.clear();
    }
};

static inline bool mmi_graphs_enabled() {
    const char* e = getenv("MMI_NO_GRAPH");
    return !(e && e[0] && e[0] != '0');
}
