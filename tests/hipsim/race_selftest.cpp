// Negative control of the schedule test (tests/test_schedule_sim.py): the same tiny kernel with and without the barrier between an
// LDS write and a read of ANOTHER wave's slot.  Without it the output depends on which wave runs first - the simulator's perturbed
// schedules must show that, or the schedule test proves nothing.  TEST INFRASTRUCTURE ONLY.
#include "mmi_device.h"

static void k_exchange(float* out, int racy) {
    MMI_SHARED float slot[128];
    const int t = threadIdx.x;
    slot[t] = 0.0f;
    __syncthreads();
    slot[t] = (float)(t + 1);
    if (!racy) __syncthreads();                  // the barrier a correct kernel has
    out[t] = slot[(t + 64) & 127];               // the other wave's slot
}

extern "C" void race_selftest_run(float* out, int racy) {
    hipLaunchKernelGGL(k_exchange, dim3(1), dim3(128), 0, nullptr, out, racy);
}
