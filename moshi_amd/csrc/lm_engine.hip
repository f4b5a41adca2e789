// placeholder until the LM engine lands (next milestone): the ABI symbols exist and fail loudly.
#include "mmi_common.h"
extern "C" int mmi_lm_create(const mmi_lm_cfg*, const mmi_tensor_desc*, int32_t, int32_t, mmi_lm**) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" void mmi_lm_destroy(mmi_lm*) {}
extern "C" int mmi_lm_streaming_start(mmi_lm*, int32_t, const mmi_sampling*, mmi_stream) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_streaming_stop(mmi_lm*) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_set_exec_mask(mmi_lm*, const uint8_t*, mmi_stream) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_reset(mmi_lm*, const uint8_t*, mmi_stream) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_step(mmi_lm*, const int64_t*, int32_t, int64_t*, float*, float*, const float*, int32_t, int32_t*, mmi_stream) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_profile_begin(mmi_lm*) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
extern "C" int mmi_lm_profile_end(mmi_lm*, double*, int64_t*, int64_t*, const char**) { return mmi_fail(MMI_ERR_UNSUPPORTED, "LM engine not built yet"); }
