// Session batcher behind the C ABI (include/moshi_mi.h, "Session batcher"): the model loop that packs many live
// full-duplex dialogue sessions into the batched frame step of one GPU.
//
// The reference's Python server holds ONE session under an asyncio lock (moshi/moshi/server.py:45,57,154-169); the
// reference's Rust server is the model for batching: a fixed number of slots, a channel per slot with a PCM FIFO, and a
// loop that every iteration (rust/moshi-server/src/batched_asr.rs:188-276)
//   pre_process  (:279-374)  pulls one 1920-sample frame per channel that has one -> stream mask; resets newly opened rows
//   step                     runs the batched model step with that mask
//   post_process (:376-437)  routes the per-row results back to the channel that owned the row when the step started.
// Here the step is the duplex path  Mimi encode -> LMGen.step -> Mimi decode  through the public entry points only
// (this file uses nothing of the engines but include/moshi_mi.h), so it is also the worked example of driving the ABI.
#include "mmi_common.h"

#include <deque>
#include <mutex>

namespace {

struct OutFrame {
    std::vector<float> pcm;
    std::vector<int64_t> tokens;
};

struct Channel {                 // batched_asr.rs:61-69
    int64_t id = 0;
    bool live = false;
    bool pending_reset = false;  // opened since the last step: the row's streaming state is reset before it runs
    long frames = 0;             // input frames consumed
    std::deque<float> in;        // PCM FIFO
    std::deque<OutFrame> out;
};

// One step's host<->device staging, laid out so that each direction is ONE copy.
struct Staging {
    // host -> device
    float* pcm = nullptr;        // [B][F]
    uint8_t* exec = nullptr;     // [B] rows that have a frame this step
    uint8_t* first = nullptr;    // [B] rows on their first frame (codec state dropped after the encode)
    uint8_t* reset = nullptr;    // [B] rows opened since the last step
    size_t h2d_bytes = 0;
    // device -> host
    int64_t* tokens = nullptr;   // [B][1 + dep_q]
    float* pcm_out = nullptr;    // [B][F]
    uint8_t* played = nullptr;   // [B] rows whose tokens were valid (decoder executed)
    size_t d2h_bytes = 0;
    unsigned char *up = nullptr, *down = nullptr;   // block bases
};

size_t align_up(size_t n) { return (n + 255) & ~(size_t)255; }

// up / down null: only the block sizes are wanted (no pointer is formed from a null base)
void carve(Staging* st, unsigned char* up, unsigned char* down, int B, int F, int NTOK) {
    auto at = [](unsigned char* base, size_t o) -> unsigned char* { return base ? base + o : nullptr; };
    size_t o = 0;
    st->up = up;
    st->pcm = reinterpret_cast<float*>(at(up, o)); o += align_up((size_t)B * F * sizeof(float));
    st->exec = at(up, o); o += align_up(B);
    st->first = at(up, o); o += align_up(B);
    st->reset = at(up, o); o += align_up(B);
    st->h2d_bytes = o;
    o = 0;
    st->down = down;
    st->tokens = reinterpret_cast<int64_t*>(at(down, o)); o += align_up((size_t)B * NTOK * sizeof(int64_t));
    st->pcm_out = reinterpret_cast<float*>(at(down, o)); o += align_up((size_t)B * F * sizeof(float));
    st->played = at(down, o); o += align_up(B);
    st->d2h_bytes = o;
}

// out tokens [B][1 + dep_q] (-2 = not generated yet, lm.py:781-782) -> decoder exec mask and codes:
// a row's audio is decoded only when it ran this step AND its tokens are valid; the codes are clamped into the codebook
// exactly where the reference indexes them unchecked (vq.py:144-146).
__global__ void k_batcher_route(const long* __restrict__ tokens, const uint8_t* __restrict__ exec, uint8_t* __restrict__ played,
                                long* __restrict__ codes, int B, int dep_q, int card) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= B) return;
    const long* row = tokens + (long)b * (1 + dep_q);
    bool ok = exec[b] != 0;
    for (int j = 0; j <= dep_q; ++j) ok = ok && row[j] >= 0;
    for (int k = 0; k < dep_q; ++k) {
        long c = row[1 + k];
        codes[(long)b * dep_q + k] = c < 0 ? 0 : (c >= card ? card - 1 : c);
    }
    played[b] = ok ? 1 : 0;
}

}  // namespace

struct mmi_batcher {
    mmi_mimi* mimi = nullptr;
    mmi_lm* lm = nullptr;
    mmi_batcher_cfg cfg;
    int B = 0, F = 0, K = 0, dep_q = 0, card = 0, NTOK = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool models_streaming = false;
    Staging host, dev;            // pinned host blocks and their device mirrors (same carving)
    int64_t* d_codes = nullptr;   // [B][K][1] user codes from the encoder
    int64_t* d_dec_codes = nullptr;
    std::mutex mu;                // guards channels / stats (batched_asr.rs:438 Channels = Arc<Mutex<..>>)
    std::vector<Channel> channels;
    std::vector<int64_t> row_owner;   // channel id that owned each row when the current step started
    int64_t next_id = 1;
    mmi_batcher_stats stats;
};

namespace {

Channel* find_channel(mmi_batcher* b, int64_t id) {
    for (auto& c : b->channels)
        if (c.live && c.id == id) return &c;
    return nullptr;
}

void release(mmi_batcher* b) {
    if (b->models_streaming) {
        hipStreamSynchronize(b->stream);
        mmi_lm_streaming_stop(b->lm);
        mmi_mimi_streaming_stop(b->mimi);
    }
    if (b->host.up) hipHostFree(b->host.up);
    if (b->host.down) hipHostFree(b->host.down);
    if (b->dev.up) hipFree(b->dev.up);
    if (b->dev.down) hipFree(b->dev.down);
    if (b->d_codes) hipFree(b->d_codes);
    if (b->d_dec_codes) hipFree(b->d_dec_codes);
    if (b->ev_begin) hipEventDestroy(b->ev_begin);
    if (b->ev_end) hipEventDestroy(b->ev_end);
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

int create_impl(mmi_batcher* b) {
    mmi_mimi_cfg mc;
    mmi_lm_cfg lc;
    int rc;
    if ((rc = mmi_mimi_get_cfg(b->mimi, &mc)) || (rc = mmi_lm_get_cfg(b->lm, &lc))) return rc;
    b->F = mc.frame_size * mc.channels;
    b->K = mmi_mimi_num_codebooks(b->mimi);
    b->dep_q = lc.dep_q;
    b->card = lc.card;
    b->NTOK = 1 + lc.dep_q;
    if (b->K < lc.n_q - lc.dep_q)
        return mmi_fail(MMI_ERR_SHAPE, "the codec produces fewer codebooks than the LM expects from the user stream");   // lm.py:683-686
    // the encoder produces K = max(dep_q, n_q - dep_q) codebooks (loaders.py:284-291); the decoder takes the dep_q the LM generates
    if (lc.dep_q > mc.q_n_q) return mmi_fail(MMI_ERR_SHAPE, "the LM generates more codebooks than the codec has");
    if (mc.q_bins != lc.card) return mmi_fail(MMI_ERR_SHAPE, "codec cardinality != LM card");
    const int B = b->B;
    MMI_HIP_CHECK(hipStreamCreate(&b->stream));
    MMI_HIP_CHECK(hipEventCreate(&b->ev_begin));
    MMI_HIP_CHECK(hipEventCreate(&b->ev_end));
    Staging probe;
    carve(&probe, nullptr, nullptr, B, b->F, b->NTOK);
    unsigned char *hu = nullptr, *hd = nullptr, *du = nullptr, *dd = nullptr;
    MMI_HIP_CHECK(hipHostMalloc((void**)&hu, probe.h2d_bytes, 0));
    b->host.up = hu;
    MMI_HIP_CHECK(hipHostMalloc((void**)&hd, probe.d2h_bytes, 0));
    b->host.down = hd;
    MMI_HIP_CHECK(hipMalloc((void**)&du, probe.h2d_bytes));
    b->dev.up = du;
    MMI_HIP_CHECK(hipMalloc((void**)&dd, probe.d2h_bytes));
    b->dev.down = dd;
    carve(&b->host, hu, hd, B, b->F, b->NTOK);
    carve(&b->dev, du, dd, B, b->F, b->NTOK);
    MMI_HIP_CHECK(hipMemset(dd, 0, probe.d2h_bytes));   // rows (or, for an ASR model, the whole PCM block) nobody writes read as zero
    memset(hu, 0, probe.h2d_bytes);
    memset(hd, 0, probe.d2h_bytes);
    MMI_HIP_CHECK(hipMalloc((void**)&b->d_codes, (size_t)B * b->K * sizeof(int64_t)));
    MMI_HIP_CHECK(hipMalloc((void**)&b->d_dec_codes, (size_t)B * (b->dep_q > 0 ? b->dep_q : 1) * sizeof(int64_t)));
    // streaming_forever(batch) on both models (server.py:59-60)
    if ((rc = mmi_mimi_streaming_start(b->mimi, B, b->stream))) return rc;
    b->models_streaming = true;
    const mmi_guidance* guide = (b->cfg.guidance.cfg_coef != 0.f && b->cfg.guidance.cfg_coef != 1.f) || b->cfg.guidance.condition_sum || b->cfg.guidance.condition_cross
                                    ? &b->cfg.guidance : nullptr;
    if (guide && guide->cfg_coef == 0.f) b->cfg.guidance.cfg_coef = 1.f;    // condition only
    if ((rc = mmi_lm_streaming_start_guided(b->lm, B, &b->cfg.sampling, guide, b->stream))) return rc;
    b->channels.resize(B);
    b->row_owner.assign(B, 0);
    memset(&b->stats, 0, sizeof(b->stats));
    b->stats.total_slots = B;
    return MMI_OK;
}

}  // namespace

extern "C" int mmi_batcher_create(mmi_mimi* mimi, mmi_lm* lm, const mmi_batcher_cfg* cfg, mmi_batcher** out) {
    MmiDeviceGuard dev_guard_(lm ? mmi_lm_device(lm) : -1);
    if (!mimi || !lm || !cfg || !out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (cfg->slots <= 0) return mmi_fail(MMI_ERR_INVALID, "slots must be positive");
    mmi_batcher* b = new mmi_batcher();
    b->mimi = mimi;
    b->lm = lm;
    b->cfg = *cfg;
    if (b->cfg.max_buffered_frames <= 0) b->cfg.max_buffered_frames = 250;   // 20 s of audio per channel
    b->B = cfg->slots;
    int rc = create_impl(b);
    if (rc) {
        release(b);
        return rc;
    }
    *out = b;
    return MMI_OK;
}

extern "C" void mmi_batcher_destroy(mmi_batcher* b) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (b) release(b);
}

extern "C" int mmi_batcher_open(mmi_batcher* b, int64_t* channel_id) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b || !channel_id) return mmi_fail(MMI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(b->mu);
    for (auto& c : b->channels) {
        if (c.live) continue;
        c = Channel();
        c.live = true;
        c.pending_reset = true;
        c.id = b->next_id++;
        *channel_id = c.id;
        b->stats.used_slots += 1;
        return MMI_OK;
    }
    return mmi_fail(MMI_ERR_BUSY, "no free slot");   // py_module.rs:443-470: `channels()` yields None
}

extern "C" int mmi_batcher_close(mmi_batcher* b, int64_t channel_id) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b) return mmi_fail(MMI_ERR_INVALID, "null handle");
    std::lock_guard<std::mutex> g(b->mu);
    Channel* c = find_channel(b, channel_id);
    if (!c) return mmi_fail(MMI_ERR_NO_CHANNEL, "unknown channel");
    *c = Channel();
    b->stats.used_slots -= 1;
    return MMI_OK;
}

extern "C" int mmi_batcher_push_pcm(mmi_batcher* b, int64_t channel_id, const float* pcm, int32_t n_samples) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b || (!pcm && n_samples > 0) || n_samples < 0) return mmi_fail(MMI_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> g(b->mu);
    Channel* c = find_channel(b, channel_id);
    if (!c) return mmi_fail(MMI_ERR_NO_CHANNEL, "unknown channel");
    if (c->in.size() + (size_t)n_samples > (size_t)b->cfg.max_buffered_frames * b->F)
        return mmi_fail(MMI_ERR_BUSY, "channel input buffer full");
    c->in.insert(c->in.end(), pcm, pcm + n_samples);
    return MMI_OK;
}

extern "C" int mmi_batcher_step(mmi_batcher* b, int32_t* n_active) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b) return mmi_fail(MMI_ERR_INVALID, "null handle");
    const int B = b->B, F = b->F;
    Staging& h = b->host;
    Staging& d = b->dev;
    int active = 0, resets = 0, firsts = 0;
    {   // ---- pre_process (batched_asr.rs:279-374)
        std::lock_guard<std::mutex> g(b->mu);
        for (int r = 0; r < B; ++r) {
            Channel& c = b->channels[r];
            h.exec[r] = h.first[r] = h.reset[r] = 0;
            b->row_owner[r] = c.live ? c.id : 0;
            float* dst = h.pcm + (size_t)r * F;
            if (c.live && c.pending_reset) {
                h.reset[r] = 1;
                c.pending_reset = false;
                ++resets;
            }
            if (c.live && c.in.size() >= (size_t)F) {
                std::copy(c.in.begin(), c.in.begin() + F, dst);
                c.in.erase(c.in.begin(), c.in.begin() + F);
                h.exec[r] = 1;
                if (c.frames == 0 && b->cfg.reset_codec_after_first_frame) {
                    h.first[r] = 1;
                    ++firsts;
                }
                c.frames += 1;
                ++active;
            } else {
                memset(dst, 0, (size_t)F * sizeof(float));
            }
        }
    }
    if (n_active) *n_active = active;
    if (active == 0 && resets == 0) return MMI_OK;
    hipStream_t s = b->stream;
    int rc;
    MMI_HIP_CHECK(hipEventRecord(b->ev_begin, s));
    MMI_HIP_CHECK(hipMemcpyAsync(d.up, h.up, h.h2d_bytes, hipMemcpyHostToDevice, s));
    if (resets) {   // handle_chat: mimi.reset_streaming(); lm_gen.reset_streaming() (server.py:163-164), for the new rows only
        if ((rc = mmi_mimi_reset(b->mimi, d.reset, s))) return rc;
        if ((rc = mmi_lm_reset(b->lm, d.reset, s))) return rc;
    }
    if (active == 0) {
        MMI_HIP_CHECK(hipStreamSynchronize(s));
        return MMI_OK;
    }
    // ---- the frame step with the stream mask (batched_asr.rs:233-262; server.py:132-146)
    if ((rc = mmi_mimi_set_exec_mask(b->mimi, d.exec, s))) return rc;
    if ((rc = mmi_lm_set_exec_mask(b->lm, d.exec, s))) return rc;
    if ((rc = mmi_mimi_encode_step(b->mimi, d.pcm, b->d_codes, B, 1, s))) return rc;
    if (firsts && (rc = mmi_mimi_reset(b->mimi, d.first, s))) return rc;   // server.py:135-141
    int valid = 0;
    if ((rc = mmi_lm_step(b->lm, b->d_codes, b->K, d.tokens, nullptr, nullptr, nullptr, B, &valid, s))) return rc;
    MMI_LAUNCH(k_batcher_route, mmi_cdiv(B, 64), 64, 0, s, (const long*)d.tokens, (const uint8_t*)d.exec, d.played,
               (long*)b->d_dec_codes, B, b->dep_q, b->card);
    MMI_CHECK_LAUNCH();
    if (b->dep_q > 0) {   // an ASR-style model (dep_q = 0) generates no audio: the step ends at the text token
        if ((rc = mmi_mimi_set_exec_mask(b->mimi, d.played, s))) return rc;   // rows still inside the LM delay do not touch the decoder
        if ((rc = mmi_mimi_decode_step(b->mimi, b->d_dec_codes, d.pcm_out, B, b->dep_q, 1, s))) return rc;
    }
    MMI_HIP_CHECK(hipMemcpyAsync(h.down, d.down, h.d2h_bytes, hipMemcpyDeviceToHost, s));
    MMI_HIP_CHECK(hipEventRecord(b->ev_end, s));
    MMI_HIP_CHECK(hipEventSynchronize(b->ev_end));
    float ms = 0.f;
    hipEventElapsedTime(&ms, b->ev_begin, b->ev_end);
    {   // ---- post_process (batched_asr.rs:376-437): a row's result goes to the channel that owned the row at pre_process
        std::lock_guard<std::mutex> g(b->mu);
        b->stats.steps += 1;
        b->stats.frames += active;
        b->stats.last_step_ms = ms;
        for (int r = 0; r < B; ++r) {
            if (!h.exec[r] || !h.played[r]) continue;
            Channel& c = b->channels[r];
            if (!c.live || c.id != b->row_owner[r]) continue;   // closed (or re-opened by someone else) meanwhile
            if ((int)c.out.size() >= b->cfg.max_buffered_frames) {
                c.out.pop_front();
                b->stats.dropped_frames += 1;
            }
            OutFrame f;
            f.pcm.assign(h.pcm_out + (size_t)r * F, h.pcm_out + (size_t)(r + 1) * F);
            f.tokens.assign(h.tokens + (size_t)r * b->NTOK, h.tokens + (size_t)(r + 1) * b->NTOK);
            c.out.push_back(std::move(f));
        }
    }
    return MMI_OK;
}

extern "C" int mmi_batcher_pop(mmi_batcher* b, int64_t channel_id, float* pcm, int64_t* tokens, int32_t* got) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b || !got) return mmi_fail(MMI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(b->mu);
    Channel* c = find_channel(b, channel_id);
    if (!c) return mmi_fail(MMI_ERR_NO_CHANNEL, "unknown channel");
    *got = 0;
    if (c->out.empty()) return MMI_OK;
    OutFrame& f = c->out.front();
    if (pcm) std::copy(f.pcm.begin(), f.pcm.end(), pcm);
    if (tokens) std::copy(f.tokens.begin(), f.tokens.end(), tokens);
    c->out.pop_front();
    *got = 1;
    return MMI_OK;
}

extern "C" int mmi_batcher_get_stats(mmi_batcher* b, mmi_batcher_stats* out) {
    MmiDeviceGuard dev_guard_(b ? mmi_lm_device(b->lm) : -1);
    if (!b || !out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(b->mu);
    *out = b->stats;
    return MMI_OK;
}
