// Duplex pipeline behind the C ABI (include/moshi_mi.h, "Duplex pipeline"): the frame step
//     MimiModel.encode -> LMGen.step -> MimiModel.decode            (moshi/moshi/server.py:132-146)
// with the three calls of consecutive frames overlapped on three HIP streams.  Like batcher.hip this file uses nothing of the
// engines but the public entry points: the engines take a stream per call, so the pipeline is streams + hand-offs around them.
//
// Dependencies of frame t (E = encoder stream, L = LM stream, D = decoder stream):
//     E: encode(t)      after encode(t-1) [stream order]
//     L: LMGen.step(t)  after step(t-1)   [stream order]      and after encode(t)
//     D: decode(t)      after decode(t-1) [stream order]      and after LMGen.step(t)
// The user codes are handed over through a two-slot ring indexed by frame parity: a slot is reused by frame t+2, which the host
// only enqueues once LMGen.step(t) has completed (flow control in mmi_duplex_submit).  The step's tokens go through a three-slot
// ring: decode(t-2) runs beside the depth-transformer phase of step t-1 and ends about when that step does, so step t writes the
// slot of decode(t-3) instead of waiting for it (no measurable difference at 32 sessions - the device-clock stamps,
// mmi_duplex_get_stamps, show the LM's queue idle for 15 us between two steps either way - but no dependency either).
//
// Gate: the temporal transformer's GEMMs are chip-filling, HBM-bound launches with one workgroup per CU and a
// static tile partition - a codec workgroup that takes a CU for 10 us makes such a launch 10 us late.  The depth-transformer
// phase (1.6 of the LM's 5.7 ms at 32 sessions) is dep_q x 33 small dependent launches on 32-192 of the 256 CUs.  So the codec
// work of the neighbouring frames - decode(t-1), encode(t+1): 1.5 ms of latency-bound launches - is held until step t reaches
// that phase (mmi_lm_set_phase_callback) and runs in its shadow.
//
// Hand-offs between streams are counters in device memory, published by a one-thread kernel and polled by a one-wave kernel
// (mmi_device.h mmi_flag_*), NOT hipStreamWaitEvent: on this stack a wait that stays pending makes the command processor poll
// the producer queue's signal and slows every dependent launch of the producer by ~1.2 us (scripts/stream_probe.hip test 1e:
// 1.60 -> 2.77 ms per 1000 launches; a resident polling wave: 1.71 ms) - the decoder stream waiting for a 487-launch LM step
// cost that step 0.6 ms.  Events remain where the wait is already satisfied when the queue reaches it, and for the host.
#include "mmi_common.h"

namespace {

__global__ void k_flag_publish(long* flag, long v) { mmi_flag_publish(flag, v); }
__global__ void k_flag_wait(const long* flag, long v) {
    if (threadIdx.x == 0) mmi_flag_wait(flag, v);
}

// diagnostic stamps (mmi_duplex_get_stamps): device clock at the pipeline's hand-off points, four frames deep
enum { S_IN = 0, S_ENC0, S_ENC1, S_WAIT0, S_WAIT1, S_LM0, S_PHASE, S_LM1, S_DEC0, S_DEC1, S_COUNT };
__global__ void k_stamp(long* p) { *p = mmi_wall_clock(); }
__global__ void k_flag_wait_stamped(const long* flag, long v, long* p) {
    if (threadIdx.x == 0) {
        p[0] = mmi_wall_clock();
        mmi_flag_wait(flag, v);
        p[1] = mmi_wall_clock();
    }
}

struct Pending { bool live = false, valid = false; float* pcm_out = nullptr; int64_t* tokens_out = nullptr; long frame = 0; };

enum { F_ENC = 0, F_LM = 1, F_DEC = 2, F_IN = 3, F_COUNT = 4 };

}  // namespace

struct mmi_duplex {
    mmi_mimi* mimi = nullptr;
    mmi_lm* lm = nullptr;
    int device = -1;
    int B = 0, F = 0, K = 0, dep_q = 0, NTOK = 0;
    hipStream_t sE = nullptr, sL = nullptr, sD = nullptr;
    hipEvent_t ev_lm[2] = {nullptr, nullptr};      // host flow control: LMGen.step(t) done (never waited on by a stream)
    hipEvent_t ev_dec[3] = {nullptr, nullptr, nullptr};   // decode(t) done: waited on by L `slots` frames later, when it has long completed
    static constexpr int slots = 3;                // token ring
    long* flags = nullptr;                         // [F_COUNT][16] device counters, one cache line each: frames completed per phase
    long phase_frame = 0;                          // frame whose phase the callback records
    // The gate is kept by the HOST: submit(t) returns to enqueueing only once step t-1 has reached its depth-transformer phase
    // (ev_phase), so the gated work - encode(t), decode(t-2) - needs no device-side wait at all and no polling wave sits on the
    // codec queues through the LM's temporal phase.  (Measured and dropped, profiles/r03_logs/duplex_ab_a_to_l.txt: a device-side
    // gate through a flag, gating only one codec half, no gate, event hand-offs, other stream priorities, one codec stream.)
    hipEvent_t ev_phase[2] = {nullptr, nullptr};
    // diagnostic timeline (mmi_duplex_set_timeline): timestamps of the last frame's phases
    bool timeline = false;
    long* stamps = nullptr;                        // [4][S_COUNT] device clock stamps of frames t & 3 (timeline on)
    hipEvent_t tl[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // in, enc0, enc1, lm0, -, lm1, dec0, dec1
    Pending pend[3];                               // a frame's decode, enqueued two submits later (behind step t+1's phase) or by join / flush
    int64_t* codes[2] = {nullptr, nullptr};        // [B][K][1]         encoder -> LM
    int64_t* tokens[3] = {nullptr, nullptr, nullptr};   // [B][1 + dep_q][1] LM -> decoder
    // the frame's input is copied (on the caller's stream, ahead of the F_IN publish) into a slot of this ring, so the caller may
    // refill its own buffer as soon as submit returns - as it may after MimiModel.encode on one stream (compression.py:376-388);
    // 4 slots: encode(t) has completed before submit(t+2) returns (flow control on step t ... t-2), one spare
    static constexpr int in_slots = 4;
    float* pcm_in[4] = {nullptr, nullptr, nullptr, nullptr};
    long frame = 0;
    // A failure half-way through a submit leaves the three stream orders inconsistent (the encoder state has advanced, a waiter
    // may be resident, the frame counter has not moved): the handle is dead from then on - every later submit / join / flush
    // returns MMI_ERR_STATE until destroy - and the flags are released so that no polling wave is left without its producer.
    bool failed = false;
};

namespace {

long* flag_of(mmi_duplex* d, int which) { return d->flags + 16 * which; }

// producer side: frame `t` of phase `which` is complete once everything enqueued on s so far has run
int publish(mmi_duplex* d, int which, long t, hipStream_t s) {
    MMI_LAUNCH(k_flag_publish, 1, 1, 0, s, flag_of(d, which), t + 1);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}
// consumer side: s goes on once frame `t` of phase `which` is complete
int await(mmi_duplex* d, int which, long t, hipStream_t s) {
    MMI_LAUNCH(k_flag_wait, 1, 64, 0, s, (const long*)flag_of(d, which), t + 1);
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

long* stamp_of(mmi_duplex* d, long t, int which) { return d->stamps + (t & 3) * S_COUNT + which; }
int stamp(mmi_duplex* d, long t, int which, hipStream_t s) {
    if (!d->timeline || !d->stamps) return MMI_OK;
    MMI_LAUNCH(k_stamp, 1, 1, 0, s, stamp_of(d, t, which));
    MMI_CHECK_LAUNCH();
    return MMI_OK;
}

int phase_callback(void* user, mmi_stream stream) {
    mmi_duplex* d = (mmi_duplex*)user;
    if (int rc = stamp(d, d->phase_frame, S_PHASE, (hipStream_t)stream)) return rc;
    MMI_HIP_CHECK(hipEventRecord(d->ev_phase[d->phase_frame & 1], (hipStream_t)stream));
    return MMI_OK;
}

// a step that failed half-way may have left a polling wave without its producer: let every waiter through
void open_flags(mmi_duplex* d) {
    if (!d->flags) return;
    std::vector<long> big((size_t)F_COUNT * 16, (long)1 << 62);
    hipMemcpy(d->flags, big.data(), big.size() * sizeof(long), hipMemcpyHostToDevice);
}

int fail_sticky(mmi_duplex* d, int rc) {
    if (rc != MMI_OK && !d->failed) {
        d->failed = true;
        open_flags(d);
    }
    return rc;
}

void release(mmi_duplex* d) {
    open_flags(d);
    for (hipStream_t s : {d->sE, d->sL, d->sD})
        if (s) { hipStreamSynchronize(s); hipStreamDestroy(s); }
    for (int i = 0; i < 2; ++i) {
        if (d->ev_lm[i]) hipEventDestroy(d->ev_lm[i]);
        if (d->ev_phase[i]) hipEventDestroy(d->ev_phase[i]);
        if (d->codes[i]) hipFree(d->codes[i]);
    }
    for (int i = 0; i < 3; ++i) {
        if (d->ev_dec[i]) hipEventDestroy(d->ev_dec[i]);
        if (d->tokens[i]) hipFree(d->tokens[i]);
    }
    for (float* p : d->pcm_in) if (p) hipFree(p);
    if (d->flags) hipFree(d->flags);
    if (d->stamps) hipFree(d->stamps);
    for (hipEvent_t e : d->tl) if (e) hipEventDestroy(e);
    delete d;
}

int create_impl(mmi_duplex* d) {
    mmi_mimi_cfg mc;
    mmi_lm_cfg lc;
    int rc;
    if ((rc = mmi_mimi_get_cfg(d->mimi, &mc)) || (rc = mmi_lm_get_cfg(d->lm, &lc))) return rc;
    d->B = mmi_mimi_streaming_batch(d->mimi);
    if (d->B <= 0 || mmi_lm_streaming_batch(d->lm) <= 0)
        return mmi_fail(MMI_ERR_STATE, "mmi_duplex_create: both models must be streaming");                // lm.py:673-676
    if (d->B != mmi_lm_streaming_batch(d->lm)) return mmi_fail(MMI_ERR_SHAPE, "the codec and the LM stream different batch sizes");
    d->F = mc.frame_size * mc.channels;
    d->K = mmi_mimi_num_codebooks(d->mimi);
    d->dep_q = lc.dep_q;
    d->NTOK = 1 + lc.dep_q;
    if (d->K < lc.n_q - lc.dep_q)
        return mmi_fail(MMI_ERR_SHAPE, "the codec produces fewer codebooks than the LM expects from the user stream");   // lm.py:683-686
    if (lc.dep_q < 1 || lc.dep_q > mc.q_n_q) return mmi_fail(MMI_ERR_SHAPE, "the LM must generate between 1 and n_q codebooks for the codec");
    if (mc.q_bins != lc.card) return mmi_fail(MMI_ERR_SHAPE, "codec cardinality != LM card");
    // Stream priorities.  Streams of one priority share a pool of GPU_MAX_HW_QUEUES (4) hardware queues with everything else the
    // process created at that priority, and two streams that land on one queue do not overlap (profiles/r03_logs: the encoder
    // and the LM shared queue 4 without priorities).  The LM and the codec are therefore given DIFFERENT priorities (distinct queue
    // pools); which of the two is the high one matters little (below).
    MMI_HIP_CHECK(hipMalloc((void**)&d->flags, (size_t)F_COUNT * 16 * sizeof(long)));
    MMI_HIP_CHECK(hipMemset(d->flags, 0, (size_t)F_COUNT * 16 * sizeof(long)));
    int lo = 0, hi = 0;
    // MMI_DUPLEX_CODEC_CUS=n (experiment, round 6): the codec streams confined to n of the CUs (a CU mask of n bits, which the driver
    // deals out over the XCDs), so that the depth transformer's launches - 96..176 workgroups that need a whole CU's wave slots each -
    // find free CUs while the codec of the neighbouring frames runs beside them; such streams have the default priority
    const int codec_cus = getenv("MMI_DUPLEX_CODEC_CUS") ? atoi(getenv("MMI_DUPLEX_CODEC_CUS")) : 0;
    if (codec_cus > 0 && codec_cus < 256) {
        unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < codec_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
        int plo = 0, phi = 0;
        if (hipDeviceGetStreamPriorityRange(&plo, &phi) == hipSuccess && plo != phi) MMI_HIP_CHECK(hipStreamCreateWithPriority(&d->sL, hipStreamNonBlocking, plo));
        else MMI_HIP_CHECK(hipStreamCreateWithFlags(&d->sL, hipStreamNonBlocking));
        MMI_HIP_CHECK(hipExtStreamCreateWithCUMask(&d->sE, 8, mask));
        MMI_HIP_CHECK(hipExtStreamCreateWithCUMask(&d->sD, 8, mask));
    } else
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
        // The LM's stream in the high-priority pool, the codec's two in the low one: the step's critical path is the LM's launch
        // chain, the codec only has to be done by the time the next step wants its codes.  Same-box pairs (round 6,
        // profiles/r06_logs/ab_stream_priorities.txt): -0.02 ms per 32-session step on average against the reverse (rounds 3-5),
        // -0.06 ms at 64 sessions; equal priorities: +0.5 ms.  MMI_DUPLEX_PRIO = codec / none select those for an A/B.
        const char* pe = getenv("MMI_DUPLEX_PRIO");
        const int mid = (lo + hi) / 2;
        const int pl = pe && pe[0] == 'c' ? lo : (pe && pe[0] == 'n' ? mid : hi), pc = pe && pe[0] == 'c' ? hi : (pe && pe[0] == 'n' ? mid : lo);
        MMI_HIP_CHECK(hipStreamCreateWithPriority(&d->sL, hipStreamNonBlocking, pl));
        MMI_HIP_CHECK(hipStreamCreateWithPriority(&d->sE, hipStreamNonBlocking, pc));
        MMI_HIP_CHECK(hipStreamCreateWithPriority(&d->sD, hipStreamNonBlocking, pc));
    } else {
        MMI_HIP_CHECK(hipStreamCreateWithFlags(&d->sL, hipStreamNonBlocking));
        MMI_HIP_CHECK(hipStreamCreateWithFlags(&d->sE, hipStreamNonBlocking));
        MMI_HIP_CHECK(hipStreamCreateWithFlags(&d->sD, hipStreamNonBlocking));
    }
    for (int i = 0; i < 2; ++i) {
        MMI_HIP_CHECK(hipEventCreateWithFlags(&d->ev_lm[i], hipEventDisableTiming));
        MMI_HIP_CHECK(hipEventCreateWithFlags(&d->ev_phase[i], hipEventDisableTiming));
        MMI_HIP_CHECK(hipMalloc((void**)&d->codes[i], (size_t)d->B * d->K * sizeof(int64_t)));
    }
    for (int i = 0; i < 3; ++i) {
        MMI_HIP_CHECK(hipEventCreateWithFlags(&d->ev_dec[i], hipEventDisableTiming));
        MMI_HIP_CHECK(hipMalloc((void**)&d->tokens[i], (size_t)d->B * d->NTOK * sizeof(int64_t)));
    }
    for (int i = 0; i < d->in_slots; ++i) MMI_HIP_CHECK(hipMalloc((void**)&d->pcm_in[i], (size_t)d->B * d->F * sizeof(float)));
    return MMI_OK;
}

// decode(t) of the audio columns of step t's output, read in place; rows still inside the LM's delay hold -2, which the
// decoder's gather clamps into the codebook exactly where the reference would index unchecked (vq.py:144-146).
// host_saw_step: the host has seen the frame's LM step complete (no device-side wait); else the decoder's stream waits for it.
int enqueue_decode(mmi_duplex* d, int p, bool host_saw_step) {
    Pending& q = d->pend[p];
    if (!q.live) return MMI_OK;
    int rc;
    if (!host_saw_step && (rc = await(d, F_LM, q.frame, d->sD))) return rc;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[6], d->sD));
    stamp(d, q.frame, S_DEC0, d->sD);
    if (q.tokens_out)
        MMI_HIP_CHECK(hipMemcpyAsync(q.tokens_out, d->tokens[p], (size_t)d->B * d->NTOK * sizeof(int64_t), hipMemcpyDeviceToDevice, d->sD));
    if (q.valid && (rc = mmi_mimi_decode_step_strided(d->mimi, d->tokens[p] + 1, d->NTOK, q.pcm_out, d->B, d->dep_q, 1, d->sD))) return rc;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[7], d->sD));
    stamp(d, q.frame, S_DEC1, d->sD);
    if ((rc = publish(d, F_DEC, q.frame, d->sD))) return rc;
    MMI_HIP_CHECK(hipEventRecord(d->ev_dec[p], d->sD));
    q.live = false;
    return MMI_OK;
}

}  // namespace

extern "C" int mmi_duplex_create(mmi_mimi* mimi, mmi_lm* lm, mmi_duplex** out) {
    MmiDeviceGuard dev_guard_(lm ? mmi_lm_device(lm) : -1);
    if (!mimi || !lm || !out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (mmi_lm_device(lm) != mmi_mimi_device(mimi)) return mmi_fail(MMI_ERR_INVALID, "the codec and the LM live on different devices");
    mmi_duplex* d = new mmi_duplex();
    d->mimi = mimi;
    d->lm = lm;
    d->device = mmi_lm_device(lm);
    int rc = create_impl(d);
    if (rc) {
        release(d);
        return rc;
    }
    *out = d;
    return MMI_OK;
}

extern "C" void mmi_duplex_destroy(mmi_duplex* d) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (d) release(d);
}

namespace {
int submit_impl(mmi_duplex* d, const float* pcm_caller, float* pcm_out, int64_t* tokens_out, int32_t* valid, mmi_stream caller) {
    const long t = d->frame;
    const int p = (int)(t & 1);                            // codes slot, flow-control events
    const int q = (int)(t % d->slots);                     // tokens slot, pending decode, ev_dec
    const int q_m2 = (int)((t + d->slots - 2) % d->slots);     // slot of frame t-2
    int rc;
    // flow control: the host runs at most two LM steps ahead of the device.  It blocks here until LMGen.step(t-2) - the previous
    // writer of codes slot p - has completed, which also bounds the lifetime the caller owes its pcm_in buffers.
    if (t >= 2) MMI_HIP_CHECK(hipEventSynchronize(d->ev_lm[p]));
    // host-kept gate: the frame is enqueued once LMGen.step(t-1) has reached its depth-transformer phase
    if (t >= 1) MMI_HIP_CHECK(hipEventSynchronize(d->ev_phase[p ^ 1]));
    // the frame's input -> the pipeline's own ring, in the caller's stream order (the caller's buffer is free again
    // when this call returns, as after MimiModel.encode)
    float* pcm_in = d->pcm_in[t % d->in_slots];
    MMI_HIP_CHECK(hipMemcpyAsync(pcm_in, pcm_caller, (size_t)d->B * d->F * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)caller));
    // everything the caller enqueued so far (the frame's input; mask / reset calls made after a join) comes first: the encoder
    // waits for it, and the LM and the decoder of this frame wait for the encoder
    if ((rc = publish(d, F_IN, t, (hipStream_t)caller))) return rc;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[0], (hipStream_t)caller));
    stamp(d, t, S_IN, (hipStream_t)caller);
    if ((rc = await(d, F_IN, t, d->sE))) return rc;
    // ---- E: encode(t) into codes slot p (the host-kept gate above: step t-1 is in its depth-transformer phase)
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[1], d->sE));
    stamp(d, t, S_ENC0, d->sE);
    if ((rc = mmi_mimi_encode_step(d->mimi, pcm_in, d->codes[p], d->B, 1, d->sE))) return rc;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[2], d->sE));
    stamp(d, t, S_ENC1, d->sE);
    if ((rc = publish(d, F_ENC, t, d->sE))) return rc;
    // ---- L: LMGen.step(t) into tokens slot q, whose last reader decode(t - slots) was enqueued by an earlier submit
    // host-kept gate: decode(t-2) - step t-2 is complete, step t-1 in its depth-transformer phase - goes out now
    if ((rc = enqueue_decode(d, q_m2, true))) return rc;
    if (d->timeline && d->stamps) {      // the LM's wait for the encoder, with the clock at its begin and end
        MMI_LAUNCH(k_flag_wait_stamped, 1, 64, 0, d->sL, (const long*)flag_of(d, F_ENC), t + 1, stamp_of(d, t, S_WAIT0));
        MMI_CHECK_LAUNCH();
    } else if ((rc = await(d, F_ENC, t, d->sL))) return rc;
    if (t >= d->slots) MMI_HIP_CHECK(hipStreamWaitEvent(d->sL, d->ev_dec[q], 0));
    int ok = 0;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[3], d->sL));
    stamp(d, t, S_LM0, d->sL);
    d->phase_frame = t;
    if ((rc = mmi_lm_set_phase_callback(d->lm, phase_callback, d))) return rc;
    rc = mmi_lm_step(d->lm, d->codes[p], d->K, d->tokens[q], nullptr, nullptr, nullptr, d->B, &ok, d->sL);
    mmi_lm_set_phase_callback(d->lm, nullptr, nullptr);
    if (rc) return rc;
    if (d->timeline) MMI_HIP_CHECK(hipEventRecord(d->tl[5], d->sL));
    stamp(d, t, S_LM1, d->sL);
    if ((rc = publish(d, F_LM, t, d->sL))) return rc;
    MMI_HIP_CHECK(hipEventRecord(d->ev_lm[p], d->sL));
    d->pend[q] = Pending{true, ok != 0, pcm_out, tokens_out, t};
    // ---- D: this frame's decode is enqueued by submit(t+2) - beside the depth-transformer phase of step t+1 - or by join / flush
    if (valid) *valid = ok;
    d->frame += 1;
    return MMI_OK;
}
}  // namespace

extern "C" int mmi_duplex_submit(mmi_duplex* d, const float* pcm_in, float* pcm_out, int64_t* tokens_out, int32_t batch, int32_t* valid,
                                 mmi_stream caller) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d || !pcm_in || !pcm_out) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (d->failed) return mmi_fail(MMI_ERR_STATE, "mmi_duplex: an earlier call failed half-way; destroy the pipeline");
    if (batch != d->B) return mmi_fail(MMI_ERR_SHAPE, "mmi_duplex_submit: batch != the streaming batch");     // lm.py:679-682
    // LMGen's per-step hooks run on the stream the caller made current; the pipeline steps the LM on its own stream, so a hooked
    // step would race it: refused rather than silently wrong (drive a hooked LMGen through its own step entry point)
    if (mmi_lm_has_hooks(d->lm)) return mmi_fail(MMI_ERR_UNSUPPORTED, "mmi_duplex_submit: the LM has per-step hooks installed");
    return fail_sticky(d, submit_impl(d, pcm_in, pcm_out, tokens_out, valid, caller));
}

extern "C" int32_t mmi_duplex_batch(const mmi_duplex* d) { return d ? d->B : 0; }

extern "C" int mmi_duplex_join(mmi_duplex* d, mmi_stream caller) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (d->failed) return mmi_fail(MMI_ERR_STATE, "mmi_duplex: an earlier call failed half-way; destroy the pipeline");
    if (d->frame == 0) return MMI_OK;
    const long t = d->frame - 1;                  // the last frame: everything earlier precedes it on each stream
    const int q = (int)(t % d->slots), q_m1 = (int)((t + d->slots - 1) % d->slots);
    int rc;
    if ((rc = enqueue_decode(d, q_m1, false)) || (rc = enqueue_decode(d, q, false))) return fail_sticky(d, rc);     // a decode still held back for its gate: now
    hipStream_t s = (hipStream_t)caller;
    if ((rc = await(d, F_DEC, t, s))) return fail_sticky(d, rc);  // decode(t) implies step(t) implies encode(t)
    return MMI_OK;
}

extern "C" int mmi_duplex_flush(mmi_duplex* d) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (d->failed) return mmi_fail(MMI_ERR_STATE, "mmi_duplex: an earlier call failed half-way; destroy the pipeline");
    if (d->frame == 0) return MMI_OK;
    const long t = d->frame - 1;
    int rc;
    for (long f : {t - 1, t}) {                 // older frame first; the host sees the step complete, so the decode needs no device-side wait
        if (f < 0) continue;
        const int slot = (int)(f % d->slots);
        if (!d->pend[slot].live || d->pend[slot].frame != f) continue;
        MMI_HIP_CHECK(hipEventSynchronize(d->ev_lm[f & 1]));
        if ((rc = enqueue_decode(d, slot, true))) return fail_sticky(d, rc);
    }
    MMI_HIP_CHECK(hipEventSynchronize(d->ev_lm[t & 1]));
    MMI_HIP_CHECK(hipEventSynchronize(d->ev_dec[t % d->slots]));     // the decoder's stream order: every earlier decode precedes it
    return MMI_OK;
}

extern "C" int mmi_duplex_set_timeline(mmi_duplex* d, int32_t on) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d) return mmi_fail(MMI_ERR_INVALID, "null handle");
    if (on && !d->tl[0])
        for (hipEvent_t& e : d->tl) MMI_HIP_CHECK(hipEventCreate(&e));
    if (on && !d->stamps) {
        MMI_HIP_CHECK(hipMalloc((void**)&d->stamps, (size_t)4 * S_COUNT * sizeof(long)));
        MMI_HIP_CHECK(hipMemset(d->stamps, 0, (size_t)4 * S_COUNT * sizeof(long)));
    }
    d->timeline = on != 0;
    return MMI_OK;
}

extern "C" int mmi_duplex_get_stamps(mmi_duplex* d, double* ms40, int64_t* last_frame) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d || !ms40 || !last_frame) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!d->stamps) return mmi_fail(MMI_ERR_STATE, "the timeline was never switched on");
    for (hipStream_t s : {d->sE, d->sL, d->sD}) MMI_HIP_CHECK(hipStreamSynchronize(s));
    long raw[4 * S_COUNT];
    MMI_HIP_CHECK(hipMemcpy(raw, d->stamps, sizeof(raw), hipMemcpyDeviceToHost));
    int khz = 0;                                 // wall_clock64's rate (100 MHz on gfx950)
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d->device) != hipSuccess || khz <= 0) khz = 100000;
    long t0 = 0;
    for (long v : raw) if (v > 0 && (t0 == 0 || v < t0)) t0 = v;
    for (int i = 0; i < 4 * S_COUNT; ++i) ms40[i] = raw[i] > 0 ? (double)(raw[i] - t0) / (double)khz : -1.0;
    *last_frame = d->frame - 1;
    return MMI_OK;
}

extern "C" int mmi_duplex_get_timeline(mmi_duplex* d, float* ms) {
    MmiDeviceGuard dev_guard_(d ? d->device : -1);
    if (!d || !ms) return mmi_fail(MMI_ERR_INVALID, "null argument");
    if (!d->timeline || d->frame == 0) return mmi_fail(MMI_ERR_STATE, "no frame was submitted with the timeline on");
    for (hipStream_t s : {d->sE, d->sL, d->sD}) MMI_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 1; i < 8; ++i) {
        ms[i - 1] = -1.f;
        if (i == 4) continue;
        if (hipEventElapsedTime(&ms[i - 1], d->tl[0], d->tl[i]) != hipSuccess) ms[i - 1] = -1.f;
    }
    (void)hipGetLastError();
    return MMI_OK;
}
