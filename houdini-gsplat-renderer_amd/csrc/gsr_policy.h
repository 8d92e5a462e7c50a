// gsr_policy.h -- the host-side POLICIES of a context as small state machines (host code only, no HIP).
//
// None of them can change a pixel: whatever a policy answers, the frame is the depth-ordered composite of every splat that can
// reach it (culled frames verify themselves and are rendered again when a horizon broke).  What they decide
// is how much work a frame does.  They used to live as a dozen loose integers inside gsr_api.hip's frame_begin / frame_finish /
// frame_check; here every EVENT is a method, so that DESIGN.md section 4's state table can name it and tests/test_policy.py can
// drive every transition on the CPU, one by one (gsr_debug_policy in gsr_api.hip is the test door: a pure function of
// (state, event) -> state).  The randomised soak (tools/fuzz_parity.py) stays the net under their interactions.
//
// The reference has no counterpart: it streams every splat through the GL pipeline every frame
// (/root/reference/gsplat_plugin/src/GSplatRenderer.C:605-657); its only per-frame decision is "has the camera position moved?"
// (:165-186), which is the sort cache of gsr_api.hip.
#pragma once
#include <algorithm>
#include <cstdint>

// ---- occlusion culling against the previous frame's depth horizons (GSR_OPT_OCCLUSION_CULL = 1: the policy; >= 2: forced) ----
struct GsrCullPolicy {
    bool pays = false;          // the kernels' verdict on the frame before: >= 30 % of the drawing tiles went opaque in the first 70 % of their list
    bool weak = false;          // the last culled frame kept > 70 % of what an unculled frame keeps
    uint32_t vis_unculled = 0;  // splats kept by the last frame that was NOT culled (0 = none yet)
    int holdoff = 0;            // frames for which culling stays off
    int backoff = 8;            // the next hold-off after a horizon broke at full dilation: 8, 32, 128, 512, 1024
    int streak = 0;             // culled frames in a row that held
    int dilate = 2;             // tiles by which rects are widened before they are compared with the horizons
    int opt_dilate = 2;         // ... its starting (and smallest) value (GSR_OPT_CULL_DILATE)

    enum : int { HOLDOFF_WEAK = 256, DILATE_MAX = 16, STREAK_GOOD = 64, BACKOFF_FIRST = 8, BACKOFF_MAX = 1024 };

    // a new cloud: nothing of the previous one applies
    void on_upload() { pays = false; weak = false; vis_unculled = 0; holdoff = 0; backoff = BACKOFF_FIRST; streak = 0; dilate = opt_dilate; }
    void set_dilate_option(int v) { opt_dilate = v < 0 ? 0 : (v > 64 ? 64 : v); dilate = opt_dilate; }
    // may THIS frame be culled (given usable horizons)?  opt = GSR_OPT_OCCLUSION_CULL (1 = policy, 2 = whenever horizons exist)
    bool allows(int opt) const { return opt >= 2 || (pays && holdoff == 0); }
    // once per frame that could have been culled (after allows() was asked)
    void tick() { if (holdoff > 0) holdoff -= 1; }
    // the kernels' verdict arrives with the frame's pair count
    void on_kernel_verdict(bool tiles_go_opaque_early) { pays = tiles_go_opaque_early; }
    // how many splats the frame kept: an unculled frame sets the yardstick, a culled one is measured against it
    void on_kept(bool frame_was_culled, uint32_t kept, int opt)
    {
        if (!frame_was_culled) { vis_unculled = kept; return; }
        weak = vis_unculled > 0 && (unsigned long long)kept * 10ull > (unsigned long long)vis_unculled * 7ull;
        if (opt == 1 && weak) holdoff = HOLDOFF_WEAK;
    }
    // a culled frame checked itself: every tile stayed in front of its horizon ...
    void on_frame_held()
    {
        if (++streak >= STREAK_GOOD) { streak = 0; backoff = BACKOFF_FIRST; if (dilate > opt_dilate) dilate -= 1; }
    }
    // ... or a tile looked past it (the frame is rendered again without culling): first widen the neighbourhood the horizons are
    // compared over; only at DILATE_MAX tiles leave culling alone for a while, four times as long each time
    void on_horizon_broke()
    {
        if (dilate < DILATE_MAX) {
            dilate = std::max(2 * dilate, 1);
        } else {
            holdoff = std::max(holdoff, backoff);
            backoff = backoff >= 256 ? BACKOFF_MAX : 4 * backoff;
        }
        streak = 0;
    }
};

// ---- front-slab frames (GSR_OPT_FRONT_SLAB = 1): two phases where occlusion culling pays but no horizons apply ----
struct GsrSlabPolicy {
    int holdoff = 0;            // frames for which the regime stays off
    enum : int { HOLDOFF = 256, MIN_VISIBLE = 1500000 };
    void on_upload() { holdoff = 0; }
    // (forced: GSR_OPT_FRONT_SLAB >= 2 or GSR_OPT_OCCLUSION_CULL = 3)
    bool allows(bool forced, const GsrCullPolicy& cull) const { return forced || (cull.pays && cull.vis_unculled >= (uint32_t)MIN_VISIBLE && holdoff == 0); }
    void tick() { if (holdoff > 0) holdoff -= 1; }
    // both phases done: a slab behind which more than a third of an unculled frame still had to be drawn costs more than it saves
    void on_frame_done(uint32_t kept_both_phases, uint32_t vis_unculled)
    {
        if (vis_unculled > 0 && (unsigned long long)kept_both_phases * 3ull > (unsigned long long)vis_unculled) holdoff = HOLDOFF;
    }
};

// ---- the small-frame sort (GSR_OPT_LOCAL_SORT = 1), per frame slot: back-off for geometries that defeat it every frame ----
struct GsrLocalSortPolicy {
    int fails = 0;              // small-frame sorts in a row that gave a bucket up
    int holdoff = 0;            // frames this slot stays with the three global passes
    enum : int { FAILS_MAX = 3, HOLDOFF = 64, MAX_KEPT = 500000 };
    void on_upload() { fails = 0; holdoff = 0; }
    // once per frame, before the sort is chosen; returns "held": the global passes whatever the prediction says (opt >= 2 forces the small-frame sort)
    bool begin_frame(int opt, bool static_redraw)
    {
        if (fails >= FAILS_MAX) { fails = 0; holdoff = HOLDOFF; }
        const bool held = holdoff > 0 && opt < 2;
        if (holdoff > 0 && !static_redraw) holdoff -= 1;
        return held;
    }
    void on_sort_result(bool gave_a_bucket_up) { fails = gave_a_bucket_up ? fails + 1 : 0; }
};

// The test door (gsr_debug_policy): state <-> 16 ints
//   [0] cull.pays [1] cull.weak [2] cull.vis_unculled [3] cull.holdoff [4] cull.backoff [5] cull.streak [6] cull.dilate [7] cull.opt_dilate
//   [8] slab.holdoff [9] local.fails [10] local.holdoff [11] (out) the event's answer (allows / held), else 0
#define GSR_POLICY_STATE_INTS 16
enum GsrPolicyEvent : int {
    GSR_PE_UPLOAD = 0,            // a new cloud
    GSR_PE_CULL_ALLOWS = 1,       // a = GSR_OPT_OCCLUSION_CULL                       -> [11]
    GSR_PE_CULL_TICK = 2,
    GSR_PE_KERNEL_VERDICT = 3,    // a = tiles go opaque early (0 / 1)
    GSR_PE_KEPT = 4,              // a = kept, b = frame was culled (0 / 1) | opt << 1
    GSR_PE_FRAME_HELD = 5,
    GSR_PE_HORIZON_BROKE = 6,
    GSR_PE_SLAB_ALLOWS = 7,       // a = forced (0 / 1)                                -> [11]
    GSR_PE_SLAB_TICK = 8,
    GSR_PE_SLAB_DONE = 9,         // a = kept by both phases
    GSR_PE_LOCAL_BEGIN = 10,      // a = GSR_OPT_LOCAL_SORT, b = static redraw (0 / 1) -> [11] held
    GSR_PE_LOCAL_RESULT = 11,     // a = gave a bucket up (0 / 1)
    GSR_PE_SET_DILATE = 12,       // a = GSR_OPT_CULL_DILATE
};
inline void gsr_policy_apply(int32_t* st, int event, long long a, long long b)
{
    GsrCullPolicy c; GsrSlabPolicy s; GsrLocalSortPolicy l;
    c.pays = st[0] != 0; c.weak = st[1] != 0; c.vis_unculled = (uint32_t)st[2]; c.holdoff = st[3]; c.backoff = st[4]; c.streak = st[5]; c.dilate = st[6]; c.opt_dilate = st[7];
    s.holdoff = st[8]; l.fails = st[9]; l.holdoff = st[10];
    int answer = 0;
    switch (event) {
    case GSR_PE_UPLOAD: c.on_upload(); s.on_upload(); l.on_upload(); break;
    case GSR_PE_CULL_ALLOWS: answer = c.allows((int)a) ? 1 : 0; break;
    case GSR_PE_CULL_TICK: c.tick(); break;
    case GSR_PE_KERNEL_VERDICT: c.on_kernel_verdict(a != 0); break;
    case GSR_PE_KEPT: c.on_kept((b & 1) != 0, (uint32_t)a, (int)(b >> 1)); break;
    case GSR_PE_FRAME_HELD: c.on_frame_held(); break;
    case GSR_PE_HORIZON_BROKE: c.on_horizon_broke(); break;
    case GSR_PE_SLAB_ALLOWS: answer = s.allows(a != 0, c) ? 1 : 0; break;
    case GSR_PE_SLAB_TICK: s.tick(); break;
    case GSR_PE_SLAB_DONE: s.on_frame_done((uint32_t)a, c.vis_unculled); break;
    case GSR_PE_LOCAL_BEGIN: answer = l.begin_frame((int)a, b != 0) ? 1 : 0; break;
    case GSR_PE_LOCAL_RESULT: l.on_sort_result(a != 0); break;
    case GSR_PE_SET_DILATE: c.set_dilate_option((int)a); break;
    default: answer = -1; break;
    }
    st[0] = c.pays; st[1] = c.weak; st[2] = (int32_t)c.vis_unculled; st[3] = c.holdoff; st[4] = c.backoff; st[5] = c.streak; st[6] = c.dilate; st[7] = c.opt_dilate;
    st[8] = s.holdoff; st[9] = l.fails; st[10] = l.holdoff; st[11] = answer;
}
