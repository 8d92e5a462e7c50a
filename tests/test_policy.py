"""The host-side policies, one transition per test, on the CPU (csrc/gsr_policy.h through gsr_debug_policy; DESIGN.md section 4's
state table names exactly these events).  No policy can change a pixel -- the GPU exactness tests and the soak prove that -- so what is
pinned here is only WHEN the library does how much work: the hold-offs, the back-off ladder, the dilation radius, the slab's and the
small-frame sort's own brakes.  A pure function (state, event, a, b) -> state: no context, no GPU."""
import numpy as np
import pytest

UPLOAD, CULL_ALLOWS, CULL_TICK, KERNEL_VERDICT, KEPT, FRAME_HELD, HORIZON_BROKE, SLAB_ALLOWS, SLAB_TICK, SLAB_DONE, LOCAL_BEGIN, LOCAL_RESULT, SET_DILATE = range(13)
F = ("cull_pays", "cull_weak", "vis_unculled", "cull_holdoff", "cull_backoff", "cull_streak", "cull_dilate", "opt_dilate", "slab_holdoff",
     "local_fails", "local_holdoff", "answer")


class Policy:
    def __init__(self, lib):
        self.L = lib
        self.st = np.zeros(16, np.int32)
        self.st[4], self.st[6], self.st[7] = 8, 2, 2          # the defaults of a fresh context
        self.ev(UPLOAD)

    def ev(self, event, a=0, b=0):
        assert self.L.gsr_debug_policy(self.st.ctypes.data, event, a, b) == 0
        return int(self.st[11])

    def __getattr__(self, name):
        return int(self.st[F.index(name)])


@pytest.fixture
def pol(pkg):
    return Policy(pkg.load_library())


def test_fresh_cloud_does_not_cull_until_the_kernels_say_it_pays(pol):
    assert pol.ev(CULL_ALLOWS, 1) == 0                       # policy mode: no verdict yet
    assert pol.ev(CULL_ALLOWS, 2) == 1                       # forced mode asks nobody
    pol.ev(KERNEL_VERDICT, 1)
    assert pol.cull_pays == 1 and pol.ev(CULL_ALLOWS, 1) == 1
    pol.ev(KERNEL_VERDICT, 0)                                # (scenes where nothing is hidden: a wall, a terrain under open sky)
    assert pol.ev(CULL_ALLOWS, 1) == 0


def test_unculled_frame_sets_the_yardstick_and_a_weak_culled_frame_holds_culling_off(pol):
    pol.ev(KERNEL_VERDICT, 1)
    pol.ev(KEPT, 1_000_000, 0 | (1 << 1))                    # an unculled frame kept 1 M splats
    assert pol.vis_unculled == 1_000_000 and pol.cull_weak == 0
    pol.ev(KEPT, 700_000, 1 | (1 << 1))                      # a culled frame that keeps exactly 70 %: not weak
    assert pol.cull_weak == 0 and pol.cull_holdoff == 0
    pol.ev(KEPT, 700_001, 1 | (1 << 1))                      # ... one more: weak -> 256 frames without culling (policy mode only)
    assert pol.cull_weak == 1 and pol.cull_holdoff == 256
    assert pol.ev(CULL_ALLOWS, 1) == 0 and pol.ev(CULL_ALLOWS, 2) == 1
    for _ in range(255):
        pol.ev(CULL_TICK)
    assert pol.cull_holdoff == 1 and pol.ev(CULL_ALLOWS, 1) == 0
    pol.ev(CULL_TICK)
    assert pol.cull_holdoff == 0 and pol.ev(CULL_ALLOWS, 1) == 1
    pol.ev(CULL_TICK)                                        # never below zero
    assert pol.cull_holdoff == 0
    p2 = Policy(pol.L)
    p2.ev(KEPT, 1_000_000, 0 | (2 << 1)); p2.ev(KEPT, 900_000, 1 | (2 << 1))   # forced mode records the weakness but holds nothing off
    assert p2.cull_weak == 1 and p2.cull_holdoff == 0


def test_a_broken_horizon_first_widens_the_dilation_radius(pol):
    assert pol.cull_dilate == 2
    for want in (4, 8, 16):
        pol.ev(FRAME_HELD); pol.ev(FRAME_HELD)
        assert pol.cull_streak == 2
        pol.ev(HORIZON_BROKE)
        assert pol.cull_dilate == want and pol.cull_holdoff == 0 and pol.cull_streak == 0
    p0 = Policy(pol.L)
    p0.ev(SET_DILATE, 0)                                     # GSR_OPT_CULL_DILATE = 0: the ladder starts at 0 -> 1 -> 2 ...
    assert p0.cull_dilate == 0 and p0.opt_dilate == 0
    p0.ev(HORIZON_BROKE); assert p0.cull_dilate == 1
    p0.ev(HORIZON_BROKE); assert p0.cull_dilate == 2


def test_at_full_dilation_a_broken_horizon_holds_culling_off_on_a_x4_ladder(pol):
    for _ in range(3):
        pol.ev(HORIZON_BROKE)
    assert pol.cull_dilate == 16
    ladder = []
    for _ in range(6):
        pol.ev(HORIZON_BROKE)
        ladder.append((pol.cull_holdoff, pol.cull_backoff))
        while pol.cull_holdoff:
            pol.ev(CULL_TICK)
    assert ladder == [(8, 32), (32, 128), (128, 512), (512, 1024), (1024, 1024), (1024, 1024)]
    assert pol.cull_dilate == 16                            # the radius does not grow past 16 tiles


def test_sixty_four_frames_that_hold_shrink_the_radius_and_reset_the_ladder(pol):
    for _ in range(4):
        pol.ev(HORIZON_BROKE)                                # radius 16, back-off 32
    assert (pol.cull_dilate, pol.cull_backoff) == (16, 32)
    for _ in range(63):
        pol.ev(FRAME_HELD)
    assert pol.cull_streak == 63 and pol.cull_dilate == 16
    pol.ev(FRAME_HELD)
    assert (pol.cull_streak, pol.cull_dilate, pol.cull_backoff) == (0, 15, 8)
    for _ in range(64 * 20):
        pol.ev(FRAME_HELD)
    assert pol.cull_dilate == 2                             # back to the option's value, never below it
    pol.ev(HORIZON_BROKE)
    for _ in range(63):
        pol.ev(FRAME_HELD)
    pol.ev(HORIZON_BROKE)                                    # a break in between starts the count again
    assert pol.cull_streak == 0


def test_a_new_cloud_forgets_everything_but_the_option(pol):
    pol.ev(SET_DILATE, 3)
    pol.ev(KERNEL_VERDICT, 1); pol.ev(KEPT, 2_000_000, 2); pol.ev(KEPT, 1_900_000, 3)
    for _ in range(6):
        pol.ev(HORIZON_BROKE)
    pol.ev(SLAB_DONE, 1_500_000); pol.ev(LOCAL_RESULT, 1)
    assert pol.cull_holdoff and pol.slab_holdoff and pol.local_fails and pol.cull_dilate > 3
    pol.ev(UPLOAD)
    assert [pol.cull_pays, pol.cull_weak, pol.vis_unculled, pol.cull_holdoff, pol.cull_backoff, pol.cull_streak, pol.cull_dilate, pol.opt_dilate,
            pol.slab_holdoff, pol.local_fails, pol.local_holdoff] == [0, 0, 0, 0, 8, 0, 3, 3, 0, 0, 0]


def test_front_slab_needs_a_heavy_scene_where_culling_pays(pol):
    assert pol.ev(SLAB_ALLOWS, 0) == 0 and pol.ev(SLAB_ALLOWS, 1) == 1      # forced (GSR_OPT_FRONT_SLAB = 2 / OCCLUSION_CULL = 3) asks nobody
    pol.ev(KERNEL_VERDICT, 1); pol.ev(KEPT, 1_499_999, 2)
    assert pol.ev(SLAB_ALLOWS, 0) == 0                      # C3's 0.8 M visible splats do not repay two phases' launches
    pol.ev(KEPT, 1_500_000, 2)
    assert pol.ev(SLAB_ALLOWS, 0) == 1
    pol.ev(KERNEL_VERDICT, 0)
    assert pol.ev(SLAB_ALLOWS, 0) == 0


def test_a_slab_that_keeps_more_than_a_third_holds_itself_off(pol):
    pol.ev(KERNEL_VERDICT, 1); pol.ev(KEPT, 3_000_000, 2)
    pol.ev(SLAB_DONE, 1_000_000)                             # exactly a third: fine
    assert pol.slab_holdoff == 0
    pol.ev(SLAB_DONE, 1_000_001)                             # B1, the ball from afar: 0.46 -> plain frames for 256 frames
    assert pol.slab_holdoff == 256 and pol.ev(SLAB_ALLOWS, 0) == 0 and pol.ev(SLAB_ALLOWS, 1) == 1
    for _ in range(256):
        pol.ev(SLAB_TICK)
    assert pol.slab_holdoff == 0 and pol.ev(SLAB_ALLOWS, 0) == 1
    assert pol.cull_holdoff == 0                             # (its own brake: occlusion culling against the previous frame is not touched)


def test_small_frame_sort_backs_off_after_three_failures_in_a_row(pol):
    assert pol.ev(LOCAL_BEGIN, 1, 0) == 0
    pol.ev(LOCAL_RESULT, 1); pol.ev(LOCAL_RESULT, 1)
    pol.ev(LOCAL_RESULT, 0)                                  # a success in between starts the count again
    assert pol.local_fails == 0
    for _ in range(3):
        assert pol.ev(LOCAL_BEGIN, 1, 0) == 0
        pol.ev(LOCAL_RESULT, 1)
    assert pol.local_fails == 3
    assert pol.ev(LOCAL_BEGIN, 1, 0) == 1                    # held: 64 frames on the three global passes ...
    assert (pol.local_fails, pol.local_holdoff) == (0, 63)
    assert pol.ev(LOCAL_BEGIN, 1, 1) == 1 and pol.local_holdoff == 63        # ... static redraws do not count
    assert pol.ev(LOCAL_BEGIN, 2, 0) == 0 and pol.local_holdoff == 62        # GSR_OPT_LOCAL_SORT = 2 forces it through (but the frames still count)
    for _ in range(61):
        assert pol.ev(LOCAL_BEGIN, 1, 0) == 1
    assert pol.local_holdoff == 1
    assert pol.ev(LOCAL_BEGIN, 1, 0) == 1 and pol.local_holdoff == 0
    assert pol.ev(LOCAL_BEGIN, 1, 0) == 0


def test_unknown_event_is_refused(pol):
    assert pol.L.gsr_debug_policy(pol.st.ctypes.data, 99, 0, 0) != 0
    assert pol.L.gsr_debug_policy(None, 0, 0, 0) != 0
