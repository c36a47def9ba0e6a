"""Failure paths a careful user would fall into: a kernel that gives up on the device must say so —
never a silent wrong result behind SCG_OK."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_util import dev, make_pair
from util import HP, chain_classifiers, random_states, random_weights
from skill_chaining_with_graphs_amd._lib import ScgError

pytestmark = pytest.mark.gpu


def _fit_problem(seed, sizes):
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    xy = rng.random((off[-1], 2)).astype(np.float32)
    lab = (((xy[:, 0] - 0.5) ** 2 + (xy[:, 1] - 0.45) ** 2) < 0.3 ** 2).astype(np.uint8)
    w = (rng.standard_normal((len(sizes), 8)) * 0.05).astype(np.float32)
    return xy, lab, off, w


def test_async_error_is_sticky_refuses_every_launch_and_clears():
    ctx, orc, m = make_pair("pinball_simple", 256)
    assert ctx.async_status(synchronize=True) == 0
    ctx.lib.scg_debug_raise_async(ctx._ctx, C.c_uint32(0x1 | (0x100 << 2)))     # what fit_kernel writes when it gives up
    x = torch.rand(256, device="cuda:0")
    with pytest.raises(ScgError, match="scg_fit_initiation gave up"):
        ctx.async_status()
    with pytest.raises(ScgError, match="earlier launch failed"):
        ctx.classifier_predict(x, x.clone(), torch.zeros(8, device="cuda:0"))
    with pytest.raises(ScgError, match="problem mask 0x4"):
        ctx.features([x, x.clone(), x.clone(), x.clone()])
    xy, lab, off, w = _fit_problem(1, [500])
    w_d = dev(w.copy())
    with pytest.raises(ScgError):
        ctx.fit_initiation(dev(xy).view(-1), dev(lab), dev(off), w_d.view(-1), iters=5)
    assert np.array_equal(w_d.cpu().numpy(), w)                                    # nothing ran
    ctx.clear_async_error()
    assert ctx.async_status(synchronize=True) == 0
    ctx.fit_initiation(dev(xy).view(-1), dev(lab), dev(off), w_d.view(-1), iters=5, lr=2.0, l2=1e-3)
    orc.fit_initiation(xy, lab, off, w, iters=5, lr=2.0, l2=1e-3)
    assert np.array_equal(w_d.cpu().numpy(), w)


def _busy_stream(ms_target=150):
    """A side stream that keeps every CU busy for a while (large f32 GEMMs from the vendor library)."""
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda:0")
    b = torch.randn(4096, 4096, device="cuda:0")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(max(1, ms_target // 2)):          # ~1.5-2 ms each
            a = (a @ b) * 1e-3
    return side, a


def test_fit_is_exact_while_another_stream_keeps_the_card_busy():
    """The fit's eight workgroups per problem exchange partial sums every iteration and need to run together; with the
    card shared they may start late. Late partners are waited for (wall clock), so the result is still bit-exact."""
    ctx, orc, m = make_pair("pinball_simple", 256)
    xy, lab, off, w = _fit_problem(2, [30000, 0, 7000, 1, 12000])
    w_o = w.copy()
    orc.fit_initiation(xy, lab, off, w_o, iters=60, lr=2.0, l2=1e-3)
    xy_d, lab_d, off_d = dev(xy).view(-1), dev(lab), dev(off)
    for rep in range(3):
        side, keep = _busy_stream()
        w_d = dev(w.copy())
        ctx.fit_initiation(xy_d, lab_d, off_d, w_d.view(-1), iters=60, lr=2.0, l2=1e-3)     # waits for its own stream + status
        assert np.array_equal(w_d.cpu().numpy(), w_o), rep
        side.synchronize()
    assert ctx.async_status(synchronize=True) == 0


def test_fit_with_no_patience_is_exact_or_reports_and_leaves_the_rows_alone():
    """scg_set_fit_timeout(0): a partner that is not there at once makes the problem give up. Whatever happens, a row of w
    is either the exact result or untouched, and an abandoned fit raises — never SCG_OK over a half-done row."""
    ctx, orc, m = make_pair("pinball_simple", 256)
    xy, lab, off, w = _fit_problem(3, [20000, 9000, 4000])
    w_o = w.copy()
    orc.fit_initiation(xy, lab, off, w_o, iters=40, lr=2.0, l2=1e-3)
    ctx.set_fit_timeout(0.0)
    side, keep = _busy_stream()
    w_d = dev(w.copy())
    try:
        ctx.fit_initiation(dev(xy).view(-1), dev(lab), dev(off), w_d.view(-1), iters=40, lr=2.0, l2=1e-3)
        gave_up = False
    except ScgError as e:
        gave_up = True
        assert "gave up" in str(e)
    side.synchronize()
    torch.cuda.synchronize()
    got = w_d.cpu().numpy()
    for q in range(3):
        assert np.array_equal(got[q], w_o[q]) or (gave_up and np.array_equal(got[q], w[q])), q
    if not gave_up:
        assert np.array_equal(got, w_o)
    ctx.clear_async_error()
    ctx.set_fit_timeout(2.0)


def test_step_handoff_failure_is_reported_like_any_asynchronous_failure():
    """A hand-off poll inside the step kernel that runs out ORs SCG_ASYNC_STEP_HANDOFF into the status word:
    the next entry point refuses with SCG_ERR_ASYNC and names the step, and clearing re-arms the context."""
    from skill_chaining_with_graphs_amd._lib import ASYNC_STEP_HANDOFF
    from skill_chaining_with_graphs_amd.core import EnvState
    ctx, orc, m = make_pair("pinball_simple", 300, n_options=1)
    st = EnvState(300, ctx.device, m)
    W = torch.zeros(2 * 5 * 1296, device="cuda:0"); clf = torch.zeros(2 * 8, device="cuda:0")
    ctx.step(st, W, clf, 0, 0)
    assert ctx.async_status(synchronize=True) == 0                        # a healthy step raises nothing
    ctx.lib.scg_debug_raise_async(ctx._ctx, C.c_uint32(ASYNC_STEP_HANDOFF))  # what td_kernel writes when a bounded poll runs out
    with pytest.raises(ScgError, match="hand-off poll"):
        ctx.step(st, W, clf, 0, 1)
    ctx.clear_async_error()
    ctx.step(st, W, clf, 0, 1)
    assert ctx.async_status(synchronize=True) == 0


def test_announced_trigger_buffers_are_held_and_validated():
    """ADVICE r3 (medium) / VERDICT r3 item 7: scg_arm_collect keeps raw device pointers that every later scg_step reads.
    (i) the Python wrapper holds the announced tensors itself, so dropping the caller's references is harmless;
    (ii) at the C-ABI, a step with an announced buffer that is no longer device memory is refused (SCG_ERR_STATE) instead
    of launched, and the trigger is disarmed."""
    import gc
    from skill_chaining_with_graphs_amd.core import EnvState, _ptr
    ctx, orc, m = make_pair("pinball_simple", 512, n_options=1)
    st = EnvState(512, ctx.device, m)
    W = torch.zeros(2 * 5 * 1296, device="cuda:0"); clf = torch.zeros(2 * 8, device="cuda:0")
    ctx.set_trace_buffers(16)
    xy = torch.zeros(2 * 1024, device="cuda:0"); lab = torch.zeros(1024, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda:0"); prev = torch.zeros(512, dtype=torch.uint8, device="cuda:0")
    ctx.step(st, W, clf, 0, 0)
    ctx.collect_examples(1, prev, 4, 4, xy, lab, cnt)                      # rearm=True: the library now holds prev / cnt addresses
    assert ctx._armed is not None and ctx._armed[0] is prev and ctx._armed[1] is cnt
    del prev, cnt
    gc.collect(); torch.cuda.empty_cache()
    ctx.step(st, W, clf, 0, 1)                                             # reads the announced buffers: still alive inside ctx._armed
    torch.cuda.synchronize()
    prev2, cnt2 = torch.zeros(512, dtype=torch.uint8, device="cuda:0"), torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ctx.collect_examples(1, prev2, 4, 4, xy, lab, cnt2)                    # new buffers: the old pair is released
    assert ctx._armed[0] is prev2
    ctx.disarm_collect()
    assert ctx._armed is None
    # (ii) straight at the C-ABI: announce a buffer, free it for real, step
    big = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")        # its own 64 MiB segment: goes back to the driver on empty_cache
    cnt3 = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    ctx._call("scg_arm_collect", C.c_uint32(1), C.c_void_p(big.data_ptr()), 4, 4, _ptr(cnt3))
    del big
    gc.collect(); torch.cuda.empty_cache()
    with pytest.raises(ScgError, match="no longer device memory"):
        ctx.step(st, W, clf, 0, 2)
    ctx.step(st, W, clf, 0, 2)                                             # disarmed by the refusal: steps run again
    assert ctx.async_status(synchronize=True) == 0


def test_a_real_handoff_give_up_voids_the_step_and_is_reported():
    """the give-up path of the step kernel itself, not its imitation through scg_debug_raise_async. The fault-injection
    build (csrc/Makefile `faultinj`) drops ONE hand-off on purpose — helper wave 0 of block 1 never arrives at step t = 0x7e57 — so
    the other helper waves' bounded poll runs out for real. Expected: the block goes inert, the reduce launch of the same step
    neither applies the update nor commits results (state and weights stay what the previous step left), the status word names the
    step, every later launch is refused until the error is cleared, and the context runs on afterwards."""
    import os
    import skill_chaining_with_graphs_amd as scg
    from skill_chaining_with_graphs_amd.core import EnvState, ScgContext
    path = os.path.join(os.path.dirname(scg.LIB_PATH), "libscg_hip_faultinj.so")
    if not os.path.exists(path):
        pytest.skip("fault-injection build missing (make -C skill-chaining-with-graphs_amd/csrc faultinj)")
    m = scg.load_map("pinball_simple")
    n, T = 1024, 0x7e57
    ctx = ScgContext(n, 1, m, seed=3, block_envs=256, library=path, **HP)
    st = EnvState(n, ctx.device, m)
    x, y, vx, vy = random_states(m, n, 5, vmax=1.0)
    for k, v in (("x", x), ("y", y), ("vx", vx), ("vy", vy)):
        getattr(st, k).copy_(dev(v))
    W = dev(random_weights(2, 7, std=0.05)).view(-1)
    clf = dev(chain_classifiers(m, 1)).view(-1)
    for t in (T - 2, T - 1):
        ctx.step(st, W, clf, 0b10, t)
    assert ctx.async_status(synchronize=True) == 0
    keys = ("x", "y", "vx", "vy", "option_id", "opt_steps", "ep_steps", "qcache", "action", "reward", "done")
    before = {k: getattr(st, k).clone() for k in keys}
    W_before = W.clone()
    ctx.step(st, W, clf, 0b10, T)                                          # launches fine; gives up on the device
    with pytest.raises(ScgError, match="hand-off poll"):
        ctx.async_status(synchronize=True)
    for k in keys:
        assert torch.equal(getattr(st, k), before[k]), f"{k} was written by a voided step"
    assert torch.equal(W, W_before), "a voided step applied its update"
    with pytest.raises(ScgError, match="hand-off poll"):                   # sticky
        ctx.step(st, W, clf, 0b10, T + 1)
    ctx.clear_async_error()
    ctx.step(st, W, clf, 0b10, T + 1)
    assert ctx.async_status(synchronize=True) == 0
    assert not torch.equal(st.x, before["x"])                              # and the context steps again
