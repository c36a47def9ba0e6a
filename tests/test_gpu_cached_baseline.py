"""SPEC §5.4, the cached baseline (SCG_STEP_CACHED_QSA): Q(s, a) of the root's items and of a block's prefix option's own items comes
from what the previous step evaluated (one update stale) instead of a second evaluation under the current weights. The parity cases of
the other files are run again with the mode on BOTH sides (gpu_util.set_cached_baseline couples the oracle's cache validity to what
the test does to the context): fused rollouts, acting-only steps and invalidations in between (exact steps that re-fill the cache),
gestation (off-policy items stay exact), ragged batches, the small-block builds; then the agent-level contract."""
import ctypes as C

import numpy as np
import pytest
import torch

import gpu_util
import sc_oracle
import test_gpu_parity as P
import test_gpu_scale as S
import test_gpu_stress as T
import test_outer_loop as L

pytestmark = pytest.mark.gpu


@pytest.fixture
def cached():
    gpu_util.set_cached_baseline(True)
    yield
    gpu_util.set_cached_baseline(False)


def test_fused_rollouts_bit_exact(cached):
    P.test_fused_step_rollout_bit_exact("pinball_simple", 4096, 1, 10)
    P.test_fused_step_rollout_bit_exact("pinball_maze", 1000, 5, 12)
    P.test_fused_step_rollout_bit_exact("pinball_simple", 1, 0, 25)


def test_the_mode_changes_the_update_and_only_the_update(cached):
    """Same seeds with and without the mode: identical first step (the cache is being filled), different weights afterwards,
    identical integer outputs as long as the greedy actions agree (they do over these few steps)."""
    from gpu_util import dev, make_pair, state_to_device
    from util import chain_classifiers, random_states, random_weights
    n, nopt, mask = 2000, 2, 0b110
    res = []
    for on in (True, False):
        gpu_util.set_cached_baseline(on)
        ctx, orc, m = make_pair("pinball_simple", n, n_options=nopt, seed=3, enabled_mask=mask)
        st = sc_oracle.new_state(n, m)
        x, y, vx, vy = random_states(m, n, 5, vmax=1.0)
        st["x"][:], st["y"][:], st["vx"][:], st["vy"][:] = x, y, vx, vy
        W = random_weights(nopt + 1, 6, std=0.05)
        st_d, W_d, clf_d = state_to_device(st, ctx), dev(W.copy()), dev(chain_classifiers(m, nopt))
        snaps = []
        for t in range(4):
            ctx.step(st_d, W_d.view(-1), clf_d.view(-1), mask, t)
            snaps.append(W_d.cpu().numpy().copy())
        res.append(snaps)
        assert ctx.baseline_cache_valid() == on
    assert np.array_equal(res[0][0], res[1][0])                # step 0: exact on both (nothing cached yet)
    assert not np.array_equal(res[0][1], res[1][1])            # step 1: the baseline is one update stale
    assert np.allclose(res[0][3], res[1][3], rtol=0.2, atol=2e-2)


def test_acting_only_steps_invalidations_and_split_apply_bit_exact(cached):
    P.test_fused_step_act_only_and_split_apply()
    P.test_env_order_prepared_by_the_previous_step_and_invalidated_on_outside_writes()
    for n, dist in [(129, "uniform"), (1000, "all3"), (1000, "heavy"), (4100, "one_each"), (2048, "no_root")]:
        P.test_env_order_layouts_bit_exact(n, dist)


def test_random_configurations_with_gestation_bit_exact(cached):
    rng = np.random.default_rng(4242)
    for _ in range(16):
        n = int(rng.choice([1, 63, 65, 255, 256, 257, 511, 1000, 1500, 3000]))
        T.test_random_configuration_rollout_bit_exact(n, int(rng.integers(0, 6)), str(rng.choice(["pinball_simple", "pinball_maze", "pinball_empty"])),
                                                      int(rng.integers(0, 1 << 20)))


def test_long_rollouts_bit_exact(cached):
    T.test_long_rollout_bit_exact()
    import long_parity
    goals, wmax = long_parity.run(400, 2048, 3, check_every=20)
    assert goals > 100 and np.isfinite(wmax)


def test_gestation_trace_collect_and_many_workgroups_bit_exact(cached):
    L.test_gestation_and_device_side_collect_bit_exact_on_gpu()
    S.test_more_than_256_workgroups_bit_exact()


@pytest.mark.parametrize("block", [64, 128])
def test_small_block_builds_bit_exact(cached, block):
    gpu_util.set_block_envs(block)
    try:
        P.test_fused_step_rollout_bit_exact("pinball_maze", 1000, 5, 12)
        T.test_random_configuration_rollout_bit_exact(500, 4, "pinball_simple", 99)
    finally:
        gpu_util.set_block_envs(None)


def test_agent_checkpoint_resume_continues_bit_identically_in_the_mode(tmp_path):
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    from util import HP, chain_classifiers, random_weights

    def make():
        ag = SkillChainingAgent("pinball_simple", 3000, 3, seed=21, cached_baseline=True, **HP)
        ag.clf.copy_(torch.as_tensor(chain_classifiers(ag.map, 3), device="cuda:0"))
        ag.enabled_mask = 0b1110
        ag.W.copy_(torch.as_tensor(random_weights(4, 2, std=0.05), device="cuda:0"))
        return ag

    a = make()
    a.domain.reset_random(seed=4, v_max=1.0)
    for _ in range(9):
        a.step_batch()
    assert a.ctx.baseline_cache_valid()
    path = str(tmp_path / "agent.pt")
    a.save(path)
    for _ in range(6):
        a.step_batch()
    b = make()
    b.load(path)
    assert b.ctx.baseline_cache_valid()                        # restored: the first step after the load uses it like a's 10th did
    for _ in range(6):
        b.step_batch()
    for f in SkillChainingAgent._STATE_FIELDS:
        assert torch.equal(getattr(a.state, f), getattr(b.state, f)), f
    assert torch.equal(a.W, b.W) and torch.equal(a.ctx.baseline_cache, b.ctx.baseline_cache)
    b.step_batch(learn=False)                                  # an acting-only step leaves the cache behind
    assert not b.ctx.baseline_cache_valid()
    b.step_batch()
    assert b.ctx.baseline_cache_valid()
    b.domain.reset_random(seed=5, v_max=1.0)                   # outside writes -> invalidate_order -> exact step next
    assert not b.ctx.baseline_cache_valid()


def test_the_flag_without_a_cache_is_refused():
    from skill_chaining_with_graphs_amd import SkillChainingAgent, _lib
    from util import HP
    ag = SkillChainingAgent("pinball_simple", 512, 0, seed=1, **HP)
    ag.step_batch()
    st = ag.state
    ptrs = [C.c_void_p(t.data_ptr()) for t in (st.x, st.y, st.vx, st.vy, st.option_id, st.opt_steps, st.ep_steps, st.qcache, st.action,
                                                 st.reward, st.done, ag.W, ag.clf)]
    rc = ag.ctx.lib.scg_step(ag.ctx._ctx, *ptrs, C.c_uint32(0), C.c_uint64(1), C.c_uint32(_lib.STEP_LEARN | _lib.STEP_APPLY | _lib.STEP_CACHED_QSA), None)
    assert rc == -4 and b"scg_set_baseline_cache" in ag.ctx.lib.scg_last_error(ag.ctx._ctx)
    ag.step_batch()                                            # the context is still usable


def test_a_freed_cache_buffer_is_refused_not_dereferenced():
    """The baseline cache is the caller's buffer, read and written by every flagged step (the Python context holds it; a C-ABI
    caller could free it): a step whose cache is no longer device memory is refused (SCG_ERR_STATE) and the mode turned off."""
    import gc
    from skill_chaining_with_graphs_amd import ScgError, SkillChainingAgent
    from util import HP
    ag = SkillChainingAgent("pinball_simple", 512, 0, seed=1, cached_baseline=True, **HP)
    ag.step_batch(); ag.step_batch()
    assert ag.ctx.baseline_cache_valid()
    big = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")        # its own segment: goes back to the driver on empty_cache
    ag.ctx._call("scg_set_baseline_cache", C.c_void_p(big.data_ptr()), C.c_int32(0))
    del big
    gc.collect(); torch.cuda.empty_cache()
    with pytest.raises(ScgError, match="no longer device memory"):
        ag.step_batch()
    ag.ctx.baseline_cache = None                               # what the library did on its side: the mode is off
    ag.step_batch()
    assert ag.ctx.async_status(synchronize=True) == 0 and not ag.ctx.baseline_cache_valid()
