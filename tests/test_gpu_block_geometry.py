"""SPEC §5's block size (envs per workgroup = per accumulation chain) is a build parameter: csrc/Makefile builds the library for
64-, 128- and 256-env blocks from one source, a context picks one (block_envs=...), and each build is held bit for bit to the
oracle BUILT FOR THE SAME block size. The 256-env build is what every other GPU test runs; here the parity cases are run again
under the two small-batch builds (1.8x at BASELINE configs[1])."""
import numpy as np
import pytest

import gpu_util
import test_gpu_parity as P
import test_gpu_scale as S
import test_gpu_stress as T
import test_outer_loop as L

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[64, 128])
def small_blocks(request):
    gpu_util.set_block_envs(request.param)
    yield request.param
    gpu_util.set_block_envs(None)


def test_fused_rollouts_bit_exact(small_blocks):
    P.test_fused_step_rollout_bit_exact("pinball_simple", 4096, 1, 10)         # BASELINE configs[1]
    P.test_fused_step_rollout_bit_exact("pinball_maze", 1000, 5, 12)           # full chain, ragged last block
    P.test_fused_step_rollout_bit_exact("pinball_simple", 1, 0, 25)


def test_un_fused_q_update_and_env_order_layouts_bit_exact(small_blocks):
    P.test_q_update_bit_exact(700, 2)
    for n, dist in [(129, "uniform"), (1000, "all3"), (1000, "heavy"), (4100, "one_each"), (2048, "no_root")]:
        P.test_env_order_layouts_bit_exact(n, dist)
    P.test_fused_step_act_only_and_split_apply()


def test_random_configurations_bit_exact(small_blocks):
    rng = np.random.default_rng(900 + small_blocks)
    for _ in range(10):
        n = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 255, 257, 500, 1000, 3000]))
        T.test_random_configuration_rollout_bit_exact(n, int(rng.integers(0, 6)), str(rng.choice(["pinball_simple", "pinball_maze"])),
                                                      int(rng.integers(0, 1 << 20)))


def test_gestation_trace_and_collect_bit_exact(small_blocks):
    L.test_trace_and_harvest_bit_exact_on_gpu()
    L.test_gestation_and_device_side_collect_bit_exact_on_gpu()


def test_more_workgroups_than_cus_bit_exact(small_blocks):
    S.test_more_than_256_workgroups_bit_exact()                                  # 70 000 envs, 5 options, learn on: 1094 / 547 workgroups


def test_contexts_of_different_geometry_live_side_by_side():
    """Two contexts of one process on different builds: same envs, same seeds — every integer output and the states agree bit for
    bit (the block size only orders the partial sums of G), and each reports its own geometry."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    ags = [SkillChainingAgent("pinball_simple", 2048, 0, seed=3, alpha=0.0, max_episode_steps=50, block_envs=b) for b in (64, 256)]
    assert [a.ctx.block_envs for a in ags] == [64, 256]
    for t in range(20):
        for a in ags:
            a.step_batch(learn=False)
    for k in ("x", "y", "vx", "vy", "action", "ep_steps", "reward", "done"):
        assert torch.equal(getattr(ags[0].state, k), getattr(ags[1].state, k)), k


def test_block_geometry_is_chosen_from_the_env_count(monkeypatch):
    """block_envs=None (and no SCG_BLOCK_ENVS): the smallest build whose workgroups fit the chip's 256 CUs in one round — a pure
    function of the env count, so two agents of equal size always agree; an explicit argument or the environment variable wins."""
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    monkeypatch.delenv("SCG_BLOCK_ENVS", raising=False)
    picks = {}
    for n in (1, 4096, 16384, 16385, 32768, 40000):
        ags = [SkillChainingAgent("pinball_simple", n, 1, seed=s) for s in (1, 2)]
        assert ags[0].ctx.block_envs == ags[1].ctx.block_envs
        picks[n] = ags[0].ctx.block_envs
        ags[0].step_batch()                                                       # (and the chosen build runs)
        del ags
    assert picks == {1: 64, 4096: 64, 16384: 64, 16385: 128, 32768: 128, 40000: 256}
    assert SkillChainingAgent("pinball_simple", 4096, 0, block_envs=256).ctx.block_envs == 256
    monkeypatch.setenv("SCG_BLOCK_ENVS", "128")
    assert SkillChainingAgent("pinball_simple", 4096, 0).ctx.block_envs == 128
