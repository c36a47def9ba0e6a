"""One-off randomised parity sweep beyond the fixed cases of tests/test_gpu_stress.py: N random configurations (batch size,
options, map incl. the crowded synthetic ones, masks, option graph, epsilon, seeds), 14-step fused rollouts with acting-only
steps in between, every output compared bit for bit with the CPU oracle.   Usage: [FUZZ_BLOCK=64|128] python tests/fuzz_parity.py [N] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]   # (lives under tests/: it uses the CPU oracle, the checker)
import numpy as np
import test_gpu_stress as T
from util import dense_map, hub_map
import gpu_util, skill_chaining_with_graphs_amd as scg

# FUZZ_BLOCK=64|128: the small-block builds of both sides
if os.environ.get("FUZZ_BLOCK"):
    gpu_util.set_block_envs(int(os.environ["FUZZ_BLOCK"]))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(seed0)
extra = {"dense160": dense_map(), "hub12": hub_map(12), "hub20": hub_map(20)}
real_load = scg.load_map
scg.load_map = lambda name: extra[name] if name in extra else real_load(name)      # make_pair() resolves names through the package
gpu_util.scg.load_map = scg.load_map
maps = ["pinball_simple", "pinball_maze", "pinball_empty"] + list(extra)
ok = 0
for c in range(n_cases):
    n = int(rng.choice([1, 7, 63, 64, 65, 128, 129, 191, 255, 256, 257, 500, 511, 513, 1000, 1500, 3000]))
    nopt = int(rng.integers(0, 6)); mp = str(rng.choice(maps)); seed = int(rng.integers(0, 1 << 20))
    try:
        T.test_random_configuration_rollout_bit_exact(n, nopt, mp, seed)
        ok += 1
    except AssertionError as e:
        print(f"MISMATCH case {c}: n={n} nopt={nopt} map={mp} seed={seed}: {str(e)[:200]}", flush=True)
print(f"{ok} of {n_cases} random configurations bit-exact (seed0 {seed0}"
      + (f", {os.environ['FUZZ_BLOCK']}-env blocks" if os.environ.get("FUZZ_BLOCK") else "") + ")")
sys.exit(0 if ok == n_cases else 1)
