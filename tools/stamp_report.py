"""Diagnostic: per-section cycle shares of the fused kernel from the SCG_STAMPS build (wave 0 of each block).
Usage: python tools/stamp_report.py [--options N] [--steps K]   (needs `make -C .../csrc stamps`)."""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
ap = argparse.ArgumentParser(); ap.add_argument("--options", type=int, default=5); ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
from skill_chaining_with_graphs_amd import _lib
_lib.LIB_PATH = os.environ.get("SCG_STAMPS_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libscg_hip_stamps.so")   # SCG_STAMPS_LIB: a variant stamps build
import numpy as np, torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n = 65536
agent = SkillChainingAgent(bench.MAP, n, args.options, seed=0, **bench.HP)
agent.clf.copy_(torch.as_tensor(bench.chain_discs(agent.map, args.options)))
for k in range(1, args.options + 1): agent.enable_option(k)
agent.init_weights(std=1e-3); agent.domain.reset_random(seed=1000)
for _ in range(10): agent.step_batch()
torch.cuda.synchronize()
lib, ctx = agent.ctx.lib, agent.ctx._ctx
lib.scg_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
nblk = n // lib.scg_block_envs()
lib.scg_diag_stamps(ctx, None, 1)
lib.scg_profile_reset(ctx, 4)                      # HIP event pairs round every 4th fused-kernel launch: the same launches' wall time
for _ in range(args.steps): agent.step_batch()
torch.cuda.synchronize()
ms, cnt = C.c_double(0.0), C.c_int64(0)
lib.scg_profile_read(ctx, C.byref(ms), C.byref(cnt))
kernel_us = 1e3 * ms.value / max(cnt.value, 1)
out = np.zeros((nblk, 48), np.uint64)
lib.scg_diag_stamps(ctx, out.ctypes.data_as(C.c_void_p), 0)
names = ["phase P (rest: trace, hist, barrier)", "pass 0: phase Z + flags + (W staging)", "P: physics, pooled (env, edge) pair groups + hand-offs", "pass 0: E (eval, both VFs)", "pass 0: U1 (only when no helper ran it)",
         "pass 0: Z + lists until the pass stamp", "pass 0: U2 (last chunk: wait)", "pass 0: wait before U2", "eval-only: staging (own share)", "U2: own products of a chunk",
         "(helper: W_0 staged)", "(helper: + wait for the published states)", "(helper: + option, W_B, Z(s))", "(helper: + wait for the actions)", "(helper: + lists, hand-off)", "slab stores",
         "P: qcache gathers, Philox, action", "P: perm + state gathers", "P: physics, own part (refine, free flight, pair lists)", "P: bookkeeping, options, result line",
         "U2: prologue / wait for the previous chunk's products", "U2: build", "U2: wait for the other waves' build", "flags + ballots", "W staging (when not under P)", "barrier behind the staging", "eval-only: wait for region R", "eval-only: staging + units on the matrix pipe", "(helper wave 8: kernel start -> its U1 done; not part of the total)", "(count: eval-only pairs wave 0 took)", "(count: eval-only pairs in the block)", "eval-only: barrier behind the staging"]
mean = out.astype(np.float64).mean(0) / args.steps
tot = mean[:32].sum() - mean[28] - mean[29] - mean[30] - mean[10:15].sum()
print(f"wave-0 cycles per launch (mean over {nblk} blocks), total {tot:.0f} cycles (s_memtime ticks = shader cycles... 100MHz? see below)")
for nme, v in zip(names, mean): print(f"  {nme:36s} {v:10.0f}  {100*v/tot:5.1f} %")
print(f"fused kernel by HIP events (this build, {cnt.value} launches): {kernel_us:.2f} us -> wave 0's {tot:.0f} cycles in that time = {tot / kernel_us / 1e3:.3f} GHz "
      f"(lower bound of the shader clock: the workgroup starts a little after the launch and ends a little before it does)")
print("E phase per wave (cycles, mean over blocks):", np.round(mean[32:48]).astype(int).tolist())
per_block = out[:, :32].astype(np.float64).sum(1) / args.steps
print(f"per-block wave-0 total ticks per launch: min {per_block.min():.0f}  mean {per_block.mean():.0f}  max {per_block.max():.0f}"
      f"  (mean/max = {per_block.mean()/per_block.max():.2f}: one workgroup per CU, the launch lasts as long as its slowest)")
print("  by block index (env order, 16 bins):", np.round(per_block.reshape(16, -1).mean(1)).astype(int).tolist())
