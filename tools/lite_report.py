"""Phase boundaries of the fused kernel from the LIGHT timing build (`make -C .../csrc lite`; SCG_LITE_LIB: a variant build): four
waves (env wave 0, the first pool wave, the first and the last helper wave) x eight boundaries, cycles since the wave's kernel entry,
mean over blocks and launches — and the same launches' wall time by HIP events, so that the build's distance from the product
(whose time bench.py / tools/ab_bench.py give) is on the table.   python tools/lite_report.py [--steps K]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
ap = argparse.ArgumentParser(); ap.add_argument("--options", type=int, default=5); ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--u2", action="store_true", help="the library is the SCG_STAMPS_LITE=3 variant (make lite_u2): waves 12, 13, 0, 3 stamp inside U2 of pass 0")
args = ap.parse_args()
from skill_chaining_with_graphs_amd import _lib
_lib.LIB_PATH = os.environ.get("SCG_LITE_LIB") or os.path.join(os.path.dirname(_lib.LIB_PATH), "libscg_hip_lite.so")
import numpy as np, torch
import bench
from skill_chaining_with_graphs_amd import SkillChainingAgent
n = 65536
agent = SkillChainingAgent(bench.MAP, n, args.options, seed=0, **bench.HP)
agent.clf.copy_(torch.as_tensor(bench.chain_discs(agent.map, args.options)))
for k in range(1, args.options + 1): agent.enable_option(k)
agent.init_weights(std=1e-3); agent.domain.reset_random(seed=1000)
for _ in range(250): agent.step_batch()
torch.cuda.synchronize()
lib, ctx = agent.ctx.lib, agent.ctx._ctx
lib.scg_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
nblk = n // lib.scg_block_envs()
lib.scg_diag_stamps(ctx, None, 1)
lib.scg_profile_reset(ctx, 4)
for _ in range(args.steps): agent.step_batch()
torch.cuda.synchronize()
ms, cnt = C.c_double(0.0), C.c_int64(0)
lib.scg_profile_read(ctx, C.byref(ms), C.byref(cnt))
out = np.zeros((nblk, 48), np.uint64)
lib.scg_diag_stamps(ctx, out.ctypes.data_as(C.c_void_p), 0)
m = out[:, :32].astype(np.float64).mean(0).reshape(4, 8) / args.steps
if args.u2:
    names = ["whole kernel", "U2 starts (E barrier passed)", "chunk 0 built, barrier passed", "chunk 0: own products done", "chunk 0 consumed (barrier)", "chunk 1 built, barrier passed",
             "chunk 1: own products done", "pass 0 done"]
    print(f"{os.path.basename(_lib.LIB_PATH)}: fused kernel by HIP events {1e3 * ms.value / max(cnt.value, 1):.2f} us ({cnt.value} launches); cycles since kernel entry, mean over {nblk} blocks")
    print(f"{'':32s} {'wave 12':>12s} {'wave 13':>12s} {'wave 0':>12s} {'wave 3':>12s}   (12, 13: the no-op's; blocks with one chunk leave 4-6 at 0: means are over all blocks)")
    for i in [1, 2, 3, 4, 5, 6, 7, 0]:
        print(f"{names[i]:32s} " + " ".join(f"{m[r, i]:12.0f}" for r in range(4)))
    two = out[:, 6] > 0
    mm = out[two][:, :32].astype(np.float64).mean(0).reshape(4, 8) / args.steps
    print(f"blocks whose LAST launch had two chunks: {int(two.sum())} of {nblk}; own products of chunk 0 / chunk 1, cycles: " +
          " | ".join(f"{mm[r, 3] - mm[r, 2]:.0f} / {mm[r, 6] - mm[r, 5]:.0f}" for r in range(4)))
    sys.exit(0)
names = ["whole kernel", "start barrier passed", "own phase-P work done", "phase-P barrier passed", "E starts", "own E done", "E barrier passed (U2 starts)", "pass 0 done"]
print(f"{os.path.basename(_lib.LIB_PATH)}: fused kernel by HIP events {1e3 * ms.value / max(cnt.value, 1):.2f} us ({cnt.value} launches); cycles since kernel entry, mean over {nblk} blocks")
print(f"{'':32s} {'env wave 0':>12s} {'pool wave':>12s} {'helper 0':>12s} {'last helper':>12s}")
for i in [1, 2, 3, 4, 5, 6, 7, 0]:
    print(f"{names[i]:32s} " + " ".join(f"{m[r, i]:12.0f}" for r in range(4)))
r0, r1 = out[:, 32].astype(np.int64), out[:, 33].astype(np.int64)
print(f"last launch on the 100 MHz clock: workgroups start within {(r0.max() - r0.min()) * 10} ns of one another, end within {(r1.max() - r1.min()) * 10} ns; "
      f"first start -> last end {(r1.max() - r0.min()) / 100:.2f} us; mean workgroup {np.mean(r1 - r0) / 100:.2f} us")
# per-block spread: the launch lasts as long as its slowest workgroup
dur = (r1 - r0) / 100.0
order = np.argsort(dur)
ph = out[:, :8].astype(np.float64) / args.steps          # env wave 0's boundaries of every block (mean over launches)
print("workgroup wall time (last launch), us: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max()))
print("by XCD (block id % 8), mean us:", " ".join(f"{dur[x::8].mean():.2f}" for x in range(8)))
print("by block index (16 bins), mean us:", " ".join(f"{v:.2f}" for v in dur.reshape(16, -1).mean(1)))
seg = np.stack([ph[:, 3], ph[:, 4] - ph[:, 3], ph[:, 6] - ph[:, 4], ph[:, 7] - ph[:, 6], ph[:, 0] - ph[:, 7]], 1)
lab = ["head (-> P barrier)", "Z + lists", "E (-> barrier)", "U2", "tail"]
print("env wave 0's segments, cycles (mean over launches): mean over blocks / mean of the 16 slowest blocks / of the 16 fastest")
for j, l in enumerate(lab):
    print(f"  {l:22s} {seg[:, j].mean():9.0f} {seg[order[-16:], j].mean():9.0f} {seg[order[:16], j].mean():9.0f}")
print("slowest blocks:", order[-12:].tolist(), " fastest:", order[:12].tolist())
