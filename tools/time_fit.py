"""Times the second north_star kernel, scg_fit_initiation (SPEC §6 batched logistic regression), at the sizes the outer
loop uses (SkillChainingAgent.chain_skills: 2k-40k examples, 400 iterations, 1-5 options) and prints one JSON line per
case: milliseconds per fit, example-iterations per second, and the HBM view BASELINE.json asks for (algorithmic bytes =
every example read ONCE: 9 B, they stay in registers for all iterations; achieved GB/s against the 8 TB/s peak —
the kernel is bound by the per-iteration dependency chain, not by HBM)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
import skill_chaining_with_graphs_amd as scg
from skill_chaining_with_graphs_amd.core import ScgContext

ITERS = 400
ctx = ScgContext(1024, 5, scg.load_map("pinball_simple"))
rng = np.random.default_rng(0)
for n_fit, m in ((1, 2000), (1, 10000), (1, 40000), (5, 2000), (5, 40000)):
    xy = torch.as_tensor(rng.random((n_fit * m, 2)).astype(np.float32), device="cuda:0")
    c = (xy[:, 0] - 0.6) ** 2 + (xy[:, 1] - 0.4) ** 2 < 0.09
    lab = c.to(torch.uint8).contiguous()
    off = torch.arange(0, n_fit + 1, dtype=torch.int32, device="cuda:0") * m
    w = torch.zeros((n_fit, 8), dtype=torch.float32, device="cuda:0")

    def run():
        w.zero_()
        ctx.fit_initiation(xy.view(-1), lab, off, w.view(-1), iters=ITERS, lr=3.0, l2=1e-4)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    acc = float(((ctx.classifier_predict(xy[:m, 0].contiguous(), xy[:m, 1].contiguous(), w[0].contiguous()) == lab[:m]).float().mean()))
    algo = n_fit * m * 9
    print(json.dumps({"kernel": "fit_kernel", "options": n_fit, "examples_per_option": m, "iters": ITERS, "ms_per_fit": ms,
                      "example_iterations_per_s": n_fit * m * ITERS / (ms * 1e-3), "algorithmic_bytes": algo,
                      "achieved_GBps": algo / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0, "train_accuracy": acc}))
