#!/usr/bin/env python3
"""bench.py — env-steps/sec of the fused skill-chaining step-batch on N MI355X (BASELINE.json metric).

A "step" = one fused step-batch (scg_step: act + Pinball physics + option logic + Fourier features +
Q + TD + weight update) over this rank's env shard, inputs resident in HBM. Workload = BASELINE
configs[2]/[3]: 65 536 envs per GPU, full skill chain (5 options), independent env shards, no
collective ("scaling": "weak"); --shared-weights switches to configs[4] (RCCL all-reduce of dW).

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line. `roofline` prices the dominant kernel (td_kernel<FUSED>) from HIP events
recorded on the launch stream inside libscg_hip.so; `cpu_baseline` times the in-repo CPU oracle (kind
"port": the upstream reference ships no code to time) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

ENVS_PER_GPU = 65536
N_OPTIONS = 5
MAP = "pinball_simple"
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32-input MFMA peak (= the FP32 vector peak)
BYTES_PER_ENV_STEP = 46         # SURVEY.md §8(d) algorithmic HBM bytes per env-step
# (r_option_success = 100 is the hyper-parameter set of rounds 1-4, kept so that the headline workload stays what it was; the library's
#  default is 0 since round 5 — SPEC §4.2 — and the discovered-chain figure below runs with that)
HP = dict(gamma=0.99, alpha=1e-3, epsilon=0.05, r_option_success=100.0, max_episode_steps=2000,
          max_option_steps=250)


def chain_discs(pmap, n_options):
    """Synthetic skill chain standing in for discovered options: nested discs round the goal."""
    import numpy as np
    clf = np.zeros((n_options + 1, 8), np.float32)
    tx, ty, _ = pmap.target
    for k in range(1, n_options + 1):
        r = 0.18 + 0.17 * (k - 1)
        uc, vc, rr = 2 * tx - 1, 2 * ty - 1, 2 * r
        clf[k, :6] = [rr * rr - uc * uc - vc * vc, 2 * uc, 2 * vc, -1.0, 0.0, -1.0]
    return clf


def flops_per_item(evaluated: bool, updated: bool) -> float:
    """Algorithmic flops of one TD item: phi = 2 flop/feature (1 mul + 1 fma counted 3),
    Q(s',.) 5 fma/feature, Q(s,a) 1 fma/feature, accumulate 1 fma/feature."""
    f = 0.0
    if evaluated:
        f += 1296 * (3 + 10)
    if updated:
        f += 1296 * (3 + 2 + 2)
    return f


# BASELINE.json's metric, verbatim (the file travels with the repo; the literal is the fallback)
METRIC = "env-steps/sec (whole node), 65\u202f536 parallel Pinball envs at 1/2/4/8 MI355X"
try:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "BASELINE.json")) as _f:
        METRIC = json.load(_f).get("metric", METRIC)
except (OSError, ValueError):
    pass


TRAFFIC_FILES = ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json")      # newest first


def measured_traffic(envs_per_gpu, n_options):
    """(bytes, source): HBM bytes per launch of the dominant kernel from the newest committed PMC profile (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE, separate passes, the guide's corrections applied; profiles/rNN_traffic.json) — only
    when it was taken on this exact workload. A STATIC figure read from a tracked file: this run does not measure it."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            w = t["workload"]
            if w["envs_per_gpu"] == envs_per_gpu and w["n_options"] == n_options and w["map"] == MAP:
                return t["traffic_bytes_per_launch"], f"profiles/{name} (static: rocprofv3 PMC passes of an earlier run of this workload, not measured by this run)"
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def live_traffic(envs_per_gpu, n_options, timeout_s=150):
    """(bytes, source, raw) or None: HBM bytes per launch of the step kernel measured NOW — two child runs of this file under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes with --kernel-trace only, the program itself behind `--`),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE doubled, WRITE_SIZE exact; units of KB).
    Called before this process has touched the GPU (children of a GPU-initialised process are not allowed on the pool), never
    under a profiler, single-GPU runs only; any failure returns None and the line falls back to the tracked static figure."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or os.environ.get("HSA_TOOLS_LIB") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None                                  # this run is itself being profiled: its process has the GPU open already
    raw = {}
    try:
        tmp = tempfile.mkdtemp(prefix="scg_traffic_", dir="/tmp")
    except OSError:
        return None
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "30", "--warmup", "5", "--ramp", "20", "--envs-per-gpu", str(envs_per_gpu),
                   "--options", str(n_options), "--no-cpu-baseline", "--no-extras", "--no-live-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if "td_kernel<0>" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None
            raw[ctr] = sum(per.values()) / len(per)
            raw[ctr + "_dispatches"] = len(per)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError):
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return (int((2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024),
            "measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two child passes of this workload (30 timed step-batches each), "
            "per launch of td_kernel<0>; gfx950 correction per MI355X_MICROARCH.md: 2 x FETCH_SIZE + WRITE_SIZE, KB", raw)


def host_cpu_share():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota (a 1-GPU box of the pool shows
    256 CPUs in its mask and a quota of 16; 256 OpenMP threads inside that quota run 6x SLOWER than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                                   # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:                                                                        # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(seconds_target=15.0):
    """Time the CPU oracle on a bounded sample of the same workload (same map, options, hyper-params): first
    one thread (a third of the budget), then all host cores (SURVEY §8d asks for both)."""
    import numpy as np
    import sc_oracle
    import skill_chaining_with_graphs_amd as scg
    from skill_chaining_with_graphs_amd.core import fourier_scale_table
    cores = host_cpu_share()
    m = scg.load_map(MAP)
    clf = chain_discs(m, N_OPTIONS)

    def timed(threads, budget):
        n = 256 * threads                     # one 256-env block (SPEC §5) per thread
        orc = sc_oracle.Oracle(m, fourier_scale_table(), n_envs=n, n_options=N_OPTIONS, seed=0,
                               enabled_mask=sum(1 << k for k in range(1, N_OPTIONS + 1)), n_threads=threads, **HP)
        st = sc_oracle.new_state(n, m)
        rng = np.random.default_rng(0)
        pos = m.sample_free(n, rng)
        st["x"][:], st["y"][:] = pos[:, 0], pos[:, 1]
        v = rng.uniform(-1, 1, (2, n)).astype(np.float32)
        st["vx"][:], st["vy"][:] = v[0], v[1]
        W = (rng.standard_normal((N_OPTIONS + 1, 5, 1296)) * 1e-3).astype(np.float32)
        G, nk = orc.step(st, W, clf, 0)       # warm-up
        orc.apply(W, G, nk)
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < budget and steps < 100000:
            G, nk = orc.step(st, W, clf, steps + 1)
            orc.apply(W, G, nk)
            steps += 1
        dt = time.perf_counter() - t0
        return n * steps / dt, n, steps, dt

    v1, n1, s1, d1 = timed(1, seconds_target / 3.0)
    vc, nc, sc, dc = timed(cores, seconds_target * 2.0 / 3.0)
    return {"value": vc, "unit": "env-steps/s", "cores": cores, "kind": "port", "single_thread_value": v1,
            "sample": f"in-repo CPU oracle (oracle/sc_oracle.c, scalar fmaf chains, OpenMP over 256-env blocks; `cores` = "
                      f"the CPUs this process may use: affinity mask capped by the cgroup quota), same workload: "
                      f"{nc} envs x {sc} step-batches on {cores} threads in {dc:.1f} s; {n1} envs x {s1} step-batches "
                      f"on 1 thread in {d1:.1f} s; the upstream reference ships no code to time"}


def _time_steps(agent, steps, warmup, ramp=200):
    import torch
    for _ in range(ramp + warmup):
        agent.step_batch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        agent.step_batch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def extra_measurements(steps, warmup):
    """Two more single-GPU figures beside the headline (which keeps its synthetic nested-disc options so that rounds stay
    comparable): BASELINE configs[1] (4096 envs, root + 1 chained option) and configs[2] with DISCOVERED options — the outer
    loop (SkillChainingAgent.chain_skills: trajectory ring -> device-side trigger -> GPU logistic regression) runs untimed
    on the 65 536-env agent until <= 5 options exist, then the steady state is timed with those classifiers."""
    import torch
    from skill_chaining_with_graphs_amd import SkillChainingAgent
    ex = {}
    # configs[1] on the block geometry a 4096-env context picks by itself (round 5: the smallest build that fits the chip in one
    # round of workgroups — 64-env blocks, 64 workgroups)
    n1 = 4096
    ag = SkillChainingAgent(MAP, n1, 1, seed=0, **HP)
    ag.clf.copy_(torch.as_tensor(chain_discs(ag.map, 1)))
    ag.enable_option(1)
    ag.init_weights(std=1e-3, seed=0)
    ag.domain.reset_random(seed=1000, v_max=1.0)
    dt = _time_steps(ag, max(steps, 200), warmup)
    ex["config1_4096_envs_root_plus_1_option"] = {"value": n1 / dt, "unit": "env-steps/s", "us_per_step": dt * 1e6, "block_envs": ag.ctx.block_envs,
                                                  "block_envs_chosen": "automatically from the env count", "workgroups": -(-n1 // ag.ctx.block_envs),
                                                  "envs_in_an_option_at_end": int((ag.state.option_id > 0).sum())}
    del ag
    # configs[2] on discovered options
    hp = dict(HP, alpha=0.02, r_option_success=0.0, update_count_floor=ENVS_PER_GPU // 16)       # a learning rate at which the root reaches the goal within the untimed
    ag = SkillChainingAgent(MAP, ENVS_PER_GPU, N_OPTIONS, seed=0, **hp)     # warm-up; small value functions take steps as if they had n_envs / 16 items (SPEC §5 apply: DESIGN §9)
    ag.enable_tracing(64)
    warm = 3000
    for _ in range(warm):
        ag.step_batch()
    t_before = ag.t
    report = ag.chain_skills(steps_per_option=400, min_examples=3000, max_examples=60000, start_coverage=2.0)   # never "covered": up to 5 options
    discovery_steps = ag.t - t_before
    ag.ctx.disarm_collect()
    dt = _time_steps(ag, max(steps, 200), warmup)
    per_opt = [int((ag.state.option_id == k).sum()) for k in range(N_OPTIONS + 1)]
    ex["discovered_chain_65536_envs"] = {"value": ENVS_PER_GPU / dt, "unit": "env-steps/s", "us_per_step": dt * 1e6,
                                          "options_created": len(report), "untimed_root_warmup_steps": warm,
                                          "untimed_discovery_steps": discovery_steps,
                                          "fit_accuracy": [round(float(r["accuracy"]), 3) for r in report],
                                          "envs_per_running_option_at_end": per_opt,
                                          "envs_in_an_option_at_end": int(sum(per_opt[1:])), "tracing": "on (trajectory ring + events)",
                                          "hparams": hp}
    del ag
    torch.cuda.empty_cache()
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--options", type=int, default=N_OPTIONS)
    ap.add_argument("--shared-weights", action="store_true", help="configs[4]: all-reduce dW over RCCL each step")
    ap.add_argument("--ordered-sum", action="store_true", help="with --shared-weights: one all-gather + the sum in rank order "
                    "inside the apply launch (scg_apply_update_slots: identical weights on every rank of the run, reproducible by the oracle for any number of ranks) instead of the all-reduce")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = CPU-side rehearsal of the N>1 path on a 1-GPU box (every rank computes on cuda:0)")
    ap.add_argument("--block-envs", type=int, default=None, choices=[64, 128, 256],
                    help="SPEC §5 block size = library build (default 256, the throughput build; 64 / 128: the small-batch builds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-learn", action="store_true", help="diagnostic: act + physics + qcache only")
    ap.add_argument("--diag-no-td", action="store_true", help="diagnostic: physics + option logic only")
    ap.add_argument("--diag-fresh-sort", action="store_true", help="diagnostic: stand-alone sort kernels every step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the two extra single-GPU measurements printed beside the headline "
                    "(BASELINE configs[1]: 4096 envs, root + 1 option; configs[2] on DISCOVERED options: chain_skills run untimed first)")
    ap.add_argument("--event-every", type=int, default=0, help="HIP event pair round every n-th fused-kernel launch "
                    "(0 = max(8, steps // 8): each pair costs a few us of queue bubble — 3 us per step when every other launch is sampled)")
    ap.add_argument("--ramp", type=int, default=200, help="untimed clock-ramp step-batches before the warm-up "
                    "(a 20-step run otherwise times a cold GPU and the first step's stand-alone sort)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the tracked profile instead of two rocprofv3 counter "
                    "passes of this workload run first (single-GPU runs; skipped anyway with --no-extras, under a profiler, or when rocprofv3 is missing)")
    args = ap.parse_args()

    live = None
    if ("WORLD_SIZE" not in os.environ and args.gpus == 1 and not args.no_live_traffic and not args.no_extras
            and not (args.no_learn or args.diag_no_td or args.diag_fresh_sort or args.shared_weights)):
        live = live_traffic(args.envs_per_gpu, args.options)      # BEFORE anything here touches the GPU (no torch import yet)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: become the launcher. Nothing in this process has touched the GPU (no torch
        # import yet): start N fresh rank processes through torch.distributed.run and relay rank 0's JSON line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if lines:
            print(lines[-1], flush=True)
        sys.exit(r.returncode if r.returncode else (0 if lines else 1))

    import numpy as np
    import torch
    import torch.distributed as dist
    import skill_chaining_with_graphs_amd as scg
    from skill_chaining_with_graphs_amd import SkillChainingAgent

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    args.gpus = world
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0                      # all ranks share the one card; gloo carries the (tiny) collectives
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    n_local = args.envs_per_gpu
    n_opt = args.options
    lo = rank * n_local
    group = dist.group.WORLD if (distributed and args.shared_weights) else None
    agent = SkillChainingAgent(MAP, n_local, n_opt, device=local_rank, seed=0, env_id_base=lo, group=group,
                               block_envs=args.block_envs, ordered_sum=args.ordered_sum, **HP)
    agent.clf.copy_(torch.as_tensor(chain_discs(agent.map, n_opt)))
    for k in range(1, n_opt + 1):
        agent.enable_option(k)
    agent.init_weights(std=1e-3, seed=0)
    agent.domain.reset_random(seed=1000 + rank, v_max=1.0)
    lib, ctx = agent.ctx.lib, agent.ctx._ctx

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    learn = not args.no_learn
    if args.diag_no_td or args.diag_fresh_sort:
        import ctypes as _C
        from skill_chaining_with_graphs_amd.core import _ptr
        def _step(_learn=True):
            st = agent.state
            agent.ctx._call("scg_step", _ptr(st.x), _ptr(st.y), _ptr(st.vx), _ptr(st.vy), _ptr(st.option_id),
                            _ptr(st.opt_steps), _ptr(st.ep_steps), _ptr(st.qcache), _ptr(st.action),
                            _ptr(st.reward), _ptr(st.done), _ptr(agent.W), _ptr(agent.clf),
                            _C.c_uint32(agent.enabled_mask), _C.c_uint64(agent.t), _C.c_uint32(0x203 if args.diag_fresh_sort else 0x100),
                            agent.ctx._stream())
            agent.t += 1
        agent.step_batch = _step
    for _ in range(args.ramp):                    # untimed, not counted as warm-up: clocks up, env order prepared
        agent.step_batch(learn)
    for _ in range(args.warmup):
        agent.step_batch(learn)
    barrier()
    event_every = args.event_every if args.event_every > 0 else max(8, args.steps // 8)
    lib.scg_profile_reset(ctx, event_every)       # HIP events round the fused kernel, on the launch stream
    if group is not None:
        agent.time_allreduce(event_every)         # ... and round the shared-weights all-reduce, as the step's stream sees it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agent.step_batch(learn)
    barrier()
    dt = time.perf_counter() - t0
    import ctypes as C
    k_ms, k_n = C.c_double(), C.c_int64()
    lib.scg_profile_read(ctx, C.byref(k_ms), C.byref(k_n))
    lib.scg_profile_reset(ctx, 0)

    ar = agent.time_allreduce(0) if group is not None else None
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dt_ranks = [dt]
    world_seen = 1
    if distributed:
        world_seen = dist.get_world_size()        # what the process group really holds (not the --gpus flag)
        every = [torch.zeros_like(t) for _ in range(world_seen)]
        dist.all_gather(every, t)
        dt_ranks = [float(v.item()) for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    total_env_steps = float(n_local) * world * args.steps
    value = total_env_steps / dt_max

    out = None
    if rank == 0:
        kern_ms = k_ms.value / max(k_n.value, 1)
        units = n_local
        achieved = units * BYTES_PER_ENV_STEP / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic, traffic_source = measured_traffic(n_local, n_opt)
        traffic_raw = None
        if live is not None:
            traffic, traffic_source, traffic_raw = live
        out = {
            "metric": METRIC,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n_local} envs/GPU x {world} GPU, map {MAP}, Fourier order 5 (1296 terms), "
                                   f"root + {n_opt} chained options (synthetic nested-disc initiation sets; SPEC 4.2 value-gated entry: see mfma.envs_in_an_option_at_end), "
                                   f"{('shared option-Q weights, ' + ('all-gather of dW + sum in rank order' if args.ordered_sum else 'RCCL all-reduce of dW')) if group is not None else 'independent env shards, no collective'}",
                       "envs_per_gpu": n_local, "n_options": n_opt, "map": MAP, "hparams": HP,
                       "block_envs": agent.ctx.block_envs,
                       "untimed_ramp_steps": args.ramp,      # step-batches run BEFORE the warm-up: clocks up, env order prepared;
                                                              # `value` is therefore a steady-state figure
                       "backend": ("none" if not distributed else args.backend)},
            "ranks": {"world_size_seen": world_seen, "ms_per_step_min": min(dt_ranks) / args.steps * 1e3,
                      "ms_per_step_max": max(dt_ranks) / args.steps * 1e3,
                      "allreduce": (None if ar is None else dict(ar, bytes=int(agent.ctx.grad_packed().numel()) * 4,
                                    note="event pair on the step's stream round rank 0's collective on the packed operand (the all-reduce, or with "
                                         "--ordered-sum the all-gather): the collective as the step sees it (fully exposed: the next "
                                         "launch needs its result)"))},
            "roofline": {"bound": "hbm", "kernel": "td_kernel<MODE_FUSED>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_counters_kb": traffic_raw, "kernel_ms": kern_ms, "launches": int(k_n.value),
                         "algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                         "note": "the fused kernel is compute- and latency-bound (f32 matrix pipe 50 % busy), not HBM-bound (SURVEY.md \u00a78d, "
                                 "DESIGN.md): see `mfma` for the binding roofline"},
        }
        # the binding (matrix-pipe) roofline: algorithmic flops of the TD items this rank processed per step
        st = agent.state
        opt = st.option_id.cpu().numpy()
        n_opt_items = int((opt > 0).sum())
        flops_step = units * flops_per_item(True, True) + n_opt_items * flops_per_item(True, True)
        out["mfma"] = {"bound": "mfma", "dtype": "f32", "achieved": flops_step / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                       "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": (flops_step / (kern_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if kern_ms > 0 else 0.0,
                       "algorithmic_flop_per_step_batch": flops_step,
                       "envs_in_an_option_at_end": n_opt_items,
                       "envs_inside_an_initiation_set_but_staying_out_at_end": int((opt < 0).sum()),      # SPEC 4.2 (round 5): option_id = -k
                       "note": "algorithmic flops of the direct formulation (phi, Q, accumulate: DESIGN.md) over the "
                               "dense f32 MFMA peak (= the f32 vector peak, MI355X_MICROARCH.md); the kernel runs the "
                               "contractions in factorised form on v_mfma_f32_16x16x4_f32 and issues ~1.9x these flops"}
        if world == 1 and not args.no_extras and not (args.diag_no_td or args.diag_fresh_sort or args.no_learn):
            del agent                                   # free the headline agent's buffers first
            torch.cuda.empty_cache()
            try:                                        # the extras never cost the headline its line
                out["extras"] = extra_measurements(args.steps, args.warmup)
            except Exception as e:                      # noqa: BLE001 — reported in the line, loudly, not swallowed
                out["extras"] = {"error": f"{type(e).__name__}: {e}"}
                print(f"bench.py: the extra measurements failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
