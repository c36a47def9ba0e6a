"""ctypes wrapper of oracle/libsc_oracle.so — the scalar CPU restatement of SPEC.md.

TEST INFRASTRUCTURE ONLY (PARITY UNPINNED — see sc_oracle.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCO_LIB") or os.path.join(_HERE, "libsc_oracle.so")   # SCO_LIB: the sanitizer build
NACT, NF, CLF_STRIDE, BLOCK_ENVS = 5, 1296, 8, 256


class Params(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_options", C.c_int32), ("env_id_base", C.c_int64), ("seed", C.c_uint64),
        ("gamma", C.c_float), ("alpha", C.c_float), ("epsilon", C.c_float), ("r_option_success", C.c_float),
        ("max_episode_steps", C.c_int32), ("max_option_steps", C.c_int32),
        ("enabled_mask", C.c_uint32), ("n_threads", C.c_int32),
        ("n_edges", C.c_int32), ("n_starts", C.c_int32),
        ("edges", C.c_void_p), ("starts", C.c_void_p),
        ("radius", C.c_float), ("hstep", C.c_float), ("r2", C.c_float),
        ("tx", C.c_float), ("ty", C.c_float), ("tr2", C.c_float),
        ("scale", C.c_void_p),
        ("ring_x", C.c_void_p), ("ring_y", C.c_void_p), ("events", C.c_void_p), ("ev_len", C.c_void_p),
        ("ring_len", C.c_int32),
        ("parents", C.c_int32 * 8),
        ("gest_mask", C.c_uint32), ("gest_succ", C.c_void_p),
        ("exit_rule", C.c_int32), ("select_rule", C.c_int32), ("nk_floor", C.c_int32), ("reoffer_period", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sc_oracle.c")
    if os.environ.get("SCO_LIB"):
        return LIB_PATH
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB_PATH


_lib = None
_libs = {}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.sco_sigmoid.restype = C.c_float
        _lib.sco_sigmoid.argtypes = [C.c_float]
        _lib.sco_q_update_grad.restype = C.c_int
        _libs[LIB_PATH] = _lib
    return _lib


def use_block_envs(block_envs: int = 256) -> None:
    """Point the checker at its build for another SPEC §5 block size (64 / 128 / 256: the HIP library is built for the same
    three, and each is checked against the oracle of its own geometry)."""
    global _lib, BLOCK_ENVS
    if os.environ.get("SCO_LIB") and block_envs == 256:
        path = LIB_PATH
    else:
        path = os.path.join(_HERE, "libsc_oracle.so" if block_envs == 256 else f"libsc_oracle_b{block_envs}.so")
    if path not in _libs:
        build()
        if not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE, "-s", os.path.basename(path)], check=True, capture_output=True)
        l = C.CDLL(path)
        l.sco_sigmoid.restype = C.c_float
        l.sco_sigmoid.argtypes = [C.c_float]
        l.sco_q_update_grad.restype = C.c_int
        assert l.sco_block_envs() == block_envs, (path, l.sco_block_envs())
        _libs[path] = l
    _lib = _libs[path]
    BLOCK_ENVS = block_envs


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Oracle:
    """Holds the map tables + hyper-parameters; methods mirror the C-ABI entry points on numpy arrays."""

    def __init__(self, pmap, scale, n_envs=1, n_options=0, seed=0, env_id_base=0, gamma=0.99, alpha=1e-3,
                 epsilon=0.05, r_option_success=0.0, max_episode_steps=10000, max_option_steps=250,
                 enabled_mask=0, n_threads=1, update_count_floor=0, reoffer_period=4):
        self.L = lib()
        self.edges = _f32(pmap.edges)
        self.starts = _f32(pmap.starts)
        self.scale = _f32(scale)
        sc = pmap.scalars
        self.p = Params(n_envs=n_envs, n_options=n_options, env_id_base=env_id_base, seed=seed, gamma=gamma,
                        alpha=alpha, epsilon=epsilon, r_option_success=r_option_success,
                        max_episode_steps=max_episode_steps, max_option_steps=max_option_steps,
                        enabled_mask=enabled_mask, n_threads=n_threads,
                        n_edges=len(self.edges), n_starts=len(self.starts),
                        edges=self.edges.ctypes.data, starts=self.starts.ctypes.data,
                        radius=sc[0], hstep=sc[1], r2=sc[2], tx=sc[3], ty=sc[4], tr2=sc[5],
                        scale=self.scale.ctypes.data)
        self.p.nk_floor = update_count_floor
        self.p.reoffer_period = reoffer_period
        self.p.exit_rule, self.p.select_rule = 2, 1            # SPEC §4.2 (other values: experiments of tests/oracle_chain_evidence.py)
        self.n_vf = n_options + 1
        self.set_parents(list(range(-1, 7)))            # default chain: 1 -> goal, k -> k-1

    # ---- primitives
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
        self.L.sco_philox4x32_10(c, k, o)
        return list(o)

    def sincospi(self, t):
        c, s = C.c_float(), C.c_float()
        self.L.sco_sincospi(C.c_float(t), C.byref(c), C.byref(s))
        return c.value, s.value

    def sigmoid(self, z):
        return float(self.L.sco_sigmoid(C.c_float(z)))

    # ---- un-fused entry points (arrays are modified in place like the C-ABI)
    def pinball_step(self, x, y, vx, vy, action):
        n = len(x)
        reward = np.empty(n, np.float32); goal = np.empty(n, np.uint8)
        self.L.sco_pinball_step(C.byref(self.p), n, _p(x), _p(y), _p(vx), _p(vy), _p(action), _p(reward), _p(goal))
        return reward, goal

    def features(self, x, y, vx, vy):
        n = len(x)
        phi = np.empty((n, NF), np.float32)
        self.L.sco_features(n, _p(x), _p(y), _p(vx), _p(vy), _p(phi))
        return phi

    def q_values(self, x, y, vx, vy, Wk):
        n = len(x)
        q = np.empty((NACT, n), np.float32)
        self.L.sco_q_values(n, _p(x), _p(y), _p(vx), _p(vy), _p(_f32(Wk)), _p(q))
        return q

    def classifier_predict(self, x, y, w8):
        out = np.empty(len(x), np.uint8)
        self.L.sco_classifier_predict(len(x), _p(x), _p(y), _p(_f32(w8)), _p(out))
        return out

    def q_update_grad(self, s, action, r, cont, sn, Wk):
        n = len(action)
        s = [_f32(a) for a in s]; sn = [_f32(a) for a in sn]
        S = (C.c_void_p * 4)(*[a.ctypes.data for a in s])
        SN = (C.c_void_p * 4)(*[a.ctypes.data for a in sn])
        G = np.zeros((NACT, NF), np.float32)
        cnt = self.L.sco_q_update_grad(C.byref(self.p), n, S, _p(action), _p(_f32(r)), _p(_f32(cont)), SN,
                                       _p(_f32(Wk)), _p(G))
        return G, int(cnt)

    def apply(self, W, G, n_k):
        n_k = np.ascontiguousarray(n_k, np.int32)
        self.L.sco_apply(C.byref(self.p), len(n_k), _p(W), _p(_f32(G)), _p(n_k))

    def step(self, st, W, clf, t, enabled_mask=None):
        """st: dict of numpy arrays (x,y,vx,vy,option_id,opt_steps,ep_steps,qcache[5,N],action,reward,done),
        modified in place. Returns (G[n_vf,5,1296], n_k[n_vf]); W is not modified (call apply)."""
        if enabled_mask is not None:
            self.p.enabled_mask = enabled_mask
        G = np.zeros((self.n_vf, NACT, NF), np.float32)
        n_k = np.zeros(self.n_vf, np.int32)
        self.L.sco_step(C.byref(self.p), _p(st["x"]), _p(st["y"]), _p(st["vx"]), _p(st["vy"]),
                        _p(st["option_id"]), _p(st["opt_steps"]), _p(st["ep_steps"]), _p(st["qcache"]),
                        _p(st["action"]), _p(st["reward"]), _p(st["done"]), _p(W), _p(_f32(clf)),
                        C.c_uint64(t), _p(G), _p(n_k))
        return G, n_k

    def set_parents(self, parents):
        """SPEC §4.2 option graph: parents[k] = target option of k (0 = goal)."""
        for k in range(8):
            self.p.parents[k] = max(int(parents[k]), 0) if k < len(parents) else 0

    def set_gestation(self, gest_mask):
        """SPEC §4.4: options in gestation + their success counters (self.gest_succ, int32[n_vf])."""
        self.gest_succ = np.zeros(self.n_vf, np.int32)
        self.p.gest_mask = gest_mask
        self.p.gest_succ = self.gest_succ.ctypes.data

    def collect_examples(self, bits, prev_in, l_pos, l_neg, ex_xy, ex_label, count):
        """SPEC §7 device-side trigger, mirrored: appends to ex_xy[cap,2] / ex_label[cap]; count = np.int32[1] in/out."""
        self.L.sco_collect_examples(self.p.n_envs, _p(self.events), _p(prev_in), C.c_uint32(bits), _p(self.ring_x),
                                    _p(self.ring_y), self.p.ring_len, _p(self.ev_len), l_pos, l_neg, _p(ex_xy),
                                    _p(ex_label), _p(count), len(ex_label))

    def set_trace(self, ring_len):
        """SPEC §7: allocate + attach trace buffers (ring_x, ring_y [ring_len, N], events [N], ev_len [N])."""
        n = self.p.n_envs
        self.ring_x = np.zeros((ring_len, n), np.float32); self.ring_y = np.zeros((ring_len, n), np.float32)
        self.events = np.zeros(n, np.uint8); self.ev_len = np.zeros(n, np.int32)
        self.p.ring_x = self.ring_x.ctypes.data; self.p.ring_y = self.ring_y.ctypes.data
        self.p.events = self.events.ctypes.data; self.p.ev_len = self.ev_len.ctypes.data
        self.p.ring_len = ring_len

    def harvest(self, sel_env, l_pos, l_neg):
        sel = np.ascontiguousarray(sel_env, np.int32)
        L = l_pos + l_neg
        xy = np.zeros((len(sel), L, 2), np.float32); lab = np.zeros((len(sel), L), np.uint8)
        self.L.sco_harvest(len(sel), _p(sel), _p(self.ring_x), _p(self.ring_y), self.p.ring_len, self.p.n_envs,
                           _p(self.ev_len), l_pos, l_neg, _p(xy), _p(lab))
        return xy, lab

    def fit_initiation(self, xy, label, offsets, w, iters, lr, l2):
        offsets = np.ascontiguousarray(offsets, np.int32)
        self.L.sco_fit_initiation(len(offsets) - 1, _p(_f32(xy)), _p(np.ascontiguousarray(label, np.uint8)),
                                  _p(offsets), _p(w), int(iters), C.c_float(lr), C.c_float(l2))


def new_state(n, pmap):
    """Host mirror of core.EnvState."""
    sx, sy = pmap.starts[0]
    return dict(
        x=np.full(n, sx, np.float32), y=np.full(n, sy, np.float32),
        vx=np.zeros(n, np.float32), vy=np.zeros(n, np.float32),
        option_id=np.zeros(n, np.int32), opt_steps=np.zeros(n, np.int32), ep_steps=np.zeros(n, np.int32),
        qcache=np.zeros((NACT, n), np.float32), action=np.zeros(n, np.uint8),
        reward=np.zeros(n, np.float32), done=np.zeros(n, np.uint8))
