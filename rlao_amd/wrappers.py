"""The thin wrappers drl4ao puts around its env (MAIN/PO4AO/util_simple.py), for the batched env.

``TorchWrapper`` (util_simple.py:201-221) converts torch <-> numpy around a NumPy env; the batched env
already speaks torch on the device, so here it only reproduces the *return convention* (float32
tensors, info as a list of pairs).  ``TimeDelayEnv`` (util_simple.py:25-52) is an action FIFO.
"""
from __future__ import annotations


def _torch():
    import torch
    return torch


class TorchWrapper:
    def __init__(self, env):
        self._env = env
        self.env = env

    def __getattr__(self, name):                 # gym.Wrapper attribute forwarding
        return getattr(self._env, name)

    def step(self, i, action):
        torch = _torch()
        if getattr(self._env, "output", "torch") == "numpy":
            next_obs, wfsf, reward, strehl, done, info = self._env.step(i, action.cpu().numpy())
            return (torch.tensor(next_obs, dtype=torch.float32), wfsf, reward, strehl, done,
                    [(k, torch.tensor(v, dtype=torch.float32)) for k, v in info.items()])
        next_obs, wfsf, reward, strehl, done, info = self._env.step(i, action)
        return next_obs.float(), wfsf, reward, strehl, done, [(k, v.float()) for k, v in info.items()]

    def reset(self):
        return _torch().as_tensor(self._env.reset(), dtype=_torch().float32)

    def reset_soft(self):
        return _torch().as_tensor(self._env.reset_soft(), dtype=_torch().float32)


class TimeDelayEnv:
    """Control delay of ``delay`` frames: the env receives the action issued ``delay`` steps earlier."""

    def __init__(self, env, delay):
        self._env = env
        self.env = env
        self.d = int(delay)
        self.action_buffer = []
        self._fill()

    def __getattr__(self, name):
        return getattr(self._env, name)

    def _zero(self):
        e = self._env
        if getattr(e, "output", "torch") == "numpy":
            import numpy as np
            return np.zeros((e.nActuator, e.nActuator))
        return _torch().zeros((e.n_envs, e.nActuator, e.nActuator), device=e.device, dtype=e.tdtype)

    def _fill(self):
        self.action_buffer = [self._zero() for _ in range(self.d)]

    def reset(self):
        obs = self._env.reset()
        self._fill()
        return obs

    def reset_soft(self):
        obs = self._env.reset_soft()
        self._fill()
        return obs

    def step(self, i, action):
        self.action_buffer.append(action)
        out = self._env.step(i, self.action_buffer[0])
        del self.action_buffer[0]
        return out


class Box:
    """Minimal stand-in for ``gym.spaces.Box`` (gymnasium is not a dependency of the env)."""

    def __init__(self, low, high, shape, dtype="float32"):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class DeviceHistory:
    """Sliding window of the last ``n_history`` images of every env, newest first, on the device: a MIRRORED ring buffer
    ``[n, 2 H, A, A]`` -- every image is written to slots p and p + H, so the window ``buf[:, p : p + H]`` is always one
    contiguous-in-time view.  A push is two slot writes (2 x n x A x A elements); nothing is rolled or concatenated (the
    reference rolls the whole history every step, OOPAOEnv_VPG.py:660-681: 131 MB per step at 1024 envs of the 41 x 41 DM)."""

    def __init__(self, n, n_history, n_act, device, dtype=None):
        torch = _torch()
        self.H = int(n_history)
        self.buf = torch.zeros((n, 2 * self.H, n_act, n_act), device=device, dtype=dtype or torch.float32)
        self.p = 0

    def push(self, img):
        self.p = (self.p - 1) % self.H
        self.buf[:, self.p] = img
        self.buf[:, self.p + self.H] = img

    def window(self):
        """[n, H, A, A] view, newest image at index 0."""
        return self.buf[:, self.p:self.p + self.H]

    def clear(self):
        self.buf.zero_()
        self.p = 0


class HistoryEnv:
    """gymnasium-style facade with the observation history kept on the device
    (MAIN/OOPAOEnv/OOPAOEnv_VPG.py:77-95 spaces, :553-608 step, :660-681 roll_buffer):

    * ``step(action) -> (obs_history, reward, terminated, truncated, info)`` -- no frame index argument; the wrapper
      counts frames itself (and wraps at ``nLoop``, the length of the env's telemetry);
    * ``obs_history`` is ``[n_envs, n_history, nAct, nAct]`` (``[n_history, nAct, nAct]`` for one env), newest image
      at index 0, rolled one slot per step;
    * the reward is the Strehl ratio (``reward = strehl``, :601-603);
    * actions pass through a FIFO of ``delay`` frames (``action_buffer``, :562-566), a 1-D action is first scattered to
      the actuator image (``vec_to_img``, :559-560).

    Nothing leaves the device for a batched env; for the single-env NumPy flavour the history is returned as a NumPy
    array like the reference's ``obs_history.cpu().numpy()``."""

    def __init__(self, env, n_history=20, delay=1):
        if delay < 1:
            raise ValueError("delay must be >= 1 (the reference applies action_buffer[0] after appending the new action)")
        self._env = env
        self.env = env
        self.n_history = int(n_history)
        self.delay = int(delay)
        self.t = 0
        A = env.nActuator
        self.single = getattr(env, "output", "torch") == "numpy"
        lead = () if self.single or env.n_envs == 1 else (env.n_envs,)
        self.observation_space = Box(-float("inf"), float("inf"), lead + (self.n_history, A, A))
        self.action_space = Box(-10.0, 10.0, lead + (A, A))
        self._alloc()

    def __getattr__(self, name):
        return getattr(self._env, name)

    def _alloc(self):
        torch = _torch()
        e = self._env
        dev = getattr(e, "device", "cpu")
        n = 1 if self.single else e.n_envs
        A = e.nActuator
        self._hist = DeviceHistory(n, self.n_history, A, dev)
        self.action_buffer = [torch.zeros((n, A, A), device=dev, dtype=torch.float32) for _ in range(self.delay - 1)]

    @property
    def obs_history(self):
        """[n, n_history, nAct, nAct], newest first: a view of the mirrored ring buffer (valid until the next step)."""
        return self._hist.window()

    def _out(self):
        if self.single:
            return self.obs_history[0].cpu().numpy()
        return self.obs_history[0] if self._env.n_envs == 1 else self.obs_history

    def reset(self, seed=None, options=None):
        """New turbulence (``generateNewPhaseScreen(seed)``), flat DM, one measurement, empty histories."""
        e = self._env
        e.atm.generateNewPhaseScreen(seed)
        e.dm.coefs = 0
        e.dm_prev = 0                                           # reset() also clears the integrator state (OOPAOEnv_VPG.py:120-121)
        e.tel * e.dm * e.wfs
        obs = _torch().as_tensor(e.reset_soft(), dtype=_torch().float32)
        self._alloc()
        self.t = 0
        self._push(obs)
        return self._out(), {}

    def _push(self, obs):
        torch = _torch()
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self._hist.buf.device)
        if obs.dim() == 2:
            obs = obs.unsqueeze(0)
        self._hist.push(obs)

    def step(self, action):
        torch = _torch()
        e = self._env
        a = torch.as_tensor(action, dtype=torch.float32, device=self._hist.buf.device)
        if a.shape[-1] != e.nActuator or a.dim() == 1 or (a.dim() == 2 and not self.single and e.n_envs > 1):
            a = e.vec_to_img(a, True)                              # command vector(s) -> actuator image(s)
        if a.dim() == 2:
            a = a.unsqueeze(0)
        self.action_buffer.append(a)
        act = self.action_buffer.pop(0)
        n_loop = int(e.param.nLoop) if hasattr(e, "param") else 1 << 30
        i = self.t % n_loop
        self.t += 1
        obs, _, _, strehl, _, info = e.step(i, act[0].cpu().numpy() if self.single else act)
        self._push(obs)
        terminated = truncated = False
        if not self.single and e.n_envs > 1:
            terminated = torch.zeros(e.n_envs, dtype=torch.bool, device=self._hist.buf.device)
            truncated = terminated.clone()
        return self._out(), strehl, terminated, truncated, {"strehl": strehl}
