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
