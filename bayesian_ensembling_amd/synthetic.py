"""Seeded synthetic member x timestep fields: SplitMix64 -> Box-Muller; per member a quadratic
trend 1.5*a*tau^2 + b with a = 0.8 + 0.4u, b = 0.1z, AR(1) realisation noise (phi = 0.6, innovation sd 0.12);
observations use a = 1, b = 0.  The oracle carries its own copy of this generator; tests compare the two."""
from __future__ import annotations

import math

import numpy as np

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next_u64(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self):
        return ((self.next_u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0)

    def normal(self):
        u1 = self.uniform()
        u2 = self.uniform()
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def synthetic_ensemble(M, N, R=8, R_obs=3, seed=0xB4E5):
    """Returns (data (M,R,N), obs (R_obs,N))."""
    rng = SplitMix64(seed)
    tau = np.arange(N, dtype=np.float64) / max(N - 1, 1)
    data = np.empty((M, R, N))

    def ar1():
        e = np.empty(N)
        prev = 0.0
        for t in range(N):
            prev = 0.6 * prev + 0.12 * rng.normal()
            e[t] = prev
        return e

    for m in range(M):
        a = 0.8 + 0.4 * rng.uniform()
        b = 0.1 * rng.normal()
        sig = 1.5 * a * tau * tau + b
        for r in range(R):
            data[m, r] = sig + ar1()
    obs = np.empty((R_obs, N))
    for r in range(R_obs):
        obs[r] = 1.5 * tau * tau + ar1()
    return data, obs
