"""Stand-in for torchdiffeq.odeint: fixed-grid explicit Euler only (integrators.py:106-113)."""
import torch


def odeint(func, y0, t, method="euler", atol=None, rtol=None, **kw):
    if method != "euler":
        raise NotImplementedError("oracle shim restates fixed-grid Euler only (dopri5 is out of scope)")
    ys = [y0]
    y = y0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        y = y + dt * func(t[i], y)
        ys.append(y)
    return torch.stack(ys, 0)
