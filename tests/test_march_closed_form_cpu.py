"""The closed form of the march's stepping loop (csrc/common.h: ssd_run_to_const; r05, off by default in the kernels: no speed-up) replayed in numpy fp32.

`do { t += dt; } while (t < tt);` is a chain of fp32 additions; inside one binade it is an exact arithmetic progression t + k D with D = fl(t + dt) - t, so the
chain's first member >= tt is fma(k, D, t) for the right k.  The device code estimates k with a hardware reciprocal (1 ulp), corrects by one step either way and
falls back to the loop whenever its invariants (same binade, equal first two differences, c >= tt > c - D) do not hold.  This test replays exactly that control
flow -- with the reciprocal perturbed by +-1 ulp to force both corrections -- against the loop, on random (t, tt) over the parameter range of the renderer."""
import numpy as np

F = np.float32


def _loop(dt, t, tt):
    while True:
        t = F(t + dt)
        if not (t < tt):
            return t


def _closed(dt, t, tt, rcp_ulps):
    t1 = F(t + dt)
    if not (t1 < tt):
        return t1, "one step"
    t2 = F(t1 + dt)
    d1, d2 = F(t1 - t), F(t2 - t1)
    rcp = np.nextafter(F(1.0) / d1, F(np.inf if rcp_ulps > 0 else -np.inf)) if rcp_ulps else F(1.0) / d1
    k = F(np.ceil(F(F(tt - t) * F(rcp))))
    c = F(np.float64(k) * np.float64(d1) + np.float64(t))              # fma: one rounding of the exact value
    if c < tt:
        c = F(c + d1)
    if not (F(c - d1) < tt):
        c = F(c - d1)
    same_binade = ((t.view(np.uint32) ^ c.view(np.uint32)) >> 23) == 0
    if d1 == d2 and same_binade and not (c < tt) and F(c - d1) < tt:
        return c, "closed"
    x = t2
    while x < tt:
        x = F(x + dt)
    return x, "fallback"


def test_closed_form_stepping_equals_the_loop():
    rng = np.random.default_rng(11)
    kinds = {"one step": 0, "closed": 0, "fallback": 0}
    for dt in (F(2.0 * 1.7320508075688772 / 256.0), F(2.0 * 1.7320508075688772 / 64.0), F(0.01)):
        for _ in range(40000):
            t = F(rng.uniform(0.05, 6.0))
            tt = F(t + F(rng.uniform(0.0, 0.3)))
            want = _loop(dt, t, tt)
            for ulps in (0, 1, -1):
                got, kind = _closed(dt, t, tt, ulps)
                kinds[kind] += 1
                assert got == want, (float(dt), float(t), float(tt), ulps, kind, float(got), float(want))
    assert kinds["closed"] > 100000 and kinds["fallback"] > 100 and kinds["one step"] > 1000          # every branch is exercised
