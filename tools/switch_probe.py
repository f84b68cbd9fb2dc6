"""What a kernel pays for NOT following a launch of itself.  Three graphs of 100 launches each, same two GEMM shapes (to_out:
2048 x 1280 x 1280 and ff.out: 2048 x 1280 x 5120, wave-specialised 64 x 160, independent operands unless stated):
  grouped      A x 50, then B x 50                  (every launch follows its own code; operands L2 / MALL-warm)
  alternating  A B A B ...                          (every launch follows the OTHER kernel instantiation... the same code object here, other shape)
  other_kernel A G A G ... with G = a GroupNorm apply over an unrelated tensor (different code between every two A's)
  dependent    A reads what the previous launch wrote (chain x -> A -> y -> A' -> x ...), the forward's situation
gpurun -- python tools/switch_probe.py > profiles/rNN_kernel_switch_probe.txt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from imagharmony_amd.ctx import Ctx
DEV = "cuda:0"; dtype = torch.bfloat16


def timed(build, reps=5):
    rec = Ctx(DEV, dtype, record=True)
    n = build(rec)
    rec.capture()
    rec.replay(); torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rec.replay(); e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3
        best = t if best is None else min(best, t)
    return best, n


g = torch.Generator(device="cpu").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV, dtype)
xa, wa, ya, ra = rn(2048, 1280), rn(1280, 1280, sc=1280 ** -0.5), torch.empty(2048, 1280, device=DEV, dtype=dtype), rn(2048, 1280)
xb, wb, yb = rn(2048, 5120), rn(1280, 5120, sc=5120 ** -0.5), torch.empty(2048, 1280, device=DEV, dtype=dtype)
gx = rn(2, 1024, 1280); gam, bet = torch.ones(1280, device=DEV), torch.zeros(1280, device=DEV)
A = lambda c: c.gemm(xa, wa, out=ya, residual=ra, cfg=(2464, 160, 1))
B = lambda c: c.gemm(xb, wb, out=yb, cfg=(2464, 160, 1))


def grouped(c):
    for _ in range(50): A(c)
    for _ in range(50): B(c)
    return 100


def alternating(c):
    for _ in range(50): A(c); B(c)
    return 100


def only(f, n=100):
    def b(c):
        for _ in range(n): f(c)
        return n
    return b


def other_kernel(c):
    for _ in range(50):
        A(c)
        c.groupnorm(gx, gam, bet, 32, 1e-5, silu=False)
    return 100


def dependent(c):
    # x -> A -> y -> A' -> x: each launch's token operand is what the previous launch wrote
    x0, x1 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
    for _ in range(50):
        c.gemm(x0, wa, out=x1, cfg=(2464, 160, 1))
        c.gemm(x1, wa, out=x0, cfg=(2464, 160, 1))
    return 100


def independent_same(c):
    x0, x1 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
    x2, x3 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
    for _ in range(50):
        c.gemm(x0, wa, out=x1, cfg=(2464, 160, 1))
        c.gemm(x2, wa, out=x3, cfg=(2464, 160, 1))
    return 100


ta, _ = timed(only(A)); tb, _ = timed(only(B)); tg, _ = timed(only(lambda c: c.groupnorm(gx, gam, bet, 32, 1e-5, silu=False), 50))
print(f"A alone x100: {ta / 100:.2f} us each; B alone x100: {tb / 100:.2f} us each; GroupNorm alone x50: {tg / 50:.2f} us each")
for name, f in (("grouped", grouped), ("alternating", alternating)):
    t, n = timed(f)
    print(f"{name:12s}: {t:.1f} us for {n} launches; expected from the alone times {(ta + tb) / 2:.1f}; per launch pair penalty {(t - (ta + tb) / 2) / 50:.2f} us")
t, n = timed(other_kernel)
print(f"other_kernel: {t:.1f} us for 50 A + 50 GroupNorm; expected {ta / 2 + tg:.1f}; penalty per A {(t - ta / 2 - tg) / 50:.2f} us")
ti, _ = timed(independent_same); td, _ = timed(dependent)
print(f"same kernel, independent operands: {ti / 100:.2f} us each; each launch reading what the previous one wrote: {td / 100:.2f} us each")

# ---- does the dependency penalty depend on WHERE the producer wrote?  XCD cell shapes (Ctx.xcd_cells): with 8 x 1 cells XCD x owns the
# same 256 token rows in producer and consumer (it reads what it wrote itself); with 1 x 8 every XCD reads rows all eight wrote
for cells in (0, 2, 3, 4, 5):
    def dep(c):
        c.xcd_cells = cells
        x0, x1 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
        for _ in range(50):
            c.gemm(x0, wa, out=x1, cfg=(2464, 160, 1))
            c.gemm(x1, wa, out=x0, cfg=(2464, 160, 1))
        return 100

    def ind(c):
        c.xcd_cells = cells
        x0, x1 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
        x2, x3 = rn(2048, 1280), torch.empty(2048, 1280, device=DEV, dtype=dtype)
        for _ in range(50):
            c.gemm(x0, wa, out=x1, cfg=(2464, 160, 1))
            c.gemm(x2, wa, out=x3, cfg=(2464, 160, 1))
        return 100
    ti, _ = timed(ind); td, _ = timed(dep)
    print(f"xcd cells {cells}: independent {ti / 100:.2f} us, dependent {td / 100:.2f} us per launch (+{(td - ti) / 100:.2f})")
