"""The finest level of a 2000x2000 pyramid (4 Mpixel: the q-recomputing kernel's smallest level) under different forms / walks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octane_amd import capi
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kit = 6
pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=kit))
lev = kit - 1
def t(**kn):
    for k, v in kn.items():
        pl.tune(k, v)
    r = min(pl.probe(lev, 60)[0] for _ in range(3)) * 1e3
    return r
base = dict(fused_q=1, xcd=4, fused_rows=0)
print(f"{n}x{n} finest level, us per (even) iteration:")
print("  q-form, walk 4 (default):", f"{t(**base):.1f}")
for w in (0, 1, 3):
    print(f"  q-form, walk {w}:", f"{t(fused_q=1, xcd=w):.1f}")
print("  stored q, 128x16 tiles:", f"{t(fused_q=0, xcd=4, fused_rows=2):.1f}")
print("  stored q, 128x8 tiles:", f"{t(fused_q=0, xcd=4, fused_rows=1):.1f}")
print("  stored q, walk 0, 128x16:", f"{t(fused_q=0, xcd=0, fused_rows=2):.1f}")
for k, v in base.items():
    pl.tune(k, v)
pl.close()
