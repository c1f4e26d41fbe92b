#!/usr/bin/env python3
"""us per launch of k_assemble at the finest level of a one-level plan, with the fast exact forms of round 3 (three-instruction division by
alpha, ...) and with IEEE divisions throughout (OCTANE_TUNE_ASM_FAST=0); each in a process of its own.
With --nc the same for 1, 2 and 3 channels (round 4: template instances for two and three channels; OCTANE_TUNE_ASM_GENERIC=1 forces
the generic instance, any channel count and every switch at run time, for the comparison).
usage: time_assembly.py [--nc] [size ...]"""
import os; os.environ.setdefault("OCTANE_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "liboctane_vof_diag.so"))  # the OCTANE_TUNE_* tuning variables exist in the diagnostic library only (round 5)
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n, nc=1):
    sys.path.insert(0, ROOT)
    import torch
    from octane_amd import capi, synth
    capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
    a, b = synth.lattice_scene(n, n, seed=20240615, nchan=nc, device="cuda")
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    pl = capi.Plan(n, n, nc, capi.FlowParams(kiters=1, liters=3, cgiters=2))
    s = torch.cuda.current_stream().cuda_stream
    best = None
    for rep in range(3):
        u.zero_(); v.zero_()
        pl.set_profiling(rep > 0)
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        if rep > 0:
            p = pl.profile()
            us = p.assemble_ms / max(1, p.assemble_launches) * 1e3
            best = us if best is None else min(best, us)
    bits = capi.lib().octane_selftest_assembly_math_bits(0, 5.0)
    print("RESULT " + json.dumps({"us": round(best, 1), "launches": int(p.assemble_launches), "bits": bits}), flush=True)
    pl.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 1)
        sys.exit(0)
    if "--nc" in sys.argv:
        for n in [int(x) for x in sys.argv[1:] if x != "--nc"] or [5000]:
            for nc in (1, 2, 3):
                for generic in ("0", "1"):
                    env = dict(os.environ, OCTANE_TUNE_ASM_GENERIC=generic, OCTANE_TUNE_PLACEMENT_TRIALS="1")
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n), str(nc)], env=env, capture_output=True, text=True, timeout=600)
                    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                    print(f"{n}x{n} nchan={nc} {'generic instance' if generic == '1' else 'template instance'}: " + (line[0][7:] if line else f"FAILED {r.stderr[-300:]}"), flush=True)
        sys.exit(0)
    for n in [int(x) for x in sys.argv[1:]] or [5000, 2000]:
        for fast in ("1", "0"):
            env = dict(os.environ, OCTANE_TUNE_ASM_FAST=fast, OCTANE_TUNE_PLACEMENT_TRIALS="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"{n}x{n} OCTANE_TUNE_ASM_FAST={fast}: " + (line[0][7:] if line else f"FAILED {r.stderr[-300:]}"), flush=True)
