#!/usr/bin/env python3
"""A/B timing of kernel variants built by tools/build_variants.sh (octane_amd/variants/*.so): every variant runs in a process of
its own (OCTANE_LIB selects the library), creates its plan with the usual best-of-n placement, and reports

  probe  -- us per launch of the finest-level fused PCG kernel with the stop test held open (octane_vof_plan_probe: varying
            weights, launches with and without x work alternate; the planes hold whatever they hold, so ablation builds --
            wrong results by design -- can be timed), an event pair around every launch;
  solve  -- for builds whose results are meant to be right: mean us per finest-level launch of a real one-level solve
            (3 GNC steps x 3 x 30 iterations, a third of them with unit weights) and whether its flow has the bits of the
            first variant's (a hash of u, v).

usage: time_variants.py [--size 5000] [--reps 2] [--only name,name] [--ablation name,name]   (run on the GPU box)"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(size, solve):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from octane_amd import capi, synth
    capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
    n = size
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=3, cgiters=30))
    out = {"trials_ms": [round(t, 4) for t in pl.placement_trials() if t > 0]}
    pr = [pl.probe(0, 41)[0] * 1e3 for _ in range(3)]
    out["probe_us"] = round(min(pr), 2)
    if solve:
        a, b = synth.lattice_scene(n, n, seed=20240615, device="cuda")
        u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        best = None
        for _ in range(2):
            u.zero_(); v.zero_()
            pl.set_profiling(True)
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
            torch.cuda.synchronize()
            p = pl.profile()
            pl.set_profiling(False)
            us = p.pass_a_ms / max(1, p.pass_a_launches) * 1e3
            best = us if best is None else min(best, us)
        out["solve_us"] = round(best, 2)
        out["launches"] = int(p.pass_a_launches)
        out["iters"] = pl.last_iterations()
        h = hashlib.sha1(u.cpu().numpy().tobytes()); h.update(v.cpu().numpy().tobytes())
        out["hash"] = h.hexdigest()[:12]
    pl.close()
    print("RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=5000)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--ablation", default="", help="variants whose results are wrong by design (probe only); names starting with 'abl' are always")
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        return child(args.size, args.child == "solve")
    vdir = os.path.join(ROOT, "octane_amd", "variants")
    names = sorted(f[:-3] for f in os.listdir(vdir) if f.endswith(".so"))
    if args.only:
        names = [n for n in names if n in args.only.split(",")]
    if "base" in names:
        names.remove("base"); names.insert(0, "base")
    abl = set(args.ablation.split(",")) if args.ablation else set()
    res = {n: [] for n in names}
    for rep in range(args.reps):
        for n in names:
            env = dict(os.environ, OCTANE_LIB=os.path.join(vdir, n + ".so"))
            mode = "probe" if (n.startswith("abl") or n in abl) else "solve"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--size", str(args.size), "--child", mode],
                               env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(f"{n}: FAILED rc={r.returncode} {r.stderr[-300:]}", flush=True)
                continue
            d = json.loads(line[0][7:])
            res[n].append(d)
            print(f"{args.size} rep {rep} {n:10s} probe {d['probe_us']:7.2f} us" + (f"  solve {d['solve_us']:7.2f} us ({d['launches']} launches, {d['iters']} iterations, flow {d['hash']})" if "solve_us" in d else "")
                  + (f"  placement trials min {min(d['trials_ms']):.4f} ms" if d['trials_ms'] else ""), flush=True)
    ref = None
    print(f"--- summary at {args.size}^2 (best of {args.reps})")
    for n in names:
        if not res[n]:
            continue
        pb = min(d["probe_us"] for d in res[n])
        sv = min((d["solve_us"] for d in res[n] if "solve_us" in d), default=None)
        hs = {d.get("hash") for d in res[n]} - {None}
        if ref is None:
            ref = (pb, sv, hs)
        print(f"{n:10s} probe {pb:7.2f} us ({pb / ref[0] - 1:+.1%})" + (f"   solve {sv:7.2f} us ({sv / ref[1] - 1:+.1%})  bits {'same' if hs == ref[2] else 'DIFFERENT'}" if sv and ref[1] else ""))


if __name__ == "__main__":
    main()
