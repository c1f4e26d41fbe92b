#!/usr/bin/env python3
"""EXPERIMENT (round 5): what would forming the updated flow INSIDE k_assemble cost?

The review proposed fusing the flow update u += x (+ alpha p of the last one or two iterations) into the next assembly.  DESIGN 5 puts the
idea down with arithmetic; this script MEASURES the part of it that can be measured without building the real thing: a variant of
k_assemble whose 18 loads of the 3 x 3 neighbourhood of u, v each become `u + x + a * p` (two more loads and an FMA per point, from the
planes the update kernel reads) and which stores the centre pixel's new u, v (8 B/pixel, into two unused planes), built OUT OF TREE from a
patched copy of vof_kernels.hip into octane_amd/variants/asm_fuse_probe.so.  Values are not meaningful (x, p hold whatever the last
solve left), the instruction and memory streams are those a fused form could not avoid.  If the assembly slows down by more than the
separate update kernel costs (193 us at 5000^2, 44 at 2500^2), fusion loses whatever else is done.

   python tools/experiments/asm_fuse_probe.py build          (here: hipcc cross-compiles)
   python tools/experiments/asm_fuse_probe.py time [sizes]   (on the GPU box: product library against the variant)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "octane_amd", "csrc")
OUT = os.path.join(ROOT, "octane_amd", "variants", "asm_fuse_probe.so")


def build():
    src = open(os.path.join(CSRC, "vof_kernels.hip")).read()
    a = src.index("      const float *U = L.u, *V = L.v;")
    b = src.index("        float Ue = sq(ue - uc)")
    body = src[a:b]
    body, nu = re.subn(r"\bU\[([^\]]+)\]", r"U_AT(\1)", body)
    body, nv = re.subn(r"\bV\[([^\]]+)\]", r"V_AT(\1)", body)
    body = body.replace("      const float *U = L.u, *V = L.v;",
                        "      const float *U = L.u, *V = L.v;\n"
                        "#define U_AT(i) (U[i] + (L.xu[i] + 0.37f * L.pf_u[0][i]))\n"
                        "#define V_AT(i) (V[i] + (L.xv[i] + 0.37f * L.pf_v[0][i]))")
    assert nu == 9 and nv == 9, (nu, nv)
    patched = src[:a] + body + src[b:]
    store = "        const size_t o = rc + ii;\n        L.a1[o] = a1;"
    assert store in patched
    patched = patched.replace(store, "        const size_t o = rc + ii;\n        L.mu[o] = uc; L.mv[o] = vc;      // the fused form's store of the updated flow\n        L.a1[o] = a1;")
    os.makedirs("/tmp/octane_vb", exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = os.path.join(CSRC, "_asm_fuse_probe_tmp.hip")          # next to its headers; removed again below
    open(tmp, "w").write(patched)
    try:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
                               "-Wno-unused-function", "-c", tmp, "-o", "/tmp/octane_vb/asm_fuse_probe.o"])
    finally:
        os.remove(tmp)
    objs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and not f.endswith(".diag.o") and f != "vof_kernels.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["/tmp/octane_vb/asm_fuse_probe.o", "-lpthread"])
    print("built", OUT)


def child(n):
    sys.path.insert(0, ROOT)
    import torch
    from octane_amd import capi, synth
    a, b = synth.lattice_scene(n, n, seed=20240615, device="cuda")
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=3, cgiters=2))
    s = torch.cuda.current_stream().cuda_stream
    best = None
    for rep in range(4):
        u.zero_(); v.zero_()
        pl.set_profiling(rep > 0)
        pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), s)
        torch.cuda.synchronize()
        if rep > 0:
            p = pl.profile()
            us = p.assemble_ms / max(1, p.assemble_launches) * 1e3
            best = us if best is None else min(best, us)
    print("RESULT " + json.dumps({"assemble_us": round(best, 1), "update_us": round(p.update_ms / max(1, p.update_launches) * 1e3, 1) if hasattr(p, "update_ms") else None}), flush=True)
    pl.close()


def main():
    if sys.argv[1:2] == ["build"]:
        return build()
    if sys.argv[1:2] == ["--child"]:
        return child(int(sys.argv[2]))
    sizes = [int(x) for x in sys.argv[2:]] or [5000, 2500, 1250]
    for n in sizes:
        for name, lib in (("product", None), ("u + x + a p formed in the assembly", OUT)):
            env = dict(os.environ)
            if lib:
                env["OCTANE_LIB"] = lib
            else:
                env.pop("OCTANE_LIB", None)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            print(f"{n}x{n} k_assemble, {name}: " + (line[0][7:] if line else f"FAILED {r.stderr[-400:]}"), flush=True)


if __name__ == "__main__":
    main()
