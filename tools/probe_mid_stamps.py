"""Where thread 0 of the persistent mid-level solve's workgroups spends an iteration (shader-clock stamps at the seams,
pcg_persist_diag.hip), next to the time per iteration of the production kernel.
usage: probe_mid_stamps.py [size ...]"""
import os as _os  # the stamped kernels live in the diagnostic library (make -C octane_amd/csrc DIAG=1), never in the product
_os.environ.setdefault('OCTANE_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'octane_amd', 'liboctane_vof_diag.so'))
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth
capi = capi.dev()      # tune / probe / self-tests: the DIAGNOSTIC library's binding (the product library does not export them)
sizes = [int(a) for a in sys.argv[1:]] or [157, 313, 625, 1000, 1250]
L = capi.lib()
L.octane_vof_mid_stamps.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
names = ["wave 0 waits for sums + ring", "slowest wave has them (barrier)", "scalars, ring pixel, update loop", "barrier", "stencil loop",
         "s_acc + barrier", "workgroup sums + edges published"]
CG = 30
for n in sizes:
    a, b = synth.lattice_scene(n, n, seed=n, device="cuda")
    u = torch.zeros(n, n, device="cuda"); v = torch.zeros(n, n, device="cuda")
    pl = capi.Plan(n, n, 1, capi.FlowParams(kiters=1, liters=1, cgiters=CG))
    if os.environ.get("SMALL_MAX"):
        pl.tune("small_max", int(os.environ["SMALL_MAX"]))
    st = torch.cuda.current_stream().cuda_stream
    for diag in (0, 1):
        pl.tune("persist_diag", diag)
        buf = (C.c_ulonglong * 32)()
        L.octane_vof_mid_stamps(0, buf)          # clear
        for _ in range(3):
            u.zero_(); v.zero_(); pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        its = pl.last_iterations()
        if diag:
            assert L.octane_vof_mid_stamps(0, buf) == 0
            print(f"--- {n}x{n}: stamped build, {dt * 1e6:.0f} us per pyramid of one level ({its} iterations)")
            for off, what in ((0, "interior sub-domains (fast path)"), (16, "border / partial sub-domains")):
                s = list(buf)[off:off + 16]
                if not s[14]:
                    continue
                cnt = s[14]
                tot = sum(s[:7])
                print(f"  {what}: {cnt // max(1, its)} workgroups, {tot / cnt:.0f} cycles per iteration")
                for i in range(7):
                    print(f"    {names[i]:36s} {s[i] / cnt:8.0f} cycles  {100.0 * s[i] / tot:5.1f} %")
                print(f"    failed polling rounds of thread 0 per iteration: {s[7] / cnt:.2f}")
        else:
            print(f"=== {n}x{n}: production build, {dt * 1e6:.0f} us per pyramid of one level ({its} iterations)")
    pl.tune("persist_diag", 0)
    pl.close()
