import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octane_amd import capi, synth
n = 5000
dev = torch.device('cuda', 0)
a, b = synth.lattice_scene(n, n, seed=1, device=dev)
u = torch.zeros(n, n, device=dev); v = torch.zeros(n, n, device=dev)
stream = torch.cuda.current_stream().cuda_stream
prm = capi.FlowParams(kiters=1, liters=1, cgiters=30)
for rep in range(2):
    for skew in [0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144, 524288]:
        os.environ['OCTANE_TUNE_SKEW'] = str(skew)
        pl = capi.Plan(n, n, 1, prm)
        pl.set_profiling(True)
        for _ in range(2):
            u.zero_(); v.zero_()
            pl.run_device(a.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), stream)
            torch.cuda.synchronize()
        p = pl.profile()
        print(f'skew {skew:7d}: A {p.pass_a_ms/p.pass_a_launches*1e3:7.1f} us  B {p.pass_b_ms/p.pass_b_launches*1e3:7.1f} us', flush=True)
        pl.close()
