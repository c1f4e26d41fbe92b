"""Does RCCL accept two ranks on ONE device?  (why the row bands' fallback transport is runtime copies, not RCCL.)
Starts two ranks that both use cuda:0 and tries an all-reduce over the nccl (= RCCL) backend."""
import os, sys, subprocess, socket

if "RANK" not in os.environ:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, __file__], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(port), NCCL_DEBUG="WARN"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out = p.communicate(timeout=120)[0]
        except subprocess.TimeoutExpired:
            p.kill(); out = "TIMEOUT"
        print(f"--- rank {r}: exit {p.returncode}\n" + "\n".join(l for l in out.splitlines() if "amdgpu.ids" not in l)[-1500:])
    sys.exit(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda:0")
dist.all_reduce(t)
torch.cuda.synchronize()
print("all_reduce over two ranks on cuda:0 gave", t.tolist())
dist.destroy_process_group()
