#!/usr/bin/env python3
"""ISA-level guard for the counted wait of the LDS-DMA PCG kernel (octane_amd/csrc/pcg_fused_q_dma.hip).

Phase 0 of a DMA-staged tile waits with `s_waitcnt vmcnt(N)` (N = the tile's own register loads issued since the DMA)
instead of `vmcnt(0)`: vector-memory operations complete in issue order on gfx9-class hardware, so once at most N are
outstanding and at least N were issued after the last `global_load_lds`, the DMA has landed.  The compiler does not
track the inline-asm DMA; if it ever emits FEWER than N vector-memory instructions on some path between the last DMA
and that wait (a merged or hoisted load, a load turned into an s_load, a conditional one), the wait passes early and
phase 1 reads a stale tile from LDS -- timing dependent, invisible to the parity tests.

This tool proves the property on the built code object.  It disassembles the gfx950 code (llvm-objdump
--symbolize-operands), builds the control-flow graph of every kernel, and runs a forward data-flow analysis whose
state is "the minimum, over all paths, of the number of vector-memory instructions issued since the last
global_load_lds" (INF = no DMA can be outstanding: kernel entry, or a wait that covers it).  Every marked wait --
`s_waitcnt vmcnt(N)` directly followed by `s_setprio 0`, the marker the kernel's inline asm emits -- must be reached
with state >= N on every path.  Loops (the DMA is issued in one trip of the tile loop and waited for in the next) are
handled by iterating to the fixed point.

usage: check_dma_wait.py <object or code-object file> [--expect-kernels N] [--objdump PATH]
exit 0 = every marked wait is safe; 1 = some marked wait can pass before the DMA has landed; 2 = cannot analyse."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
INF = 1 << 20
CAP = 255          # counts saturate here (vmcnt has 6 bits: nothing above 63 can matter)
VMEM_PREFIXES = ("global_load", "global_store", "global_atomic", "buffer_", "scratch_load", "scratch_store", "flat_load",
                 "flat_store", "flat_atomic", "image_", "tbuffer_")


def device_code_object(path: str, workdir: str) -> str:
    """A host object with an offload bundle -> the extracted gfx950 code object; a bare code object -> itself."""
    with open(path, "rb") as f:
        head = f.read(20)
    if head[:4] == b"\x7fELF" and head[18:20] == b"\xe0\x00":       # e_machine = EM_AMDGPU
        return path
    local = os.path.join(workdir, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=workdir)
    cands = [f for f in os.listdir(workdir) if "amdgcn" in f and f.startswith(os.path.basename(path))]
    if not cands:
        raise RuntimeError(f"no amdgcn code object inside {path}")
    return os.path.join(workdir, cands[0])


def disassemble(path: str):
    """-> {kernel name: [(mnemonic, operands, [labels at this instruction])]}"""
    out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", path], check=True, capture_output=True, text=True).stdout
    funcs, cur, pending = {}, None, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:\s*$", line)
        if m:
            name = m.group(1)
            if re.fullmatch(r"L\d+", name):
                pending.append(name)
            else:
                cur = funcs.setdefault(name, [])
                pending = []
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else "", pending))
        pending = []
    return funcs


def analyse(name, ins):
    """-> (list of (index, N, min state) for every marked wait, number of DMA instructions)"""
    label_at = {}
    for i, (_, _, labels) in enumerate(ins):
        for lb in labels:
            label_at[lb] = i
    n = len(ins)
    succ = [[] for _ in range(n)]
    for i, (mn, ops, _) in enumerate(ins):
        if mn in ("s_endpgm", "s_endpgm_saved"):
            continue
        if mn in ("s_setpc_b64", "s_swappc_b64", "s_call_b64", "s_cbranch_g_fork", "s_cbranch_join"):
            raise RuntimeError(f"{name}: indirect control flow ({mn}) -- cannot analyse")
        if mn == "s_branch" or mn.startswith("s_cbranch"):
            tgt = ops.split()[-1].strip()
            if tgt not in label_at:
                raise RuntimeError(f"{name}: branch to unknown label {tgt!r}")
            succ[i].append(label_at[tgt])
            if mn == "s_branch":
                continue
        if i + 1 < n:
            succ[i].append(i + 1)

    def transfer(i, s):
        mn, ops, _ = ins[i]
        if mn.startswith("global_load_lds"):
            return 0
        if mn.startswith(VMEM_PREFIXES):
            return s if s >= INF else min(s + 1, CAP)
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m and s < INF and s >= int(m.group(1)):
                return INF
            if not m and re.fullmatch(r"\s*(0x[0-9a-f]+|\d+)\s*", ops):     # a raw immediate: decode vmcnt (bits 3:0 and 15:14 on gfx9)
                imm = int(ops.strip(), 0)
                k = (imm & 0xF) | (((imm >> 14) & 0x3) << 4)
                if s < INF and s >= k:
                    return INF
        return s

    state_in = [None] * n
    state_in[0] = INF
    work = [0]
    while work:
        i = work.pop()
        out = transfer(i, state_in[i])
        for j in succ[i]:
            if state_in[j] is None or out < state_in[j]:
                state_in[j] = out
                work.append(j)
    marked = []
    for i, (mn, ops, _) in enumerate(ins):
        if mn == "s_waitcnt" and i + 1 < n and ins[i + 1][0] == "s_setprio" and ins[i + 1][1].strip() in ("0", "0x0"):
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if not m:
                raise RuntimeError(f"{name}: marked wait without a vmcnt field: {ops!r}")
            if state_in[i] is not None:                  # (None: unreachable code)
                marked.append((i, int(m.group(1)), state_in[i]))
    ndma = sum(1 for mn, _, _ in ins if mn.startswith("global_load_lds"))
    return marked, ndma


def check(path: str, verbose: bool = True):
    """-> (ok, report lines, number of kernels with a marked wait)"""
    with tempfile.TemporaryDirectory() as wd:
        funcs = disassemble(device_code_object(path, wd))
    ok, lines, nk = True, [], 0
    for name, ins in funcs.items():
        if not ins:
            continue
        marked, ndma = analyse(name, ins)
        if ndma and not marked:
            lines.append(f"{name}: {ndma} global_load_lds instructions and NO marked wait")
            ok = False
            continue
        if marked:
            nk += 1
        for i, want, have in marked:
            safe = have >= want
            ok = ok and safe
            what = "no DMA can be outstanding (an earlier wait covers it)" if have >= INF else f"at least {have} vector-memory instructions since the last DMA on every path"
            lines.append(f"{name}: s_waitcnt vmcnt({want}) at instruction {i}: {what} -> {'safe' if safe else 'UNSAFE: the wait can pass before the DMA has landed'}")
    return ok, lines, nk


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    expect = None
    if "--expect-kernels" in argv:
        expect = int(argv[argv.index("--expect-kernels") + 1])
    try:
        ok, lines, nk = check(argv[1])
    except (RuntimeError, subprocess.CalledProcessError, OSError) as e:
        print(f"check_dma_wait: {e}")
        return 2
    print("\n".join(lines))
    if expect is not None and nk != expect:
        print(f"check_dma_wait: {nk} kernels with a marked wait, expected {expect}")
        return 1
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv))
