#!/usr/bin/env python3
"""Compact timeline of the vector-memory events of a kernel in a gfx950 code object, in program order: loads (L), LDS-DMA (D),
stores (S), `s_waitcnt vmcnt(N)` (wN), barriers (|), branch targets (:), the marked DMA wait (W*).  Runs of the same event are
counted.  What it is for: seeing where the compiler put full waits (w0) relative to the inline-asm DMAs it does not count
(round 3: a `vmcnt(4)` at the top of the tile loop and `vmcnt(0)` in both arms of a conditional load made the 18 own loads of
a tile go out in three round trips).
usage: isa_vm_timeline.py <object> [substring of the kernel name]"""
import sys
import os
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_dma_wait as c  # noqa: E402


def timeline(ins):
    out, last, n = [], None, 0

    def push(tok):
        nonlocal last, n
        if tok == last:
            n += 1
            return
        if last is not None:
            out.append(last if n == 1 else f"{last}x{n}")
        last, n = tok, 1
    for i, (mn, ops, labels) in enumerate(ins):
        if labels:
            push(":")
        if mn.startswith("global_load_lds"):
            push("D")
        elif mn.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            push("L")
        elif mn.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
            push("S")
        elif mn == "s_barrier":
            push("|")
        elif mn == "s_waitcnt" and "vmcnt" in ops:
            k = ops.split("vmcnt(")[1].split(")")[0]
            marked = i + 1 < len(ins) and ins[i + 1][0] == "s_setprio"
            push(f"W*{k}" if marked else f"w{k}")
            last_n = None
        elif mn == "s_endpgm":
            push("END")
    push(None)
    return " ".join(out)


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as wd:
        funcs = c.disassemble(c.device_code_object(sys.argv[1], wd))
    for name, ins in funcs.items():
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        if not ins:
            continue
        print(f"== {name} ({len(ins)} instructions)")
        print(timeline(ins))
