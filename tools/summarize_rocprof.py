#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run of bench.py into the
per-kernel / per-pyramid-level table kept under profiles/.

usage: summarize_rocprof.py <dir with *_kernel_trace.csv> <out.md> [size kiters liters cgiters]
Launch order inside one pyramid is fixed (coarse -> fine; per level 3*liters solves of
1 assemble + cgiters x (pass A, pass B) + 1 update), so the level of a dispatch follows from its
index among the dispatches of the same kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict


def norm(short):
    """Template instances and the fused flow update under one name each."""
    if short.startswith("k_pcg_pass_a"):
        return "k_pcg_pass_a"
    if short.startswith("k_pcg_fused"):
        return "k_pcg_fused"
    if short.startswith("k_flow_update"):
        return "k_flow_update"
    if short.startswith("k_assemble"):
        return "k_assemble"
    return short


def pmc_section(d, size, kiters, per_level, two_pass_levels, qform=False, qname="k_pcg_fused_q"):
    """HBM traffic of the finest level from the FETCH_SIZE / WRITE_SIZE passes (sub-directories fetch/ and
    write/ of the profile directory).  Units and the gfx950 correction follow MI355X_MICROARCH.md 'HBM':
    both counters are in KiB; FETCH_SIZE reads exactly half the bytes of a wide (16 B/lane) coalesced
    stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores."""
    out = []
    vals = {}
    for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        fs = glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not fs:
            return out
        agg = defaultdict(list)
        with open(fs[0]) as f:
            for r in csv.DictReader(f):
                if "octane::" not in r["Kernel_Name"] or r["Counter_Name"] != cname:
                    continue
                short = r["Kernel_Name"].split("octane::")[1].split("(")[0]
                short = norm(short)
                agg[short].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        if "k_assemble" in agg:      # placement-trial launches precede the first assembly
            d_first = min(d for d, _ in agg["k_assemble"])
            for k in ("k_pcg_pass_a", "k_pcg_pass_b", "k_pcg_fused"):
                if k in agg:
                    agg[k] = [x for x in agg[k] if x[0] > d_first]
        for k, n in per_level.items():
            if k not in agg:
                continue
            v = [x for _, x in sorted(agg[k])]
            nl = len(two_pass_levels[k])
            per_pyr = n * nl
            if len(v) < per_pyr:
                continue
            sel = []
            for p in range(len(v) // per_pyr):
                sel += v[p * per_pyr + (nl - 1) * n:(p + 1) * per_pyr]
            vals[(k, cname)] = sum(sel) / len(sel)
    px = size * size
    # pass A reads 36 B/px, 28 in the first of the three GNC steps (wx / wy are the constant -1 there): mean 33.33
    # the fused iteration reads r q p + five coefficient planes = 44 B/px (36 in the first GNC step) and, every second launch,
    # x and the p before last (16): mean 52 - 8/3; it writes r p q = 24 and x every second launch (8): mean 28;
    # its flow update also reads the last p (the pending x update).  The q-recomputing form (k_pcg_fused_q, large levels)
    # neither reads nor writes q: 8 B/px less on either side
    alg = {"k_pcg_pass_a": (100.0 / 3.0, 16), "k_pcg_pass_b": (40, 16),
           "k_pcg_fused": (44 - 8.0 / 3.0, 20) if qform else (52 - 8.0 / 3.0, 28), "k_assemble": (52, 36),
           "k_flow_update": (24 if ("k_pcg_fused", "FETCH_SIZE") in vals else 16, 16 if ("k_pcg_fused", "FETCH_SIZE") in vals else 8)}
    out += ["", "## HBM traffic per launch at the finest level (rocprofv3 --pmc, separate passes)", "",
            "FETCH_SIZE x2 (gfx950 wide-load correction), WRITE_SIZE x1, both KiB -> bytes. Infinity-Cache hits are",
            "counted by these counters (they sit on the L2's fabric side), so this is L2<->fabric traffic, an upper",
            "bound on HBM bytes.", "",
            "| kernel | read MB (FETCH_SIZE x 2) | algorithmic read MB | write MB (WRITE_SIZE) | algorithmic write MB | total / algorithmic |",
            "|---|---|---|---|---|---|"]
    traffic = {}
    for k in ("k_pcg_fused", "k_pcg_pass_a", "k_pcg_pass_b", "k_assemble", "k_flow_update"):
        if (k, "FETCH_SIZE") not in vals:
            continue
        rd = vals[(k, "FETCH_SIZE")] * 1024 * 2
        wr = vals[(k, "WRITE_SIZE")] * 1024
        ar, aw = alg[k][0] * px, alg[k][1] * px
        traffic[k] = {"read_bytes": round(rd), "write_bytes": round(wr), "size": size,
                      "kernel": qname if (k == "k_pcg_fused" and qform) else k,
                      "how": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-load correction) and --pmc WRITE_SIZE, separate passes, "
                             "mean over the finest-level launches of bench.py"}
        out.append(f"| {k} | {rd / 1e6:.0f} | {ar / 1e6:.0f} | {wr / 1e6:.0f} | {aw / 1e6:.0f} | {(rd + wr) / (ar + aw):.3f} |")
    if traffic:
        import json
        import subprocess
        try:
            traffic["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or os.environ.get("OCTANE_COMMIT", "unknown")
        except OSError:
            traffic["commit"] = os.environ.get("OCTANE_COMMIT", "unknown")
        # the text of the dominant kernel these counters were measured on: bench.py reports the traffic only while the text is the same
        traffic["kernel_source_sha1"] = kernel_source_sha1()
        with open(os.path.join(os.path.dirname(os.path.abspath(sys.argv[2])), "traffic.json"), "w") as f:
            json.dump(traffic, f, indent=1)
    return out


def kernel_source_sha1():
    """sha1 over the sources of the finest-level PCG kernel (what bench.py's roofline block is about)."""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "octane_amd", "csrc")
    h = hashlib.sha1()
    for name in ("pcg_fused_q_dma.hip", "pcg_fused_q_phase1.inc", "pcg_fused_q_phase2.inc", "device_util.hpp", "vof_kernels.hpp"):
        with open(os.path.join(root, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def main():
    d, out = sys.argv[1], sys.argv[2]
    size, kiters, liters, cgiters = (int(x) for x in (sys.argv[3:7] if len(sys.argv) >= 7 else (5000, 8, 3, 30)))
    cands = glob.glob(os.path.join(d, "trace", "**", "*_kernel_trace.csv"), recursive=True) or \
        glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    trace = cands[0]
    rows = defaultdict(list)
    meta = {}
    qform = False
    qname = "k_pcg_fused_q"
    with open(trace) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "octane::" not in name:
                continue
            short = name.split("octane::")[1].split("(")[0]
            if short.startswith("k_pcg_fused_q"):
                qform, qname = True, short.split("<")[0]
            short = norm(short)
            m = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Workgroup_Size_X"])
            rows[short].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), m))
            meta[short] = m
    # the plan's placement trials (vof_plan.hip) launch a few PCG passes before the first pyramid: drop everything
    # that starts before the first assembly
    if "k_assemble" in rows:
        t_first = min(r[0] for r in rows["k_assemble"])
        for k in ("k_pcg_pass_a", "k_pcg_pass_b", "k_pcg_fused"):
            if k in rows:
                rows[k] = [r for r in rows[k] if r[0] > t_first]
    per_level = {"k_pcg_fused": 3 * liters * cgiters, "k_pcg_pass_a": 3 * liters * cgiters, "k_pcg_pass_b": 3 * liters * cgiters,
                 "k_assemble": 3 * liters, "k_flow_update": 3 * liters}
    # levels of at most 6144 pixels are solved by k_pcg_solve_small (one launch per solve, flow update included)
    def lev_w(lev):
        return int(size * 0.5 ** (kiters - 1 - lev) + 0.5)
    small = [lev for lev in range(kiters) if lev_w(lev) ** 2 <= 6144]
    # mid-size levels are solved by k_pcg_solve_mid (pcg_persist.hip: one launch per solve, flow update included) unless
    # OCTANE_TUNE_PERSIST=0: the rule of pcg_mid_config on a 256-CU device, levels of at most 2 Mi pixels
    def is_mid(wl):
        if os.environ.get("OCTANE_TUNE_PERSIST", "1") == "0" or wl * wl > (2 << 20):
            return False
        gx = (wl + 63) // 64
        return any(gx * ((wl + 8 * P - 1) // (8 * P)) <= 256 for P in (4, 6, 8, 10, 12, 14, 16))
    mid = [lev for lev in range(kiters) if lev not in small and is_mid(lev_w(lev))]
    two_pass_levels = {k: ([l for l in range(kiters) if l not in small and l not in mid] if k != "k_assemble" else list(range(kiters)))
                       for k in per_level}
    lines = [f"# rocprofv3 kernel-trace summary: bench.py, {size}x{size}, kiters={kiters} liters={liters} cgiters={cgiters}", "",
             "Source: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py ...` on one MI355X;",
             "durations are End-Start of each dispatch in ns, averaged over every pyramid in the run (warm-up included).", "",
             "| kernel | VGPR | AGPR | SGPR | LDS B | scratch | level (size) | launches | mean us | min us | max us | grid (threads) |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, n_per_level in per_level.items():
        if k not in rows:
            continue
        v = sorted(rows[k])
        levs = two_pass_levels[k]
        per_pyr = n_per_level * len(levs)
        npyr = len(v) // per_pyr
        if npyr == 0:          # only the handful of launches of the plan's placement probe
            continue
        for li, lev in enumerate(levs):
            sel = []
            for p in range(npyr):
                sel += v[p * per_pyr + li * n_per_level: p * per_pyr + (li + 1) * n_per_level]
            durs = [s[1] for s in sel]
            f = 0.5 ** (kiters - 1 - lev)
            lw = int(size * f + 0.5)
            m = sel[-1][3]         # template instances differ per level (tile height, q stored or recomputed)
            lines.append(f"| {k} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} | {lev} ({lw}x{lw}) | {len(durs)} | "
                         f"{sum(durs) / len(durs) / 1e3:.2f} | {min(durs) / 1e3:.2f} | {max(durs) / 1e3:.2f} | {sel[0][2]} |")
    # the persistent mid-level solves: one launch per solve, told apart by their grid (workgroups = sub-domains of the level)
    mids = defaultdict(list)
    for k, v in rows.items():
        if k.startswith("k_pcg_solve_mid"):
            for r in v:
                mids[(r[2] // 512, k)].append(r)
    if mids:
        lines += ["", f"| persistent solve (one launch = {cgiters} PCG iterations + flow update) | workgroups | VGPR | LDS B | scratch | launches | mean us | us per iteration |",
                  "|---|---|---|---|---|---|---|---|"]
        for (g, k), v in sorted(mids.items()):
            durs = [r[1] for r in v]
            m = v[-1][3]
            lines.append(f"| {k} | {g} | {m[0]} | {m[3]} | {m[4]} | {len(durs)} | {sum(durs) / len(durs) / 1e3:.1f} | {sum(durs) / len(durs) / 1e3 / max(1, cgiters):.2f} |")
    lines += ["", "| other kernels | launches | mean us | total ms |", "|---|---|---|---|"]
    for k, v in sorted(rows.items()):
        if k in per_level or k.startswith("k_pcg_solve_mid"):
            continue
        durs = [s[1] for s in v]
        lines.append(f"| {k} | {len(durs)} | {sum(durs) / len(durs) / 1e3:.2f} | {sum(durs) / 1e6:.3f} |")
    fin = {k: None for k in ("k_pcg_fused", "k_pcg_pass_a", "k_pcg_pass_b")}
    for k in fin:
        if k in rows and len(rows[k]) >= per_level[k] * len(two_pass_levels[k]):
            v = sorted(rows[k]); n = per_level[k]; nl = len(two_pass_levels[k]); per_pyr = n * nl; npyr = len(v) // per_pyr
            d_ = []
            for p in range(npyr):
                d_ += [s[1] for s in v[p * per_pyr + (nl - 1) * n: (p + 1) * per_pyr]]
            fin[k] = sum(d_) / len(d_)
    px = size * size
    if fin["k_pcg_fused"]:
        f_ns = fin["k_pcg_fused"]
        bpp = 184 / 3 if qform else 232 / 3
        what = ("fused PCG iteration, q recomputed: 61.33 B/px (64 on average over odd / even launches: 52 / 76; 56 in the first GNC step, a "
                if qform else
                "fused PCG iteration: 77.33 B/px (80 on average over odd / even launches: 68 / 92; 72 in the first GNC step, a ")
        lines += ["", "## Finest level against the HBM roofline (algorithmic bytes, DESIGN.md)", "",
                  f"* {what}"
                  f"third of the launches) x {px} px = {bpp * px / 1e9:.3f} GB per launch / {f_ns / 1e3:.1f} us = "
                  f"**{bpp * px / f_ns:.0f} GB/s** ({bpp * px / f_ns / 80:.1f} % of 8 TB/s)",
                  f"* the same iteration at SURVEY 8(d)'s 116 B/px (pass A + pass B with seven coefficient planes): "
                  f"{116 * px / f_ns:.0f} GB/s ({116 * px / f_ns / 80:.1f} % of 8 TB/s)"]
    elif fin["k_pcg_pass_a"] and fin["k_pcg_pass_b"]:
        lines += ["", "## Finest level against the HBM roofline (algorithmic bytes, DESIGN.md)", "",
                  f"* pass A: 49.33 B/px (52; 44 in the first GNC step, a third of the launches) x {px} px = {148 / 3 * px / 1e9:.3f} GB "
                  f"per launch / {fin['k_pcg_pass_a'] / 1e3:.1f} us = **{148 / 3 * px / fin['k_pcg_pass_a']:.0f} GB/s** "
                  f"({148 / 3 * px / fin['k_pcg_pass_a'] / 80:.1f} % of 8 TB/s)",
                  f"* pass B: 56 B/px x {px} px = {56 * px / 1e9:.3f} GB per launch / {fin['k_pcg_pass_b'] / 1e3:.1f} us = "
                  f"**{56 * px / fin['k_pcg_pass_b']:.0f} GB/s** ({56 * px / fin['k_pcg_pass_b'] / 80:.1f} % of 8 TB/s)",
                  f"* one PCG iteration at SURVEY 8(d)'s 116 B/px: {116 * px / (fin['k_pcg_pass_a'] + fin['k_pcg_pass_b']):.0f} GB/s "
                  f"({116 * px / (fin['k_pcg_pass_a'] + fin['k_pcg_pass_b']) / 80:.1f} % of 8 TB/s)"]
    lines += pmc_section(d, size, kiters, per_level, two_pass_levels, qform, qname)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
