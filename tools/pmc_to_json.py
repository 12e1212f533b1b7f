#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE and WRITE_SIZE summaries written by
tools/rocprof_summary.py --pmc (separate rocprofv3 passes).

traffic = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): the x2 on FETCH_SIZE is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section), calibrated there on 16-byte-per-lane streaming
reads; rk3dc_fused reads 16 B per lane, rk3dq_fused and the 2-D kernels read 8 B per lane (their read side is
an upper estimate).  WRITE_SIZE is taken as reported (it matches the algorithmic write bytes of
these kernels to within 2 %)."""
import json
import re
import sys

WORKLOADS = {"rk3dq_fused": "c5 512x512x512", "rk3dc_fused": "c5 512x512x512", "rk3d_fused": "c5 512x512x512", "rk3d_collide": "c5 512x512x512",
             "rk3d_phase_field": "c5 512x512x512", "rk2d_fused": None, "rk2d_fused_tracer": None, "sc2d_fused": "c3 2048x2048"}


def parse(path, counter):
    """{kernel name as printed: (launches, average per launch)}.  Instances whose printed (truncated) names coincide are one kernel
    launched over parts of the lattice -- the tracer step: the tile rows at the lattice's ends and the rest -- and are folded into ONE
    entry per time step: the sum over the instances of launches x average, divided by the smallest launch count (= the steps)."""
    rows = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+%s\s+(\d+)\s+([0-9.]+)\s*$" % counter, line.rstrip())
        if m:
            rows.setdefault(m.group(1).strip(), []).append((int(m.group(2)), float(m.group(3))))
    out = {}
    for name, inst in rows.items():
        steps = min(n for n, _ in inst)
        out[name] = (steps, sum(n * v for n, v in inst) / steps)
    return out


def short(name):
    m = re.search(r"(rk3d[cq]?_\w+|rk2d_\w+|sc2d_\w+)", name)
    return m.group(1) if m else name


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    collect(fetch, write, kernels, "")
    # further passes of `bench.py --c5-state STATE --no-secondary`:  STATE fetch.txt write.txt  -> keys rk3dq_fused[STATE]
    rest = sys.argv[3:]
    while len(rest) >= 3:
        collect(parse(rest[1], "FETCH_SIZE"), parse(rest[2], "WRITE_SIZE"), kernels, "[%s]" % rest[0], only="rk3dq_fused")
        rest = rest[3:]
    print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                                "`python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5-legs` (and `--c5-state S --no-secondary` for the "
                                "[S] entries), MI355X (tools/profile_round.sh)",
                      "correction": "traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> B); x2 = the guide's gfx950 FETCH_SIZE correction "
                                    "(calibrated on 16-B lanes; exact for rk3dc_fused, upper estimate for the 8-B-lane 2-D kernels)",
                      "kernels": kernels}, indent=1))


def collect(fetch, write, kernels, suffix, only=None):
    for name, (n, f) in fetch.items():
        k = short(name)
        if k not in WORKLOADS or (only and k != only):
            continue
        w = write.get(name, (0, 0.0))[1]
        rec = {"launches_profiled": n, "fetch_size_kb": f, "write_size_kb": w,
               "traffic_bytes_per_launch": (2.0 * f + w) * 1024.0, "workload": WORKLOADS[k]}
        if k == "rk2d_fused_tracer":      # the tracer step of c4 (one launch since round 4; earlier: three of rk2d_fused<.., true, ..>)
            k = "rk2d_fused"
            name = name.replace("rk2d_fused_tracer<", "rk2d_fused<true, true, ")
        if k == "rk2d_fused":             # c2 (no tracer) and c4 (tracer) are different template instances
            rec["workload"] = "c4 2048x2048" if re.search(r"rk2d_fused<(true|false), true", name) else "c2 1024x1024"
            k = "rk2d_fused" if rec["workload"].startswith("c2") else "rk2d_fused[tracer]"
        if k == "sc2d_fused" and re.search(r"sc2d_fused<false[,>]", name):      # SRT instance = the 128 x 128 droplet of configs[0]
            rec["workload"] = "c1 128x128"
            k = "sc2d_fused[c1]"
        m3 = re.search(r"rk3d[cq]?_fused<([^>]*)>", name)
        if m3:                            # <.., FIRST, MRT>: first-step instances carry one launch; SRT is the secondary entry
            targs = [t.strip() for t in m3.group(1).split(",")]
            if targs[-2] == "true":
                continue
            if targs[-1] == "false":
                k += "[SRT]"
        k += suffix
        if k in kernels and kernels[k]["launches_profiled"] >= n:
            continue                      # e.g. the first-step instantiation of the 3-D kernel: one launch only
        kernels[k] = rec


if __name__ == "__main__":
    main()
