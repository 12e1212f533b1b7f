#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE and WRITE_SIZE summaries written by
tools/rocprof_summary.py --pmc (separate rocprofv3 passes).

traffic = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE (KB -> bytes).  The factors are MEASURED in the same counter passes:
lbmpm_hbm_stream_test (which bench.py runs) launches calibration kernels that move a known number of bytes with one access width each
(csrc/lbmpm_common.hip: calib_read / calib_copy at 8 and 16 bytes per lane, calib_pull19_b64 = the 19-plane access shape of
rk3dq_fused), and a lattice kernel takes the factor of its own width: rk3dq_fused (global_load_dwordx2) calib_pull19_b64's, the 2-D
kernels and rk3dc_fused (16-byte population pairs) calib_copy_b128's.  Profiles without calibration rows (rounds 1 - 4) fall back on the
guide's gfx950 correction, x 2 on FETCH_SIZE (/opt/skills/guides/MI355X_MICROARCH.md, HBM section), and WRITE_SIZE as reported."""
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


GIB = float(1 << 30)
# true bytes (read, written) of one launch of each calibration kernel (csrc/lbmpm_common.hip)
CALIB = {"calib_read_b64": (GIB, 0.), "calib_copy_b64": (GIB, GIB), "calib_read_b128": (GIB, 0.), "calib_copy_b128": (GIB, GIB),
         "calib_pull19_b64": (19. * (48 << 20), 19. * (48 << 20))}
# which calibration a lattice kernel's accesses look like
WIDTH_OF = {"rk3dq_fused": "calib_pull19_b64", "rk3dc_fused": "calib_copy_b128", "rk3d_fused": "calib_copy_b64", "rk3d_collide": "calib_copy_b64",
            "rk3d_phase_field": "calib_read_b64", "rk2d_fused": "calib_copy_b128", "rk2d_fused_tracer": "calib_copy_b128", "sc2d_fused": "calib_copy_b128"}


def calibration(fetch, write):
    """{calib kernel: {"true_read_bytes", "fetch_size_kb", "fetch_factor" = true bytes / counted bytes, the same for writes}}"""
    out = {}
    for name, (_n, f) in fetch.items():
        m = re.search(r"(calib_\w+)", name)
        if not m or m.group(1) not in CALIB:
            continue
        rd, wr = CALIB[m.group(1)]
        w = write.get(name, (0, 0.0))[1]
        out[m.group(1)] = {"true_read_bytes": rd, "fetch_size_kb": f, "fetch_factor": round(rd / (f * 1024.), 4) if f > 0 else None,
                           "true_write_bytes": wr, "write_size_kb": w, "write_factor": round(wr / (w * 1024.), 4) if w > 0 and wr > 0 else None}
    return out


def short(name):
    m = re.search(r"(rk3d[cq]?_\w+|rk2d_\w+|sc2d_\w+)", name)
    return m.group(1) if m else name


CAL = {}


def main():
    fetch, write = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    global CAL
    CAL = calibration(fetch, write)
    collect(fetch, write, kernels, "")
    # further passes of `bench.py --c5-state STATE --no-secondary`:  STATE fetch.txt write.txt  -> keys rk3dq_fused[STATE]
    rest = sys.argv[3:]
    while len(rest) >= 3:
        collect(parse(rest[1], "FETCH_SIZE"), parse(rest[2], "WRITE_SIZE"), kernels, "" if rest[0] == "-" else "[%s]" % rest[0], only="rk3dq_fused")
        rest = rest[3:]
    print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                                "`python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-c5-legs` (and `--c5-state S --no-secondary` for the "
                                "[S] entries), MI355X (tools/profile_round.sh)",
                      "correction": ("traffic = fetch_factor * FETCH_SIZE + write_factor * WRITE_SIZE (KB -> B), the factors measured in the same passes on "
                                     "calibration kernels of the lattice kernel's own access width (`calibration`; every kernel entry names the one it took)")
                                    if CAL else "traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> B); x2 = the guide's gfx950 FETCH_SIZE correction (no calibration rows in this profile)",
                      "calibration": CAL,
                      "kernels": kernels}, indent=1))


def collect(fetch, write, kernels, suffix, only=None):
    for name, (n, f) in fetch.items():
        k = short(name)
        if k not in WORKLOADS or (only and k != only):
            continue
        w = write.get(name, (0, 0.0))[1]
        cal = CAL.get(WIDTH_OF.get(k, ""), {})
        ff, wf = cal.get("fetch_factor") or 2.0, cal.get("write_factor") or 1.0
        rec = {"launches_profiled": n, "fetch_size_kb": f, "write_size_kb": w, "fetch_factor": ff, "write_factor": wf,
               "factors_from": WIDTH_OF.get(k) if cal else "guide (x 2 on FETCH_SIZE)",
               "traffic_bytes_per_launch": (ff * f + wf * w) * 1024.0, "workload": WORKLOADS[k]}
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
            # rk3dq_fused<FIRST, MRT, RAGGED> (round 6; <FIRST, MRT> before), rk3dc_fused<TY, FIRST, MRT>, rk3d_fused<TX, TY, FIRST, MRT>
            first, mrt = (targs[0], targs[1]) if "rk3dq_fused" in name else (targs[-2], targs[-1])
            if first == "true":
                continue
            if mrt == "false":
                k += "[SRT]"
        k += suffix
        if k in kernels and kernels[k]["launches_profiled"] >= n:
            continue                      # e.g. the first-step instantiation of the 3-D kernel: one launch only
        kernels[k] = rec


if __name__ == "__main__":
    main()
