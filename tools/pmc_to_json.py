#!/usr/bin/env python
"""Condense a tools/gpu_profile.sh PMC summary into the JSON record bench.py quotes
(profiles/rNN_pmc.json): HBM bytes per launch of lh_encode_kernel with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of wide coalesced reads: doubled; WRITE_SIZE as
reported; both are in KiB), VALU utilisation (a wave64 VALU instruction occupies its SIMD-32 for two
cycles; 1024 SIMDs) and the share of wave cycles with an instruction in flight.
usage: pmc_to_json.py <summ_pmc.txt> <summ_kernel_stats.txt> "<workload>" """
import datetime
import json
import subprocess
import sys


def workload_streams(workload):
    w = workload.split()
    return w[w.index("--streams") + 1] if "--streams" in w else 1024


def main(pmc, stats, workload):
    v = {}
    others = {}                 # the split pipeline's analysis kernels: counters per kernel
    kernel_name = "lh_encode_kernel"
    for line in open(pmc):
        p = line.split()
        if len(p) >= 5 and p[0].startswith("lh_encode"):
            v[p[1]] = (int(p[2]), float(p[3]), float(p[4]))
            kernel_name = p[0]
        elif len(p) >= 5 and p[0].startswith("lh_") and not p[0].startswith("lh_summary"):
            others.setdefault(p[0], {})[p[1]] = float(p[4])
    launches = max([x[0] for x in v.values()] + [1])

    def per(name):
        return v[name][2] if name in v else None
    fetch, write = per("FETCH_SIZE"), per("WRITE_SIZE")
    out = {"date": datetime.date.today().isoformat(), "workload": workload, "kernel": kernel_name,
           "launches_profiled": launches}
    try:
        out["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        out["commit"] = None    # (no git on the GPU box: the digest below is what ties the record to its sources)
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    out["csrc_sha256"] = bench.csrc_digest()
    if fetch is not None and write is not None:
        # (all kernels of a launch: the analysis kernels' traffic is part of the launch's)
        ofetch = sum(o.get("FETCH_SIZE", 0.0) for o in others.values())
        owrite = sum(o.get("WRITE_SIZE", 0.0) for o in others.values())
        out["hbm_bytes_per_launch"] = int((fetch + ofetch) * 1024 * 2 + (write + owrite) * 1024)
        out["hbm_bytes_per_launch_encode_kernel"] = int(fetch * 1024 * 2 + write * 1024)
        out["fetch_kib_reported"] = fetch + ofetch
        out["write_kib_reported"] = write + owrite
    for name in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
                 "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"):
        if name in v:
            out[name] = per(name)
    if "SQ_ACTIVE_INST_ANY" in v and "SQ_WAVE_CYCLES" in v:
        out["issue_active_frac"] = round(per("SQ_ACTIVE_INST_ANY") / per("SQ_WAVE_CYCLES"), 4)
    for line in open(stats):
        if line.startswith("{"):
            try:
                out["frames_per_launch"] = json.loads(line)["roofline"]["frames_per_launch"]
            except (ValueError, KeyError):
                pass
        if "lh_encode_kernel" in line and "," in line:
            f = [x.strip('"') for x in line.split(",")]
            try:
                out["kernel_avg_ns"] = float(f[3])
            except (ValueError, IndexError):
                pass
        elif line.startswith('"lh_') and "," in line:
            f = [x.strip('"') for x in line.split(",")]
            try:
                others.setdefault(f[0], {})["avg_ns"] = float(f[3])
            except (ValueError, IndexError):
                pass
    for name in ("SQ_INSTS_VMEM", "SQ_INSTS_FLAT", "SQ_INSTS_SMEM", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE", "GRBM_COUNT"):
        if name in v:
            out[name] = per(name)
    if "SQ_INSTS_VALU" in v and "kernel_avg_ns" in out and "SQ_WAVE_CYCLES" in v:
        # The clock the kernel ran at, from the counters themselves.  GRBM_GUI_ACTIVE counts the cycles the graphics
        # engine is busy (one counter per XCD, summed: / 8): for a lone kernel that is its length in shader cycles.
        # Without that pass: SQ_BUSY_CYCLES (cycles a shader engine holds waves, summed over 32 engines) -- a lower
        # bound, an engine is idle once its last stream is through.  (Round 3/4's first records took 4 x SQ_WAVE_CYCLES /
        # waves for the kernel's length: that is the MEAN residence of a wave, 10 % short of the launch at 1024 unequal
        # streams -- it is reported as wave_residency now.)
        waves = 2 * int(workload_streams(workload))
        cycles = None
        if "GRBM_GUI_ACTIVE" in v:
            for div in (8.0, 1.0, 32.0):
                ghz = per("GRBM_GUI_ACTIVE") / div / out["kernel_avg_ns"]
                if 1.2 < ghz < 2.7:
                    cycles = per("GRBM_GUI_ACTIVE") / div
                    out["clock_from"] = "GRBM_GUI_ACTIVE / %d" % div
                    break
        if cycles is None and "SQ_BUSY_CYCLES" in v:
            cycles = per("SQ_BUSY_CYCLES") / 32.0
            out["clock_from"] = "SQ_BUSY_CYCLES / 32 (lower bound)"
        if cycles is None:
            cycles = 4.0 * per("SQ_WAVE_CYCLES") / waves
            out["clock_from"] = "4 x SQ_WAVE_CYCLES / waves (lower bound)"
        out["shader_cycles_per_launch"] = round(cycles)
        out["shader_clock_ghz"] = round(cycles / out["kernel_avg_ns"], 3)
        # mean share of the launch a wave is resident (a launch ends with its last stream)
        out["wave_residency"] = round(4.0 * per("SQ_WAVE_CYCLES") / waves / cycles, 4)
        # a wave64 VALU instruction occupies its SIMD-32 for two cycles; 1024 SIMDs
        out["valu_frac"] = round(per("SQ_INSTS_VALU") * 2 / (1024 * cycles), 4)
        insts = sum(per(n) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH") if n in v)
        insts += per("SQ_INSTS_VMEM") if "SQ_INSTS_VMEM" in v else (per("SQ_INSTS_FLAT") or 0)
        insts += per("SQ_INSTS_SMEM") or 0
        out["wave_insts_per_launch"] = round(insts)
        out["insts_per_cycle_per_simd"] = round(insts / (1024 * cycles), 4)
        # the fraction that binds: what a SIMD issued against what tools/ubench/lat2.hip measured it can issue for two
        # waves (2 / 4.63 cycles = 0.43 instructions per cycle; four waves: 0.61, eight: 0.76)
        out["issue_ceiling_2waves"] = 0.43
        out["issue_frac"] = round(out["insts_per_cycle_per_simd"] / 0.43, 4)
    if "hbm_bytes_per_launch" in out and out.get("frames_per_launch"):
        out["hbm_bytes_per_frame"] = round(out["hbm_bytes_per_launch"] / out["frames_per_launch"], 1)
    if out.get("wave_insts_per_launch") and out.get("frames_per_launch"):
        out["wave_insts_per_frame"] = round(out["wave_insts_per_launch"] / out["frames_per_launch"], 1)
        out["shader_cycles_per_frame"] = round(out["shader_cycles_per_launch"] * int(workload_streams(workload))
                                               / out["frames_per_launch"], 1)
    if others:
        ks = {}
        for name, o in sorted(others.items()):
            k = {"avg_ns": o.get("avg_ns")}
            insts = sum(o.get(n, 0.0) for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM",
                                                "SQ_INSTS_SMEM"))
            k["wave_insts_per_launch"] = round(insts)
            if o.get("avg_ns") and out.get("shader_clock_ghz"):
                cyc = o["avg_ns"] * out["shader_clock_ghz"]
                k["insts_per_cycle_per_simd"] = round(insts / (1024 * cyc), 4)
            if "SQ_WAVE_CYCLES" in o and o["SQ_WAVE_CYCLES"] > 0:
                k["wait_frac"] = round(o.get("SQ_WAIT_ANY", 0.0) / o["SQ_WAVE_CYCLES"], 4)
            if "SQ_ACTIVE_INST_LDS" in o:
                k["lds_busy_quad_cycles"] = round(o["SQ_ACTIVE_INST_LDS"])
                k["lds_bank_conflict_quad_cycles"] = round(o.get("SQ_LDS_BANK_CONFLICT", 0.0))
            if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
                k["hbm_bytes_per_launch"] = int(o["FETCH_SIZE"] * 2048 + o["WRITE_SIZE"] * 1024)
                if out.get("frames_per_launch"):
                    k["hbm_bytes_per_frame"] = round(k["hbm_bytes_per_launch"] / out["frames_per_launch"], 1)
            ks[name] = k
        out["analysis_kernels"] = ks
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
