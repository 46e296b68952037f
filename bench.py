#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native MP3 encode inner loop.

Metric (BASELINE.json): encoded audio seconds per wall-clock second (x real-time) at 44.1 kHz
stereo CBR 128 kb/s.  A "step" is one pass of the hot path (psycho-acoustics + polyphase/MDCT +
quantisation loop -> side-info payload in HBM) over one batch of synthetic streams whose PCM is
already resident in HBM (BASELINE config[1]: batch = 1024 streams x 60 s per GPU).

Multi-GPU: streams are independent, so a batch of N x 1024 streams is sharded
statically, rank r owning the contiguous block lamehip.shard_streams(N * 1024, N, r); there is no
collective anywhere on the data path and no RCCL.  `python bench.py --gpus N` spawns the N ranks
itself (one process per device, a directory of marker files as the barrier); when a launcher has
already created the ranks (python -m torch.distributed.run ... bench.py --gpus N: WORLD_SIZE / RANK /
LOCAL_RANK in the environment) the ranks only use torch.distributed's gloo backend on the CPU for
the barrier and the maximum of the times.  Either way rank 0 prints ONE JSON line with the
whole-job aggregate.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s
CHECKED_STREAMS = 64                # streams whose payload is compared with the CPU oracle after the timed region (rank 0; every
                                    # other rank checks the first and the last stream of its own block)
# what tools/ubench/lat2.hip measured a SIMD can issue per cycle (all instruction classes): for the two waves the encode kernel
# gives it, and for eight
ISSUE_CEILING_2_WAVES = 0.43
ISSUE_CEILING_8_WAVES = 0.76


def alg_bytes_per_frame(sr):
    """PCM in + side-info payload out per frame of the whole path -- 4608 + 5184 B for the two-granule
    frames of MPEG-1 (32 / 44.1 / 48 kHz), 2304 + 2624 B for the one-granule frames of MPEG-2 / 2.5."""
    return 9792 if sr >= 32000 else 4928


def alg_bytes_by_kernel(sr):
    """The same per kernel of the split pipeline (csrc/lh_device.h): the analysis kernels read the PCM and leave the small
    record and the long-block masking (576 + 6144 B; the short-block masking only for granules that switch), the sub-band
    kernel reads the PCM and leaves the spectra (9216 B), the encode kernel reads both and leaves the payload."""
    g = 2 if sr >= 32000 else 1
    pcm, payload = 2304 * g, 2592 * g if g == 2 else 2624
    small, lng, xr = 576, 6144, 9216
    return {"analysis": pcm + small + lng, "subband": pcm + xr, "encode": small + lng + xr + payload}
CPU_SAMPLE_SECONDS = 20.0           # audio per stream in the CPU leg


def synth_on_device(torch, batch, n, sr, seed0, device, chunk=32, bursts_per_s=3.0):
    """SURVEY.md 8(d)'s synthetic signal (the recipe of tests/helpers.py:synth_stream), generated on the GPU with
    torch's generator (no dataset access): eight partials 220 * 2^(k/2) Hz at 0.5/(k+1) with a 5 Hz vibrato of
    0.1 % and independent phases per channel, band-limited brown noise (a random walk minus its 200-sample moving
    average) at 0.02 of full scale, decaying white-noise bursts (castanet-like: they trigger short blocks)
    bursts_per_s times a second, normalised to 0.8 of full scale.  Returns an int16 tensor [batch, 2, n]."""
    out = torch.empty((batch, 2, n), dtype=torch.int16, device=device)
    t = torch.arange(n, device=device, dtype=torch.float64) / sr
    vib = 1.0 + 0.001 * torch.sin(2 * 3.141592653589793 * 5 * t)
    step = max(int(sr / bursts_per_s), 1)
    blen = min(2000, step)
    env = torch.exp(-torch.arange(blen, device=device, dtype=torch.float32) / 300.0)
    k200 = 200
    for b0 in range(0, batch, chunk):
        b1 = min(batch, b0 + chunk)
        g = torch.Generator(device=device)
        g.manual_seed(0x4C414D45 + seed0 + b0)
        x = torch.zeros((b1 - b0, 2, n), device=device)
        for k in range(8):
            f = 220.0 * 2 ** (k / 2.0)
            ph = torch.rand((b1 - b0, 2, 1), generator=g, device=device, dtype=torch.float64) * 6.283185307179586
            x += ((0.5 / (k + 1)) * torch.sin(6.283185307179586 * f * t * vib + ph)).float()
        brown = torch.cumsum(torch.randn((b1 - b0, 2, n), generator=g, device=device, dtype=torch.float64), dim=2)
        padded = torch.cat([brown[:, :, :1].expand(-1, -1, k200), brown], dim=2)
        cs = torch.cumsum(padded, dim=2)
        brown = brown - (cs[:, :, k200:] - cs[:, :, :-k200]) / k200
        x += (0.02 * brown / (brown.abs().amax(dim=(1, 2), keepdim=True) + 1e-9)).float()
        del brown, padded, cs
        for s in range(step // 2, n - blen, step):
            x[:, :, s:s + blen] += 0.6 * env * torch.randn((b1 - b0, 2, blen), generator=g, device=device)
        x = x / x.abs().amax(dim=(1, 2), keepdim=True) * (0.8 * 32767)
        out[b0:b1] = x.to(torch.int16)
    return out


SIGNAL = ("SURVEY 8(d) recipe on the device: 8 partials with 5 Hz vibrato, band-limited brown noise, noise bursts "
          "%.0f/s, seeded per stream")


# ---------------------------------------------------------------------------------------------
# CPU leg: the compiled reference (oracle/_ref) on the host's physical cores, one process per core,
# each encoding one of the SAME streams the GPU leg encodes (their first 20 s).
def _cpu_worker(job):
    """One process of the CPU leg (spawned: no GPU context): encode stream `idx' of the saved PCM
    over and over for about `budget' seconds; returns (audio seconds encoded, elapsed)."""
    path, idx, sr, brate, vbr_q, abr, budget, kind, vbr_mode = job
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    pcm = np.load(path, mmap_mode="r")[idx]
    pcm = np.ascontiguousarray(pcm)
    seconds = pcm.shape[1] / float(sr)
    if kind == "reference":
        ref = helpers.Reference()

        def run():
            ref.encode(pcm, sr, brate, vbr_q=vbr_q, abr=abr, vbr_mode=vbr_mode)
    else:
        import lamehip
        orc = helpers.Oracle()
        enc = lamehip.Encoder(sr, brate, require_device=False, vbr_q=vbr_q, abr=abr, vbr_mode=vbr_mode)
        cfg, tab = enc.config(), enc.tables()

        def run():
            orc.encode_frames(cfg, tab, pcm)
    run()                       # first touch (page-in) outside the clock
    t0 = time.perf_counter()
    done = 0
    while True:
        run()
        done += 1
        if time.perf_counter() - t0 >= budget:
            break
    return done * seconds, time.perf_counter() - t0


def _physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _cpu_allowance():
    """CPUs this process may actually use at once: the affinity mask and the cgroup's CPU quota
    (a container on a big host is often limited to a fraction of its cores)."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return allowed, quota


def _cpu_budget():
    allowed, quota = _cpu_allowance()
    return max(1, int(min(allowed, quota) if quota else allowed))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(host_pcm, sr, brate, vbr_q=None, abr=None, budget=10.0, vbr_mode=4):
    """P concurrent encodes, P = physical cores (at most as many as there are saved streams),
    plus one process alone for the per-core figure."""
    import multiprocessing as mp
    import numpy as np
    import helpers
    kind = "reference" if helpers.have_reference() else "port"
    cores = _physical_cores()
    allowed, quota = _cpu_allowance()
    usable = min(cores, allowed, int(quota) if quota and quota >= 1 else cores)
    procs = max(1, min(usable, host_pcm.shape[0]))
    tmp = tempfile.NamedTemporaryFile(suffix=".npy", delete=False)
    tmp.close()
    np.save(tmp.name, host_pcm)
    try:
        ctx = mp.get_context("spawn")
        with ctx.Pool(1) as pool:
            a, t = pool.map(_cpu_worker, [(tmp.name, 0, sr, brate, vbr_q, abr, min(budget, 4.0), kind, vbr_mode)])[0]
        single = a / t
        with ctx.Pool(procs) as pool:
            res = pool.map(_cpu_worker, [(tmp.name, i, sr, brate, vbr_q, abr, budget, kind, vbr_mode) for i in range(procs)],
                           chunksize=1)
    finally:
        os.unlink(tmp.name)
    aggregate = sum(a / t for a, t in res)
    what = ("ABR %d" % abr) if abr is not None else ("CBR %d" % brate) if vbr_q is None else ("VBR -V%d%s" % (vbr_q, " --vbr-old" if vbr_mode == 2 else ""))
    return {"value": round(aggregate, 1), "unit": "x real-time", "cores": procs, "kind": kind,
            "per_core": round(aggregate / procs, 2), "one_process_alone": round(single, 2),
            # what the whole host would give if every physical core ran at the measured per-core rate (the container is
            # limited to `cores' of them): an extrapolation, labelled as such
            "host_physical_cores_estimate": round(aggregate / procs * cores, 1),
            "physical_cores": cores, "logical_cpus": os.cpu_count(), "cpus_in_affinity_mask": allowed,
            "cgroup_cpu_quota": quota, "cpu_model": _cpu_model(),
            "sample": "%d processes at once, each encoding the first %.0f s of one of the GPU leg's streams "
                      "(%s, %d Hz) repeatedly for %.0f s" % (procs, host_pcm.shape[2] / float(sr), what, sr, budget)}


# ---------------------------------------------------------------------------------------------
class FileRendezvous:
    """Barrier and maximum over ranks through marker files in a directory (ranks spawned by bench.py
    itself: no torch.distributed, no RCCL)."""

    def __init__(self, path, rank, world):
        self.path, self.rank, self.world, self.n = path, rank, world, 0

    def _all(self, tag, value):
        self.n += 1
        name = os.path.join(self.path, "%s.%d" % (tag, self.n))
        with open("%s.%d.tmp" % (name, self.rank), "w") as f:
            json.dump([float(x) for x in value] if isinstance(value, (list, tuple)) else float(value), f)
        os.rename("%s.%d.tmp" % (name, self.rank), "%s.%d" % (name, self.rank))
        vals = []
        t0 = time.time()
        for r in range(self.world):
            while not os.path.exists("%s.%d" % (name, r)):
                if time.time() - t0 > 1800:
                    raise RuntimeError("rank %d never reached %s" % (r, tag))
                time.sleep(0.0005)
            vals.append(json.load(open("%s.%d" % (name, r))))
        return vals

    def barrier(self):
        self._all("barrier", 0.0)

    def max(self, value):
        return max(self._all("max", value))

    def gather(self, values):
        """every rank's list of floats, in rank order (all ranks get all of them)"""
        return self._all("gather", list(values))

    def close(self):
        pass


class GlooRendezvous:
    """The same through torch.distributed on the CPU (ranks created by a launcher)."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist = dist

    def barrier(self):
        self.dist.barrier()

    def max(self, value):
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, values):
        import torch
        mine = torch.tensor(list(values), dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, mine)         # (gloo, on the CPU: a handful of timing scalars, never stream data)
        return [[float(x) for x in t] for t in out]

    def close(self):
        self.dist.destroy_process_group()


class Alone:
    def barrier(self):
        pass

    def max(self, value):
        return value

    def gather(self, values):
        return [list(values)]

    def close(self):
        pass


def spawn_ranks(args, argv):
    """python bench.py --gpus N without a launcher: start the N ranks, pass their lines through."""
    rdv = tempfile.mkdtemp(prefix="lamehip_bench_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, WORLD_SIZE=str(args.gpus), RANK=str(r), LOCAL_RANK=str(r), LAMEHIP_BENCH_RDV=rdv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    for f in os.listdir(rdv):
        os.unlink(os.path.join(rdv, f))
    os.rmdir(rdv)
    return rc


def csrc_digest():
    """SHA-256 over the device / host sources of the library (csrc/*.h, *.hip, *.c, *.cpp, Makefile, in name
    order): what a recorded profile is tied to (there is no git on the GPU box)."""
    import hashlib
    d = os.path.join(ROOT, "deprecated-lame-mirror_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".c", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def recorded(name):
    """A measurement that needs its own profiler pass (PMC counters), recorded under profiles/ with the digest
    of the sources it was taken from; bench.py quotes it only while the sources are the same."""
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            rec = json.load(open(path))
        except ValueError:
            return None
        rec["stale"] = rec.get("csrc_sha256") != csrc_digest()
        return rec
    return None


def _check_worker(job):
    """One process of the post-check (spawned: no GPU context): the oracle's payload of one saved stream against the
    bytes the GPU left; returns None or a description of the first difference."""
    pcm_path, idx, stream, frames_path, enc_args = job
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    import lamehip
    from lamehip.types import LhFrameOut, struct_diff
    import ctypes as C
    pcm = np.ascontiguousarray(np.load(pcm_path, mmap_mode="r")[idx])
    enc = lamehip.Encoder(*enc_args[0], require_device=False, **enc_args[1])
    want = helpers.Oracle().encode_frames(enc.config(), enc.tables(), pcm)
    got = open(frames_path, "rb").read()
    fsz = C.sizeof(LhFrameOut)
    if len(got) != len(want) * fsz:
        return "stream %d: %d frames from the GPU, %d from the oracle" % (stream, len(got) // fsz, len(want))
    for f in range(len(want)):
        if got[f * fsz:(f + 1) * fsz] != bytes(want[f]):
            d = struct_diff(want[f], LhFrameOut.from_buffer_copy(got[f * fsz:(f + 1) * fsz]))
            if d:
                return "stream %d frame %d differs from the oracle: %r" % (stream, f, d[:4])
    return None


def check_against_oracle(batch, enc_args, host_streams, which, procs=1):
    """Payload of some of the bench's streams, every frame, against the CPU oracle (test infrastructure, used here only
    as the checker, after the timed region), in `procs' processes at once; raises on the first difference."""
    import multiprocessing as mp
    import numpy as np
    tmpd = tempfile.mkdtemp(prefix="lamehip_check_")
    try:
        pcm_path = os.path.join(tmpd, "pcm.npy")
        np.save(pcm_path, np.stack(host_streams))
        jobs = []
        for i, s in enumerate(which):
            fp = os.path.join(tmpd, "frames_%d.bin" % s)
            with open(fp, "wb") as f:
                f.write(bytes(batch.get_frames(s)))
            jobs.append((pcm_path, i, s, fp, enc_args))
        if procs > 1 and len(jobs) > 1:
            with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
                res = pool.map(_check_worker, jobs, chunksize=1)
        else:
            res = [_check_worker(j) for j in jobs]
    finally:
        for f in os.listdir(tmpd):
            os.unlink(os.path.join(tmpd, f))
        os.rmdir(tmpd)
    bad = [r for r in res if r]
    if bad:
        raise SystemExit("bench: " + bad[0])
    return {"streams": len(which), "first_last": [int(which[0]), int(which[-1])], "frames_each": int(batch.frames(which[0])),
            "processes": min(procs, len(jobs)), "result": "identical"}


def pmc_record_for(args):
    """The committed PMC record of the workload this run benches (recorded with tools/r06_collect.sh)."""
    if args.vbr is not None:
        return "r06_pmc_vbrold2.json" if args.vbr_old else "r06_pmc_vbr2.json"
    if args.brate == 320 and args.samplerate == 48000:
        return "r06_pmc_cbr320.json"
    if args.samplerate < 32000:
        return "r06_pmc_lsf.json"
    return "r06_pmc.json"


def roofline_block(frames, sr, kavg_s, parts, pmc_name):
    """HBM roofline of the launch, kernel by kernel: algorithmic bytes over the kernel's average time (HIP events on the
    batch's stream, recorded between the kernels: lamehip_batch_last_kernel_parts_ms).  The block's own achieved / frac are
    the dominant kernel's (the encode kernel: lh_encode_kernel_q of the split pipeline, or the fused lh_encode_kernel);
    `whole_launch' prices the whole path's bytes over the whole launch.  traffic = the recorded PMC figure per
    frame x the frames of this launch, only while the record was taken from these very sources."""
    split = bool(parts) and parts[0]
    whole = frames * alg_bytes_per_frame(sr) / kavg_s / 1e9
    kernels = []
    if split:
        by = alg_bytes_by_kernel(sr)
        names = {"analysis": "lh_attack_kernel + lh_attack_scan_kernel + lh_analysis_kernel", "subband": "lh_subband_kernel",
                 "encode": "lh_encode_kernel_q"}
        for k, ms in zip(("analysis", "subband", "encode"), parts[1]):
            a = frames * by[k] / (ms / 1e3) / 1e9 if ms > 0 else 0.0
            kernels.append({"kernel": names[k], "ms_avg": round(ms, 3), "alg_bytes_per_frame": by[k], "achieved": round(a, 3),
                            "frac": round(a / HBM_PEAK_GBS, 6)})
        # the block's own figure follows SURVEY 8(d) to the letter: the PATH's algorithmic bytes per frame (9 792 B for MPEG-1)
        # x the frames of the launch over the dominant kernel's average time; the bytes each kernel of the split pipeline
        # moves by design (the records between them included) price the entries of `kernels' only
        dom = kernels[2]
        dom_name, dom_ms, dom_bytes = dom["kernel"], dom["ms_avg"], alg_bytes_per_frame(sr)
        achieved = frames * dom_bytes / (dom_ms / 1e3) / 1e9 if dom_ms > 0 else 0.0
    else:
        achieved, dom_name, dom_ms, dom_bytes = whole, "lh_encode_kernel", kavg_s * 1e3, alg_bytes_per_frame(sr)
    pmc = recorded(pmc_name) or {}
    fresh = bool(pmc) and not pmc.get("stale") and pmc.get("hbm_bytes_per_frame")
    ipc = pmc.get("insts_per_cycle_per_simd") if fresh else None
    block = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 6),
             "traffic": int(pmc["hbm_bytes_per_frame"] * frames) if fresh else None,
             # HBM bytes moved (all kernels of the launch) over the path's algorithmic bytes: 1.0 = nothing re-read or spilled
             "traffic_ratio": round(pmc["hbm_bytes_per_frame"] / alg_bytes_per_frame(sr), 2) if fresh else None,
             "traffic_note": ("%.1f KB per frame x the frames of this launch (all kernels of the launch); per-frame figure "
                              "recorded with rocprofv3 --pmc (FETCH_SIZE x 2 + WRITE_SIZE) on %s from these sources (digest %s), "
                              "workload `%s'; bench.py does not run the profiler"
                              % (pmc.get("hbm_bytes_per_frame", 0) / 1e3, pmc.get("date"), pmc.get("csrc_sha256"),
                                 pmc.get("workload"))) if fresh
             else ("stale profile: profiles/%s was recorded from other sources (digest %s, now %s)"
                   % (pmc_name, pmc.get("csrc_sha256"), csrc_digest())) if pmc else "no recorded PMC profile in profiles/",
             "kernel": dom_name, "kernel_ms_avg": round(dom_ms, 3), "alg_bytes_per_frame": dom_bytes,
             "frames_per_launch": frames,
             "kernels": kernels if split else None,
             "whole_launch": {"ms_avg": round(kavg_s * 1e3, 3), "alg_bytes_per_frame": alg_bytes_per_frame(sr),
                              "achieved": round(whole, 3), "frac": round(whole / HBM_PEAK_GBS, 6)},
             "valu_frac": pmc.get("valu_frac") if fresh else None,
             "issue_active_frac": pmc.get("issue_active_frac") if fresh else None,
             # the fraction that binds: instructions the encode kernel's SIMDs issued per cycle (all classes, from the same
             # PMC record, at the clock the counters give) over what tools/ubench/lat2.hip measured a SIMD can issue -- for the
             # two waves that kernel gives it (how good those two waves are) and for eight (how idle the SIMD is)
             "insts_per_cycle_per_simd": ipc,
             "issue_frac": round(ipc / ISSUE_CEILING_2_WAVES, 3) if ipc else None,
             "issue_frac_of_8_wave_ceiling": round(ipc / ISSUE_CEILING_8_WAVES, 3) if ipc else None,
             "wave_insts_per_frame": pmc.get("wave_insts_per_frame") if fresh else None,
             "shader_clock_ghz": pmc.get("shader_clock_ghz") if fresh else None,
             "clock_from": pmc.get("clock_from") if fresh else None,
             # mean share of the launch a wave is resident: a launch ends with its slowest stream
             "wave_residency": pmc.get("wave_residency") if fresh else None,
             "bound_in_practice": "instruction issue of a serial search: one wave issues at most one instruction per "
                                  "~4.6 cycles (tools/ubench/lat2.hip) and a stream offers two chains (its channels), "
                                  "one SIMD's worth; HBM is idle"}
    return block


def mean_parts(parts):
    """[(split, [analysis, subband, encode] ms), ...] of several launches -> one such pair of means"""
    if not parts:
        return None
    return parts[0][0], [sum(p[1][k] for p in parts) / len(parts) for k in range(3)]


def short_run(torch, lamehip, dev, device_index, sr, B, seconds, steps, seed, bursts_per_s, pmc_name, **enc_kw):
    """A small batch of another BASELINE configuration (extras; never the headline value): its own oracle check
    (two streams, every frame) and its own roofline block."""
    n = int(seconds * sr)
    enc = lamehip.Encoder(sr, device=device_index, **enc_kw)
    b = lamehip.Batch(enc, B, n, device=device_index)
    pcm = synth_on_device(torch, B, n, sr, seed, dev, bursts_per_s=bursts_per_s)
    for s in range(B):
        b.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), n)
    which = sorted(set(int(round(k * (B - 1) / 7.0)) for k in range(8))) if B > 1 else [0]
    host_streams = [pcm[s].cpu().numpy() for s in which]
    del pcm
    b.encode(sync=True)
    torch.cuda.synchronize()
    kms, parts = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        b.encode(sync=True)
        kms.append(b.kernel_ms())
        parts.append(b.kernel_parts_ms())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    checked = check_against_oracle(b, ((sr,), dict(enc_kw)), host_streams, which, procs=min(_cpu_budget(), 8))
    assert len(b.pack(0)) > 0
    frames = sum(b.frames(s) for s in range(B))
    b.close()
    enc.close()
    v = B * seconds * steps / dt
    return {"value": round(v, 1), "unit": "x real-time", "per_stream_x_realtime": round(v / B, 2),
            "workload": "batch=%d x %.0f s, %d Hz; %s" % (B, seconds, sr, SIGNAL % bursts_per_s), "steps": steps,
            "checked_against_oracle": checked,
            "roofline": roofline_block(frames, sr, sum(kms) / len(kms) / 1e3, mean_parts(parts), pmc_name)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU")
    ap.add_argument("--seconds", type=float, default=60.0, help="audio seconds per stream")
    ap.add_argument("--samplerate", type=int, default=44100)
    ap.add_argument("--brate", type=int, default=128)
    ap.add_argument("--vbr", type=int, default=None, metavar="Q",
                    help="vbr_mtrh at quality Q (BASELINE config[2] is -V2) instead of CBR; not the default line")
    ap.add_argument("--vbr-old", action="store_true", help="with --vbr Q: the old VBR loop, lame_set_VBR(vbr_rh), instead of vbr_mtrh")
    ap.add_argument("--abr", type=int, default=None, metavar="KBPS", help="ABR at a mean of KBPS instead of CBR")
    ap.add_argument("--bursts", type=float, default=3.0, help="noise bursts per second in the synthetic signal (config[4]: 40)")
    ap.add_argument("--mode", type=int, default=None, help="MPEG mode (1 = joint stereo); default: the library's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of BASELINE configs [2] and [4]")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end_to_end object (pinned host PCM -> H2D -> kernel -> D2H -> bytes, pipelined "
                         "batches of 5 s streams; an extra object, never `value`)")
    ap.add_argument("--end-to-end", action="store_true", help="(accepted for older command lines; it is the default now)")
    ap.add_argument("--check-streams", type=int, default=CHECKED_STREAMS,
                    help="streams of the timed launch compared with the oracle afterwards (rank 0; default %d)" % CHECKED_STREAMS)
    ap.add_argument("--check-procs", type=int, default=0,
                    help="processes of that comparison (0 = the CPUs this process may use; 1 = in this process: under a "
                         "profiler, which would otherwise attach to every child)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

    import torch
    import lamehip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    ndev = torch.cuda.device_count()
    device_index = local_rank % ndev            # more ranks than devices (a 1-GPU box): they share
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world == 1:
        rdv = Alone()
    elif "LAMEHIP_BENCH_RDV" in os.environ:
        rdv = FileRendezvous(os.environ["LAMEHIP_BENCH_RDV"], rank, world)
    else:
        rdv = GlooRendezvous(rank, world)

    sr, B = args.samplerate, args.streams
    n = int(args.seconds * sr)
    lo, hi = lamehip.shard_streams(world * B, world, rank)      # this rank's block of the global batch
    assert hi - lo == B
    vbr_mode = 2 if args.vbr_old else 4
    enc_kw = {} if args.mode is None else {"mode": args.mode}
    enc = lamehip.Encoder(sr, args.brate, vbr_q=args.vbr, abr=args.abr, device=device_index, vbr_mode=vbr_mode, **enc_kw)
    batch = lamehip.Batch(enc, B, n, device=device_index)
    pcm = synth_on_device(torch, B, n, sr, lo, dev, bursts_per_s=args.bursts)   # seeds follow the global stream index
    torch.cuda.synchronize()
    for s in range(B):
        batch.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), n)
    # host copies for the checks / the CPU leg (rank 0 only): a few whole streams, and the first 20 s of
    # as many streams as there are physical cores
    # rank 0: CHECKED_STREAMS streams spread over its block (first and last among them); every other rank: the first and the
    # last stream of ITS OWN block -- on a node with N devices the output of each is compared with the oracle by its own rank
    if rank == 0:
        nck = max(1, min(args.check_streams, B))
        which = sorted(set(int(round(k * (B - 1) / max(nck - 1, 1))) for k in range(nck)))
    else:
        which = sorted(set([0, B - 1]))
    enc_args = ((sr, args.brate), dict(vbr_q=args.vbr, abr=args.abr, vbr_mode=vbr_mode, **enc_kw))
    host_streams, host_cpu = [pcm[s].cpu().numpy() for s in which], None
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            k = min(_physical_cores(), B)
            host_cpu = pcm[:k, :, :min(n, int(CPU_SAMPLE_SECONDS * sr))].cpu().numpy()
    del pcm
    torch.cuda.synchronize()
    frames = sum(batch.frames(s) for s in range(B))

    def barrier():
        torch.cuda.synchronize()
        rdv.barrier()
        torch.cuda.synchronize()

    batch.reserve()                     # the launch's buffers (payload, analysis pool) exist before the first step, warm-up or not
    for _ in range(args.warmup):
        batch.encode()
    kernel_ms, parts_ms = [], []
    barrier()
    t0 = time.perf_counter()
    c0 = time.process_time()
    for _ in range(args.steps):
        batch.encode(sync=True)         # (an encoded batch starts over from the initial state by itself)
        kernel_ms.append(batch.kernel_ms())
        parts_ms.append(batch.kernel_parts_ms())
    own_dt = time.perf_counter() - t0
    own_cpu = time.process_time() - c0
    barrier()
    dt = rdv.max(time.perf_counter() - t0)
    # what each rank did, for the one line rank 0 prints (timing scalars only; no stream data ever crosses ranks)
    # every rank checks its own streams (after the timed region; a difference ends the rank with an error, and with it the run)
    checked = check_against_oracle(batch, enc_args, host_streams, which,
                                   procs=(args.check_procs or _cpu_budget()) if rank == 0 else 1)
    per_rank = rdv.gather([rank, device_index, lo, hi, own_dt, own_cpu, sum(kernel_ms) / len(kernel_ms), float(checked["streams"])])

    if rank == 0:
        assert len(batch.pack(0)) > 0           # and the payload survives the host packer's consistency checks
        audio_s = world * B * args.seconds * args.steps
        value = audio_s / dt
        kavg = sum(kernel_ms) / len(kernel_ms) / 1e3
        what = ("ABR%d" % args.abr if args.abr is not None else "CBR%d" % args.brate if args.vbr is None else "VBR -V%d%s" % (args.vbr, " --vbr-old" if args.vbr_old else ""))
        res = {
            "metric": "encoded audio seconds/sec (x real-time) at %gkHz stereo %s" % (sr / 1000.0, what),
            "value": round(value, 1), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic: " + SIGNAL % args.bursts,
            "config": {"workload": ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, CBR %d kb/s, "
                                    "per GPU (BASELINE config[1])" % (B, sr / 1000.0, args.seconds, args.brate))
                       if args.vbr is None and args.abr is None else
                       ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, ABR %d kb/s, per GPU"
                        % (B, sr / 1000.0, args.seconds, args.abr)) if args.abr is not None else
                       ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, VBR -V%d (vbrquantize.c path), "
                        "per GPU (BASELINE config[2])" % (B, sr / 1000.0, args.seconds, args.vbr)) if not args.vbr_old else
                       ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, VBR -V%d --vbr-old (VBR_old_iteration_loop), "
                        "per GPU" % (B, sr / 1000.0, args.seconds, args.vbr)),
                       "streams_per_gpu": B, "seconds_per_stream": args.seconds,
                       "per_stream_x_realtime": round(value / (world * B), 2),
                       "parallelism": "static sharding: rank r owns the contiguous block shard_streams(N*B, N, r); "
                                      "no collective, no RCCL",
                       "ranks": "spawned by bench.py (file barrier)" if "LAMEHIP_BENCH_RDV" in os.environ
                       else ("launcher (gloo barrier on the CPU)" if world > 1 else "single process"),
                       "devices_visible": ndev},
            "per_rank": [{"rank": int(r[0]), "device": int(r[1]), "streams": [int(r[2]), int(r[3])],
                          "s_per_step": round(r[4] / args.steps, 6),
                          "value": round((r[3] - r[2]) * args.seconds * args.steps / r[4], 1),
                          "kernel_ms_avg": round(r[6], 3),
                          # host CPU seconds this rank's process burned per wall second of its timed region: the
                          # kernel path needs no host work, the rest is the runtime waiting for the stream
                          "host_cpu_per_wall_s": round(r[5] / r[4], 3),
                          # streams of this rank's own block that its own process compared with the oracle after the timed
                          # region (a difference would have ended the run)
                          "checked_streams": int(r[7]), "checked": "identical"} for r in per_rank],
            "collectives_on_data_path": 0,
            "pipeline": ({"kind": "split", "kernels_ms_avg": dict(zip(("analysis", "subband", "encode"),
                          [round(sum(p[1][k] for p in parts_ms) / len(parts_ms), 3) for k in range(3)]))}
                         if parts_ms and parts_ms[0][0] else {"kind": "fused"}),
            "roofline": roofline_block(frames, sr, kavg, mean_parts(parts_ms), pmc_record_for(args)),
            "checked_against_oracle": checked,
        }
        if host_cpu is not None:
            res["cpu_baseline"] = cpu_baseline(host_cpu, sr, args.brate, vbr_q=args.vbr, abr=args.abr, vbr_mode=vbr_mode)
        if not args.no_extras and world == 1 and args.vbr is None and args.abr is None:
            batch.close()
            batch = None
            res["extra"] = {
                "vbr_v2_config2": short_run(torch, lamehip, dev, device_index, 44100, 1024, 5.0, 2, 5000, 3.0,
                                            "r06_pmc_vbr2.json", vbr_q=2),
                "vbr_old_v2": short_run(torch, lamehip, dev, device_index, 44100, 1024, 5.0, 2, 5000, 3.0,
                                        "r06_pmc_vbrold2.json", vbr_q=2, vbr_mode=2),
                "cbr320_48k_bursts_config4": short_run(torch, lamehip, dev, device_index, 48000, 1024, 5.0, 2, 9000,
                                                       40.0, "r06_pmc_cbr320.json", brate=320, mode=1),
                # MPEG-2 (one granule of 576 samples per frame): the kernel object compiled with -DLH_LSF
                "mpeg2_22k_cbr64": short_run(torch, lamehip, dev, device_index, 22050, 1024, 10.0, 2, 7000, 3.0,
                                             "r06_pmc_lsf.json", brate=64),
            }
        if not args.no_end_to_end and not args.no_extras and world == 1:
            if batch is not None:
                batch.close()
                batch = None
            res["end_to_end"] = end_to_end(torch, lamehip, enc, min(B, 1024), sr, dev)
        print(json.dumps(res))
    rdv.barrier()
    rdv.close()


def end_to_end(torch, lamehip, enc, B, sr, dev, seconds=30.0, rounds=8, nbatch=2):
    """SURVEY.md 8(d) region R2: s16 PCM in pinned host memory -> H2D -> kernel -> D2H -> mp3 bytes in host
    memory, as a pipeline of `nbatch' batch objects (each with its own HIP stream) that are reused round-robin
    for `rounds' batches of B streams x `seconds': batch n's kernel runs while batch n+1's PCM goes up and batch
    n-1's bytes come down.  device_packed (the product's path for batches: the kernel assembles the bytes,
    lh_dev_emit.h) and, for comparison, the host packer on all cores (payload D2H + lamehip_batch_pack_all in a
    thread per batch).  Before the clock: the PCM is in the batches' pinned mirrors and every object has run
    once.  Checked: the device-packed bytes of four streams against the host packer's."""
    import threading
    n = int(seconds * sr)
    # (two pinned PCM mirrors of B x seconds: 5.4 GB each at 1024 x 30 s; the library's default cap is 4 GB in all)
    os.environ.setdefault("LAMEHIP_PINNED_MAX_MB", "16384")
    host = synth_on_device(torch, B, n, sr, 777, dev).cpu().numpy()
    threads = min(32, _cpu_budget())
    objs = []
    for k in range(nbatch):
        b = lamehip.Batch(enc, B, n)
        b.pcm_host()[:, :, :n] = host           # "pinned host PCM" is where R2 starts
        for s in range(B):
            b.set_length(s, n)
        objs.append(b)
    which = sorted(set([0, 1, B // 2, B - 1]))

    def mark(b):
        for s in range(B):
            b.mark_pcm(s)

    # -- the same batch HBM-resident, payload only (region R1 on this sample: what the pipeline is held against) --
    b = objs[0]
    mark(b)
    b.encode(sync=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b.encode(sync=True)
    resident = 3 * B * seconds / (time.perf_counter() - t0)

    # -- device-packed pipeline --
    for b in objs:
        b.set_device_packing()
        mark(b)
        b.encode(sync=False)
        b.fetch()
    sizes = 0
    for b in objs:
        sizes = sum(len(b.bytes_view(s)) for s in range(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    home = []                   # when each batch's bytes were in host memory
    for r in range(rounds):
        b = objs[r % nbatch]
        if r >= nbatch:
            got = sum(len(b.bytes_view(s)) for s in which)      # the previous batch of this object is home
            assert got > 0
            home.append(time.perf_counter())
        mark(b)
        b.upload()
        b.encode(sync=False)
        b.fetch()
    for k in range(nbatch):
        objs[(rounds + k) % nbatch].bytes_view(0)
        home.append(time.perf_counter())
    dt_dev = time.perf_counter() - t0
    # batch to batch once the pipeline is full: from the first batch's arrival to the last one's (the whole-run figure above
    # also holds the first upload and the last fetch, which nothing overlaps)
    steady = (len(home) - 1) * B * seconds / (home[-1] - home[0])
    dev_bytes = {s: bytes(objs[0].bytes_view(s)) for s in which}
    # phases of one batch alone (no overlap), for the record
    b = objs[0]
    torch.cuda.synchronize()
    p0 = time.perf_counter()
    mark(b)
    b.upload()
    b.sync()
    p1 = time.perf_counter()
    b.encode(sync=True)
    p2 = time.perf_counter()
    b.fetch()
    b.bytes_view(0)
    p3 = time.perf_counter()

    # -- host-packed pipeline (payload D2H + pack_all on `threads' host threads, one packing thread per batch) --
    for b in objs:
        b.set_device_packing(False)
    res_sizes = [None] * nbatch

    def pack(k):
        _, _, sz = objs[k].pack_all(threads, as_bytes=False)
        res_sizes[k] = int(sz.sum())

    for k, b in enumerate(objs):
        mark(b)
        b.encode(sync=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    workers = [None] * nbatch
    for r in range(rounds):
        k = r % nbatch
        if workers[k] is not None:
            workers[k].join()
        b = objs[k]
        mark(b)
        b.upload()
        b.encode(sync=False)
        workers[k] = threading.Thread(target=pack, args=(k,))
        workers[k].start()                      # pack_all waits for the batch's stream itself
    for w in workers:
        if w is not None:
            w.join()
    dt_host = time.perf_counter() - t0
    same = all(objs[0].pack(s) == dev_bytes[s] for s in which)
    for b in objs:
        b.close()
    if not same:
        raise SystemExit("end_to_end: device-packed bytes differ from the host packer's")
    audio = rounds * B * seconds
    return {"value": round(audio / dt_host, 1), "unit": "x real-time", "host_threads": threads,
            "mp3_bytes_per_batch": res_sizes[0], "hbm_resident_same_sample": round(resident, 1),
            "device_packed": {"value": round(audio / dt_dev, 1), "unit": "x real-time", "mp3_bytes_per_batch": int(sizes),
                              "steady_state": round(steady, 1),
                              "steady_state_note": "batches per second between the first and the last batch's arrival in host "
                                                   "memory (pipeline full); `value' is the whole run of %d batches, fill and "
                                                   "drain included" % rounds,
                              "one_batch_alone_s": {"h2d": round(p1 - p0, 4), "kernel": round(p2 - p1, 4),
                                                    "d2h": round(p3 - p2, 4)},
                              "bytes_checked_against_host_packer": {"streams": which, "result": "identical"}},
            "pipeline": "%d batch objects round-robin over %d batches, each on its own HIP stream" % (nbatch, rounds),
            "sample": "%d streams x %.0f s per batch, pinned host s16 in -> mp3 bytes in pinned host memory out"
                      % (B, seconds)}


if __name__ == "__main__":
    main()
