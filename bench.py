#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X-native MP3 encode inner loop.

Metric (BASELINE.json): encoded audio seconds per wall-clock second (x real-time)
at 44.1 kHz stereo CBR 128 kb/s.  A "step" is one pass of the hot path (psycho-
acoustics + polyphase/MDCT + quantisation loop -> side-info payload in HBM) over
one batch of synthetic streams whose PCM is already resident in HBM
(BASELINE config[1]: batch = 1024 streams x 60 s per GPU).  With --gpus N each
rank encodes its own 1024 streams (static sharding, no collective in the data
path; weak scaling); rank 0 prints ONE JSON line with the whole-job aggregate.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_FRAME = 9792          # 4608 B PCM in + 5184 B side info out
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def synth_on_device(torch, batch, n, sr, seed0, device, chunk=32):
    """Seeded music-like stereo s16 PCM generated on the GPU (no dataset access):
    8 partials with independent phases per channel + low-level noise + decaying
    noise bursts every 1/3 s.  Returns an int16 tensor [batch, 2, n] on device."""
    out = torch.empty((batch, 2, n), dtype=torch.int16, device=device)
    t = torch.arange(n, device=device, dtype=torch.float32) / sr
    step = sr // 3
    env = torch.exp(-torch.arange(2000, device=device, dtype=torch.float32) / 300.0)
    for b0 in range(0, batch, chunk):
        b1 = min(batch, b0 + chunk)
        g = torch.Generator(device=device)
        g.manual_seed(0x4C414D45 + seed0 + b0)
        x = torch.zeros((b1 - b0, 2, n), device=device)
        for k in range(8):
            f = 220.0 * 2 ** (k / 2.0)
            ph = torch.rand((b1 - b0, 2, 1), generator=g, device=device) * 6.2831853
            x += (0.5 / (k + 1)) * torch.sin(6.2831853 * f * t + ph)
        x += 0.01 * torch.randn((b1 - b0, 2, n), generator=g, device=device)
        for s in range(step // 2, n - 2000, step):
            x[:, :, s:s + 2000] += 0.6 * env * torch.randn((b1 - b0, 2, 2000), generator=g, device=device)
        x = x / x.abs().amax(dim=(1, 2), keepdim=True) * (0.8 * 32767)
        out[b0:b1] = x.to(torch.int16)
    return out


def cpu_baseline(sr, brate, seconds_budget=12.0, vbr_q=None, abr=None):
    """Time the compiled reference (oracle/_ref, kind 'reference') -- or the CPU
    restatement (kind 'port') when the reference build is absent -- on ONE host
    core over a bounded sample of the same workload."""
    import helpers
    import numpy as np
    n = sr * 20
    pcm = helpers.synth_stream(12345, n, sr)
    if helpers.have_reference():
        ref = helpers.Reference()
        kind = "reference"

        def run():
            ref.encode(pcm, sr, brate, vbr_q=vbr_q, abr=abr)
    else:
        import lamehip
        orc = helpers.Oracle()
        enc = lamehip.Encoder(sr, brate, require_device=False, vbr_q=vbr_q, abr=abr)
        cfg, tab = enc.config(), enc.tables()
        kind = "port"

        def run():
            orc.encode_frames(cfg, tab, pcm)
    t0 = time.time()
    reps = 0
    while True:
        run()
        reps += 1
        if time.time() - t0 > seconds_budget or reps >= 8:
            break
    dt = time.time() - t0
    return {"value": round(reps * 20.0 / dt, 2), "unit": "x real-time", "cores": 1, "kind": kind,
            "sample": "%d x 20 s seeded synthetic 44.1 kHz stereo, %s, one host core"
                      % (reps, ("ABR %d" % abr) if abr is not None else ("CBR %d" % brate) if vbr_q is None
                         else ("VBR -V%d" % vbr_q))}


def end_to_end(torch, lamehip, enc, B, sr, dev, seconds=5.0):
    """SURVEY.md 8(d) region R2: s16 PCM in host memory -> H2D -> kernel -> D2H -> host bit packing
    to mp3 bytes in memory, all host cores packing.  A separate, smaller sample (B x 5 s)."""
    n = int(seconds * sr)
    host = synth_on_device(torch, B, n, sr, 777, dev).cpu().numpy()
    threads = min(32, os.cpu_count() or 1)      # measured best on the 256-thread host: 32
    b = lamehip.Batch(enc, B, n)
    best = None
    for _ in range(2):
        b.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(B):
            b.set_pcm(s, host[s, 0], host[s, 1])
        b.encode(sync=True)
        t1 = time.perf_counter()
        _, _, sizes = b.pack_all(threads, as_bytes=False)
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1, int(sizes.sum()))
    # the same with the bit packer on the device: D2H of finished bytes, no host packing
    stride = (b.frames(0) + 2) * (1500 if enc.config().vbr else 1100)
    dbest = None
    for _ in range(2):
        b.reset()
        b.set_device_packing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(B):
            b.set_pcm(s, host[s, 0], host[s, 1])
        b.encode(sync=True)
        t1 = time.perf_counter()
        _, dsizes = b.get_bytes_all(stride)
        t2 = time.perf_counter()
        if dbest is None or t2 - t0 < dbest[0]:
            dbest = (t2 - t0, t1 - t0, t2 - t1, int(dsizes.sum()))
    b.close()
    return {"value": round(B * seconds / best[0], 1), "unit": "x real-time", "host_threads": threads,
            "h2d_plus_kernel_s": round(best[1], 3), "d2h_plus_pack_s": round(best[2], 3), "mp3_bytes": best[3],
            "device_packed": {"value": round(B * seconds / dbest[0], 1), "unit": "x real-time",
                              "h2d_plus_kernel_s": round(dbest[1], 3), "d2h_s": round(dbest[2], 3),
                              "mp3_bytes": dbest[3]},
            "sample": "%d streams x %.0f s, host s16 in -> mp3 bytes out" % (B, seconds)}


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
    ap.add_argument("--abr", type=int, default=None, metavar="KBPS", help="ABR at a mean of KBPS instead of CBR")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time host PCM -> H2D -> kernel -> D2H -> host bit packing (threads) on a "
                         "5 s sample; reported as an extra object, never as `value`")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import lamehip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    sr, B = args.samplerate, args.streams
    n = int(args.seconds * sr)
    enc = lamehip.Encoder(sr, args.brate, vbr_q=args.vbr, abr=args.abr)
    batch = lamehip.Batch(enc, B, n)
    # static sharding: rank r owns global streams [r*B, (r+1)*B); seeds follow the global index
    pcm = synth_on_device(torch, B, n, sr, rank * B, dev)
    torch.cuda.synchronize()
    for s in range(B):
        batch.set_pcm_device(s, pcm[s, 0].data_ptr(), pcm[s, 1].data_ptr(), n)
    del pcm
    torch.cuda.synchronize()
    frames = sum(batch.frames(s) for s in range(B))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.reset()
        batch.encode()
    kernel_ms = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.reset()
        batch.encode(sync=True)
        kernel_ms.append(batch.kernel_ms())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # sanity outside the timed region: the payload must survive the host packer's checks
    nbytes = len(batch.pack(0))
    assert nbytes > 0

    if rank == 0:
        audio_s = world * B * args.seconds * args.steps
        value = audio_s / dt
        kavg = sum(kernel_ms) / len(kernel_ms) / 1e3
        achieved = frames * ALG_BYTES_PER_FRAME / kavg / 1e9
        res = {
            "metric": "encoded audio seconds/sec (x real-time) at 44.1kHz stereo "
                      + ("ABR%d" % args.abr if args.abr is not None else "CBR128" if args.vbr is None
                         else "VBR -V%d" % args.vbr),
            "value": round(value, 1), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, CBR %d kb/s, "
                                    "per GPU (BASELINE config[1])" % (B, sr / 1000.0, args.seconds, args.brate))
                       if args.vbr is None and args.abr is None else
                       ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, ABR %d kb/s, per GPU"
                        % (B, sr / 1000.0, args.seconds, args.abr)) if args.abr is not None else
                       ("batch=%d synthetic %.1f kHz stereo streams x %.0f s, VBR -V%d (vbrquantize.c path), "
                        "per GPU (BASELINE config[2])" % (B, sr / 1000.0, args.seconds, args.vbr)),
                       "streams_per_gpu": B, "seconds_per_stream": args.seconds,
                       "per_stream_x_realtime": round(value / (world * B), 2),
                       "parallelism": "static stream sharding, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "traffic_note": "not collected here (needs rocprofv3 --pmc passes); offline measurement on the "
                                         "1024 x 5 s workload: profiles/r01_traffic.json, DESIGN.md section 4",
                         "kernel": "lh_encode_kernel", "kernel_ms_avg": round(kavg * 1e3, 3),
                         "alg_bytes_per_frame": ALG_BYTES_PER_FRAME, "frames_per_launch": frames},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(sr, args.brate, vbr_q=args.vbr, abr=args.abr)
        if args.end_to_end and world == 1:
            res["end_to_end"] = end_to_end(torch, lamehip, enc, B, sr, dev)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
