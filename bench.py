#!/usr/bin/env python3
"""Headline bench: Msamples/s of G.711 mu-law decode + RMS/peak/byte-mean meter,
65 536 channels @ 8 kHz per GPU, F = 128 frames (20 ms each) per launch.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (igdsp_decode_meter through the C ABI) over one
device-resident batch payload[F][C][160] (1.34 GB per GPU: larger than the 256 MB
Infinity Cache, so the figure is an HBM figure).  Channels shard by contiguous range
across ranks (weak scaling: 65 536 ch per GPU; N = 8 is BASELINE configs[3], 524 288 ch)
with ONE 896-byte RCCL all-reduce per launch for the node-wide sum-of-squares / peak,
issued on a side stream behind the kernel's event.

Buffers come from the C ABI: the whole input / output set of the step from ONE igdsp_io_alloc call, which places inputs,
records and bulk outputs in different classes of device memory (on MI355X a launch that reads one class and writes another
is ~13 % faster) — this file carries no placement logic of its own.  The same step is ALSO timed on plain
consecutive igdsp_dev_alloc buffers — what a host gets that follows include/igdsp.h literally — twice: straight after the
allocation with W warm-up launches only (`roofline.frac_plain_cold`) and after the same ~60 ms clock pre-warm the headline
gets (`roofline.frac_plain_warm`: placement is then the ONLY difference to `roofline.frac`).  `placement_setup_ms` (top
level) is what the placed set cost at start-up.  Before the W warm-up steps of the headline measurement the launch is
repeated for ~60 ms so the clocks settle (a gateway that runs continuously is always in that state); none of this is inside a step.

Rank 0 prints ONE JSON line; `roofline` is measured live with one pair of HIP events on the launch
stream around the K timed launches (average launch duration = elapsed / K); a second pass of >= 20 launches, each between
its own pair of events, gives `roofline.median_ms / min_ms / max_ms` (BASELINE.md section 3 asks for the median; a pair
per launch adds event handling to every launch, so that pass is not the headline).  `cpu_baseline` is the CPU oracle timed
on this box's host cores (N = 1 only), one frame per call through its single-frame entry.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = 160
# algorithmic bytes per sample, meter-only = (160 payload + 1 codec id + 16 result) / 160
BYTES_PER_SAMPLE = {"meter": (160 + 1 + 16) / 160.0, "store": (160 + 1 + 16 + 320) / 160.0,
                    "roundtrip": (160 + 1 + 16 + 160) / 160.0,   # config #5: read 1 + write 1 + record
                    "depayload": (180 + 160 + 2 + 8) / 160.0,
                    "rtp": (192 + 1 + 16 + 8) / 160.0,           # fused: 192 B packet slot in, record + info out
                    "packets": (180 + 1 + 16 + 8) / 160.0,       # fused, packets packed at their natural 180 B stride
                    "window": (180 + 1 + 16 + 8) / 160.0,        # the same + the ED-137 gated window and the silence run folded in the same pass (hold / probe / run summaries: < 0.1 % more)
                    "wav": (160 + 320) / 160.0,                  # 8(f) rank 2: payload in, [b, 0x00] file images out (+ 44 B per channel)
                    "encode": (320 + 1 + 160) / 160.0}           # a2: int16 in, code out    # 8(f) rank 1: 180 B packet in, dense payload + len + info out
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=65536, help="channels PER GPU")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--frame-bytes", type=int, default=160,
                    help="meter / store / roundtrip modes: bytes (= samples) per frame; 160 is the 20 ms frame of the headline, the reference's "
                         "other sizes (24, 80, 164, 168, 240) run k_meter_strided / k_roundtrip_strided, other multiples of 4 the "
                         "LDS-image kernel")
    ap.add_argument("--total-channels", type=int, default=0,
                    help="strong scaling: this many channels in total, split evenly over the ranks (SURVEY 8d: 524288 over 1/2/4/8 "
                         "GPUs); overrides --channels and reports \"scaling\": \"strong\"")
    ap.add_argument("--mode", choices=["meter", "store", "roundtrip", "depayload", "rtp", "packets", "window", "encode", "wav"], default="meter")
    ap.add_argument("--variant", type=int, default=0,
                    help="igdsp_set_variant: 0 tuned default, 1 wave-per-frame, 2 chunk64, 3 chunk64 fat waves, 4 round trip through the compressor cell table")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on GPUs; gloo only to rehearse the N>1 path on one GPU")
    ap.add_argument("--one-gpu-rehearsal", action="store_true", help="all ranks use cuda:0 (with --backend gloo)")
    ap.add_argument("--no-agg", action="store_true", help="experiment: skip the launch aggregate (N=1 only)")
    ap.add_argument("--no-stream-calib", action="store_true", help="skip the read-only stream calibration kernel")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed launches of the step before the W warmup steps until this much GPU time has passed: the "
                         "GPU needs ~20 ms of load to reach its steady clocks (first 30 launches measure ~6 %% slow)")
    ap.add_argument("--wav-offset", type=int, default=84, help="wav mode: byte offset of file 0 inside its buffer (84: data bytes line-aligned; 0: headers line-aligned)")
    ap.add_argument("--placement", choices=["both", "abi", "plain"], default="both",
                    help="both: headline on igdsp_io_alloc buffers + the plain-allocation figures (cold and pre-warmed) on igdsp_dev_alloc buffers; "
                         "abi / plain: only that one (plain = the headline itself runs on plain buffers)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: bring up a 1-rank nccl (= RCCL) group and run the N > 1 step sequence (kernel -> event -> "
                         "side-stream all_reduce(int64[112]) -> next launch); reports the per-launch cost against the plain step")
    return ap.parse_args()


def cpu_baseline(seconds: float):
    """The oracle's table-decode + exact sum-of-squares + peak + sqrt loop, one frame
    at a time, on a bounded sample of the SAME workload (first 4 096 ch x 32 frames of the D-uniform
    stream), all host cores and single thread."""
    import numpy as np

    from oracle import oracle as orc   # checker/baseline only

    C_, F_ = 4096, 32
    payload = orc.gen_uniform(F_ * C_ * N_SAMPLES).reshape(F_, C_, N_SAMPLES)
    codec = np.zeros((C_,), np.uint8)
    samples = payload.size
    cores = len(os.sched_getaffinity(0))
    t1 = orc.time_single_frame(payload, codec, 1, 1)
    reps1 = max(1, int(0.2 * seconds / max(t1, 1e-6)))
    t1 = orc.time_single_frame(payload, codec, 1, reps1) / reps1
    tn = orc.time_single_frame(payload, codec, cores, 1)
    repsn = max(1, int(0.4 * seconds / max(tn, 1e-6)))
    tn = orc.time_single_frame(payload, codec, cores, repsn) / repsn
    tl = orc.time_decode_meter(payload, codec, cores, max(1, repsn // 2)) / max(1, repsn // 2)     # the same arithmetic as one batch loop
    tb = orc.time_byte_mean(payload, cores, max(1, repsn // 2)) / max(1, repsn // 2)
    return {
        "value": round(samples / tn / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"oracle B1 (scalar table decode + u64 sum x^2 + peak + sqrt, -O2), ONE FRAME PER CALL through the oracle's "
                  f"single-frame entry (call id -> slot, as igdsp_on_rtp_frame takes it), on the first {C_} ch x {F_} frames "
                  f"({samples / 1e6:.0f} MB: cache-resident on the host) of the same D-uniform stream, x{repsn} passes, {cores} pthreads",
        "single_thread_value": round(samples / t1 / 1e6, 2),
        "batch_loop_value": round(samples / tl / 1e6, 2),
        "reference_loop_byte_mean_value": round(samples / tb / 1e6, 2),
        "cpu_model": _cpu_model(), "compiler_flags": "gcc -O2 -funsigned-char (oracle/Makefile; -O2 as the reference's .pro:88)",
    }


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _block_form(channels, env):
    """The launchers' rule for the block-owned form of the round trip / the fused window (igdsp_k_codec.hip launch_roundtrip,
    igdsp_capi.hip igdsp_decode_meter_window): as many channel groups per block (<= 4) as still give every CU a block, taken when
    the blocks fill whole rounds of the CUs to 85 % or (window: always; round trip: from 0.6 of a round) within one round.  Only used to NAME the dominant kernel in the JSON line."""
    import torch
    e = os.environ.get(env)
    if e is not None:
        return int(e) != 0
    n_groups, cus = channels // 64, max(1, torch.cuda.get_device_properties(0).multi_processor_count)
    if n_groups == 0:
        return False
    gpb = next((g for g in (4, 2) if n_groups % g == 0 and n_groups // g >= cus), 1)
    blocks = n_groups // gpb
    rounds = (blocks + cus - 1) // cus
    if env == "IGDSP_WIN_BLK":
        return rounds == 1 or blocks * 100 >= rounds * cus * 85
    return rounds == 1 and blocks * 10 >= cus * 6


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from igate4xsoftphonedsp_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the igdsp kernels have no CPU fallback")
    if args.one_gpu_rehearsal:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    elif args.force_collective:
        import socket

        with socket.socket() as sk:                      # a free local port for the 1-rank rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", local))
    if world > capi.AGG_MAX_RANKS:
        raise SystemExit("aggregate vector has 8 peak slots")

    if args.total_channels:
        if args.total_channels % (world * 64):
            raise SystemExit("--total-channels must be a multiple of 64 x the number of ranks")
        args.channels = args.total_channels // world
    C_, F_, n = args.channels, args.frames, N_SAMPLES
    if args.frame_bytes != N_SAMPLES:
        if args.mode not in ("meter", "store", "roundtrip") or args.frame_bytes < 4 or args.frame_bytes > 256 or args.frame_bytes % 4:
            raise SystemExit("--frame-bytes: meter mode only, a multiple of 4 in 4..256")
        n = args.frame_bytes
    C_total = C_ * world
    ctx = capi.Context(device=local, max_channels=1024)
    ctx.set_variant(args.variant)
    main_s = torch.cuda.Stream()       # explicit launch stream: kernels, resets and timers all live on it
    comm_s = torch.cuda.Stream()       # side stream for the per-launch all-reduce
    torch.cuda.set_stream(main_s)
    hs = main_s.cuda_stream
    assert hs != 0

    # ---- buffers.  Every big buffer of the step comes from the C ABI, twice:
    #   "plain": igdsp_dev_alloc one after the other, inputs first — what a host writes when it follows the header literally;
    #   "abi"  : ONE igdsp_io_alloc call for the whole set — the library places inputs / records / bulk outputs in different
    #            classes of device memory (no search or arena lives in this file any more).
    # The driver line's `value` / `roofline.frac` are measured on the "abi" set, `roofline.frac_plain_warm` / `_cold` on the "plain" one.
    U8, I16, I64 = torch.uint8, torch.int16, torch.int64
    MODE = args.mode
    spec = []                                    # (name, shape, dtype, role)
    if MODE in ("meter", "store", "roundtrip", "wav"):
        spec.append(("pl", (F_, C_, n), U8, capi.IO_INPUT))
    wav_stride = (44 + 2 * F_ * n + 127) // 128 * 128
    if MODE == "wav":
        spec.append(("files", (C_ * wav_stride + 128,), U8, capi.IO_BULK))
    if MODE == "depayload":
        spec.append(("pk", (F_, C_, 180), U8, capi.IO_INPUT))
    if MODE == "encode":
        spec.append(("pcm_in", (F_, C_, n), I16, capi.IO_INPUT))
    if MODE == "rtp":
        spec.append(("slots", (F_, C_, 192), U8, capi.IO_INPUT))
    if MODE in ("packets", "window"):
        spec.append(("slots", (F_, C_, 180), U8, capi.IO_INPUT))
    if MODE not in ("encode", "wav"):
        # igdsp_frame_stats[F][C]; at 16 .. 32-byte frames the records are 40 % of the launch's traffic: its bulk output
        spec.append(("st", (F_ * C_ * 2,), I64, capi.IO_BULK if (MODE == "meter" and n <= 32) else capi.IO_RECORD))
    if MODE in ("rtp", "packets", "window", "depayload"):
        spec.append(("info", (F_ * C_,), I64, capi.IO_RECORD))
    if MODE == "window":
        spec.append(("hold", (C_ * 4,), I64, capi.IO_RECORD))
        spec.append(("probe", (C_,), I64, capi.IO_RECORD))
        spec.append(("work", (C_ * 48,), I64, capi.IO_RECORD))                 # igdsp_window_work_bytes(C) = 8 segments x C x 48 B
    if MODE == "depayload":
        spec.append(("len", (F_ * C_,), I16, capi.IO_RECORD))
        spec.append(("dense", (F_, C_, n), U8, capi.IO_BULK))
    if MODE == "store":
        spec.append(("pcm", (F_, C_, n), I16, capi.IO_BULK))
    if MODE in ("encode", "roundtrip"):
        spec.append(("out", (F_, C_, n), U8, capi.IO_BULK))
    if MODE == "roundtrip":
        spec.append(("hold", (C_ * 4,), I64, capi.IO_RECORD))

    def nbytes(shape, dtype):
        return int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()

    class BufSet:
        def __init__(self, how):
            self.how, self.report, self.ioset, self.raw = how, None, None, []
            sizes = [nbytes(sh, dt) for _, sh, dt, _ in spec]
            if how == "abi":
                self.ioset, ptrs, self.report = ctx.io_alloc([(b, role) for b, (_, _, _, role) in zip(sizes, spec)])
            else:
                ptrs = [ctx.dev_alloc(b) for b in sizes]
                self.raw = ptrs
            self.t = {name: capi.as_tensor(p, b, dt, sh) for p, b, (name, sh, dt, _) in zip(ptrs, sizes, spec)}
            for name, _, _, role in spec:
                if role != capi.IO_INPUT:
                    self.t[name].zero_()
            self.fill_inputs()

        def fill_inputs(self):
            """Synthetic, generated on the device, shard-invariant: this rank holds channels [rank*C, (rank+1)*C) of
            the global [F][C_total][160] D-uniform array."""
            t = self.t
            if "pl" in t:
                for f in range(F_):
                    ctx.gen_uniform(t["pl"][f], C_ * n, first_byte=(f * C_total + rank * C_) * n, stream=hs)
            if "pk" in t:                                                     # [F][C][180] ED-137 packets (header bytes arbitrary but PT 0)
                ctx.gen_uniform(t["pk"], t["pk"].numel(), seed=7, stream=hs)
                t["pk"][:, :, 0] = 0x90
                t["pk"][:, :, 1] = 0
            if "pcm_in" in t:
                ctx.gen_uniform(t["pcm_in"], t["pcm_in"].numel() * 2, seed=11, stream=hs)
            if "slots" in t:
                ctx.gen_uniform(t["slots"], t["slots"].numel(), seed=7, stream=hs)
                if MODE == "rtp":                                             # [F][C][192] slots: size 180, PT 0, payload at +32
                    t["slots"][:, :, 0] = 180
                    t["slots"][:, :, 1:12] = 0
                    t["slots"][:, :, 12] = 0x90
                    t["slots"][:, :, 13] = 0
                else:                                                         # [F][C][180] ED-137 packets, PT 0, all full
                    t["slots"][:, :, 0] = 0x90
                    t["slots"][:, :, 1] = 0
            if MODE == "window":
                ctx.hold_reset(t["hold"], C_, stream=hs)
                t["probe"].zero_()
            if MODE == "roundtrip":                                           # BASELINE configs[4]: mixed A-law / mu-law, D-speech
                # D-speech: two-tone + noise, amplitude 1000*(1 + c mod 30), encoded with the oracle's encoder.  The
                # generator is CPU test infrastructure, so a [F][480][160] tile (16 amplitude periods, both laws) is generated once
                # and replicated across the channels on the device — synthetic data either way.
                from oracle import oracle as orc

                tile_c = 480
                tile = orc.gen_speech(tile_c, F_, n, (np.arange(tile_c) & 1).astype(np.uint8) * 8)
                d_tile = torch.from_numpy(tile).cuda()
                t["pl"].copy_(d_tile[:, torch.arange(C_, device="cuda") % tile_c, :])
                del d_tile
                ctx.hold_reset(t["hold"], C_, stream=hs)
            torch.cuda.synchronize()

        def close(self):
            torch.cuda.synchronize()
            self.t = {}
            if self.ioset is not None:
                self.ioset.close()
            for p in self.raw:
                ctx.dev_free(p)
            self.raw = []

    d_cd = torch.zeros((C_,), dtype=U8, device="cuda")                         # mu-law (RTP PT 0) everywhere
    if MODE == "roundtrip":
        d_cd[1::2] = 8
    d_radio = torch.ones((C_,), dtype=U8, device="cuda")

    # one pre-zeroed aggregate per step (896 B each): a launch ADDS into its aggregate, so nothing has to be cleared
    # between launches and, for N > 1, all-reduce k runs on the side stream on its own buffer while launch k + 1 runs
    n_steps_total = args.warmup + args.steps
    use_coll = world > 1 or args.force_collective
    agg_ring = torch.zeros((n_steps_total, capi.AGG_WORDS), dtype=torch.int64, device="cuda")
    ev_k = [torch.cuda.Event() for _ in range(n_steps_total)] if use_coll else []
    region = ctx.timer()                             # HIP events on the launch stream around the K timed launches

    NO_INFO = os.environ.get("IGDSP_BENCH_NO_INFO") == "1"       # experiments: the packet modes without the igdsp_rtp_info output
    NO_RECORDS = os.environ.get("IGDSP_BENCH_WINDOW_ONLY") == "1"   # window mode without per-frame records / info: the windows alone

    def launch(agg, B):
        t = B.t
        if MODE == "encode":
            ctx.encode(t["pcm_in"], d_cd, C_, F_, n, t["out"], stream=hs)
        elif MODE == "wav":
            # file images start 84 bytes into the buffer: base + 84 + 44 = the payload-derived bytes of every file are 128-byte aligned
            ctx.wav_expand(t["pl"], C_, F_, n, t["files"].data_ptr() + args.wav_offset, wav_stride, stream=hs)
        elif MODE == "rtp":
            ctx.decode_meter_rtp(t["slots"], d_cd, C_, F_, t["st"], info=t["info"], agg=agg, rank=rank, stream=hs)
        elif MODE == "packets":
            ctx.decode_meter_packets(t["slots"], None, d_cd, C_, F_, 180, 20, t["st"], info=None if NO_INFO else t["info"], agg=agg, rank=rank, stream=hs)
        elif MODE == "window":
            win = ctx.window(t["hold"], gate_mode=capi.GATE_SQU_OR_PTT, probe=t["probe"], work=t["work"])
            ctx.decode_meter_window(capi.PKT_PACKED, t["slots"], None, d_cd, None, C_, F_, 180, 20, None if NO_RECORDS else t["st"], win,
                                    info=None if (NO_INFO or NO_RECORDS) else t["info"], agg=agg, rank=rank, stream=hs)
        elif MODE == "depayload":
            ctx.depayload(t["pk"], None, d_radio, C_, F_, 180, n, t["dense"], t["len"], t["info"], stream=hs)
        elif MODE == "roundtrip":
            ctx.roundtrip_peakhold(t["pl"], d_cd, C_, F_, n, t["out"], t["st"], t["hold"], stream=hs)
        else:
            ctx.decode_meter(t["pl"], d_cd, C_, F_, n, t["st"], pcm=t.get("pcm"), agg=None if args.no_agg else agg, rank=rank, stream=hs)

    def gpu_ms(fn, reps):
        t = ctx.timer()
        t.start(hs)
        for _ in range(reps):
            fn()
        t.stop(hs)
        ms = t.elapsed_ms() / reps
        t.close()
        return ms

    scratch_agg = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")

    def prewarm(B):
        """clock pre-warm (not steps: no aggregate ring, no collective): the GPU needs ~20 ms of load to reach its steady clocks"""
        spent = 0.0
        while spent < args.prewarm_ms:
            spent += 20 * gpu_ms(lambda: launch(scratch_agg, B), 20)

    # ---- the plain figures: consecutive igdsp_dev_alloc buffers (a host that follows the header literally).  Cold = W warm-up
    # launches, K timed ones, nothing else; warm = the same K launches after the headline's own clock pre-warm, so that the
    # ONLY difference between frac_plain_warm and frac is where the buffers sit.
    plain_cold_ms = plain_warm_ms = None
    if args.placement in ("both", "plain"):
        P = BufSet("plain")
        for _ in range(args.warmup):
            launch(scratch_agg, P)
        plain_cold_ms = gpu_ms(lambda: launch(scratch_agg, P), args.steps)
        if args.placement == "both":
            prewarm(P)
            for _ in range(args.warmup):
                launch(scratch_agg, P)
            plain_warm_ms = gpu_ms(lambda: launch(scratch_agg, P), args.steps)
            P.close()
            del P
    if args.placement == "plain":
        OUT = P
    else:
        OUT = BufSet("abi")

    def step(i: int, coll: bool = True):
        agg = agg_ring[i] if coll else scratch_agg   # the comparison pass of --force-collective must not add into the ring again
        launch(agg, OUT)
        if use_coll and coll:                      # node-wide sum / peak: one 896-byte all-reduce per launch, side stream
            ev_k[i].record(main_s)
            with torch.cuda.stream(comm_s):
                comm_s.wait_event(ev_k[i])
                dist.all_reduce(agg, op=dist.ReduceOp.SUM)

    if use_coll:                                   # communicator / channel setup is not part of any step (holds for --warmup 0 too)
        prime = torch.zeros((capi.AGG_WORDS,), dtype=torch.int64, device="cuda")
        with torch.cuda.stream(comm_s):
            dist.all_reduce(prime, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    prewarm(OUT)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    region.start(hs)
    for i in range(args.steps):
        step(args.warmup + i)
    region.stop(hs)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # average launch duration: the K launches run back to back on `hs`, bracketed by ONE pair of HIP events on that
    # stream (bracketing every launch with its own pair adds ~10 us of event handling to each 0.25 ms launch)
    kern_avg_ms = region.elapsed_ms() / args.steps
    if args.placement == "plain":
        plain_warm_ms = kern_avg_ms                   # the headline itself ran on plain buffers, pre-warmed
    # "median of >= 20 launches after 3 warm-ups" — a pass of its own, every launch between its own
    # pair of events (no aggregate ring, no collective: the kernel alone)
    n_med = max(20, args.steps)
    for _ in range(3):
        launch(scratch_agg, OUT)
    pairs = [ctx.timer() for _ in range(n_med)]
    for t in pairs:
        t.start(hs)
        launch(scratch_agg, OUT)
        t.stop(hs)
    per_launch = sorted(t.elapsed_ms() for t in pairs)
    for t in pairs:
        t.close()
    median_ms = per_launch[n_med // 2] if n_med % 2 else 0.5 * (per_launch[n_med // 2 - 1] + per_launch[n_med // 2])
    coll_info = None
    if args.force_collective and world == 1:
        # the same K steps again WITHOUT the event + side-stream all-reduce: what the N > 1 sequence costs a launch
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, coll=False)
        torch.cuda.synchronize()
        dt_plain = time.perf_counter() - t1
        coll_info = {"backend": "nccl (RCCL)", "world": 1, "ms_per_step_with_allreduce": round(dt / args.steps * 1e3, 4),
                     "ms_per_step_without": round(dt_plain / args.steps * 1e3, 4),
                     "overhead_frac": round(dt / dt_plain - 1.0, 4), "vector": f"int64[{capi.AGG_WORDS}] on the device, side stream behind the launch's event"}

    # node-wide aggregate from the last launch (after the all-reduce every rank holds all peak slots)
    from igate4xsoftphonedsp_amd import dist as igdist

    node = igdist.node_view(agg_ring[n_steps_total - 1])

    samples_per_step_rank = C_ * F_ * n
    total_samples = samples_per_step_rank * world * args.steps
    value = total_samples / dt / 1e6
    bps = BYTES_PER_SAMPLE[args.mode] if n == N_SAMPLES else (n + 1 + 16 + {"store": 2 * n, "roundtrip": n}.get(args.mode, 0)) / n
    achieved = samples_per_step_rank * bps / (kern_avg_ms * 1e-3) / 1e9
    kernel_name = "k_encode_lut16" if args.mode == "encode" else "k_wav_expand16" if args.mode == "wav" else "k_meter_rtp64" if args.mode in ("rtp", "packets", "window") else "k_depayload64" if args.mode == "depayload" else ("k_roundtrip_chunk64" if args.variant == 4 else ("k_roundtrip_blk64" if _block_form(C_, "IGDSP_RT_BLK") else "k_roundtrip_lut64") if n == N_SAMPLES else "k_roundtrip_strided") if args.mode == "roundtrip" else ("k_meter_wave_per_frame" if args.variant == 1 else "k_meter_chunk64" if n == N_SAMPLES else "k_meter_tiny" if (16 <= n <= 32 and args.mode == "meter") else "k_meter_strided" if ((n >> 4) in (1, 4, 5, 6, 8, 10, 12, 15) and (n >> 2) & 3 != 3 and n not in (244, 248)) else "k_meter_image")

    def frac_of(ms):
        return None if ms is None else round(samples_per_step_rank * bps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)

    out = {
        "metric": "Msamples/s G.711 decode+RMS, 65536ch@8kHz; %HBM roofline at 1/2/4/8 GPU",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong" if args.total_channels else "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"{C_} ch/GPU x {F_} frames x {n} samples {'mixed A-law/mu-law' if args.mode == 'roundtrip' else 'mu-law'} "
                        f"decode+meter ({args.mode}), device-resident {nbytes(spec[0][1], spec[0][2]) / 1e9:.2f} GB/GPU, "
                        f"{'D-speech tile x channels' if args.mode == 'roundtrip' else 'D-uniform seed 0x20241218'}",
            "channels_per_gpu": C_, "channels_total": C_total, "frames_per_launch": F_, "samples_per_frame": n,
            "sharding": "contiguous channel ranges, no data-path collective; one 896 B all-reduce per launch" if world > 1 else "single GPU",
            "kernel_variant": args.variant,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_plain_warm": frac_of(plain_warm_ms), "frac_plain_cold": frac_of(plain_cold_ms),
            "kernel_avg_ms_plain_warm": None if plain_warm_ms is None else round(plain_warm_ms, 4),
            "kernel_avg_ms_plain_cold": None if plain_cold_ms is None else round(plain_cold_ms, 4),
            "traffic": None,
            "kernel": kernel_name, "kernel_avg_ms": round(kern_avg_ms, 4),
            "median_ms": round(median_ms, 4), "min_ms": round(per_launch[0], 4), "max_ms": round(per_launch[-1], 4),
            "median_launches": n_med, "frac_median": frac_of(median_ms),
            "algorithmic_bytes_per_sample": round(bps, 5),
            "algorithmic_bytes_per_launch": int(samples_per_step_rank * bps),
        },
        "aggregate": {"node_rms": round(node["rms"], 3), "node_peak": node["peak"], "samples": node["samples"]},
    }

    if not args.no_stream_calib:
        # two bare calibration kernels on the same buffer: a read-only stream, and the meter kernel's exact
        # traffic pattern (10 KiB read + 1 KiB record store per super-chunk) with no per-sample work
        import ctypes as CT

        sink = torch.zeros((1,), dtype=torch.int64, device="cuda")
        d_in = OUT.t[spec[0][0]]                        # the mode's main input
        in_b = d_in.numel() * d_in.element_size()
        tm = ctx.timer()
        for _ in range(3):
            ctx.stream_read(d_in, in_b, sink, stream=hs)
        tm.start(hs)
        for _ in range(10):
            ctx.stream_read(d_in, in_b, sink, stream=hs)
        tm.stop(hs)
        out["roofline"]["stream_read_GBs"] = round(in_b * 10 / (tm.elapsed_ms() * 1e-3) / 1e9, 1)
        if "st" in OUT.t and in_b // 10 <= OUT.t["st"].numel() * 8:
            fn = ctx.L.igdsp_internal_stream_rw
            fn.restype = CT.c_int
            fn.argtypes = [CT.c_void_p, CT.c_void_p, CT.c_size_t, CT.c_void_p, CT.c_void_p]
            for _ in range(3):
                fn(ctx.h, d_in.data_ptr(), in_b, OUT.t["st"].data_ptr(), hs)
            tm.start(hs)
            for _ in range(10):
                fn(ctx.h, d_in.data_ptr(), in_b, OUT.t["st"].data_ptr(), hs)
            tm.stop(hs)
            rw_ms = tm.elapsed_ms() / 10
            out["roofline"]["same_traffic_stream_ms"] = round(rw_ms, 4)
            if args.mode == "meter":
                out["roofline"]["frac_of_same_traffic_stream"] = round(rw_ms / kern_avg_ms, 4)

    # HBM traffic per launch from the PMC counters is NOT measured in this run (counters need rocprofv3 passes of their own):
    # it is the figure recorded by tools/profile.sh for this mode and shape, labelled with the files it came from
    traffic_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(traffic_file):
        try:
            with open(traffic_file) as fh:
                tr = json.load(fh).get(args.mode + (str(args.frame_bytes) if args.frame_bytes != N_SAMPLES else ""))
            if tr and tr.get("kernel") == kernel_name and tr.get("channels") == C_ and tr.get("frames") == F_:
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = tr.get("source")
        except Exception:
            pass

    out["config"]["placement"] = {"headline_buffers": OUT.how, "io_alloc_report": OUT.report}
    out["placement_setup_ms"] = None if not OUT.report else OUT.report.get("setup_ms")     # start-up cost of the placed set (not in any step)
    if coll_info:
        out["collective"] = coll_info
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None

    region.close()
    OUT.close()
    ctx.close()
    if world > 1:
        dist.barrier()
    if use_coll:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
