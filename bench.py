#!/usr/bin/env python
"""Headline benchmark: U2PL semi-supervised training step, ResNet-101 + DeepLabv3+,
769x769 Cityscapes-shaped synthetic crops, per-GPU batch 2 labeled + 2 unlabeled
(BASELINE.json configs[2]/[3]: drop-in for train_semi.py's hot loop), fp32.

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by torch.distributed.run, one rank per GPU, RCCL over xGMI)

A step = one full optimizer step: teacher eval fwd (pseudo labels), CutMix, student
fwd+bwd on 4 images, teacher train-mode fwd, OHEM sup loss (+aux), entropy /
exact-percentile reliability split, unsup CE, memory-bank contrastive loss,
gradient all-reduce, SGD, teacher EMA.  images/s = N * 4 / step time.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--crop", type=int, default=769)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharpen", type=float, default=4.0,
                    help="std of the calibrated teacher/student logits (see calibrate()): random-init logits are "
                         "near-uniform, which would leave the contrastive path (anchors need p>0.3) idle")
    ap.add_argument("--no-calibrate", action="store_true", help="round-1 workload: classifier last layer x sharpen only")
    ap.add_argument("--no-bank-prefill", action="store_true")
    ap.add_argument("--bf16", action="store_true",
                    help="BASELINE configs[4] (config 5): student convolutions on the bf16 matrix cores (fp32 accumulate, fp32 "
                         "master weights, fp32 EMA teacher); a SEPARATE line, never the headline (use with --crop 801)")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--no-direct-leg", action="store_true", help="skip ms_per_step_direct / conv_error_vs_f64 (N = 1 only)")
    ap.add_argument("--no-config5-leg", action="store_true", help="skip the short BASELINE configs[4] leg (801x801, bf16 student)")
    return ap.parse_args()


def synth_batch(B, S, C, device, gen):
    """SURVEY 8d synthetic step inputs: images N(0,1); labels block-constant randint on a
    (S/16+1)^2 grid nearest-upsampled, first 8 rows = 255."""
    img_l = torch.randn(B, 3, S, S, device=device, generator=gen)
    img_u = torch.randn(B, 3, S, S, device=device, generator=gen)
    gsz = S // 16 + 1
    coarse = torch.randint(0, C, (B, gsz, gsz), device=device, generator=gen)
    iy = (torch.arange(S, device=device) * gsz // S).clamp(max=gsz - 1)
    lab = coarse[:, iy][:, :, iy].contiguous()
    lab[:, :8] = 255
    return img_l, lab, img_u


def calibrate(model, teacher, calib, batches, sharpen):
    """Put the random-init networks into a TRAINED-LIKE state so that the loss path sees realistic inputs (there are no
    checkpoints or datasets offline): (1) BatchNorm running statistics = cumulative average over the synthetic batches
    (a fresh network's running stats are 0 / 1, which makes the eval-mode pseudo-label pass collapse onto 1-2 classes);
    (2) the classifier's last layer is standardised per class on the teacher's eval-mode logits (every class wins about
    1/19 of the pixels, logit std = `sharpen`: confident, class-balanced predictions); (3) student = teacher (as after EMA
    convergence); (4) the labeled targets are the teacher's own arg-max (predictions agree with the labels, like a trained
    model).  The conv / BN workload is unchanged; all 19 classes now yield anchors, prototypes and negative keys."""
    from u2pl_amd import nn as KN
    bns = [m for m in teacher.modules() if isinstance(m, KN.BatchNorm2d)]
    saved = [m.momentum for m in bns]
    teacher.train()
    with torch.no_grad():
        for k in range(4):
            for m in bns:
                m.momentum = 1.0 / (k + 1)
            il, _, iu = calib[k % len(calib)]
            teacher(torch.cat((il, iu)), need_aux=True)
        for m, mo in zip(bns, saved):
            m.momentum = mo
        teacher.eval()
        il, _, iu = calib[0]
        pred = teacher(torch.cat((il, iu)), need_aux=False, need_rep=False)["pred"].float()
        mu = pred.mean(dim=(0, 2, 3))
        sd = pred.std(dim=(0, 2, 3)).clamp_min(1e-6)
        last = teacher.decoder.classifier[8]
        last.weight.mul_((sharpen / sd).view(-1, 1, 1, 1))
        last.bias.copy_((last.bias - mu) * (sharpen / sd))
        model.load_state_dict(teacher.state_dict())
        from u2pl_amd import hipops as H
        out = []
        for il, ll, iu in batches:
            lab = H.pseudo_label(H.bilinear_up(teacher(il, need_aux=False, need_rep=False)["pred"], ll.shape[1:]))[1]
            lab[:, :8] = 255
            out.append((il, lab, iu))
    return out


_WD = {"thread": None, "phase": "setup", "step": -1, "last": time.time(), "done": False, "t_timed": None, "steps_done": 0}


def _progress(phase=None, step=None):
    """every phase change and every step is progress: the watchdog measures the time since the LAST one"""
    if phase is not None:
        _WD["phase"] = phase
    if step is not None:
        _WD["step"] = step
    _WD["last"] = time.time()


def _watchdog_arm(args, rank, world):
    """If the run makes no progress (no phase change, no completed step) for U2PL_BENCH_WATCHDOG_S seconds (default 600;
    multi-GPU runs only), EVERY stuck rank prints ONE JSON line that says where it stopped -- phase, step, its own ms per step
    over the timed steps it did complete, collectives issued, the communicator configuration -- and exits: a hang (e.g. a
    collective order mismatch on RCCL) then costs minutes, not the whole lease, and leaves a record.  A healthy but long
    run (first-time build, long diagnostics) is not killed: the limit applies to the time since the last progress mark."""
    import threading
    if world <= 1:
        return
    limit = float(os.environ.get("U2PL_BENCH_WATCHDOG_S", "600"))

    def watch():
        while not _WD["done"]:
            time.sleep(2.0)
            if _WD["done"] or time.time() - _WD["last"] <= limit:
                continue
            from u2pl_amd import nn as KN
            so_far = None
            if _WD["t_timed"] is not None and _WD["steps_done"] > 0:
                so_far = round((_WD["t_steps_end"] - _WD["t_timed"]) / _WD["steps_done"] * 1e3, 3)
            line = {"metric": "train images/sec at 769x769 (R101-DeepLabv3+)", "value": None, "unit": "images/s", "n_gpus": world,
                    "error": "watchdog: no progress for %.0f s" % (time.time() - _WD["last"]), "rank": rank, "phase": _WD["phase"],
                    "step": _WD["step"], "timed_steps_completed_by_this_rank": _WD["steps_done"],
                    "per_rank_ms": {str(rank): so_far}, "comm_exposed_ms": None,
                    "collectives_issued_by_this_rank": KN.COMM_DEBUG["issued"], "comm_stats": dict(KN.COMM_STATS),
                    "teacher_communicator": os.environ.get("U2PL_TEACHER_COMM", "0"),
                    "bucket_overlap": os.environ.get("U2PL_NO_BUCKET_OVERLAP") is None,
                    "hint": "rerun with U2PL_COMM_DEBUG=1 (per-step comparison of the ranks' collective sequences) and "
                            "U2PL_NO_BUCKET_OVERLAP=1 (all gradient buckets after backward)"}
            # ONE line on stdout (rank 0's, the one the driver parses); the other ranks report on stderr
            print(json.dumps(line), flush=True, file=sys.stdout if rank == 0 else sys.stderr)
            os._exit(3)

    t = threading.Thread(target=watch, daemon=True)
    t.start()
    _WD["thread"] = t


def wino_vs_direct_error(dev):
    """measured accuracy side of the Winograd / direct trade: the step's most frequent 3x3 layer
    (layer3: 256 -> 256, dilation 2, 97 x 97 maps) through both algorithms against a float64 convolution on the host"""
    import torch.nn.functional as F
    from u2pl_amd import nn as KN
    saved = dict(KN.CONV_ALGO)
    g = torch.Generator(device=dev).manual_seed(11)
    conv = KN.Conv2d(256, 256, 3, padding=2, dilation=2, bias=False).to(dev)
    x = torch.randn(1, 256, 97, 97, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    ref = F.conv2d(x.double().cpu(), conv.weight.detach().double().cpu(), padding=2, dilation=2)
    out = {}
    try:
        for name, w in (("winograd_f4", 4), ("direct", 0)):
            KN.CONV_ALGO.update(wino=w)
            with torch.no_grad():
                y = conv(x)
            out[name] = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
    finally:
        KN.CONV_ALGO.update(saved)
    return {"layer": "3x3 256->256 d=2 @97x97 (layer3 conv2)", "max_err_over_max_vs_f64": {k: float("%.3g" % v) for k, v in out.items()},
            "winograd_over_direct": round(out["winograd_f4"] / max(out["direct"], 1e-30), 2)}


def main():
    args = parse()
    from u2pl_amd import configs

    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: re-launch ourselves one rank per GPU (the driver calls torch.distributed.run directly)
        import subprocess
        port = 29500 + os.getpid() % 2000
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port)] + sys.argv))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    assert world == args.gpus or world == 1, (world, args.gpus)
    ndev = max(torch.cuda.device_count(), 1)
    local_rank %= ndev   # (tests share one GPU between two gloo ranks; production: one GPU per rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # U2PL_DIST_SINGLE=1 at N = 1: a process group of ONE rank on RCCL -- the step takes every multi-rank code path (SyncBatchNorm
    # exchanges, bucketed gradient all-reduce from the backward hooks, count / key all-gathers, meter reductions, no HIP graphs)
    # with every collective really issued and equal to the identity (u2pl_amd.comm.dist_active): what the N > 1 path costs in
    # launches and latency, measurable on a one-GPU box (the bandwidth terms of a real exchange are not in it)
    single_rccl = world == 1 and os.environ.get("U2PL_DIST_SINGLE", "0") == "1"
    multi = world > 1 or single_rccl
    if single_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if multi:
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a first run on RCCL must not be able to hang the lease silently: collectives time out, RCCL reports what it was doing,
        # and a host watchdog (below) prints a diagnostic JSON line and exits if the run stops making progress
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        os.environ.setdefault("TORCH_NCCL_DUMP_ON_TIMEOUT", "1")
        # "nccl" is RCCL on ROCm; U2PL_DIST_BACKEND=gloo only for the shared-GPU functional test
        dist.init_process_group(os.environ.get("U2PL_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("U2PL_COLLECTIVE_TIMEOUT_S", "120"))))
    _watchdog_arm(args, rank, world)

    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    if args.bf16:
        from u2pl_amd import nn as KN0
        KN0.CONV_ALGO["bf16"] = 1

    torch.manual_seed(2)
    np.random.seed(2)
    cfg = configs.cityscapes_semi(arch=args.arch, crop=args.crop, batch_size=args.batch, sync_bn=True)
    C = cfg["net"]["num_classes"]
    model = ModelBuilder(cfg["net"]).to(dev)
    teacher = ModelBuilder(cfg["net"]).to(dev)
    if args.no_calibrate:
        with torch.no_grad():
            for m in (model, teacher):
                m.decoder.classifier[8].weight.mul_(args.sharpen)
    trainer = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
    if not args.no_bank_prefill:   # steady state of BASELINE configs[3]: queues at capacity (30000 x 256; class 0: 50000)
        gb = torch.Generator(device=dev).manual_seed(7)
        for c in range(C):
            trainer.memobank.load_logical(c, torch.randn(trainer.memobank.cap[c], 256, device=dev, generator=gb))
    gen = torch.Generator(device=dev).manual_seed(2 + rank)
    batches = [synth_batch(args.batch, args.crop, C, dev, gen) for _ in range(2)]
    if not args.no_calibrate:
        # A synthetic network is fragile: ONE SGD step at the reference's lr 0.01 on noise images collapses the
        # pseudo-labels onto a single class (measured), which would idle the contrastive path from the second step on.
        # The learning rate is a scalar argument of the same SGD launch: 1e-6 keeps the calibrated state through
        # warm-up and timing without changing the work of a step.
        trainer.base_lr = 1e-6
        # weights are calibrated on batches that are IDENTICAL on every rank (replicas must stay bit-identical);
        # each rank's own batches only get their labeled targets from the calibrated teacher
        gc = torch.Generator(device=dev).manual_seed(1234)
        calib = [synth_batch(args.batch, args.crop, C, dev, gc) for _ in range(2)]
        batches = calibrate(model, teacher, calib, batches, args.sharpen)

    # every train_step of this process is counted (tools/parse_pmc_traffic.py divides a profiled run's loss-path traffic by it)
    n_steps_total = [0]
    _ts = trainer.train_step

    def _counted_step(*a, **k):
        n_steps_total[0] += 1
        return _ts(*a, **k)
    trainer.train_step = _counted_step

    def step(i):
        il, ll, iu = batches[i % len(batches)]
        _progress(step=i)
        return trainer.train_step(il, ll, iu, epoch=1)

    _progress("warmup")
    # HIP graphs capture a static segment at its (WARM + 1)-th execution (u2pl_amd/graphs.py): with fewer warm-up steps than
    # that, the capture would land inside the timed region -- run the missing ones here, un-timed, before the W warm-up steps
    from u2pl_amd import graphs as GRw
    priming = max(0, GRw.WARM + 1 - args.warmup) if GRw.enabled() else 0
    for i in range(priming):
        step(i)
    for i in range(args.warmup):
        step(priming + i)
    torch.cuda.synchronize()
    from u2pl_amd import nn as KN
    from u2pl_amd import hipops as HO
    rccl_ranks = None
    if multi:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                      # an ACTUAL collective on the process group: the ranks it reached
        rccl_ranks = int(ones.item())
        dist.barrier()
    torch.cuda.synchronize()
    comm0 = dict(KN.COMM_STATS)
    route0 = HO.split_route_stats()
    from u2pl_amd import _lib as LIBc, graphs as GR
    from u2pl_amd.utils import utils as UU
    calls0, launches0, gstats0, blocked0 = LIBc.CALLS[0], LIBc.lib().cdll.u2pl_kernel_launches(), dict(GR.STATS), UU.BLOCKED_S[0]
    _progress("timed")
    t0 = time.perf_counter()
    _WD["t_timed"] = _WD["t_steps_end"] = time.time()
    host_s = 0.0       # time the Python thread spends inside train_step (enqueue + the step's one blocking read)
    for i in range(args.steps):
        th = time.perf_counter()
        meters = step(args.warmup + i)
        host_s += time.perf_counter() - th
        _WD["steps_done"], _WD["t_steps_end"] = i + 1, time.time()
    calls_timed = (LIBc.CALLS[0] - calls0) / args.steps
    launches_timed = (LIBc.lib().cdll.u2pl_kernel_launches() - launches0) / args.steps
    replays_timed = (GR.STATS["replays"] - gstats0["replays"]) / args.steps
    blocked_s = UU.BLOCKED_S[0] - blocked0
    host_tail = time.perf_counter()
    torch.cuda.synchronize()
    host_tail = time.perf_counter() - host_tail      # GPU work still queued when the host left the last step
    own_dt = time.perf_counter() - t0                # this rank alone: its K steps + its own queue drained, before the barrier
    _progress("timed-barrier")
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    per_rank_ms = [round(own_dt / args.steps * 1e3, 3)]
    if multi:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        own = torch.tensor([own_dt / args.steps * 1e3], device=dev, dtype=torch.float64)
        allv = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(allv, own)
        per_rank_ms = [round(float(v), 3) for v in allv]
    dt = float(tt)
    _progress("diagnostics")
    ms = dt / args.steps * 1e3
    imgs = world * 2 * args.batch
    value = imgs / (ms / 1e3)
    comm = {k: (KN.COMM_STATS.get(k, 0) - comm0.get(k, 0)) / args.steps for k in set(comm0) | set(KN.COMM_STATS)}
    route1 = HO.split_route_stats()
    from u2pl_amd import roofline as RL

    def timed_steps(n, first, fn=None):
        """n more steps, bracketed like the timed region (diagnostic lines below: not the headline)"""
        fn = fn or step
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for i in range(n):
            fn(first + i)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        _progress()
        tb = torch.tensor([time.perf_counter() - ta], device=dev, dtype=torch.float64)
        if multi:
            dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        return float(tb) / n * 1e3

    diag = {}
    if multi:
        # exposed communication of the bucketed gradient all-reduce: the same steps with every bucket launched AFTER backward
        os.environ["U2PL_NO_BUCKET_OVERLAP"] = "1"
        diag["ms_per_step_no_bucket_overlap"] = round(timed_steps(3, args.warmup + args.steps), 3)
        del os.environ["U2PL_NO_BUCKET_OVERLAP"]
        diag["comm_exposed_ms"] = round(diag["ms_per_step_no_bucket_overlap"] - ms, 3)

    diag["per_rank_ms"] = per_rank_ms       # each rank's own K steps + queue drain, before the closing barrier (max = the headline's clock)
    if world == 1:
        diag["comm_exposed_ms"] = 0.0
    # per-phase time (the reference's meters: train_semi.py:563-587): ONE extra step with the side streams folded
    # into the main stream (teacher passes, weight gradients) and an event at every phase boundary.  Serialised, the phases
    # add up to more than the overlapped step (`phase_sum_ms` vs `ms_per_step`): what the side streams hide is the difference.
    KNp = KN
    trainer.train_step(*batches[0], epoch=1)     # (warm: same allocator / operand state as the timed steps)
    torch.cuda.synchronize()
    saved_side, saved_wg = getattr(trainer, "_side", None), KNp._WGRAD["enabled"]
    trainer._side, KNp._WGRAD["enabled"] = torch.cuda.current_stream(), False
    try:
        # (this configuration has its own graph key -- the weight-gradient fork is part of it -- so the step runs eagerly: one
        #  un-timed step first, which lets the caching allocator grow to the eager step's working set outside the measurement)
        trainer.train_step(*batches[0], epoch=1)
        torch.cuda.synchronize()
        trainer.phase_log = []
        trainer.train_step(*batches[1 % len(batches)], epoch=1)
        torch.cuda.synchronize()
        log = trainer.phase_log
    finally:
        trainer.phase_log = None
        trainer._side, KNp._WGRAD["enabled"] = saved_side, saved_wg
    phase_ms = {b[0]: round(a[1].elapsed_time(b[1]), 3) for a, b in zip(log[:-1], log[1:])}
    diag["phase_ms"] = dict(phase_ms, comm_exposed=diag.get("comm_exposed_ms"))
    diag["phase_sum_ms"] = round(sum(phase_ms.values()), 3)
    diag["phase_ms_note"] = ("one extra step serialised on ONE stream (teacher passes and weight gradients folded into the main stream), "
                             "HIP events at the phase boundaries; teacher_eval = pseudo-label pass + CutMix, student_fwd includes the "
                             "supervised loss heads, contrastive = unsupervised CE + bank + InfoNCE forward, bwd = whole backward incl. "
                             "weight gradients [+ bucket all-reduce launches], opt_ema = all-reduce join + SGD + EMA + operand re-split")
    _progress()
    if multi:
        dist.barrier()

    # the roofline leg runs ONE extra (un-timed) step with per-call HIP events; it contains the step's
    # collectives, so every rank executes it
    KN.AMAX_STATS.update(fused=0, standalone=0)
    roof = RL.measure(trainer, batches[0], args, ms)
    # split-fp16 operand maxima of that (eager) step: how many came out of their producers' own launches, how many needed a pass
    roof["operand_maxima_per_step"] = dict(KN.AMAX_STATS, conv_h=bool(KN.CONV_H["on"]))
    _progress()
    if multi:
        dist.barrier()
    if not args.no_calibrate:
        # the timed steps ran at lr 1e-6 (see above).  FIVE steps at the configuration's lr 0.01, each from the re-instated
        # calibrated state (weights, momentum, teacher, BatchNorm buffers restored and the derived operands rebuilt, un-timed:
        # the step after an lr-0.01 step would see collapsed pseudo-labels), through the same graph replays as the timed steps:
        # the scalar does not change the time of a step
        import statistics as _st
        bufs = [b for mdl in (model, teacher) for b in mdl.buffers()]
        snap = dict(s=trainer.arena.flat.clone(), t=trainer.t_arena.flat.clone(),
                    m=None if trainer.arena.momentum_buf is None else trainer.arena.momentum_buf.clone(),
                    b=[b.clone() for b in bufs])

        def restore():
            trainer.arena.flat.copy_(snap["s"])
            trainer.t_arena.flat.copy_(snap["t"])
            if snap["m"] is not None:
                trainer.arena.momentum_buf.copy_(snap["m"])
            for b, b0 in zip(bufs, snap["b"]):
                b.copy_(b0)
            KN.invalidate_weights()
            KN.presplit(trainer.arena.params, trainer.arena)
            KN.presplit(trainer.t_arena.params, trainer.t_arena)
            torch.cuda.synchronize()

        trainer.base_lr = cfg["trainer"]["optimizer"]["kwargs"]["lr"]
        samples = []
        for i in range(5):
            restore()
            samples.append(round(timed_steps(1, i), 3))
        trainer.base_lr = 1e-6
        restore()
        diag["ms_step_lr0.01"] = round(_st.median(samples), 3)
        diag["ms_step_lr0.01_samples"] = samples
        # the reference's FIRST epoch (sup_only_epoch = 0: teacher <- student aliasing around every step, train_semi.py:309-315 --
        # two more 267 MB copies + a re-split of the teacher's operands per step) next to the headline's ordinary steps (ADVICE r5:
        # rounds <= 4 timed these, the CPU baseline still runs them)

        def step_e0(i):
            il, ll, iu = batches[i % len(batches)]
            return trainer.train_step(il, ll, iu, epoch=0)

        step_e0(0)
        diag["ms_per_step_epoch0_aliasing"] = round(timed_steps(3, 1, step_e0), 3)
        restore()
    wt = KN.CONV_ALGO["wino"]
    if wt in (2, 4) and not args.bf16 and world == 1 and not args.no_direct_leg:
        # the accuracy / speed trade of the default algorithm, in the line itself: the SAME workload with
        # every 3x3 layer on the direct implicit-GEMM kernel (U2PL_CONV_WINO=0), and both algorithms' error against float64
        KN.CONV_ALGO["wino"] = 0
        g_env = os.environ.get("U2PL_GRAPHS")
        os.environ["U2PL_GRAPHS"] = "0"          # eager steps for this leg (graphs are keyed on the algorithm; no capture inside the timing)
        try:
            step(0)                              # builds the direct kernels' operands (3x3 split planes), un-timed
            diag["ms_per_step_direct"] = round(timed_steps(3, 1), 3)
        finally:
            KN.CONV_ALGO["wino"] = wt
            if g_env is None:
                del os.environ["U2PL_GRAPHS"]
            else:
                os.environ["U2PL_GRAPHS"] = g_env
        step(0)
        torch.cuda.synchronize()
        diag["conv_error_vs_f64"] = wino_vs_direct_error(dev)
        _progress()
    conv_algo = ("student: bf16-operand implicit GEMM on v_mfma_f32_32x32x16_bf16 (fwd, dgrad, wgrad; direct, no Winograd); "
                 "teacher: fp32 as in the headline" if args.bf16 else
                 "fp32 implicit GEMM for every layer (U2PL_CONV_WINO=0)" if wt not in (2, 4) else
                 f"fp32 Winograd F({wt}x{wt},3x3) for the stride-1 3x3 layers whose tile padding leaves >= "
                 f"{KN.CONV_ALGO['min_gain']}x fewer multiplies (forward, data and weight gradients; component products on "
                 "the same implicit-GEMM kernel), direct fp32 implicit GEMM elsewhere; U2PL_CONV_WINO=0|2|4 selects")
    from u2pl_amd import roofline as _RLq
    if not args.bf16:
        conv_algo += ("; fp32 products: " + (("THREE fp16 piece products of a two-piece split of the per-tensor power-of-two-scaled "
                      "operands, fp32 accumulate on the 16-bit matrix cores (v_mfma_f32_32x32x16_f16; the layers with <= 64 output "
                      "channels and U2PL_CONV_H=0: " if KN.CONV_H["on"] else "(") + "exact three-way bf16 split of both operands, six "
                      "piece products accumulated in fp32 on the bf16 matrix cores); error vs float64 <= the fp32 MFMA path's "
                      "(tools/bench_igemm_wsh.py, tests/test_gpu_igemm_ws.py; U2PL_CONV_SPLIT=0 selects v_mfma_f32_32x32x2_f32)"
                      if _RLq.conv_split_on() else "v_mfma_f32_32x32x2_f32 (U2PL_CONV_SPLIT=0)"))
    if rank == 0:
        out = {
            "metric": "train images/sec at %dx%d (R101-DeepLabv3+)" % (args.crop, args.crop), "value": round(value, 4), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (student conv operands; fp32 accumulate / tensors / master weights / teacher)" if args.bf16 else "f32",
            "data": ("synthetic (N(0,1) images; random-init weights put into a trained-like state: BN running statistics and "
                     "the classifier's last layer calibrated on the synthetic batches to confident, class-balanced "
                     "predictions with logit std %g, student = teacher, labeled targets = teacher arg-max; lr 1e-6 so that "
                     "state persists over the timed steps -- same launches as at lr 0.01; steps run as epoch 1 of 200 = the "
                     "ordinary semi-supervised step: in epoch 0 = sup_only_epoch the reference additionally aliases teacher <- "
                     "student around every step, train_semi.py:309-315, two more 267 MB copies + operand re-splits)" % args.sharpen)
                    if not args.no_calibrate else
                    "synthetic (N(0,1) images, block labels; random-init weights, classifier last layer x%g)" % args.sharpen,
            "config": {"workload": f"U2PL semi {args.arch}-DeepLabv3+ Cityscapes-shaped {args.crop}x{args.crop}, per-GPU "
                                   f"batch {args.batch} labeled + {args.batch} unlabeled, C=19, OHEM+aux, cutmix, "
                                   "contrastive bank 30000x256 pre-filled (BASELINE configs[2]/[3])",
                       "global_batch": imgs, "parallelism": f"dp{world}", "conv_algo": conv_algo},
            "losses_last_step": [round(float(x), 5) for x in meters.cpu()],
            # collectives issued per step by this rank (0 at N = 1: SyncBatchNorm exchanges only happen under a process group)
            "syncbn_collectives_per_step": comm["syncbn_allreduce"], "bucket_allreduces_per_step": comm["bucket_allreduce"],
            "rccl_ranks": rccl_ranks, "rccl_world_of_one": bool(single_rccl),
            "emulated_collective_latency_us": float(os.environ.get("U2PL_EMULATE_COLL_US", "0") or 0) if multi else 0.0,
            "split_launches_timed": route1[0] - route0[0], "split_second_barrier_launches_timed": route1[1] - route0[1],
            "split_ledger_fallbacks": HO.SPLIT_FALLBACKS["ledger"],
            # host side of the timed steps (rank 0): wall time inside train_step per step -- the enqueue of ~3000 C-ABI calls plus
            # the step's one blocking device-to-host read -- and the GPU work still queued when the host left the last step
            # (tools/host_overhead.py has the cProfile breakdown).  host_enqueue_ms close to ms_per_step with a small tail =
            # the host is the bound; well below it = the GPU is
            # host_in_step_ms = wall time inside train_step; host_blocked_ms = the part of it spent inside the step's one blocking
            # device-to-host read (the host waiting for the GPU); host_enqueue_ms = the difference = what the host needs to issue a step
            "host_in_step_ms": round(host_s / args.steps * 1e3, 2), "host_blocked_ms": round(blocked_s / args.steps * 1e3, 2),
            "host_enqueue_ms": round((host_s - blocked_s) / args.steps * 1e3, 2),
            "gpu_tail_after_last_enqueue_ms": round(host_tail * 1e3, 2),
        }
        # the roofline objects right behind the contract's keys (a truncated key list still shows them)
        head = {k: roof.pop(k) for k in ("roofline", "roofline_hbm", "roofline_wgrad", "roofline_bf16") if k in roof}
        out = {**{k: out[k] for k in list(out)[:13]}, **head, **{k: out[k] for k in list(out)[13:]}}
        out.update(diag)
        out.update(roof)
        # calls / launches of the TIMED steps (a HIP-graph replay of a static segment is one hipGraphLaunch, not a C-ABI call); the
        # roofline leg's counts come from its eager per-call profile step and are kept under *_eager_profile
        for k in ("abi_calls_per_step", "kernel_launches_per_step"):
            if k in out:
                out[k + "_eager_profile"] = out.pop(k)
        out["abi_calls_per_step"] = round(calls_timed, 1)
        out["kernel_launches_issued_by_host_per_step"] = round(launches_timed, 1)
        out["graph_replays_per_step"] = round(replays_timed, 2)
        out["train_steps_in_process"] = n_steps_total[0]
        out["graphs"] = dict(GR.STATS, enabled=GR.enabled(), warm=GR.WARM, priming_steps_before_warmup=priming,
                             segments="teacher eval pass, teacher train pass, student forward, student backward")
        if not args.no_cpu_baseline and world == 1:   # CPU baseline: rank 0 at N=1 only (the checker's port, never the product)
            from oracle import step_ref
            out["cpu_baseline"] = step_ref.timed_cpu_baseline(crop=args.crop, arch=args.arch, batch=args.batch)
            import glob
            ref_ts = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_cpu_reference_timing.json")))
            ref_t = ref_ts[-1] if ref_ts else ""
            if os.path.exists(ref_t):
                rt = json.load(open(ref_t))
                if rt.get("port_images_per_s") and rt.get("reference_images_per_s"):
                    # bridge between the two CPU numbers: port / reference on the SAME machine (build container)
                    out["cpu_baseline"]["port_over_reference"] = round(rt["port_images_per_s"] / rt["reference_images_per_s"], 3)
                out["cpu_baseline"]["reference_train_in_build_container"] = {
                    "images_per_s": round(rt["reference_images_per_s"], 5), "cores": rt["cores"], "file": "profiles/" + os.path.basename(ref_t),
                    "note": "the reference's own train_semi.train() through oracle/ref_shim.py, 1 warm-up + 2 timed steps, "
                            "same configuration; /root/reference does not exist on the GPU box"}
        if world == 1 and not args.bf16 and not args.no_config5_leg and args.crop == 769:
            # BASELINE configs[4] ("config 5": 801x801, reduced-precision student, fp32 EMA teacher) as a short leg of the default
            # run, so that the driver's line carries it; own process, after the headline measurement
            import gc
            import subprocess
            try:
                # the child needs ~100 GB of its own: hand this process's cached blocks and graph pools back first
                trainer.__dict__.pop("_graph_cache", None)
                gc.collect()
                torch.cuda.empty_cache()
                r5 = subprocess.run([sys.executable, os.path.abspath(__file__), "--bf16", "--crop", "801", "--steps", "4", "--warmup", "3",
                                     "--no-cpu-baseline", "--no-config5-leg"], capture_output=True, text=True, timeout=600)
                l5 = [ln for ln in r5.stdout.splitlines() if ln.startswith("{")]
                if not l5:
                    raise RuntimeError("no JSON line; rc %s; stderr tail: %s" % (r5.returncode, r5.stderr[-600:]))
                j5 = json.loads(l5[-1])
                out["config5"] = {"images_per_s": j5["value"], "ms_per_step": j5["ms_per_step"], "crop": 801, "steps": j5["steps"],
                                  "warmup": j5["warmup"], "dtype": j5["dtype"], "conv_algo": j5["config"]["conv_algo"],
                                  "roofline_bf16": j5.get("roofline_bf16"),
                                  "note": "python bench.py --bf16 --crop 801: student conv operands rounded to bf16 while staged into LDS "
                                          "(bf16 matrix cores, fp32 accumulate), activations still fp32 in HBM; teacher as in the headline"}
            except Exception as e:      # the leg must never cost the headline line
                out["config5"] = {"error": repr(e)[:900]}
        _WD["done"] = True
        front = [k for k in list(out)[:13]] + [k for k in ("roofline", "cpu_baseline", "roofline_hbm", "roofline_wgrad", "roofline_bf16") if k in out]
        out = {**{k: out[k] for k in front}, **{k: v for k, v in out.items() if k not in front}}
        # RCCL's version banner (NCCL_DEBUG=WARN above) sits in the C library's stdout buffer until the process exits: flush it
        # first, so that the JSON line is the LAST line of rank 0's stdout
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    _WD["done"] = True
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
