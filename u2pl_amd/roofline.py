"""Roofline accounting for bench.py: per-entry-point HIP-event timing of one extra
(un-timed) training step on the stream the kernels run on, algorithmic FLOPs /
bytes from SURVEY.md section 8(d), peaks from MI355X_MICROARCH.md."""
import os
import time

import torch

from . import _lib

PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6290 GB/s measured float4 copy)
ALGO_TFLOP_PER_IMAGE_769 = 6.60   # 26.4 TFLOP per step of 2+2 images at 769^2 (direct-convolution count)

# positions of the geometry ints inside each conv entry point's argument list
_CONV_GEOM = {
    "u2pl_conv2d_fwd_f32": 6, "u2pl_conv2d_fwd_bnstats_f32": 6, "u2pl_conv2d_fwd_bnact_f32": 6, "u2pl_conv2d_dgrad_f32": 5, "u2pl_conv2d_wgrad_f32": 7,
    "u2pl_conv2d_fwd_bf16op_f32": 6, "u2pl_conv2d_fwd_bnstats_bf16op_f32": 6, "u2pl_conv2d_dgrad_bf16op_f32": 5,
    "u2pl_conv2d_wgrad_bf16op_f32": 7,
    "u2pl_conv2d_fwd_ws_f32": 6, "u2pl_conv2d_fwd_bnstats_ws_f32": 6, "u2pl_conv2d_fwd_bnact_ws_f32": 6, "u2pl_conv2d_dgrad_ws_f32": 5,
    # split-fp16 (round 6): the operand-maximum pointer shifts the geometry by one (two in the weight gradient)
    "u2pl_conv2d_fwd_wsh_f32": 7, "u2pl_conv2d_fwd_bnstats_wsh_f32": 7, "u2pl_conv2d_fwd_bnact_wsh_f32": 7, "u2pl_conv2d_dgrad_wsh_f32": 6,
    "u2pl_conv2d_wgrad_h_f32": 9,
}
# entry points whose fp32 products are THREE fp16 piece products (csrc/conv_geom.h "split-fp16"); every other split entry point: six
H_NAMES = ("u2pl_conv2d_fwd_wsh_f32", "u2pl_conv2d_fwd_bnstats_wsh_f32", "u2pl_conv2d_fwd_bnact_wsh_f32", "u2pl_conv2d_dgrad_wsh_f32",
           "u2pl_gemm_batched_wsh_f32", "u2pl_conv2d_wgrad_h_f32", "u2pl_wgrad_batched_h_f32")


def piece_products(name):
    return 3 if name in H_NAMES else 6
PEAK_BF16_MFMA_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16, dense
# split fp32 (csrc/conv.hip BF == 3): one fp32 product = six bf16 piece products -> the matrix-pipe bound of the algorithm in
# fp32-equivalent FLOP/s
PEAK_SPLIT_F32_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def conv_split_on():
    return bool(_lib.lib().cdll.u2pl_conv_get_split())


def _mfma_fields(ach, pipe_flops=None, flops=None):
    """roofline fields of an fp32 GEMM-like group: against the pipe its instructions actually run on.  pipe_flops / flops: the
    matrix-pipe FLOPs the group's launches issue (executed fp32 FLOPs x piece products per product: 3 for the fp16 split, 6 for
    the bf16 split) over their executed fp32 FLOPs -- the group's fp32-equivalent bound is 2500 / that ratio"""
    if conv_split_on():
        ratio = (pipe_flops / flops) if (pipe_flops and flops) else 6.0
        peak = PEAK_BF16_MFMA_TFLOPS / ratio
        return {"peak": round(peak, 1), "frac": round(ach / peak, 4),
                "frac_vs_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "piece_products_per_fp32_product": round(ratio, 3),
                "matrix_pipe_tflops": round(ach * ratio, 1),
                # the same fp32-equivalent rate against the bound rounds 3-5 quoted (six bf16 piece products: 2500 / 6 = 416.7 TF)
                "frac_vs_six_product_bound": round(ach / PEAK_SPLIT_F32_TFLOPS, 4),
                "arithmetic": "fp32 products on the 16-bit matrix cores, fp32 accumulate: THREE fp16 piece products of a two-piece "
                              "split of the power-of-two-scaled operands (v_mfma_f32_32x32x16_f16; default since round 6, "
                              "csrc/conv_geom.h) or SIX bf16 piece products of an exact three-piece split "
                              "(v_mfma_f32_32x32x16_bf16: U2PL_CONV_H=0, and the <= 64-channel layers); U2PL_CONV_SPLIT=0: "
                              "v_mfma_f32_32x32x2_f32.  achieved = fp32-equivalent FLOP/s; peak = 2500 TF (bf16 = fp16 dense) / "
                              "piece_products_per_fp32_product of the launches; frac = matrix-pipe FLOP/s over 2500; "
                              "frac_vs_fp32_mfma_peak = achieved / 157.3"}
    return {"peak": PEAK_F32_MFMA_TFLOPS, "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "arithmetic": "v_mfma_f32_32x32x2_f32 (U2PL_CONV_SPLIT=0)"}


def _conv_flops(name, args):
    if name == "u2pl_gemm_batched_f32":      # (x, ldx, zx, w, zw, y, ldy, zy, M, K, Nn, batch)
        M, K, Nn, batch = args[8:12]
        return 2.0 * M * K * Nn * batch
    if name == "u2pl_gemm_batched_ws_f32":   # (x, ldx, zx, wsplit, y, ldy, zy, M, K, Nn, batch)
        M, K, Nn, batch = args[7:11]
        return 2.0 * M * K * Nn * batch
    if name == "u2pl_gemm_batched_wsh_f32":  # (x, ldx, zx, x_amax, wsplit, y, ldy, zy, M, K, Nn, batch)
        M, K, Nn, batch = args[8:12]
        return 2.0 * M * K * Nn * batch
    if name == "u2pl_wgrad_batched_f32":     # (dy, lddy, zdy, x, ldx, zx, part, M, Cin, Cout, batch)
        M, Cin, Cout, batch = args[7:11]
        return 2.0 * M * Cin * Cout * batch
    if name == "u2pl_wgrad_batched_h_f32":   # (dy, lddy, zdy, dy_amax, x, ldx, zx, x_amax, part, M, Cin, Cout, batch)
        M, Cin, Cout, batch = args[9:13]
        return 2.0 * M * Cin * Cout * batch
    i = _CONV_GEOM[name]
    N, Hin, Win, Cin, Hout, Wout, Cout, R, S = args[i:i + 9]
    return 2.0 * N * Hout * Wout * Cout * R * S * Cin


def _conv_bytes(name, args):
    """ALGORITHMIC HBM bytes of one igemm-group launch: the A operand read once (an R x S gather re-reads overlapping
    pixels from cache, not from memory), the B operand once (fp32 weights: 4 B/element; pre-split planes: 6 B), the
    output written once.  What the kernel moves above this figure is re-reads of A across column tiles / of B across
    row tiles that L2 / MALL did not absorb."""
    if name in ("u2pl_gemm_batched_f32", "u2pl_gemm_batched_ws_f32", "u2pl_gemm_batched_wsh_f32"):
        M, K, Nn, batch = args[7:11] if name == "u2pl_gemm_batched_ws_f32" else args[8:12]
        wb = 6 if name.endswith("_ws_f32") else 4           # (three bf16 planes: 6 B per weight; two fp16 planes or fp32: 4 B)
        return batch * (4.0 * M * K + wb * Nn * K + 4.0 * M * Nn)
    i = _CONV_GEOM[name]
    N, Hin, Win, Cin, Hout, Wout, Cout, R, S = args[i:i + 9]
    wb = 6 if "_ws_" in name else 4
    return 4.0 * N * Hin * Win * Cin + wb * Cout * R * S * Cin + 4.0 * N * Hout * Wout * Cout


def profile_step(step_fn):
    _lib.PROFILE = []
    try:
        step_fn()
        torch.cuda.synchronize()
        rec = _lib.PROFILE
    finally:
        _lib.PROFILE = None
    agg, shapes = {}, {}
    agg["_dense"] = replay_dense(rec)
    for name, args, e0, e1, nk, full in [r[:4] + (r[7], r[6]) for r in rec]:
        d = agg.setdefault(name, dict(ms=0.0, calls=0, flops=0.0, kernels=0, bytes=0.0))
        if name in MFMA_GROUPS["igemm"]:
            d["bytes"] += _conv_bytes(name, args)
            ri = 23 if "_wsh_" in name else 22
            if "_bnact" in name and len(full) > ri and full[ri] is not None:
                # the epilogue's residual operand is read once too (eval-mode identity; round 5: the residual GRADIENT that the
                # pointwise data-gradient launch adds)
                i = _CONV_GEOM[name]
                d["bytes"] += 4.0 * args[i] * args[i + 4] * args[i + 5] * args[i + 6]
        ms = e0.elapsed_time(e1)
        d["ms"] += ms
        d["calls"] += 1
        d["kernels"] += nk
        if name in ("u2pl_wgrad_batched_f32", "u2pl_wgrad_batched_h_f32"):
            d["flops"] += _conv_flops(name, args)
        elif name in ("u2pl_gemm_batched_f32", "u2pl_gemm_batched_ws_f32", "u2pl_gemm_batched_wsh_f32"):
            fl = _conv_flops(name, args)
            d["flops"] += fl
            o = 7 if name == "u2pl_gemm_batched_ws_f32" else 8
            key = ("wino_gemm", args[o + 3], args[o], 1, args[o + 1], args[o], 1, args[o + 2], 1, 1, 1, 0, 1)
            sd = shapes.setdefault(key, dict(ms=0.0, calls=0, flops=0.0))
            sd["ms"] += ms
            sd["calls"] += 1
            sd["flops"] += fl
        elif name in _CONV_GEOM:
            fl = _conv_flops(name, args)
            d["flops"] += fl
            i = _CONV_GEOM[name]
            key = (name[12:-4].replace("fwd_bnstats", "fwd").replace("fwd_bnact", "fwd").replace("_bf16op", "@bf16").replace("_wsh", "@wsh").replace("_ws", "@ws"),) + tuple(args[i:i + 9]) + tuple(args[i + 9:i + 12])
            sd = shapes.setdefault(key, dict(ms=0.0, calls=0, flops=0.0))
            sd["ms"] += ms
            sd["calls"] += 1
            sd["flops"] += fl
    agg["_shapes"] = shapes
    return agg


MFMA_GROUPS = {
    "igemm": ("u2pl_conv2d_fwd_f32", "u2pl_conv2d_fwd_bnstats_f32", "u2pl_conv2d_fwd_bnact_f32", "u2pl_conv2d_dgrad_f32",
              "u2pl_gemm_batched_f32", "u2pl_conv2d_fwd_ws_f32", "u2pl_conv2d_fwd_bnstats_ws_f32", "u2pl_conv2d_fwd_bnact_ws_f32",
              "u2pl_conv2d_dgrad_ws_f32", "u2pl_gemm_batched_ws_f32",
              "u2pl_conv2d_fwd_wsh_f32", "u2pl_conv2d_fwd_bnstats_wsh_f32", "u2pl_conv2d_fwd_bnact_wsh_f32",
              "u2pl_conv2d_dgrad_wsh_f32", "u2pl_gemm_batched_wsh_f32"),
    "wgrad": ("u2pl_conv2d_wgrad_f32", "u2pl_wgrad_batched_f32", "u2pl_conv2d_wgrad_h_f32", "u2pl_wgrad_batched_h_f32"),
    "bf16": ("u2pl_conv2d_fwd_bf16op_f32", "u2pl_conv2d_fwd_bnstats_bf16op_f32", "u2pl_conv2d_dgrad_bf16op_f32",
             "u2pl_conv2d_wgrad_bf16op_f32"),
}


def replay_dense(rec):
    """SUSTAINED-load time of the MFMA-bound launches of the profiled step: every recorded call of a group is re-issued
    back to back (same arguments, same buffers) behind a spinning kernel, ONE HIP-event pair around the whole train.
    The per-call event pairs of profile_step leave idle gaps between the kernels (two marker packets + Python per
    call), in which the chip boosts its clock: they read ~11 % shorter than the same kernels in steady state, which is
    what rocprofv3 sees over a run (profiles/README.md) -- this replay is the figure that agrees with it."""
    out = {}
    for gname, names in MFMA_GROUPS.items():
        calls = [(r[4], r[5]) for r in rec if r[0] in names]
        if not calls:
            continue
        sp = _lib.stream_ptr()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(40_000_000)
        e0.record()
        for fn, conv in calls:
            fn(*conv, sp)
        e1.record()
        torch.cuda.synchronize()
        out[gname] = e0.elapsed_time(e1)
    return out


def replay_hbm_group(reps=10):
    """GPU time of the HBM-bound group (entropy + exact select + masks; classify / compaction / prototypes; bank
    append; InfoNCE forward + gradient scatter) on the LAST step's own tensors: every stage is re-issued `reps`
    times behind a spinning kernel, so the host has finished enqueueing before the first timed kernel starts and
    ONE HIP-event pair brackets each stage's launches on the stream they run on (per-call event pairs would add
    two marker packets, ~20 us, to kernels that take 5-30 us)."""
    from . import hipops as H

    R = H.REPLAY or {}
    out = {}

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(60_000_000)      # ~30 ms of GPU spin: the host enqueues the whole replay meanwhile
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3    # us per repetition

    saved, H.REPLAY = H.REPLAY, None
    try:
        if "rel" in R:
            tl, hw, label_l, label_u, low_shape, percents, neg_high = R["rel"]
            out["reliability_us"] = timed(lambda: H.reliability_split(tl, hw, label_l, label_u, low_shape, percents,
                                                                      negative_high_entropy=neg_high))
        if "phase1" in R:
            out["phase1_us"] = timed(lambda: H.contra_phase1(*R["phase1"]))
        if "append" in R:
            dd, n, D, ld, mx = R["append"][:5]
            out["bank_append_us"] = timed(lambda: _lib.call("u2pl_bank_append_multi_f32", dd, n, D, ld, mx))
        if "enqueue" in R and R.get("bank") is not None:
            # the device-resident enqueue of the last step, replayed on a scratch copy of the ring state and a scratch
            # storage buffer (a replay on the real bank would append the step's keys ten more times)
            bank = R["bank"]
            rows, ld, idx, idx_stride, counts_dev = R["enqueue"]
            st0 = bank.state.clone()
            scratch = torch.empty_like(bank.storage)
            st = st0.clone()

            def enq():
                _lib.call("u2pl_bank_enqueue_f32", st, scratch, bank.D, rows, ld, idx, idx_stride, None, counts_dev, bank.C)
            out["bank_append_us"] = timed(enq)
            del scratch
        if "infonce" in R:
            rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups, keep = R["infonce"]

            def nce():
                H.infonce_kernels_once(rep_rows, jobs_dev, njobs, Q, K, temp, valid_seg, groups)
            out["infonce_fwd_bwd_us"] = timed(nce)
    finally:
        H.REPLAY = saved
    return out


def split_phase_table():
    """mean time of block 0 in every phase of the persistent reliability split over all launches of this process (stamps
    the kernel keeps in its workspace: csrc/relfused.hip), microseconds"""
    from . import hipops as H
    names = [(18, "A_tile_and_labels"), (19, "A_entropies"), (10, "A_owner_hist"), (11, "P_scan_prefix_totals"), (12, "P_counting_sort"), (13, "P_run_stores"), (1, "P_drain"),
             (2, "barrier"), (14, "C_totals_scan"), (15, "C_ranks_bins_lists"), (16, "G_prefix_pairs"), (17, "G_block_prefix"),
             (3, "G_members"), (8, "D_sync"), (9, "D_select"), (5, "D_thresholds"), (6, "apply"), (7, "end")]
    out = {}
    for slot in H._RF_WS.values():
        acc = slot[1][7168:7200].cpu().numpy().astype("int64")
        if acc[31] <= 0:
            continue
        prev = 0.0
        for k, nm in names:
            cur = acc[k] / acc[31]
            out[nm] = round((cur - prev) / 100.0, 2)
            prev = cur
        out["launches"] = int(acc[31])
        break
    return out


def kernel_source_hash():
    """sha256 (first 12 hex digits) over the HIP sources and headers of the library: ties a profile to the kernels it measured
    (the GPU box has no .git)"""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    hs = hashlib.sha256()
    files = []
    for d in (os.path.join(root, "csrc"), os.path.join(os.path.dirname(root), "include")):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h"))]
    for f in files:
        hs.update(os.path.basename(f).encode())
        hs.update(open(f, "rb").read())
    return hs.hexdigest()[:12]


def hbm_algorithmic_bytes(B, C, H, W, h, w, D, stats):
    """entropy+reliability 4C+24 B/pixel + low-res outputs; contrastive with
    the Q0 skip (only images {0,B} referenced) from the measured counts."""
    px = B * H * W
    rel = px * (4 * C + 24) + (2 * 4 + 8) * 2 * B * h * w
    Q, K = stats.get("Q", 256), stats.get("K", 50)
    con = (2 * D * h * w * 4 + 2 * B * C * h * w * 4 + 2 * B * h * w * 9 + stats.get("n_keys", 0) * D * 4
           + stats.get("njobs", 0) * Q * (2 + K) * D * 4 + stats.get("njobs", 0) * Q * D * 4)
    return rel, con


def measure(trainer, batch, args, ms_per_step):
    from .utils import loss_helper as LH

    from . import hipops as H

    il, ll, iu = batch
    # per-kernel HIP-event timing needs serial execution: run the profiled step without the side stream
    saved = getattr(trainer, "_side", None)
    trainer._side = torch.cuda.current_stream()
    H.REPLAY = {}
    try:
        agg = profile_step(lambda: trainer.train_step(il, ll, iu, epoch=1))
        replay = replay_hbm_group()
    finally:
        trainer._side = saved
        H.REPLAY = None
    shapes = agg.pop("_shapes")
    dense = agg.pop("_dense")
    out = {}
    if os.environ.get("U2PL_BENCH_SHAPES"):
        top = sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])[:40]
        out["conv_shapes"] = [dict(op=k[0], N=k[1], Hin=k[2], Cin=k[4], Hout=k[5], Cout=k[7], k=k[8], s=k[10], d=k[12],
                                   calls=v["calls"], ms=round(v["ms"], 2),
                                   tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)) for k, v in top]
    ig = [(n_, agg.get(n_)) for n_ in MFMA_GROUPS["igemm"]]
    pipe = sum(x["flops"] * piece_products(n_) for n_, x in ig if x)
    ig = [x for _, x in ig if x]
    if ig:
        fl, t_ev, n = sum(x["flops"] for x in ig), sum(x["ms"] for x in ig), sum(x["calls"] for x in ig)
        nker = sum(x["kernels"] for x in ig)
        t = dense.get("igemm", t_ev)      # sustained-load time (dense replay); the gapped per-call events read shorter
        ach = fl / (t * 1e-3) / 1e12
        out["roofline"] = {"kernel": "k_igemm_ws / k_conv_igemm (direct conv fwd [+BN-stat / eval-BN epilogue] + dgrad, and the batched Winograd component GEMMs; executed fp32 FLOPs)", "bound": "mfma",
                           "achieved": round(ach, 2), **_mfma_fields(ach, pipe, fl), "unit": "TFLOP/s", "traffic": None,
                           "abi_calls_per_step": n, "kernel_launches_per_step": nker, "avg_launch_ms": round(t / max(nker, 1), 4),
                           "avg_abi_call_ms": round(t / n, 4),
                           "executed_tflop_per_step": round(fl / 1e12, 3), "ms_per_step": round(t, 2),
                           "ms_per_step_gapped_events": round(t_ev, 2),
                           "algorithmic_bytes_per_step": round(sum(x.get("bytes", 0.0) for x in ig)),
                           "algorithmic_bytes_note": "sum over the group's launches of A read once + B read once + output written "
                                                     "once; compare with traffic (PMC) x launches",
                           "method": "all launches of the group re-issued back to back behind a spinning kernel, one HIP-event "
                                     "pair (sustained clocks, like a rocprofv3 run); ms_per_step_gapped_events = sum of "
                                     "per-call event pairs with idle gaps (boost clocks)"}
        # SURVEY 8(d) ALGORITHMIC figure (direct-convolution FLOPs of the whole step, Winograd savings not deducted)
        # over the whole step time: the "effective" rate the headline images/s corresponds to
        if args.crop == 769 and args.arch == "resnet101":
            algo = ALGO_TFLOP_PER_IMAGE_769 * 2 * args.batch
            out["roofline"].update(algorithmic_tflop_per_step=round(algo, 2),
                                   algorithmic_achieved=round(algo / (ms_per_step * 1e-3), 2),
                                   algorithmic_frac=round(algo / (ms_per_step * 1e-3) / PEAK_F32_MFMA_TFLOPS, 4),
                                   note="frac/achieved: executed FLOPs of this kernel's launches over their own HIP-event "
                                        "time (serialised extra step); algorithmic_*: 26.4 TFLOP (SURVEY 8d, every layer "
                                        "counted as a direct convolution) over the WHOLE timed step, against the 157.3 TF of the fp32 MFMA "
                                        "instruction (can exceed 1: Winograd multiplies less, the split form runs on the bf16 pipe)")
    bfs = [agg.get(n) for n in ("u2pl_conv2d_fwd_bf16op_f32", "u2pl_conv2d_fwd_bnstats_bf16op_f32", "u2pl_conv2d_dgrad_bf16op_f32",
                                "u2pl_conv2d_wgrad_bf16op_f32")]
    bfs = [x for x in bfs if x]
    if bfs:   # config 5: the student's products on the bf16 matrix cores (operands rounded in LDS, fp32 tensors in HBM)
        fl, t, n = sum(x["flops"] for x in bfs), sum(x["ms"] for x in bfs), sum(x["calls"] for x in bfs)
        t = dense.get("bf16", t)
        ach = fl / (t * 1e-3) / 1e12
        out["roofline_bf16"] = {"kernel": "k_conv_igemm<BF> / k_conv_wgrad_bf16 (student fwd + dgrad + wgrad, bf16 operands, fp32 accumulate)",
                                "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None, "abi_calls_per_step": n,
                                "kernel_launches_per_step": sum(x["kernels"] for x in bfs),
                                "ms_per_step": round(t, 2), "executed_tflop_per_step": round(fl / 1e12, 3),
                                "note": "fp32 activations / weights are read from HBM and rounded on the way into LDS: these "
                                        "launches are HBM / LDS bound long before the 2.5 PFLOP/s matrix-core peak"}
    wgs = [(n_, agg.get(n_)) for n_ in MFMA_GROUPS["wgrad"]]
    wpipe = sum(x["flops"] * piece_products(n_) for n_, x in wgs if x)
    wgs = [x for _, x in wgs if x]
    wg = dict(flops=sum(x["flops"] for x in wgs), ms=sum(x["ms"] for x in wgs), calls=sum(x["calls"] for x in wgs),
              kernels=sum(x["kernels"] for x in wgs)) if wgs else None
    if wg:
        wg["ms"] = dense.get("wgrad", wg["ms"])
        ach = wg["flops"] / (wg["ms"] * 1e-3) / 1e12
        out["roofline_wgrad"] = {"kernel": "k_conv_wgrad (direct, + ordered slab reduce; and the batched Winograd component products; executed FLOPs)", "bound": "mfma",
                                 "achieved": round(ach, 2), **_mfma_fields(ach, wpipe, wg["flops"]), "unit": "TFLOP/s", "traffic": None,
                                 "abi_calls_per_step": wg["calls"], "kernel_launches_per_step": wg["kernels"],
                                 "ms_per_step": round(wg["ms"], 2)}
    B, H, W = ll.shape
    C = trainer.num_classes
    h, w = (H - 1) // 4 + 1, (W - 1) // 4 + 1
    rel_names = ["u2pl_entropy_f32", "u2pl_entropy_up_f32", "u2pl_select_f32", "u2pl_apply_drop_i64",
                 "u2pl_reliability_masks", "u2pl_reliability_apply", "u2pl_reliability_fused"]
    con_names = ["u2pl_contra_classify", "u2pl_compact_lists", "u2pl_class_prototypes", "u2pl_contra_phase1", "u2pl_bank_append_f32",
                 "u2pl_bank_append_multi_f32", "u2pl_infonce_f32", "u2pl_infonce_fused_f32", "u2pl_infonce_reduce_f32",
                 "u2pl_scatter_rows_ordered_f32", "u2pl_zero_rows_f32"]
    rel_b, con_b = hbm_algorithmic_bytes(B, C, H, W, h, w, 256, LH.LAST_STATS)
    t_rel = sum(agg[n]["ms"] for n in rel_names if n in agg)
    t_con = sum(agg[n]["ms"] for n in con_names if n in agg)
    per_call = {"reliability_us": round(t_rel * 1e3, 1), "contrastive_us": round(t_con * 1e3, 1)}
    if "reliability_us" in replay and "phase1_us" in replay:   # back-to-back replay of the same launches (see replay_hbm_group)
        t_rel = replay["reliability_us"] * 1e-3
        t_con = (replay["phase1_us"] + replay.get("bank_append_us", 0.0) + replay.get("infonce_fwd_bwd_us", 0.0)) * 1e-3
    if t_rel > 0 and t_con > 0:
        ach = (rel_b + con_b) / ((t_rel + t_con) * 1e-3) / 1e9
        out["roofline_hbm"] = {"kernel": "k_reliability_fused (entropy + exact percentiles + target + masks: one persistent launch, one device-wide barrier) + contrastive (phase 1 in three launches: classify, prototype stream, compaction write || prototype finish; bank append; InfoNCE with the loss reduction and the stale-row clearing in the same launch; ordered row-sparse bwd)",
                               "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                               "algorithmic_MB": round((rel_b + con_b) / 1e6, 1), "reliability_us": round(t_rel * 1e3, 1),
                               "contrastive_us": round(t_con * 1e3, 1), "stats": dict(LH.LAST_STATS),
                               "stages_us": {k: round(v, 1) for k, v in replay.items()},
                               "per_call_event_us": per_call, "split_phases_us_block0": split_phase_table(),
                               "method": "each stage re-issued 10x on the last step's tensors behind a spinning kernel, one "
                                         "HIP-event pair per stage (per-call event pairs, kept in per_call_event_us, add "
                                         "~20 us of marker packets to 5-30 us kernels)"}
    # HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections) cannot be collected from
    # inside this process: the value below is a STATIC record of the PMC passes committed under profiles/ (with the
    # commit they were taken at), not a measurement of this run
    import glob
    pdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    tjs = sorted(glob.glob(os.path.join(pdir, "r[0-9][0-9]_traffic.json")))
    tj = tjs[-1] if tjs else ""
    if os.path.exists(tj) and "roofline" in out:
        import json
        tr = json.load(open(tj))
        sha = kernel_source_hash()
        if tr.get("kernel_sources_sha") != sha:
            # a PMC record of OTHER kernel sources is not this build's traffic: refuse it rather than quote a stale number
            out["roofline"]["traffic_source"] = ("none: profiles/%s was taken with kernel sources %s (commit %s), this build is %s; "
                                                 "re-run tools/run_r6_profiles.sh" % (os.path.basename(tj), tr.get("kernel_sources_sha", "?"),
                                                                                      tr.get("commit", "?"), sha))
            if "roofline_hbm" in out:
                out["roofline_hbm"]["traffic_source"] = out["roofline"]["traffic_source"]
        else:
            out["roofline"]["traffic"] = tr.get("igemm_bytes_per_launch", tr.get("k_conv_igemm_bytes_per_launch"))
            # per step = the record's average bytes per launch x THIS run's launches of the group per step.  (Rounds 2-4 divided
            # the profiled run's total by its number of timed + diagnostic steps, which counted the calibration passes' ~800
            # launches as step traffic: 187.6 GB = "1.61x algorithmic" in round 4 was 150 GB = 1.29x by this accounting; the
            # record's own figure is kept beside it)
            nl = out["roofline"].get("kernel_launches_per_step")
            if out["roofline"]["traffic"] and nl:
                out["roofline"]["traffic_per_step"] = int(out["roofline"]["traffic"] * nl)
                ab = out["roofline"].get("algorithmic_bytes_per_step")
                if ab:
                    out["roofline"]["traffic_over_algorithmic"] = round(out["roofline"]["traffic_per_step"] / ab, 3)
            out["roofline"]["traffic_per_step_by_set_count"] = tr.get("igemm_bytes_per_step")
            out["roofline"]["traffic_source"] = ("static: profiles/%s (rocprofv3 PMC passes at commit %s, kernel sources %s = this "
                                                 "build), not measured in this run" % (os.path.basename(tj), tr.get("commit", "?"), sha))
            if "roofline_hbm" in out and tr.get("hbm_group_bytes_per_step") is not None:
                out["roofline_hbm"]["traffic"] = tr["hbm_group_bytes_per_step"]
                out["roofline_hbm"]["traffic_source"] = out["roofline"]["traffic_source"]
    top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:12]
    out["abi_calls_per_step"] = sum(v["calls"] for v in agg.values())
    out["kernel_launches_per_step"] = sum(v["kernels"] for v in agg.values())
    out["kernel_ms_per_step"] = {k: round(v["ms"], 2) for k, v in top}
    out["kernel_ms_total_profiled"] = round(sum(v["ms"] for v in agg.values()), 2)
    return out
