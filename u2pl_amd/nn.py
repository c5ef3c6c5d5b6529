"""Layer modules of the MI355X-native DeepLabv3+ / dilated ResNet stack.

Each module keeps torch's parameter/buffer NAMES (checkpoint wire format of the
reference: ``encoder.layer3.5.conv2.weight`` ...) but its math is a HIP kernel
from libu2pl_hip.so reached through a ``torch.autograd.Function``.  Activations
are logical NCHW tensors stored channels_last (NHWC "rows"); conv weights are
logical OIHW stored channels_last ([Cout][R][S][Cin]) so 1x1 convs are plain
GEMMs and state_dicts stay interchangeable with the reference.

``nn.ReLU`` / ``nn.Dropout2d`` / ``nn.AdaptiveAvgPool2d`` instances inside
``nn.Sequential`` containers are only *markers* (they keep the reference's
Sequential indices); ``run_seq`` fuses them into the BatchNorm apply kernel.
"""
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from ._lib import HipError, call, query

from .comm import (COMM_DEBUG, COMM_STATS, _all_reduce, _digest_of, _world, check_comm_sequence, comm_sequence_digest,  # noqa: F401
                   dist_active, note_collective)
from .layout import _CL, _ws, as_rows, new_act  # noqa: F401
from .operands import (AMAX_STATS, CONV_H, CONV_WS, PRESPLIT, WEIGHT_EPOCH, _AMAX_POOL, _DerivedCache, _derived, _split_of,  # noqa: F401
                       _weight_stamp, _ws_ok, amax_of, amax_pool_reset, amax_slot, bump_weight_epoch, invalidate_weights, presplit,
                       set_amax, ws_dgrad, ws_forward, ws_wino)


# Weight-gradient kernels are leaves of the backward graph: they run on a side HIP stream so the
# memory-bound BatchNorm-backward passes and the next layer's dgrad (main stream) overlap the MFMA-bound
# wgrad.  wgrad_stream_sync() must be called before the optimizer reads the gradient arena.
_WGRAD = {"stream": None, "enabled": os.environ.get("U2PL_NO_WGRAD_STREAM") is None}


def _wgrad_stream():
    if not _WGRAD["enabled"] or _lib.PROFILE is not None:
        return None
    if _WGRAD["stream"] is None:
        _WGRAD["stream"] = torch.cuda.Stream()
    return _WGRAD["stream"]


def wgrad_stream_sync():
    _WGRAD["queued"] = False
    if _WGRAD["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_WGRAD["stream"])
    _lib.SIDE_WORK.discard("wgrad")


def _queue_wgrad_join():
    """Join the side stream when the running backward pass finishes (once per pass)."""
    if not _WGRAD.get("queued"):
        _WGRAD["queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_stream_sync)


def _grad_sink(p):
    """Arena gradient view of a parameter (set by ParamArena) or None."""
    return getattr(p, "_u2pl_grad", None)


def _mark_ready(sink):
    """the gradient of this parameter is complete for this step: lets the arena launch the all-reduce of a finished
    bucket while the rest of the backward pass is still running (no-op without a process group)"""
    if sink is not None:
        owner = getattr(sink, "_u2pl_ready", None)
        if owner is not None:
            owner[0].mark_ready(owner[1])


# ------------------------------------------------------------------ convolution
# Stride-1 "same" 3x3 convolutions can run in Winograd form (csrc/wino.hip): F(4x4,3x3) does 1/4 of the direct
# multiplies (F(2x2,3x3): 4/9) in fp32 on the same matrix cores, at the cost of two HBM-bound transform passes.
# U2PL_CONV_WINO = 0 (direct implicit GEMM only) | 2 | 4 (tile size; default 4).  Dilated convolutions are
# decomposed into d*d sub-images; a layer only takes the Winograd path when the multiply reduction that is
# left after tile padding is worth the transforms.
# U2PL_CONV_BF16 = 1 (BASELINE configs[4], "config 5"): the STUDENT's convolutions (every call that records a gradient)
# round their operands to bf16 while staging them into LDS and run on the bf16 matrix cores with fp32 accumulation
# (forward, data and weight gradient; direct kernel -- no Winograd on 8-bit mantissas); tensors, master weights, the EMA
# teacher (no_grad calls) and the 3-channel stem stay fp32.  Never the default: the headline is the reference's fp32.
CONV_ALGO = {"wino": int(os.environ.get("U2PL_CONV_WINO", "4")), "min_gain": float(os.environ.get("U2PL_WINO_MIN_GAIN", "1.7")),
             "wgrad": int(os.environ.get("U2PL_WINO_WGRAD", "1")), "bf16": int(os.environ.get("U2PL_CONV_BF16", "0"))}


def wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W):
    """-> Winograd output-tile size (2 / 4) for this layer, or 0 for the direct kernel."""
    mt = CONV_ALGO["wino"]
    if mt not in (2, 4) or R != 3 or S != 3 or stride != 1 or pad != dil or Cin % 32 or Cout % 4 or Cout < 32:
        return 0
    th, tw = -(-(-(-H // dil)) // mt), -(-(-(-W // dil)) // mt)
    eff = (H * W) / float(dil * dil * th * tw * mt * mt)          # useful / computed outputs
    gain = 9.0 * mt * mt / ((mt + 2) ** 2) * eff                    # direct multiplies / Winograd multiplies
    return mt if gain >= CONV_ALGO["min_gain"] else 0


def _wino_conv(x, ldx, weight, bias, y, N, H, W, Cin, Cout, dil, mt, transposed, pivot=None, epi=None, y_amax=None):
    """y <- conv3x3(x) in Winograd form; transposed: data-gradient (x = dY with Cout channels, y = dX with Cin).
    Returns the fused BatchNorm partial sums (or None).  epi = (mean, invstd, gamma, beta, res, ldr, relu): eval-mode
    BatchNorm applied by the output transform."""
    dev = x.device
    a2 = (mt + 2) ** 2
    Ci, Co = (Cout, Cin) if transposed else (Cin, Cout)
    tiles = query("u2pl_wino_tiles", N, H, W, dil, mt)
    V = torch.empty(a2 * tiles * Ci, dtype=torch.float32, device=dev)
    ws = _ws_ok(Co, Ci) and weight.is_leaf
    Mb = torch.empty(a2 * tiles * Co, dtype=torch.float32, device=dev)
    if ws and CONV_H["on"]:      # split-fp16: the transform leaves max |V| as it writes
        v_amax = amax_slot(dev)
        call("u2pl_wino_input_amax_f32", x, ldx, N, H, W, Ci, dil, mt, V, v_amax)
        set_amax(V, v_amax)
        call("u2pl_gemm_batched_wsh_f32", V, Ci, tiles * Ci, v_amax, ws_wino(weight, transposed, mt, True), Mb, Co, tiles * Co,
             tiles, Ci, Co, a2)
    elif ws:
        call("u2pl_wino_input_f32", x, ldx, N, H, W, Ci, dil, mt, V)
        call("u2pl_gemm_batched_ws_f32", V, Ci, tiles * Ci, ws_wino(weight, transposed, mt, False), Mb, Co, tiles * Co, tiles, Ci,
             Co, a2)
    else:
        call("u2pl_wino_input_f32", x, ldx, N, H, W, Ci, dil, mt, V)
        U = torch.empty(a2 * Cout * Cin, dtype=torch.float32, device=dev)
        call("u2pl_wino_weight_f32", weight, Cout, Cin, int(transposed), mt, U)
        call("u2pl_gemm_batched_f32", V, Ci, tiles * Ci, U, Co * Ci, Mb, Co, tiles * Co, tiles, Ci, Co, a2)
    part = None
    if pivot is not None:
        nblk = query("u2pl_wino_stat_blocks", tiles, Co)
        part = torch.empty((nblk, 2, Co), dtype=torch.float32, device=dev)
    if epi is not None and y_amax is not None:      # split-fp16: the fused output is the next convolution's operand
        call("u2pl_wino_output_bnact_amax_f32", Mb, N, H, W, Co, dil, mt, bias, y, Co, *epi, y_amax)
    elif epi is not None:
        call("u2pl_wino_output_bnact_f32", Mb, N, H, W, Co, dil, mt, bias, y, Co, *epi)
    else:
        call("u2pl_wino_output_f32", Mb, N, H, W, Co, dil, mt, bias, y, Co, part, pivot)
    return part, V


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil, wsink, bsink, pivot=None, recording=True, link=None):
        """pivot (a BatchNorm running_mean) requests the fused train-mode BN statistics of the output:
        returns (y, partials) with partials = float32 [nblk][2][C] (S1, S2 pivot-shifted, per row block of the epilogue);
        finished_sums() / the BatchNorm's fused finish turn them into the double [2C] sums."""
        x_in = x
        x, ldx = as_rows(x)
        N, Cin, H, W = x.shape
        Cout, _, R, S = weight.shape
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
        if not weight.is_contiguous(memory_format=_CL) and not (R == 1 and S == 1 and weight.is_contiguous()):
            raise HipError("conv weight must be stored channels_last ([Cout][R][S][Cin])")
        # the second output (the fused BatchNorm sums) is not differentiable: without this autograd materialises a zero fp64
        # gradient for it in front of every backward call (116 fill launches per step, profiles/r05_bench_serial_kernel_stats.csv)
        ctx.set_materialize_grads(False)
        ctx.link = link
        ctx.x_amax = ctx.col_amax = None        # split-fp16: maxima of the saved operands (device scalars), where the forward had them
        y = new_act(N, Cout, Ho, Wo, x.device)
        col = None
        # `recording`: grad mode at the call site (inside forward() it is always off); the teacher's calls run under no_grad
        use_bf = (bool(CONV_ALGO.get("bf16", 0)) and recording and any(ctx.needs_input_grad[:3]) and Cin % 32 == 0
                  and H * W > 1)
        ctx.bf = use_bf
        sfx = "_bf16op_f32" if use_bf else "_f32"
        if H * W == 1 and R == 1 and S == 1 and stride == 1 and pad == 0 and N <= 16 and Cin % 4 == 0:
            # image-pooling branch of the ASPP: float64-accumulated dense layer (see csrc/nn.hip:k_dense_small)
            call("u2pl_dense_small_f32", x, ldx, weight, bias, y, Cout, N, Cin, Cout)
            pivot = None if pivot is None else False   # statistics by the stand-alone pass
        elif Cin % 32:
            Kp = ((R * S * Cin + 31) // 32) * 32
            col = torch.empty((N * Ho * Wo, Kp), dtype=torch.float32, device=x.device)
            call("u2pl_im2col_f32", x, ldx, col, Kp, N, H, W, Cin, Ho, Wo, R, S, stride, pad, dil)
            wp = torch.zeros((Cout, Kp), dtype=torch.float32, device=x.device)
            wp[:, : R * S * Cin] = weight.permute(0, 2, 3, 1).reshape(Cout, R * S * Cin)
            call("u2pl_conv2d_fwd_f32", col, Kp, wp, bias, y, Cout, N, Ho, Wo, Kp, Ho, Wo, Cout, 1, 1, 1, 0, 1)
        elif not use_bf and wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W):
            mt = wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W)
            part, V = _wino_conv(x, ldx, weight, bias, y, N, H, W, Cin, Cout, dil, mt, False, pivot)
            if ctx.needs_input_grad[1]:
                col = V     # the transformed input is the weight gradient's operand: keep it instead of redoing it
                ent = getattr(V, "_u2pl_amax", None)
                ctx.col_amax = ent[0] if ent is not None else None
            if pivot is not None:
                sums = part      # raw [nblk][2][Cout] partials: the BatchNorm finishes them (fused with its finalisation, round 5)
        elif pivot is not None and pivot is not False:
            ws = not use_bf and _ws_ok(Cout, R * S * Cin)
            nblk = query("u2pl_igemm_ws_stat_blocks", N, Ho, Wo) if ws else query("u2pl_conv2d_fwd_stat_blocks", N, Ho, Wo, Cout)
            part = torch.empty((nblk, 2, Cout), dtype=torch.float32, device=x.device)
            if ws and CONV_H["on"]:
                ctx.x_amax = amax_of(x_in, x, ldx)
                call("u2pl_conv2d_fwd_bnstats_wsh_f32", x, ldx, ctx.x_amax, ws_forward(weight, True), bias, y, Cout, N, H, W, Cin,
                     Ho, Wo, Cout, R, S, stride, pad, dil, pivot, part)
            elif ws:
                call("u2pl_conv2d_fwd_bnstats_ws_f32", x, ldx, ws_forward(weight, False), bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout,
                     R, S, stride, pad, dil, pivot, part)
            else:
                call("u2pl_conv2d_fwd_bnstats" + sfx, x, ldx, weight, bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout, R, S, stride,
                     pad, dil, pivot, part)
            sums = part          # raw partials, see above
        elif not use_bf and _ws_ok(Cout, R * S * Cin):
            if CONV_H["on"]:
                ctx.x_amax = amax_of(x_in, x, ldx)
                call("u2pl_conv2d_fwd_wsh_f32", x, ldx, ctx.x_amax, ws_forward(weight, True), bias, y, Cout, N, H, W, Cin, Ho, Wo,
                     Cout, R, S, stride, pad, dil)
            else:
                call("u2pl_conv2d_fwd_ws_f32", x, ldx, ws_forward(weight, False), bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout, R, S,
                     stride, pad, dil)
        else:
            call("u2pl_conv2d_fwd" + sfx, x, ldx, weight, bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil)
        ctx.save_for_backward(x, weight, col)
        ctx.geom = (N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ldx)
        ctx.has_bias = bias is not None
        ctx.wsink, ctx.bsink = wsink, bsink
        if pivot is not None:
            if Cin % 32 or pivot is False:   # stem (im2col path) / pooled dense layer: statistics by the stand-alone pass
                return y, None
            ctx.mark_non_differentiable(sums)
            return y, sums
        return y

    @staticmethod
    def backward(ctx, gy, gsums=None):
        if gy is None:      # (only the non-differentiable sums were used downstream)
            return (None,) * 11
        x, weight, col = ctx.saved_tensors
        N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ldx = ctx.geom
        gy_in = gy          # (the tensor autograd handed over: it carries the producer's fused maximum, if any)
        gy, ldg = as_rows(gy)
        dev = gy.device
        M = N * Ho * Wo
        Cp = Cout
        if Cout % 32:  # narrow heads (num_classes outputs): zero-pad the gradient columns once
            Cp = (Cout + 31) // 32 * 32
            gp = torch.zeros((M, Cp), dtype=torch.float32, device=dev)
            call("u2pl_copy_cols_f32", gy, ldg, gp, Cp, M, Cout)
            gy, ldg = gp, Cp
            wpad = torch.zeros((Cp, Cin, R, S), dtype=torch.float32, device=dev).contiguous(memory_format=_CL)
            wpad[:Cout] = weight
            weight_k = wpad
        else:
            weight_k = weight
        dx = dw = db = None
        # what the OTHER consumers of this conv's input have contributed to its gradient so far (GradJoin): folded into this
        # launch where a fused form exists, added in place otherwise; handed on or returned by join.settle below
        join = ctx.link
        prev = join.acc if join is not None else None
        if join is not None and not ctx.needs_input_grad[0]:
            raise HipError("gradient join on a convolution whose input needs no gradient")
        if ctx.needs_input_grad[0]:
            dx = new_act(N, Cin, H, W, dev)
            included = prev is None
            epi = None
            if prev is not None and Cin % 4 == 0:
                one, zero = _identity_bn(Cin, dev)
                rr, ldr = as_rows(prev)
                epi = (zero, one, one, zero, rr, ldr, 0)     # identity BatchNorm + `res` = the running sum: (v - 0) * 1 * 1 + 0 + res
            mt = wino_tile(Cp, Cin, R, S, stride, pad, dil, H, W) if (Cp == Cout and not ctx.bf) else 0
            if mt:   # data gradient = the same convolution with rotated taps and swapped channel roles
                _wino_conv(gy, ldg, weight_k, None, dx, N, H, W, Cin, Cout, dil, mt, True, None, epi)
                included = included or epi is not None     # (the output transform's residual epilogue)
            elif (epi is not None and R == 1 and S == 1 and stride == 1 and pad == 0 and Cp == Cout and not ctx.bf
                  and _ws_ok(Cin, Cout)):
                # pointwise: the data gradient IS a pointwise forward convolution with the transposed weight planes -- run it
                # through the forward kernel's eval-BatchNorm epilogue with identity parameters and the running sum as its
                # `res`: dx = (gy . W) + prev in ONE pass instead of the GEMM + autograd's elementwise add over the 4x-wide
                # block input (46 adds, 3.2 ms per step before round 5)
                if CONV_H["on"]:
                    call("u2pl_conv2d_fwd_bnact_wsh_f32", gy, ldg, amax_of(gy_in, gy, ldg), ws_dgrad(weight, True), None, dx, Cin, N,
                         H, W, Cout, H, W, Cin, 1, 1, 1, 0, 1, *epi, None)
                else:
                    call("u2pl_conv2d_fwd_bnact_ws_f32", gy, ldg, ws_dgrad(weight, False), None, dx, Cin, N, H, W, Cout, H, W, Cin, 1, 1,
                         1, 0, 1, *epi)
                included = True
            elif Cp == Cout and not ctx.bf and _ws_ok(Cin, R * S * Cout):
                if CONV_H["on"]:
                    call("u2pl_conv2d_dgrad_wsh_f32", gy, ldg, amax_of(gy_in, gy, ldg), ws_dgrad(weight, True), dx, Cin, N, H, W, Cin,
                         Ho, Wo, Cout, R, S, stride, pad, dil)
                else:
                    call("u2pl_conv2d_dgrad_ws_f32", gy, ldg, ws_dgrad(weight, False), dx, Cin, N, H, W, Cin, Ho, Wo, Cout, R, S,
                         stride, pad, dil)
            else:
                wT = torch.empty(Cin * R * S * Cp, dtype=torch.float32, device=dev)
                call("u2pl_weight_transpose_f32", weight_k, wT, Cp, R * S, Cin)
                call("u2pl_conv2d_dgrad_bf16op_f32" if ctx.bf else "u2pl_conv2d_dgrad_f32", gy, ldg, wT, dx, Cin, N, H, W, Cin,
                     Ho, Wo, Cp, R, S, stride, pad, dil)
            if join is not None:
                dx = join.settle(dx, included)      # None unless this was the last consumer to report
        # split-fp16 weight gradients (k_wgrad_tr only): the operands' maxima are taken on THIS stream, before the side stream forks
        wg_h = (CONV_H["on"] and ctx.needs_input_grad[1] and Cp == Cout and not ctx.bf and Cin % 32 == 0
                and query("u2pl_wgrad_h_eligible", Cin, Cout) == 1)
        wg_wino = wg_h and bool(CONV_ALGO.get("wgrad", 1) and wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W))
        gy_amax = x_amax = None
        if wg_h and not wg_wino:
            gy_amax = amax_of(gy_in, gy, ldg)
            x_amax = ctx.x_amax if ctx.x_amax is not None else amax_of(x, x, ldx)
        side = _wgrad_stream() if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            _lib.SIDE_WORK.add("wgrad")
            for t_ in (gy, x, col, gy_amax, x_amax, ctx.col_amax):
                if t_ is not None:
                    t_.record_stream(side)
            _queue_wgrad_join()
            stream_ctx = torch.cuda.stream(side)
            stream_ctx.__enter__()
        if ctx.needs_input_grad[1]:
            sink = ctx.wsink
            if Cin % 32:  # 3-channel stem (col = im2col patches): gradient of the padded [Cout][Kp] patch-matrix weights
                Kp = col.shape[1]
                tmp = torch.empty((Cp, Kp), dtype=torch.float32, device=dev)
                wsb = _ws(query("u2pl_conv2d_wgrad_workspace_bytes", N, Ho, Wo, Kp, Cp, 1, 1), dev)
                call("u2pl_conv2d_wgrad_f32", gy, ldg, col, Kp, tmp, wsb, 0, N, Ho, Wo, Kp, Ho, Wo, Cp, 1, 1, 1, 0, 1)
                g = tmp[:Cout, : R * S * Cin].reshape(Cout, R, S, Cin).permute(0, 3, 1, 2)
                if sink is not None:
                    sink.add_(g)
                else:
                    dw = g.contiguous(memory_format=_CL)
            elif Cp == Cout and not ctx.bf and CONV_ALGO.get("wgrad", 1) and wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W):
                # Winograd weight gradient: dU = sum_tiles (A dY A^T) (x) (B^T x B), dW = G^T dU G
                mt = wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W)
                a2 = (mt + 2) ** 2
                tiles = query("u2pl_wino_tiles", N, H, W, dil, mt)
                V, v_amax = col, ctx.col_amax      # saved by the forward
                if V is None:
                    V = torch.empty(a2 * tiles * Cin, dtype=torch.float32, device=dev)
                    if wg_h:
                        v_amax = amax_slot(dev)
                        call("u2pl_wino_input_amax_f32", x, ldx, N, H, W, Cin, dil, mt, V, v_amax)
                    else:
                        call("u2pl_wino_input_f32", x, ldx, N, H, W, Cin, dil, mt, V)
                Mg = torch.empty(a2 * tiles * Cout, dtype=torch.float32, device=dev)
                ns = query("u2pl_wgrad_batched_splits", tiles, Cin, Cout, a2)
                part = torch.empty(ns * Cout * a2 * Cin, dtype=torch.float32, device=dev)
                if wg_h:
                    if v_amax is None:
                        v_amax = amax_slot(dev)
                        call("u2pl_absmax_f32", V, Cin, a2 * tiles, Cin, v_amax, 0)
                    mg_amax = amax_slot(dev)
                    call("u2pl_wino_gy_amax_f32", gy, ldg, N, H, W, Cout, dil, mt, Mg, mg_amax)
                    call("u2pl_wgrad_batched_h_f32", Mg, Cout, tiles * Cout, mg_amax, V, Cin, tiles * Cin, v_amax, part, tiles, Cin,
                         Cout, a2)
                else:
                    call("u2pl_wino_gy_f32", gy, ldg, N, H, W, Cout, dil, mt, Mg)
                    call("u2pl_wgrad_batched_f32", Mg, Cout, tiles * Cout, V, Cin, tiles * Cin, part, tiles, Cin, Cout, a2)
                tgt = sink if sink is not None else torch.empty_like(weight)
                call("u2pl_wino_wgrad_finish_f32", part, ns, Cout, Cin, mt, int(sink is not None), tgt)
                if sink is None:
                    dw = tgt
            else:
                wsb = _ws(query("u2pl_conv2d_wgrad_workspace_bytes", N, Ho, Wo, Cin, Cp, R, S), dev)
                direct = sink is not None and Cp == Cout
                tgt = sink if direct else torch.empty_like(weight_k)
                if gy_amax is not None:
                    call("u2pl_conv2d_wgrad_h_f32", gy, ldg, gy_amax, x, ldx, x_amax, tgt, wsb, int(direct), N, H, W, Cin, Ho, Wo,
                         Cp, R, S, stride, pad, dil)
                else:
                    call("u2pl_conv2d_wgrad_bf16op_f32" if ctx.bf else "u2pl_conv2d_wgrad_f32", gy, ldg, x, ldx, tgt, wsb,
                         int(direct), N, H, W, Cin, Ho, Wo, Cp, R, S, stride, pad, dil)
                if not direct:
                    if sink is not None:
                        sink.add_(tgt[:Cout])
                    else:
                        dw = tgt[:Cout] if Cp != Cout else tgt
        if ctx.needs_input_grad[1]:
            _mark_ready(ctx.wsink)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            sums = torch.empty(2 * Cp, dtype=torch.float64, device=dev)
            wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, Cp), dev)
            call("u2pl_colsum_f32", gy, ldg, M, 1, Cp, wsb, sums)
            if ctx.bsink is not None:
                call("u2pl_sums_to_f32", sums, Cout, 1.0, 1, ctx.bsink)
                _mark_ready(ctx.bsink)
            else:
                db = torch.empty(Cout, dtype=torch.float32, device=dev)
                call("u2pl_sums_to_f32", sums, Cout, 1.0, 0, db)
        if side is not None:
            stream_ctx.__exit__(None, None, None)
            for t_ in (dw, db):
                if t_ is not None:   # returned to autograd on the main stream
                    torch.cuda.current_stream().wait_stream(side)
                    break
        return dx, dw, db, None, None, None, None, None, None, None, None


_IDENT_BN = {}


def prewarm_globals(modules, dev):
    """allocate the process-lifetime caches a pass may touch (identity BatchNorm parameters of the fused gradient joins) OUTSIDE a
    HIP-graph capture: born inside one they would live in the graph's private memory pool"""
    for mod in modules:
        for m in mod.modules():
            if isinstance(m, Conv2d):
                _identity_bn(m.in_channels, dev)


def _identity_bn(C, dev):
    k = (C, str(dev))
    v = _IDENT_BN.get(k)
    if v is None:
        v = _IDENT_BN[k] = (torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev))
    return v


# A tensor with several consumers gets its gradient as the SUM of their contributions, which autograd forms with elementwise adds
# (12 B per element; a bottleneck's input -- conv1 + the residual add inside bn3 -- is the 4x-wide tensor of the block).  GradJoin
# moves the sum into the consumers' own backward launches: the module that owns the fan-out creates one join for its n wired
# consumers and hands it to each; a consumer's backward folds the running sum (`acc`) into its own launch where a fused form exists
# (pointwise / Winograd data gradients: the forward kernels' residual epilogue with identity BatchNorm parameters), adds it in place
# otherwise, and either parks the new sum in the join and reports None to autograd, or -- the last one -- returns the total.
# Order-agnostic; every wired consumer's backward MUST run in the pass (they are wired only where all outputs reach the loss).
# U2PL_NO_RES_GRAD_FUSION=1: off.
FUSE_RES_GRAD = os.environ.get("U2PL_NO_RES_GRAD_FUSION") is None


class GradJoin:
    __slots__ = ("left", "acc", "armed")

    def __init__(self, n):
        self.left, self.acc, self.armed = n, None, False

    def _check(self):
        """end of the backward pass: every wired consumer must have reported -- otherwise a parked partial sum was dropped
        (torch.autograd.grad on a sub-graph, a caller back-propagating through one head only)"""
        if self.left > 0 and self.acc is not None:
            raise HipError("GradJoin: %d wired consumer(s) never ran their backward -- the gradient of the shared input is "
                           "incomplete (set U2PL_NO_RES_GRAD_FUSION=1 for partial backward passes)" % self.left)

    def settle(self, t, included=False):
        """t: this consumer's contribution (already containing `acc` when included) -> the tensor to return to autograd"""
        if not self.armed:
            self.armed = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._check)
            except RuntimeError:        # (called outside a backward pass -- host-logic tests: there is no pass whose end to check)
                pass
        if self.acc is not None and not included:
            t.add_(self.acc)
        self.left -= 1
        if self.left > 0:
            self.acc = t
            return None
        self.acc = None
        return t


def grad_join(x, n):
    """a join for n consumers of x, or None (no graph being recorded / x needs no gradient / fewer than two consumers)"""
    if n >= 2 and FUSE_RES_GRAD and torch.is_grad_enabled() and x.requires_grad:
        return GradJoin(n)
    return None


def residual_grad_link(x):
    """bottleneck without a downsample branch: the block input feeds conv1 and the residual add of bn3"""
    return grad_join(x, 2)


def dgrad_fusable(conv, x):
    """does this conv's data gradient have a fused accumulate form (pointwise stride-1 on the pre-split kernels, or Winograd)?"""
    Cout, Cin, R, S = conv.weight.shape
    H, W = x.shape[2], x.shape[3]
    if R == 1 and S == 1 and conv.stride == 1 and conv.padding == 0:
        return _ws_ok(Cin, Cout) and Cin % 4 == 0 and Cout % 32 == 0
    return bool(wino_tile(Cout, Cin, R, S, conv.stride, conv.padding, conv.dilation, H, W)) and Cout % 32 == 0 and Cin % 4 == 0


class Conv2d(nn.Module):
    """nn.Conv2d(groups=1) with torch's default init, weight stored channels_last."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        k = kernel_size
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (k, k), stride, padding, dilation
        w = torch.empty(out_channels, in_channels, k, k)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))  # == nn.Conv2d.reset_parameters (same RNG draws)
        self.weight = nn.Parameter(w.contiguous(memory_format=_CL))
        if bias:
            fan_in = in_channels * k * k
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            b = torch.empty(out_channels)
            nn.init.uniform_(b, -bound, bound)
            self.bias = nn.Parameter(b)
        else:
            self.register_parameter("bias", None)

    def forward(self, x, stat_pivot=None, grad_link=None):
        """stat_pivot: running_mean of a following train-mode BatchNorm -> returns (y, fused BN sums).
        grad_link: a GradJoin shared with the other consumers of x (grad_join)."""
        out = _ConvFn.apply(x, self.weight, self.bias, self.stride, self.padding, self.dilation,
                            _grad_sink(self.weight), _grad_sink(self.bias) if self.bias is not None else None, stat_pivot,
                            torch.is_grad_enabled(), grad_link)
        return out

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, k={self.kernel_size}, s={self.stride}, p={self.padding}, d={self.dilation}"


# ------------------------------------------------------------------ batch norm (+res +relu +dropout)
def finished_sums(pre, C, out=None):
    """conv-epilogue statistics -> double [2C(+1)] pivot-shifted sums: `pre` is either the raw float32 [nblk][2][C] partials
    (ordered finish here) or already-finished double sums"""
    if pre.dtype == torch.float32:
        if out is None:
            out = torch.empty(2 * C + 1, dtype=torch.float64, device=pre.device)
        call("u2pl_colreduce_finish_f32", pre, pre.shape[0], C, out)
        return out
    if out is not None:
        out.copy_(pre)
        return out
    return pre


class _BNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, drop, mod, relu, gsink, bsink, pre_sums=None, link=None):
        x, ldx = as_rows(x)
        N, C, H, W = x.shape
        M = N * H * W
        dev = x.device
        ctx.link = link if res is not None else None
        rr = ldr = None
        if res is not None:
            rr, ldr = as_rows(res)
        y = new_act(N, C, H, W, dev)
        training = mod.training
        sync = mod.sync and dist_active()
        if training:
            pivot = mod.running_mean
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
            count = float(M)
            if pre_sums is not None and pre_sums.dtype == torch.float32 and not sync and FUSE_BN_FINISH:
                # single rank: ordered finish of the conv epilogue's partials + finalisation in ONE launch (same bits)
                call("u2pl_bn_finish_finalize_f32", pre_sums, pre_sums.shape[0], C, count, pivot, mod.eps, mod.momentum, mean,
                     invstd, mod.running_mean, mod.running_var, None)
            else:
                if pre_sums is not None:     # statistics came out of the producing conv's epilogue
                    sums = finished_sums(pre_sums, C)
                else:
                    sums = torch.empty(2 * C + 1, dtype=torch.float64, device=dev)
                    wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), dev)
                    call("u2pl_bn_stats_f32", x, ldx, M, C, pivot, wsb, sums)
                if sync:
                    # (2C doubles travel; the row count does not: equal per-rank shapes -- drop_last loaders -- make it
                    #  M * world on every rank.  Writing it into a slot of the buffer from the host cost a torch setitem per layer:
                    #  ~100 us of host time, 21 ms per step -- measured on RCCL in a world of one, tools/host_overhead.py)
                    _all_reduce(sums[:2 * C], "syncbn_allreduce", group=mod.group)
                    count = float(M * _world())
                call("u2pl_bn_finalize_f32", sums, count, pivot, C, mod.eps, mod.momentum, mean, invstd, mod.running_mean,
                     mod.running_var)
            mod._nbt += 1   # host counter; the buffer is materialised lazily (see BatchNorm2d)
        else:
            mean = mod.running_mean
            invstd = _eval_invstd(mod, C, dev)
            count = float(M)
        if CONV_H["on"]:        # split-fp16: max |y| for the convolutions that read y
            y_amax = amax_slot(dev)
            call("u2pl_bn_apply_amax_f32", x, ldx, mean, invstd, gamma, beta, rr, ldr or 0, int(relu), drop, H * W, y, C, M, C, y_amax)
            set_amax(y, y_amax)
        else:
            call("u2pl_bn_apply_f32", x, ldx, mean, invstd, gamma, beta, rr, ldr or 0, int(relu), drop, H * W, y, C, M, C)
        # backward's ReLU mask: [y > 0] read from y, or -- no residual -- recomputed from x with the forward's own expression (same
        # bits; y is then neither saved nor read: 4 of 12-16 bytes per element in each of the two backward passes)
        ctx.mask_from_x = bool(relu and res is None and RELU_MASK_FROM_X and beta is not None)
        ctx.save_for_backward(x, y if (relu and not ctx.mask_from_x) else None, mean, invstd, gamma, drop,
                              beta if ctx.mask_from_x else None)
        ctx.meta = (N, C, H, W, ldx, training, sync, count, res is not None)
        ctx.gsink, ctx.bsink, ctx.group = gsink, bsink, mod.group
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, invstd, gamma, drop, rbeta = ctx.saved_tensors
        N, C, H, W, ldx, training, sync, count, has_res = ctx.meta
        M = N * H * W
        gy, ldg = as_rows(gy)
        dev = gy.device
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), dev)
        if rbeta is not None:
            call("u2pl_bn_bwd_sums_mx_f32", gy, ldg, x, ldx, mean, invstd, gamma, rbeta, drop, H * W, M, C, wsb, sums)
        else:
            call("u2pl_bn_bwd_sums_f32", gy, ldg, x, ldx, y, C, mean, invstd, drop, H * W, M, C, wsb, sums)
        dgamma = dbeta = None
        # single rank, gradients into the arena: the two parameter-gradient writes ride in the apply launch (same arithmetic)
        pg_fused = (FUSE_BN_FINISH and ctx.needs_input_grad[1] and ctx.gsink is not None and ctx.needs_input_grad[0]
                    and (not (sync and training) or CONV_H["on"] or rbeta is not None))
        # (SyncBN: the parameter gradients are the LOCAL sums, the apply needs the all-reduced ones -- one device copy of the local
        #  sums rides into the apply launch instead of two parameter-gradient launches in front of the exchange)
        psums = sums
        # parameter gradients are LOCAL sums (DDP averages them later), like torch SyncBN
        if ctx.needs_input_grad[1] and not pg_fused:
            if ctx.gsink is not None:
                call("u2pl_sums_to_f32", sums[C:], C, 1.0, 1, ctx.gsink)
                call("u2pl_sums_to_f32", sums, C, 1.0, 1, ctx.bsink)
                _mark_ready(ctx.gsink)
                _mark_ready(ctx.bsink)
            else:
                dgamma = torch.empty(C, dtype=torch.float32, device=dev)
                dbeta = torch.empty(C, dtype=torch.float32, device=dev)
                call("u2pl_sums_to_f32", sums[C:], C, 1.0, 0, dgamma)
                call("u2pl_sums_to_f32", sums, C, 1.0, 0, dbeta)
        if sync and training:
            if pg_fused:
                psums = sums.clone()
            _all_reduce(sums, "syncbn_allreduce", group=ctx.group)
        dx = new_act(N, C, H, W, dev) if ctx.needs_input_grad[0] else None
        dres = new_act(N, C, H, W, dev) if has_res and ctx.needs_input_grad[3] else None
        if (CONV_H["on"] or rbeta is not None) and dx is not None:
            # split-fp16: max |dx| / max |dres| for the data / weight gradients that read them; rbeta: the mask from x
            dx_amax = amax_slot(dev) if CONV_H["on"] else None
            dres_amax = amax_slot(dev) if (dres is not None and CONV_H["on"]) else None
            call("u2pl_bn_bwd_apply_amax_f32", gy, ldg, x, ldx, y, C, mean, invstd, gamma, drop, H * W,
                 sums if training else None, count, dx, C, dres, C, M, C, psums if pg_fused else None,
                 ctx.gsink if pg_fused else None, ctx.bsink if pg_fused else None, 1, dx_amax, dres_amax, rbeta)
            if dx_amax is not None:
                set_amax(dx, dx_amax)
            if dres_amax is not None:
                set_amax(dres, dres_amax)
            if pg_fused:
                _mark_ready(ctx.gsink)
                _mark_ready(ctx.bsink)
        elif pg_fused:
            call("u2pl_bn_bwd_apply_pg_f32", gy, ldg, x, ldx, y, C, mean, invstd, gamma, drop, H * W,
                 sums if training else None, count, dx, C, dres, C, M, C, sums, ctx.gsink, ctx.bsink, 1)
            _mark_ready(ctx.gsink)
            _mark_ready(ctx.bsink)
        elif dx is not None:
            call("u2pl_bn_bwd_apply_f32", gy, ldg, x, ldx, y, C, mean, invstd, gamma, drop, H * W,
                 sums if training else None, count, dx, C, dres, C, M, C)
        if ctx.link is not None and dres is not None:
            dres = ctx.link.settle(dres)    # GradJoin: parked for the block's other consumer (conv1), or the total if it ran first
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None


# ---- packed SyncBatchNorm exchanges ------------------------------------------------------------------------------------
# A BatchNorm's statistics depend on the previous layer's normalised output, so consecutive layers of one network cannot
# share a collective -- but INDEPENDENT BatchNorms can: the five ASPP branches (base.py:23-83) and, in the first
# bottleneck of every ResNet stage, bn3 and the downsample BatchNorm (resnet.py:120-140).  Each group exchanges ONE packed
# buffer forward and one backward instead of one all-reduce per layer.  Only used in train mode under a
# process group with sync=True; everything else goes through _BNFn unit by unit (the single-GPU path is untouched).
def _bn_apply(x, ldx, mean, invstd, gamma, beta, res, ldr, relu, drop, hw, y, ldy, M, C):
    """u2pl_bn_apply_f32; with split-fp16 on, the form that leaves max |y| for the convolutions that read y"""
    if CONV_H["on"]:
        y_amax = amax_slot(y.device)
        call("u2pl_bn_apply_amax_f32", x, ldx, mean, invstd, gamma, beta, res, ldr, relu, drop, hw, y, ldy, M, C, y_amax)
        set_amax(y, y_amax)
    else:
        call("u2pl_bn_apply_f32", x, ldx, mean, invstd, gamma, beta, res, ldr, relu, drop, hw, y, ldy, M, C)


def _bn_local_sums(x, ldx, M, C, mod, pre_sums, out):
    """pivot-shifted sums of one BatchNorm into out (double [2C+1], slot 2C = the local row count)"""
    if pre_sums is not None:
        finished_sums(pre_sums, C, out)
    else:
        wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), x.device)
        call("u2pl_bn_stats_f32", x, ldx, M, C, mod.running_mean, wsb, out)
    # (slot 2C -- the row count of the packed protocol -- stays what the caller's zero fill left: the count is M * world by
    #  construction and a host-side setitem per unit costs ~100 us)


def _bn_finalize(sums, count, mod, C, dev):
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    call("u2pl_bn_finalize_f32", sums, count, mod.running_mean, C, mod.eps, mod.momentum, mean, invstd, mod.running_mean,
         mod.running_var)
    mod._nbt += 1
    return mean, invstd


def _param_grads(sums, C, gsink, bsink, need, dev):
    """LOCAL parameter gradients from the (un-reduced) backward sums: dbeta = sums[:C], dgamma = sums[C:]"""
    if not need:
        return None, None
    if gsink is not None:
        call("u2pl_sums_to_f32", sums[C:], C, 1.0, 1, gsink)
        call("u2pl_sums_to_f32", sums, C, 1.0, 1, bsink)
        _mark_ready(gsink)
        _mark_ready(bsink)
        return None, None
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    call("u2pl_sums_to_f32", sums[C:], C, 1.0, 0, dgamma)
    call("u2pl_sums_to_f32", sums, C, 1.0, 0, dbeta)
    return dgamma, dbeta


class _BNGroupFn(torch.autograd.Function):
    """n independent train-mode SyncBatchNorms (+ReLU, Dropout2d scale) with ONE statistics all-reduce forward and ONE
    backward.  args: n, then per unit (x, gamma, beta, drop, pre_sums), then the per-unit python metadata list."""

    @staticmethod
    def forward(ctx, n, *flat):
        meta = flat[5 * n]                      # [(mod, relu, gsink, bsink)] * n
        units, total = [], 0
        for i in range(n):
            x, gamma, beta, drop, pre = flat[5 * i:5 * i + 5]
            x, ldx = as_rows(x)
            N, C, H, W = x.shape
            units.append((x, ldx, N, C, H, W, gamma, beta, drop, pre))
            total += 2 * C + 1
        dev = units[0][0].device
        group = meta[0][0].group
        packed = torch.zeros(total, dtype=torch.float64, device=dev)
        off = 0
        for (x, ldx, N, C, H, W, gamma, beta, drop, pre), (mod, relu, gs, bs) in zip(units, meta):
            _bn_local_sums(x, ldx, N * H * W, C, mod, pre, packed[off:off + 2 * C + 1])
            off += 2 * C + 1
        _all_reduce(packed, "syncbn_allreduce", group=group)
        outs, saved, ctx.meta, off = [], [], [], 0
        W_ = _world()
        for (x, ldx, N, C, H, W, gamma, beta, drop, pre), (mod, relu, gs, bs) in zip(units, meta):
            M = N * H * W
            count = float(M * W_)
            mean, invstd = _bn_finalize(packed[off:off + 2 * C + 1], count, mod, C, dev)
            off += 2 * C + 1
            y = new_act(N, C, H, W, dev)
            _bn_apply(x, ldx, mean, invstd, gamma, None if beta is None else beta, None, 0, int(relu), drop, H * W, y, C, M, C)
            outs.append(y)
            saved += [x, y if relu else None, mean, invstd, gamma, drop]
            ctx.meta.append((N, C, H, W, ldx, count, gs, bs))
        ctx.save_for_backward(*saved)
        ctx.n, ctx.group = n, group
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        n, sv = ctx.n, ctx.saved_tensors
        dev = gys[0].device
        total = sum(2 * m[1] for m in ctx.meta)
        packed = torch.empty(total, dtype=torch.float64, device=dev)
        work, off = [], 0
        for i in range(n):
            x, y, mean, invstd, gamma, drop = sv[6 * i:6 * i + 6]
            N, C, H, W, ldx, count, gs, bs = ctx.meta[i]
            M = N * H * W
            gy, ldg = as_rows(gys[i])
            sums = packed[off:off + 2 * C]
            off += 2 * C
            wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), dev)
            call("u2pl_bn_bwd_sums_f32", gy, ldg, x, ldx, y, C, mean, invstd, drop, H * W, M, C, wsb, sums)
            dg, db = _param_grads(sums, C, gs, bs, ctx.needs_input_grad[1 + 5 * i + 1], dev)
            work.append((gy, ldg, x, ldx, y, mean, invstd, gamma, drop, N, C, H, W, count, sums, dg, db))
        _all_reduce(packed, "syncbn_allreduce", group=ctx.group)
        grads = [None]
        for i, (gy, ldg, x, ldx, y, mean, invstd, gamma, drop, N, C, H, W, count, sums, dg, db) in enumerate(work):
            dx = None
            if ctx.needs_input_grad[1 + 5 * i]:
                dx = new_act(N, C, H, W, dev)
                call("u2pl_bn_bwd_apply_f32", gy, ldg, x, ldx, y, C, mean, invstd, gamma, drop, H * W, sums, count, dx, C, None, C,
                     N * H * W, C)
            grads += [dx, dg, db, None, None]
        grads.append(None)
        return tuple(grads)


class _BNResPairFn(torch.autograd.Function):
    """out = relu(bn_a(xa) + bn_b(xb)) -- a bottleneck's bn3 and its downsample BatchNorm (resnet.py:120-140), train mode,
    SyncBatchNorm: the two statistics exchanges travel in one all-reduce, forward and backward (both backward sums are taken
    from the same masked gradient: d out / d bn_a = d out / d bn_b = relu mask)."""

    @staticmethod
    def forward(ctx, xa, ga, ba, pre_a, xb, gb, bb, pre_b, meta):
        (mod_a, gsa, bsa), (mod_b, gsb, bsb) = meta
        xa, lda = as_rows(xa)
        xb, ldb = as_rows(xb)
        N, C, H, W = xa.shape
        M, dev = N * H * W, xa.device
        packed = torch.zeros(2 * (2 * C + 1), dtype=torch.float64, device=dev)
        _bn_local_sums(xa, lda, M, C, mod_a, pre_a, packed[:2 * C + 1])
        _bn_local_sums(xb, ldb, M, C, mod_b, pre_b, packed[2 * C + 1:])
        _all_reduce(packed, "syncbn_allreduce", group=mod_a.group)
        count = float(M * _world())
        mean_a, inv_a = _bn_finalize(packed[:2 * C + 1], count, mod_a, C, dev)
        mean_b, inv_b = _bn_finalize(packed[2 * C + 1:], count, mod_b, C, dev)
        ident = new_act(N, C, H, W, dev)
        call("u2pl_bn_apply_f32", xb, ldb, mean_b, inv_b, gb, bb, None, 0, 0, None, H * W, ident, C, M, C)
        y = new_act(N, C, H, W, dev)
        _bn_apply(xa, lda, mean_a, inv_a, ga, ba, ident, C, 1, None, H * W, y, C, M, C)
        ctx.save_for_backward(xa, xb, y, mean_a, inv_a, mean_b, inv_b, ga, gb)
        ctx.meta = (N, C, H, W, lda, ldb, count, gsa, bsa, gsb, bsb, mod_a.group)
        return y

    @staticmethod
    def backward(ctx, gy):
        xa, xb, y, mean_a, inv_a, mean_b, inv_b, ga, gb = ctx.saved_tensors
        N, C, H, W, lda, ldb, count, gsa, bsa, gsb, bsb, group = ctx.meta
        M, dev = N * H * W, gy.device
        gy, ldg = as_rows(gy)
        packed = torch.empty(4 * C, dtype=torch.float64, device=dev)
        sa, sb = packed[:2 * C], packed[2 * C:]
        wsb = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), dev)
        call("u2pl_bn_bwd_sums_f32", gy, ldg, xa, lda, y, C, mean_a, inv_a, None, H * W, M, C, wsb, sa)
        wsb2 = _ws(query("u2pl_colreduce_workspace_bytes", M, 1, C), dev)
        call("u2pl_bn_bwd_sums_f32", gy, ldg, xb, ldb, y, C, mean_b, inv_b, None, H * W, M, C, wsb2, sb)
        dga, dba = _param_grads(sa, C, gsa, bsa, ctx.needs_input_grad[1], dev)
        dgb, dbb = _param_grads(sb, C, gsb, bsb, ctx.needs_input_grad[5], dev)
        _all_reduce(packed, "syncbn_allreduce", group=group)
        dxa = dxb = None
        if ctx.needs_input_grad[0]:
            dxa = new_act(N, C, H, W, dev)
            call("u2pl_bn_bwd_apply_f32", gy, ldg, xa, lda, y, C, mean_a, inv_a, ga, None, H * W, sa, count, dxa, C, None, C, M, C)
        if ctx.needs_input_grad[4]:
            dxb = new_act(N, C, H, W, dev)
            call("u2pl_bn_bwd_apply_f32", gy, ldg, xb, ldb, y, C, mean_b, inv_b, gb, None, H * W, sb, count, dxb, C, None, C, M, C)
        return dxa, dga, dba, None, dxb, dgb, dbb, None, None


def _packable(bns):
    return (dist_active() and os.environ.get("U2PL_NO_SYNCBN_PACK") is None
            and all(b.training and b.sync for b in bns) and len({id(b.group) for b in bns}) == 1)


def _conv_with_stats(conv, bn, x, grad_link=None):
    """conv output + its fused train-mode BN sums (None where the epilogue form does not exist: stem, pooled 1x1)"""
    if conv.in_channels % 32 == 0:
        y, sums = conv(x, stat_pivot=bn.running_mean, grad_link=grad_link)
        return y, sums
    return conv(x, grad_link=grad_link), None


def conv_bn_group(units):
    """units: [(conv, bn, x, relu, drop)], mutually independent -> list of outputs.  Under a process group in train mode
    the SyncBatchNorm statistics of all units are exchanged in ONE all-reduce (forward and backward)."""
    units = [tuple(u) + (None,) * (6 - len(u)) for u in units]        # optional 6th entry: a GradJoin for the conv's input
    if not _packable([u[1] for u in units]) or len(units) < 2:
        return [conv_bn(c, b, x, relu=r, drop=d, grad_link=j) for c, b, x, r, d, j in units]
    flat, meta = [], []
    for conv, bn, x, relu, drop, j in units:
        y, sums = _conv_with_stats(conv, bn, x, j)
        flat += [y, bn.weight, bn.bias, drop, sums]
        meta.append((bn, relu, _grad_sink(bn.weight), _grad_sink(bn.bias)))
    return list(_BNGroupFn.apply(len(units), *flat, meta))


def conv_bn_res_pair(conv_a, bn_a, xa, conv_b, bn_b, xb, grad_link_b=None):
    """relu(bn_a(conv_a(xa)) + bn_b(conv_b(xb))): the tail of a bottleneck with a downsample branch; grad_link_b: a GradJoin
    for xb (the block input, also consumed by the block's conv1)"""
    if not _packable([bn_a, bn_b]):
        identity = conv_bn(conv_b, bn_b, xb, grad_link=grad_link_b)
        return conv_bn(conv_a, bn_a, xa, res=identity, relu=True)
    ya, sa = _conv_with_stats(conv_a, bn_a, xa)
    yb, sb = _conv_with_stats(conv_b, bn_b, xb, grad_link_b)
    meta = ((bn_a, _grad_sink(bn_a.weight), _grad_sink(bn_a.bias)), (bn_b, _grad_sink(bn_b.weight), _grad_sink(bn_b.bias)))
    return _BNResPairFn.apply(ya, bn_a.weight, bn_a.bias, sa, yb, bn_b.weight, bn_b.bias, sb, meta)


def _bn_write_nbt(m, prefix, keep_vars):
    m.num_batches_tracked.fill_(m._nbt)


def _bn_read_nbt(m, incompatible):
    m._nbt = int(m.num_batches_tracked)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d / nn.SyncBatchNorm (base.py:6-8) with torch defaults (eps 1e-5,
    momentum 0.1, affine, running stats).  `sync=True` exchanges the per-channel
    sums across ranks (one small all-reduce per layer, fwd and bwd)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, sync=False):
        super().__init__()
        self.num_features, self.eps, self.momentum, self.sync = num_features, eps, momentum, sync
        self.group = None     # process group of the statistics exchange (None: the default group); see use_process_group
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        # torch bumps num_batches_tracked with one tiny kernel per BN per forward (236 launches / step);
        # the count is kept on the host and written into the buffer only when a state_dict is taken.
        self._nbt = 0
        self.register_state_dict_pre_hook(_bn_write_nbt)       # (module-level functions: lambdas would make the module unpicklable)
        self.register_load_state_dict_post_hook(_bn_read_nbt)

    def forward(self, x, res=None, relu=False, drop=None, pre_sums=None, res_link=None):
        return _BNFn.apply(x, self.weight, self.bias, res, drop, self, relu, _grad_sink(self.weight),
                           _grad_sink(self.bias), pre_sums, res_link)


FUSE_EVAL_BN = os.environ.get("U2PL_NO_EVAL_BN_FUSION") is None
# round 6: a BatchNorm + ReLU without a residual recomputes its backward's ReLU mask from x instead of reading y (same bits)
RELU_MASK_FROM_X = os.environ.get("U2PL_NO_RELU_MASK_FROM_X") is None
# round 5: fewer tiny launches on the conv -> statistics -> normalise chain (same arithmetic, same bits; U2PL_NO_BN_FINISH_FUSION=1:
# the separate launches): finish + finalize in one, parameter gradients inside the backward apply, eval-mode invstd of a whole
# model in one launch (eval_invstd)
FUSE_BN_FINISH = os.environ.get("U2PL_NO_BN_FINISH_FUSION") is None
_EVAL_INVSTD = {}
import weakref  # noqa: E402
_EVAL_PREP = weakref.WeakKeyDictionary()


def _eval_invstd(bn, C, dev):
    v = _EVAL_INVSTD.get(id(bn))
    if v is not None:
        return v
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    call("u2pl_bn_eval_invstd_f32", bn.running_var, C, bn.eps, invstd)
    return invstd


class eval_invstd:
    """with eval_invstd(model): <eval-mode pass(es)> -- 1 / sqrt(running_var + eps) of EVERY BatchNorm of the model by ONE launch
    up front (u2pl_bn_eval_invstd_multi_f32) instead of one launch per layer and pass.  Valid while no train-mode pass of the
    model runs inside the block (eval-mode passes do not touch the running statistics)."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        import numpy as np
        if not FUSE_BN_FINISH:
            self.ids = []
            return self
        m = self.model
        bns = [b for b in m.modules() if isinstance(b, BatchNorm2d)]
        key = tuple(b.running_var.data_ptr() for b in bns)
        prep = _EVAL_PREP.get(m)
        if prep is None or prep["key"] != key:
            if not bns or not bns[0].running_var.is_cuda:
                self.ids = []
                return self
            dev = bns[0].running_var.device
            total = sum(b.num_features for b in bns)
            out = torch.empty(total, dtype=torch.float32, device=dev)
            jobs = np.zeros(len(bns), dtype=np.dtype([("rv", "<u8"), ("out", "<u8"), ("begin", "<i8"), ("C", "<i4"), ("eps", "<f4")]))
            assert jobs.dtype.itemsize == query("u2pl_bn_eval_invstd_job_bytes")
            off, views = 0, []
            for i, b in enumerate(bns):
                jobs[i] = (b.running_var.data_ptr(), out.data_ptr() + 4 * off, off, b.num_features, b.eps)
                views.append(out[off:off + b.num_features])
                off += b.num_features
            from .hipops import h2d
            # (kept OUTSIDE the module -- a WeakKeyDictionary: torch.save(model) / deepcopy must not serialise device pointers, and a
            # copied model gets a prep of its own)
            prep = _EVAL_PREP[m] = dict(key=key, bns=bns, out=out, views=views, total=total,
                                        jobs=h2d(torch.from_numpy(jobs.view(np.uint8).copy()), dev))
        call("u2pl_bn_eval_invstd_multi_f32", prep["jobs"], len(prep["bns"]), prep["total"])
        self.ids = [id(b) for b in prep["bns"]]
        for b, v in zip(prep["bns"], prep["views"]):
            _EVAL_INVSTD[id(b)] = v
        return self

    def __exit__(self, *a):
        for i in self.ids:
            _EVAL_INVSTD.pop(i, None)
        return False


def conv_bn_eval(conv, bn, x, res=None, relu=False):
    """conv -> eval-mode BatchNorm (+residual, ReLU) in ONE pass over the conv output: the normalisation runs in the
    GEMM epilogue / the Winograd output transform (u2pl_conv2d_fwd_bnact_f32, u2pl_wino_output_bnact_f32) with the
    arithmetic of u2pl_bn_apply_f32, so the result has the bits of the two-kernel form.  No autograd graph: only for
    calls that record nothing (teacher pseudo-label pass, validate(), eval.py).  Returns None when the layer has no
    fused form (pooled 1x1 branch)."""
    x_in = x
    x, ldx = as_rows(x)
    N, Cin, H, W = x.shape
    weight, bias = conv.weight, conv.bias
    Cout, _, R, S = weight.shape
    stride, pad, dil = conv.stride, conv.padding, conv.dilation
    if H * W == 1 or Cout % 4:
        return None
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    dev = x.device
    invstd = _eval_invstd(bn, Cout, dev)
    rr, ldr = (None, 0) if res is None else as_rows(res)
    epi = (bn.running_mean, invstd, bn.weight, bn.bias, rr, ldr, int(relu))
    y = new_act(N, Cout, Ho, Wo, dev)
    if Cin % 32:
        Kp = ((R * S * Cin + 31) // 32) * 32
        col = torch.empty((N * Ho * Wo, Kp), dtype=torch.float32, device=dev)
        call("u2pl_im2col_f32", x, ldx, col, Kp, N, H, W, Cin, Ho, Wo, R, S, stride, pad, dil)
        wp = torch.zeros((Cout, Kp), dtype=torch.float32, device=dev)
        wp[:, : R * S * Cin] = weight.permute(0, 2, 3, 1).reshape(Cout, R * S * Cin)
        call("u2pl_conv2d_fwd_bnact_f32", col, Kp, wp, bias, y, Cout, N, Ho, Wo, Kp, Ho, Wo, Cout, 1, 1, 1, 0, 1, *epi)
    elif wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W):
        y_amax = amax_slot(dev) if CONV_H["on"] else None
        _wino_conv(x, ldx, weight, bias, y, N, H, W, Cin, Cout, dil, wino_tile(Cin, Cout, R, S, stride, pad, dil, H, W),
                   False, None, epi, y_amax)
        if y_amax is not None:
            set_amax(y, y_amax)
    elif _ws_ok(Cout, R * S * Cin) and CONV_H["on"]:
        y_amax = amax_slot(dev)
        call("u2pl_conv2d_fwd_bnact_wsh_f32", x, ldx, amax_of(x_in, x, ldx), ws_forward(weight, True), bias, y, Cout, N, H, W, Cin, Ho,
             Wo, Cout, R, S, stride, pad, dil, *epi, y_amax)
        set_amax(y, y_amax)
    elif _ws_ok(Cout, R * S * Cin):
        call("u2pl_conv2d_fwd_bnact_ws_f32", x, ldx, ws_forward(weight, False), bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout, R, S,
             stride, pad, dil, *epi)
    else:
        call("u2pl_conv2d_fwd_bnact_f32", x, ldx, weight, bias, y, Cout, N, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil,
             *epi)
    return y


def conv_bn(conv, bn, x, res=None, relu=False, drop=None, grad_link=None, res_link=None):
    """conv -> BatchNorm (+residual, ReLU, Dropout2d scale).  In training mode the BN statistics are
    produced by the conv kernel's epilogue (no separate read pass over the conv output); in eval mode without a
    recorded graph the whole BatchNorm runs in the conv's epilogue (conv_bn_eval)."""
    if (FUSE_EVAL_BN and not bn.training and drop is None
            and not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or bn.weight.requires_grad))):
        y = conv_bn_eval(conv, bn, x, res=res, relu=relu)
        if y is not None:
            return y
    if bn.training and conv.in_channels % 32 == 0:
        y, sums = conv(x, stat_pivot=bn.running_mean, grad_link=grad_link)
        return bn(y, res=res, relu=relu, drop=drop, pre_sums=sums, res_link=res_link)
    return bn(conv(x, grad_link=grad_link), res=res, relu=relu, drop=drop, res_link=res_link)


def use_process_group(model, group):
    """Route the SyncBN statistics exchange of every BatchNorm of `model` through `group`.  Collectives of one communicator
    execute in issue order on its own stream: the teacher's passes (side HIP stream) and the student's forward (main
    stream) only overlap across ranks when their per-layer all-reduces do not queue behind each other, so the trainer
    gives the teacher a communicator of its own."""
    for m in model.modules():
        if isinstance(m, BatchNorm2d):
            m.group = group


class SyncBatchNorm(BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__(num_features, eps, momentum, sync=True)


# Parity runs feed explicit keep-masks to every Dropout2d (hard part 10): a callable
# hook(module, N, C) -> (N, C) float32 CPU tensor holding 0 or 1/(1-p), or None to let the layer draw.
DROPOUT_HOOK = None


def dropout2d_scale(mod, N, C, device):
    """nn.Dropout2d(p) keep-mask as a per-(n,c) scale (0 or 1/(1-p)); None in eval mode."""
    p = mod.p
    if not mod.training or p <= 0:
        return None
    if DROPOUT_HOOK is not None:
        s = DROPOUT_HOOK(mod, N, C)
        if s is not None:
            return s.to(torch.float32).contiguous().pin_memory().to(device, non_blocking=True)
    u = None
    if DROPOUT_POOL is not None:      # uniforms of this PASS drawn ahead in one call (see dropout_pool)
        off, n = DROPOUT_POOL["off"], N * C
        if off + n <= DROPOUT_POOL["u"].numel():
            u = DROPOUT_POOL["u"][off:off + n].view(N, C)
            DROPOUT_POOL["off"] = off + n
        elif _lib.CAPTURING[0]:
            raise HipError("dropout pool exhausted inside a graph capture")
    if u is None:
        u = torch.rand((N, C), device=device)
    keep = (u >= p).to(torch.float32)
    return keep.mul_(1.0 / (1.0 - p))


# The keep-masks of ONE forward pass come from one block of uniforms drawn BEFORE the pass (one generator call per pass instead
# of one per Dropout2d layer).  Reason: a HIP graph that contains generator-driven kernels reads seed / offset from tensors that
# belong to the GENERATOR, refreshed at every replay on the replaying stream; the teacher's graph (side stream) and the student's
# (main stream) share the default generator, so one replay's refresh raced the other graph's kernels (measured: masks of the
# late layers differed from the eager step's).  With the uniforms in a buffer filled outside the graphs the passes draw
# identically eager or replayed, and no generator state is read inside a graph.
DROPOUT_POOL = None


class dropout_pool:
    """with dropout_pool(u): every Dropout2d of the enclosed pass takes its (N, C) uniforms from u, in call order"""

    def __init__(self, u):
        self.u = u

    def __enter__(self):
        global DROPOUT_POOL
        self.prev = DROPOUT_POOL
        DROPOUT_POOL = None if self.u is None else {"u": self.u, "off": 0}
        return self

    def __exit__(self, *a):
        global DROPOUT_POOL
        DROPOUT_POOL = self.prev
        return False


def dropout_uniforms_needed(model, N):
    """N x (channels in front of every Dropout2d of the model): the size of a pass's uniform block (an upper bound when the
    pass skips heads)"""
    total = 0
    for seq in model.modules():
        if isinstance(seq, nn.Sequential):
            mods = list(seq)
            for j, m in enumerate(mods):
                if isinstance(m, nn.Dropout2d) and m.p > 0:
                    bn = next((q for q in reversed(mods[:j]) if isinstance(q, BatchNorm2d)), None)
                    total += N * (bn.num_features if bn is not None else 0)
    return total


def run_seq(seq, x, grad_link=None):
    """Execute an nn.Sequential of {Conv2d, BatchNorm2d, ReLU, Dropout2d} fusing
    BN + ReLU + Dropout2d into one kernel (indices/names unchanged).  grad_link: a GradJoin for x, given to the Sequential's
    FIRST convolution (the consumer of x)."""
    mods = list(seq)
    i = 0
    pending_conv = None
    first_conv = next((m for m in mods if isinstance(m, Conv2d)), None) if grad_link is not None else None
    if grad_link is not None and (first_conv is None or first_conv is not mods[0]):
        raise HipError("run_seq: a gradient join needs the Sequential to start with its convolution")
    while i < len(mods):
        m = mods[i]
        if isinstance(m, Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], BatchNorm2d):
            pending_conv = m      # executed together with the BatchNorm that follows (fused statistics)
            i += 1
            continue
        if isinstance(m, BatchNorm2d):
            relu, drop, j = False, None, i + 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu, j = True, j + 1
            if j < len(mods) and isinstance(mods[j], nn.Dropout2d):
                drop = dropout2d_scale(mods[j], x.shape[0], m.num_features, x.device)
                j += 1
            if pending_conv is not None:
                x = conv_bn(pending_conv, m, x, relu=relu, drop=drop, grad_link=grad_link if pending_conv is first_conv else None)
                pending_conv = None
            else:
                x = m(x, relu=relu, drop=drop)
            i = j
        elif isinstance(m, (nn.ReLU, nn.Dropout2d)):
            raise HipError("un-fused ReLU/Dropout2d marker in Sequential")
        else:
            x = m(x)
            i += 1
    return x


# ------------------------------------------------------------------ pooling / resize / concat
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x, ldx = as_rows(x)
        N, C, H, W = x.shape
        Ho = -(-(H + 2 - 3) // 2) + 1
        Wo = -(-(W + 2 - 3) // 2) + 1
        if (Ho - 1) * 2 >= H + 1:
            Ho -= 1
        if (Wo - 1) * 2 >= W + 1:
            Wo -= 1
        y = new_act(N, C, Ho, Wo, x.device)
        tap = torch.empty((N * Ho * Wo, C), dtype=torch.uint8, device=x.device)
        call("u2pl_maxpool3s2_fwd_f32", x, ldx, N, H, W, C, Ho, Wo, y, C, tap)
        ctx.save_for_backward(tap)
        ctx.meta = (N, C, H, W, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, gy):
        (tap,) = ctx.saved_tensors
        N, C, H, W, Ho, Wo = ctx.meta
        gy, ldg = as_rows(gy)
        dx = new_act(N, C, H, W, gy.device)
        call("u2pl_maxpool3s2_bwd_f32", gy, ldg, tap, N, H, W, C, Ho, Wo, dx, C)
        return dx


class MaxPool3x3s2Ceil(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True) (resnet.py:189-191)."""

    def forward(self, x):
        return _MaxPoolFn.apply(x)


class _GapFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d((1,1)) (base.py:24)."""

    @staticmethod
    def forward(ctx, x):
        x, ldx = as_rows(x)
        N, C, H, W = x.shape
        sums = torch.empty((N, 2, C), dtype=torch.float64, device=x.device)
        wsb = _ws(query("u2pl_colreduce_workspace_bytes", H * W, N, C), x.device)
        call("u2pl_colsum_f32", x, ldx, H * W, N, C, wsb, sums)
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        for n in range(N):
            call("u2pl_sums_to_f32", sums[n], C, 1.0 / (H * W), 0, y[n])
        ctx.meta = (N, C, H, W)
        return y.reshape(N, C, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.meta
        g = gy.reshape(N, C).contiguous()
        dx = new_act(N, C, H, W, g.device)
        call("u2pl_broadcast_rows_f32", g, C, 1.0 / (H * W), dx, C, H * W, N * H * W, C)
        return dx


def global_avg_pool(x):
    return _GapFn.apply(x)


class _BroadcastFn(torch.autograd.Function):
    """bilinear(align_corners=True) up-sampling of a 1x1 map == broadcast (base.py:92-94)."""

    @staticmethod
    def forward(ctx, v, H, W):
        N, C = v.shape[0], v.shape[1]
        vv = v.reshape(N, C).contiguous()
        y = new_act(N, C, H, W, v.device)
        call("u2pl_broadcast_rows_f32", vv, C, 1.0, y, C, H * W, N * H * W, C)
        ctx.meta = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.meta
        gy, ldg = as_rows(gy)
        sums = torch.empty((N, 2, C), dtype=torch.float64, device=gy.device)
        wsb = _ws(query("u2pl_colreduce_workspace_bytes", H * W, N, C), gy.device)
        call("u2pl_colsum_f32", gy, ldg, H * W, N, C, wsb, sums)
        g = torch.empty((N, C), dtype=torch.float32, device=gy.device)
        for n in range(N):
            call("u2pl_sums_to_f32", sums[n], C, 1.0, 0, g[n])
        return g.reshape(N, C, 1, 1), None, None


class _UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        x, ldx = as_rows(x)
        N, C, h, w = x.shape
        y = new_act(N, C, H, W, x.device)
        call("u2pl_bilinear_rows_fwd_f32", x, ldx, N, h, w, C, H, W, y, C)
        ctx.meta = (N, C, h, w, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, C, h, w, H, W = ctx.meta
        gy, ldg = as_rows(gy)
        dx = new_act(N, C, h, w, gy.device)
        call("u2pl_bilinear_rows_bwd_f32", gy, ldg, N, h, w, C, H, W, dx, C)
        return dx, None, None


def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) on feature maps."""
    H, W = int(size[0]), int(size[1])
    if x.shape[2] == 1 and x.shape[3] == 1:
        return _BroadcastFn.apply(x, H, W)
    return _UpsampleFn.apply(x, H, W)


class _CatFn(torch.autograd.Function):
    """torch.cat(tensors, dim=1) on NHWC rows: strided row copies into channel slices."""

    @staticmethod
    def forward(ctx, *xs):
        N, _, H, W = xs[0].shape
        Cs = [t.shape[1] for t in xs]
        out = new_act(N, sum(Cs), H, W, xs[0].device)
        o = 0
        for t, c in zip(xs, Cs):
            t, ld = as_rows(t)
            call("u2pl_copy_rows_f32", t, ld, out[:, o:o + c], sum(Cs), N * H * W, c, 0)
            o += c
        ctx.Cs = Cs
        return out

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for c in ctx.Cs:
            outs.append(g[:, o:o + c])
            o += c
        return tuple(outs)


def cat_channels(tensors):
    return _CatFn.apply(*tensors)
from .arena import ParamArena  # noqa: E402,F401
