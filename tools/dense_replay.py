"""The dense replay of bench.py's MFMA roofline ALONE, for a rocprofv3 --kernel-trace run: builds the headline workload,
records one step, then re-issues every k_conv_igemm launch of that step back to back (u2pl_amd.roofline.replay_dense) and
prints {executed FLOPs, HIP-event time, number of kernel launches in the replay}.  tools/parse_dense_trace.py turns the
kernel trace into `frac` = FLOPs / (max End - min Start of the replay's launches) / 157.3 TFLOP/s, i.e. the number
bench.py reports can be re-derived from a tracked CSV."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from u2pl_amd import _lib, configs, roofline as RL
    from u2pl_amd.models.model_helper import ModelBuilder
    from u2pl_amd.trainer import SemiTrainer
    from u2pl_amd.utils.loss_helper import get_criterion
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    np.random.seed(2)
    cfg = configs.cityscapes_semi(arch="resnet101", crop=769, batch_size=2, sync_bn=True)
    C = cfg["net"]["num_classes"]
    model, teacher = ModelBuilder(cfg["net"]).to(dev), ModelBuilder(cfg["net"]).to(dev)
    trainer = SemiTrainer(cfg, model, teacher, get_criterion(cfg), steps_per_epoch=163)
    gb = torch.Generator(device=dev).manual_seed(7)
    for c in range(C):
        trainer.memobank.load_logical(c, torch.randn(trainer.memobank.cap[c], 256, device=dev, generator=gb))
    gen = torch.Generator(device=dev).manual_seed(2)
    batches = [bench.synth_batch(2, 769, C, dev, gen) for _ in range(2)]
    trainer.base_lr = 1e-6
    gc = torch.Generator(device=dev).manual_seed(1234)
    calib = [bench.synth_batch(2, 769, C, dev, gc) for _ in range(2)]
    batches = bench.calibrate(model, teacher, calib, batches, 4.0)
    for i in range(2):
        trainer.train_step(*batches[i % 2], epoch=1)
    torch.cuda.synchronize()
    trainer._side = torch.cuda.current_stream()
    _lib.PROFILE = []
    trainer.train_step(*batches[0], epoch=1)
    torch.cuda.synchronize()
    rec, _lib.PROFILE = _lib.PROFILE, None
    names = RL.MFMA_GROUPS["igemm"]
    calls = [r for r in rec if r[0] in names]
    flops = sum(RL._conv_flops(r[0], r[1]) for r in calls)
    # matrix-pipe FLOPs: executed fp32 FLOPs x piece products per fp32 product (3: split-fp16 launches, 6: the bf16-split ones)
    pipe = sum(RL._conv_flops(r[0], r[1]) * RL.piece_products(r[0]) for r in calls)
    nker = sum(r[7] for r in calls)
    out = {}
    for rep in range(2):           # the LAST repetition is the one tools/parse_dense_trace.py evaluates
        sp = _lib.stream_ptr()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(40_000_000)
        e0.record()
        for r in calls:
            r[4](*r[5], sp)
        e1.record()
        torch.cuda.synchronize()
        out = dict(executed_tflop=flops / 1e12, matrix_pipe_tflop=pipe / 1e12, piece_products_per_fp32_product=pipe / flops,
                   hip_event_ms=e0.elapsed_time(e1), kernel_launches=nker, abi_calls=len(calls))
    out["tflops"] = out["executed_tflop"] / out["hip_event_ms"] * 1e3
    out["frac_of_157.3"] = out["tflops"] / 157.3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
