"""Micro-benchmark of the loss / reliability / contrastive kernels at BASELINE
config-3 sizes (769^2, B=2+2, C=19, bank pre-filled).  GPU only."""
import json
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import hipops as H  # noqa: E402
from u2pl_amd.utils import loss_helper as LH  # noqa: E402

DEV = "cuda"
CFG = dict(negative_high_entropy=True, low_rank=3, high_rank=20, current_class_threshold=0.3,
           current_class_negative_threshold=1, low_entropy_threshold=20, num_negatives=50, num_queries=256,
           temperature=0.5)


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    B, C, S, s, D = 2, 19, 769, 193, 256
    g = torch.Generator(device=DEV).manual_seed(2)
    low = (torch.randn(2 * B, C, s, s, device=DEV, generator=g) * 3).contiguous(memory_format=torch.channels_last)
    rep = torch.randn(2 * B, D, s, s, device=DEV, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rep_t = torch.randn(2 * B, D, s, s, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    label_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    label_l[:, :8] = 255
    res = {}
    large = H.bilinear_up(low[B:], (S, S))
    res["bilinear_up_us"] = timeit(lambda: H.bilinear_up(low[B:], (S, S)))
    _, label_u = H.pseudo_label(large + torch.randn(large.shape, device=DEV, generator=g))
    res["pseudo_label_us"] = timeit(lambda: H.pseudo_label(large))

    def rel():
        ws = H.new_select_ws(DEV, B * S * S)
        ent = H.entropy_map(large, label_u, ws)
        thr = H.run_select(ent, ws, [("pct", 80.0), ("pct", 20.0), ("pct", 80.0)])
        return ent, thr, H.reliability_masks(ent, thr[1:2], thr[2:3], label_l, label_u, (s, s))

    res["reliability_total_us"] = timeit(rel)

    def rel_fused():
        ws = H.new_select_ws(DEV, B * S * S)
        ent = H.entropy_map_up(low[B:], (S, S), label_u, ws)
        thr = H.run_select(ent, ws, [("pct", 80.0), ("pct", 20.0), ("pct", 80.0)])
        return H.reliability_apply(ent, thr, label_l, label_u, (s, s))

    res["reliability_fused_total_us"] = timeit(rel_fused)
    res["reliability_persistent_us"] = timeit(
        lambda: H.reliability_split(low[B:], (S, S), label_l, label_u, (s, s), [80.0, 20.0, 80.0], fused=True), n=40)
    wsf = H.new_select_ws(DEV, B * S * S)
    res["entropy_up_us"] = timeit(lambda: H.entropy_map_up(low[B:], (S, S), label_u, wsf))
    ws = H.new_select_ws(DEV, B * S * S)
    res["entropy_us"] = timeit(lambda: H.entropy_map(large, label_u, ws))
    ent, thr, (lo, hi, lbits) = rel()
    res["select3_us"] = timeit(lambda: H.run_select(ent, ws, [("pct", 80.0), ("pct", 20.0), ("pct", 80.0)]))
    res["masks_us"] = timeit(lambda: H.reliability_masks(ent, thr[1:2], thr[2:3], label_l, label_u, (s, s)))
    pred = large.clone().requires_grad_(True)
    tgt = label_u.clone()

    def ce():
        pred.grad = None
        LH.H.cross_entropy(pred, tgt, 255, True).backward()

    res["ce_fwd_bwd_us"] = timeit(ce)
    crit = LH.CriterionOhem(0.0, thresh=0.7, min_kept=100000)
    res["ohem_fwd_bwd_us"] = timeit(lambda: crit(pred, label_l).backward())
    prob = torch.softmax(low, 1).contiguous(memory_format=torch.channels_last)      # (the trainer's probabilities are channels-last rows)
    bank = H.DeviceMemoryBank(C, [50000] + [30000] * (C - 1), D, DEV)
    for c in range(C):
        bank.load_logical(c, torch.randn(bank.cap[c], D, device=DEV, generator=g))

    def contra():
        rep.grad = None
        keys, loss = LH.contra_memobank_core(rep, lbits, B, prob[:B], prob[B:], lo, hi, CFG, bank, rep_t)
        loss.backward()
        return keys

    keys = contra()
    res["contra_new_keys"] = [int(k) for k in keys]
    res["contra_fwd_bwd_us"] = timeit(contra, n=40, warm=3)     # (40 repetitions: the first, cold call -- ~1.5x -- would be 1/9 of a 9-call average)
    t0 = time.time()
    contra()
    torch.cuda.synchronize()
    res["contra_wall_us"] = (time.time() - t0) * 1e6
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
