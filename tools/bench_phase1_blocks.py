"""Block timelines of the three phase-1 kernels of the contrastive loss at config-3 sizes (debug stamps armed through
u2pl_debug_phase1_times).  GPU only.  Needs the instrumented build of the library:
    python -m u2pl_amd.build_ext --variant p1dbg -DU2PL_P1_DBG"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["U2PL_LIB_PATH"] = os.path.join(ROOT, "u2pl_amd", "lib", "variants", "libu2pl_hip_p1dbg.so")
from u2pl_amd import hipops as H, _lib  # noqa: E402
from u2pl_amd.utils import loss_helper as LH  # noqa: E402
from tools.bench_loss_path import CFG  # noqa: E402

DEV = "cuda"


def main():
    B, C, S, s, D = 2, 19, 769, 193, 256
    g = torch.Generator(device=DEV).manual_seed(2)
    low = (torch.randn(2 * B, C, s, s, device=DEV, generator=g) * 3).contiguous(memory_format=torch.channels_last)
    rep = torch.randn(2 * B, D, s, s, device=DEV, generator=g).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rep_t = torch.randn(2 * B, D, s, s, device=DEV, generator=g).contiguous(memory_format=torch.channels_last)
    label_l = torch.randint(0, C, (B, S, S), device=DEV, generator=g)
    label_l[:, :8] = 255
    large = H.bilinear_up(low[B:], (S, S))
    _, label_u = H.pseudo_label(large + torch.randn(large.shape, device=DEV, generator=g))
    rs = H.reliability_split(low[B:], (S, S), label_l, label_u, (s, s), [80.0, 20.0, 80.0])
    lo, hi, lbits = rs["low_mask"], rs["high_mask"], rs["lbits"]
    prob = torch.softmax(low, 1).contiguous(memory_format=torch.channels_last)
    bank = H.DeviceMemoryBank(C, [50000] + [30000] * (C - 1), D, DEV)
    for c in range(C):
        bank.load_logical(c, torch.randn(bank.cap[c], D, device=DEV, generator=g))

    def contra():
        rep.grad = None
        keys, loss = LH.contra_memobank_core(rep, lbits, B, prob[:B], prob[B:], lo, hi, CFG, bank, rep_t)
        loss.backward()

    for _ in range(3):
        contra()
    buf = torch.zeros(3 * 4096 * 2, dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    out = {}
    acc = []
    for rep_i in range(5):
        buf.zero_()
        assert _lib.lib().cdll.u2pl_debug_phase1_times(_lib.ctypes.c_void_p(buf.data_ptr())) == 0
        contra()
        torch.cuda.synchronize()
        acc.append(buf.cpu().numpy().astype(np.int64).reshape(3, 4096, 2) & 0xFFFFFFFF)
    _lib.lib().cdll.u2pl_debug_phase1_times(None)
    a = acc[-1]
    names = ["classify", "proto_stream", "tail"]
    t_first = None
    for k in range(3):
        v = a[k]
        on = v[:, 1] != 0
        st, en = v[on, 0], v[on, 1]
        if t_first is None:
            t_first = st.min()
        dur = (en - st) / 100.0
        out[names[k]] = {
            "blocks": int(on.sum()),
            "first_start_us": float((st.min() - t_first) / 100.0),
            "last_start_us": float((st.max() - t_first) / 100.0),
            "last_end_us": float((en.max() - t_first) / 100.0),
            "span_us": float((en.max() - st.min()) / 100.0),
            "block_us_mean": float(dur.mean()), "block_us_p50": float(np.median(dur)), "block_us_max": float(dur.max()),
            "block_us_min": float(dur.min()),
        }
        for slot in (1, 2, 3):
            m = a[k][1024 * slot:1024 * slot + 1024, 0][on[:1024]]
            ok = m != 0
            if ok.any():
                d = ((m[ok] - st[:len(m)][ok]) & 0xFFFFFFFF) / 100.0
                out[names[k]]["mark%d_us_mean" % slot] = float(d.mean())
                out[names[k]]["mark%d_us_max" % slot] = float(d.max())
        if k == 2:
            nwb = (4 * s * s + 1023) // 1024
            w = np.arange(4096)[on] < nwb
            out["tail"]["write_block_us_mean"] = float(dur[w].mean())
            out["tail"]["write_block_us_max"] = float(dur[w].max())
            out["tail"]["finish_block_us_mean"] = float(dur[~w].mean())
            out["tail"]["finish_block_us_max"] = float(dur[~w].max())
            for slot in (1, 2):
                m = a[k][1024 * slot:1024 * slot + 1024, 0][on[:1024]]
                d = ((m - st) & 0xFFFFFFFF) / 100.0
                out["tail"]["write_mark%d_mean_max" % slot] = [float(d[w].mean()), float(d[w].max())]
                out["tail"]["finish_mark%d_mean_max" % slot] = [float(d[~w].mean()), float(d[~w].max())]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
