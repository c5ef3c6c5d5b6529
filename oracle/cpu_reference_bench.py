"""TEST INFRASTRUCTURE ONLY -- the CPU baseline of SURVEY 8(d) / BASELINE.md section 3: the REFERENCE'S OWN
train() (train_semi.py:234-594, imported read-only through oracle/ref_shim.py) timed on this machine's host cores at the
headline configuration (R101-DeepLabv3+, 769x769, 2 labeled + 2 unlabeled, C=19, OHEM + aux, CutMix, contrastive
bank, dropout on), 1 warm-up + 2 timed optimizer steps, and the port (oracle/step_ref.CpuStepRef) the same way with
the same thread count.  /root/reference exists only in the build container, so this runs HERE; bench.py's
`cpu_baseline` leg times the port on the GPU box (kind "port") and quotes this file's reference figure next to it.

    python oracle/cpu_reference_bench.py [reference|port|both]  ->  profiles/r04_cpu_reference_timing.json (U2PL_CPU_TIMING_FILE: other name)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "profiles", os.environ.get("U2PL_CPU_TIMING_FILE", "r04_cpu_reference_timing.json"))


def time_reference(threads):
    sys.path.insert(0, HERE)
    import gen_golden as G
    import ref_shim
    ns = ref_shim.load()
    S, B, C, steps = 769, 2, 19, 3
    cfg = G._train_cfg(False, "resnet101", C, min_kept=100000, class_thr=0.3, epochs=200)
    data = G.survey_step_inputs(2, B, S, C, steps)
    t0 = time.perf_counter()
    r = G._run_reference_train(ns, cfg, data, steps, [0], False, p_drop=0.1, dropout_seed=1234, sharpen=4.0, threads=threads)
    wall = time.perf_counter() - t0
    bt = [float(x) for x in r["meters"][:, 5]]        # the loop's own batch_time meter (train_semi.py:563-565), seconds
    return dict(reference_train_s_per_step=bt, reference_timed_s_per_step=float(np.mean(bt[1:])),
                reference_images_per_s=float(2 * B / np.mean(bt[1:])), reference_total_wall_s=wall)


def time_port(threads):
    sys.path.insert(0, ROOT)
    os.environ["U2PL_CPU_BASELINE_THREADS"] = str(threads)
    from oracle import step_ref
    r = step_ref.timed_cpu_baseline(crop=769, arch="resnet101", batch=2, warmup=1, steps=2)
    return dict(port_s_per_step=r["s_per_step"], port_images_per_s=r["value"])


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    threads = int(os.environ.get("U2PL_CPU_BASELINE_THREADS", os.cpu_count() or 1))
    torch.set_num_threads(threads)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    out.update(machine="build container", cores=os.cpu_count(), threads=threads,
               config="R101-DeepLabv3+ 769x769, 2+2, C=19, OHEM+aux, CutMix, contrastive; dropout on; 1 warm-up + 2 timed steps")
    if what in ("reference", "both"):
        out.update(time_reference(threads))
    if what in ("port", "both"):
        out.update(time_port(threads))
    if "port_images_per_s" in out and "reference_images_per_s" in out:
        out["port_over_reference"] = out["port_images_per_s"] / out["reference_images_per_s"]
    print(json.dumps(out))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    json.dump(out, open(OUT, "w"), indent=1)


if __name__ == "__main__":
    main()
