"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
prescribes) of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline` into profiles/rNN_traffic.json
(usage: parse_pmc_traffic.py <fetch_dir> <write_dir> <out.json> 6 10 [commit]  -- 6 full steps (1 warm-up + 3 timed + the recorded
roofline step + the lr-0.01 sanity step), 10 replay repetitions of the HBM-bound group).
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced stream -> x2."""
import collections
import csv
import glob
import json
import sys


def _sha():
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from u2pl_amd.roofline import kernel_source_hash
    return kernel_source_hash()


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    per = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[k][0] += float(r["Counter_Value"])
        per[k][1] += 1
    return per


def main(fetch_dir, write_dir, out, steps_profiled, hbm_reps=0, commit="?"):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        f, nf = fe.get(k, [0.0, 0])
        w, nw = wr.get(k, [0.0, 0])
        n = max(nf, nw, 1)
        rows[k] = dict(launches=n, fetch_KiB_raw=f, write_KiB=w, bytes_per_launch=(2 * f + w) * 1024 / n)
    ig = [v for k, v in rows.items() if k.startswith(("k_conv_igemm", "k_igemm_ws"))]
    ig_l = sum(v["launches"] for v in ig)
    ig_b = sum(v["bytes_per_launch"] * v["launches"] for v in ig) / max(ig_l, 1)
    hbm_names = ("k_entropy", "k_sel_", "k_reliability", "k_apply_drop", "k_contra", "k_compact", "k_proto", "k_phase1", "k_bank",
                 "k_infonce", "k_scatter_add", "k_scatter_rows", "k_zero_rows")
    # bench.py's roofline leg re-issues every stage of the HBM-bound group hbm_reps more times (replay_hbm_group)
    hb = sum(v["bytes_per_launch"] * v["launches"] for k, v in rows.items() if k.startswith(hbm_names)) / (steps_profiled + hbm_reps)
    res = dict(note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py; bytes = (2*FETCH_SIZE + "
                    "WRITE_SIZE) KiB (gfx950 FETCH_SIZE x2 correction, WRITE_SIZE uncalibrated); Infinity-Cache hits are counted",
               commit=commit, kernel_sources_sha=_sha(),
               igemm_bytes_per_launch=round(ig_b), igemm_launches=ig_l,
               # by SET COUNT: total / (steps_profiled + one dense replay).  Over-estimates: bench.py's calibration passes (~800 more
               # launches of the group) are in the total but not in the divisor; bench.py therefore reports bytes_per_launch x its own
               # launches per step (473) as traffic_per_step and keeps this figure as traffic_per_step_by_set_count
               igemm_bytes_per_step=round(ig_b * ig_l / (steps_profiled + 1)),
               k_conv_igemm_bytes_per_launch=round(ig_b), k_conv_igemm_launches=ig_l,
               hbm_group_bytes_per_step=round(hb), kernels={k: v for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:25]})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("commit", "kernel_sources_sha", "igemm_bytes_per_launch", "igemm_launches", "igemm_bytes_per_step",
                                          "hbm_group_bytes_per_step")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0,
         sys.argv[6] if len(sys.argv) > 6 else "?")
