"""frac of the fp32 MFMA peak from a rocprofv3 --kernel-trace CSV of tools/dense_replay.py:
    python tools/parse_dense_trace.py <kernel_trace.csv> <dense_replay.json> [out.json]
takes the LAST `kernel_launches` k_conv_igemm / k_igemm_ws rows (the final replay), span = max End - min Start."""
import csv
import json
import sys


def main(trace, meta, out=None):
    m = json.loads([l for l in open(meta).read().splitlines() if l.startswith("{")][-1])
    rows = [r for r in csv.DictReader(open(trace)) if "k_conv_igemm" in r["Kernel_Name"] or "k_igemm_ws" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = int(m["kernel_launches"])
    last = rows[-n:]
    t0, t1 = min(int(r["Start_Timestamp"]) for r in last), max(int(r["End_Timestamp"]) for r in last)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    span_ms = (t1 - t0) / 1e6
    res = dict(kernel_launches=n, igemm_rows_in_trace=len(rows), span_ms=span_ms, sum_of_durations_ms=busy / 1e6,
               executed_tflop=m["executed_tflop"], tflops_from_span=m["executed_tflop"] / span_ms * 1e3,
               frac_from_span=m["executed_tflop"] / span_ms * 1e3 / 157.3,
               # against the pipe the instructions run on: matrix-pipe FLOPs (executed x piece products per product) over 2500 TF
               piece_products_per_fp32_product=m.get("piece_products_per_fp32_product", 6.0),
               frac_of_split_bound_from_span=m.get("matrix_pipe_tflop", 6.0 * m["executed_tflop"]) / span_ms * 1e3 / 2500.0,
               frac_of_split_bound_from_sum_of_durations=m.get("matrix_pipe_tflop", 6.0 * m["executed_tflop"]) / (busy / 1e6) * 1e3 / 2500.0,
               frac_from_sum_of_durations=m["executed_tflop"] / (busy / 1e6) * 1e3 / 157.3,
               hip_event_ms=m["hip_event_ms"], frac_from_hip_events=m["frac_of_157.3"])
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
