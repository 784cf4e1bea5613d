"""Markdown table of the CURRENT numbers from one bench.py JSON line.

    python tools/design_numbers.py BENCH.json [BENCH_K20.json] > table.md

The table in DESIGN.md is this script's output on the bench lines of the round's final build (kept under profiles/)."""
import json
import sys


def load(path):
    with open(path) as f:
        text = f.read()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    d = json.loads(lines[-1])
    if "run" in d and "stdout_tail" in d.get("run", {}):  # a driver record (BENCH_rNN.json): the line is inside
        raise SystemExit("pass the bench line itself, not the driver's record")
    return d


def main():
    d = load(sys.argv[1])
    k20 = load(sys.argv[2]) if len(sys.argv) > 2 else None
    e = d["extras"]
    rf = d["roofline"]
    rows = []

    def row(what, us, nbytes, frac, note=""):
        rows.append("| %s | %s | %s | %s | %s |" % (what, "%.2f" % us if us is not None else "–",
                                                  "%.1f MB" % (nbytes / 1e6) if nbytes else "–",
                                                  "**%.3f**" % frac if frac is not None else "–", note))

    row("**headline**: per-channel int8 QDQ, 4096² bf16 → bf16 (`%s`)" % rf["kernel"].replace("sbq::", ""), rf["kernel_avg_us"],
        rf["algorithmic_bytes_per_launch"], rf["frac"],
        "event windows %.2f–%.2f µs; `frac_rocprof` %s (committed profile, %s µs); `frac_wall` %.3f at K = %d%s; traffic %s B = %.3f × algorithmic"
        % (rf["kernel_avg_us_windows_min"], rf["kernel_avg_us_windows_max"], rf.get("frac_rocprof"), rf.get("rocprof_kernel_avg_us"),
           rf["frac_wall"], d["steps"], (", %.3f at K = %d (value %.3e)" % (k20["roofline"]["frac_wall"], k20["steps"], k20["value"])) if k20 else "",
           rf.get("traffic"), (rf["traffic"] / rf["algorithmic_bytes_per_launch"]) if rf.get("traffic") else float("nan")))
    row("same, cache-resident (one buffer pair)", e["cache_resident_us"], 67108864, 67108864 / e["cache_resident_us"] / 1e3 / 8000)
    row("bf16 → fp32 out (parity mode)", e["bf16_to_fp32_us"], 4096 * 4096 * 6, 4096 * 4096 * 6 / e["bf16_to_fp32_us"] / 1e3 / 8000)
    row("per-channel min-max observer of the weight", e["minmax_observer_us"], 4096 * 4096 * 2, 4096 * 4096 * 2 / e["minmax_observer_us"] / 1e3 / 8000)
    row("fused observe + qparams + QDQ (one read)", e["fused_observe_qdq_us"], 67108864, 67108864 / e["fused_observe_qdq_us"] / 1e3 / 8000,
        "vs %.2f + %.2f µs as two launches" % (e["minmax_observer_us"], rf["kernel_avg_us"]))
    row("53 ResNet-50 weights, ONE launch (fp32, 4 bit)", e["resnet50_53_weights_one_launch_us"], None,
        e["resnet50_weights_one_launch_GBps"] / 8000, "%.0f µs one launch per layer" % e["resnet50_weights_launch_per_layer_us"])
    cfg = e["configs"]

    def legs(prefix, dct):
        for k, v in dct.items():
            if isinstance(v, dict) and "us" in v:
                extra = []
                if "frac_of_fp32_vector_peak" in v:
                    extra.append("%.3f of the fp32 peak" % v["frac_of_fp32_vector_peak"])
                if "frac_of_fp32_matrix_peak" in v:
                    extra.append("%.3f of the fp32 matrix peak" % v["frac_of_fp32_matrix_peak"])
                if "one_launch_per_matrix_us" in v:
                    extra.append("%.2f µs as one launch per matrix" % v["one_launch_per_matrix_us"])
                if "one_by_one_us" in v:
                    extra.append("%.0f µs one by one" % v["one_by_one_us"])
                if v.get("parity") is not True:
                    extra.append("GATE %s" % v.get("parity"))
                row("%s %s" % (prefix, k), v["us"], v["algorithmic_bytes"], v["frac"], "; ".join(extra))
                legs(prefix + " " + k + " /", v)
            elif isinstance(v, dict):
                legs(prefix + " " + k + " /", v)

    legs("config 1:", cfg["config1_resnet18_minmax_trt"])
    legs("config 2 (MSE):", {"per_channel": cfg["config2_mse_per_channel"]})
    legs("config 3 (percentile):", cfg["config3_percentile"])
    legs("config 4 (GPTQ 4-bit g128):", cfg["config4_gptq_4bit_g128"])
    legs("config 5 (mask + LSQ 4 bit):", cfg["config5_mask_lsq_4bit"])
    legs("model-wide calibration (53 tensors):", {k: v for k, v in e["model_wide_calibration"].items() if isinstance(v, dict)})
    print("| Leg | µs | algorithmic bytes | fraction of 8 TB/s | notes |")
    print("|---|---|---|---|---|")
    print("\n".join(rows))
    print()
    ar = e.get("rccl_world_size_1", {})
    if ar.get("ran"):
        lat = ar["latency"]
        print("RCCL, one rank (N = 1 floor): " + ", ".join("%s %.1f" % (k.replace("_us", ""), v) for k, v in lat.items() if k.endswith("_us")) + " µs")
    e2e = e.get("e2e_resnet20_b16_forward", {})
    if "eager_us" in e2e:
        print("ResNet-20 (QuantOpr harness), batch 16, host wall clock per forward: generic %.0f µs → plans %.0f → graph %.0f (frozen weights %.0f); "
              "float model %.0f eager / %.0f graph; quantizers inside the graph %.0f µs" % (
                  e2e["eager_us"], e2e["plan_us"], e2e["graph_us"], e2e["graph_frozen_weights_us"], e2e["float_model_eager_us"],
                  e2e["float_model_graph_us"], e2e["quantizers_cost_in_graph_us"]))
    cb = d.get("cpu_baseline") or {}
    if cb:
        print("CPU baseline (%s, %d of %d cores): %.3e elements/s; GPU / CPU = %.0f×" % (cb["kind"], cb["cores"], cb["host_cores"], cb["value"],
                                                                                         d["value"] / cb["value"]))
    print("value %.4e elements/s at K = %d (ms_per_step %.5f); all config gates pass: %s" % (d["value"], d["steps"], d["ms_per_step"],
                                                                                            e.get("all_config_gates_pass")))


if __name__ == "__main__":
    main()
