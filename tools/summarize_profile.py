#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_bench.sh into profiles/<tag>_*.{md,json}.

    python tools/summarize_profile.py r01                                   # bench.py, k_render_spheres
    python tools/summarize_profile.py r01_final k_render_scene scene_final  # tools/profile_scene.sh output

Reads gpurun_out/prof_<tag>/trace/bench_kernel_stats.csv (kernel-trace --stats) and the
counter_collection.csv of each counter pass; writes a summary the repository keeps.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_render_spheres"


def main():
    global KERNEL
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        KERNEL = sys.argv[2]
    out_name = sys.argv[3] if len(sys.argv) > 3 else "bench_c2"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    stats_rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
    bench, bench_text = None, ""
    if os.path.exists(os.path.join(src, "trace_bench.json")):
        with open(os.path.join(src, "trace_bench.json")) as fp:
            bench = json.loads(fp.read().strip().splitlines()[-1])
    elif os.path.exists(os.path.join(src, "trace_bench.txt")):
        with open(os.path.join(src, "trace_bench.txt")) as fp:
            bench_text = "".join(l for l in fp if "amdgpu.ids" not in l).strip()
    config_line = None  # tools/bench_config.py prints one JSON line
    try:
        config_line = json.loads(bench_text.splitlines()[-1]) if bench_text else None
    except ValueError:
        config_line = None

    counters = collections.OrderedDict()
    launches = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "bench_counter_collection.csv"))):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, vals in per.items():
            counters[name] = sum(vals) / len(vals)  # per launch
            launches[name] = len(vals)

    k = next(r for r in stats_rows if KERNEL in r["Name"])
    avg_ms = float(k["AverageNs"]) / 1e6
    c = counters
    derived = {}
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        derived["shader_clock_GHz"] = cyc / (avg_ms * 1e-3) / 1e9
        derived["valu_wave_instr_per_cycle_per_simd"] = c["SQ_INSTS_VALU"] / 1024.0 / cyc
        derived["cycles_per_valu_wave_instr"] = 1.0 / derived["valu_wave_instr_per_cycle_per_simd"]
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            # active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU counts lane-cycles, SQ_ACTIVE_INST_VALU
            # instruction-cycles, both in the same quad-cycle unit)
            derived["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        if "SQ_WAVE_CYCLES" in c:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in c:
                    derived[n + "_share_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        # gfx950: FETCH_SIZE (KiB) reports half the bytes of wide coalesced reads; here reads are few, scattered
        # and small, so the factor is not applied -- the figure is an upper bound on the order of magnitude only
        derived["hbm_read_MB_per_launch_uncorrected"] = c["FETCH_SIZE"] * 1024 / 1e6
    if "WRITE_SIZE" in c:
        derived["hbm_write_MB_per_launch"] = c["WRITE_SIZE"] * 1024 / 1e6
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        derived["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # share of cycles in which a SIMD issues a VALU instruction, pricing one at the measured ~4.1 cycles
        derived["valu_busy_estimate"] = min(1.0, derived["valu_wave_instr_per_cycle_per_simd"] * 4.1)
    if "SQ_INSTS_VALU_FLOPS_FP32" in c and "SQ_INSTS_VALU_FLOPS_FP64" in c:
        # the counters give flop per lane of every VALU wave-instruction; a wave-instruction occupies all 64 lanes
        flop = 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP32"] + c["SQ_INSTS_VALU_FLOPS_FP64"] + c.get("SQ_INSTS_VALU_FLOPS_FP64_TRANS", 0.0))
        derived["executed_vector_flop_per_launch"] = flop
        derived["executed_vector_TFLOPs"] = flop / (avg_ms * 1e-3) / 1e12
        derived["executed_flop_fp64_share"] = 64.0 * (c["SQ_INSTS_VALU_FLOPS_FP64"] + c.get("SQ_INSTS_VALU_FLOPS_FP64_TRANS", 0.0)) / flop
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
        derived["l1_requests_served_without_l2"] = 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]

    if bench is not None:
        command = "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-pipelined"
        title = "bench.py, One-Weekend final scene"
        script = "tools/profile_bench.sh"
    elif config_line is not None:
        command = f"python3 tools/bench_config.py {config_line['config_name']} {config_line['spp']}"
        title = f"{config_line['workload']}, {config_line['width']}x{config_line['height']}x{config_line['spp']}"
        script = "tools/profile_config.sh"
        if "executed_vector_flop_per_launch" in derived:
            derived["executed_flop_per_ray"] = derived["executed_vector_flop_per_launch"] / config_line["rays"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            derived["hbm_GBs_under_profiler"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / (avg_ms * 1e-3) / 1e9
    else:
        command = "python3 tools/bench_scenes.py <spp> <scene>"
        title = f"tools/bench_scenes.py ({out_name})"
        script = "tools/profile_scene.sh"
    sys.path.insert(0, ROOT)
    from bench import kernel_source_sha256  # the tree the library under the profiler was built from
    summary = {"tag": tag, "command": command + " (under rocprofv3)", "kernel_source_sha256": kernel_source_sha256(),
               "kernel_stats": stats_rows, "bench_line_under_profiler": bench if bench is not None else (config_line if config_line is not None else bench_text),
               "counters_per_launch": counters, "derived": derived}
    with open(os.path.join(out_dir, f"{tag}_{out_name}.json"), "w") as fp:
        json.dump(summary, fp, indent=1)

    lines = [f"# rocprofv3 summary `{tag}`: {title}, 1 x MI355X", "",
             f"Command: `rocprofv3 --kernel-trace --stats -- {command}`",
             f"(counter passes: one `rocprofv3 --pmc ...` run each; see {script}).", "",
             "## Kernel trace (--stats)", "", "| kernel | calls | average ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
    for r in stats_rows:
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.3f} | "
                     f"{float(r['MinNs']) / 1e6:.3f} | {float(r['MaxNs']) / 1e6:.3f} | {r['Percentage']} |")
    if bench is not None:
        lines += ["", f"bench.py's own event-timed average for the same kernel: {bench['roofline']['kernel_ms']:.3f} ms "
                      f"(value {bench['value']:.1f} Msamples/s, roofline fraction {bench['roofline']['frac']:.3f})."]
    else:
        lines += ["", "Output of the profiled command (event-timed kernel time of the library):", "", "```", bench_text, "```"]
    lines += ["", f"## Counters, per launch of `{KERNEL}`", "", "| counter | value |", "|---|---|"]
    for n, v in counters.items():
        lines.append(f"| {n} | {v:.6g} |")
    lines += ["", "## Derived", "", "| quantity | value |", "|---|---|"]
    for n, v in derived.items():
        lines.append(f"| {n} | {v:.4g} |")
    lines.append("")
    with open(os.path.join(out_dir, f"{tag}_{out_name}.md"), "w") as fp:
        fp.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
