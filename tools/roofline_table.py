#!/usr/bin/env python3
"""profiles/rNN_roofline.md from the committed rocprof summaries: every fraction quoted in DESIGN.md / README.md is a row of this table.

    python tools/roofline_table.py r05        (reads profiles/r05_*, writes profiles/r05_roofline.md)

Per workload (headline render per arithmetic mode, DFNet_dm step, DFNet training step N2, NeRF-H training step N1) and kernel:
calls per step, mean duration (rocprofv3 --kernel-trace --stats), time per step, and — where a PMC pass of the same command exists —
the matrix instructions issued per step (SQ_INSTS_MFMA), the FLOP rate they amount to against the nominal dense peak of their
instruction class (2 500 TFLOP/s for v_mfma_f32_32x32x16_f16: 32 768 FLOP each; 157.3 for v_mfma_f32_32x32x2_f32: 4 096 FLOP each),
and the vector instructions per matrix instruction.  "algorithmic" = FLOPs the mathematics needs (unpadded,
2 x MAC, x 3 MFMAs per product in split-f16) for the kernels where that is a closed formula; the MFMA-count rate includes tile padding.
HBM-bound stage kernels: algorithmic bytes per launch over the mean duration against 8 TB/s (from the bench JSON's `hbm` records)."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
F16_PEAK, F32_PEAK = 2500.0, 157.3
RAYS_PER_PASS = 61440
FINE_FLOP = 2.0 * 182720 * 192 * RAYS_PER_PASS          # per launch of the fine kernel (one pass of a 640x480 frame)
COARSE_FLOP = 2.0 * 130944 * 64 * RAYS_PER_PASS


def stats(path):
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(open(path)))


def short(name):
    n = name.replace("void ", "").replace("dfn::", "")
    return (n.split("(")[0])[:78]


def is_f32(name):
    return "PrecF32" in name or "conv_kernel<PrecF32" in name or "gemm_" in name or "wgrad_kernel<" in name and "x3" not in name


def pmc_of(pmc, name):
    """PMC record of a kernel: the summaries key by (truncated) kernel name without arguments."""
    if not pmc:
        return None
    key = short(name)
    for k, v in pmc.items():
        kk = k.replace("dfn::", "").replace("void ", "").replace("(anonymous namespace)::", "")
        if not kk:      # (a name that began with "(anonymous namespace)::" was cut to nothing by the collector: matches no kernel)
            continue
        if kk == key or key.startswith(kk) or kk.startswith(key):
            return v
    return None


def mfma_per_call(rec, per_dispatch=False):
    """Matrix instructions per dispatch of a PMC record.  tools/gpu_pmc.sh and the headline passes store TOTALS over `dispatches`;
    tools/gpu_train_pmc.sh stores per-dispatch AVERAGES (marked `per_dispatch_average`; the unmarked round-5 file is of that kind too:
    `per_dispatch`).  Dividing an average by the dispatch count again was the 8-9x error of the round-5 N1 section."""
    m = rec.get("mfma_insts", rec.get("SQ_INSTS_MFMA", 0.0))
    if rec.get("per_dispatch_average") or per_dispatch:
        return m
    return m / (rec.get("dispatches", 0) or 1)


def step_fraction(rows, steps, pmc, per_dispatch=False, step_ms=None):
    """Matrix instructions issued per step over all kernels with a PMC record x 32 768 FLOP / the step's kernel time (or step_ms)
    against the nominal f16 peak: (MFMA per step, fraction)."""
    tot_m, tot_ns = 0.0, 0.0
    for r in rows:
        tot_ns += float(r["TotalDurationNs"]) / steps
        rec = pmc_of(pmc, r["Name"])
        if rec and not is_f32(r["Name"]):
            tot_m += mfma_per_call(rec, per_dispatch) * int(r["Calls"]) / steps
    t = (step_ms * 1e-3) if step_ms else tot_ns * 1e-9
    return tot_m, tot_m * 32768.0 / t / 1e12 / F16_PEAK


def n1_section(tag):
    """(rows, steps, pmc) of the NeRF-H training step of round `tag`, or None."""
    rows = stats(os.path.join(P, f"{tag}_train_step_kernel_stats.csv"))
    pj = os.path.join(P, f"{tag}_train_step_pmc.json")
    if not rows or not os.path.exists(pj):
        return None
    steps = max(1, int(next((r["Calls"] for r in rows if "train_fwd_chain_kernel<true" in r["Name"]), 1)))
    return rows, steps, json.load(open(pj))


def table(title, rows, steps, pmc, pmc_steps, algo=None, min_share=0.004, per_dispatch=False):
    out = [f"### {title}", "", "| kernel | calls / step | mean us | ms / step | MFMA / step | TFLOP/s (issued) | of nominal | VALU / MFMA | HBM per call: read (2 x FETCH_SIZE) + written, MB -> TB/s | algorithmic |",
           "|---|---|---|---|---|---|---|---|---|---|"]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        if float(r["TotalDurationNs"]) < min_share * tot:
            continue
        name, calls = r["Name"], int(r["Calls"])
        ms = float(r["TotalDurationNs"]) / 1e6 / steps
        rec = pmc_of(pmc, name)
        mf = tf = fr = vm = hb = ""
        if rec and ("fetch_bytes_x2_per_dispatch" in rec or "FETCH_SIZE" in rec):
            rd = rec.get("fetch_bytes_x2_per_dispatch", 2048.0 * rec.get("FETCH_SIZE", 0.0) / (1 if (rec.get("per_dispatch_average") or per_dispatch) else (rec.get("dispatches", 1) or 1)))
            wr = rec.get("write_bytes_per_dispatch", 1024.0 * rec.get("WRITE_SIZE", 0.0) / (1 if (rec.get("per_dispatch_average") or per_dispatch) else (rec.get("dispatches", 1) or 1)))
            hb = f"{rd / 1e6:.1f} + {wr / 1e6:.1f} -> {(rd + wr) / (float(r['AverageNs']) * 1e-9) / 1e12:.2f}"
        if rec:
            m = rec.get("mfma_insts", rec.get("SQ_INSTS_MFMA", 0.0))
            if m:
                per_call = mfma_per_call(rec, per_dispatch)
                per_step = per_call * calls / steps
                flop = 4096.0 if is_f32(name) else 32768.0
                rate = per_step * flop / (ms * 1e-3) / 1e12
                mf, tf, fr = f"{per_step / 1e6:.2f} M", f"{rate:.0f}", f"{rate / (F32_PEAK if is_f32(name) else F16_PEAK):.3f}"
                v = rec.get("valu_per_mfma")
                if v is None and rec.get("SQ_INSTS_VALU"):
                    v = rec["SQ_INSTS_VALU"] / m
                vm = f"{v:.2f}" if v is not None else ""
        al = ""
        if algo:
            for key, fn in algo.items():
                if key in name:
                    al = fn(calls / steps, float(r["AverageNs"]) / 1e3)
        out.append(f"| `{short(name)}` | {calls / steps:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {ms:.3f} | {mf} | {tf} | {fr} | {vm} | {hb} | {al} |")
    out.append(f"| **all kernels** | {sum(int(r['Calls']) for r in rows) / steps:.0f} | | {tot / 1e6 / steps:.3f} | | | | | | |")
    out.append("")
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    L = [f"# Roofline table, round {tag[1:]} — generated by tools/roofline_table.py from profiles/{tag}_*", "",
         "Peaks: dense f16 MFMA 2 500 TFLOP/s, fp32 MFMA 157.3 TFLOP/s, HBM 8 TB/s (MI355X_MICROARCH.md).  Split-f16 issues three f16 MFMAs per",
         "product: its fractions are of the 2 500 TFLOP/s the instructions could reach, i.e. algorithmic TFLOP/s x 3 / 2 500.", ""]
    # ---- headline render, per arithmetic mode (bench.py --steps 3 --warmup 1: four frames of five passes)
    for prec, x, peak in (("f16x3", 3, F16_PEAK), ("f32", 1, F32_PEAK), ("f16", 1, F16_PEAK)):
        rows = stats(os.path.join(P, f"{tag}_kernel_stats_{prec}.csv"))
        if not rows:
            continue
        pj = os.path.join(P, f"{tag}_pmc_summary_{prec}.json")
        pmc = json.load(open(pj)) if os.path.exists(pj) else None
        frames = max(1, int(next((r["Calls"] for r in rows if "nerfh_fine_kernel" in r["Name"]), 20)) // 5)
        algo = {"nerfh_fine_kernel": lambda c, us, x=x, peak=peak: f"{FINE_FLOP / 1e12:.3f} TFLOP / launch -> {FINE_FLOP * x / (us * 1e-6) / 1e12 / peak:.3f} of nominal",
                "nerfh_coarse_kernel": lambda c, us, x=x, peak=peak: f"{COARSE_FLOP / 1e12:.3f} TFLOP / launch -> {COARSE_FLOP * x / (us * 1e-6) / 1e12 / peak:.3f} of nominal"}
        L += table(f"Headline render, 640x480 at 64+128, arithmetic `{prec}` (per frame = step)", rows, frames, pmc, frames, algo)
    bj = os.path.join(P, f"{tag}_bench.json")
    if os.path.exists(bj):
        b = json.load(open(bj))
        L += ["### HBM-bound stage kernels of the headline render (bench.py `hbm` records: algorithmic bytes / mean launch time)", "",
              "| kernel | algorithmic bytes / ray | GB/s | of 8 TB/s |", "|---|---|---|---|"]
        for k, v in (b.get("hbm") or {}).items():
            if isinstance(v, dict) and "achieved_GBps" in v:
                L.append(f"| `{k}` | {v.get('bytes_per_ray', '')} | {v['achieved_GBps']:.0f} | {v.get('frac_of_hbm_peak', v['achieved_GBps'] / 8000.0):.3f} |")
        L.append("")
    # ---- steps
    for title, csvname, pmcname, steps, algo in (
            ("DFNet_dm step (BASELINE configs[4] per-GPU shape: batch 4, 240x320, render 60x80 at 64+128); 25 steps traced", "dm_step_kernel_stats.csv", "dm_step_pmc.json", 25,
             {"nerfh_fine_backward_kernel": lambda c, us: f"1.347 TFLOP x 3 / launch -> {1.347 * 3 / (us * 1e-6) / 1e0 / F16_PEAK:.3f} of nominal",
              "nerfh_fine_kernel": lambda c, us: f"1.347 TFLOP x 3 / launch -> {1.347 * 3 / (us * 1e-6) / F16_PEAK:.3f} of nominal",
              "conv_wgrad_s_kernel<3": lambda c, us: "0.563 TFLOP (f16 x 3) per step over all 3x3 layers; per layer: " + f"{tag}_wgrad_layers.txt"}),
            ("DFNet training step N2 (run_feature.py: 8 siamese + 4 synthesised frames of 240x320); 21 steps traced", "feature_train_kernel_stats.csv", "feature_train_pmc.json", 21, None),
            ("NeRF-H training step N1 (1 536 rays, 64+128)", "train_step_kernel_stats.csv", "train_step_pmc.json", None, None)):
        rows = stats(os.path.join(P, f"{tag}_{csvname}"))
        if not rows:
            continue
        if steps is None:   # N1: one fine forward chain per step (warm-up steps included in the trace)
            steps = max(1, int(next((r["Calls"] for r in rows if "train_fwd_chain_kernel<true" in r["Name"]), 1)))
        pj = os.path.join(P, f"{tag}_{pmcname}")
        pmc = json.load(open(pj)) if os.path.exists(pj) else None
        n1 = "train_step" in csvname
        L += table(title, rows, steps, pmc, steps, algo, per_dispatch=n1)
        if n1 and pmc:
            m, fr = step_fraction(rows, steps, pmc, per_dispatch=True)
            L[-1:] = [f"N1 step as a whole: {m / 1e6:.2f} M matrix instructions issued per step over {sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6:.3f} ms "
                      f"of kernel time = **{fr:.3f} of the nominal f16 peak** (issued; the one-plane weight-gradient stream issues ONE MFMA per "
                      "product, so this sits below bench.py's algorithmic x 3 convention — `f16_mfma_issued_frac_of_nominal` in the bench line is "
                      "this number from the byte / MFMA model).", ""]
            for k, v in pmc.items():
                if "FETCH_SIZE" in v and "wgrad_stream" in k:
                    us = next((float(r["AverageNs"]) / 1e3 for r in rows if short(r["Name"]).startswith(k.replace("dfn::", "")[:40])), None)
                    if us:
                        rd = 2.0 * v["FETCH_SIZE"] * 1024.0
                        L[-1:] = [f"`{k}`: reads 2 x FETCH_SIZE = {rd / 1e9:.2f} GB per dispatch in {us:.0f} us = {rd / (us * 1e-6) / 1e12:.2f} TB/s "
                                  f"({rd / (us * 1e-6) / 8e12:.2f} of 8 TB/s).", ""]
    wl = os.path.join(P, f"{tag}_wgrad_layers.txt")
    if os.path.exists(wl):
        L += ["### Split-storage conv weight gradient, layer by layer, stand-alone (tools/gpu_wgrad_layers.py)", "", "```", open(wl).read().rstrip(), "```", ""]
    dl = os.path.join(P, f"{tag}_dfnet_layers.txt")
    if os.path.exists(dl):
        L += ["### DFNet forward, 4 x 480x640, kernel by kernel (tools/gpu_dfnet_layers.py)", "", "```", open(dl).read().rstrip(), "```", ""]
    path = os.path.join(P, f"{tag}_roofline.md")
    open(path, "w").write("\n".join(L) + "\n")
    print(path, len(L), "lines")


if __name__ == "__main__":
    main()
